#!/bin/bash
# Bench build variants of the library side by side (run on the GPU box): tools/var_bench.sh <tag> <bench args...>
# Every lamsa_amd/lib/var/lib_*.so plus the default library; one bare bench line per variant in gpurun_out/<tag>_var.txt.
set -o pipefail
tag=$1; shift
out=gpurun_out/${tag}_var.txt
: > $out
for lib in default lamsa_amd/lib/var/lib_*.so; do
  if [ "$lib" = default ]; then unset LAMSA_HP_LIB; else export LAMSA_HP_LIB=$PWD/$lib; fi
  echo "== $lib" >> $out
  LAMSA_NO_BUILD=1 timeout -k 10 300 python3 bench.py "$@" --bare 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print(json.dumps({'reads_per_s': d.get('reads_per_s'), 'seq': d.get('reads_per_s_one_step_at_a_time_rank0'), 'ms': d.get('ms_per_step'), 'bad': d.get('reads_not_ok'), 'launch_ms': d.get('launch_ms')}))
" >> $out || { echo "variant failed: stopping" >> $out; break; }
done
cat $out
