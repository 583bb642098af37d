#!/bin/bash
# One bare bench line per setting of an environment variable (run on the GPU box): tools/env_bench.sh <tag> <VAR> "<v1 v2 ...>" <bench args...>
tag=$1; var=$2; vals=$3; shift 3
out=gpurun_out/${tag}_env.txt; : > $out
for v in $vals; do
  echo "== $var=$v" >> $out
  env $var=$v LAMSA_NO_BUILD=1 timeout -k 10 300 python3 bench.py "$@" --bare 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print(json.dumps({'reads_per_s': d.get('reads_per_s'), 'ms': d.get('ms_per_step'), 'bad': d.get('reads_not_ok')}))
" >> $out
done
cat $out
