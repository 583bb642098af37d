#!/bin/bash
# experiment: the step with the wave-wide extensions (and global alignments) of the fill replaced by stubs -- wrong results, timing only
for v in ${VARIANTS:-noext nodp}; do
  LAMSA_HP_LIB=$PWD/lamsa_amd/lib/lib_$v.so timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --sequential --bare > gpurun_out/exp_$v.json 2> gpurun_out/exp_$v.err
  python3 - gpurun_out/exp_$v.json $v <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], json.dumps({"ms_per_step": d["ms_per_step"], "bad": d["reads_not_ok"], "launch_ms": d["launch_ms"]}))
PY
done
