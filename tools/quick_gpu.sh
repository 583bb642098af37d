#!/bin/bash
# One development round on the GPU box (through gpurun): the whole-path GPU parity tests, a bare sequential bench line of the default
# workload, and -- when lamsa_amd/lib/liblamsa_hp_prof.so (make EXTRA=-DHP_PROF OUT=../lib/liblamsa_hp_prof.so) travelled -- the per-phase
# cycle counters of the same step.   tools/quick_gpu.sh <tag> [pytest -k expression]
set -o pipefail
tag=$1; kexpr=${2:-}
mkdir -p gpurun_out
if [ -n "$kexpr" ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "$kexpr" > gpurun_out/${tag}_pytest.log 2>&1 || { tail -20 gpurun_out/${tag}_pytest.log; exit 1; }
  tail -2 gpurun_out/${tag}_pytest.log
fi
timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --sequential --bare > gpurun_out/${tag}_seq.json 2> gpurun_out/${tag}_seq.err || { tail -5 gpurun_out/${tag}_seq.err; exit 1; }
python3 - gpurun_out/${tag}_seq.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(json.dumps({"reads_per_s": d["reads_per_s"], "ms_per_step": d["ms_per_step"], "bad": d["reads_not_ok"], "launch_ms": d["launch_ms"]}))
PY
if [ -f lamsa_amd/lib/liblamsa_hp_prof.so ]; then
  LAMSA_HP_LIB=$PWD/lamsa_amd/lib/liblamsa_hp_prof.so timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --sequential --bare > gpurun_out/${tag}_prof.json 2> gpurun_out/${tag}_prof.err
  grep "HP_PROF" gpurun_out/${tag}_prof.err | head -24
fi
