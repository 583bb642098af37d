#!/bin/bash
# rocprofv3 kernel stats + counter passes of the non-default workloads (run on the GPU box): tools/prof_others.sh <tag> [reads per step of the 20-kbp workload = 32768]
tag=$1; pb20=${2:-32768}
cd $GRAFT_REPO_ROOT
for w in sv10k pb5k mol5k pb20k; do
  extra=""; [ $w = pb20k ] && extra="--reads $pb20"
  bash tools/prof_bench.sh ${tag}_$w --workload $w $extra --steps 2 --warmup 1 > gpurun_out/prof_${tag}_$w.log 2>&1 || { echo "$w failed"; tail -3 gpurun_out/prof_${tag}_$w.log; exit 1; }
  python3 - profiles/r04_bench_$w.json /tmp/bench_$w.json <<'PY'
import sys
l = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]; open(sys.argv[2], "w").write(l)
PY
  python3 tools/summarize_prof.py gpurun_out/prof/${tag}_$w gpurun_out/r04_$w "bench.py --workload $w $extra --steps 2 --warmup 1 --sequential --bare (tools/prof_bench.sh)" /tmp/bench_$w.json > /dev/null 2>&1 && echo "$w summarised"
done
