#!/bin/bash
# bench lines of the non-default workloads (run on the GPU box): tools/bench_others.sh <tag>
tag=$1
for w in sv10k pb5k mol5k; do python3 bench.py --workload $w --steps 6 --warmup 1 --stream-chunks 8 --cpu-seconds 8 > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err || exit 1; done
python3 bench.py --workload pb20k --reads 16384 --steps 6 --warmup 1 --stream-chunks 8 --cpu-seconds 8 > gpurun_out/${tag}_bench_pb20k.json 2> gpurun_out/${tag}_bench_pb20k.err || exit 1
echo "bench_others done"
