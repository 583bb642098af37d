#!/bin/bash
# The round's bench lines (run on the GPU box): the default workload with every leg, the other workloads with short CPU legs.
tag=$1
python3 bench.py --steps 8 --warmup 1 > gpurun_out/${tag}_bench_ont10k.json 2> gpurun_out/${tag}_bench_ont10k.err || exit 1
for w in sv10k pb5k; do python3 bench.py --workload $w --steps 6 --warmup 1 --stream-chunks 8 --cpu-seconds 8 > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err || exit 1; done
python3 bench.py --workload pb20k --reads 16384 --steps 6 --warmup 1 --stream-chunks 8 --cpu-seconds 8 > gpurun_out/${tag}_bench_pb20k.json 2> gpurun_out/${tag}_bench_pb20k.err || exit 1
python3 bench.py --workload mol5k --steps 6 --warmup 1 --stream-chunks 8 --cpu-seconds 8 > gpurun_out/${tag}_bench_mol5k.json 2> gpurun_out/${tag}_bench_mol5k.err || exit 1
echo "bench_all done"
