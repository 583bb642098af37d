#!/bin/bash
# second half of tools/bench_all.sh: the 20-kbp and the 1 %-error workloads, with the 20-kbp profile passes
tag=$1
python3 bench.py --workload pb20k --reads 16384 --steps 6 --warmup 1 --stream-chunks 8 --cpu-seconds 8 > gpurun_out/${tag}_bench_pb20k.json 2> gpurun_out/${tag}_bench_pb20k.err || exit 1
PMC_SETS="FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" bash tools/prof_bench.sh r02u_pb20k --workload pb20k --reads 16384 --steps 2 --warmup 1 > gpurun_out/${tag}_prof.log 2>&1
python3 bench.py --workload mol5k --steps 6 --warmup 1 --stream-chunks 8 --cpu-seconds 8 > gpurun_out/${tag}_bench_mol5k.json 2> gpurun_out/${tag}_bench_mol5k.err || exit 1
echo "bench_rest done"
