bash tools/prof_bench.sh r02u_ont10k --steps 2 --warmup 1 > gpurun_out/r02u_prof.log 2>&1
PMC_SETS="FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" bash tools/prof_bench.sh r02u_sv10k --workload sv10k --steps 2 --warmup 1 >> gpurun_out/r02u_prof.log 2>&1
PMC_SETS="FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" bash tools/prof_bench.sh r02u_pb5k --workload pb5k --steps 2 --warmup 1 >> gpurun_out/r02u_prof.log 2>&1
PMC_SETS="FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" bash tools/prof_bench.sh r02u_pb20k --workload pb20k --reads 16384 --steps 2 --warmup 1 >> gpurun_out/r02u_prof.log 2>&1
tail -3 gpurun_out/r02u_prof.log
