#!/bin/bash
# Profile one bench configuration with rocprofv3 (run on the GPU box through gpurun):
#   tools/prof_bench.sh <tag> <bench args...>
# writes gpurun_out/prof/<tag>/{stats,pmc1,pmc2,...}; copy the summaries you want judged into profiles/.
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof/$tag
mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# one launch at a time (--sequential): the timed region of a plain run queues steps two deep, and the trace would show every kernel stretched by the one it shares the GPU with
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py "$@" --sequential --bare > $out/bench_stats.log 2>&1
# PMC_SETS="FETCH_SIZE;WRITE_SIZE" limits the counter passes (default: all five)
if [ -n "${PMC_SETS:-}" ]; then IFS=';' read -ra SETS <<< "$PMC_SETS"; else
SETS=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"
      "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
      "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
      "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES"); fi
for set in "${SETS[@]}"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  # counter passes serialise the dispatches anyway: one launch at a time (--sequential)
  rocprofv3 --pmc $set --output-format csv -d $out/pmc_$n -- python3 bench.py "$@" --sequential --bare > $out/bench_pmc_$n.log 2>&1
  echo "pmc pass $n done" 
done
find $out -name "*.csv" | head -50 > $out/files.txt
