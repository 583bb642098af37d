#!/bin/bash
# Kernel times and the HBM-traffic / instruction counters of one bench configuration, printed per kernel (run on the GPU box through gpurun):
#   tools/pmc_quick.sh <tag> [bench args]      -> gpurun_out/<tag>_pmc.json, gpurun_out/<tag>_kernel_stats.csv
set -o pipefail
tag=$1; shift
PMC_SETS="${PMC_SETS:-FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU}" bash tools/prof_bench.sh $tag --steps 2 --warmup 1 --stream-chunks 0 "$@" > gpurun_out/${tag}_prof.log 2>&1
python3 tools/summarize_prof.py gpurun_out/prof/$tag gpurun_out/$tag "bench.py --steps 2 --warmup 1 --stream-chunks 0 $* --bare (tools/pmc_quick.sh)" > /dev/null 2>&1
python3 - gpurun_out/${tag}_pmc.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d['kernels'].items():
    p = v.get('per_dispatch', {})
    print("%-15s %8.3f ms  fetch %6.1f GB  write %6.1f GB  wait %s issue %s  vmem rd %.3g wr %.3g  valu %.3g salu %.3g" % (k, v.get('avg_ms_rocprof', 0), p.get('FETCH_SIZE', 0) * 1024 / 1e9, p.get('WRITE_SIZE', 0) * 1024 / 1e9,
          v.get('wait_any_frac'), v.get('issuing_frac'), p.get('SQ_INSTS_VMEM_RD', 0), p.get('SQ_INSTS_VMEM_WR', 0), p.get('SQ_INSTS_VALU', 0), p.get('SQ_INSTS_SALU', 0)))
PY
