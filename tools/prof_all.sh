#!/bin/bash
# Profile passes of the non-default workloads (run on the GPU box): kernel trace + the traffic / instruction / issue-port counters, one
# rocprofv3 pass per set, summarised into profiles/<tag>_<workload>_{pmc.json,kernel_stats.csv}.   tools/prof_all.sh <tag> [workloads...]
tag=$1; shift
ws=${@:-sv10k pb5k mol5k pb20k}
SETS="FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR;GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU"
for w in $ws; do
  extra=""; [ $w = pb20k ] && extra="--reads 16384"
  PMC_SETS="$SETS" bash tools/prof_bench.sh ${tag}_$w --workload $w $extra --steps 2 --warmup 1 > gpurun_out/${tag}_${w}_prof.log 2>&1
  grep -h "^{" gpurun_out/prof/${tag}_$w/bench_stats.log | tail -1 > /tmp/bj_$w.json
  python3 tools/summarize_prof.py gpurun_out/prof/${tag}_$w profiles/${tag}_$w "bench.py --workload $w $extra --steps 2 --warmup 1 --sequential --bare (tools/prof_all.sh)" /tmp/bj_$w.json > /dev/null 2>&1
  cp profiles/${tag}_${w}_pmc.json profiles/${tag}_${w}_kernel_stats.csv gpurun_out/ 2>/dev/null
  python3 - profiles/${tag}_${w}_pmc.json $w <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["kernels"].items():
    r = v.get("round1") or v
    print("%-6s %-15s %8.3f ms  wait %s valu_busy %s salu_busy %s  hbm GB/step %.1f-%.1f" % (sys.argv[2], k, r.get("avg_ms_rocprof", 0), r.get("wait_any_frac"), r.get("valu_busy_frac"), r.get("salu_busy_frac"), v.get("hbm_bytes_per_step_lower", 0) / 1e9, v.get("hbm_bytes_per_step_upper", 0) / 1e9))
PY
done
