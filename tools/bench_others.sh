#!/bin/bash
# bench lines of the non-default workloads (run on the GPU box): tools/bench_others.sh <tag> [reads per step of the 20-kbp workload = 32768]
tag=$1; pb20=${2:-32768}
for w in sv10k pb5k mol5k; do python3 bench.py --workload $w --steps 6 --warmup 1 --stream-chunks 8 --cpu-seconds 8 --default-run-reads 384 > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err || exit 1; done
python3 bench.py --workload pb20k --reads $pb20 --steps 6 --warmup 1 --stream-chunks 4 --cpu-seconds 8 --default-run-reads 128 > gpurun_out/${tag}_bench_pb20k.json 2> gpurun_out/${tag}_bench_pb20k.err || exit 1
python3 - gpurun_out/${tag}_bench_sv10k.json gpurun_out/${tag}_bench_pb5k.json gpurun_out/${tag}_bench_mol5k.json gpurun_out/${tag}_bench_pb20k.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1]); lm = d["roofline"]["launch_ms_one_step_at_a_time"] or d["launch_ms"]
    rb = (d["cpu_baseline"] or {}).get("reference_binary") or {}
    print(f.split("_bench_")[1][:-5], "reads/s", d["reads_per_s"], "bad", d["reads_not_ok"], "streamed", d["pcie_inclusive_streamed_reads_per_s"]["pageable_host_arrays"],
          "| one step: chain1 %.1f list %.1f wave %.1f lane %.1f fill %.1f drain_fill %.1f" % (lm["chain1"], lm["list1_within_fill1"], lm["wave_dp1_within_fill1"], lm["dp1_within_fill1"] - lm["list1_within_fill1"] - lm["wave_dp1_within_fill1"], lm["fill1"] - lm["dp1_within_fill1"], lm["drain_fill1"]),
          "| ref", rb.get("gpu_equals_reference_on_sample"), "default run", rb.get("gpu_equals_reference_on_sample_default_run"))
PY
echo "bench_others done"
