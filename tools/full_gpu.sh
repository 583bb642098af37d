#!/bin/bash
# One full check on the GPU box (through gpurun): every -m gpu test, then the default bench line as the driver runs it.   tools/full_gpu.sh <tag>
tag=${1:-full}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; tail -3 gpurun_out/${tag}_pytest.log
( time timeout -k 10 600 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err ) 2>&1 | grep real
python3 - gpurun_out/${tag}_bench.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value", d["value"], "reads/s", d["reads_per_s"], "ms/step", d["ms_per_step"], "bad", d["reads_not_ok"])
print("launch_ms", d["launch_ms"])
print("one step at a time", d["roofline"]["launch_ms_one_step_at_a_time"])
print("roofline", {k: d["roofline"][k] for k in ("kernel", "kernel_ms", "achieved", "frac", "traffic", "gcups", "gcups_within_fill_launches", "valu_lane_slots_per_cell")})
print("pcie", d["pcie_inclusive_streamed_reads_per_s"], d["pcie_inclusive_reads_per_s"])
c = d["cpu_baseline"]; print("cpu", c.get("reads_per_s"), c.get("gpu_equals_cpu_on_sample")); print("reference", c.get("reference_binary"))
PY
