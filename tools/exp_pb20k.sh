cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests/test_path_gpu.py tests/test_dp_gpu.py -x -q 2>&1 | tail -2
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --bare --sequential > gpurun_out/r4k_seq.json 2> gpurun_out/r4k_seq.err
timeout -k 10 400 python3 bench.py --workload pb20k --reads 16384 --steps 4 --warmup 1 --bare > gpurun_out/r4k_pb20_16k.json 2> gpurun_out/r4k_pb20_16k.err
timeout -k 10 500 python3 bench.py --workload pb20k --reads 32768 --steps 4 --warmup 1 --bare > gpurun_out/r4k_pb20_32k.json 2> gpurun_out/r4k_pb20_32k.err
for f in r4k_seq r4k_pb20_16k r4k_pb20_32k; do python3 - gpurun_out/$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); lm = d["launch_ms"]
    print(sys.argv[1], "reads/s", d["reads_per_s"], "ms/step", d["ms_per_step"], "bad", d["reads_not_ok"], "chain1 %.1f list %.1f wave %.1f lane %.1f fill %.1f drains %.1f %.1f" % (lm["chain1"], lm["list1_within_fill1"], lm["wave_dp1_within_fill1"], lm["dp1_within_fill1"] - lm["list1_within_fill1"] - lm["wave_dp1_within_fill1"], lm["fill1"] - lm["dp1_within_fill1"], lm["drain_chain1"], lm["drain_fill1"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
