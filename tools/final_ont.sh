#!/bin/bash
# the round's final numbers of the default workload: profile passes first (bench.py reads the committed summary for roofline.traffic;
# here the fresh one is summarised on the box before the bench line is taken), then the full bench line
tag=$1
bash tools/prof_bench.sh ${tag}_ont10k --steps 2 --warmup 1 > gpurun_out/${tag}_prof.log 2>&1
grep -h "^{" gpurun_out/prof/${tag}_ont10k/bench_stats.log | tail -1 > /tmp/bj.json
python3 tools/summarize_prof.py gpurun_out/prof/${tag}_ont10k profiles/r03_ont10k "bench.py --steps 2 --warmup 1 --sequential --bare (tools/final_ont.sh -> tools/prof_bench.sh: one rocprofv3 pass per counter set)" /tmp/bj.json > /dev/null 2>&1
cp profiles/r03_ont10k_pmc.json profiles/r03_ont10k_kernel_stats.csv gpurun_out/ 2>/dev/null
python3 bench.py --steps 8 --warmup 1 > gpurun_out/${tag}_bench_ont10k.json 2> gpurun_out/${tag}_bench_ont10k.err
echo "final_ont done"
