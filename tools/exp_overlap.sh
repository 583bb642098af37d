#!/bin/bash
# experiment (GPU box): grids capped below a CU's capacity so that the launches of the two batches in flight share the CUs; occupancy variants
# of k_filldp_wave (lamsa_amd/lib/var/liblamsa_hp_w<waves per SIMD>.so: make EXTRA="-DHP_WJ_WAVES_PER_SIMD=8 -DHP_WJ_LDS_WORDS=1280" OUT=...)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --steps 6 --warmup 1 --bare $SEQ > gpurun_out/exp_$tag.json 2> gpurun_out/exp_$tag.err; python3 - gpurun_out/exp_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    lm = d["launch_ms"]
    print(sys.argv[2], "reads/s", d["reads_per_s"], "ms/step", d["ms_per_step"], "bad", d["reads_not_ok"], "chain1 %.1f fill1 %.1f (list %.1f wave %.1f lane %.1f fill %.1f)" % (lm["chain1"], lm["fill1"], lm["list1_within_fill1"], lm["wave_dp1_within_fill1"], lm["dp1_within_fill1"] - lm["list1_within_fill1"] - lm["wave_dp1_within_fill1"], lm["fill1"] - lm["dp1_within_fill1"]))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
W8=LAMSA_HP_LIB=$PWD/lamsa_amd/lib/var/liblamsa_hp_w8.so
W6=LAMSA_HP_LIB=$PWD/lamsa_amd/lib/var/liblamsa_hp_w6.so
timeout -k 10 400 python3 -m pytest tests/test_path_gpu.py -x -q 2>&1 | tail -2
SEQ=--sequential
run seq_base A=1
SEQ=
run base A=1
run w8_c8_f16_w16 $W8 LAMSA_HP_CHAIN_PER_CU=8 LAMSA_HP_FILL_PER_CU=16 LAMSA_HP_WJ_PER_CU=16
run w8_c8_f16_w20 $W8 LAMSA_HP_CHAIN_PER_CU=8 LAMSA_HP_FILL_PER_CU=16 LAMSA_HP_WJ_PER_CU=20
run w8_c10_f16_w16 $W8 LAMSA_HP_CHAIN_PER_CU=10 LAMSA_HP_FILL_PER_CU=16 LAMSA_HP_WJ_PER_CU=16
run w8_c8_f12_w16 $W8 LAMSA_HP_CHAIN_PER_CU=8 LAMSA_HP_FILL_PER_CU=12 LAMSA_HP_WJ_PER_CU=16
run w8_c6_f16_w16 $W8 LAMSA_HP_CHAIN_PER_CU=6 LAMSA_HP_FILL_PER_CU=16 LAMSA_HP_WJ_PER_CU=16
run w6_c8_f16_w12 $W6 LAMSA_HP_CHAIN_PER_CU=8 LAMSA_HP_FILL_PER_CU=16 LAMSA_HP_WJ_PER_CU=12
