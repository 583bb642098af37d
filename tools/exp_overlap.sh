#!/bin/bash
# experiment (GPU box): grids capped below a CU's capacity so that the launches of the two batches in flight share the CUs; occupancy variants of k_filldp_wave
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --steps 6 --warmup 1 --bare > gpurun_out/exp_$tag.json 2> gpurun_out/exp_$tag.err; python3 - gpurun_out/exp_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    lm = d["launch_ms"]
    print(sys.argv[2], "reads/s", d["reads_per_s"], "ms/step", d["ms_per_step"], "bad", d["reads_not_ok"], "chain1 %.1f fill1 %.1f (list %.1f wave %.1f)" % (lm["chain1"], lm["fill1"], lm["list1_within_fill1"], lm["wave_dp1_within_fill1"]))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run base A=1
run half LAMSA_HP_CHAIN_PER_CU=8 LAMSA_HP_FILL_PER_CU=16 LAMSA_HP_WJ_PER_CU=8
run c8w8 LAMSA_HP_CHAIN_PER_CU=8 LAMSA_HP_WJ_PER_CU=8
run c8f24w8 LAMSA_HP_CHAIN_PER_CU=8 LAMSA_HP_FILL_PER_CU=24 LAMSA_HP_WJ_PER_CU=8
run c12w12 LAMSA_HP_CHAIN_PER_CU=12 LAMSA_HP_WJ_PER_CU=12
run w8 LAMSA_HP_LIB=$PWD/lamsa_amd/lib/var/liblamsa_hp_w8.so
run w6 LAMSA_HP_LIB=$PWD/lamsa_amd/lib/var/liblamsa_hp_w6.so
run w8half LAMSA_HP_LIB=$PWD/lamsa_amd/lib/var/liblamsa_hp_w8.so LAMSA_HP_CHAIN_PER_CU=8 LAMSA_HP_WJ_PER_CU=16
