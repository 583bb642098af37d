#!/bin/bash
# experiment (GPU box): how many waves per CU each launch gets when two batches are in flight (LAMSA_HP_*_PER_CU override slab_plan's caps;
# LAMSA_HP_FULL_GRIDS=1: no caps)
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
V=$PWD/lamsa_amd/lib/var
SEQ=
run prio0 LAMSA_HP_LIB=$V/liblamsa_hp_prio0.so
run prio3 LAMSA_HP_LIB=$V/liblamsa_hp_prio3.so
run prio3_full LAMSA_HP_LIB=$V/liblamsa_hp_prio3.so LAMSA_HP_FULL_GRIDS=1
run prio3_c8_f16_w32 LAMSA_HP_LIB=$V/liblamsa_hp_prio3.so LAMSA_HP_WJ_PER_CU=32
run prio3_c12_f16_w8 LAMSA_HP_LIB=$V/liblamsa_hp_prio3.so LAMSA_HP_CHAIN_PER_CU=12 LAMSA_HP_WJ_PER_CU=8
run prio3_c8_f8_w16 LAMSA_HP_LIB=$V/liblamsa_hp_prio3.so LAMSA_HP_FILL_PER_CU=8
SEQ=--sequential
run prio3_seq LAMSA_HP_LIB=$V/liblamsa_hp_prio3.so
