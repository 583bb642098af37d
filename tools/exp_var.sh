#!/bin/bash
# experiment (GPU box): a workload under library variants, one step at a time.   tools/exp_var.sh <workload> <reads> <variant> ...
# (variants: lamsa_amd/lib/var/liblamsa_hp_<v>.so; "base" = the product library)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
wl=$1; n=$2; shift 2
for v in "$@"; do
  lib=$PWD/lamsa_amd/lib/var/liblamsa_hp_$v.so; [ $v = base ] && lib=$PWD/lamsa_amd/lib/liblamsa_hp.so
  LAMSA_HP_LIB=$lib timeout -k 10 400 python3 bench.py --workload $wl --reads $n --steps 3 --warmup 1 --bare --sequential > gpurun_out/expv_${wl}_$v.json 2> gpurun_out/expv_${wl}_$v.err
  python3 - gpurun_out/expv_${wl}_$v.json $wl $v <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); lm = d["launch_ms"]
    print(sys.argv[2], sys.argv[3], "reads/s", d["reads_per_s"], "ms/step", d["ms_per_step"], "bad", d["reads_not_ok"], "chain1 %.1f list %.1f wave %.1f lane %.1f fill %.1f chain2 %.1f" % (lm["chain1"], lm["list1_within_fill1"], lm["wave_dp1_within_fill1"], lm["dp1_within_fill1"] - lm["list1_within_fill1"] - lm["wave_dp1_within_fill1"], lm["fill1"] - lm["dp1_within_fill1"], lm["chain2"]))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "failed", e)
PY
done
