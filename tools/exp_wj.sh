#!/bin/bash
# experiment (GPU box): the wave-per-job launch under library variants and slab sizes, one step at a time.
# tools/exp_wj.sh <variant> ...  (variants: lamsa_amd/lib/var/liblamsa_hp_<v>.so; "base" = the product library; LAMSA_HP_WJ_SLAB_KB: bytes of a wave's ordinary slab)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; v=$2; shift 2
  lib=$PWD/lamsa_amd/lib/var/liblamsa_hp_$v.so; [ $v = base ] && lib=$PWD/lamsa_amd/lib/liblamsa_hp.so
  env LAMSA_HP_LIB=$lib "$@" timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --bare --sequential > gpurun_out/expw_$tag.json 2> gpurun_out/expw_$tag.err
  python3 - gpurun_out/expw_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); lm = d["launch_ms"]
    print(sys.argv[2], "ms/step", d["ms_per_step"], "bad", d["reads_not_ok"], "chain1 %.1f" % lm["chain1"], "list %.1f" % lm["list1_within_fill1"], "wave %.1f" % lm["wave_dp1_within_fill1"],
          "lane %.1f" % (lm["dp1_within_fill1"] - lm["list1_within_fill1"] - lm["wave_dp1_within_fill1"]), "k_fill %.1f" % (lm["fill1"] - lm["dp1_within_fill1"]))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for v in "$@"; do run $v $v A=1; done
