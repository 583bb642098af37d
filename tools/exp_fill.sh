#!/bin/bash
# experiment (GPU box): occupancy variants of one kernel, one step at a time.  tools/exp_fill.sh <variant> ...   (lamsa_amd/lib/var/liblamsa_hp_<variant>.so; "base" = the product library)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  lib=$PWD/lamsa_amd/lib/var/liblamsa_hp_$v.so; [ $v = base ] && lib=$PWD/lamsa_amd/lib/liblamsa_hp.so
  LAMSA_HP_LIB=$lib timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --bare --sequential > gpurun_out/expf_$v.json 2> gpurun_out/expf_$v.err
  python3 - gpurun_out/expf_$v.json $v <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); lm = d["launch_ms"]
print(sys.argv[2], "ms/step", d["ms_per_step"], "bad", d["reads_not_ok"], "chain1 %.1f" % lm["chain1"], "list %.1f" % lm["list1_within_fill1"], "wave %.1f" % lm["wave_dp1_within_fill1"],
      "lane %.1f" % (lm["dp1_within_fill1"] - lm["list1_within_fill1"] - lm["wave_dp1_within_fill1"]), "k_fill %.1f" % (lm["fill1"] - lm["dp1_within_fill1"]))
PY
done
