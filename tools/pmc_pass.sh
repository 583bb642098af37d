#!/bin/bash
# One rocprofv3 --pmc pass per counter group over a bench configuration (run on the GPU box through gpurun):
#   tools/pmc_pass.sh <tag> "<counters of group 1>" ["<group 2>" ...] -- <bench args...>
# writes gpurun_out/prof/<tag>/pmc_<i>/ and prints per-counter sums for k_align_batch.
set -u
tag=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof/$tag
mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for set in "${groups[@]}"; do
  i=$((i+1))
  # counter collection serialises the dispatches anyway: one launch at a time (--sequential)
  rocprofv3 --pmc $set --output-format csv -d $out/pmc_$i -- python3 bench.py "$@" --sequential --bare > $out/bench_pmc_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/bench_pmc_$i.log; exit 1; }
  python3 - "$out/pmc_$i" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); nd = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_align_batch" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); nd[row["Counter_Name"]].add(row["Dispatch_Id"])
for k in sorted(tot):
    print("%-40s %18.0f per dispatch (%d dispatches)" % (k, tot[k] / max(1, len(nd[k])), len(nd[k])))
PY
done
