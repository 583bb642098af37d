#!/usr/bin/env python3
"""Summarise one tools/prof_bench.sh run (gpurun_out/prof/<tag>/) into the two files kept under profiles/:
    <out>_kernel_stats.csv   the rocprofv3 --kernel-trace --stats table, as written by rocprofv3
    <out>_pmc.json           per-dispatch averages of every counter over the k_align_batch dispatches, plus the
                             HBM traffic derived from FETCH_SIZE / WRITE_SIZE as MI355X_MICROARCH.md prescribes
usage: tools/summarize_prof.py gpurun_out/prof/<tag> profiles/<name> "<command line that produced it>" [bench.json]
"""
import csv
import glob
import json
import os
import shutil
import sys

KERNEL = "k_align_batch"


def main():
    src, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    bench = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else None
    stats = glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True)
    if not stats:
        sys.exit("no kernel_stats.csv under %s/stats" % src)
    shutil.copy(stats[0], out + "_kernel_stats.csv")
    avg_ms = calls = None
    for row in csv.DictReader(open(stats[0])):
        if row["Name"].startswith(KERNEL):
            avg_ms = float(row["AverageNs"]) / 1e6; calls = int(row["Calls"])
    per = {}
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            if not row["Kernel_Name"].startswith(KERNEL):
                continue
            acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for name, by_disp in acc.items():
            per[name] = sum(by_disp.values()) / len(by_disp)
    doc = {"command": cmd, "kernel": KERNEL, "dispatches_per_pass": calls, "kernel_avg_ms_rocprof": round(avg_ms, 3) if avg_ms else None}
    if bench:
        doc["kernel_avg_ms_bench_hip_events"] = bench["roofline"]["kernel_ms"]
        doc["algorithmic_bytes_per_dispatch"] = bench["roofline"]["algorithmic_bytes_per_launch"]
        doc["reads_per_step"] = bench["config"]["reads_per_step_per_gpu"]
        doc["workload"] = bench["config"]["workload"].split(":")[0]
    doc["per_dispatch"] = per
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        doc["hbm_traffic_note"] = ("FETCH_SIZE/WRITE_SIZE are in KiB. Per MI355X_MICROARCH.md FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950; "
                                   "this kernel mixes narrow gathers and coalesced record loads, so lower (as reported) and upper (reads x2) bounds are given; "
                                   "`traffic` in bench.py's roofline uses the upper bound.")
        doc["hbm_bytes_per_dispatch_lower"] = (per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024
        doc["hbm_bytes_per_dispatch_upper"] = (2 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024
    json.dump(doc, open(out + "_pmc.json", "w"), indent=1)
    print(json.dumps({k: doc[k] for k in doc if k != "per_dispatch"}, indent=1))


if __name__ == "__main__":
    main()
