#!/usr/bin/env python3
"""Summarise one tools/prof_bench.sh run (gpurun_out/prof/<tag>/) into the two files kept under profiles/:
    <out>_kernel_stats.csv   the rocprofv3 --kernel-trace --stats table, as written by rocprofv3
    <out>_pmc.json           per kernel of the hot path: rocprofv3's average duration and the per-dispatch average of every
                             counter, the HBM traffic derived from FETCH_SIZE / WRITE_SIZE as MI355X_MICROARCH.md prescribes
                             (FETCH_SIZE in KiB; as reported = lower bound, reads doubled for the gfx950 under-count of wide
                             coalesced reads = upper bound), and the issue / wait shares of wave cycles
usage: tools/summarize_prof.py gpurun_out/prof/<tag> profiles/<name> "<command line that produced it>" [bench.json]
"""
import csv
import glob
import json
import os
import shutil
import sys

KERNELS = ("k_chain1", "k_filllist", "k_filldp_small", "k_filldp_wave", "k_fill", "k_chain2", "k_publish", "k_align_batch", "k_dp_batch")


def kname(s):
    import re
    m = re.match(r"(?:void )?(?:hp::)?(k_[A-Za-z0-9_]+)", s.strip())
    return m.group(1) if m and m.group(1) in KERNELS else None


def main():
    src, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    bench = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else None
    stats = glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True)
    if not stats:
        sys.exit("no kernel_stats.csv under %s/stats" % src)
    shutil.copy(stats[0], out + "_kernel_stats.csv")
    kern = {}
    for row in csv.DictReader(open(stats[0])):
        k = kname(row["Name"])
        if k:
            kern[k] = {"calls": int(row["Calls"]), "avg_ms_rocprof": round(float(row["AverageNs"]) / 1e6, 3), "total_ms_rocprof": round(float(row["TotalDurationNs"]) / 1e6, 3)}
    # The fill launches run twice per step (round 1, round 2) and round 2 is nearly empty on the bench data: an average over both says
    # little about either.  The dispatches of such a kernel are told apart by their order (round 1 first): `round1` holds the same figures
    # for the first of every pair.
    TWICE = ("k_fill", "k_filllist", "k_filldp_small", "k_filldp_wave")
    for f in glob.glob(os.path.join(src, "stats", "**", "*_kernel_trace.csv"), recursive=True):
        by = {}
        for row in csv.DictReader(open(f)):
            k = kname(row["Kernel_Name"])
            if k in TWICE:
                by.setdefault(k, []).append((int(row["Dispatch_Id"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6))
        for k, v in by.items():
            v.sort()
            first = [ms for i, (_, ms) in enumerate(v) if i % 2 == 0]
            if k in kern and first and len(v) % 2 == 0:
                kern[k].setdefault("round1", {})["avg_ms_rocprof"] = round(sum(first) / len(first), 3)
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            k = kname(row["Kernel_Name"])
            if not k:
                continue
            acc.setdefault((k, row["Counter_Name"]), {}).setdefault(row["Dispatch_Id"], 0.0)
            acc[(k, row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for (k, name), by_disp in acc.items():
            kern.setdefault(k, {}).setdefault("per_dispatch", {})[name] = sum(by_disp.values()) / len(by_disp)
            kern[k].setdefault("pmc_dispatches", {})[name] = len(by_disp)
            if k in TWICE and len(by_disp) % 2 == 0:
                ids = sorted(by_disp, key=int)
                first = [by_disp[i] for n_, i in enumerate(ids) if n_ % 2 == 0]
                kern[k].setdefault("round1", {}).setdefault("per_dispatch", {})[name] = sum(first) / len(first)
    for k, d0 in list(kern.items()):
      for d in ([d0, d0["round1"]] if "round1" in d0 else [d0]):
          per = d.get("per_dispatch", {})
          if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
              d["hbm_bytes_per_dispatch_lower"] = (per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024
              d["hbm_bytes_per_dispatch_upper"] = (2 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024
          wc = per.get("SQ_WAVE_CYCLES")
          if wc:
              for c, key in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_ACTIVE_INST_ANY", "issuing_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_frac")):
                  if c in per:
                      d[key] = round(per[c] / wc, 4)
          if "TCC_HIT_sum" in per and "TCC_MISS_sum" in per:
              d["l2_hit_rate"] = round(per["TCC_HIT_sum"] / max(1.0, per["TCC_HIT_sum"] + per["TCC_MISS_sum"]), 4)
          if per.get("GRBM_GUI_ACTIVE"):
              # GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md): cycles of the dispatch = / 8; 1024 SIMDs, a VALU or scalar instruction
              # holds its issue port of a SIMD for 4 cycles
              cyc = per["GRBM_GUI_ACTIVE"] / 8.0
              if "SQ_ACTIVE_INST_VALU" in per:
                  d["valu_busy_frac"] = round(per["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, 4)
              if "SQ_INST_CYCLES_SALU" in per:
                  d["salu_busy_frac"] = round(per["SQ_INST_CYCLES_SALU"] * 4 / 1024 / cyc, 4)
          if "SQ_INSTS_VALU" in per and d.get("avg_ms_rocprof"):
              # vector instructions issued per second against what 1024 SIMDs can issue (one wave64 VALU instruction per 2 cycles at 2.4 GHz)
              d["valu_issue_frac"] = round(per["SQ_INSTS_VALU"] / (d["avg_ms_rocprof"] * 1e-3) / (1024 * 2.4e9 / 2), 4)
    doc = {"command": cmd, "kernels": kern,
           "hbm_traffic_note": "FETCH_SIZE/WRITE_SIZE are in KiB.  Per MI355X_MICROARCH.md FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950; these kernels mix "
                               "narrow gathers and coalesced loads, so a lower (as reported) and an upper (reads x2) bound are given; bench.py's roofline.traffic uses the upper one."}
    if bench:
        doc["reads_per_step"] = bench["config"]["reads_per_step_per_gpu"]
        doc["workload"] = bench["config"]["workload"].split(":")[0]
        doc["algorithmic_bytes_per_step"] = bench["roofline"].get("algorithmic_bytes_per_step", bench["roofline"].get("algorithmic_bytes_per_launch"))
        doc["launch_ms_bench_hip_events"] = bench.get("launch_ms")
    # one step of the main pass = chain1, [filllist, filldp, fill] (round 1), chain2, [filllist, filldp, fill] (round 2), publish: traffic of a step
    main = [k for k in ("k_chain1", "k_filllist", "k_filldp_small", "k_filldp_wave", "k_fill", "k_chain2", "k_publish") if k in kern and "hbm_bytes_per_dispatch_upper" in kern[k]]
    steps = kern.get("k_chain1", {}).get("calls") or 1
    for k in main:
        mult = kern[k]["calls"] / steps                      # dispatches of this kernel per step
        kern[k]["dispatches_per_step"] = round(mult, 3)
        kern[k]["hbm_bytes_per_step_lower"] = kern[k]["hbm_bytes_per_dispatch_lower"] * mult
        kern[k]["hbm_bytes_per_step_upper"] = kern[k]["hbm_bytes_per_dispatch_upper"] * mult
        kern[k]["ms_per_step_rocprof"] = round(kern[k]["avg_ms_rocprof"] * mult, 3)
    if main:
        doc["hbm_bytes_per_step_lower"] = sum(kern[k]["hbm_bytes_per_step_lower"] for k in main)
        doc["hbm_bytes_per_step_upper"] = sum(kern[k]["hbm_bytes_per_step_upper"] for k in main)
    json.dump(doc, open(out + "_pmc.json", "w"), indent=1)
    brief = {k: {x: v for x, v in d.items() if x not in ("per_dispatch", "pmc_dispatches")} for k, d in kern.items()}
    print(json.dumps({"kernels": brief, **{k: doc[k] for k in doc if k.startswith("hbm_bytes")}}, indent=1))


if __name__ == "__main__":
    main()
