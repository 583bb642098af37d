#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference itself.

Runs only in the build container (needs oracle/_ref, i.e. /root/reference compiled by
`make -C oracle ref`, and for the whole-path fixtures the bundled GEM binaries).  The
fixtures are DATA: inputs + the reference's outputs.  No reference source is stored.

  python tools/make_golden.py dp        -> tests/golden/dp_vectors.npz
  python tools/make_golden.py reads     -> tests/golden/<scenario>/...   (whole-path SAM goldens)
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

DP_CASES = [  # (read_type, seed, n, max_len, (sub, ins, del))
    ("default", 1, 120, 260, (0.01, 0.01, 0.01)),
    ("pacbio", 2, 120, 420, (0.015, 0.09, 0.045)),
    ("ont2d", 3, 120, 420, (0.04, 0.04, 0.04)),
]


def dp_calls(P):
    return [(0, P.band_w, 0), (0, 5, 0), (1, P.band_w, 50), (1, P.band_w, P.hash_len * P.match), (2, 0, 100), (2, 0, P.hash_len * P.match)]


def make_dp():
    import dpjobs
    import reflib
    out = {}
    for rt, seed, n, mx, err in DP_CASES:
        jobs = dpjobs.make_jobs(seed, n, mx, err)
        P = reflib.ref_para(rt)
        out[rt + "_q"] = np.concatenate([j[0] for j in jobs]); out[rt + "_qlen"] = np.array([len(j[0]) for j in jobs], np.int32)
        out[rt + "_t"] = np.concatenate([j[1] for j in jobs]); out[rt + "_tlen"] = np.array([len(j[1]) for j in jobs], np.int32)
        for ci, (kind, w, h0) in enumerate(dp_calls(P)):
            r = reflib.ref_dp(jobs, P, kind, w, h0)
            k = "%s_c%d_" % (rt, ci)
            out[k + "kwh"] = np.array([kind, w, h0], np.int32)
            out[k + "score"] = r["score"]; out[k + "qle"] = r["qle"]; out[k + "tle"] = r["tle"]
            out[k + "cign"] = np.array([len(c) for c in r["cigars"]], np.int32)
            out[k + "cig"] = np.array([x for c in r["cigars"] for x in c], np.int32)
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, "dp_vectors.npz"), **out)
    print("wrote dp_vectors.npz", os.path.getsize(os.path.join(GOLD, "dp_vectors.npz")))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("dp", "all"):
        make_dp()
    if what in ("reads", "all"):
        from make_golden_reads import make_reads
        make_reads()
