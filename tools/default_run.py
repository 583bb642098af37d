#!/usr/bin/env python3
"""The reference's DEFAULT run at scale (run on the GPU box): `lamsa aln` WITHOUT -R 0 -- stage 4, the BWT rescue of unaligned read parts
(src/bwt_aln.c:398-409, on by default: src/lamsa_aln.c:1304) -- of the compiled reference (oracle/_ref/lamsa) against the product binary's,
per bench workload, on a stand-in whose FM index the product's own `lamsa index --from-pac` builds on the spot (bench.default_run_check).
usage: tools/default_run.py [reads per workload = 2000] [reference bases = 300000000] [workload ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
bp = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000_000
names = sys.argv[3:] or ["ont10k", "pb5k", "sv10k", "mol5k", "pb20k"]
threads = os.cpu_count() or 8
for w in names:
    k = n if w != "pb20k" else max(200, n // 4)
    r = bench.default_run_check(w, bench.WORKLOADS[w], min(threads, 64), k, bp)
    print(json.dumps(r), flush=True)
