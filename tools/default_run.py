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
import subprocess   # noqa: E402
import tempfile     # noqa: E402
import time         # noqa: E402
import simbatch     # noqa: E402
import simfiles     # noqa: E402
simbatch.build()
d = tempfile.mkdtemp(prefix="lamsa_dflt_", dir=os.environ.get("TMPDIR", "/tmp"))
ref = simbatch.SimRef(bp, n_contigs=12, seed=17, threads=min(threads, 32))
simfiles.write_index(d + "/ref.fa", ref)
t = time.time()
q = subprocess.run([os.path.join(ROOT, "lamsa_amd", "bin", "lamsa"), "index", "--from-pac", d + "/ref.fa"], capture_output=True, text=True, env=dict(os.environ, LAMSA_INDEX_THREADS=str(min(threads, 64)), LAMSA_TRACE="1"))
t_index = time.time() - t
print("# stand-in of %d bp, `lamsa index --from-pac` on %d threads: rc %d, %.1f s  %s" % (int(ref.l_pac), min(threads, 64), q.returncode, t_index, " | ".join(l for l in q.stderr.splitlines() if l.startswith("[index]"))), flush=True)
for w in names:
    k = n if w != "pb20k" else max(200, n // 4)
    r = bench.default_run_check(w, bench.WORKLOADS[w], min(threads, 64), k, bp, keep=d, prebuilt=(ref, round(t_index, 1)))
    print(json.dumps(r), flush=True)
    if w in ("ont10k", "pb5k"):     # once more with work for stage 4: three reads in ten have 15 % of their seeds, in the middle, without a hit
        r = bench.default_run_check(w, bench.WORKLOADS[w], min(threads, 64), k, bp, keep=d, prebuilt=(ref, round(t_index, 1)), unseeded=(0.3, 0.15))
        print(json.dumps(r), flush=True)
import shutil       # noqa: E402
shutil.rmtree(d, ignore_errors=True)
