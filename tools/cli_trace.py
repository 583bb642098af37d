#!/usr/bin/env python3
"""Where the ingest of the product binary spends its time (run on the GPU box; no GPU work): simulated reads + GEM map text as files, then
`lamsa aln --parse-only` with LAMSA_TRACE=1 -- per chunk the sequential scan, the parse on all threads, the merge.
usage: tools/cli_trace.py [n_reads] [threads]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import simbatch   # noqa: E402
import simfiles   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
threads = int(sys.argv[2]) if len(sys.argv) > 2 else min(os.cpu_count() or 8, 32)
d = tempfile.mkdtemp(prefix="clit_", dir=os.environ.get("TMPDIR", "/tmp"))
t = time.time(); ref = simbatch.SimRef(1_000_000_000, n_contigs=24, seed=5, threads=min(threads, 16))
B = simbatch.SimBatch(ref, n, 10000, "ont2d", seed=31, threads=min(threads, 16))
simfiles.write_index(d + "/ref.fa", ref); simfiles.write_reads(d + "/reads.fa", B, workers=min(threads, 32))
print("files written in %.1f s; %d threads; %d cpus visible, affinity %d" % (time.time() - t, threads, os.cpu_count(), len(os.sched_getaffinity(0))), flush=True)
exe = os.path.join(ROOT, "lamsa_amd", "bin", "lamsa")
for rep in range(1):
    for th in (threads,):
        p = subprocess.run([exe, "aln", "-N", "-T", "ont2d", "-R", "0", "-t", str(th), "--batch", "16384", "--parse-only", "-o", d + "/out.sam", d + "/ref.fa", d + "/reads.fa"],
                           capture_output=True, text=True, env=dict(os.environ, LAMSA_TRACE="1"))
        print("-t %d:" % th)
        print("\n".join(l for l in p.stderr.splitlines() if l.startswith("[scan]") or l.startswith("[prepare]") or "wall" in l), flush=True)
# a whole run from the hit stream: the stages behind the GPU ([write]: result streams -> records, ranking + SAM text, the file)
subprocess.run([exe, "aln", "-N", "-T", "ont2d", "-R", "0", "-t", str(threads), "--batch", "16384", "--parse-only", "--save-hits", d + "/hits.bin", "-o", d + "/out.sam", d + "/ref.fa", d + "/reads.fa"], capture_output=True, text=True)
for rep in range(2):
    p = subprocess.run([exe, "aln", "-T", "ont2d", "-R", "0", "-t", str(threads), "--batch", "16384", "--hits", d + "/hits.bin", "-o", d + "/out.sam", d + "/ref.fa", d + "/reads.fa"],
                       capture_output=True, text=True, env=dict(os.environ, LAMSA_TRACE="1"))
    print("--hits, whole run:")
    print("\n".join(l for l in p.stderr.splitlines() if l.startswith("[write]") or l.startswith("[prepare]") or "wall" in l), flush=True)
