#!/usr/bin/env python3
"""Time the product binary end to end on files (run on the GPU box): simulated reads + hits written as the files
`lamsa aln -N` reads (FASTA, GEM map text, .pac/.ann), then `lamsa_amd/bin/lamsa aln -N ...` on them, page cache warm.
Prints the binary's own stage accounting and reads/s.  usage: tools/cli_bench.py [n_reads] [read_len] [ref_bp] [reads per chunk] [threads]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import simbatch   # noqa: E402
import simfiles   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
ref_bp = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000_000
threads = int(sys.argv[5]) if len(sys.argv) > 5 else (os.cpu_count() or 8)      # (-t of the runs is min(threads, 32); a one-GPU box grants 16 cores)
d = tempfile.mkdtemp(prefix="clib_", dir=os.environ.get("TMPDIR", "/tmp"))
t = time.time(); ref = simbatch.SimRef(ref_bp, n_contigs=24, seed=5, threads=min(threads, 16))
B = simbatch.SimBatch(ref, n, L, "ont2d", seed=31, threads=min(threads, 16))
simfiles.write_index(d + "/ref.fa", ref); simfiles.write_reads(d + "/reads.fa", B, workers=min(threads, 32))
print("files written in %.1f s: %d reads, %.0f hits/read, map %.1f MB" % (time.time() - t, n, B.n_hits / n, os.path.getsize(d + "/reads.fa.seed.gem.map") / 1e6), flush=True)
batch = sys.argv[4] if len(sys.argv) > 4 else "16384"
for rep, extra in enumerate((["--parse-only"], ["--save-hits", d + "/hits.bin"], [], ["--hits", d + "/hits.bin", "--parse-only"], ["--hits", d + "/hits.bin"], ["--hits", d + "/hits.bin"])):
    t = time.time()
    p = subprocess.run([os.path.join(ROOT, "lamsa_amd", "bin", "lamsa"), "aln", "-N", "-T", "ont2d", "-R", "0", "-t", str(min(threads, 32)), "--batch", batch, "-o", d + "/out.sam"] + extra + [d + "/ref.fa", d + "/reads.fa"],
                       capture_output=True, text=True)
    dt = time.time() - t
    print("run %d %s: rc %d, %.2f s wall -> %.0f reads/s end to end" % (rep, " ".join(extra), p.returncode, dt, n / dt))
    print("\n".join(l for l in p.stderr.splitlines() if "wall" in l or "Mapping done" in l or "failed" in l), flush=True)
    if rep == 4:                # once more with the stage trace of every chunk
        q = subprocess.run([os.path.join(ROOT, "lamsa_amd", "bin", "lamsa"), "aln", "-N", "-T", "ont2d", "-R", "0", "-t", str(min(threads, 32)), "--batch", batch, "-o", d + "/out.sam"] + extra + [d + "/ref.fa", d + "/reads.fa"],
                           capture_output=True, text=True, env=dict(os.environ, LAMSA_TRACE="1"))
        print("\n".join(l for l in q.stderr.splitlines() if l.startswith("[write]") or l.startswith("[prepare]")), flush=True)
# --shard i/N: every shard is a process with its own parser (meant to run one per GPU; here one after the other on this GPU, since two
# processes sharing one GPU time-slice its persistent grids and its memory: measured 5-10x slower, see profiles/r03_cli_bench.txt);
# their outputs concatenated are out.sam
exe = os.path.join(ROOT, "lamsa_amd", "bin", "lamsa")
for n_sh, src in ((2, ["--hits", d + "/hits.bin"]), (2, [])):
    dts, rcs = [], []
    for i in range(n_sh):
        t = time.time()
        q = subprocess.run([exe, "aln", "-N", "-T", "ont2d", "-R", "0", "-t", str(min(threads, 32)), "--batch", batch, "--shard", "%d/%d" % (i, n_sh), "-o", d + "/out.%d.sam" % i] + src + [d + "/ref.fa", d + "/reads.fa"],
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        dts.append(time.time() - t); rcs.append(q.returncode)
        if q.returncode:
            print(q.stderr[-400:])
    same = not any(rcs) and subprocess.run("cat %s/out.0.sam %s/out.1.sam | grep -v ^@PG | cmp - <(grep -v ^@PG %s/out.sam)" % (d, d, d), shell=True, executable="/bin/bash").returncode == 0
    print("%d shards %s, one after the other: rc %s, %s s wall each -> %.0f reads/s if they ran on %d GPUs; concatenated == unsharded SAM: %s" % (
        n_sh, " ".join(src) or "(GEM map text)", rcs, ", ".join("%.2f" % x for x in dts), n / max(dts), n_sh, same), flush=True)
# Eight shards at once on the host alone (--parse-only makes no GPU call): what the host side of an 8-GPU node has to keep up with.  This box
# grants one GPU's share of the cores (16), so every shard gets an eighth of it -- a lower bound for a node that brings 16 cores per GPU.
for src, what in ((["--hits", d + "/hits.bin"], "hit stream"), ([], "GEM map text")):
    per = max(1, min(threads, 32) // 8)
    t = time.time()
    ps = [subprocess.Popen([exe, "aln", "-N", "-T", "ont2d", "-R", "0", "-t", str(per), "--batch", batch, "--shard", "%d/8" % i, "--parse-only"] + src + [d + "/ref.fa", d + "/reads.fa"],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for i in range(8)]
    rcs = [q.wait() for q in ps]
    dt = time.time() - t
    print("8 shards --parse-only at once (%s), %d threads each on %d cores: rc %s, %.2f s wall -> %.0f reads/s ingested in aggregate" % (what, per, threads, rcs, dt, n / dt), flush=True)
# does the time to reserve the device buffers depend on a process that has just released its own?
time.sleep(20)
t = time.time()
p = subprocess.run([exe, "aln", "-N", "-T", "ont2d", "-R", "0", "-t", str(min(threads, 32)), "--batch", batch, "-o", d + "/out2.sam", "--hits", d + "/hits.bin", d + "/ref.fa", d + "/reads.fa"], capture_output=True, text=True)
print("after 20 idle seconds: rc %d, %.2f s wall" % (p.returncode, time.time() - t))
print("\n".join(l for l in p.stderr.splitlines() if "wall" in l))
print("SAM %.1f MB" % (os.path.getsize(d + "/out.sam") / 1e6))
