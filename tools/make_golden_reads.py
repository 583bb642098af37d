"""Whole-path golden fixtures: simulated reads -> real GEM seeding -> reference SAM.

Shared index: tests/golden/ref/ref.fa.{ann,amb,pac}.  For every scenario: tests/golden/<name>/{reads.fa,
reads.fa.seed.gem.map.gz,args.txt,golden_R0.sam[,golden_full.sam if it differs]}.  `golden_R0.sam` is the
reference run with `-N -I -R 0` (stage 4, the BWT rescue, disabled from the command line),
`golden_full.sam` the default run.  Inputs and outputs only -- nothing of the reference's code.
"""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import simdata  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
LAMSA = os.path.join(ROOT, "oracle", "_ref", "lamsa")

REF = dict(seed=23, contigs=[260000, 140000], repeats=[(300, 260), (1000, 60), (6000, 12)])
SCEN = [  # name, args, n_reads, length, profile, extra
    ("c1_perfect", [], 16, 5000, "perfect", {}),
    ("c2_pacbio", ["-T", "pacbio"], 10, 5000, "pacbio", {}),
    ("c3_ont", ["-T", "ont2d"], 6, 10000, "ont", {}),
    ("c4_pb20k", ["-T", "pacbio", "-w", "200"], 2, 20000, "pb20k", {}),
    ("c5_sv", ["-V", "10000"], 10, 10000, "lowerr", {"sv_frac": 0.67}),
    ("c6_edge", ["-T", "ont2d"], 6, 2500, "ont", {"n_frac": 0.01, "edge": True}),
]


def run(cmd, **kw):
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, **kw)


def make_reads():
    rng = np.random.default_rng(REF["seed"])
    contigs = simdata.make_reference(rng, REF["contigs"], REF["repeats"])
    tmp = tempfile.mkdtemp(prefix="lamsa_gold_")
    ref = os.path.join(tmp, "ref.fa")
    simdata.write_fasta(ref, [("chr%d" % (i + 1), c) for i, c in enumerate(contigs)])
    run([LAMSA, "index", ref])
    for si, (name, args, n, length, prof, extra) in enumerate(SCEN):
        d = os.path.join(GOLD, name)
        os.makedirs(d, exist_ok=True)
        rng = np.random.default_rng(1000 + si)
        sub, ins, dele = simdata.PROFILES[prof]
        reads = simdata.simulate_reads(rng, contigs, n, length, sub, ins, dele, extra.get("sv_frac", 0.0), n_frac=extra.get("n_frac", 0.0))
        if extra.get("edge"):
            reads.append(("short_read", rng.integers(0, 4, 30, dtype=np.uint8)))                 # shorter than a seed
            reads.append(("random_read", rng.integers(0, 4, 3000, dtype=np.uint8)))               # no seed hits
            reads.append(("all_n", np.full(600, 4, np.uint8)))
            chim = np.concatenate([contigs[0][5000:7000], simdata.COMP[contigs[1][9000:11000][::-1]]])   # chimeric / inversion-like
            reads.append(("chimera", chim))
        rd = os.path.join(tmp, name + ".fa")
        simdata.write_fasta(rd, reads, width=0)
        run([LAMSA, "aln"] + args + ["-t", "1", ref, rd, "-o", os.path.join(tmp, name + ".full.sam")])
        run([LAMSA, "aln"] + args + ["-t", "1", "-N", "-I", "-R", "0", ref, rd, "-o", os.path.join(tmp, name + ".R0.sam")])
        os.makedirs(os.path.join(GOLD, "ref"), exist_ok=True)
        for ext in (".ann", ".amb", ".pac"):
            shutil.copy(ref + ext, os.path.join(GOLD, "ref", "ref.fa" + ext))
        shutil.copy(rd, os.path.join(d, "reads.fa"))
        with open(rd + ".seed.gem.map", "rb") as f, gzip.GzipFile(os.path.join(d, "reads.fa.seed.gem.map.gz"), "wb", mtime=0) as g:
            g.write(f.read())
        body = {}
        for kind in ("full", "R0"):
            with open(os.path.join(tmp, "%s.%s.sam" % (name, kind))) as f:
                body[kind] = [l for l in f if not l.startswith("@PG")]
        with open(os.path.join(d, "golden_R0.sam"), "w") as g:
            g.writelines(body["R0"])
        if body["full"] != body["R0"]:          # stage 4 (BWT rescue) changed the output: keep the default run too
            with open(os.path.join(d, "golden_full.sam"), "w") as g:
                g.writelines(body["full"])
        with open(os.path.join(d, "args.txt"), "w") as f:
            f.write(" ".join(args) + "\n")
        sz = sum(os.path.getsize(os.path.join(d, x)) for x in os.listdir(d))
        print(name, "reads", len(reads), "bytes", sz)
    shutil.rmtree(tmp)


if __name__ == "__main__":
    make_reads()
