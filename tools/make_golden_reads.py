"""Whole-path golden fixtures: simulated reads -> real GEM seeding -> reference SAM.

Shared index: tests/golden/ref/ref.fa.{ann,amb,pac,bwt,sa}.  For every scenario: tests/golden/<name>/{reads.fa.gz,
reads.fa.seed.gem.map.gz,args.txt,golden_R0.sam.gz[,golden_full.sam.gz if it differs]}.  `golden_R0.sam` is the
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
    ("c1_perfect", [], 100, 5000, "perfect", {}),
    ("c2_pacbio", ["-T", "pacbio"], 100, 5000, "pacbio", {}),
    ("c3_ont", ["-T", "ont2d"], 100, 10000, "ont", {}),
    ("c4_pb20k", ["-T", "pacbio", "-w", "200"], 50, 20000, "pb20k", {}),
    ("c5_sv", ["-V", "10000"], 100, 10000, "lowerr", {"sv_frac": 0.67}),
    ("c6_edge", ["-T", "ont2d"], 100, 2500, "ont", {"n_frac": 0.01, "edge": True}),
    # stage 4 (BWT rescue): reads carrying 40-320 bp pieces of other loci between their flanks; default -R
    ("c7_rescue", [], 100, 4000, "lowerr", {"rescue": True}),
    ("c8_rescue_ont", ["-T", "ont2d"], 100, 6000, "ont", {"rescue": True}),
    # rearranged reads: an inversion, a tandem duplication or a foreign tail each (the inter-line, DUP and tail-region branches of the path)
    ("c9_rearr", [], 120, 6000, "lowerr", {"rearr": True}),
    ("c10_rearr_ont", ["-T", "ont2d"], 80, 6000, "ont", {"rearr": True}),
]


def rearranged_reads(rng, contigs, n, length, sub, ins, dele, copies=()):
    """Reads of contig 0, one rearrangement each: the middle 300-900 bases inverted; 100-1500 bases at the middle duplicated in tandem;
    the last 100-600 bases replaced by random sequence; or 150-360 bases of the other contig at one end.  Independent errors; half of the reads reverse-complemented."""
    reads = []
    for k in range(n):
        p = int(rng.integers(10000, 220000)); base = contigs[0][p:p + length].copy()
        kind = ("inv", "dup", "tail", "ftail")[k % 4]
        mid = length // 2 + int(rng.integers(-800, 800))
        if kind == "inv":
            w = int(rng.integers(300, 900)); base[mid:mid + w] = simdata.COMP[base[mid:mid + w][::-1]]
        elif kind == "dup":
            w = int(rng.integers(100, 1500)); base = np.concatenate([base[:mid + w], base[mid:mid + w], base[mid + w:]])[:length + w]
        elif kind == "tail":
            w = int(rng.integers(100, 600)); base[length - w:] = rng.integers(0, 4, w, dtype=np.uint8)
        else:              # two or three seeds' worth of another locus at one end: a tiny line of its own at the edge of the read
            w = int(rng.integers(150, 360)); q = int(rng.integers(5000, 120000)); piece = contigs[1][q:q + w]
            kb = [c for c in copies if c[0] == 1000]
            if k % 16 >= 8 and kb:          # ... out of a planted 1-kbp repeat: several tiny lines over the same read bases, one per copy
                w = int(rng.integers(160, 245))                                        # (two seeds' worth in the 100-bp-step mode)
                _, ci, off = kb[int(rng.integers(0, len(kb)))]
                o2 = int(rng.integers(0, 1000 - w)); piece = contigs[ci][off + o2:off + o2 + w]
            if k % 8 < 4:
                base[length - w:] = piece
            else:
                base[:w] = piece
        r = simdata.mutate(rng, base, sub, ins, dele)
        if rng.random() < 0.5:
            r = simdata.COMP[r[::-1]]
        reads.append(("%s_%d_w%d" % (kind, k, w), np.ascontiguousarray(r, dtype=np.uint8)))
    return reads


def rescue_reads(rng, contigs, n, length, sub, ins, dele):
    """Reads of contig 0 with one to three short foreign pieces (other contig or 50 kbp away, either strand) spliced in or
    replacing as many bases; independent errors everywhere; half of the reads reverse-complemented."""
    reads = []
    for k in range(n):
        p = int(rng.integers(20000, 200000)); base = contigs[0][p:p + length].copy()
        pieces, cur = [], 0
        for m in sorted(rng.integers(300, length - 300, int(rng.integers(1, 4))).tolist()):
            if m <= cur + 200:
                continue
            w = int(rng.integers(40, 320)); q = int(rng.integers(5000, 120000))
            piece = contigs[1][q:q + w].copy() if rng.random() < 0.7 else contigs[0][(p + 50000) % 200000:(p + 50000) % 200000 + w].copy()
            if rng.random() < 0.4:
                piece = simdata.COMP[piece[::-1]]
            pieces += [base[cur:m], piece]; cur = m + (w if rng.random() < 0.5 else 0)
        pieces.append(base[cur:])
        r, out, e = np.concatenate(pieces), [], None
        e = rng.random(len(r))
        for i, c in enumerate(r):
            if e[i] < sub:
                out.append((int(c) + 1 + int(rng.integers(0, 3))) % 4)
            elif e[i] < sub + ins:
                out += [int(c), int(rng.integers(0, 4))]
            elif e[i] >= sub + ins + dele:
                out.append(int(c))
        r = np.array(out, dtype=np.uint8)
        if rng.random() < 0.5:
            r = simdata.COMP[r[::-1]]
        reads.append(("rescue_%d" % k, r))
    return reads


def run(cmd, **kw):
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, **kw)


def make_reads(only=None):
    """only: names of the scenarios to (re)generate; default all."""
    rng = np.random.default_rng(REF["seed"])
    copies = []
    contigs = simdata.make_reference(rng, REF["contigs"], REF["repeats"], copies=copies)
    tmp = tempfile.mkdtemp(prefix="lamsa_gold_")
    ref = os.path.join(tmp, "ref.fa")
    simdata.write_fasta(ref, [("chr%d" % (i + 1), c) for i, c in enumerate(contigs)])
    run([LAMSA, "index", ref])
    for si, (name, args, n, length, prof, extra) in enumerate(SCEN):
        if only and name not in only:
            continue
        d = os.path.join(GOLD, name)
        os.makedirs(d, exist_ok=True)
        rng = np.random.default_rng(1000 + si)
        sub, ins, dele = simdata.PROFILES[prof]
        if extra.get("rescue"):
            reads = rescue_reads(rng, contigs, n, length, sub, ins, dele)
        elif extra.get("rearr"):
            reads = rearranged_reads(rng, contigs, n, length, sub, ins, dele, copies)
        else:
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
        for ext in (".ann", ".amb", ".pac", ".bwt", ".sa"):      # .bwt / .sa: the FM index stage 4 searches
            shutil.copy(ref + ext, os.path.join(GOLD, "ref", "ref.fa" + ext))
        for old in ("reads.fa", "golden_R0.sam", "golden_full.sam", "golden_full.sam.gz"):
            if os.path.exists(os.path.join(d, old)):
                os.remove(os.path.join(d, old))
        for src, dst in ((rd, "reads.fa.gz"), (rd + ".seed.gem.map", "reads.fa.seed.gem.map.gz")):
            with open(src, "rb") as f, gzip.GzipFile(os.path.join(d, dst), "wb", mtime=0) as g:
                g.write(f.read())
        body = {}
        for kind in ("full", "R0"):
            with open(os.path.join(tmp, "%s.%s.sam" % (name, kind))) as f:
                body[kind] = [l for l in f if not l.startswith("@PG")]
        with gzip.GzipFile(os.path.join(d, "golden_R0.sam.gz"), "wb", mtime=0) as g:
            g.write("".join(body["R0"]).encode())
        if body["full"] != body["R0"]:          # stage 4 (BWT rescue) changed the output: keep the default run too
            with gzip.GzipFile(os.path.join(d, "golden_full.sam.gz"), "wb", mtime=0) as g:
                g.write("".join(body["full"]).encode())
        with open(os.path.join(d, "args.txt"), "w") as f:
            f.write(" ".join(args) + "\n")
        sz = sum(os.path.getsize(os.path.join(d, x)) for x in os.listdir(d))
        print(name, "reads", len(reads), "bytes", sz)
    shutil.rmtree(tmp)


if __name__ == "__main__":
    make_reads(set(sys.argv[1:]) or None)
