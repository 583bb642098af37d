#!/usr/bin/env python3
"""Synthetic reference / long-read simulator (SURVEY.md section 8d recipe).

Everything is seeded, so fixtures can be regenerated bit-for-bit.  Used by
tools/make_golden.py (parity fixtures, run through the real reference + GEM in
the build container) and by bench.py (bench inputs, generated on the GPU box).

Reference: uniform ACGT contigs, optionally with planted repeat families
(each copy `div` diverged by substitutions) -- the GRCh37 stand-in needs
repeats, otherwise seed-hit chaining cost is hidden.
Reads: uniform start, 50% reverse strand, independent per-base sub/ins/del
errors, optional single SV (deletion or novel insertion) at the midpoint.
"""
import argparse
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.array([3, 2, 1, 0], dtype=np.uint8)


def make_reference(rng, contig_lens, repeats=(), div=0.05, copies=None):
    """Return list of uint8 arrays (values 0..3), one per contig.

    repeats: iterable of (unit_len, n_copies).  copies: a list that receives (unit_len, contig, offset) of every planted copy."""
    contigs = [rng.integers(0, 4, size=n, dtype=np.uint8) for n in contig_lens]
    total = sum(contig_lens)
    bounds = np.cumsum([0] + list(contig_lens))
    for unit_len, n_copies in repeats:
        unit = rng.integers(0, 4, size=unit_len, dtype=np.uint8)
        for _ in range(n_copies):
            g = int(rng.integers(0, total - unit_len))
            ci = int(np.searchsorted(bounds, g, side="right") - 1)
            off = g - bounds[ci]
            if off + unit_len > contig_lens[ci]:
                continue
            cp = unit.copy()
            nmut = rng.binomial(unit_len, div)
            if nmut:
                pos = rng.integers(0, unit_len, size=nmut)
                cp[pos] = (cp[pos] + rng.integers(1, 4, size=nmut)) & 3
            if rng.random() < 0.5:
                cp = COMP[cp[::-1]]
            contigs[ci][off:off + unit_len] = cp
            if copies is not None:
                copies.append((unit_len, ci, int(off)))
    return contigs


def mutate(rng, seq, sub, ins, dele):
    """Apply independent per-base errors; returns the mutated sequence."""
    n = len(seq)
    r = rng.random(n)
    out = []
    keep = r >= dele                       # deletion of the reference base
    is_sub = (r >= dele) & (r < dele + sub)
    s = seq.copy()
    k = int(is_sub.sum())
    if k:
        s[is_sub] = (s[is_sub] + rng.integers(1, 4, size=k)) & 3
    is_ins = rng.random(n) < ins           # insert one random base after position
    pieces = np.empty(n * 2, dtype=np.int16)
    pieces[0::2] = np.where(keep, s.astype(np.int16), np.int16(-1))
    pieces[1::2] = np.where(is_ins, rng.integers(0, 4, size=n).astype(np.int16), np.int16(-1))
    res = pieces[pieces >= 0].astype(np.uint8)
    return res


def simulate_reads(rng, contigs, n_reads, length, sub, ins, dele, sv_frac=0.0,
                   sv_del=(1000, 10000), sv_ins=(1000, 5000), n_frac=0.0):
    """Yield (name, uint8 seq 0..4)."""
    lens = np.array([len(c) for c in contigs])
    reads = []
    for i in range(n_reads):
        while True:
            ci = int(rng.integers(0, len(contigs)))
            if lens[ci] > length + 12000:
                break
        span = length
        sv = None
        if rng.random() < sv_frac:
            if rng.random() < 0.5:
                sv = ("D", int(rng.integers(sv_del[0], sv_del[1] + 1)))
                span = length + sv[1]
            else:
                sv = ("I", int(rng.integers(sv_ins[0], sv_ins[1] + 1)))
                span = max(length - sv[1], 200)
        pos = int(rng.integers(0, lens[ci] - span))
        frag = contigs[ci][pos:pos + span]
        if sv is not None:
            mid = span // 2
            if sv[0] == "D":
                frag = np.concatenate([frag[:mid - sv[1] // 2], frag[mid - sv[1] // 2 + sv[1]:]])
            else:
                frag = np.concatenate([frag[:mid], rng.integers(0, 4, size=sv[1], dtype=np.uint8), frag[mid:]])
        rd = mutate(rng, frag, sub, ins, dele)
        strand = "+"
        if rng.random() < 0.5:
            rd = COMP[rd[::-1]]
            strand = "-"
        rd = rd.copy()
        if n_frac > 0:
            m = rng.random(len(rd)) < n_frac
            rd[m] = 4
        svs = "" if sv is None else "_%s%d" % sv
        reads.append(("r%d_c%d_%d_%s%s" % (i, ci + 1, pos + 1, "f" if strand == "+" else "r", svs), rd))
    return reads


def write_fasta(path, records, width=60):
    tab = np.frombuffer(b"ACGTN", dtype=np.uint8)
    with open(path, "wb") as f:
        for name, seq in records:
            f.write(b">" + name.encode() + b"\n")
            s = tab[seq].tobytes()
            if width:
                for i in range(0, len(s), width):
                    f.write(s[i:i + width] + b"\n")
            else:
                f.write(s + b"\n")


PROFILES = {
    # name: (sub, ins, del)
    "perfect": (0.0, 0.0, 0.0),
    "lowerr": (0.004, 0.003, 0.003),
    "pacbio": (0.015, 0.09, 0.045),
    "ont": (0.04, 0.04, 0.04),
    "pb20k": (0.01, 0.09, 0.05),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--contigs", type=str, default="100000,100000")
    ap.add_argument("--repeats", type=str, default="", help="unit:copies,unit:copies")
    ap.add_argument("--ref-out", type=str)
    ap.add_argument("--reads-out", type=str)
    ap.add_argument("--n-reads", type=int, default=20)
    ap.add_argument("--length", type=int, default=5000)
    ap.add_argument("--profile", type=str, default="perfect")
    ap.add_argument("--sv-frac", type=float, default=0.0)
    ap.add_argument("--n-frac", type=float, default=0.0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    clens = [int(x) for x in a.contigs.split(",")]
    reps = [tuple(int(y) for y in x.split(":")) for x in a.repeats.split(",") if x]
    contigs = make_reference(rng, clens, reps)
    if a.ref_out:
        write_fasta(a.ref_out, [("chr%d" % (i + 1), c) for i, c in enumerate(contigs)])
    if a.reads_out:
        sub, ins, dele = PROFILES[a.profile]
        reads = simulate_reads(rng, contigs, a.n_reads, a.length, sub, ins, dele, a.sv_frac, n_frac=a.n_frac)
        write_fasta(a.reads_out, reads, width=0)


if __name__ == "__main__":
    main()
