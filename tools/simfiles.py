"""Write a simulated batch as the FILES `lamsa aln` reads: <prefix>.ann/.amb/.pac for the reference stand-in,
reads.fa and reads.fa.seed.gem.map (GEM map text: 5 tab-separated columns, hits `chr:strand:pos:gigar`, `-` when a seed
has none; src/gem_parse.c:212-286).  The gigar strings are built so that parsing them gives back exactly the simulated
CIGAR, NM and length difference of every hit.  Used by tests/test_cli_gpu.py for the end-to-end run at scale.
"""
import numpy as np

BASES = np.frombuffer(b"ACGTN", dtype=np.uint8)


def write_index(prefix, ref):
    n = len(ref.seq_len)
    with open(prefix + ".ann", "w") as f:
        f.write("%d %d %d\n" % (int(ref.l_pac), n, 11))
        for i in range(n):
            f.write("0 chr%d\n%d %d 0\n" % (i + 1, int(ref.seq_off[i]), int(ref.seq_len[i])))
    with open(prefix + ".amb", "w") as f:
        f.write("%d %d 0\n" % (int(ref.l_pac), n))
    nbytes = int(ref.l_pac) // 4 + 1
    pac = np.zeros(nbytes + 1, np.uint8)
    m = min(nbytes, len(ref.pac))
    pac[:m] = ref.pac[:m]
    with open(prefix + ".pac", "wb") as f:
        f.write(pac[:nbytes].tobytes())
        if int(ref.l_pac) % 4 == 0:
            f.write(b"\0")
        f.write(bytes([int(ref.l_pac) % 4]))


def gigar(words, nm, reverse):
    """GEM gigar of one hit from its CIGAR words (len << 4 | op, op 0 M / 1 I / 2 D) and its edit distance."""
    ops = [(int(w) & 0xf, int(w) >> 4) for w in words]
    if reverse:
        ops = ops[::-1]
    mm = nm - sum(l for o, l in ops if o in (1, 2))
    out = []
    for o, l in ops:
        if o == 0:
            x = min(mm, l) if mm > 0 else 0
            mm -= x
            if l - x > 0:
                out.append(str(l - x))
            out.append("A" * x)
        elif o == 1:
            out.append(">%d-" % l)
        else:
            out.append(">%d+" % l)
    assert mm == 0, "mismatches do not fit the match runs"
    return "".join(out)


def _write_part(args):
    path, r0, r1, seed_len, seed_step = args
    B = _SHARED["B"]
    cig_off = _SHARED.get("cig_off")              # 64-bit starts of the seed CIGARs (a batch may hold more than 2^31 elements: B.h_cig_off wraps)
    with open(path + ".fa", "w") as fa, open(path + ".map", "w") as mp:
        for r in range(r0, r1):
            seq = B.read_seq[int(B.read_off[r]):int(B.read_off[r + 1])]
            fa.write(">r%d\n%s\n" % (r, BASES[seq].tobytes().decode()))
            L = len(seq)
            seed_all = 0 if L < seed_len else 1 + (L - seed_len) // seed_step
            assert seed_all == int(B.seed_all[r])
            slot_of = {int(B.seed_id[s]): s for s in range(int(B.seed_off[r]), int(B.seed_off[r + 1]))}
            un = _SHARED.get("unseeded")              # (share of the reads, share of a read's seeds): a stretch in the middle of such a read without any seed hit
            un_lo = un_hi = 0
            if un and (r * 2654435761 % 1000) < 1000 * un[0]:
                un_lo = int(seed_all * 0.4); un_hi = un_lo + max(1, int(seed_all * un[1]))
            for sd in range(1, seed_all + 1):
                s = slot_of.get(sd) if not (un_lo <= sd < un_hi) else None
                hits = []
                if s is not None:
                    for k in range(int(B.hit_off[s]), int(B.hit_off[s + 1])):
                        co, cn = int(cig_off[k]) if cig_off is not None else int(B.h_cig_off[k]), int(B.h_cig_n[k])
                        st = int(B.h_strand[k])
                        hits.append("chr%d:%s:%d:%s" % (int(B.h_chr[k]), "+" if st > 0 else "-", int(B.h_pos[k]), gigar(B.cig[co:co + cn], int(B.h_nm[k]), st < 0)))
                mp.write("r%d_%d\tN\t*\t0\t%s\n" % (r, sd, ",".join(hits) if hits else "-"))
    return path


_SHARED = {}


def write_reads(path, B, seed_len=50, seed_step=25, workers=1, unseeded=None):
    """reads.fa + reads.fa.seed.gem.map for batch B (names r0, r1, ...).  workers > 1: the reads are split over forked
    worker processes (the batch is shared copy-on-write), their part files concatenated in order."""
    import os
    import shutil
    n = B.n_reads
    workers = max(1, min(workers, n // 64 if n >= 128 else 1))
    _SHARED["B"] = B
    if unseeded:
        _SHARED["unseeded"] = unseeded
    if int(getattr(B, "n_cig", 0)) > 0x7fffffff:    # the CIGARs lie back to back in hit order (tools/simhits.c)
        import numpy as np
        cn = np.asarray(B.h_cig_n[:B.n_hits], np.int64)
        _SHARED["cig_off"] = np.concatenate([[0], np.cumsum(cn)[:-1]])
    per = (n + workers - 1) // workers
    jobs = [(path + ".part%d" % w, w * per, min(n, (w + 1) * per), seed_len, seed_step) for w in range(workers) if w * per < n]
    if workers == 1:
        parts = [_write_part(j) for j in jobs]
    else:
        import multiprocessing as mpc
        with mpc.get_context("fork").Pool(workers) as pool:
            parts = pool.map(_write_part, jobs)
    with open(path, "wb") as fa, open(path + ".seed.gem.map", "wb") as mp:
        for q in parts:
            for ext, dst in ((".fa", fa), (".map", mp)):
                with open(q + ext, "rb") as f:
                    shutil.copyfileobj(f, dst, 1 << 24)
                os.remove(q + ext)
    _SHARED.clear()
