"""Python face of tools/simhits.c: GRCh37-like stand-in reference + simulated reads + seed hits at scale,
returned as numpy arrays in the layout of lamsa_hp_batch (see simhits.c for what is simulated and why)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libsimhits.so")


def build():
    src = os.path.join(HERE, "simhits.c")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", SO, src, "-lpthread"], check=True)


class SimCfg(C.Structure):
    _fields_ = [("n_reads", C.c_int), ("length", C.c_int), ("sub", C.c_double), ("ins", C.c_double), ("dele", C.c_double), ("sv_frac", C.c_double),
                ("seed_len", C.c_int), ("seed_step", C.c_int), ("max_edit", C.c_int), ("max_mis", C.c_int), ("min_match", C.c_int),
                ("max_indel", C.c_int), ("per_loci", C.c_int)]


class SimBatchC(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("n_slots", C.c_int64), ("n_hits", C.c_int64), ("n_cig", C.c_int64)] + \
               [(n, C.c_void_p) for n in ("read_off", "read_seq", "seed_all", "last_len", "seed_off", "seed_id", "hit_off", "h_pos", "h_chr", "h_strand",
                                           "h_nm", "h_len_dif", "h_cig_off", "h_cig_n", "cig", "t_chr", "t_pos", "t_strand")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.environ.get("LAMSA_NO_BUILD"):          # bench.py sets it: no gcc child at run time (it may run under a profiler)
            build()
        elif not os.path.exists(SO):
            raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first" % SO)
        L = C.CDLL(SO)
        L.sim_ref_new.restype = C.c_void_p
        L.sim_ref_new.argtypes = [C.c_uint64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.sim_ref_free.argtypes = [C.c_void_p]
        for f, rt in (("sim_ref_l_pac", C.c_int64), ("sim_ref_pac", C.c_void_p), ("sim_ref_seq_off", C.c_void_p), ("sim_ref_seq_len", C.c_void_p), ("sim_ref_n_copies", C.c_int64)):
            getattr(L, f).restype = rt; getattr(L, f).argtypes = [C.c_void_p]
        L.sim_reads_new.restype = C.POINTER(SimBatchC)
        L.sim_reads_new.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(SimCfg), C.c_int]
        L.sim_batch_free.argtypes = [C.POINTER(SimBatchC)]
        _lib = L
    return _lib


# repeat landscape of the stand-in: (unit bp, copies per family, divergence of a copy from its consensus); the number of
# families of each shape is scaled so that repeats cover ~45% of the genome.
SHAPES = [(300, 800, 0.05), (1000, 400, 0.05), (6000, 150, 0.05), (400, 40, 0.03), (2000, 8, 0.02)]
SHARE = [0.045, 0.04, 0.06, 0.03, 0.03]


class SimRef:
    """Reference stand-in resident in host memory (packed 2 bit/base)."""

    def __init__(self, total_bp, n_contigs=24, seed=5, threads=8, repeats=True):
        L = lib()
        lens = np.full(n_contigs, total_bp // n_contigs, np.int64)
        unit = np.array([s[0] for s in SHAPES], np.int32); copies = np.array([s[1] for s in SHAPES], np.int32)
        div = np.array([s[2] for s in SHAPES], np.float64)
        count = np.array([max(0, int(round(total_bp * sh / (u * c)))) if repeats else 0 for (u, c, _), sh in zip(SHAPES, SHARE)], np.int32)
        self._p = L.sim_ref_new(seed, n_contigs, lens.ctypes.data, len(SHAPES), unit.ctypes.data, copies.ctypes.data, count.ctypes.data, div.ctypes.data, threads)
        self.l_pac = int(L.sim_ref_l_pac(self._p)); self.n_copies = int(L.sim_ref_n_copies(self._p))
        nb = self.l_pac // 4 + 1
        self.pac = np.ctypeslib.as_array(C.cast(L.sim_ref_pac(self._p), C.POINTER(C.c_uint8)), (nb,))      # view, owned by the C side
        self.seq_off = np.ctypeslib.as_array(C.cast(L.sim_ref_seq_off(self._p), C.POINTER(C.c_int64)), (n_contigs,)).copy()
        self.seq_len = np.ctypeslib.as_array(C.cast(L.sim_ref_seq_len(self._p), C.POINTER(C.c_int32)), (n_contigs,)).copy()
        self.families = count.tolist()

    def close(self):
        if self._p:
            lib().sim_ref_free(self._p); self._p = None


PROFILES = {  # read error model + GEM thresholds per read type (gem_map.sh arguments: -m mis_rate -e ed_rate --min-matched-bases mat_rate)
    "default": dict(sub=0.004, ins=0.003, dele=0.003, seed_step=100, max_edit=2, max_mis=2, min_match=40),
    "pacbio": dict(sub=0.015, ins=0.09, dele=0.045, seed_step=25, max_edit=15, max_mis=2, min_match=35),
    "ont2d": dict(sub=0.04, ins=0.04, dele=0.04, seed_step=25, max_edit=12, max_mis=3, min_match=30),
    "pb20k": dict(sub=0.01, ins=0.09, dele=0.05, seed_step=25, max_edit=15, max_mis=2, min_match=35),
    # BASELINE.json config 5: 1 % errors, one SV (deletion 1-10 kbp or novel insertion 1-5 kbp at the read's middle) in 2/3 of the reads
    "sv10k": dict(sub=0.004, ins=0.003, dele=0.003, seed_step=100, max_edit=2, max_mis=2, min_match=40, sv_frac=0.67),
}


class SimBatch:
    """Simulated reads + seed hits (numpy arrays named like the fields of lamsa_hp_batch) against a SimRef."""

    def __init__(self, ref, n_reads, length, profile, seed=7, threads=8):
        L = lib()
        p = PROFILES[profile]
        cfg = SimCfg(n_reads, length, p["sub"], p["ins"], p["dele"], p.get("sv_frac", 0.0), 50, p["seed_step"], p["max_edit"], p["max_mis"], p["min_match"], 3, 200)
        bp = L.sim_reads_new(ref._p, seed, C.byref(cfg), threads)
        if not bp:
            need = 2 * length + 16 + (10016 if p.get("sv_frac", 0.0) > 0 else 0)
            raise ValueError("no contig of the reference is longer than %d bp: a read of %d bases (profile %s) has no locus to be drawn from" % (need, length, profile))
        b = bp.contents
        n, ns, nh = b.n_reads, b.n_slots, b.n_hits

        def arr(ptr, cnt, dt):
            cnt = int(cnt)
            if cnt <= 0:
                return np.zeros(4, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), (cnt,)).copy()
        self.n_reads = n
        # more than 2^31-1 seed CIGAR elements: h_cig_off (int32) wraps, such a batch goes through the boundary's compact form only
        # (lamsa_amd.hp.compact_batch rebuilds the offsets from the lengths)
        self.n_cig = int(b.n_cig)
        self.read_off = arr(b.read_off, n + 1, np.int64); self.read_seq = arr(b.read_seq, self.read_off[n], np.uint8)
        self.seed_all = arr(b.seed_all, n, np.int32); self.last_len = arr(b.last_len, n, np.int32)
        self.seed_off = arr(b.seed_off, n + 1, np.int64); self.seed_id = arr(b.seed_id, ns, np.int32); self.hit_off = arr(b.hit_off, ns + 1, np.int64)
        self.h_pos = arr(b.h_pos, nh, np.int64); self.h_chr = arr(b.h_chr, nh, np.int32); self.h_strand = arr(b.h_strand, nh, np.int8)
        self.h_nm = arr(b.h_nm, nh, np.int16); self.h_len_dif = arr(b.h_len_dif, nh, np.int16); self.h_cig_off = arr(b.h_cig_off, nh, np.int32)
        self.h_cig_n = arr(b.h_cig_n, nh, np.uint8); self.cig = arr(b.cig, b.n_cig, np.int32)
        self.t_chr = arr(b.t_chr, n, np.int32); self.t_pos = arr(b.t_pos, n, np.int64); self.t_strand = arr(b.t_strand, n, np.int8)
        self.n_slots, self.n_hits = int(ns), int(nh)
        L.sim_batch_free(bp)
        self.pac, self.l_pac, self.seq_off, self.seq_len = ref.pac, ref.l_pac, ref.seq_off, ref.seq_len


def take(B, idx):
    """The reads `idx` of a batch, in that order, as a new batch object (the seed-CIGAR arena is shared)."""
    class _Sub:
        pass
    s = _Sub()
    idx = [int(i) for i in idx]
    n = len(idx)
    s.n_reads = n
    lens = np.array([B.read_off[i + 1] - B.read_off[i] for i in idx], np.int64)
    s.read_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    s.read_seq = np.concatenate([B.read_seq[int(B.read_off[i]):int(B.read_off[i + 1])] for i in idx] + [np.zeros(8, np.uint8)])
    s.seed_all = np.ascontiguousarray(B.seed_all[idx]) if n else np.zeros(4, np.int32)
    s.last_len = np.ascontiguousarray(B.last_len[idx]) if n else np.zeros(4, np.int32)
    ns = np.array([B.seed_off[i + 1] - B.seed_off[i] for i in idx], np.int64)
    s.seed_off = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    slots = np.concatenate([np.arange(B.seed_off[i], B.seed_off[i + 1]) for i in idx] + [np.zeros(0, np.int64)]).astype(np.int64)
    s.seed_id = np.concatenate([B.seed_id[slots], np.zeros(4, np.int32)]).astype(np.int32)
    nh = (B.hit_off[slots + 1] - B.hit_off[slots]) if len(slots) else np.zeros(0, np.int64)
    s.hit_off = np.concatenate([[0], np.cumsum(nh)]).astype(np.int64)
    hits = np.concatenate([np.arange(B.hit_off[B.seed_off[i]], B.hit_off[B.seed_off[i + 1]]) for i in idx] + [np.zeros(0, np.int64)]).astype(np.int64)
    for k in ("h_pos", "h_chr", "h_strand", "h_nm", "h_len_dif", "h_cig_off", "h_cig_n"):
        a = getattr(B, k)
        setattr(s, k, np.concatenate([a[hits], np.zeros(4, a.dtype)]))
    s.cig = B.cig
    s.n_slots, s.n_hits = int(len(slots)), int(len(hits))
    s.pac, s.l_pac, s.seq_off, s.seq_len = B.pac, B.l_pac, B.seq_off, B.seq_len
    return s
