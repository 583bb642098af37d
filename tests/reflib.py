"""ctypes bindings to the checkers used by the tests (test infrastructure).

* ``oracle()``  -> oracle/liblamsa_oracle.so, our plain-C restatement (always buildable)
* ``ref()``     -> oracle/_ref/liblamsa_ref.so, the reference itself compiled from
                   /root/reference (only present where ``make -C oracle ref`` could run,
                   or where the prebuilt .so travelled); ``None`` when absent.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class LoCigv(C.Structure):
    _fields_ = [("c", C.POINTER(C.c_int32)), ("n", C.c_int), ("m", C.c_int)]


class LoPara(C.Structure):
    """Mirror of lo_para (oracle/lo.h)."""
    _fields_ = [(n, C.c_int32) for n in ("seed_len", "seed_step", "seed_inv", "per_aln_m", "first_loci_thd",
                                          "SV_len_thd", "ske_max")] + [("ovlp_rat", C.c_float)] + \
               [(n, C.c_int32) for n in ("bwt_seed_len", "bwt_max_len", "bwt_min_len", "split_len", "split_pen",
                                          "res_mul_max", "hash_len", "hash_key_len", "hash_step", "hash_size",
                                          "match_dis", "mismatch_thd", "ins_gapo", "ins_gape", "del_gapo", "del_gape",
                                          "ins_ext_o", "ins_ext_e", "del_ext_o", "del_ext_e", "match", "mis")] + \
               [("sc_mat", C.c_int8 * 25)] + \
               [(n, C.c_int32) for n in ("band_w", "end_bonus", "zdrop")] + [("id_rate", C.c_float)] + \
               [(n, C.c_int32) for n in ("read_type", "aln_mode", "supp_soft", "comm")]


class RefPara(C.Structure):
    """Mirror of lamsa_aln_para (reference src/lamsa_aln.h:386-436) for calling into liblamsa_ref.so."""
    _fields_ = [("n_thread", C.c_int), ("seed_len", C.c_int), ("seed_step", C.c_int), ("seed_inv", C.c_int),
                ("per_aln_m", C.c_int), ("first_loci_thd", C.c_int), ("SV_len_thd", C.c_int), ("ske_max", C.c_int),
                ("ovlp_rat", C.c_float), ("bwt_seed_len", C.c_int), ("bwt_max_len", C.c_int), ("bwt_min_len", C.c_int),
                ("fastest", C.c_int), ("split_len", C.c_int), ("split_pen", C.c_int), ("res_mul_max", C.c_int),
                ("hash_len", C.c_int), ("hash_key_len", C.c_int), ("hash_step", C.c_int), ("hash_size", C.c_int),
                ("supp_soft", C.c_uint8), ("comm", C.c_uint8), ("outp", C.c_void_p),
                ("match_dis", C.c_int), ("mismatch_thd", C.c_int), ("del_thd", C.c_int), ("ins_thd", C.c_int),
                ("frag_score_table", C.c_void_p),
                ("ins_gapo", C.c_int), ("ins_gape", C.c_int), ("del_gapo", C.c_int), ("del_gape", C.c_int),
                ("ins_ext_o", C.c_int), ("ins_ext_e", C.c_int), ("del_ext_o", C.c_int), ("del_ext_e", C.c_int),
                ("match", C.c_int), ("mis", C.c_int), ("sc_mat", C.c_int8 * 25),
                ("band_w", C.c_int), ("end_bonus", C.c_int), ("zdrop", C.c_int),
                ("ed_rate", C.c_float), ("mis_rate", C.c_float), ("id_rate", C.c_float), ("mat_rate", C.c_float),
                ("read_type", C.c_int), ("aln_mode", C.c_uint8)]


_oracle = None
_ref = False


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def oracle():
    global _oracle
    if _oracle is None:
        so = os.path.join(ORACLE_DIR, "liblamsa_oracle.so")
        if not os.environ.get("LAMSA_NO_BUILD"):          # bench.py sets it: no make / gcc children at run time (it may run under a profiler)
            build_oracle()
        elif not os.path.exists(so):
            raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first" % so)
        L = C.CDLL(so)
        L.lo_para_init.argtypes = [C.POINTER(LoPara)]
        L.lo_para_finish.argtypes = [C.POINTER(LoPara)]
        _oracle = L
    return _oracle


def ref():
    """The compiled reference, or None."""
    global _ref
    if _ref is False:
        p = os.path.join(ORACLE_DIR, "_ref", "liblamsa_ref.so")
        _ref = C.CDLL(p) if os.path.exists(p) else None
    return _ref


READ_TYPES = {"default": 0, "pacbio": 1, "ont2d": 2}


def lo_para(read_type="default", **over):
    L = oracle()
    P = LoPara()
    L.lo_para_init(C.byref(P))
    P.read_type = READ_TYPES[read_type]
    for k, v in over.items():
        setattr(P, k, v)
    L.lo_para_finish(C.byref(P))
    return P


def ref_para(read_type="default", **over):
    R = ref()
    P = RefPara()
    R.init_aln_para(C.byref(P))
    P.read_type = READ_TYPES[read_type]
    for k, v in over.items():
        setattr(P, k, v)
    R.lamsa_set_aln_mode(C.byref(P))
    P.seed_inv = P.seed_step - P.seed_len
    R.lamsa_fill_mat(P.match, P.mis, P.sc_mat)
    return P


def cig_list(v):
    return [int(v.c[i]) for i in range(v.n)]


def u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


# ---------------------------------------------------------------- DP helpers (oracle / reference / CPU emulation)
KIND_GLOBAL, KIND_EXTEND, KIND_BI = 0, 1, 2


def oracle_dp(jobs, P, kind, w, h0):
    """Run DP jobs through the plain-C oracle.  Returns the same dict shape as LamsaHp.dp_batch."""
    L = oracle()
    n = len(jobs)
    kind = np.broadcast_to(kind, n); w = np.broadcast_to(w, n); h0 = np.broadcast_to(h0, n)
    score = np.zeros(n, np.int32); qle = np.zeros(n, np.int32); tle = np.zeros(n, np.int32)
    cigars = []
    for i, (q, t) in enumerate(jobs):
        ql, tl = len(q), len(t)
        _q, qp = u8(q if ql else np.zeros(1, np.uint8)); _t, tp = u8(t if tl else np.zeros(1, np.uint8))
        v = LoCigv(); L.lo_cigv_init(C.byref(v))
        a, b = C.c_int(0), C.c_int(0)
        if kind[i] == KIND_GLOBAL:
            score[i] = L.lo_ksw_global(ql, qp, tl, tp, P.sc_mat, P.del_gapo, P.del_gape, P.ins_gapo, P.ins_gape, int(w[i]), C.byref(v))
        elif kind[i] == KIND_EXTEND:
            score[i] = L.lo_ksw_extend(ql, qp, tl, tp, P.sc_mat, int(w[i]), int(h0[i]), C.byref(P), C.byref(a), C.byref(b), C.byref(v))
            qle[i], tle[i] = a.value, b.value
        else:
            score[i] = L.lo_ksw_bi_extend(ql, qp, tl, tp, int(h0[i]), int(h0[i]), C.byref(P), C.byref(v))
        cigars.append(cig_list(v))
        L.lo_cigv_free(C.byref(v))
    return dict(score=score, qle=qle, tle=tle, status=np.zeros(n, np.int32), cigars=cigars)


def ref_dp(jobs, P, kind, w, h0):
    """Same through the compiled reference (ksw_global2 / ksw_extend_core / ksw_bi_extend)."""
    R = ref()
    n = len(jobs)
    kind = np.broadcast_to(kind, n); w = np.broadcast_to(w, n); h0 = np.broadcast_to(h0, n)
    score = np.zeros(n, np.int32); qle = np.zeros(n, np.int32); tle = np.zeros(n, np.int32)
    cigars = []
    for i, (q, t) in enumerate(jobs):
        ql, tl = len(q), len(t)
        _q, qp = u8(q if ql else np.zeros(1, np.uint8)); _t, tp = u8(t if tl else np.zeros(1, np.uint8))
        nc, mc = C.c_int(0), C.c_int(0); cg = C.POINTER(C.c_int32)()
        a, b = C.c_int(0), C.c_int(0)
        if kind[i] == KIND_GLOBAL:
            score[i] = R.ksw_global2(ql, qp, tl, tp, 5, P.sc_mat, P.del_gapo, P.del_gape, P.ins_gapo, P.ins_gape, int(w[i]), C.byref(nc), C.byref(cg))
        elif kind[i] == KIND_EXTEND:
            score[i] = R.ksw_extend_core(ql, qp, tl, tp, 5, P.sc_mat, int(w[i]), int(h0[i]), C.byref(P), C.byref(a), C.byref(b), C.byref(cg), C.byref(nc), C.byref(mc))
            qle[i], tle[i] = a.value, b.value
        else:
            score[i] = R.ksw_bi_extend(ql, qp, tl, tp, 5, P.sc_mat, int(h0[i]), int(h0[i]), C.byref(P), C.byref(cg), C.byref(nc), C.byref(mc))
        cigars.append([int(cg[k]) for k in range(nc.value)])
    return dict(score=score, qle=qle, tle=tle, status=np.zeros(n, np.int32), cigars=cigars)


_emu = None
# LAMSA_EMU_COVERAGE=1 (tools/device_coverage.sh): the emulation build instrumented for gcov, in its own directory
EMU_COV = bool(os.environ.get("LAMSA_EMU_COVERAGE"))
EMU_DIR = os.path.join(ROOT, "tests", "_build_cov" if EMU_COV else "_build")
EMU_FLAGS = ["-O0", "--coverage"] if EMU_COV else ["-O1"]


def emu():
    """The device sources compiled against the CPU lane emulation (tests/emu)."""
    global _emu
    if _emu is None:
        os.makedirs(EMU_DIR, exist_ok=True)
        out = os.path.join(EMU_DIR, "libhp_emu.so")
        srcs = [os.path.join(ROOT, "tests", "emu", "emu_api.cpp")]
        deps = srcs + [os.path.join(ROOT, "tests", "emu", "hp", "wave.h")] + \
            [os.path.join(ROOT, "lamsa_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "lamsa_amd", "csrc")) if f.endswith(".h")]
        if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
            subprocess.run(["g++"] + EMU_FLAGS + ["-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                            "-I", os.path.join(ROOT, "tests", "emu"), "-I", os.path.join(ROOT, "lamsa_amd", "csrc"),
                            "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-o", out] + srcs, check=True, cwd=EMU_DIR)
        _emu = C.CDLL(out)
    return _emu


def emu_cli():
    """tests/_build/lamsa_emu: the product's host program (lamsa_amd/host) linked against the emulated C-ABI
    (tests/emu/emu_capi.cpp) instead of liblamsa_hp.so -- exercises file IO, GEM parsing, ranking and SAM on the CPU."""
    os.makedirs(EMU_DIR, exist_ok=True)
    out = os.path.join(EMU_DIR, "lamsa_emu")
    host = os.path.join(ROOT, "lamsa_amd", "host")
    srcs = [os.path.join(host, "main.cpp"), os.path.join(host, "lamsa_host.cpp"), os.path.join(host, "rescue.cpp"), os.path.join(host, "index.cpp"),
            os.path.join(ROOT, "tests", "emu", "emu_api.cpp"), os.path.join(ROOT, "tests", "emu", "emu_capi.cpp")]
    deps = srcs + [os.path.join(host, "lamsa_host.h"), os.path.join(host, "rescue.h"), os.path.join(ROOT, "include", "lamsa_hp.h"), os.path.join(ROOT, "tests", "emu", "hp", "wave.h")] + \
        [os.path.join(ROOT, "lamsa_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "lamsa_amd", "csrc")) if f.endswith(".h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.run(["g++"] + EMU_FLAGS + ["-g", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "tests", "emu"),
                        "-I", os.path.join(ROOT, "lamsa_amd", "csrc"), "-o", out] + srcs + ["-lz", "-lpthread"], check=True, cwd=EMU_DIR)
    return out


def emu_capi_lib():
    """tests/_build/liblamsa_hp_emu.so: the C-ABI of include/lamsa_hp.h on the CPU lane emulation, as a shared library
    (what tests/test_glue_cpu.py links the patched reference against instead of liblamsa_hp.so)."""
    os.makedirs(EMU_DIR, exist_ok=True)
    out = os.path.join(EMU_DIR, "liblamsa_hp_emu.so")
    srcs = [os.path.join(ROOT, "tests", "emu", "emu_api.cpp"), os.path.join(ROOT, "tests", "emu", "emu_capi.cpp")]
    deps = srcs + [os.path.join(ROOT, "include", "lamsa_hp.h"), os.path.join(ROOT, "tests", "emu", "hp", "wave.h")] + \
        [os.path.join(ROOT, "lamsa_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "lamsa_amd", "csrc")) if f.endswith(".h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.run(["g++"] + EMU_FLAGS + ["-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", os.path.join(ROOT, "tests", "emu"),
                        "-I", os.path.join(ROOT, "lamsa_amd", "csrc"), "-o", out] + srcs + ["-lpthread"], check=True, cwd=EMU_DIR)
    return out


def emu_dp(jobs, hp_para, kind, w, h0, slab_bytes=64 << 20, pk=True, stats=None):
    """Run DP jobs through the emulated device code (same kernel sources, CPU lanes).  pk=False: extensions of 63 .. 254 query bases take the
    int32 register sets instead of the packed int16 routine; stats: a list that receives [extensions through the packed routine, through the int32 sets, global alignments of more than 62
    query bases through the packed routine, through the LDS rows, extensions of more than 254 query bases through the window-in-registers routine]."""
    import sys
    sys.path.insert(0, ROOT)
    from lamsa_amd.hp import pack_jobs
    E = emu()
    n = len(jobs)
    seq, q_off, qlen, t_off, tlen = pack_jobs(jobs)
    kind = np.ascontiguousarray(np.broadcast_to(kind, n), np.int32)
    w = np.ascontiguousarray(np.broadcast_to(w, n), np.int32)
    h0 = np.ascontiguousarray(np.broadcast_to(h0, n), np.int32)
    cap = np.zeros(n + 1, np.int64)
    cap[1:] = np.cumsum(qlen.astype(np.int64) + tlen + 8)
    score = np.zeros(n, np.int32); qle = np.zeros(n, np.int32); tle = np.zeros(n, np.int32)
    st = np.zeros(n, np.int32); cn = np.zeros(n, np.int32); cig = np.zeros(int(cap[n]) + 4, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    E.emu_dp_batch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 15 + [C.c_size_t]
    E.emu_set_pk(1 if pk else 0); E.emu_stat_reset(); E.emu_stat.restype = C.c_longlong
    E.emu_dp_batch(C.byref(hp_para), n, p(seq), p(q_off), p(qlen), p(t_off), p(tlen), p(kind), p(w), p(h0),
                   p(score), p(qle), p(tle), p(st), p(cn), p(cap), p(cig), slab_bytes)
    E.emu_set_pk(1)
    if stats is not None:
        stats[:] = [int(E.emu_stat(16)), int(E.emu_stat(17)), int(E.emu_stat(18)), int(E.emu_stat(19)), int(E.emu_stat(22))]
    cigars = [cig[cap[i]:cap[i] + cn[i]].tolist() for i in range(n)]
    return dict(score=score, qle=qle, tle=tle, status=st, cigars=cigars)


# ---------------------------------------------------------------- whole-path batches (SoA) + result streams
class LoRef(C.Structure):
    _fields_ = [("pac", C.c_void_p), ("l_pac", C.c_int64), ("n_seqs", C.c_int), ("seq_offset", C.c_void_p), ("seq_len", C.c_void_p)]


class LoIndex(C.Structure):
    _fields_ = [("ref", LoRef), ("name", C.c_void_p), ("off", C.c_void_p), ("len", C.c_void_p), ("pac", C.c_void_p)]


class LoBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("n_slots", C.c_int64), ("n_hits", C.c_int64), ("n_cig", C.c_int64)] + \
               [(n, C.c_void_p) for n in ("read_off", "read_seq", "seed_all", "last_len", "seed_off", "seed_id", "hit_off",
                                           "h_pos", "h_chr", "h_strand", "h_nm", "h_len_dif", "h_cig_off", "h_cig_n", "cig")]


def _np(ptr, n, dt):
    n = int(n)
    if n <= 0 or not ptr:
        return np.zeros(max(n, 0), dt)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), (n,)).copy()


class Batch:
    """A batch of reads + seed hits as numpy arrays in the layout of lamsa_hp_batch, plus the packed reference."""

    def __init__(self, ref_prefix, reads, P, max_reads=0):
        L = oracle()
        ix = LoIndex()
        if L.lo_index_load(C.byref(ix), ref_prefix.encode()) != 0:
            raise RuntimeError("cannot load index " + ref_prefix)
        b = LoBatch()
        L.lo_batch_load.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_long, C.c_void_p]
        if L.lo_batch_load(C.byref(ix), reads.encode(), C.byref(P), max_reads, C.byref(b)) != 0:
            raise RuntimeError("cannot load batch " + reads)
        n, ns, nh = b.n_reads, b.n_slots, b.n_hits
        self.n_reads = n
        self.read_off = _np(b.read_off, n + 1, np.int64); self.read_seq = _np(b.read_seq, self.read_off[n] if n else 0, np.uint8)
        self.seed_all = _np(b.seed_all, n, np.int32); self.last_len = _np(b.last_len, n, np.int32)
        self.seed_off = _np(b.seed_off, n + 1, np.int64); self.seed_id = _np(b.seed_id, ns, np.int32); self.hit_off = _np(b.hit_off, ns + 1, np.int64)
        self.h_pos = _np(b.h_pos, nh, np.int64); self.h_chr = _np(b.h_chr, nh, np.int32); self.h_strand = _np(b.h_strand, nh, np.int8)
        self.h_nm = _np(b.h_nm, nh, np.int16); self.h_len_dif = _np(b.h_len_dif, nh, np.int16)
        self.h_cig_off = _np(b.h_cig_off, nh, np.int32); self.h_cig_n = _np(b.h_cig_n, nh, np.uint8); self.cig = _np(b.cig, b.n_cig, np.int32)
        ns_ = ix.ref.n_seqs
        self.l_pac = ix.ref.l_pac
        self.pac = _np(ix.ref.pac, ix.ref.l_pac // 4 + 1, np.uint8)
        self.seq_off = _np(ix.ref.seq_offset, ns_, np.int64); self.seq_len = _np(ix.ref.seq_len, ns_, np.int32)
        L.lo_batch_free(C.byref(b)); L.lo_index_free(C.byref(ix))
        for name in ("read_seq", "seed_id", "h_pos", "h_chr", "h_strand", "h_nm", "h_len_dif", "h_cig_off", "h_cig_n", "cig"):
            a = getattr(self, name)
            if len(a) == 0:
                setattr(self, name, np.zeros(4, a.dtype))      # keep pointers valid

    def take(self, idx):
        """A new Batch holding the reads idx (list of indices), renumbered."""
        o = object.__new__(Batch)
        for k in ("pac", "l_pac", "seq_off", "seq_len", "cig"):
            setattr(o, k, getattr(self, k))
        ro, so, ho = [0], [0], [0]
        seqs, sid, sall, last = [], [], [], []
        hsel = []
        for r in idx:
            seqs.append(self.read_seq[self.read_off[r]:self.read_off[r + 1]]); ro.append(ro[-1] + len(seqs[-1]))
            sall.append(self.seed_all[r]); last.append(self.last_len[r])
            for s in range(self.seed_off[r], self.seed_off[r + 1]):
                sid.append(self.seed_id[s]); a, b = self.hit_off[s], self.hit_off[s + 1]
                hsel.extend(range(a, b)); ho.append(ho[-1] + (b - a))
            so.append(len(sid))
        hsel = np.array(hsel, np.int64)
        o.n_reads = len(idx)
        o.read_off = np.array(ro, np.int64); o.read_seq = np.concatenate(seqs + [np.zeros(4, np.uint8)]).astype(np.uint8)
        o.seed_all = np.array(sall, np.int32); o.last_len = np.array(last, np.int32); o.seed_off = np.array(so, np.int64)
        o.seed_id = np.array(sid + [0], np.int32); o.hit_off = np.array(ho, np.int64)
        for k, dt in (("h_pos", np.int64), ("h_chr", np.int32), ("h_strand", np.int8), ("h_nm", np.int16), ("h_len_dif", np.int16), ("h_cig_off", np.int32), ("h_cig_n", np.uint8)):
            setattr(o, k, np.concatenate([getattr(self, k)[hsel] if len(hsel) else np.zeros(0, dt), np.zeros(4, dt)]).astype(dt))
        return o

    def c_lo_batch(self):
        b = LoBatch()
        b.n_reads = self.n_reads; b.n_slots = int(self.seed_off[-1]); b.n_hits = int(self.hit_off[-1]); b.n_cig = len(self.cig)
        for name in ("read_off", "read_seq", "seed_all", "last_len", "seed_off", "seed_id", "hit_off", "h_pos", "h_chr", "h_strand",
                     "h_nm", "h_len_dif", "h_cig_off", "h_cig_n", "cig"):
            a = np.ascontiguousarray(getattr(self, name)); setattr(self, name, a); setattr(b, name, a.ctypes.data)
        return b

    def c_lo_ref(self):
        r = LoRef()
        r.pac = self.pac.ctypes.data; r.l_pac = int(self.l_pac); r.n_seqs = len(self.seq_len)
        r.seq_offset = self.seq_off.ctypes.data; r.seq_len = self.seq_len.ctypes.data
        return r


def split_streams(stream, off, ln):
    return [stream[int(off[i]):int(off[i]) + int(ln[i])].tolist() for i in range(len(off))]


def oracle_streams(batch, P, n_threads=4):
    """Per-read result streams (lists of ints) from the oracle."""
    L = oracle()
    n = batch.n_reads
    b, r = Batch.c_lo_batch(batch), Batch.c_lo_ref(batch)
    sp = C.POINTER(C.c_int32)(); nw = C.c_int64(0)
    off = np.zeros(max(n, 1), np.int64); ln = np.zeros(max(n, 1), np.int32); st = np.zeros(max(n, 1), np.int32)
    L.lo_batch_align_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.lo_batch_align_stream(C.byref(b), C.byref(r), C.byref(P), n_threads, C.byref(sp), C.byref(nw), off.ctypes.data, ln.ctypes.data, st.ctypes.data)
    stream = np.ctypeslib.as_array(sp, (max(nw.value, 1),)).copy()
    C.CDLL(None).free(sp)
    return split_streams(stream, off[:n], ln[:n])


def emu_lane_dp(jobs, hp_para, kind, w, h0):
    """DP jobs through the lane-per-job routines of hp_lanedp.h (what k_filldp runs), CPU lane emulation."""
    from lamsa_amd.hp import pack_jobs
    E = emu()
    n = len(jobs)
    seq, q_off, qlen, t_off, tlen = pack_jobs(jobs)
    CIG = 160 + 256 + 8
    score = np.zeros(n, np.int32); qle = np.zeros(n, np.int32); tle = np.zeros(n, np.int32); cn = np.zeros(n, np.int32); cig = np.zeros(n * CIG + 4, np.int32)
    p = lambda a: a.ctypes.data
    E.emu_lane_dp.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p] * 5
    rc = E.emu_lane_dp(C.byref(hp_para), n, p(seq), p(q_off), p(qlen), p(t_off), p(tlen), int(kind), int(w), int(h0), p(score), p(qle), p(tle), p(cn), p(cig))
    assert rc == 0
    return dict(score=score, qle=qle, tle=tle, cigars=[cig[i * CIG:i * CIG + cn[i]].tolist() for i in range(n)])


def split_indel_map_three_ways(read, refw, ref_offset, lp, rp, hp_para, ref_start=0, ref_len=None):
    """split_indel_map (src/split_mapping.c:829) on one read gap and reference window: (oracle, reference or None, emulated device code),
    each as (return value, CIGAR words).  The window is refw[ref_start : ref_start + ref_len] (default: all of refw); the DUP branch of
    split_mapping hands over a window with hash_len - dis fetched bases on either side of it (src/frag_check.c:529-545)."""
    L = oracle(); E = emu()
    _r, rptr = u8(read); _t, tbase = u8(refw)
    full = refw
    refw = full[ref_start:ref_start + (len(full) - ref_start if ref_len is None else ref_len)]
    tptr = C.c_void_p(C.cast(tbase, C.c_void_p).value + int(ref_start))
    v = LoCigv(); L.lo_cigv_init(C.byref(v))
    L.lo_split_indel_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    o_ret = L.lo_split_indel_map(C.byref(v), rptr, len(read), tptr, len(refw), int(ref_offset), C.byref(lp))
    o = (int(o_ret), cig_list(v)); L.lo_cigv_free(C.byref(v))
    r = None
    if rp is not None:
        Rf = ref(); libc = C.CDLL(None)
        libc.calloc.restype = C.c_void_p; libc.calloc.argtypes = [C.c_size_t, C.c_size_t]
        hash_num = C.c_void_p(libc.calloc(rp.hash_size, 4)); hash_node = C.c_void_p(libc.calloc(rp.hash_size, 8))
        cg = C.POINTER(C.c_int32)(); nc, mc = C.c_int(0), C.c_int(0)
        Rf.split_indel_map.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        rr = Rf.split_indel_map(C.byref(cg), C.byref(nc), C.byref(mc), rptr, len(read), tptr, len(refw), int(ref_offset), C.byref(rp), C.byref(hash_num), C.byref(hash_node))
        r = (int(rr), [int(cg[k]) for k in range(nc.value)])
    cap = 2 * (len(read) + len(refw)) + 64
    cig = np.zeros(cap, np.int32); ret = C.c_int32(0); st = C.c_int32(0)
    E.emu_split_indel_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    n = E.emu_split_indel_map(C.byref(hp_para), rptr, len(read), tptr, len(refw), int(ref_offset), cig.ctypes.data, cap, C.byref(ret), C.byref(st))
    e = (int(ret.value), cig[:n].tolist(), int(st.value))
    return o, r, e


def emu_wave_job(jobs, hp_para, wtype, w, h0, slab_bytes=64 << 20):
    """Jobs as the wave-per-job launch of the read path runs them (lamsa_amd/csrc/hp_wavejob.h, CPU lane emulation): wtype 1 = a junction's
    ksw_bi_extend(h0, h0), 2 = ksw_global2(w), 3 / 4 = a line's head / tail extension (ksw_extend_r / ksw_extend_c with (w, h0), the rest of the
    query soft-clipped, the head's CIGAR turned round)."""
    from lamsa_amd.hp import pack_jobs
    E = emu()
    n = len(jobs)
    seq, q_off, qlen, t_off, tlen = pack_jobs(jobs)
    cig_off = np.zeros(n + 1, np.int64)
    cig_off[1:] = np.cumsum(qlen.astype(np.int64) + tlen + 72)
    score = np.zeros(n, np.int32); qle = np.zeros(n, np.int32); tle = np.zeros(n, np.int32); st = np.zeros(n, np.int32); cn = np.zeros(n, np.int32)
    cig = np.zeros(int(cig_off[-1]) + 4, np.int32)
    p = lambda a: a.ctypes.data
    E.emu_wave_job.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_int, C.c_size_t] + [C.c_void_p] * 7
    rc = E.emu_wave_job(C.byref(hp_para), n, p(seq), p(q_off), p(qlen), p(t_off), p(tlen), int(wtype), int(w), int(h0), slab_bytes, p(score), p(qle), p(tle), p(st), p(cn), p(cig), p(cig_off))
    assert rc == 0, rc
    return dict(score=score, qle=qle, tle=tle, status=st, cigars=[cig[cig_off[i]:cig_off[i] + cn[i]].tolist() for i in range(n)])


def end_extension_from_oracle(jobs, lp, head, w, h0):
    """What frag_head_bound_fix / frag_tail_bound_fix make of an extension (src/frag_check.c:640-648, :699-703), from the oracle's
    ksw_extend_core: the head runs on both sequences reversed; unless the query was consumed the rest of it is soft-clipped; the head's
    CIGAR is turned round."""
    js = [(q[::-1].copy(), t[::-1].copy()) for q, t in jobs] if head else jobs
    want = oracle_dp(js, lp, 1, w, h0)
    cigs = []
    for k, (q, t) in enumerate(js):
        c = list(want["cigars"][k])
        if want["qle"][k] != len(q):
            rest = (len(q) - int(want["qle"][k])) << 4 | 4
            if rest >> 4:
                if c and (c[-1] & 0xf) == 4:
                    c[-1] += (rest >> 4) << 4
                else:
                    c.append(rest)
        cigs.append(c[::-1] if head else c)
    return dict(score=want["score"], qle=want["qle"], tle=want["tle"], cigars=cigs)


def emu_streams(batch, hp_para, scale=1, slab_bytes=256 << 20, phased=True, unit_cap=0, cl_cap=0, lane_dp=True, gaptab_cap=0, gap_mcap=0, stats=None, wave_jobs=True, wj_small=0, frag_block_min=0):
    """Per-read result streams from the device sources compiled with the CPU lane emulation.
    phased: scale-1 batches go through the launches of hp_phase.h (the product's main pass) instead of the one-kernel path.
    cl_cap: clusters of more hits than this take the HBM path of the main chaining pass instead of the LDS one (hp_cluster.h).
    gaptab_cap / gap_mcap: reads with more seed slots scan the gaps of a line by seed range instead of by cluster / gaps with more
    survivors take the wave-wide mini DP (hp_gaps.h); < 0 = none qualifies.  wave_jobs: False = no wave-per-job launch (hp_wavejob.h), the fill
    runs the junctions beyond a lane job and the end extensions itself; wj_small: the ordinary slab of a wave job in bytes (jobs that need more go to
    the waves that own a big slab).  stats: a list that receives the HP_STAT path counters."""
    from lamsa_amd.hp import HpRef, HpBatch
    E = emu()
    E.emu_set_phased(1 if phased else 0); E.emu_set_unit_cap(int(unit_cap)); E.emu_set_cl_cap(int(cl_cap)); E.emu_set_lane_dp(1 if lane_dp else 0)
    E.emu_set_gap_caps(int(gaptab_cap), int(gap_mcap)); E.emu_set_wave_jobs(1 if wave_jobs else 0); E.emu_set_wj_small(int(wj_small)); E.emu_set_frag_block_min(int(frag_block_min)); E.emu_stat_reset(); E.emu_stat.restype = C.c_longlong
    n = batch.n_reads
    hb = hp_batch_struct(batch, HpBatch)
    hr = HpRef(batch.pac.ctypes.data, int(batch.l_pac), len(batch.seq_len), batch.seq_off.ctypes.data, batch.seq_len.ctypes.data)
    cap = 4096 + 64 * n + 16 * int(batch.read_off[-1]) * scale
    stream = np.zeros(cap, np.int32); nw = C.c_int64(0)
    off = np.zeros(max(n, 1), np.int64); ln = np.zeros(max(n, 1), np.int32); st = np.zeros(max(n, 1), np.int32)
    E.emu_align_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    E.emu_align_batch(C.byref(hp_para), C.byref(hr), C.byref(hb), scale, slab_bytes, stream.ctypes.data, cap, C.byref(nw), off.ctypes.data, ln.ctypes.data, st.ctypes.data)
    if stats is not None:
        stats[:] = [int(E.emu_stat(i)) for i in range(32)]
    E.emu_set_gap_caps(0, 0); E.emu_set_wave_jobs(1); E.emu_set_wj_small(0); E.emu_set_frag_block_min(0)
    return split_streams(stream, off[:n], ln[:n]), st[:n].copy()


def hp_batch_struct(batch, HpBatch):
    b = HpBatch()
    b.n_reads = batch.n_reads
    for name in ("read_off", "read_seq", "seed_all", "last_len", "seed_off", "seed_id", "hit_off", "h_pos", "h_chr", "h_strand",
                 "h_nm", "h_len_dif", "h_cig_off", "h_cig_n", "cig", "cig8"):
        if getattr(batch, name, None) is None:              # the compact form of the boundary (lamsa_amd.hp.compact_batch)
            setattr(b, name, None)
            continue
        a = np.ascontiguousarray(getattr(batch, name)); setattr(batch, name, a); setattr(b, name, a.ctypes.data)
    b.n_cig = len(batch.cig8) if getattr(batch, "cig8", None) is not None else len(batch.cig)
    return b
