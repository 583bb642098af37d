"""ctypes binding of include/lamsa_hp.h (the C-ABI of lib/liblamsa_hp.so).

Mirrors the reference's worker interface as the survey draws it:
create(para, ref) / dp_batch / align_batch / destroy.  Fails loudly when the HIP library
is missing -- there is deliberately no fallback.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "liblamsa_hp.so")

_I32 = ("seed_len", "seed_step", "seed_inv", "per_aln_m", "first_loci_thd", "SV_len_thd", "ske_max")
_I32b = ("bwt_seed_len", "bwt_max_len", "bwt_min_len", "split_len", "split_pen", "res_mul_max",
         "hash_len", "hash_key_len", "hash_step", "hash_size", "match_dis", "mismatch_thd",
         "ins_gapo", "ins_gape", "del_gapo", "del_gape", "ins_ext_o", "ins_ext_e", "del_ext_o", "del_ext_e",
         "match", "mis", "band_w", "end_bonus", "zdrop")


class HpPara(C.Structure):
    """struct lamsa_hp_para"""
    _fields_ = [(n, C.c_int32) for n in _I32] + [("ovlp_rat", C.c_float)] + [(n, C.c_int32) for n in _I32b] + \
               [("id_rate", C.c_float), ("read_type", C.c_int32), ("aln_mode", C.c_int32)]


class HpRef(C.Structure):
    _fields_ = [("pac", C.c_void_p), ("l_pac", C.c_int64), ("n_seqs", C.c_int32),
                ("seq_offset", C.c_void_p), ("seq_len", C.c_void_p)]


class HpDpJobs(C.Structure):
    _fields_ = [("n_jobs", C.c_int32), ("seq", C.c_void_p), ("seq_bytes", C.c_int64),
                ("q_off", C.c_void_p), ("qlen", C.c_void_p), ("t_off", C.c_void_p), ("tlen", C.c_void_p),
                ("kind", C.c_void_p), ("w", C.c_void_p), ("h0", C.c_void_p)]


class HpDpOut(C.Structure):
    _fields_ = [("score", C.POINTER(C.c_int32)), ("qle", C.POINTER(C.c_int32)), ("tle", C.POINTER(C.c_int32)),
                ("status", C.POINTER(C.c_int32)), ("cig_off", C.POINTER(C.c_int64)), ("cigar", C.POINTER(C.c_int32))]


class HpBatch(C.Structure):
    """struct lamsa_hp_batch"""
    _fields_ = [("n_reads", C.c_int32)] + [(n, C.c_void_p) for n in (
        "read_off", "read_seq", "seed_all", "last_len", "seed_off", "seed_id", "hit_off", "h_pos", "h_chr", "h_strand",
        "h_nm", "h_len_dif", "h_cig_off", "h_cig_n", "cig")] + [("n_cig", C.c_int64), ("cig8", C.c_void_p)]


class HpResult(C.Structure):
    """struct lamsa_hp_result"""
    _fields_ = [("stream", C.POINTER(C.c_int32)), ("stream_words", C.c_int64), ("read_off", C.POINTER(C.c_int64)),
                ("read_len", C.POINTER(C.c_int32)), ("read_status", C.POINTER(C.c_int32)), ("read_tbases", C.POINTER(C.c_int32)), ("read_work", C.POINTER(C.c_int32))]


EXPORTS = ("lamsa_hp_para_init", "lamsa_hp_para_finish", "lamsa_hp_create", "lamsa_hp_destroy",
           "lamsa_hp_last_error", "lamsa_hp_dp_batch", "lamsa_hp_last_kernel_ms", "lamsa_hp_set_scratch_limit",
           "lamsa_hp_align_batch", "lamsa_hp_upload_batch", "lamsa_hp_run_uploaded",
           "lamsa_hp_submit_batch", "lamsa_hp_collect_batch", "lamsa_hp_host_alloc", "lamsa_hp_host_free",
           "lamsa_hp_start_uploaded", "lamsa_hp_finish_uploaded", "lamsa_hp_reserve")

_lib = None


_lib_path = None


def loaded_library():
    """(path, sha256 of the file) of the library load_library has loaded -- bench.py puts both into its JSON line."""
    import hashlib
    if _lib_path is None:
        return None, None
    return _lib_path, hashlib.sha256(open(_lib_path, "rb").read()).hexdigest()


def load_library(path=None):
    """Load liblamsa_hp.so; raises if it has not been built (no fallback).  Without an explicit path, LAMSA_HP_LIB may name another
    build of the same library (diagnostic builds with cycle counters: make -C lamsa_amd/csrc OUT=... EXTRA=-DHP_PROF)."""
    global _lib, _lib_path
    if path is None:
        path = os.environ.get("LAMSA_HP_LIB", LIB_PATH)
    if _lib is None:
        _lib_path = os.path.abspath(path)
        if not os.path.exists(path):
            raise RuntimeError("HIP hot-path library missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % path)
        L = C.CDLL(path)
        L.lamsa_hp_para_init.argtypes = [C.POINTER(HpPara)]
        L.lamsa_hp_para_finish.argtypes = [C.POINTER(HpPara)]
        L.lamsa_hp_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(HpPara), C.POINTER(HpRef), C.c_int]
        L.lamsa_hp_create.restype = C.c_int
        L.lamsa_hp_destroy.argtypes = [C.c_void_p]
        L.lamsa_hp_last_error.argtypes = [C.c_void_p]
        L.lamsa_hp_last_error.restype = C.c_char_p
        L.lamsa_hp_dp_batch.argtypes = [C.c_void_p, C.POINTER(HpDpJobs), C.POINTER(HpDpOut)]
        L.lamsa_hp_dp_batch.restype = C.c_int
        L.lamsa_hp_align_batch.argtypes = [C.c_void_p, C.POINTER(HpBatch), C.POINTER(HpResult)]
        L.lamsa_hp_align_batch.restype = C.c_int
        L.lamsa_hp_upload_batch.argtypes = [C.c_void_p, C.POINTER(HpBatch)]
        L.lamsa_hp_upload_batch.restype = C.c_int
        L.lamsa_hp_run_uploaded.argtypes = [C.c_void_p, C.POINTER(HpResult)]
        L.lamsa_hp_run_uploaded.restype = C.c_int
        L.lamsa_hp_last_kernel_ms.argtypes = [C.c_void_p, C.c_int]
        L.lamsa_hp_last_kernel_ms.restype = C.c_float
        L.lamsa_hp_set_scratch_limit.argtypes = [C.c_void_p, C.c_size_t]
        L.lamsa_hp_set_scratch_limit.restype = C.c_int
        L.lamsa_hp_submit_batch.argtypes = [C.c_void_p, C.POINTER(HpBatch)]
        L.lamsa_hp_submit_batch.restype = C.c_int
        L.lamsa_hp_collect_batch.argtypes = [C.c_void_p, C.POINTER(HpResult)]
        L.lamsa_hp_collect_batch.restype = C.c_int
        L.lamsa_hp_host_alloc.argtypes = [C.c_size_t]
        L.lamsa_hp_host_alloc.restype = C.c_void_p
        L.lamsa_hp_host_free.argtypes = [C.c_void_p]
        L.lamsa_hp_start_uploaded.argtypes = [C.c_void_p]
        L.lamsa_hp_start_uploaded.restype = C.c_int
        L.lamsa_hp_finish_uploaded.argtypes = [C.c_void_p, C.POINTER(HpResult)]
        L.lamsa_hp_finish_uploaded.restype = C.c_int
        _lib = L
    return _lib


READ_TYPES = {"default": 0, "pacbio": 1, "ont2d": 2}


def make_para(read_type="default", **over):
    L = load_library()
    P = HpPara()
    L.lamsa_hp_para_init(C.byref(P))
    P.read_type = READ_TYPES[read_type]
    for k, v in over.items():
        setattr(P, k, v)
    L.lamsa_hp_para_finish(C.byref(P))
    return P


def pack_jobs(jobs):
    """jobs: list of (query u8 array, target u8 array) -> (seq, q_off, qlen, t_off, tlen)."""
    n = len(jobs)
    q_off = np.zeros(n, np.int64); t_off = np.zeros(n, np.int64)
    qlen = np.zeros(n, np.int32); tlen = np.zeros(n, np.int32)
    parts, pos = [], 0
    for i, (q, t) in enumerate(jobs):
        q_off[i] = pos; qlen[i] = len(q); parts.append(q); pos += len(q)
        t_off[i] = pos; tlen[i] = len(t); parts.append(t); pos += len(t)
    seq = np.concatenate(parts + [np.zeros(16, np.uint8)]).astype(np.uint8) if parts else np.zeros(16, np.uint8)
    return seq, q_off, qlen, t_off, tlen


class LamsaHp:
    """One handle per GPU (not thread-safe), like one worker's thread_aux_t in the reference."""

    def __init__(self, para, ref=None, device=0):
        self.L = load_library()
        self.para = para
        self._h = C.c_void_p()
        self._keep = None
        r = None
        if ref is not None:
            pac, l_pac, seq_off, seq_len = ref
            pac = np.ascontiguousarray(pac, np.uint8); seq_off = np.ascontiguousarray(seq_off, np.int64)
            seq_len = np.ascontiguousarray(seq_len, np.int32)
            self._keep = (pac, seq_off, seq_len)
            r = HpRef(pac.ctypes.data, int(l_pac), len(seq_len), seq_off.ctypes.data, seq_len.ctypes.data)
        rc = self.L.lamsa_hp_create(C.byref(self._h), C.byref(para), C.byref(r) if r is not None else None, device)
        if rc != 0:
            raise RuntimeError("lamsa_hp_create failed: %d (no usable MI355X / HIP device?)" % rc)

    def close(self):
        if self._h:
            self.L.lamsa_hp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scratch_limit(self, nbytes):
        """Cap the first pass's per-wave scratch (0 = automatic); reads that do not fit go to the 8x retry pass."""
        rc = self.L.lamsa_hp_set_scratch_limit(self._h, int(nbytes))
        if rc != 0:
            raise RuntimeError("lamsa_hp_set_scratch_limit failed: %d" % rc)

    def last_kernel_ms(self, which=0):
        return float(self.L.lamsa_hp_last_kernel_ms(self._h, which))

    def dp_batch(self, jobs, kind, w, h0):
        """Run DP jobs; returns dict(score, qle, tle, status, cigars=list of lists)."""
        seq, q_off, qlen, t_off, tlen = pack_jobs(jobs)
        n = len(jobs)
        kind = np.ascontiguousarray(np.broadcast_to(kind, n), np.int32)
        w = np.ascontiguousarray(np.broadcast_to(w, n), np.int32)
        h0 = np.ascontiguousarray(np.broadcast_to(h0, n), np.int32)
        J = HpDpJobs(n, seq.ctypes.data, int(len(seq)), q_off.ctypes.data, qlen.ctypes.data, t_off.ctypes.data,
                     tlen.ctypes.data, kind.ctypes.data, w.ctypes.data, h0.ctypes.data)
        O = HpDpOut()
        rc = self.L.lamsa_hp_dp_batch(self._h, C.byref(J), C.byref(O))
        if rc != 0:
            raise RuntimeError("lamsa_hp_dp_batch: %d %s" % (rc, self.L.lamsa_hp_last_error(self._h).decode()))
        score = np.ctypeslib.as_array(O.score, (n,)).copy() if n else np.zeros(0, np.int32)
        qle = np.ctypeslib.as_array(O.qle, (n,)).copy() if n else np.zeros(0, np.int32)
        tle = np.ctypeslib.as_array(O.tle, (n,)).copy() if n else np.zeros(0, np.int32)
        st = np.ctypeslib.as_array(O.status, (n,)).copy() if n else np.zeros(0, np.int32)
        off = np.ctypeslib.as_array(O.cig_off, (n + 1,)).copy()
        tot = int(off[n])
        cg = np.ctypeslib.as_array(O.cigar, (max(tot, 1),)).copy()
        cigars = [cg[off[i]:off[i + 1]].tolist() for i in range(n)]
        return dict(score=score, qle=qle, tle=tle, status=st, cigars=cigars)

    # ---- the hot path proper
    def _batch_struct(self, batch):
        b = HpBatch()
        b.n_reads = batch.n_reads
        self._keep_batch = []
        for name in ("read_off", "read_seq", "seed_all", "last_len", "seed_off", "seed_id", "hit_off", "h_pos", "h_chr", "h_strand",
                     "h_nm", "h_len_dif", "h_cig_off", "h_cig_n", "cig", "cig8"):
            v = getattr(batch, name, None)
            if v is None:                                   # optional: h_cig_off (CIGARs back to back), cig / cig8 (one of them)
                setattr(b, name, None)
                continue
            a = np.ascontiguousarray(v); self._keep_batch.append(a); setattr(b, name, a.ctypes.data)
        b.n_cig = len(batch.cig8) if getattr(batch, "cig8", None) is not None else len(batch.cig)
        return b

    def _result(self, R, n):
        if n == 0:
            return [], np.zeros(0, np.int32)
        off = np.ctypeslib.as_array(R.read_off, (max(n, 1),))[:n].copy()
        ln = np.ctypeslib.as_array(R.read_len, (max(n, 1),))[:n].copy()
        st = np.ctypeslib.as_array(R.read_status, (max(n, 1),))[:n].copy()
        stream = np.ctypeslib.as_array(R.stream, (max(int(R.stream_words), 1),)).copy()
        self.last_tbases = np.ctypeslib.as_array(R.read_tbases, (max(n, 1),))[:n].copy()
        self.last_work = np.ctypeslib.as_array(R.read_work, (max(4 * n, 4),))[:4 * n].copy()
        self.last_stream_words = int(R.stream_words)
        return [stream[int(off[i]):int(off[i]) + int(ln[i])].tolist() for i in range(n)], st

    def align_batch(self, batch):
        """batch: object with the numpy arrays of lamsa_hp_batch. Returns (per-read result streams, status array)."""
        b = self._batch_struct(batch)
        R = HpResult()
        rc = self.L.lamsa_hp_align_batch(self._h, C.byref(b), C.byref(R))
        if rc != 0:
            raise RuntimeError("lamsa_hp_align_batch: %d %s" % (rc, self.L.lamsa_hp_last_error(self._h).decode()))
        return self._result(R, batch.n_reads)

    def upload_batch(self, batch):
        b = self._batch_struct(batch)
        rc = self.L.lamsa_hp_upload_batch(self._h, C.byref(b))
        if rc != 0:
            raise RuntimeError("lamsa_hp_upload_batch: %d %s" % (rc, self.L.lamsa_hp_last_error(self._h).decode()))
        self._n_up = batch.n_reads

    def run_uploaded(self, fetch=True, raw=False):
        """Align the resident batch.  fetch=False leaves the results on the device; raw=True returns the numpy
        views (stream, read_off, read_len, status) without splitting them into per-read lists."""
        R = HpResult()
        rc = self.L.lamsa_hp_run_uploaded(self._h, C.byref(R) if fetch else None)
        if rc != 0:
            raise RuntimeError("lamsa_hp_run_uploaded: %d %s" % (rc, self.L.lamsa_hp_last_error(self._h).decode()))
        if not fetch:
            return None
        if raw:
            n = self._n_up
            self.last_tbases = np.ctypeslib.as_array(R.read_tbases, (max(n, 1),))[:n]
            self.last_work = np.ctypeslib.as_array(R.read_work, (max(4 * n, 4),))[:4 * n]
            self.last_stream_words = int(R.stream_words)
            return (np.ctypeslib.as_array(R.stream, (max(int(R.stream_words), 1),)), np.ctypeslib.as_array(R.read_off, (max(n, 1),))[:n],
                    np.ctypeslib.as_array(R.read_len, (max(n, 1),))[:n], np.ctypeslib.as_array(R.read_status, (max(n, 1),))[:n])
        return self._result(R, self._n_up)

    def start_uploaded(self):
        """Queue a run of the resident batch (two may be in flight)."""
        rc = self.L.lamsa_hp_start_uploaded(self._h)
        if rc != 0:
            raise RuntimeError("lamsa_hp_start_uploaded: %d %s" % (rc, self.L.lamsa_hp_last_error(self._h).decode()))

    def finish_uploaded(self, fetch=True, raw=False):
        """Wait for the oldest run of the resident batch; results as run_uploaded returns them."""
        R = HpResult()
        rc = self.L.lamsa_hp_finish_uploaded(self._h, C.byref(R) if fetch else None)
        if rc != 0:
            raise RuntimeError("lamsa_hp_finish_uploaded: %d %s" % (rc, self.L.lamsa_hp_last_error(self._h).decode()))
        if not fetch:
            return None
        n = self._n_up
        if raw:
            self.last_tbases = np.ctypeslib.as_array(R.read_tbases, (max(n, 1),))[:n]
            self.last_work = np.ctypeslib.as_array(R.read_work, (max(4 * n, 4),))[:4 * n]
            self.last_stream_words = int(R.stream_words)
            return (np.ctypeslib.as_array(R.stream, (max(int(R.stream_words), 1),)), np.ctypeslib.as_array(R.read_off, (max(n, 1),))[:n],
                    np.ctypeslib.as_array(R.read_len, (max(n, 1),))[:n], np.ctypeslib.as_array(R.read_status, (max(n, 1),))[:n])
        return self._result(R, n)

    # ---- streaming form: up to two batches in flight
    def submit_batch(self, batch):
        """Upload `batch` behind the running kernel and queue its kernel; returns once the copy is done."""
        b = self._batch_struct(batch)
        rc = self.L.lamsa_hp_submit_batch(self._h, C.byref(b))
        if rc != 0:
            raise RuntimeError("lamsa_hp_submit_batch: %d %s" % (rc, self.L.lamsa_hp_last_error(self._h).decode()))
        self._n_fly = getattr(self, "_n_fly", []) + [batch.n_reads]

    def collect_batch(self, raw=False):
        """Results of the oldest submitted batch, as align_batch returns them (raw=True: numpy views, see run_uploaded)."""
        R = HpResult()
        rc = self.L.lamsa_hp_collect_batch(self._h, C.byref(R))
        if rc != 0:
            raise RuntimeError("lamsa_hp_collect_batch: %d %s" % (rc, self.L.lamsa_hp_last_error(self._h).decode()))
        n = self._n_fly.pop(0)
        if raw:
            self.last_stream_words = int(R.stream_words)
            return (np.ctypeslib.as_array(R.stream, (max(int(R.stream_words), 1),)), np.ctypeslib.as_array(R.read_off, (max(n, 1),))[:n],
                    np.ctypeslib.as_array(R.read_len, (max(n, 1),))[:n], np.ctypeslib.as_array(R.read_status, (max(n, 1),))[:n])
        return self._result(R, n)


def compact_batch(batch):
    """The compact form of the boundary for a batch whose seed CIGARs lie back to back in hit order: one byte per CIGAR
    element (cig8) and no offsets at all (h_cig_off = NULL) -- a quarter of the CIGAR bytes and 4 bytes per hit less over
    PCIe, and no 2^31 limit on the elements of a batch.  Raises when the batch is not laid out that way or an element is
    longer than 63."""
    nh = int(batch.hit_off[-1]) if len(batch.hit_off) else 0
    cn = np.asarray(batch.h_cig_n[:nh], np.int64)
    n_cig = int(cn.sum())
    # (with more than 2^31-1 elements the 32-bit offsets of the word form cannot say anything any more)
    if nh and n_cig <= 0x7fffffff and not np.array_equal(np.asarray(batch.h_cig_off[:nh], np.int64), np.concatenate([[0], np.cumsum(cn)[:-1]])):
        raise ValueError("seed CIGARs are not back to back in hit order")
    w = np.asarray(batch.cig[:n_cig])
    if n_cig and (int((w >> 4).max()) > 63 or int((w & 0xf).max()) > 2):
        raise ValueError("a seed CIGAR element does not fit one byte")

    class Compact:
        pass
    out = Compact()
    for k, v in vars(batch).items():
        setattr(out, k, v)
    out.h_cig_off = None; out.cig = None
    out.cig8 = (((w & 0xf) << 6) | (w >> 4)).astype(np.uint8) if n_cig else np.zeros(4, np.uint8)
    if n_cig == 0:
        out.cig8 = out.cig8[:0]
    return out


def pinned_batch(batch):
    """Copy of `batch` (any object with the lamsa_hp_batch arrays) whose arrays live in page-locked host memory
    (lamsa_hp_host_alloc): uploads from it run at the PCIe rate.  Keep the returned object alive while it is in use;
    `release()` frees the memory."""
    L = load_library()

    class Pinned:
        pass
    out = Pinned(); out._ptrs = []; out.n_reads = batch.n_reads
    for name in ("read_off", "read_seq", "seed_all", "last_len", "seed_off", "seed_id", "hit_off", "h_pos", "h_chr", "h_strand",
                 "h_nm", "h_len_dif", "h_cig_off", "h_cig_n", "cig", "cig8"):
        if getattr(batch, name, None) is None:
            setattr(out, name, None)
            continue
        a = np.ascontiguousarray(getattr(batch, name))
        p = L.lamsa_hp_host_alloc(max(a.nbytes, 1))
        if not p:
            raise RuntimeError("lamsa_hp_host_alloc(%d) failed" % a.nbytes)
        out._ptrs.append(p)
        v = np.ctypeslib.as_array((C.c_uint8 * max(a.nbytes, 1)).from_address(p))[:a.nbytes].view(a.dtype)
        v[...] = a.reshape(-1)
        setattr(out, name, v)

    def release():
        for p in out._ptrs:
            L.lamsa_hp_host_free(p)
        out._ptrs = []
    out.release = release
    return out
