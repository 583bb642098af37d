"""CPU-side checks of the banded-DP rows:
the oracle against the reference's golden vectors (and against the reference itself when
oracle/_ref is present), and the device kernel sources, compiled with the CPU lane
emulation, against the oracle.  Integer work: everything is bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

import dpjobs
import goldenlib
import reflib

ERR = {"default": (0.01, 0.01, 0.01), "pacbio": (0.015, 0.09, 0.045), "ont2d": (0.04, 0.04, 0.04)}


def test_oracle_matches_golden_dp_vectors():
    n = 0
    for rt, jobs, calls in goldenlib.dp_vectors():
        P = reflib.lo_para(rt)
        for kind, w, h0, exp in calls:
            got = reflib.oracle_dp(jobs, P, kind, w, h0)
            assert goldenlib.same_dp(exp, got, kind) == [], (rt, kind, w, h0)
            n += len(jobs)
    assert n > 2000


@pytest.mark.skipif(reflib.ref() is None, reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("rt", ["default", "pacbio", "ont2d"])
def test_oracle_matches_reference_live(rt):
    lp, rp = reflib.lo_para(rt), reflib.ref_para(rt)
    jobs = dpjobs.make_jobs(900 + len(rt), 150, 350, ERR[rt])
    for kind, w, h0 in ((0, lp.band_w, 0), (1, lp.band_w, 50), (1, 300, 19), (2, 0, 100)):
        assert goldenlib.same_dp(reflib.ref_dp(jobs, rp, kind, w, h0), reflib.oracle_dp(jobs, lp, kind, w, h0), kind) == []


def test_params_match_reference_presets():
    # values of src/lamsa_aln.c:1342-1420 / lamsa_aln.h:15-90
    d, p, o = reflib.lo_para("default"), reflib.lo_para("pacbio"), reflib.lo_para("ont2d")
    assert (d.seed_step, d.seed_len, d.band_w, d.match_dis, d.mis, d.ins_gapo, d.end_bonus, d.aln_mode) == (100, 50, 10, 5, 3, 5, 5, 0)
    assert (p.seed_step, p.band_w, p.match_dis, p.ins_ext_o, p.hash_len, p.hash_step, p.aln_mode) == (25, 200, 8, 2, 8, 4, 3)
    assert (o.seed_step, o.band_w, o.match_dis, o.ins_ext_o, o.bwt_min_len, o.aln_mode) == (25, 100, 3, 1, 100, 3)
    assert list(d.sc_mat)[:6] == [1, -3, -3, -3, -1, -3] and list(d.sc_mat)[20:] == [-1] * 5


def hp_para_like(rt, **over):
    """lamsa_hp_para filled without the HIP library (host logic mirror for the emulation tests)."""
    from lamsa_amd.hp import HpPara
    lp = reflib.lo_para(rt, **over)
    P = HpPara()
    for name, _ in HpPara._fields_:
        setattr(P, name, getattr(lp, name))
    return P


@pytest.mark.parametrize("rt", ["default", "pacbio", "ont2d"])
def test_emulated_kernels_match_oracle(rt):
    lp, P = reflib.lo_para(rt), hp_para_like(rt)
    jobs = dpjobs.make_jobs(31 + len(rt), 120, 330, ERR[rt])
    for kind, w, h0 in ((0, lp.band_w, 0), (0, 3, 0), (1, lp.band_w, 50), (1, lp.band_w, 8), (2, 0, 100), (2, 0, lp.hash_len * lp.match)):
        assert goldenlib.same_dp(reflib.oracle_dp(jobs, lp, kind, w, h0), reflib.emu_dp(jobs, P, kind, w, h0), kind) == []


def _mid_jobs(seed, err):
    """Queries of 58 .. 260 bases around every length near the limits of the register routines (62 | 63 .. 126 | 127 .. 190 | 191 .. 254
    | 255), targets from half to twice as long."""
    rng = np.random.default_rng(seed)
    jobs = []
    for ql in (list(range(58, 70)) + list(range(120, 133)) + list(range(186, 196)) + list(range(250, 260)) + [int(x) for x in rng.integers(63, 127, 40)]
               + [int(x) for x in rng.integers(127, 255, 50)]):
        t = rng.integers(0, 4, size=int(ql * rng.uniform(0.5, 2.0)), dtype=np.uint8)
        q = dpjobs.mutate(rng, t, *err)
        q = np.concatenate([q, rng.integers(0, 4, size=max(0, ql - len(q)), dtype=np.uint8)])[:ql]
        jobs.append((np.ascontiguousarray(q, np.uint8), np.ascontiguousarray(t, np.uint8)))
    return jobs


@pytest.mark.parametrize("rt", ["default", "pacbio", "ont2d"])
def test_emulated_extension_with_the_row_in_register_sets(rt):
    """ksw_extend_core for queries of 63 .. 254 bases (ksw_extend_regn<2 | 3 | 4>: column j in lane j & 63 of register set j >> 6):
    extension and two-sided extension, full and narrow bands (the band edge crosses from one set to the next), small and large h0."""
    lp, P = reflib.lo_para(rt), hp_para_like(rt)
    jobs = _mid_jobs(77 + len(rt), ERR[rt])
    for kind, w, h0 in ((1, lp.band_w, 50), (1, 5, 30), (1, 40, 7), (1, 70, 200), (1, 130, 90), (2, 0, 100), (2, 0, 12)):
        want = reflib.oracle_dp(jobs, lp, kind, w, h0)
        for pk in (True, False):              # the window of int16 pairs (ksw_extend_band<1 | 2>: the whole query fits it), then the int32 sets behind it
            st = []
            assert goldenlib.same_dp(want, reflib.emu_dp(jobs, P, kind, w, h0, pk=pk, stats=st), kind) == [], (kind, w, h0, pk)
            assert (st[4] > 0 and st[1] == 0) if pk else (st[4] == 0 and st[1] > 0), st
    for w in (lp.band_w, 3, 40):              # ksw_global2 on the same jobs: ksw_global_pk<1 | 2>, then the LDS rows behind it
        want = reflib.oracle_dp(jobs, lp, 0, w, 0)
        for pk in (True, False):
            st = []
            assert goldenlib.same_dp(want, reflib.emu_dp(jobs, P, 0, w, 0, pk=pk, stats=st), 0) == [], (w, pk)
            assert (st[2] > 0 and st[3] > 0) if pk else (st[2] == 0 and st[3] > 0), st          # (queries beyond 254 bases keep the LDS rows)


def test_emulated_packed_extension_refuses_scores_beyond_int16():
    """A start score that would take a cell past the int16 range sends the job to the int32 register sets (pkb_extend_ok, hp_ksw.h)."""
    lp, P = reflib.lo_para("ont2d"), hp_para_like("ont2d")
    jobs = _mid_jobs(5, ERR["ont2d"])[:40]
    mx = max(P.match, P.mis)
    qmax = max(len(q) for q, _ in jobs)
    for h0, packed in ((14000, True), (22990 - qmax * mx, True), (23000, False), (30000, False), (1 << 20, False)):
        st = []
        got = reflib.emu_dp(jobs, P, 1, lp.band_w, h0, stats=st)
        assert goldenlib.same_dp(reflib.oracle_dp(jobs, lp, 1, lp.band_w, h0), got, 1) == [], h0
        assert (st[4] > 0 and st[1] == 0) if packed else (st[4] == 0 and st[1] > 0), (h0, st)


def test_emulated_kernels_edge_cases():
    lp, P = reflib.lo_para("ont2d"), hp_para_like("ont2d")
    rng = np.random.default_rng(5)
    e = np.zeros(0, np.uint8)
    t200 = rng.integers(0, 4, 200, dtype=np.uint8)
    jobs = [(e, e), (e, t200[:7]), (t200[:7], e), (t200[:1], t200[:1]), (t200, t200), (np.full(30, 4, np.uint8), t200[:30]),
            (t200[:130], t200[:64]), (t200[:64], t200[:65]), (t200[:129], t200[:128])]
    for kind, w, h0 in ((0, 100, 0), (0, 1, 0), (1, 100, 50), (1, 2, 1), (2, 0, 100)):
        assert goldenlib.same_dp(reflib.oracle_dp(jobs, lp, kind, w, h0), reflib.emu_dp(jobs, P, kind, w, h0), kind) == []


@pytest.mark.parametrize("rt,over", [("default", {}), ("pacbio", {}), ("ont2d", {}), ("default", {"end_bonus": 40}),
                                     ("ont2d", {"del_ext_o": 30, "del_ext_e": 9}), ("pacbio", {"end_bonus": 0, "ins_ext_o": 9})])
def test_emulated_bi_extend_with_empty_query(rt, over):
    """ksw_bi_extend with nothing to align on the query side (overlapping neighbour seeds) takes a shortcut in the
    kernels; the oracle runs the full extension.  Also checked against the compiled reference when present."""
    lp, P = reflib.lo_para(rt, **over), hp_para_like(rt, **over)
    rng = np.random.default_rng(3)
    e = np.zeros(0, np.uint8)
    jobs = [(e, rng.integers(0, 5, n, dtype=np.uint8)) for n in (0, 1, 2, 3, 5, 17, 60, 64, 65, 300)]
    for h0 in (1, 2, 3, 8, 10, 50, 100):
        want = reflib.oracle_dp(jobs, lp, 2, 0, h0)
        assert goldenlib.same_dp(want, reflib.emu_dp(jobs, P, 2, 0, h0), 2) == []
        if reflib.ref() is not None:
            assert goldenlib.same_dp(reflib.ref_dp(jobs, reflib.ref_para(rt, **over), 2, 0, h0), want, 2) == []


def test_emulated_wide_band_uses_hbm_rows():
    """Bands wider than the LDS row (w > 222) take the HBM-row variants of both routines."""
    lp, P = reflib.lo_para("default"), hp_para_like("default")
    rng = np.random.default_rng(12)
    t = rng.integers(0, 4, 1500, dtype=np.uint8)
    q = dpjobs.mutate(rng, t[:1200], 0.02, 0.02, 0.02)
    jobs = [(q, t), (q[:700], t[:900]), (q[:300], t[:1300])]
    for kind, w, h0 in ((0, 400, 0), (1, 600, 5000), (1, 300, 20000)):
        assert goldenlib.same_dp(reflib.oracle_dp(jobs, lp, kind, w, h0), reflib.emu_dp(jobs, P, kind, w, h0), kind) == [], (kind, w)


def test_emulated_kernels_long_extension():
    # the rare-but-huge job of frag_head/tail_bound_fix: query ~ a whole read
    lp, P = reflib.lo_para("ont2d"), hp_para_like("ont2d")
    rng = np.random.default_rng(9)
    t = rng.integers(0, 4, 3100, dtype=np.uint8)
    q = dpjobs.mutate(rng, t[:3000], 0.04, 0.04, 0.04)
    jobs = [(q, t)]
    for kind, w, h0 in ((1, 100, 50), (2, 0, 100)):
        assert goldenlib.same_dp(reflib.oracle_dp(jobs, lp, kind, w, h0), reflib.emu_dp(jobs, P, kind, w, h0), kind) == []


def test_cabi_library_exports_declared_symbols():
    """The product library loads and exports every symbol include/lamsa_hp.h declares (no compute without a GPU)."""
    import re
    from lamsa_amd import hp
    hdr = open(os.path.join(reflib.ROOT, "include", "lamsa_hp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)          # declarations only, not prose
    declared = set(re.findall(r"\b(lamsa_hp_[a-z0-9_]+)\s*\(", hdr))
    L = hp.load_library()
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), name
    # host logic of the boundary: parameter presets agree with the oracle's
    for rt in ("default", "pacbio", "ont2d"):
        a, b = hp.make_para(rt), reflib.lo_para(rt)
        for name, _ in hp.HpPara._fields_:
            assert getattr(a, name) == getattr(b, name), (rt, name)


@pytest.mark.parametrize("preset", ["default", "pacbio", "ont2d"])
def test_lane_per_job_dp_matches_oracle(preset):
    """hp_lanedp.h (one small DP job per lane, what the k_filldp launch runs for the junctions of a line) against the oracle:
    ksw_global2, ksw_extend_core and ksw_bi_extend on ragged jobs up to the lane buffers' capacity, targets read 2 bits per base."""
    from lamsa_amd.hp import HpPara
    lp = reflib.lo_para(preset)
    P = HpPara()
    for n, _ in HpPara._fields_:
        setattr(P, n, getattr(lp, n))
    jobs = [(q, t) for q, t in dpjobs.make_jobs(4242, 900, 150, (0.05, 0.05, 0.05)) if len(q) <= 160 and len(t) <= 256 and (len(t) == 0 or t.max() < 4)]
    assert len(jobs) > 400
    for kind, w, h0 in ((0, lp.band_w, 0), (0, 7, 0), (1, lp.band_w, 50), (1, 12, 8), (2, 0, 100), (2, 0, 10)):
        want = reflib.oracle_dp(jobs, lp, kind, w, max(h0, 1))
        got = reflib.emu_lane_dp(jobs, P, kind, w, max(h0, 1))
        bad = goldenlib.same_dp(want, got, kind)
        assert bad == [], (kind, w, h0, bad[:5])


@pytest.mark.parametrize("preset", ["default", "pacbio", "ont2d"])
def test_wave_jobs_match_oracle(preset):
    """hp_wavejob.h (the jobs of the wave-per-job launch: a junction's ksw_bi_extend, a seed gap's ksw_global2, a line's head and tail
    extension with their soft clip; sequences staged from the read bytes and the packed reference, the direction matrix in that launch's LDS
    where it fits) against the oracle, on ragged jobs around the lengths where the routines and the matrix's place change (62 / 63, 126 / 127,
    254 / 255 query bases; ~75 and ~150 rows of 64 and 128 bytes in 9.5 KB)."""
    from lamsa_amd.hp import HpPara
    lp = reflib.lo_para(preset)
    P = HpPara()
    for n, _ in HpPara._fields_:
        setattr(P, n, getattr(lp, n))
    jobs = [(q, t) for q, t in dpjobs.make_jobs(778, 500, 300, (0.05, 0.05, 0.05)) if len(t) == 0 or t.max() < 4]
    jobs += [(q, t) for q, t in dpjobs.make_jobs(779, 160, 300, (0.04, 0.04, 0.04), tail_noise=True) if len(t) == 0 or t.max() < 4]
    for ql in (60, 61, 62, 63, 64, 74, 75, 76, 100, 125, 126, 127, 128, 150, 151, 200, 253, 254, 255, 256, 300):
        jobs += [(q[:ql], t[:ql + d]) for (q, t), d in zip(dpjobs.make_jobs(1000 + ql, 6, 400, (0.05, 0.05, 0.05)), (-9, -3, 0, 2, 7, 12)) if len(q) >= ql and t.max() < 4]
    assert len(jobs) > 600
    for h0 in (100, 10):
        want = reflib.oracle_dp(jobs, lp, 2, 0, h0)
        got = reflib.emu_wave_job(jobs, P, 1, 0, h0)
        assert goldenlib.same_dp(want, got, 2) == [], (preset, h0)
    for w in (lp.band_w, 7):
        want = reflib.oracle_dp(jobs, lp, 0, w, 0)
        got = reflib.emu_wave_job(jobs, P, 2, w, 0)
        assert goldenlib.same_dp(want, got, 0) == [], (preset, w)
    for head in (True, False):
        for w, h0 in ((lp.band_w, 50), (12, 8)):
            want = reflib.end_extension_from_oracle(jobs, lp, head, w, h0)
            got = reflib.emu_wave_job(jobs, P, 3 if head else 4, w, h0)
            bad = [i for i in range(len(jobs)) if (want["score"][i], want["qle"][i], want["tle"][i], list(want["cigars"][i])) != (got["score"][i], got["qle"][i], got["tle"][i], list(got["cigars"][i]))]
            assert bad == [] and (got["status"] == 0).all(), (preset, head, w, h0, bad[:5])


def test_no_caller_local_is_handed_by_address_to_a_non_inlined_device_routine(tmp_path):
    """tools/noinl_guard.py over the device sources: every address-of argument at a call of an HP_NOINL routine is one of the accepted
    context structures (tools/noinl_allow.txt); and the guard does see a planted violation."""
    import subprocess
    import sys
    guard = os.path.join(reflib.ROOT, "tools", "noinl_guard.py")
    q = subprocess.run([sys.executable, guard, "--report", os.devnull], capture_output=True, text=True)
    assert q.returncode == 0, q.stderr
    sys.path.insert(0, os.path.join(reflib.ROOT, "tools"))
    import noinl_guard
    src = "HP_NOINL int callee(ReadCtx &r, int *out)\n{\n    return 0;\n}\nHP_FN void caller(ReadCtx &r)\n{\n    int lo = 0, hi;\n    callee(r, &lo);\n}\n"
    t = noinl_guard.strip_comments(src)
    assert "callee" in noinl_guard.noinl_names({"x": t})
    assert "lo" in noinl_guard.local_scalars_before(t, t.index("callee(r, &lo)"))


def _split_cases(rng, hash_len):
    """(read gap, fetched reference bases, window start, window length, ref_offset) as split_mapping hands them to split_indel_map
    (src/frag_check.c:475-545): deletions (window = the fetched bases), duplications (a window inside hash_len - dis more bases on either
    side, ref_offset = -dis), and gaps over tandem repeats and two-copy repeats, where no k-mer of the gap is unique in the window."""
    def mut(s, p):
        out = []
        for c in s:
            x = rng.random()
            if x < p:
                out.append((int(c) + 1 + int(rng.integers(0, 3))) % 4)
            elif x < 2 * p:
                out += [int(c), int(rng.integers(0, 4))]
            elif x >= 3 * p:
                out.append(int(c))
        return np.array(out, np.uint8)
    for it in range(400):
        kind = it % 5
        if kind == 0:      # deletion
            refw = rng.integers(0, 4, int(rng.integers(200, 3000)), dtype=np.uint8)
            a = int(rng.integers(20, len(refw) // 2)); b = int(rng.integers(len(refw) // 2, len(refw) - 20))
            yield mut(np.concatenate([refw[:a], refw[b:]]), 0.01), refw, 0, len(refw), 0
        elif kind == 1:    # a tandem repeat all over: every k-mer of the gap occurs many times
            unit = rng.integers(0, 4, int(rng.integers(2, 7)), dtype=np.uint8)
            refw = np.tile(unit, 120)[:int(rng.integers(150, 500))]
            yield np.tile(unit, 120)[:int(rng.integers(60, len(refw)))], refw, 0, len(refw), 0
        elif kind == 2:    # duplication in the read: the window of the DUP branch
            g = rng.integers(0, 4, 4000, dtype=np.uint8)
            s_tlen = int(rng.integers(2 * 10 + 1, 600)); dup = int(rng.integers(15, 500)); dis = -dup
            gs = 1500; a = int(rng.integers(0, s_tlen)); w = min(dup, s_tlen - a) if s_tlen - a > 0 else 0
            gap = g[gs:gs + s_tlen]
            read = mut(np.concatenate([gap[:a + w], g[gs + a + w - dup:gs + a + w], gap[a + w:]]), 0.01)
            margin = hash_len - dis
            tb = g[gs - margin:gs + s_tlen + margin].copy()
            yield read, tb, margin, s_tlen, (-dis if rng.random() < 0.8 else 0)
        elif kind == 3:    # two copies of a block around the deleted stretch
            blk = rng.integers(0, 4, 60, dtype=np.uint8)
            refw = np.concatenate([rng.integers(0, 4, 80, dtype=np.uint8), blk, rng.integers(0, 4, int(rng.integers(50, 400)), dtype=np.uint8), blk, rng.integers(0, 4, 80, dtype=np.uint8)])
            yield mut(np.concatenate([refw[:100], refw[-100:]]), 0.01), refw, 0, len(refw), 0
        else:              # microsatellites on both flanks
            unit = rng.integers(0, 4, 3, dtype=np.uint8)
            refw = np.concatenate([np.tile(unit, 40), rng.integers(0, 4, int(rng.integers(30, 300)), dtype=np.uint8), np.tile(unit, 40)])
            yield mut(np.concatenate([np.tile(unit, 30), np.tile(unit, 30)]), 0.01), refw, 0, len(refw), 0


@pytest.mark.parametrize("preset", ["default", "ont2d"])
def test_kmer_split_mapper_three_ways(preset):
    """split_indel_map (src/split_mapping.c:829: k-mer index of the window, bucket matching, hash DP, indel CIGAR) on crafted gaps: the
    device code under the lane emulation == the oracle == the REFERENCE itself (oracle/_ref, when built).  The gaps without a unique
    k-mer run the MULTI pass of the hash DP (hp_split.h hmain_line, src/split_mapping.c:570-581), which no read of the corpus reaches."""
    from lamsa_amd.hp import HpPara
    import ctypes
    lp = reflib.lo_para(preset)
    rp = reflib.ref_para(preset) if reflib.ref() is not None else None
    P = HpPara()
    for n, _ in HpPara._fields_:
        setattr(P, n, getattr(lp, n))
    E = reflib.emu(); E.emu_stat_reset(); E.emu_stat.restype = ctypes.c_longlong
    rng = np.random.default_rng(11)
    n = 0
    for read, tb, start, ref_len, off in _split_cases(rng, lp.hash_len):
        if len(read) < lp.hash_len + 2 or ref_len < lp.hash_len + 2:
            continue
        o, r, e = reflib.split_indel_map_three_ways(read, tb, off, lp, rp, P, ref_start=start, ref_len=ref_len)
        assert e[2] == 0 and (e[0], e[1]) == o, ("emulation vs oracle", n, len(read), ref_len, off)
        if r is not None:
            assert r == o, ("oracle vs reference", n, len(read), ref_len, off)
        n += 1
    assert n > 300 and E.emu_stat(13) > 0


def _hp_para_of(lp):
    from lamsa_amd.hp import HpPara
    P = HpPara()
    for n, _ in HpPara._fields_:
        setattr(P, n, getattr(lp, n))
    return P


def _long_jobs(seed, n, lo, hi, err, noise):
    """Long extension jobs: a query of lo .. hi bases against its (mutated) target, some with an unrelated stretch of 0 .. `noise` bases in the
    middle or at the end (z-drop, local end), some shorter or longer than the target."""
    rng = np.random.default_rng(seed)
    jobs = []
    for k in range(n):
        tl = int(rng.integers(lo, hi + 1))
        t = rng.integers(0, 4, size=tl, dtype=np.uint8)
        q = dpjobs.mutate(rng, t, *err, 0.0)
        mode = k % 5
        if mode == 1:
            cut = int(rng.integers(len(q) // 4, len(q)))
            q = np.concatenate([q[:cut], rng.integers(0, 4, size=int(rng.integers(0, noise + 1)), dtype=np.uint8)])
        elif mode == 2:
            a = int(rng.integers(len(q) // 4, 3 * len(q) // 4))
            q = np.concatenate([q[:a], rng.integers(0, 4, size=int(rng.integers(1, noise + 1)), dtype=np.uint8), q[a:]])
        elif mode == 3:
            q = q[:int(rng.integers(256, len(q) + 1))] if len(q) > 256 else q
        elif mode == 4:
            t = t[:int(rng.integers(len(t) // 2, len(t) + 1))]
        jobs.append((np.ascontiguousarray(q, np.uint8), np.ascontiguousarray(t, np.uint8)))
    return jobs


@pytest.mark.parametrize("preset", ["default", "pacbio", "ont2d"])
def test_long_extensions_with_the_window_in_registers(preset):
    """ksw_extend_band (hp_ksw.h): ksw_extend_core for queries of more than 254 bases -- the end extensions of a line -- with the row's live
    window in registers, 2 / 4 / 8 columns per lane for bands up to 53 / 108 / 218; against the oracle on jobs of 255 .. 3 000 bases with
    every band class, with start scores that let the band shrink and grow again, z-drop inside unrelated stretches, queries and targets that
    end first; and the routine is really the one that ran.  With the packed routines off the same jobs take the LDS rows."""
    lp = reflib.lo_para(preset)
    P = _hp_para_of(lp)
    err = {"default": (0.01, 0.01, 0.01), "pacbio": (0.015, 0.09, 0.045), "ont2d": (0.04, 0.04, 0.04)}[preset]
    jobs = _long_jobs(5, 40, 255, 700, err, 400) + _long_jobs(6, 12, 1200, 3000, err, 600)
    for w, h0 in ((lp.band_w, 50), (30, 50), (53, 8), (54, 100), (108, 20), (109, 50), (200, 50), (218, 19), (219, 50), (7, 10)):
        want = reflib.oracle_dp(jobs, lp, 1, w, h0)
        stats = []
        got = reflib.emu_dp(jobs, P, 1, w, h0, stats=stats)
        assert goldenlib.same_dp(want, got, 1) == [], (preset, w, h0)
        assert stats[4] > 0 or w > 218, (w, stats)          # (ksw_extend_core narrows a band that the penalties cannot fill, src/ksw.c:696-704: 219 may still run here)
    stats = []
    got = reflib.emu_dp(jobs, P, 1, lp.band_w, 50, pk=False, stats=stats)
    assert goldenlib.same_dp(reflib.oracle_dp(jobs, lp, 1, lp.band_w, 50), got, 1) == [] and stats[4] == 0
