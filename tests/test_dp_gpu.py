"""GPU parity tests of the batched banded-DP kernels through the C-ABI (lamsa_hp_dp_batch):
HIP result == oracle == golden vectors of the reference, bit for bit."""
import numpy as np
import pytest

import dpjobs
import goldenlib
import reflib

pytestmark = pytest.mark.gpu

ERR = {"default": (0.01, 0.01, 0.01), "pacbio": (0.015, 0.09, 0.045), "ont2d": (0.04, 0.04, 0.04)}


@pytest.fixture(scope="module")
def handles():
    from lamsa_amd import hp
    hs = {rt: hp.LamsaHp(hp.make_para(rt)) for rt in ("default", "pacbio", "ont2d")}
    yield hs
    for h in hs.values():
        h.close()


def test_hip_matches_golden_dp_vectors(handles):
    n = 0
    for rt, jobs, calls in goldenlib.dp_vectors():
        for kind, w, h0, exp in calls:
            got = handles[rt].dp_batch(jobs, kind, w, h0)
            assert goldenlib.same_dp(exp, got, kind) == [], (rt, kind, w, h0)
            n += len(jobs)
    assert n > 2000


@pytest.mark.parametrize("rt", ["default", "pacbio", "ont2d"])
def test_hip_matches_oracle_fuzz(handles, rt):
    lp = reflib.lo_para(rt)
    jobs = dpjobs.make_jobs(4000 + len(rt), 1500, 600, ERR[rt])
    for kind, w, h0 in ((0, lp.band_w, 0), (0, 2, 0), (1, lp.band_w, 50), (1, lp.band_w, lp.hash_len * lp.match), (2, 0, 100)):
        got = handles[rt].dp_batch(jobs, kind, w, h0)
        assert goldenlib.same_dp(reflib.oracle_dp(jobs, lp, kind, w, h0), got, kind) == [], (rt, kind, w, h0)


@pytest.mark.parametrize("rt", ["default", "pacbio", "ont2d"])
def test_hip_extension_with_the_row_in_register_sets(handles, rt):
    """Queries of 58 .. 260 bases (the limits of ksw_extend_reg and ksw_extend_regn<2 | 3 | 4> and everything between), full and narrow bands."""
    import test_dp_cpu
    lp = reflib.lo_para(rt)
    jobs = test_dp_cpu._mid_jobs(77 + len(rt), ERR[rt])
    for kind, w, h0 in ((1, lp.band_w, 50), (1, 5, 30), (1, 40, 7), (1, 70, 200), (1, 130, 90), (2, 0, 100), (2, 0, 12), (0, lp.band_w, 0), (0, 3, 0), (0, 40, 0)):
        got = handles[rt].dp_batch(jobs, kind, w, h0)
        assert goldenlib.same_dp(reflib.oracle_dp(jobs, lp, kind, w, h0), got, kind) == [], (rt, kind, w, h0)


def test_hip_packed_extension_refuses_scores_beyond_int16(handles):
    """Start scores around and beyond what the int16 pairs of ksw_extend_band can hold (pkb_extend_ok, hp_ksw.h): the jobs beyond it take the
    int32 register sets -- same results as the oracle either way."""
    import test_dp_cpu
    lp = reflib.lo_para("ont2d")
    jobs = test_dp_cpu._mid_jobs(5, ERR["ont2d"])[:40]
    for h0 in (14000, 14950, 21500, 22400, 22990, 23000, 30000, 1 << 20):
        got = handles["ont2d"].dp_batch(jobs, 1, lp.band_w, h0)
        assert goldenlib.same_dp(reflib.oracle_dp(jobs, lp, 1, lp.band_w, h0), got, 1) == [], h0


def test_hip_edge_cases(handles):
    lp = reflib.lo_para("ont2d")
    rng = np.random.default_rng(5)
    e = np.zeros(0, np.uint8)
    t200 = rng.integers(0, 4, 200, dtype=np.uint8)
    jobs = [(e, e), (e, t200[:7]), (t200[:7], e), (t200[:1], t200[:1]), (t200, t200), (np.full(30, 4, np.uint8), t200[:30]),
            (t200[:130], t200[:64]), (t200[:64], t200[:65]), (t200[:129], t200[:128])]
    for kind, w, h0 in ((0, 100, 0), (0, 1, 0), (1, 100, 50), (1, 2, 1), (2, 0, 100)):
        got = handles["ont2d"].dp_batch(jobs, kind, w, h0)
        assert goldenlib.same_dp(reflib.oracle_dp(jobs, lp, kind, w, h0), got, kind) == []
    assert handles["ont2d"].dp_batch([], 0, 10, 0)["cigars"] == []


def test_hip_whole_read_extension(handles):
    """Maximum-size jobs: 10 kbp / w=100 and 20 kbp / w=200 extensions (traceback 2 MB / 8 MB)."""
    for rt, L, sub in (("ont2d", 10000, (0.04, 0.04, 0.04)), ("pacbio", 20000, (0.01, 0.09, 0.05))):
        lp = reflib.lo_para(rt)
        rng = np.random.default_rng(L)
        t = rng.integers(0, 4, L + 200, dtype=np.uint8)
        q = dpjobs.mutate(rng, t[:L], *sub)
        jobs = [(q, t), (q[:L // 2], t[:L // 2 + 300])]
        for kind, w, h0 in ((1, lp.band_w, 50), (2, 0, 100)):
            got = handles[rt].dp_batch(jobs, kind, w, h0)
            assert goldenlib.same_dp(reflib.oracle_dp(jobs, lp, kind, w, h0), got, kind) == [], (rt, kind)


def test_hip_bi_extend_with_empty_query(handles):
    """The shortcut for overlapping neighbour seeds (query side empty) against the full extension of the oracle."""
    rng = np.random.default_rng(3)
    e = np.zeros(0, np.uint8)
    jobs = [(e, rng.integers(0, 5, n, dtype=np.uint8)) for n in (0, 1, 2, 3, 5, 17, 60, 64, 65, 300)]
    for rt in ("default", "pacbio", "ont2d"):
        lp = reflib.lo_para(rt)
        for h0 in (1, 2, 3, 8, 10, 50, 100):
            assert goldenlib.same_dp(reflib.oracle_dp(jobs, lp, 2, 0, h0), handles[rt].dp_batch(jobs, 2, 0, h0), 2) == [], (rt, h0)


def test_hip_wide_band_uses_hbm_rows(handles):
    """Bands wider than the LDS row (w > 222) take the HBM-row variants of both routines."""
    lp = reflib.lo_para("default")
    rng = np.random.default_rng(12)
    t = rng.integers(0, 4, 1500, dtype=np.uint8)
    q = dpjobs.mutate(rng, t[:1200], 0.02, 0.02, 0.02)
    jobs = [(q, t), (q[:700], t[:900]), (q[:300], t[:1300])]
    for kind, w, h0 in ((0, 400, 0), (1, 600, 5000), (1, 300, 20000)):
        got = handles["default"].dp_batch(jobs, kind, w, h0)
        assert goldenlib.same_dp(reflib.oracle_dp(jobs, lp, kind, w, h0), got, kind) == [], (kind, w)


def _lane_jobs(seed, err, qmax, tmax):
    return [(q, t) for q, t in dpjobs.make_jobs(seed, 1400, 150, err) if len(q) <= qmax and len(t) <= tmax and (len(t) == 0 or t.max() < 4)]


@pytest.mark.parametrize("rt", ["default", "pacbio", "ont2d"])
def test_hip_lane_per_job_routines_match_oracle(handles, rt):
    """hp_lanedp.h on the MI355X, driven directly: kinds 4 / 5 / 6 of lamsa_hp_dp_batch run ksw_global2 / ksw_extend_core /
    ksw_bi_extend one job per LANE with 16-bit cells in lane-strided LDS -- the routines k_filldp_small runs for the read path, whose
    LDS strides and lane-interleaved direction matrix the CPU emulation cannot see.  Against the oracle, bit for bit."""
    lp = reflib.lo_para(rt)
    jobs = _lane_jobs(900 + len(rt), ERR[rt], 160, 256)
    assert len(jobs) > 600
    for kind, w, h0 in ((0, lp.band_w, 0), (0, 7, 0), (1, lp.band_w, 50), (1, 12, 8), (2, 0, 100), (2, 0, 10)):
        got = handles[rt].dp_batch(jobs, kind + 4, w, max(h0, 1))
        want = reflib.oracle_dp(jobs, lp, kind, w, max(h0, 1))
        assert goldenlib.same_dp(want, got, kind) == [], (rt, kind, w, h0)


@pytest.mark.parametrize("rt", ["default", "pacbio", "ont2d"])
def test_hip_wave_jobs_match_oracle(handles, rt):
    """hp_wavejob.h on the MI355X (kinds 8 .. 11 of lamsa_hp_dp_batch: jobs as the wave-per-job launch of the read path runs them -- k_filldp_wave's
    128 registers and 9.5 KB of LDS, sequences staged from the read bytes and the packed reference, the direction matrix of the packed routines in
    LDS where it fits, which the CPU emulation only models): a junction's ksw_bi_extend, a seed gap's ksw_global2, a line's head and tail
    extension with their soft clip, at every length around the limits of the routines (62 / 63, 126 / 127, 254 / 255 query bases) and of the
    LDS matrix (~150 rows of 64 bytes, ~75 of 128), and whole-read extensions beyond them.  Against the oracle, bit for bit."""
    lp = reflib.lo_para(rt)
    jobs = [(q, t) for q, t in dpjobs.make_jobs(880 + len(rt), 900, 320, ERR[rt]) if len(t) == 0 or t.max() < 4]
    for ql in (60, 61, 62, 63, 64, 74, 75, 76, 100, 125, 126, 127, 128, 150, 151, 152, 200, 253, 254, 255, 256, 300):
        jobs += [(q[:ql], t[:ql + d]) for (q, t), d in zip(dpjobs.make_jobs(2000 + ql, 8, 420, ERR[rt]), (-9, -3, 0, 2, 7, 12, 20, 33)) if len(q) >= ql and t.max() < 4]
    jobs += [(q, t) for q, t in dpjobs.make_jobs(41, 6, 6000, ERR[rt]) if len(q) > 1500 and t.max() < 4]
    assert len(jobs) > 900
    for h0 in (100, 10):
        got = handles[rt].dp_batch(jobs, 8, 0, h0)
        assert goldenlib.same_dp(reflib.oracle_dp(jobs, lp, 2, 0, h0), got, 2) == [], (rt, h0)
    for w in (lp.band_w, 7):
        got = handles[rt].dp_batch(jobs, 9, w, 0)
        assert goldenlib.same_dp(reflib.oracle_dp(jobs, lp, 0, w, 0), got, 0) == [], (rt, w)
    for head in (True, False):
        for w, h0 in ((lp.band_w, 50), (12, 8)):
            want = reflib.end_extension_from_oracle(jobs, lp, head, w, h0)
            got = handles[rt].dp_batch(jobs, 10 if head else 11, w, h0)
            bad = [i for i in range(len(jobs)) if (want["score"][i], want["qle"][i], want["tle"][i], list(want["cigars"][i])) != (got["score"][i], got["qle"][i], got["tle"][i], list(got["cigars"][i]))]
            assert bad == [] and (got["status"] == 0).all(), (rt, head, w, h0, bad[:5])


def test_hip_lane_kinds_fall_back_when_sixteen_bit_cells_are_not_exact():
    """Penalties beyond lj_params_ok (a gap extension of 60 makes 16-bit cells unsafe): the read path keeps such a handle's small jobs on
    the wave routines, and so do kinds 4-6 -- same results as the oracle with those penalties."""
    from lamsa_amd import hp
    over = dict(ins_gape=60, del_gape=60, ins_ext_e=60, del_ext_e=60)
    lp = reflib.lo_para("ont2d", **over)
    h = hp.LamsaHp(hp.make_para("ont2d", **over))
    try:
        jobs = _lane_jobs(61, ERR["ont2d"], 127, 255)[:300]
        for kind, w, h0 in ((0, lp.band_w, 1), (1, lp.band_w, 50), (2, 0, 100)):
            got = h.dp_batch(jobs, kind + 4, w, h0)
            assert goldenlib.same_dp(reflib.oracle_dp(jobs, lp, kind, w, h0), got, kind) == [], (kind, w, h0)
    finally:
        h.close()


def test_hip_lane_kinds_refuse_what_their_buffers_cannot_hold(handles):
    rng = np.random.default_rng(3)
    long_q = rng.integers(0, 4, 200, dtype=np.uint8)
    with pytest.raises(RuntimeError, match="beyond the lane routines"):
        handles["ont2d"].dp_batch([(long_q, long_q[:100])], 5, 100, 50)
    with pytest.raises(RuntimeError, match="mixes job classes"):
        handles["ont2d"].dp_batch([(long_q[:50], long_q[:60]), (long_q[:50], long_q[:60])], np.array([4, 1], np.int32), 100, 50)
