// hp_lanedp.h -- the small junction DPs of a line, ONE JOB PER LANE ("ragged, tiny, dependent DP jobs").
//
// A line of a noisy read has dozens of junctions between fragments whose read gap is 25..75 bases (split_mapping's
// mismatch branch, src/frag_check.c:547-559 -> ksw_bi_extend, src/ksw.c:862-926) and, in the low-error modes, a
// ksw_global2 per pair of neighbouring seeds (frag_extend, :360-400).  Every one of them is independent of the others: only
// merge_cigar (:251), which joins their CIGARs, is sequential.  The row-parallel routines of hp_ksw.h give such a job a whole
// wavefront -- a 25-column row in 64 lanes, ~300 wave instructions per row.  Here the jobs of a line are listed first and then
// run 64 at a time, each lane doing its own job cell by cell exactly as the reference's scalar loops do (same recurrences,
// same tie rules, same band and z-drop logic, same traceback): ~30 instructions per cell and lane, i.e. per wave instruction 64
// cells instead of one row.  A lane keeps its H/E row in private memory, its direction matrix lane-interleaved in the wave's
// slab, reads its query bases from the read and its target bases straight from the 2-bit reference.  The CIGARs go to a
// job arena; the fill kernel (hp_fill.h) picks them up where it would have run the DP and goes on with merge_cigar.
// Jobs that do not fit the lane's buffers, the SV branches and everything else stay with the wave-per-job routines.
#pragma once
#include "hp_fill.h"

namespace hp {

#define HP_LJ_QCAP 96            // longest query of a lane job
#define HP_LJ_TCAP 192           // longest target
#define HP_LJ_CIG  (HP_LJ_QCAP + HP_LJ_TCAP + 8)

struct LCig { cig_t *c; int n; };
HP_INL void lc_push0(LCig &v, cig_t w) { if (v.n > 0 && (v.c[v.n - 1] & 0xf) == (w & 0xf)) v.c[v.n - 1] += (w >> 4) << 4; else v.c[v.n++] = w; }      // _push_cigar0
HP_INL void lc_push1(LCig &v, cig_t w) { if ((w >> 4) != 0) lc_push0(v, w); }                                                                        // _push_cigar1
HP_INL void lc_pushv(LCig &v, const cig_t *c, int n)
{   // _push_cigar, src/frag_check.h:158-184
    if (n == 0) return;
    int j = 0;
    if (v.n > 0) {
        const cig_t last = v.c[v.n - 1], c0 = c[0];
        if ((last & 0xf) == (c0 & 0xf)) { v.c[v.n - 1] = last + ((c0 >> 4) << 4); j = 1; }
        else if (((last & 0xf) == C_I && (c0 & 0xf) == C_S) || ((last & 0xf) == C_S && (c0 & 0xf) == C_I)) { v.c[v.n - 1] = (((last >> 4) + (c0 >> 4)) << 4) | C_S; j = 1; }
    }
    for (; j < n; ++j) v.c[v.n++] = c[j];
}
HP_INL void lc_invert(LCig &v) { for (int a = 0, b = v.n - 1; a < b; ++a, --b) { const cig_t t = v.c[a]; v.c[a] = v.c[b]; v.c[b] = t; } }

// one lane's view of its job
struct LaneJob {
    const uint8_t *q; int qs, qlen;          // query: bases q[j * qs]
    const uint8_t *pac; int64_t tk; int ts, tlen;   // target: base i is the 2-bit base tk + i * ts of the packed reference
    uint8_t *z; int zl;                      // direction matrix: cell idx at z[idx * 64 + zl] (zl = the lane)
    long long cells;
};
HP_INL int lj_q(const LaneJob &J, int j) { return J.q[(long)j * J.qs]; }
HP_INL int lj_t(const LaneJob &J, int i) { const int64_t k = J.tk + (int64_t)i * J.ts; return J.pac[k >> 2] >> ((~k & 3) << 1) & 3; }      // _get_pac, bntseq.c:242
HP_INL LaneJob lj_rev(const LaneJob &J)
{   // ksw_extend_r (src/ksw.c:820): both sequences reversed (views)
    LaneJob R = J;
    R.q = J.q + (long)(J.qlen > 0 ? J.qlen - 1 : 0) * J.qs; R.qs = -J.qs;
    R.tk = J.tk + (int64_t)(J.tlen > 0 ? J.tlen - 1 : 0) * J.ts; R.ts = -J.ts;
    return R;
}

// traceback (src/ksw.c:638-649, 792-801)
HP_INL void lj_backtrack(const LaneJob &J, int n_col, int w, int i, int k, LCig &out)
{
    int which = 0;
    out.n = 0;
    while (i >= 0 && k >= 0) {
        const int off = i > w ? i - w : 0;
        which = J.z[((size_t)i * n_col + (k - off)) * 64 + J.zl] >> (which << 1) & 3;
        if (which == 0) { lc_push0(out, 1 << 4 | C_M); --i; --k; }
        else if (which == 1) { lc_push0(out, 1 << 4 | C_D); --i; }
        else { lc_push0(out, 1 << 4 | C_I); --k; }
    }
    if (i >= 0) lc_push0(out, (i + 1) << 4 | C_D);
    if (k >= 0) lc_push0(out, (k + 1) << 4 | C_I);
    lc_invert(out);
}

// ksw_global2 (src/ksw.c:543-653)
HP_INL int lj_global(const lamsa_hp_para *P, LaneJob &J, int o_del, int e_del, int o_ins, int e_ins, int w, LCig *out)
{
    const int qlen = J.qlen, tlen = J.tlen;
    { const int d = iabs(qlen - tlen) + 3; if (w < d) w = d; }                       // :549
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;                         // :559
    int H[HP_LJ_QCAP + 2], E[HP_LJ_QCAP + 2];
    int i, j;
    H[0] = 0; E[0] = HP_NEG_INF;                                                   // :569-572
    for (j = 1; j <= qlen && j <= w; ++j) { H[j] = -(o_ins + e_ins * j); E[j] = HP_NEG_INF; }
    for (; j <= qlen; ++j) H[j] = E[j] = HP_NEG_INF;
    for (i = 0; i < tlen; ++i) {
        int f = HP_NEG_INF, h1;
        const int ti = lj_t(J, i);
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : HP_NEG_INF;                    // :579
        J.cells += end > beg ? end - beg : 0;
        for (j = beg; j < end; ++j) {
            int m = H[j], e = E[j], h, t, dir;
            H[j] = h1;
            m += sub_score(P, ti, lj_q(J, j));
            dir = m >= e ? 0 : 1; h = m >= e ? m : e;                               // ties: M over E
            dir = h >= f ? dir : 2; h = h >= f ? h : f;                             //       then over F
            h1 = h;
            t = m - oe_del; e -= e_del;
            if (e > t) dir |= 1 << 2; else e = t;
            E[j] = e;
            t = m - oe_ins; f -= e_ins;
            if (f > t) dir |= 2 << 4; else f = t;
            if (out) J.z[((size_t)i * n_col + (j - beg)) * 64 + J.zl] = (uint8_t)dir;
        }
        H[end] = h1; E[end] = HP_NEG_INF;                                           // :632
    }
    const int score = H[qlen];
    if (out) {
        i = tlen - 1;
        const int k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;                    // :638
        lj_backtrack(J, n_col, w, i, k, *out);
    }
    return score;
}

// ksw_extend_core (src/ksw.c:667-807); h0 > 0, lengths >= 0 (checked by the caller)
HP_INL int lj_extend(const lamsa_hp_para *P, LaneJob &J, int w, int h0, int *qle, int *tle, LCig *out)
{
    const int qlen = J.qlen, tlen = J.tlen;
    const int o_ins = P->ins_ext_o, e_ins = P->ins_ext_e, o_del = P->del_ext_o, e_del = P->del_ext_e;
    const int end_bonus = P->end_bonus, zdrop = P->zdrop;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int i, j, k, beg, end, max, max_i, max_j, max_ie, gscore;
    int H[HP_LJ_QCAP + 2], E[HP_LJ_QCAP + 2];
    for (j = 0; j <= qlen + 1; ++j) { H[j] = 0; E[j] = 0; }
    H[0] = h0; H[1] = h0 > oe_ins ? h0 - oe_ins : 0;                                // :692-694
    for (j = 2; j <= qlen && H[j - 1] > e_ins; ++j) H[j] = H[j - 1] - e_ins;
    {   // :696-704 (double arithmetic, truncation toward zero as in the reference)
        int mx = P->match > 0 ? P->match : 0;
        if (-P->mis > mx) mx = -P->mis;
        int max_ins = (int)((double)(qlen * mx + end_bonus - o_ins) / e_ins + 1.);
        max_ins = max_ins > 1 ? max_ins : 1;
        w = w < max_ins ? w : max_ins;
        int max_del = (int)((double)(qlen * mx + end_bonus - o_del) / e_del + 1.);
        max_del = max_del > 1 ? max_del : 1;
        w = w < max_del ? w : max_del;
    }
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1;
    beg = 0; end = qlen;
    for (i = 0; i < tlen; ++i) {
        int t, f = 0, h1, m = 0, mj = -1;
        const int ti = lj_t(J, i);
        const int d_beg = i > w ? i - w : 0;
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
        else h1 = 0;
        J.cells += end > beg ? end - beg : 0;
        if (out) {                                                                  // cells of the row outside the band read as "never written" (memset 255, :707)
            const int c_hi = d_beg + n_col;
            for (j = d_beg; j < beg && j < c_hi; ++j) J.z[((size_t)i * n_col + (j - d_beg)) * 64 + J.zl] = 255;
            for (j = end > d_beg ? end : d_beg; j < c_hi; ++j) J.z[((size_t)i * n_col + (j - d_beg)) * 64 + J.zl] = 255;
        }
        for (j = beg; j < end; ++j) {
            int M = H[j], e = E[j], h, dir;
            H[j] = h1;
            M = M ? M + sub_score(P, ti, lj_q(J, j)) : 0;                           // :737
            dir = M > e ? 0 : 1; h = M > e ? M : e;                                 // ties: E over M
            dir = h > f ? dir : 2; h = h > f ? h : f;                               //       F over both
            h1 = h;
            mj = m > h ? mj : j;                                                    // last j among equals
            m = m > h ? m : h;
            t = M - oe_del; t = t > 0 ? t : 0; e -= e_del;
            if (e > t) dir |= 1 << 2; else e = t;
            E[j] = e;
            t = M - oe_ins; t = t > 0 ? t : 0; f -= e_ins;
            if (f > t) dir |= 2 << 4; else f = t;
            if (out) J.z[((size_t)i * n_col + (j - d_beg)) * 64 + J.zl] = (uint8_t)dir;
        }
        H[end] = h1; E[end] = 0;                                                    // :758
        if (j == qlen) {                                                            // :759-762
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        if (m == 0) break;
        if (m > max) { max = m; max_i = i; max_j = mj; }
        else if (zdrop > 0) {                                                       // :767-773
            if (i - max_i > mj - max_j) { if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
            else { if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
        }
        for (j = beg; j < end && H[j] == 0 && E[j] == 0; ++j) { }                   // :775-778
        beg = j;
        for (j = end; j >= beg && H[j] == 0 && E[j] == 0; --j) { }
        end = j + 2 < qlen ? j + 2 : qlen;
    }
    if (gscore <= 0 || gscore <= max - end_bonus) { i = max_i; k = max_j; }         // :785-789
    else { i = max_ie; k = qlen - 1; }
    *qle = k + 1; *tle = i + 1;
    if (out) lj_backtrack(J, n_col, w, i, k, *out);
    return max;
}

// ksw_bi_extend (src/ksw.c:862-926) with sw_mid_fix (:841-860); qlen > 0.  L, R: scratch CIGARs of the lane.
HP_INL int lj_bi_extend(const lamsa_hp_para *P, LaneJob &J, int lh0, int rh0, LCig &L, LCig &R, LCig &out)
{
    const int qlen = J.qlen, tlen = J.tlen;
    int res, lqe, lte, rqe, rte;
    out.n = 0; L.n = 0; R.n = 0;
    const int w = iabs(qlen - tlen) + 3 > P->band_w ? iabs(qlen - tlen) + 3 : P->band_w;     // :873
    lj_extend(P, J, w, lh0, &lqe, &lte, &L);
    res = lqe == qlen ? 0 : (lte == tlen ? 1 : 2);                                          // ksw_extend_c, :815-817
    if (res < 2) {                                                                          // :875-880
        lc_pushv(out, L.c, L.n);
        lc_push1(out, res == 0 ? ((tlen - lte) << 4) | C_D : ((qlen - lqe) << 4) | C_I);
        return 0;
    }
    if (bi_near_diag(P, qlen, tlen) && ((lqe << 1 > qlen) || (lte << 1 > tlen))) {          // :881-887
        lj_global(P, J, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, P->band_w, &out);
        return 0;
    }
    LaneJob Rj = lj_rev(J);
    lj_extend(P, Rj, w, rh0, &rqe, &rte, &R);
    J.cells = Rj.cells;
    res = rqe == qlen ? 0 : (rte == tlen ? 1 : 2);
    if (res < 2) {                                                                          // :892-899
        lc_push1(R, res == 0 ? ((tlen - rte) << 4) | C_D : ((qlen - rqe) << 4) | C_I);
        lc_invert(R);
        lc_pushv(out, R.c, R.n);
        return 0;
    }
    if (bi_near_diag(P, qlen, tlen) && ((rqe << 1 > qlen) || (rte << 1 > tlen))) {          // :900-906
        lj_global(P, J, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, P->band_w, &out);
        return 0;
    }
    lc_invert(R);
    {   // sw_mid_fix, :841-860
        const int Sn = qlen - lqe - rqe, Hn = tlen - lte - rte, half = P->split_len / 2;
        if (iabs(Sn) >= half || iabs(Hn) >= half || iabs(Sn - Hn) >= half) {
            lc_pushv(out, L.c, L.n);
            lc_push0(out, (cig_t)((uint32_t)Sn << 4) | C_S);
            lc_push0(out, (cig_t)((uint32_t)Hn << 4) | C_H);
            lc_pushv(out, R.c, R.n);
        } else {
            LCig g; g.c = L.c; g.n = 0;                                                     // L is not needed any more
            lj_global(P, J, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, P->band_w, &g);
            lc_pushv(out, g.c, g.n);
        }
    }
    return (qlen - lqe - rqe) >= P->split_len ? 1 : 0;                                      // :924
}

}  // namespace hp
