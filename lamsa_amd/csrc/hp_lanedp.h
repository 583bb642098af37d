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

#define HP_LJ_QCAP 160           // longest query of a lane job
#define HP_LJ_TCAP 256           // longest target (fits the 8 bits of the job record)
#define HP_LJ_QSMALL 64          // the short jobs -- most of them -- run in a kernel of their own with a third of the LDS per wave
#define HP_LJ_TSMALL 160
#define HP_LJ_CIG  (HP_LJ_QCAP + HP_LJ_TCAP + 8)
// LDS of a wave of the lane-per-job kernels: per lane a row of (qcap + 2) cells {H:16 | E:16} and the query's base codes,
// cell j of lane l at [j * 64 + l] (the lanes of a wave walk their rows together: conflict-free)
#define HP_LJ_LDS_WORDS(qcap) (((qcap) + 2) * 64 + ((qcap) + 2) * 16)
#define HP_LJ_NEG (-20000)       // MINUS_INF of ksw_global2 in 16 bits: every comparison comes out as with -0x40000000 (see lj_params_ok)

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
    const HP_G uint8_t *q; int qs, qcomp, qlen;   // query: base q[j * qs], complemented when qcomp (a '-' line reads the read backwards)
    const HP_G uint8_t *pac; int64_t tk; int ts, tlen;   // target: base i is the 2-bit base tk + i * ts of the packed reference
    HP_G uint8_t *z; int zl, zs;             // direction matrix: cell (row i, column c) at z[(i * zs + c) * 64 + zl] (zl = the lane, zs = a row stride
                                             // common to the 64 jobs of the group, so that lanes at the same cell store side by side)
    HP_L int32_t *row; HP_L uint8_t *qrow;   // this lane's cells in LDS: row[j * 64], qrow[j * 64]
    int rev;                                 // the query codes in qrow are stored for the reversed job (ksw_extend_r)
    long long cells;
};
HP_INL int lj_qbase(const LaneJob &J, int j) { const int c = J.q[(long)j * J.qs]; return J.qcomp ? (c < 4 ? 3 - c : 4) : c; }
HP_INL int lj_t(const LaneJob &J, int i) { const int64_t k = J.tk + (int64_t)i * J.ts; return J.pac[k >> 2] >> ((~k & 3) << 1) & 3; }      // _get_pac, bntseq.c:242
HP_INL void lj_stage_query(LaneJob &J) { for (int j = 0; j < J.qlen; ++j) J.qrow[j * 64] = (uint8_t)lj_qbase(J, j); J.rev = 0; }
HP_INL int lj_q(const LaneJob &J, int j) { return J.qrow[(J.rev ? J.qlen - 1 - j : j) * 64]; }
HP_INL LaneJob lj_rev(const LaneJob &J)
{   // ksw_extend_r (src/ksw.c:820): both sequences reversed (views)
    LaneJob R = J;
    R.rev = !J.rev;
    R.tk = J.tk + (int64_t)(J.tlen > 0 ? J.tlen - 1 : 0) * J.ts; R.ts = -J.ts;
    return R;
}
HP_INL int lj_sub(int sc_match, int sc_mis, int t, int q) { const int v = t == q ? sc_match : sc_mis; return q > 3 ? -1 : v; }      // lamsa_fill_mat, lamsa_aln.c:1331-1340 (target bases are 0..3)
HP_INL int lj_pack(int h, int e) { return (int)(((unsigned)e << 16) | ((unsigned)h & 0xffffu)); }
HP_INL int lj_h(int w) { return (int)(short)(w & 0xffff); }
HP_INL int lj_e(int w) { return w >> 16; }

// traceback (src/ksw.c:638-649, 792-801)
HP_INL void lj_backtrack(const LaneJob &J, int n_col, int w, int i, int k, LCig &out)
{
    int which = 0;
    out.n = 0;
    while (i >= 0 && k >= 0) {
        const int off = i > w ? i - w : 0;
        which = J.z[((size_t)i * J.zs + (k - off)) * 64 + J.zl] >> (which << 1) & 3;
        if (which == 0) { lc_push0(out, 1 << 4 | C_M); --i; --k; }
        else if (which == 1) { lc_push0(out, 1 << 4 | C_D); --i; }
        else { lc_push0(out, 1 << 4 | C_I); --k; }
    }
    if (i >= 0) lc_push0(out, (i + 1) << 4 | C_D);
    if (k >= 0) lc_push0(out, (k + 1) << 4 | C_I);
    lc_invert(out);
}

// Can the jobs of this handle run with 16-bit cells?  ksw_global2's cells are either real scores or MINUS_INF plus or minus a few
// score terms; F = (largest penalty) * (qlen + tlen + 8) bounds both the real scores' magnitude and that drift.  With HP_LJ_NEG =
// -20000 for MINUS_INF and F < 10000 the two kinds stay apart (-20000 + F < -F) and nothing leaves the 16 bits (-20000 - F > -32768),
// and the offsets from MINUS_INF are the reference's own, so every comparison of the recurrence (and with it every direction bit) is
// the one the reference makes with 32-bit cells.  ksw_extend_core's cells lie in [0, h0 + qlen * match].
HP_HD bool lj_params_ok(const lamsa_hp_para *P)
{
    int mx = P->match;
    const int v[] = {P->mis, P->ins_gapo, P->ins_gape, P->del_gapo, P->del_gape, P->ins_ext_o, P->ins_ext_e, P->del_ext_o, P->del_ext_e, 1};
    for (int i = 0; i < 10; ++i) mx = v[i] > mx ? v[i] : mx;
    return mx > 0 && mx * (HP_LJ_QCAP + HP_LJ_TCAP + 8) < 10000 && P->match > 0 && P->mis >= 0;
}

// ksw_global2 (src/ksw.c:543-653)
HP_INL int lj_global(const lamsa_hp_para *P, LaneJob &J, int o_del, int e_del, int o_ins, int e_ins, int w, LCig *out)
{
    const int qlen = J.qlen, tlen = J.tlen;
    { const int d = iabs(qlen - tlen) + 3; if (w < d) w = d; }                       // :549
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;                         // :559
    const int sc_match = P->match, sc_mis = 0 - P->mis;
    HP_L int32_t *row = J.row;
    int i, j;
    row[0] = lj_pack(0, HP_LJ_NEG);                                                // :569-572
    for (j = 1; j <= qlen && j <= w; ++j) row[j * 64] = lj_pack(-(o_ins + e_ins * j), HP_LJ_NEG);
    for (; j <= qlen; ++j) row[j * 64] = lj_pack(HP_LJ_NEG, HP_LJ_NEG);
    int t_next = tlen > 0 ? lj_t(J, 0) : 0;
    for (i = 0; i < tlen; ++i) {
        int f = HP_LJ_NEG, h1;
        const int ti = t_next;
        if (i + 1 < tlen) t_next = lj_t(J, i + 1);                                  // the next row's base is on its way while this row runs
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : HP_LJ_NEG;                     // :579
        J.cells += end > beg ? end - beg : 0;
        int wj = beg < end ? row[beg * 64] : 0, qj = beg < end ? lj_q(J, beg) : 0;
        for (j = beg; j < end; ++j) {
            const int wn = row[(j + 1) * 64], qn = lj_q(J, j + 1 < qlen ? j + 1 : j);             // next cell's operands
            int m = lj_h(wj), e = lj_e(wj), h, t, dir;
            m += lj_sub(sc_match, sc_mis, ti, qj);
            dir = m >= e ? 0 : 1; h = m >= e ? m : e;                               // ties: M over E
            dir = h >= f ? dir : 2; h = h >= f ? h : f;                             //       then over F
            t = m - oe_del; e -= e_del;
            if (e > t) dir |= 1 << 2; else e = t;
            row[j * 64] = lj_pack(h1, e);                                           // H[j] = h1 (the cell to the left), E[j] = e
            h1 = h;
            t = m - oe_ins; f -= e_ins;
            if (f > t) dir |= 2 << 4; else f = t;
            if (out) J.z[((size_t)i * J.zs + (j - beg)) * 64 + J.zl] = (uint8_t)dir;
            wj = wn; qj = qn;
        }
        row[end * 64] = lj_pack(h1, HP_LJ_NEG);                                     // :632
    }
    const int score = lj_h(row[qlen * 64]);
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
    const int sc_match = P->match, sc_mis = 0 - P->mis;
    int i, j, k, beg, end, max, max_i, max_j, max_ie, gscore;
    HP_L int32_t *row = J.row;
    {   // :692-694
        int hp = h0;
        row[0] = lj_pack(h0, 0);
        for (j = 1; j <= qlen + 1; ++j) {
            int hv = 0;
            if (j == 1) hv = h0 > oe_ins ? h0 - oe_ins : 0;
            else if (j <= qlen && hp > e_ins) hv = hp - e_ins;
            row[j * 64] = lj_pack(hv, 0);
            hp = hv;
        }
    }
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
    int t_next = tlen > 0 ? lj_t(J, 0) : 0;
    for (i = 0; i < tlen; ++i) {
        int t, f = 0, h1, m = 0, mj = -1;
        const int ti = t_next;
        if (i + 1 < tlen) t_next = lj_t(J, i + 1);
        const int d_beg = i > w ? i - w : 0;
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
        else h1 = 0;
        J.cells += end > beg ? end - beg : 0;
        if (out) {                                                                  // cells of the row outside the band read as "never written" (memset 255, :707)
            const int c_hi = d_beg + n_col;
            for (j = d_beg; j < beg && j < c_hi; ++j) J.z[((size_t)i * J.zs + (j - d_beg)) * 64 + J.zl] = 255;
            for (j = end > d_beg ? end : d_beg; j < c_hi; ++j) J.z[((size_t)i * J.zs + (j - d_beg)) * 64 + J.zl] = 255;
        }
        int wj = beg < end ? row[beg * 64] : 0, qj = beg < end ? lj_q(J, beg) : 0;
        for (j = beg; j < end; ++j) {
            const int wn = row[(j + 1) * 64], qn = lj_q(J, j + 1 < qlen ? j + 1 : j);
            int M = lj_h(wj), e = lj_e(wj), h, dir;
            M = M ? M + lj_sub(sc_match, sc_mis, ti, qj) : 0;                       // :737
            dir = M > e ? 0 : 1; h = M > e ? M : e;                                 // ties: E over M
            dir = h > f ? dir : 2; h = h > f ? h : f;                               //       F over both
            mj = m > h ? mj : j;                                                    // last j among equals
            m = m > h ? m : h;
            t = M - oe_del; t = t > 0 ? t : 0; e -= e_del;
            if (e > t) dir |= 1 << 2; else e = t;
            row[j * 64] = lj_pack(h1, e);
            h1 = h;
            t = M - oe_ins; t = t > 0 ? t : 0; f -= e_ins;
            if (f > t) dir |= 2 << 4; else f = t;
            if (out) J.z[((size_t)i * J.zs + (j - d_beg)) * 64 + J.zl] = (uint8_t)dir;
            wj = wn; qj = qn;
        }
        row[end * 64] = lj_pack(h1, 0);                                             // :758
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
        for (j = beg; j < end && row[j * 64] == 0; ++j) { }                         // :775-778 (H == 0 && E == 0)
        beg = j;
        for (j = end; j >= beg && row[j * 64] == 0; --j) { }
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
