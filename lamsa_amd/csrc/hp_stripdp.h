// hp_stripdp.h -- the junction extensions with 65 .. 127 query bases, FOUR JOBS PER WAVEFRONT (src/ksw.c:667-807 inside ksw_bi_extend :862-880).
//
// split_mapping's mismatch branch (src/frag_check.c:547-559) of a noisy read calls ksw_bi_extend on read gaps of 75, 100, 125 bases; its
// left extension (ksw_extend_c with h0 = 100) reaches the end of the query or the target for most of them, and then the job is done
// (:875-880).  The row-parallel routines of hp_ksw.h give such a job a whole wavefront: a row of ~110 cells is two 64-lane tiles, each
// with its own scan, reductions, ballots and scalar band bookkeeping -- ~450 wave instructions per row for ~110 cells.  The
// lane-per-job routines of hp_lanedp.h need a private row of 160 cells per lane (52 KB of LDS per wave: three waves per CU).
// Here a job gets a QUARTER of a wave -- one 16-lane DPP row -- and every lane a strip of HP_ST_S = 8 consecutive columns of the
// reference's eh[] array in registers ({H:16 | E:16} packed: ksw_extend_core's cells lie in [0, h0 + qlen * match], see lj_params_ok):
//   * inside its strip a lane walks the columns one after the other; across lanes the F recurrence is a max-plus prefix scan --
//     ksw.c opens a gap from M, never from H (:751-755), so F(j) = max_k<j (max(M(k) - oe_ins, 0) - (j-1-k) * e_ins) is a scan of
//     values that do not depend on F -- done with four row_shr steps that never leave the 16-lane row;
//   * row maximum (last j among equals, :743-744), H(i, end-1), first and last non-zero cell of the row (:775-778) are all-reduces
//     of the same kind; the per-job state (band limits, maximum, z-drop, g-score) is kept by every lane of the group;
//   * the direction matrix is one dword per lane and row -- eight 4-bit cells, 0xF = outside the band -- stored side by side (64 B per
//     job and row), the traceback reads it back per group.
// ~260 wave instructions per row for four jobs.  Rows of all four jobs advance together (jobs are queued by length class); a job that
// has ended idles until the others have.  Jobs whose left extension does not end the call (ksw.c:881 on) are left to the fill kernel.
#pragma once
#include "hp_lanedp.h"

namespace hp {

#define HP_ST_S 8                                  // columns per lane
#define HP_ST_QMAX (16 * HP_ST_S - 1)              // eh[] has qlen + 1 entries
#define HP_ST_TMAX 255
#define HP_ST_CIG (HP_ST_QMAX + HP_ST_TMAX + 8)    // CIGAR words of one job (one op per step at most)
#define HP_ST_ZROWS (HP_ST_TMAX + 1)
#define HP_ST_IDENT (-0x7f000000)
// scratch of a wave in its slab: four direction matrices (16 dwords per row) and four CIGAR buffers
#define HP_ST_SLAB_BYTES (4 * HP_ST_ZROWS * 16 * 4 + 4 * HP_ST_CIG * 4)

struct StripJob {                 // per lane, equal within a 16-lane group
    wv::Lane<int> on;             // the group has a job
    wv::Lane<long long> q;        // address of query base 0
    wv::Lane<int> qs, qcomp, qlen, tlen;
    wv::Lane<long long> tk;       // index of target base 0 in the packed reference (base i at tk + i)
};
struct StripRes { wv::Lane<int> score, qle, tle, n_cig, cells; };      // the CIGAR of group g: n_cig words at cbuf + g * HP_ST_CIG

HP_INL int st_lo(int w) { return (int)(short)(w & 0xffff); }
HP_INL int st_hi(int w) { return w >> 16; }
HP_INL int st_pack(int h, int e) { return (int)(((unsigned)e << 16) | ((unsigned)h & 0xffffu)); }

// ksw_extend_core(w = max(|qlen - tlen| + 3, band_w), h0) with traceback for up to four jobs.  zbuf, cbuf: this wave's scratch.
HP_INL void strip_extend(const lamsa_hp_para *P, const HP_G uint8_t *pac, const StripJob &J, int h0, uint32_t *zbuf_, cig_t *cbuf_, StripRes &R)
{
    HP_G uint32_t *zbuf = (HP_G uint32_t *)zbuf_;
    HP_G cig_t *cbuf = (HP_G cig_t *)cbuf_;
    const int o_ins = P->ins_ext_o, e_ins = P->ins_ext_e, o_del = P->del_ext_o, e_del = P->del_ext_e;
    const int end_bonus = P->end_bonus, zdrop = P->zdrop;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int sc_match = P->match, sc_mis = 0 - P->mis;
    // Registers per lane: the eight cells of its strip and their query codes (4 bits each in one word); everything else about a row is
    // recomputed where it is needed rather than kept -- the kernel's speed is its occupancy.
    wv::Lane<int> W[HP_ST_S], qpk;
    wv::Lane<int> beg, end, wl, ncol, mx, mx_i, mx_j, mx_ie, gs, alive, tnext, cells;
    int max_rows = 0;
    WAVE_FOR(l) {
        const int gl = l & 15, ql = J.qlen[l], tl = J.tlen[l];
        int w = iabs(ql - tl) + 3 > P->band_w ? iabs(ql - tl) + 3 : P->band_w;        // ksw_bi_extend, :873
        {   // :696-704 (double arithmetic, truncation toward zero as in the reference)
            int mxs = P->match > 0 ? P->match : 0;
            if (-P->mis > mxs) mxs = -P->mis;
            int max_ins = (int)((double)(ql * mxs + end_bonus - o_ins) / e_ins + 1.);
            max_ins = max_ins > 1 ? max_ins : 1;
            w = w < max_ins ? w : max_ins;
            int max_del = (int)((double)(ql * mxs + end_bonus - o_del) / e_del + 1.);
            max_del = max_del > 1 ? max_del : 1;
            w = w < max_del ? w : max_del;
        }
        wl[l] = w; ncol[l] = ql < 2 * w + 1 ? ql : 2 * w + 1;
        const int h1v = h0 > oe_ins ? h0 - oe_ins : 0;
        const HP_G uint8_t *gq = (const HP_G uint8_t *)J.q[l];
        int pk = 0;
#pragma unroll
        for (int s = 0; s < HP_ST_S; ++s) {                                           // first row, :692-694; the lane's query codes
            const int j = gl * HP_ST_S + s;
            const int hv = j == 0 ? h0 : (j == 1 ? h1v : ((j <= ql && h1v - (j - 2) * e_ins > e_ins) ? h1v - (j - 1) * e_ins : 0));
            W[s][l] = st_pack(hv, 0);
            int c = 4;
            if (J.on[l] && j < ql) { c = gq[(long)j * J.qs[l]]; if (J.qcomp[l]) c = c < 4 ? 3 - c : 4; }
            pk |= c << (4 * s);
        }
        qpk[l] = pk;
        beg[l] = 0; end[l] = ql; mx[l] = h0; mx_i[l] = -1; mx_j[l] = -1; mx_ie[l] = -1; gs[l] = -1; cells[l] = 0;
        alive[l] = J.on[l] && tl > 0;
        tnext[l] = alive[l] ? (int)(pac[J.tk[l] >> 2] >> ((~J.tk[l] & 3) << 1) & 3) : 0;
    }
    {   wv::Lane<int> tr;
        WAVE_FOR(l) tr[l] = J.on[l] ? J.tlen[l] : 0;
        max_rows = wv::reduce_max(tr); }
#define HP_ST_M(s_) (st_lo(W[s_][l]) ? st_lo(W[s_][l]) + lj_sub(sc_match, sc_mis, ti, (qpk[l] >> (4 * (s_))) & 15) : 0)       /* :737 */
    for (int i = 0; i < max_rows; ++i) {
        if (wv::ballot(alive) == 0) break;
        wv::Lane<int> tot, mkey, hlast, c7, negf, lastnz, tcur;
        // ---- A: this row's band; the lane's part of the F scan: max over its cells of (max(M - oe_ins, 0) + j * e_ins)
        WAVE_FOR(l) {
            int run = HP_ST_IDENT;
            tcur[l] = 0;
            if (alive[l]) {
                const int gl = l & 15, ql = J.qlen[l], w = wl[l];
                int b = beg[l], e = end[l];
                if (b < i - w) b = i - w;                                             // :718-720
                if (e > i + w + 1) e = i + w + 1;
                if (e > ql) e = ql;
                beg[l] = b; end[l] = e;
                if (gl == 0 && e > b) cells[l] += e - b;
                const int ti = tnext[l];
                tcur[l] = ti;
                if (i + 1 < J.tlen[l]) { const long long k = J.tk[l] + i + 1; tnext[l] = (int)(pac[k >> 2] >> ((~k & 3) << 1) & 3); }       // the next row's base is on its way
#pragma unroll
                for (int s = 0; s < HP_ST_S; ++s) {
                    const int j = gl * HP_ST_S + s;
                    int tt = HP_ST_M(s) - oe_ins; tt = tt > 0 ? tt : 0;
                    const int key = (j >= b && j < e) ? tt + j * e_ins : HP_ST_IDENT;
                    run = run > key ? run : key;
                }
            }
            tot[l] = run;
        }
        wv::row16_scan_max_excl(tot, HP_ST_IDENT);
        // ---- B: F, H, E and the direction of every cell of the row; eh[] after the row (:756-758) except the lane's first column,
        // whose new H comes from the lane below
        WAVE_FOR(l) {
            int mk = -1, hl = -1, h7 = 0;
            if (alive[l]) {
                const int gl = l & 15, b = beg[l], e = end[l], ti = tcur[l];
                int h1_init = 0;
                if (b == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }      // :722-725
                int zw = 0, run = tot[l], hprev = 0;
#pragma unroll
                for (int s = 0; s < HP_ST_S; ++s) {
                    const int j = gl * HP_ST_S + s;
                    int nib = 0xf, h = 0;
                    int hn = st_lo(W[s][l]), en = st_hi(W[s][l]);
                    if (j >= b && j < e) {
                        const int M = HP_ST_M(s);
                        int tt = M - oe_ins; tt = tt > 0 ? tt : 0;
                        int f = 0 - (j - b) * e_ins;                                  // F(i, beg) = 0 carried along the row
                        if (j > b) { const int g = run - (j - 1) * e_ins; f = g > f ? g : f; }
                        { const int key = tt + j * e_ins; run = run > key ? run : key; }
                        int dir = M > en ? 0 : 1; h = M > en ? M : en;                  // ties: E over M   :738-739
                        dir = h > f ? dir : 2;    h = h > f ? h : f;                    //       F over both :740-741
                        if (f - e_ins > tt) dir |= 8;                                   // :751-755
                        tt = M - oe_del; tt = tt > 0 ? tt : 0; en -= e_del;
                        if (en > tt) dir |= 4; else en = tt;                            // :745-750
                        nib = dir;
                        const int k = (h << 8) | j;                                     // row maximum, last j among equals (:743-744)
                        mk = mk > k ? mk : k;
                        if (j == e - 1) hl = h;
                    }
                    if (s > 0) {                                                        // eh[j].h = H(i, j - 1) (:756), eh[end] = {h1, 0} (:758)
                        if (b < e) { if (j == b) hn = h1_init; else if (j > b && j <= e) hn = hprev; }
                        else if (j == e) hn = h1_init;
                    }
                    if (j == e) en = 0;
                    W[s][l] = st_pack(hn, en);
                    hprev = h;
                    zw |= nib << (4 * s);
                }
                h7 = hprev;
                zbuf[((size_t)(l >> 4) * HP_ST_ZROWS + i) * 16 + gl] = (uint32_t)zw;
            }
            mkey[l] = mk; hlast[l] = hl; c7[l] = h7;
        }
        wv::row16_allmax(mkey); wv::row16_allmax(hlast); wv::row16_shr1(c7, 0);
        // ---- C: the lane's first column; the non-zero span of eh[] (:775-778)
        WAVE_FOR(l) {
            int fnz = 0x7fffffff, lnz = -1;
            if (alive[l]) {
                const int gl = l & 15, b = beg[l], e = end[l];
                {
                    const int j = gl * HP_ST_S;
                    int h1_init = 0;
                    if (b == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
                    int hn = st_lo(W[0][l]);
                    if (b < e) { if (j == b) hn = h1_init; else if (j > b && j <= e) hn = c7[l]; }
                    else if (j == e) hn = h1_init;
                    W[0][l] = st_pack(hn, st_hi(W[0][l]));
                }
#pragma unroll
                for (int s = 0; s < HP_ST_S; ++s) {
                    const int j = gl * HP_ST_S + s;
                    const bool nz = j >= b && j <= e && W[s][l] != 0;                   // H != 0 || E != 0
                    if (nz && j < e && j < fnz) fnz = j;
                    if (nz) lnz = j;
                }
            }
            negf[l] = fnz == 0x7fffffff ? -0x7fffffff : 0 - fnz; lastnz[l] = lnz;
        }
        wv::row16_allmax(negf); wv::row16_allmax(lastnz);
        // ---- D: g-score, maximum, z-drop (:759-773); the band of the next row (:775-778)
        WAVE_FOR(l) {
            if (alive[l]) {
                const int ql = J.qlen[l], b = beg[l], e = end[l];
                const int mrow = mkey[l] < 0 ? 0 : mkey[l] >> 8, mj = mkey[l] < 0 ? -1 : mkey[l] & 255;
                int h1_init = 0;
                if (b == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
                const int h_last = b < e ? hlast[l] : h1_init;
                if ((b < e ? e : b) == ql) { mx_ie[l] = gs[l] > h_last ? mx_ie[l] : i; gs[l] = gs[l] > h_last ? gs[l] : h_last; }
                bool stop = mrow == 0;                                                  // :763
                if (!stop) {
                    if (mrow > mx[l]) { mx[l] = mrow; mx_i[l] = i; mx_j[l] = mj; }
                    else if (zdrop > 0) {
                        if (i - mx_i[l] > mj - mx_j[l]) { if (mx[l] - mrow - ((i - mx_i[l]) - (mj - mx_j[l])) * e_del > zdrop) stop = true; }
                        else { if (mx[l] - mrow - ((mj - mx_j[l]) - (i - mx_i[l])) * e_ins > zdrop) stop = true; }
                    }
                }
                if (!stop) {
                    const int nb = negf[l] == -0x7fffffff ? e : 0 - negf[l];
                    const int jl = lastnz[l] < 0 ? nb - 1 : lastnz[l];
                    beg[l] = nb; end[l] = jl + 2 < ql ? jl + 2 : ql;
                }
                if (stop || i + 1 >= J.tlen[l]) alive[l] = 0;
            }
        }
    }
#undef HP_ST_M
    // ---- end point (:785-789) and traceback (:792-801), every group on its own matrix
    wv::Lane<int> bi, bk, bon, which, n, pend, have;
    WAVE_FOR(l) {
        int i, k;
        if (gs[l] <= 0 || gs[l] <= mx[l] - end_bonus) { i = mx_i[l]; k = mx_j[l]; }
        else { i = mx_ie[l]; k = J.qlen[l] - 1; }
        R.qle[l] = k + 1; R.tle[l] = i + 1; R.score[l] = mx[l]; R.cells[l] = cells[l];
        bi[l] = i; bk[l] = k; bon[l] = J.on[l] != 0; which[l] = 0; n[l] = 0; pend[l] = 0; have[l] = 0;
    }
    wv::sync();
#define HP_ST_PUSH(l_, w_) do { const int v_ = (w_); if (have[l_] && (pend[l_] & 0xf) == (v_ & 0xf)) pend[l_] += (v_ >> 4) << 4; \
        else { if (have[l_]) cbuf[(size_t)((l_) >> 4) * HP_ST_CIG + n[l_]++] = pend[l_]; pend[l_] = v_; have[l_] = 1; } } while (0)
    for (;;) {
        wv::Lane<int> go;
        WAVE_FOR(l) go[l] = bon[l] && bi[l] >= 0 && bk[l] >= 0;
        if (wv::ballot(go) == 0) break;
        WAVE_FOR(l) {
            if (go[l]) {
                const int ii = bi[l], kk = bk[l], w = wl[l];
                const int off = ii > w ? ii - w : 0;
                int cell = 255;                                                         // outside the row's window: never written (:707)
                if (kk >= off && kk - off < ncol[l]) {
                    const int nib = (int)(zbuf[((size_t)(l >> 4) * HP_ST_ZROWS + ii) * 16 + (kk >> 3)] >> ((kk & 7) << 2)) & 0xf;
                    cell = nib == 0xf ? 255 : ((nib & 3) | ((nib & 4) ? 1 << 2 : 0) | ((nib & 8) ? 2 << 4 : 0));
                }
                const int wh = cell >> (which[l] << 1) & 3;
                which[l] = wh;
                if (wh == 0) { HP_ST_PUSH(l, 1 << 4 | C_M); --bi[l]; --bk[l]; }
                else if (wh == 1) { HP_ST_PUSH(l, 1 << 4 | C_D); --bi[l]; }
                else { HP_ST_PUSH(l, 1 << 4 | C_I); --bk[l]; }
            }
        }
    }
    WAVE_FOR(l) {
        if (bon[l]) {
            if (bi[l] >= 0) HP_ST_PUSH(l, (bi[l] + 1) << 4 | C_D);
            if (bk[l] >= 0) HP_ST_PUSH(l, (bk[l] + 1) << 4 | C_I);
            if (have[l]) cbuf[(size_t)(l >> 4) * HP_ST_CIG + n[l]++] = pend[l];
        }
        R.n_cig[l] = bon[l] ? n[l] : 0;
    }
#undef HP_ST_PUSH
    wv::sync();
    {   // _invert_cigar, the 16 lanes of a group in parallel
        int most = 0;
        { wv::Lane<int> t; WAVE_FOR(l) t[l] = R.n_cig[l]; most = wv::reduce_max(t); }
        for (int b0 = 0; b0 < most / 2; b0 += 16) {
            WAVE_FOR(l) {
                const int a = b0 + (l & 15), nn = R.n_cig[l];
                if (a < nn / 2) { HP_G cig_t *c = cbuf + (size_t)(l >> 4) * HP_ST_CIG; const cig_t x = c[a], y = c[nn - 1 - a]; c[a] = y; c[nn - 1 - a] = x; }
            }
        }
    }
    wv::sync();
}

// Do this handle's parameters keep ksw_extend_core's cells -- [0, h0 + qlen * match] -- and the keys of the row maximum inside their fields?
HP_HD bool st_params_ok(const lamsa_hp_para *P, int h0)
{
    return lj_params_ok(P) && h0 > 0 && h0 + HP_ST_QMAX * P->match < 32000 && P->ins_ext_e > 0 && P->del_ext_e > 0;
}

}  // namespace hp
