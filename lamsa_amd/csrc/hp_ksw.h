// hp_ksw.h -- banded affine-gap DP on one wavefront.
//
//   ksw_global   <- ksw_global2      reference src/ksw.c:543-653
//   ksw_extend   <- ksw_extend_core  reference src/ksw.c:667-807
//   ksw_extend_c/_r, sw_mid_fix, ksw_bi_extend <- src/ksw.c:809-926
//
// Parallelisation (this is not how the reference computes, only what it computes):
// both recurrences take E(i+1,j) and F(i,j+1) from M(i,j) = H(i-1,j-1)+S, never from
// H(i,j), so a row depends on the previous row only.  The 64 lanes own 64 consecutive
// band columns of the current row; F along the row is a max-plus prefix scan with linear
// decay (wv::scan_max_excl on M-oe_ins+j*e_ins); row maximum, the "last j among equals"
// rule, and the band shrink of the extension routine are wave reductions / ballots.
// Rows are processed strictly in order, which keeps the reference's per-row band
// trimming, m==0 break, z-drop test and tie rules exact.  The single in-place H/E row is
// kept (same stale-cell behaviour outside the band as the reference's eh[] array).
// Scores are int32 like the reference.  No MFMA: this is max-plus, not a contraction.
#pragma once
#include "hp_core.h"

namespace hp {

// LDS direction matrix: a cell needs 4 bits (move into H: 2, "E extends": 1, "F extends": 1; the reference's byte is
// h | e << 2 | f << 5), 0xF = never written.  A row is filled with 0xFF and every computed cell ANDs its nibble in.
#define HP_ZSTRIDE(n_col) ((((n_col) + 7) >> 3) << 2)          // bytes per row
// what the wave's LDS leaves for the matrix behind the rows and the query window: HP_LDS_Z_BYTES in the fill kernel (8 waves per SIMD), ten times
// that in the job launch of hp_wavejob.h (4 waves per SIMD); the two-columns-per-lane routines keep no rows in LDS and use all of it (z_cap_pk)
HP_INL size_t z_cap(const Ctx &cx) { const int w = cx.lds_words - (2 * HP_LDS_CELLS + HP_LDS_CELLS / 4); return w > 0 ? (size_t)w * 4 : 0; }
HP_INL size_t z_cap_pk(const Ctx &cx) { return cx.lds_words > 0 ? (size_t)cx.lds_words * 4 : 0; }
#define HP_ZFITS(n_col, rows) ((size_t)HP_ZSTRIDE(n_col) * (size_t)(rows) <= z_cap(cx))
HP_INL int z_nibble(int dir) { return (dir & 3) | ((dir >> 2) & 1) << 2 | ((dir >> 5) & 1) << 3; }
HP_INL void z_row_clear(HP_L uint8_t *LZ, int row, int n_col) {
    HP_L int *p = (HP_L int *)(LZ + row * HP_ZSTRIDE(n_col));
    const int nd = HP_ZSTRIDE(n_col) >> 2;
    for (int d0 = 0; d0 < nd; d0 += 64) { WAVE_FOR(l) { if (d0 + l < nd) p[d0 + l] = -1; } }
}
HP_INL void z_put(HP_L uint8_t *LZ, int row, int n_col, int c, int dir) {          // per lane
    HP_L int *p = (HP_L int *)(LZ + row * HP_ZSTRIDE(n_col)) + (c >> 3);
    wv::lds_and(p, (int)~((unsigned)(~z_nibble(dir) & 0xf) << ((c & 7) << 2)));
}

// ---- traceback (src/ksw.c:638-649 and :792-801).  The direction matrix lives in LDS (lz) when it fits, else in the
// wave's HBM slab (gz).  Cells the forward pass never wrote read as 255 (src/ksw.c:707): outside the row's window
// always; inside it the LDS matrix holds 255 where the band did not reach, and for the HBM matrix of the extension
// routine the per-row band limits (rowb) say which cells were written.
HP_FN void dp_backtrack(Ctx &cx, const HP_L uint8_t *lz, const uint8_t *z, const int32_t *rowb, int n_col, int w, int i, int k, CigV &out, int pk_stride = 0)
{
    pk_stride = wv::uni(pk_stride);                                        // > 0: the matrix of the two-columns-per-lane routines (in HBM, or in LDS with 0xF = never written), a nibble per cell, rows of pk_stride bytes indexed by the column itself
    // Which matrix: the one in LDS exactly when no slab matrix was handed over.  (Not "lz != nullptr": a null pointer into LDS is offset 0, and the
    // matrix of the two-columns-per-lane routines BEGINS at offset 0 of the wave's LDS -- DESIGN.md, hazards.)
    const bool in_lds = z == nullptr;
    const HP_G uint8_t *gz = (const HP_G uint8_t *)wv::uni64((long long)z);
    const HP_G hp_v2i *grb = (const HP_G hp_v2i *)wv::uni64((long long)rowb);     // HBM matrix of the extension routine: [beg, end) of every row
    HP_G cig_t *oc = (HP_G cig_t *)wv::uni64((long long)out.c);
    const int cap = wv::uni(out.cap);
    n_col = wv::uni(n_col); w = wv::uni(w); i = wv::uni(i); k = wv::uni(k);
    // the run being built stays in registers; finished runs are stored and never read back (_push_cigar0 semantics)
    // The first 64 finished runs stay in a lane register (run k in lane k): a CIGAR of at most 64 elements -- nearly every one -- is stored
    // once, already inverted, and never read back.
    int n = 0, pend = 0, which = 0;
    bool have = false;
    wv::Lane<int> runs;
    WAVE_FOR(l) { runs[l] = 0; }
#define HP_BT_OUT(v_) do { if (n < cap) { if (n < 64) wv::setlane(runs, n, (v_)); else { if (n == 64) { WAVE_FOR(l) { oc[l] = runs[l]; } } oc[n] = (v_); } ++n; } else cx.status |= ST_OVERFLOW; } while (0)
#define HP_BT_PUSH(w_) do { const int v_ = (w_); if (have && (pend & 0xf) == (v_ & 0xf)) pend += (v_ >> 4) << 4; \
        else { if (have) HP_BT_OUT(pend); pend = v_; have = true; } } while (0)
    const int zstride = HP_ZSTRIDE(n_col);
    // one cell of the matrix as the reference's byte; 255 = never written (outside the window or the band)
#define HP_BT_CELL(ii, kk, cell_) do { const int off_ = (ii) > w ? (ii) - w : 0; cell_ = 255; \
        if ((kk) >= off_ && (kk) - off_ < n_col) { \
            if (in_lds && pk_stride) { const int nib_ = (lz[(ii) * pk_stride + ((kk) >> 1)] >> (((kk) & 1) << 2)) & 0xf; \
                      cell_ = nib_ == 0xf ? 255 : ((nib_ & 3) | ((nib_ & 4) ? 1 << 2 : 0) | ((nib_ & 8) ? 2 << 4 : 0)); } \
            else if (in_lds) { const int c_ = (kk) - off_, nib_ = (lz[(ii) * zstride + (c_ >> 1)] >> ((c_ & 1) << 2)) & 0xf; \
                      cell_ = nib_ == 0xf ? 255 : ((nib_ & 3) | ((nib_ & 4) ? 1 << 2 : 0) | ((nib_ & 8) ? 2 << 4 : 0)); } \
            else if (pk_stride) { const int nib_ = (gz[(long)(ii) * pk_stride + (((kk) & (2 * pk_stride - 1)) >> 1)] >> (((kk) & 1) << 2)) & 0xf; \
                   cell_ = (nib_ & 3) | ((nib_ & 4) ? 1 << 2 : 0) | ((nib_ & 8) ? 2 << 4 : 0); \
                   if (grb) { const hp_v2i be_ = grb[(ii)]; if (!((kk) >= be_.x && (kk) < be_.y)) cell_ = 255; } } \
            else { const int zc_ = gz[(long)(ii) * n_col + ((kk) - off_)]; \
                   if (grb) { const hp_v2i be_ = grb[(ii)]; cell_ = ((kk) >= be_.x && (kk) < be_.y) ? zc_ : 255; } else cell_ = zc_; } } } while (0)
    while (i >= 0 && k >= 0) {
        int cell;
        if (which == 0) {
            // In state H the walk follows the diagonal for as long as the cells say "came from M": the 64 lanes look at the
            // next 64 diagonal cells at once and the whole run is pushed in one step.
            wv::Lane<int> cl, okl;
            WAVE_FOR(l) {
                const int ii = i - l, kk = k - l;
                int c = 255;
                if (ii >= 0 && kk >= 0) HP_BT_CELL(ii, kk, c);
                cl[l] = c; okl[l] = ii >= 0 && kk >= 0 && (c & 3) == 0;
            }
            const unsigned long long mk = wv::ballot(okl);
            const int run = mk == ~0ull ? 64 : __builtin_ctzll(~mk);
            if (run > 0) { HP_BT_PUSH(run << 4 | C_M); i -= run; k -= run; continue; }
            cell = wv::bcast(cl, 0);
        } else {
            HP_BT_CELL(i, k, cell);
        }
        which = cell >> (which << 1) & 3;
        if (which == 0) { HP_BT_PUSH(1 << 4 | C_M); --i; --k; }
        else if (which == 1) { HP_BT_PUSH(1 << 4 | C_D); --i; }
        else { HP_BT_PUSH(1 << 4 | C_I); --k; }
    }
    if (i >= 0) HP_BT_PUSH((i + 1) << 4 | C_D);
    if (k >= 0) HP_BT_PUSH((k + 1) << 4 | C_I);
    if (have) HP_BT_OUT(pend);
#undef HP_BT_PUSH
#undef HP_BT_OUT
#undef HP_BT_CELL
    if (n <= 64) { WAVE_FOR(l) { if (l < n) oc[n - 1 - l] = runs[l]; } }   // _invert_cigar on the way out
    else {
        wv::sync();
        for (int b0 = 0; b0 < n / 2; b0 += 64) {                           // _invert_cigar, lane-parallel
            WAVE_FOR(l) { const int a = b0 + l; if (a < n / 2) { const cig_t x = oc[a], y = oc[n - 1 - a]; oc[a] = y; oc[n - 1 - a] = x; } }
        }
    }
    wv::sync();
    out.n = n;
}

#define HP_SCAN_IDENT (-0x7f000000)

// what an extension returns: by value, in registers (results handed back through pointers into the caller's frame
// would go through scratch memory)
struct ExtRes { int score, qle, tle; };

// ---- ksw_global2 (src/ksw.c:543-653) with the H/E row in the wave's HBM slab: only for bands wider than the LDS row.
HP_NOINL int ksw_global_wide(Ctx &cx, int qlen, Seq q, int tlen, Seq t,
                        int o_del, int e_del, int o_ins, int e_ins, int w, CigV *out)
{
    long long cells_ = 0;                                                  // in a register: a counter in cx would be a memory round trip per row
    HP_T0(tg0_);
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;                 // :559
    const size_t mark = arena_mark(cx.tmp);
    int32_t *H = (int32_t *)arena_alloc(cx, sizeof(int32_t) * ((size_t)qlen + 1));
    int32_t *E = (int32_t *)arena_alloc(cx, sizeof(int32_t) * ((size_t)qlen + 1));
    uint8_t *z = out ? (uint8_t *)arena_alloc(cx, (size_t)n_col * tlen + 1) : nullptr;
    if (!H || !E || (out && !z)) { arena_release(cx.tmp, mark); return 0; }
    const int sc_match = cx.P->match, sc_mis = -cx.P->mis;
    HP_G int32_t *gH = (HP_G int32_t *)H, *gE = (HP_G int32_t *)E;
    HP_G uint8_t *gz = (HP_G uint8_t *)z;
    const HP_G uint8_t *gq = (const HP_G uint8_t *)q.p; const int qs = q.stride;
    const HP_G uint8_t *gt = (const HP_G uint8_t *)t.p; const int ts = t.stride;
#define HP_SUB(tb, qb) (((tb) > 3 || (qb) > 3) ? -1 : ((tb) == (qb) ? sc_match : sc_mis))

    for (int j0 = 0; j0 <= qlen; j0 += 64) {                               // first row, :569-572
        WAVE_FOR(l) {
            int j = j0 + l;
            if (j <= qlen) { gH[j] = j == 0 ? 0 : (j <= w ? -(o_ins + e_ins * j) : HP_NEG_INF); gE[j] = HP_NEG_INF; }
        }
    }
    wv::sync();
    for (int i = 0; i < tlen; ++i) {
        const int ti = gt[(long)i * ts];
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        cells_ += end > beg ? end - beg : 0;
        const int h1_init = beg == 0 ? -(o_del + e_del * (i + 1)) : HP_NEG_INF;   // :579
        int carryH = gH[beg];          // H(i-1,beg-1), read before the in-place update below
        int Fin = HP_NEG_INF;         // F(i,beg)
        gH[beg] = h1_init;            // eh[beg].h = H(i,beg-1)
        for (int j0 = beg; j0 < end; j0 += 64) {
            const int nxt = j0 + 64;
            const int carry_next = nxt <= qlen ? gH[nxt] : 0;   // old value, lane 63 is about to overwrite it
            wv::Lane<int> m, e, key;
            WAVE_FOR(l) {
                int j = j0 + l;
                if (j < end) {
                    int hm = l == 0 ? carryH : gH[j];
                    const int qb = gq[(long)j * qs];
                    m[l] = hm + HP_SUB(ti, qb);
                    e[l] = gE[j];
                    key[l] = m[l] - oe_ins + j * e_ins;
                } else { m[l] = 0; e[l] = 0; key[l] = HP_SCAN_IDENT; }
            }
            wv::scan_max_excl(key, HP_SCAN_IDENT);
            wv::Lane<int> fnext;
            WAVE_FOR(l) {
                int j = j0 + l;
                fnext[l] = 0;
                if (j < end) {
                    int f = Fin - l * e_ins;                               // F(i,j) carried in from the left
                    if (l > 0) { int g = key[l] - (j - 1) * e_ins; f = g > f ? g : f; }
                    int mm = m[l], ee = e[l], h, tt;
                    int dir = mm >= ee ? 0 : 1; h = mm >= ee ? mm : ee;    // ties: M over E   :598-599
                    dir = h >= f ? dir : 2;     h = h >= f ? h : f;        //       then over F :600-601
                    tt = mm - oe_del; ee -= e_del;
                    if (ee > tt) dir |= 1 << 2; else ee = tt;              // :603-607
                    tt = mm - oe_ins; f -= e_ins;
                    if (f > tt) dir |= 2 << 4; else f = tt;                // :608-611
                    gE[j] = ee;
                    gH[j + 1] = h;                                         // eh[j+1].h = H(i,j)
                    if (z) gz[(long)i * n_col + (j - beg)] = (uint8_t)dir;
                    fnext[l] = f;
                }
            }
            Fin = wv::bcast(fnext, 63);
            carryH = carry_next;
        }
        gE[end] = HP_NEG_INF;                                              // :632
        wv::sync();
    }
    const int score = H[qlen];
    if (out) {
        int i = tlen - 1;
        int k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;                 // :638
        HP_T0(tb0_);
        dp_backtrack(cx, nullptr, z, nullptr, n_col, w, i, k, *out);
        HP_TADD(cx, 28, tb0_);
    }
    cx.n_cells += cells_;                                                  // accounting: DP cell updates (bench.py: GCUPS), once per call
    arena_release(cx.tmp, mark);
    HP_TADD(cx, 24, tg0_);
    return score;
}

// ---- ksw_extend_core (src/ksw.c:667-807) with the H/E row in the wave's HBM slab: only for bands wider than the LDS row.
// w is already adjusted (see ksw_extend).
HP_NOINL ExtRes ksw_extend_wide(Ctx &cx, int qlen, Seq q, int tlen, Seq t, int w, int h0, CigV *out)
{
    long long cells_ = 0;                                                  // in a register: a counter in cx would be a memory round trip per row
    ExtRes er; er.score = 0; er.qle = 0; er.tle = 0;
    HP_T0(te0_);
    const lamsa_hp_para *P = cx.P;
    const int o_ins = P->ins_ext_o, e_ins = P->ins_ext_e, o_del = P->del_ext_o, e_del = P->del_ext_e;
    const int end_bonus = P->end_bonus, zdrop = P->zdrop;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    const size_t mark = arena_mark(cx.tmp);
    int32_t *H = (int32_t *)arena_alloc(cx, sizeof(int32_t) * ((size_t)qlen + 2));
    int32_t *E = (int32_t *)arena_alloc(cx, sizeof(int32_t) * ((size_t)qlen + 2));
    int32_t *rowb = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 2 * ((size_t)tlen + 1));
    uint8_t *z = (uint8_t *)arena_alloc(cx, (size_t)n_col * tlen + 1);
    if (!H || !E || !rowb || !z) { arena_release(cx.tmp, mark); return er; }

    const int sc_match = P->match, sc_mis = -P->mis;
    HP_G int32_t *gH = (HP_G int32_t *)H, *gE = (HP_G int32_t *)E, *growb = (HP_G int32_t *)rowb;
    HP_G uint8_t *gz = (HP_G uint8_t *)z;
    const HP_G uint8_t *gq = (const HP_G uint8_t *)q.p; const int qs = q.stride;
    const HP_G uint8_t *gt = (const HP_G uint8_t *)t.p; const int ts = t.stride;
    // first row, :692-694: h0, h0-oe_ins, then -e_ins per column while the previous cell is > e_ins
    {
        const int h1v = h0 > oe_ins ? h0 - oe_ins : 0;
        // H[j] = h1v - (j-1)*e_ins for 2 <= j <= jmax where jmax is the last j with H[j-1] > e_ins
        for (int j0 = 0; j0 <= qlen + 1; j0 += 64) {
            WAVE_FOR(l) {
                int j = j0 + l;
                if (j <= qlen + 1) {
                    int v = 0;
                    if (j == 0) v = h0;
                    else if (j == 1) v = h1v;
                    else if (j <= qlen) { int prev = h1v - (j - 2) * e_ins; if (prev > e_ins) v = prev - e_ins; }
                    gH[j] = v; gE[j] = 0;
                }
            }
        }
    }
    wv::sync();
    int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1;
    int beg = 0, end = qlen;
    for (int i = 0; i < tlen; ++i) {
        const int ti = gt[(long)i * ts];
        const int d_beg = i > w ? i - w : 0;
        if (beg < i - w) beg = i - w;                                      // :718-720
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        cells_ += end > beg ? end - beg : 0;
        int h1_init;
        if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
        else h1_init = 0;
        growb[2 * i] = beg; growb[2 * i + 1] = end;
        int carryH = gH[beg];
        int Fin = 0;
        long long best = -1;            // (h << 32 | j): row maximum, last j among equals (:743-744)
        int h_last = h1_init;           // H(i,end-1), or the first-column value when the row is empty
        // band shrink bookkeeping (:775-778): nz(j) = eh[j].h != 0 || eh[j].e != 0 after this row
        int first_nz = -1, last_nz = -1, prev_h_nz = h1_init != 0;
        if (beg < end) gH[beg] = h1_init; else gH[end] = h1_init;            // eh[end].h = h1 when the row is empty (:758)
        for (int j0 = beg; j0 < end; j0 += 64) {
            const int nxt = j0 + 64;
            const int carry_next = nxt <= qlen + 1 ? gH[nxt] : 0;
            wv::Lane<int> m, e, key;
            WAVE_FOR(l) {
                int j = j0 + l;
                if (j < end) {
                    int hm = l == 0 ? carryH : gH[j];
                    const int qb = gq[(long)j * qs];
                    int M = hm ? hm + HP_SUB(ti, qb) : 0;                   // :737
                    int tt = M - oe_ins; tt = tt > 0 ? tt : 0;
                    m[l] = M; e[l] = gE[j]; key[l] = tt + j * e_ins;
                } else { m[l] = 0; e[l] = 0; key[l] = HP_SCAN_IDENT; }
            }
            wv::scan_max_excl(key, HP_SCAN_IDENT);
            wv::Lane<int> fnext, hnz, enz;
            wv::Lane<long long> rk;
            WAVE_FOR(l) {
                int j = j0 + l;
                fnext[l] = 0; hnz[l] = 0; enz[l] = 0; rk[l] = -1;
                if (j < end) {
                    int f = Fin - l * e_ins;
                    if (l > 0) { int g = key[l] - (j - 1) * e_ins; f = g > f ? g : f; }
                    int M = m[l], ee = e[l], h, tt;
                    int dir = M > ee ? 0 : 1; h = M > ee ? M : ee;          // ties: E over M   :738-739
                    dir = h > f ? dir : 2;    h = h > f ? h : f;            //       F over both :740-741
                    tt = M - oe_del; tt = tt > 0 ? tt : 0; ee -= e_del;
                    if (ee > tt) dir |= 1 << 2; else ee = tt;               // :745-750
                    tt = M - oe_ins; tt = tt > 0 ? tt : 0; f -= e_ins;
                    if (f > tt) dir |= 2 << 4; else f = tt;                 // :751-755
                    gE[j] = ee;
                    gH[j + 1] = h;
                    gz[(long)i * n_col + (j - d_beg)] = (uint8_t)dir;
                    fnext[l] = f; hnz[l] = h != 0; enz[l] = ee != 0;
                    rk[l] = ((long long)h << 32) | (unsigned)j;
                }
            }
            Fin = wv::bcast(fnext, 63);
            carryH = carry_next;
            {
                long long b = wv::reduce_max64(rk);
                if (b > best) best = b;
                const int cnt = end - j0 < 64 ? end - j0 : 64;            // active lanes
                unsigned long long bh = wv::ballot(hnz), be = wv::ballot(enz);
                // nz for index j0+l: E bit l | H bit (l-1); index j0+cnt (== end on the last chunk) gets H bit cnt-1
                unsigned long long nzm = be | (bh << 1) | (unsigned long long)(prev_h_nz ? 1 : 0);
                unsigned long long inrow = cnt == 64 ? ~0ull : ((1ull << cnt) - 1);
                if (first_nz < 0 && (nzm & inrow)) first_nz = j0 + __builtin_ctzll(nzm & inrow);
                if (nzm & inrow) last_nz = j0 + 63 - __builtin_clzll(nzm & inrow);
                prev_h_nz = (int)((bh >> (cnt - 1)) & 1);
            }
        }
        // H(i,end-1): value of the last computed cell (needed for gscore / eh[end].h)
        wv::sync();
        if (beg < end) h_last = gH[end];
        gE[end] = 0;                                                       // :758
        const int jj = beg < end ? end : beg;                              // loop variable j after the row
        if (jj == qlen) {                                                  // :759-762
            max_ie = gscore > h_last ? max_ie : i;
            gscore = gscore > h_last ? gscore : h_last;
        }
        int mrow = 0, mj = -1;
        if (best >= 0) { mrow = (int)(best >> 32); mj = (int)(best & 0xffffffffll); }
        if (mrow == 0) break;                                              // :763
        if (mrow > max) { max = mrow; max_i = i; max_j = mj; }
        else if (zdrop > 0) {                                              // :767-773
            if (i - max_i > mj - max_j) { if (max - mrow - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
            else { if (max - mrow - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
        }
        // shrink the band for the next row, :775-778.  index `end` itself: eh[end].h = h_last, eh[end].e = 0
        {
            int nb = first_nz >= 0 ? first_nz : end;                       // first j in [beg,end) that is non-zero
            int jl;                                                        // last j in [nb,end] that is non-zero, else nb-1
            if (h_last != 0 && end >= nb) jl = end;
            else if (last_nz >= nb && last_nz >= 0) jl = last_nz;
            else jl = nb - 1;
            beg = nb;
            end = jl + 2 < qlen ? jl + 2 : qlen;
        }
    }
    int i, k;
    if (gscore <= 0 || gscore <= max - end_bonus) { i = max_i; k = max_j; }   // :785-789
    else { i = max_ie; k = qlen - 1; }
    er.qle = k + 1; er.tle = i + 1; er.score = max;
    if (out) { wv::sync(); HP_T0(tb0_); dp_backtrack(cx, nullptr, z, rowb, n_col, w, i, k, *out); HP_TADD(cx, 28, tb0_); }
    cx.n_cells += cells_;                                                  // accounting: DP cell updates (bench.py: GCUPS), once per call
    arena_release(cx.tmp, mark);
    HP_TADD(cx, 26, te0_);
    return er;
}

// =====================================================================================================
// LDS-resident rows.  The H/E row is a circular buffer of HP_LDS_CELLS cells indexed by (column & mask): a row only
// touches the columns [beg, end], beg never decreases and end grows by at most two per row, so the live window is
// narrower than 2w+4 cells.  A column enters the window exactly once, when it first exceeds the high-water mark `hw`;
// it is given the value the reference's first-row initialisation left there (:569-572, :692-694), which is what the
// reference reads from its full-length array when the band reaches a cell it has never written ("stale cells").
// Query bases are staged into LDS 64 at a time as the window advances, target bases 64 rows at a time into a lane
// register; the direction matrix goes to LDS when it fits.  Inside the row loop there is no HBM round trip.
// =====================================================================================================
#define HP_LDS_MASK (HP_LDS_CELLS - 1)

HP_NOINL int ksw_global_lds(Ctx &cx, int qlen, Seq q, int tlen, Seq t,
                            int o_del, int e_del, int o_ins, int e_ins, int w, CigV *out)
{
    ++cx.lds_epoch;                                                         // the rows live in LDS
    long long cells_ = 0;                                                  // in a register: a counter in cx would be a memory round trip per row
    HP_T0(tg0_);
    qlen = wv::uni(qlen); tlen = wv::uni(tlen); w = wv::uni(w);            // wave-uniform: keep them in scalar registers
    o_del = wv::uni(o_del); e_del = wv::uni(e_del); o_ins = wv::uni(o_ins); e_ins = wv::uni(e_ins);
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;                 // :559
    const size_t mark = arena_mark(cx.tmp);
    const bool zl = HP_ZFITS(n_col, tlen);
    uint8_t *z = (out && !zl) ? (uint8_t *)arena_alloc(cx, (size_t)n_col * tlen + 1) : nullptr;
    if (out && !zl && !z) { arena_release(cx.tmp, mark); return 0; }
    HP_L int32_t *LH = cx.lds, *LE = cx.lds + HP_LDS_CELLS;
    HP_L uint8_t *LQ = (HP_L uint8_t *)(cx.lds + 2 * HP_LDS_CELLS), *LZ = LQ + HP_LDS_CELLS;
    HP_G uint8_t *gz = (HP_G uint8_t *)wv::uni64((long long)z);
    const int sc_match = wv::uni(cx.P->match), sc_mis = -wv::uni(cx.P->mis);
    const HP_G uint8_t *gq = (const HP_G uint8_t *)wv::uni64((long long)q.p); const int qs = wv::uni(q.stride);
    const HP_G uint8_t *gt = (const HP_G uint8_t *)wv::uni64((long long)t.p); const int ts = wv::uni(t.stride);
#define HP_GH0(j) ((j) == 0 ? 0 : ((j) <= w ? -(o_ins + e_ins * (j)) : HP_NEG_INF))
    int hw = -1, qw = -1;
    wv::Lane<int> tl;
    WAVE_FOR(l) { tl[l] = 4; }
    // Rows in blocks of 64: the block's target bases are loaded (and waited for) once per block, so that the row loop
    // itself never waits on the vector-memory counter -- which would also wait for the direction-matrix stores of the
    // previous rows whenever that matrix lives in HBM.
    bool stop_rows = false;
    for (int ib = 0; ib < tlen && !stop_rows; ib += 64) {
    { WAVE_FOR(l) { const int ii = ib + l; tl[l] = ii < tlen ? gt[(long)ii * ts] : 4; } }
    const int ti_first = wv::bcast(tl, 0);                                // consumed here: the wait for the load stays outside the row loop
    const int ie = ib + 64 < tlen ? ib + 64 : tlen;
    for (int i = ib; i < ie; ++i) {
        const int ti = i == ib ? ti_first : wv::bcast(tl, i & 63);
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        cells_ += end > beg ? end - beg : 0;
        if (end > hw) {                                                    // columns entering the window
            for (int j0 = hw + 1; j0 <= end; j0 += 64) { WAVE_FOR(l) { const int j = j0 + l; if (j <= end) { LH[j & HP_LDS_MASK] = HP_GH0(j); LE[j & HP_LDS_MASK] = HP_NEG_INF; } } }
            hw = end;
        }
        while (end - 1 > qw) { WAVE_FOR(l) { const int j = qw + 1 + l; if (j < qlen) LQ[j & HP_LDS_MASK] = gq[(long)j * qs]; } qw += 64; }
        if (out && zl) z_row_clear(LZ, i, n_col);
        wv::sync();
        const int h1_init = beg == 0 ? -(o_del + e_del * (i + 1)) : HP_NEG_INF;   // :579
        int carryH = wv::uni(LH[beg & HP_LDS_MASK]);          // H(i-1,beg-1), read before the in-place update below
        int Fin = HP_NEG_INF;                        // F(i,beg)
        wv::sync();
        LH[beg & HP_LDS_MASK] = h1_init;             // eh[beg].h = H(i,beg-1)
        for (int j0 = beg; j0 < end; j0 += 64) {
            const int nxt = j0 + 64;
            const int carry_next = wv::uni(LH[nxt & HP_LDS_MASK]);   // old value, lane 63 is about to overwrite it
            wv::Lane<int> m, e, key;
            WAVE_FOR(l) {
                const int j = j0 + l;
                if (j < end) {
                    const int hm = l == 0 ? carryH : LH[j & HP_LDS_MASK];
                    const int qb = LQ[j & HP_LDS_MASK];
                    m[l] = hm + HP_SUB(ti, qb);
                    e[l] = LE[j & HP_LDS_MASK];
                    key[l] = m[l] - oe_ins + j * e_ins;
                } else { m[l] = 0; e[l] = 0; key[l] = HP_SCAN_IDENT; }
            }
            wv::scan_max_excl(key, HP_SCAN_IDENT);
            wv::sync();
            wv::Lane<int> fnext;
            WAVE_FOR(l) {
                const int j = j0 + l;
                fnext[l] = 0;
                if (j < end) {
                    int f = Fin - l * e_ins;                               // F(i,j) carried in from the left
                    if (l > 0) { const int g = key[l] - (j - 1) * e_ins; f = g > f ? g : f; }
                    int mm = m[l], ee = e[l], h, tt;
                    int dir = mm >= ee ? 0 : 1; h = mm >= ee ? mm : ee;    // ties: M over E   :598-599
                    dir = h >= f ? dir : 2;     h = h >= f ? h : f;        //       then over F :600-601
                    tt = mm - oe_del; ee -= e_del;
                    if (ee > tt) dir |= 1 << 2; else ee = tt;              // :603-607
                    tt = mm - oe_ins; f -= e_ins;
                    if (f > tt) dir |= 2 << 4; else f = tt;                // :608-611
                    LE[j & HP_LDS_MASK] = ee;
                    LH[(j + 1) & HP_LDS_MASK] = h;                         // eh[j+1].h = H(i,j)
                    if (out) { if (zl) z_put(LZ, i, n_col, j - beg, dir); else gz[(long)i * n_col + (j - beg)] = (uint8_t)dir; }
                    fnext[l] = f;
                }
            }
            Fin = wv::bcast(fnext, 63);
            carryH = carry_next;
        }
        LE[end & HP_LDS_MASK] = HP_NEG_INF;                                // :632
        if (end > hw) hw = end;
        wv::sync();
    }
    }
    const int score = tlen > 0 ? wv::uni((int)LH[qlen & HP_LDS_MASK]) : HP_GH0(qlen);
#undef HP_GH0
    if (out) {
        const int i = tlen - 1;
        const int k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;           // :638
        HP_T0(tb0_);
        dp_backtrack(cx, zl ? LZ : nullptr, z, nullptr, n_col, w, i, k, *out);
        HP_TADD(cx, 28, tb0_);
    }
    cx.n_cells += cells_;                                                  // accounting: DP cell updates (bench.py: GCUPS), once per call
    arena_release(cx.tmp, mark);
    HP_TADD(cx, 24, tg0_);
    return score;
}

// =====================================================================================================
// Register-resident rows for short queries (at most 62 bases: most junction jobs).  Column index j of the reference's
// eh[] array lives in lane j: eh[j].h and eh[j].e are two registers, the query base of column j a third.  A row
// reads its own lane (eh[j].h holds H(i-1,j-1)), the new H values move one lane up with a DPP shift, and lanes outside
// [beg, end] simply keep their registers -- the reference's stale cells.  Only the direction matrix touches memory.
// =====================================================================================================
#define HP_REG_QMAX 62

HP_NOINL int ksw_global_reg(Ctx &cx, int qlen, Seq q, int tlen, Seq t,
                            int o_del, int e_del, int o_ins, int e_ins, int w, CigV *out)
{
    long long cells_ = 0;                                                  // in a register: a counter in cx would be a memory round trip per row
    HP_T0(tg0_);
    qlen = wv::uni(qlen); tlen = wv::uni(tlen); w = wv::uni(w);
    o_del = wv::uni(o_del); e_del = wv::uni(e_del); o_ins = wv::uni(o_ins); e_ins = wv::uni(e_ins);
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;                 // :559
    const size_t mark = arena_mark(cx.tmp);
    const bool zl = HP_ZFITS(n_col, tlen);
    uint8_t *z = (out && !zl) ? (uint8_t *)arena_alloc(cx, (size_t)n_col * tlen + 1) : nullptr;
    if (out && !zl && !z) { arena_release(cx.tmp, mark); return 0; }
    HP_L uint8_t *LZ = (HP_L uint8_t *)(cx.lds + 2 * HP_LDS_CELLS) + HP_LDS_CELLS;
    HP_G uint8_t *gz = (HP_G uint8_t *)wv::uni64((long long)z);
    const int sc_match = wv::uni(cx.P->match), sc_mis = -wv::uni(cx.P->mis);
    const HP_G uint8_t *gq = (const HP_G uint8_t *)wv::uni64((long long)q.p); const int qs = wv::uni(q.stride);
    const HP_G uint8_t *gt = (const HP_G uint8_t *)wv::uni64((long long)t.p); const int ts = wv::uni(t.stride);
    wv::Lane<int> Hs, Es, qb, tl;
    WAVE_FOR(l) {
        Hs[l] = l == 0 ? 0 : (l <= w ? -(o_ins + e_ins * l) : HP_NEG_INF);             // first row, :569-572
        Es[l] = HP_NEG_INF;
        qb[l] = l < qlen ? (int)gq[(long)l * qs] : 4;
        tl[l] = 4;
    }
    for (int ib = 0; ib < tlen; ib += 64) {
        { WAVE_FOR(l) { const int ii = ib + l; tl[l] = ii < tlen ? gt[(long)ii * ts] : 4; } }
        const int ti_first = wv::bcast(tl, 0);
        const int ie = ib + 64 < tlen ? ib + 64 : tlen;
        for (int i = ib; i < ie; ++i) {
            const int ti = i == ib ? ti_first : wv::bcast(tl, i & 63);
            const int beg = i > w ? i - w : 0;
            const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
            cells_ += end > beg ? end - beg : 0;
            if (out && zl) { z_row_clear(LZ, i, n_col); wv::sync(); }
            const int h1_init = beg == 0 ? -(o_del + e_del * (i + 1)) : HP_NEG_INF;   // :579
            wv::Lane<int> m, key, hcur;
            WAVE_FOR(l) {
                m[l] = Hs[l] + HP_SUB(ti, qb[l]);
                key[l] = (l >= beg && l < end) ? m[l] - oe_ins + l * e_ins : HP_SCAN_IDENT;
            }
            wv::scan_max_excl(key, HP_SCAN_IDENT);
            WAVE_FOR(l) {
                hcur[l] = 0;
                if (l >= beg && l < end) {
                    int f = HP_NEG_INF - (l - beg) * e_ins;                // F(i,beg) = -inf carried along the row
                    if (l > beg) { const int g = key[l] - (l - 1) * e_ins; f = g > f ? g : f; }
                    int mm = m[l], ee = Es[l], h, tt;
                    int dir = mm >= ee ? 0 : 1; h = mm >= ee ? mm : ee;    // ties: M over E   :598-599
                    dir = h >= f ? dir : 2;     h = h >= f ? h : f;        //       then over F :600-601
                    tt = mm - oe_del; ee -= e_del;
                    if (ee > tt) dir |= 1 << 2; else ee = tt;              // :603-607
                    tt = mm - oe_ins; f -= e_ins;
                    if (f > tt) dir |= 2 << 4;                             // :608-611 (f itself is not needed again: the scan redoes it)
                    Es[l] = ee;
                    hcur[l] = h;
                    if (out) { if (zl) z_put(LZ, i, n_col, l - beg, dir); else gz[(long)i * n_col + (l - beg)] = (uint8_t)dir; }
                }
            }
            wv::shr1(hcur, 0);                                             // eh[j+1].h = H(i,j)
            WAVE_FOR(l) {
                if (l == beg) Hs[l] = h1_init; else if (l > beg && l <= end) Hs[l] = hcur[l];
                if (l == end) Es[l] = HP_NEG_INF;                          // :632
            }
        }
    }
    const int score = wv::bcast(Hs, qlen);
    if (out) {
        const int i = tlen - 1;
        const int k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;           // :638
        wv::sync();
        HP_T0(tb0_);
        dp_backtrack(cx, zl ? LZ : nullptr, z, nullptr, n_col, w, i, k, *out);
        HP_TADD(cx, 28, tb0_);
    }
    cx.n_cells += cells_;                                                  // accounting: DP cell updates (bench.py: GCUPS), once per call
    arena_release(cx.tmp, mark);
    HP_TADD(cx, 24, tg0_);
#ifdef HP_PROF
    HP_TADD(cx, 62, tg0_);
#endif
    return score;
}

HP_NOINL ExtRes ksw_extend_reg(Ctx &cx, int qlen, Seq q, int tlen, Seq t, int w, int h0, CigV *out)
{
    long long cells_ = 0;                                                  // in a register: a counter in cx would be a memory round trip per row
    ExtRes er; er.score = 0; er.qle = 0; er.tle = 0;
    HP_T0(te0_);
    qlen = wv::uni(qlen); tlen = wv::uni(tlen); w = wv::uni(w); h0 = wv::uni(h0);
    const lamsa_hp_para *P = cx.P;
    const int o_ins = wv::uni(P->ins_ext_o), e_ins = wv::uni(P->ins_ext_e), o_del = wv::uni(P->del_ext_o), e_del = wv::uni(P->del_ext_e);
    const int end_bonus = wv::uni(P->end_bonus), zdrop = wv::uni(P->zdrop);
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    const size_t mark = arena_mark(cx.tmp);
    const bool zl = HP_ZFITS(n_col, tlen);
    uint8_t *z = zl ? nullptr : (uint8_t *)arena_alloc(cx, (size_t)n_col * tlen + 1);
    int32_t *rowb = zl ? nullptr : (int32_t *)arena_alloc(cx, sizeof(int32_t) * 2 * ((size_t)tlen + 1));
    if (!zl && (!z || !rowb)) { arena_release(cx.tmp, mark); return er; }
    HP_L uint8_t *LZ = (HP_L uint8_t *)(cx.lds + 2 * HP_LDS_CELLS) + HP_LDS_CELLS;
    HP_G uint8_t *gz = (HP_G uint8_t *)wv::uni64((long long)z);
    HP_G int32_t *growb = (HP_G int32_t *)wv::uni64((long long)rowb);
    const int sc_match = wv::uni(P->match), sc_mis = -wv::uni(P->mis);
    const HP_G uint8_t *gq = (const HP_G uint8_t *)wv::uni64((long long)q.p); const int qs = wv::uni(q.stride);
    const HP_G uint8_t *gt = (const HP_G uint8_t *)wv::uni64((long long)t.p); const int ts = wv::uni(t.stride);
    const int h1v = h0 > oe_ins ? h0 - oe_ins : 0;
    wv::Lane<int> Hs, Es, qb, tl;
    WAVE_FOR(l) {                                                          // first row, :692-694
        Hs[l] = l == 0 ? h0 : (l == 1 ? h1v : ((l <= qlen && h1v - (l - 2) * e_ins > e_ins) ? h1v - (l - 1) * e_ins : 0));
        Es[l] = 0;
        qb[l] = l < qlen ? (int)gq[(long)l * qs] : 4;
        tl[l] = 4;
    }
    int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1;
    int beg = 0, end = qlen;
    bool stop_rows = false;
    for (int ib = 0; ib < tlen && !stop_rows; ib += 64) {
        { WAVE_FOR(l) { const int ii = ib + l; tl[l] = ii < tlen ? gt[(long)ii * ts] : 4; } }
        const int ti_first = wv::bcast(tl, 0);
        const int ie = ib + 64 < tlen ? ib + 64 : tlen;
        for (int i = ib; i < ie; ++i) {
            const int ti = i == ib ? ti_first : wv::bcast(tl, i & 63);
            const int d_beg = i > w ? i - w : 0;
            if (beg < i - w) beg = i - w;                                  // :718-720
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
            cells_ += end > beg ? end - beg : 0;
            if (zl) { z_row_clear(LZ, i, n_col); wv::sync(); }
            else { growb[2 * i] = beg; growb[2 * i + 1] = end; }
            int h1_init;
            if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
            else h1_init = 0;
            wv::Lane<int> m, key, hcur;
            WAVE_FOR(l) {
                const int hm = Hs[l];
                const int M = hm ? hm + HP_SUB(ti, qb[l]) : 0;             // :737
                int tt = M - oe_ins; tt = tt > 0 ? tt : 0;
                m[l] = M;
                key[l] = (l >= beg && l < end) ? tt + l * e_ins : HP_SCAN_IDENT;
            }
            wv::scan_max_excl(key, HP_SCAN_IDENT);
            WAVE_FOR(l) {
                hcur[l] = -1;
                if (l >= beg && l < end) {
                    int f = 0 - (l - beg) * e_ins;                         // F(i,beg) = 0 carried along the row
                    if (l > beg) { const int g = key[l] - (l - 1) * e_ins; f = g > f ? g : f; }
                    int M = m[l], ee = Es[l], h, tt;
                    int dir = M > ee ? 0 : 1; h = M > ee ? M : ee;          // ties: E over M   :738-739
                    dir = h > f ? dir : 2;    h = h > f ? h : f;            //       F over both :740-741
                    tt = M - oe_del; tt = tt > 0 ? tt : 0; ee -= e_del;
                    if (ee > tt) dir |= 1 << 2; else ee = tt;               // :745-750
                    tt = M - oe_ins; tt = tt > 0 ? tt : 0; f -= e_ins;
                    if (f > tt) dir |= 2 << 4;                              // :751-755
                    Es[l] = ee;
                    hcur[l] = h;
                    if (zl) z_put(LZ, i, n_col, l - d_beg, dir); else gz[(long)i * n_col + (l - d_beg)] = (uint8_t)dir;
                }
            }
            // row maximum, last j among equals (:743-744)
            int mrow = 0, mj = -1;
            {
                const int hmax = wv::reduce_max(hcur);
                if (hmax >= 0) {
                    wv::Lane<int> eq;
                    WAVE_FOR(l) eq[l] = hcur[l] == hmax;
                    mrow = hmax; mj = 63 - __builtin_clzll(wv::ballot(eq));
                }
            }
            const int h_last = beg < end ? wv::bcast(hcur, end - 1) : h1_init;   // H(i,end-1), or the first-column value when the row is empty
            WAVE_FOR(l) { if (hcur[l] < 0) hcur[l] = 0; }
            wv::shr1(hcur, 0);                                             // eh[j+1].h = H(i,j)
            wv::Lane<int> nz;
            WAVE_FOR(l) {
                if (beg < end) { if (l == beg) Hs[l] = h1_init; else if (l > beg && l <= end) Hs[l] = hcur[l]; }
                else if (l == end) Hs[l] = h1_init;                        // eh[end].h = h1 when the row is empty (:758)
                if (l == end) Es[l] = 0;                                   // :758
                nz[l] = l >= beg && l <= end && (Hs[l] != 0 || Es[l] != 0);
            }
            const int jj = beg < end ? end : beg;                          // loop variable j after the row
            if (jj == qlen) {                                              // :759-762
                max_ie = gscore > h_last ? max_ie : i;
                gscore = gscore > h_last ? gscore : h_last;
            }
            if (mrow == 0) { stop_rows = true; break; }                    // :763
            if (mrow > max) { max = mrow; max_i = i; max_j = mj; }
            else if (zdrop > 0) {                                          // :767-773
                if (i - max_i > mj - max_j) { if (max - mrow - ((i - max_i) - (mj - max_j)) * e_del > zdrop) { stop_rows = true; break; } }
                else { if (max - mrow - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) { stop_rows = true; break; } }
            }
            // shrink the band for the next row, :775-778
            {
                const unsigned long long nzm = wv::ballot(nz);
                const unsigned long long lowm = nzm & (end < 64 ? ((1ull << end) - 1) : ~0ull);      // non-zero indices in [beg, end)
                const int nb = lowm ? __builtin_ctzll(lowm) : end;
                const unsigned long long upm = nzm & ~((1ull << nb) - 1);                             // non-zero indices in [nb, end]
                const int jl = upm ? 63 - __builtin_clzll(upm) : nb - 1;
                beg = nb;
                end = jl + 2 < qlen ? jl + 2 : qlen;
            }
        }
    }
    int i, k;
    if (gscore <= 0 || gscore <= max - end_bonus) { i = max_i; k = max_j; }   // :785-789
    else { i = max_ie; k = qlen - 1; }
    er.qle = k + 1; er.tle = i + 1; er.score = max;
    if (out) { wv::sync(); HP_T0(tb0_); dp_backtrack(cx, zl ? LZ : nullptr, z, rowb, n_col, w, i, k, *out); HP_TADD(cx, 28, tb0_); }
    cx.n_cells += cells_;                                                  // accounting: DP cell updates (bench.py: GCUPS), once per call
    arena_release(cx.tmp, mark);
    HP_TADD(cx, 26, te0_);
#ifdef HP_PROF
    HP_TADD(cx, 60, te0_);
#endif
    return er;
}

// ksw_extend_core for queries of 63 .. 64 * NS - 2 bases: the row in NS registers per lane -- column j of the reference's eh[] array is
// lane (j & 63) of register set (j >> 6) -- so that a row is one pass over the sets instead of NS trips through the LDS row with the
// scalar bookkeeping of a tile each (~1 050 instructions per row there for two tiles, ~450 here; the junction extensions of a noisy
// read, 75 .. 250 query bases, are most of the fill kernel's DP time).  The sets are walked in order: the F scan of a set is topped up
// with the maximum of the sets before it, and only H of the row survives a set's pass, so the live registers are three per set.
// Same recurrences, tie rules, band and z-drop logic as ksw_extend_reg.
#define HP_REGN_SETS 4
#define HP_REGN_QMAX(ns) (64 * (ns) - 2)
HP_INL unsigned long long lt_mask64(int x) { return x <= 0 ? 0ull : (x >= 64 ? ~0ull : ((1ull << x) - 1)); }      // bits below x
template <int NS>
HP_NOINL ExtRes ksw_extend_regn(Ctx &cx, int qlen, Seq q, int tlen, Seq t, int w, int h0, CigV *out)
{
    long long cells_ = 0;                                                  // in a register: a counter in cx would be a memory round trip per row
    ExtRes er; er.score = 0; er.qle = 0; er.tle = 0;
    HP_T0(te0_);
    qlen = wv::uni(qlen); tlen = wv::uni(tlen); w = wv::uni(w); h0 = wv::uni(h0);
    const lamsa_hp_para *P = cx.P;
    const int o_ins = wv::uni(P->ins_ext_o), e_ins = wv::uni(P->ins_ext_e), o_del = wv::uni(P->del_ext_o), e_del = wv::uni(P->del_ext_e);
    const int end_bonus = wv::uni(P->end_bonus), zdrop = wv::uni(P->zdrop);
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    const size_t mark = arena_mark(cx.tmp);
    const bool zl = HP_ZFITS(n_col, tlen);
    uint8_t *z = zl ? nullptr : (uint8_t *)arena_alloc(cx, (size_t)n_col * tlen + 1);
    int32_t *rowb = zl ? nullptr : (int32_t *)arena_alloc(cx, sizeof(int32_t) * 2 * ((size_t)tlen + 1));
    if (!zl && (!z || !rowb)) { arena_release(cx.tmp, mark); return er; }
#ifdef HP_PROF
    if (!zl && cx.prof) { cx.prof[52] += (long long)n_col * tlen; cx.prof[53] += 1; }
#endif
    HP_L uint8_t *LZ = (HP_L uint8_t *)(cx.lds + 2 * HP_LDS_CELLS) + HP_LDS_CELLS;
    HP_G uint8_t *gz = (HP_G uint8_t *)wv::uni64((long long)z);
    HP_G int32_t *growb = (HP_G int32_t *)wv::uni64((long long)rowb);
    const int sc_match = wv::uni(P->match), sc_mis = -wv::uni(P->mis);
    const HP_G uint8_t *gq = (const HP_G uint8_t *)wv::uni64((long long)q.p); const int qs = wv::uni(q.stride);
    const HP_G uint8_t *gt = (const HP_G uint8_t *)wv::uni64((long long)t.p); const int ts = wv::uni(t.stride);
    const int h1v = h0 > oe_ins ? h0 - oe_ins : 0;
    wv::Lane<int> Hs[NS], Es[NS], hcur[NS], qbp, je0, tl;
    WAVE_FOR(l) { qbp[l] = 0; je0[l] = l * e_ins; tl[l] = 4; }
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        WAVE_FOR(l) {                                                      // first row, :692-694
            const int j = 64 * c + l;
            Hs[c][l] = j == 0 ? h0 : (j == 1 ? h1v : ((j <= qlen && h1v - (j - 2) * e_ins > e_ins) ? h1v - (j - 1) * e_ins : 0));
            Es[c][l] = 0;
            qbp[l] |= (j < qlen ? (int)gq[(long)j * qs] : 4) << (3 * c);   // the query codes of a lane's columns, three bits per set
        }
    }
    int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1;
    int beg = 0, end = qlen;
    bool stop_rows = false;
    for (int ib = 0; ib < tlen && !stop_rows; ib += 64) {
        { WAVE_FOR(l) { const int ii = ib + l; tl[l] = ii < tlen ? gt[(long)ii * ts] : 4; } }
        const int ti_first = wv::bcast(tl, 0);
        const int ie = ib + 64 < tlen ? ib + 64 : tlen;
        for (int i = ib; i < ie; ++i) {
            const int ti = i == ib ? ti_first : wv::bcast(tl, i & 63);
            const int d_beg = i > w ? i - w : 0;
            if (beg < i - w) beg = i - w;                                  // :718-720
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
            cells_ += end > beg ? end - beg : 0;
            if (zl) { z_row_clear(LZ, i, n_col); wv::sync(); }
            else { growb[2 * i] = beg; growb[2 * i + 1] = end; }
            int h1_init;
            if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
            else h1_init = 0;
            const int fbase = beg * e_ins;                                 // F(i,beg) = 0 carried along the row: (beg - j) * e_ins at column j
            int carry = HP_SCAN_IDENT;                                     // maximum of the scan keys of the sets below
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                const int jc = 64 * c * e_ins;
                wv::Lane<int> m, key;
                WAVE_FOR(l) {
                    const int j = 64 * c + l;
                    const int hm = Hs[c][l], qb = (qbp[l] >> (3 * c)) & 7;
                    const int M = hm ? hm + HP_SUB(ti, qb) : 0;            // :737
                    int tt = M - oe_ins; tt = tt > 0 ? tt : 0;
                    m[l] = M;
                    key[l] = (j >= beg && j < end) ? tt + je0[l] + jc : HP_SCAN_IDENT;
                }
                // F along the row: an exclusive prefix maximum over all the columns = the scan of each set, topped up with the maximum
                // of the sets before it
                const int top = wv::scan_max_excl_top(key, HP_SCAN_IDENT);
                WAVE_FOR(l) {
                    const int j = 64 * c + l;
                    hcur[c][l] = -1;
                    if (j >= beg && j < end) {
                        const int je = je0[l] + jc;
                        int f = fbase - je;
                        if (j > beg) { const int kk = key[l] > carry ? key[l] : carry; const int g = kk - je + e_ins; f = g > f ? g : f; }
                        int M = m[l], ee = Es[c][l], h, tt;
                        int dir = M > ee ? 0 : 1; h = M > ee ? M : ee;      // ties: E over M   :738-739
                        dir = h > f ? dir : 2;    h = h > f ? h : f;        //       F over both :740-741
                        tt = M - oe_del; tt = tt > 0 ? tt : 0; ee -= e_del;
                        if (ee > tt) dir |= 1 << 2; else ee = tt;           // :745-750
                        tt = M - oe_ins; tt = tt > 0 ? tt : 0; f -= e_ins;
                        if (f > tt) dir |= 2 << 4;                          // :751-755
                        Es[c][l] = ee;
                        hcur[c][l] = h;
                        if (zl) z_put(LZ, i, n_col, j - d_beg, dir); else gz[(long)i * n_col + (j - d_beg)] = (uint8_t)dir;
                    }
                }
                carry = top > carry ? top : carry;
            }
            // row maximum, last j among equals (:743-744)
            int mrow = 0, mj = -1;
            {
                wv::Lane<int> hm;
                WAVE_FOR(l) {
                    int v = hcur[0][l];
#pragma unroll
                    for (int c = 1; c < NS; ++c) v = hcur[c][l] > v ? hcur[c][l] : v;
                    hm[l] = v;
                }
                const int hmax = wv::reduce_max(hm);
                if (hmax >= 0) {
                    mrow = hmax;
                    bool got = false;
#pragma unroll
                    for (int c = NS - 1; c >= 0; --c) {
                        if (got) continue;
                        wv::Lane<int> eq;
                        WAVE_FOR(l) eq[l] = hcur[c][l] == hmax;
                        const unsigned long long b = wv::ballot(eq);
                        if (b) { mj = 64 * c + 63 - __builtin_clzll(b); got = true; }
                    }
                }
            }
            int h_last = h1_init;                                          // H(i,end-1), or the first-column value when the row is empty
            if (beg < end) {
#pragma unroll
                for (int c = 0; c < NS; ++c) if (((end - 1) >> 6) == c) h_last = wv::bcast(hcur[c], (end - 1) & 63);
            }
#pragma unroll
            for (int c = 0; c < NS; ++c) { WAVE_FOR(l) { if (hcur[c][l] < 0) hcur[c][l] = 0; } }
#pragma unroll
            for (int c = NS - 1; c >= 0; --c) {                            // eh[j+1].h = H(i,j), across the sets
                const int below = c > 0 ? wv::bcast(hcur[c > 0 ? c - 1 : 0], 63) : 0;
                wv::shr1(hcur[c], below);
            }
            unsigned long long nzm[NS];
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                wv::Lane<int> nz;
                WAVE_FOR(l) {
                    const int j = 64 * c + l;
                    if (beg < end) { if (j == beg) Hs[c][l] = h1_init; else if (j > beg && j <= end) Hs[c][l] = hcur[c][l]; }
                    else if (j == end) Hs[c][l] = h1_init;                 // eh[end].h = h1 when the row is empty (:758)
                    if (j == end) Es[c][l] = 0;                            // :758
                    nz[l] = j >= beg && j <= end && (Hs[c][l] != 0 || Es[c][l] != 0);
                }
                nzm[c] = wv::ballot(nz);
            }
            const int jj = beg < end ? end : beg;                          // loop variable j after the row
            if (jj == qlen) {                                              // :759-762
                max_ie = gscore > h_last ? max_ie : i;
                gscore = gscore > h_last ? gscore : h_last;
            }
            if (mrow == 0) { stop_rows = true; break; }                    // :763
            if (mrow > max) { max = mrow; max_i = i; max_j = mj; }
            else if (zdrop > 0) {                                          // :767-773
                if (i - max_i > mj - max_j) { if (max - mrow - ((i - max_i) - (mj - max_j)) * e_del > zdrop) { stop_rows = true; break; } }
                else { if (max - mrow - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) { stop_rows = true; break; } }
            }
            // shrink the band for the next row, :775-778.  The ballots hold eh[j] != 0 for j in [beg, end] only: "the first one below end" is the
            // first one once index `end` is set aside, "the last one from there on" the last one of all -- no range masks.
            {
                const int e_set = end >> 6;
                const unsigned long long e_bit = 1ull << (end & 63);
                bool end_nz = false, got = false;
                int nb = end, jl = -1;
#pragma unroll
                for (int c = 0; c < NS; ++c) {
                    unsigned long long lo = nzm[c];
                    if (c == e_set) { end_nz = lo & e_bit; lo &= ~e_bit; }
                    if (!got && lo) { nb = 64 * c + __builtin_ctzll(lo); got = true; }
                }
                if (got) {
                    bool gl = false;
#pragma unroll
                    for (int c = NS - 1; c >= 0; --c) if (!gl && nzm[c]) { jl = 64 * c + 63 - __builtin_clzll(nzm[c]); gl = true; }
                } else jl = end_nz ? end : end - 1;
                beg = nb;
                end = jl + 2 < qlen ? jl + 2 : qlen;
            }
        }
    }
    int i, k;
    if (gscore <= 0 || gscore <= max - end_bonus) { i = max_i; k = max_j; }   // :785-789
    else { i = max_ie; k = qlen - 1; }
    er.qle = k + 1; er.tle = i + 1; er.score = max;
    if (out) { wv::sync(); HP_T0(tb0_); dp_backtrack(cx, zl ? LZ : nullptr, z, rowb, n_col, w, i, k, *out); HP_TADD(cx, 28, tb0_); }
    cx.n_cells += cells_;                                                  // accounting: DP cell updates (bench.py: GCUPS), once per call
    arena_release(cx.tmp, mark);
    HP_TADD(cx, 26, te0_);
    return er;
}

// Scores as int16 pairs (packed math, wave.h pk::): two cells per instruction.  Comparisons become sign bits of differences; every quantity
// is a sum of a few scores and penalties, and a job only goes to a packed routine (ksw_extend_band, ksw_global_pk) when none of them can
// leave the int16 range (pkb_extend_ok, pk_global_ok); the int32 register sets and the LDS rows stand behind them.
#define HP_PK_QMAX(ns) (128 * (ns) - 2)
#ifndef HP_PK_RT
#define HP_PK_RT 1                          // the tests' CPU build switches the packed routines off to reach the int32 register sets behind them
#endif
#define HP_PK_IDENT (-16000)

// ksw_extend_core for LONG queries -- the end extensions of a line, up to the whole read (frag_head_bound_fix / frag_tail_bound_fix,
// src/frag_check.c:576-707) -- with the row's live WINDOW in registers, int16 pairs, 2 * NS consecutive columns per lane.
// A row of the extension only touches the columns [beg, end], at most 2w + 2 of them, and the window only moves right: column j of the
// reference's eh[] array lives in slot j mod (128 * NS) -- lane (j / (2 NS)) mod 64, register (j mod 2 NS) / 2, half j & 1 -- so a lane holds
// 2 NS neighbouring columns (NS registers of pairs), then the 2 NS columns 128 NS further on, and so on; the slots of columns the band has
// left behind are given to the columns ahead of it eight lanes at a time, with the value the reference's first-row initialisation left
// there (:692-694), i.e. what it reads from its full-length array when the band reaches a cell it never wrote.
// What that buys over the LDS tiles (ksw_extend_lds: 64 columns per pass, ~90 instructions and eight LDS operations each, four passes
// for a band of 201): everything a row does across lanes is done ONCE per row whatever NS -- the F scan (a lane scans its own columns,
// then one exclusive prefix maximum over the lanes' totals, in two halves because the window wraps round the wave), the row maximum
// (one reduction of (H << 16 | column) keys: "last column among equals" is the larger key, :743-744), the one-column shift of H (inside
// a lane but for one wave rotation), the first and last non-zero cell (:775-778: a ballot over the lanes, then two lanes' bit patterns).
// Same recurrences, tie rules, band and z-drop logic as ksw_extend_reg; scores are bounded (pkb_extend_ok), the scan's keys use
// columns relative to the row's first one.  The direction matrix is a nibble per cell in the wave's slab, a row = 64 * NS bytes indexed by
// the slot, with the band limits of every row beside it.
// sets for a query of qlen columns under band w: the window holds the band + 2 + the eight lanes being refilled + one lane of slack -- or the
// whole query, and then never moves
HP_INL int pkb_sets_q(int qlen, int w)
{
    for (int ns = 1; ns <= 4; ns *= 2) if (2 * w + 3 + 18 * ns <= 128 * ns || qlen + 3 <= 128 * ns) return ns;
    return 0;
}
HP_INL bool pkb_extend_ok(const lamsa_hp_para *P, int qlen, int h0, int ns)
{
    const int mx = P->match > P->mis ? P->match : P->mis;
    const int pen = (P->ins_ext_o > P->del_ext_o ? P->ins_ext_o : P->del_ext_o) + (P->ins_ext_e > P->del_ext_e ? P->ins_ext_e : P->del_ext_e);
    const int ext = P->ins_ext_e > P->del_ext_e ? P->ins_ext_e : P->del_ext_e;
    return ns > 0 && mx > 0 && mx < 256 && pen >= 0 && pen < 4000 && P->ins_ext_e >= 0 && P->del_ext_e >= 0 && P->ins_ext_o >= 0 && P->del_ext_o >= 0 &&
           (long long)h0 + (long long)qlen * mx < 23000 && (long long)(128 * ns + 2) * ext < 8000 && qlen + 128 * ns < 32000;
}
template <int NS>
HP_NOINL ExtRes ksw_extend_band(Ctx &cx, int qlen, Seq q, int tlen, Seq t, int w, int h0, CigV *out)
{
    long long cells_ = 0;
    ExtRes er; er.score = 0; er.qle = 0; er.tle = 0;
    HP_T0(te0_);
    qlen = wv::uni(qlen); tlen = wv::uni(tlen); w = wv::uni(w); h0 = wv::uni(h0);
    const lamsa_hp_para *P = cx.P;
    const int o_ins = wv::uni(P->ins_ext_o), e_ins = wv::uni(P->ins_ext_e), o_del = wv::uni(P->del_ext_o), e_del = wv::uni(P->del_ext_e);
    const int end_bonus = wv::uni(P->end_bonus), zdrop = wv::uni(P->zdrop);
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    constexpr int LC = 2 * NS, WN = 128 * NS, zs = 64 * NS;              // columns per lane, slots of the window, bytes of a row of the direction matrix
    const size_t mark = arena_mark(cx.tmp);
    uint8_t *z = (uint8_t *)arena_alloc(cx, (size_t)zs * tlen + 16);
    int32_t *rowb = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 2 * ((size_t)tlen + 1));
    if (!z || !rowb) { arena_release(cx.tmp, mark); return er; }
#ifdef HP_PROF
    if (cx.prof) { cx.prof[52] += (long long)zs * tlen; cx.prof[53] += 1; }
#endif
    HP_G uint8_t *gz = (HP_G uint8_t *)wv::uni64((long long)z);
    HP_G int32_t *growb = (HP_G int32_t *)wv::uni64((long long)rowb);
    const int sc_match = wv::uni(P->match), sc_mis = -wv::uni(P->mis);
    const HP_G uint8_t *gq = (const HP_G uint8_t *)wv::uni64((long long)q.p); const int qs = wv::uni(q.stride);
    const HP_G uint8_t *gt = (const HP_G uint8_t *)wv::uni64((long long)t.p); const int ts = wv::uni(t.stride);
    const int h1v = h0 > oe_ins ? h0 - oe_ins : 0;
    const int OEI = pk::rep(oe_ins), OED = pk::rep(oe_del), EI = pk::rep(e_ins), ED = pk::rep(e_del);
    const int DSC = pk::rep(sc_match - sc_mis), MIS = pk::rep(sc_mis), IDENT = pk::rep(HP_PK_IDENT);
#define HP_PKB_EH0(j) ((j) == 0 ? h0 : ((j) == 1 ? h1v : (((j) <= qlen && h1v - ((j) - 2) * e_ins > e_ins) ? h1v - ((j) - 1) * e_ins : 0)))      /* first row, :692-694 */
    // a lane's columns jb .. jb + LC - 1: cells, query bases (one bit of four per column; N and beyond the query: score -1)
#define HP_PKB_LOAD(jb_) do { \
        _Pragma("unroll") for (int r_ = 0; r_ < NS; ++r_) { \
            int hv_[2], oh_ = 0, nn_ = 0; \
            _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_) { \
                const int j_ = (jb_) + 2 * r_ + b_; \
                hv_[b_] = HP_PKB_EH0(j_); \
                const int code_ = j_ < qlen ? (int)gq[(long)j_ * qs] : 4; \
                if (code_ < 4) oh_ |= 1 << (code_ + 16 * b_); else nn_ |= (int)(0xffffu << (16 * b_)); \
            } \
            Hs[r_][l] = pk::pack(hv_[0], hv_[1]); Es[r_][l] = 0; qoh[r_][l] = oh_; qN[r_][l] = nn_; \
        } \
        JB[l] = pk::rep(jb_); } while (0)
    wv::Lane<int> Hs[NS], Es[NS], qoh[NS], qN[NS], hcur[NS], M[NS], INB[NS], JRE[NS], pre[NS], JB, tl;
    WAVE_FOR(l) { tl[l] = 4; HP_PKB_LOAD(LC * l); }
    int top = WN;                                                          // columns [0, top) have been given their slots
    int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1;
    int beg = 0, end = qlen;
    bool stop_rows = false;
    for (int ib = 0; ib < tlen && !stop_rows; ib += 64) {
        { WAVE_FOR(l) { const int ii = ib + l; tl[l] = ii < tlen ? gt[(long)ii * ts] : 4; } }
        const int ti_first = wv::bcast(tl, 0);
        const int ie = ib + 64 < tlen ? ib + 64 : tlen;
        for (int i = ib; i < ie; ++i) {
            const int ti = i == ib ? ti_first : wv::bcast(tl, i & 63);
            if (beg < i - w) beg = i - w;                                  // :718-720
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
            cells_ += end > beg ? end - beg : 0;
            { WAVE_FOR(l) { if (l < 2) growb[2 * i + l] = l ? end : beg; } }
            int h1_init;
            if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
            else h1_init = 0;
            if (beg >= end) {
                // the row is empty: eh[end] = {h1, 0} (:758), its maximum is 0 and the loop ends (:763) -- nothing reads the cells again
                if (beg == qlen) { max_ie = gscore > h1_init ? max_ie : i; gscore = gscore > h1_init ? gscore : h1_init; }       // :759-762 (the loop variable stands at beg)
                stop_rows = true; break;
            }
            if (end + 3 > top) {                                           // the band's right edge (next row's at most two further) nears the loaded columns: eight more lanes
                const int lr = (top / LC) & 63;
                WAVE_FOR(l) { const int d = (l - lr) & 63; if (d < 8) { const int jb = top + d * LC; HP_PKB_LOAD(jb); } }
                top += 8 * LC;
            }
            const int tsh = ti & 3, tN = ti > 3 ? -1 : 0;                   // a target N scores -1 against everything
            const int BEG = pk::rep(beg), END = pk::rep(end), H1 = pk::rep(h1_init);
            const int l0 = (beg / LC) & 63;                                // the lane of the window's first column: lanes l0 .. 63, then 0 .. l0 - 1, hold ascending columns
            // ---- the lane's own columns: M, the scan keys max(M - oe_ins, 0) + (j - beg) * e_ins and their running maximum
            wv::Lane<int> ka, kb;
            WAVE_FOR(l) {
                int run = HP_PK_IDENT;
#pragma unroll
                for (int r = 0; r < NS; ++r) {
                    const int jp = pk::add(JB[l], pk::pack(2 * r, 2 * r + 1));
                    const int in = pk::neg_mask(pk::sub(jp, END)) & ~pk::neg_mask(pk::sub(jp, BEG));       // beg <= j < end, per half
                    const int eq = (qoh[r][l] >> tsh) & 0x00010001;
                    const int S = pk::add(pk::mul(eq, DSC), MIS) | qN[r][l] | tN;                          // HP_SUB(ti, qb)
                    const int hm = Hs[r][l];
                    const int m = pk::mul(pk::add(hm, S), pk::min_u(hm, 0x00010001));                      // hm ? hm + S : 0   (:737; hm is never negative)
                    const int t1 = pk::max(pk::sub(m, OEI), 0);
                    const int jre = pk::mul(pk::sub(jp, BEG), EI);
                    const int k = pk::sel(in, pk::add(t1, jre), IDENT);
                    const int klo = pk::lo(k), khi = pk::hi(k);
                    const int p0 = run; run = run > klo ? run : klo;
                    const int p1 = run; run = run > khi ? run : khi;
                    M[r][l] = m; INB[r][l] = in; JRE[r][l] = jre; pre[r][l] = pk::pack(p0, p1);
                }
                ka[l] = l >= l0 ? run : HP_PK_IDENT; kb[l] = l < l0 ? run : HP_PK_IDENT;
            }
            // F along the row: the exclusive prefix maximum over the lanes before this one in column order
            // (one scan while the window has not moved, the lanes then being in column order as they are, was measured: no gain)
            const int topA = wv::scan_max_excl_top(ka, HP_PK_IDENT);
            wv::scan_max_excl(kb, HP_PK_IDENT);
            wv::Lane<int> best;
            WAVE_FOR(l) {
                const int pl = l >= l0 ? ka[l] : (topA > kb[l] ? topA : kb[l]);
                const int PP = pk::rep(pl);
                int zw = 0, bk = -1;
#pragma unroll
                for (int r = 0; r < NS; ++r) {
                    const int jre = JRE[r][l], in = INB[r][l], m = M[r][l];
                    const int pr = pk::max(pre[r][l], PP);
                    int f = pk::max(pk::add(pk::sub(pr, jre), EI), pk::sub(0, jre));                        // F(i,beg) = 0 carried along the row
                    const int tI = pk::max(pk::sub(m, OEI), 0);
                    int ee = Es[r][l];
                    const int m1 = pk::neg_mask(pk::sub(ee, m));                                           // M > E
                    int h = pk::max(m, ee);                                                                // ties: E over M   :738-739
                    const int m2 = pk::neg_mask(pk::sub(f, h));                                            // h > F
                    int d = pk::sel(m2, ~m1 & 0x00010001, 0x00020002);                                     //       F over both :740-741
                    h = pk::max(h, f);
                    const int tD = pk::max(pk::sub(m, OED), 0);
                    ee = pk::sub(ee, ED);
                    d |= pk::neg_mask(pk::sub(tD, ee)) & 0x00040004;                                       // E extends, :745-750
                    ee = pk::max(ee, tD);
                    f = pk::sub(f, EI);
                    d |= pk::neg_mask(pk::sub(tI, f)) & 0x00080008;                                        // F extends, :751-755
                    Es[r][l] = pk::sel(in, ee, Es[r][l]);
                    const int hc = pk::sel(in, h, -1);
                    hcur[r][l] = hc;
                    zw |= ((d | (d >> 12)) & 0xff) << (8 * r);
                    // row maximum, last column among equals (:743-744): the larger of (H << 16 | column); a column outside the band gives a negative key
                    const int jp = pk::add(JB[l], pk::pack(2 * r, 2 * r + 1));
                    const int k0 = (int)(((unsigned)hc << 16) | ((unsigned)jp & 0xffffu)), k1 = (int)(((unsigned)hc & 0xffff0000u) | ((unsigned)jp >> 16));
                    bk = bk > k0 ? bk : k0; bk = bk > k1 ? bk : k1;
                }
                best[l] = bk;
                if constexpr (NS == 1) gz[(long)i * zs + l] = (uint8_t)zw;
                else if constexpr (NS == 2) *(HP_G uint16_t *)(gz + (long)i * zs + 2 * l) = (uint16_t)zw;
                else *(HP_G uint32_t *)(gz + (long)i * zs + 4 * l) = (uint32_t)zw;
            }
            int mrow = 0, mj = -1;
            { const int b = wv::reduce_max(best); if (b >= 0) { mrow = b >> 16; mj = b & 0xffff; } }
            int h_last;                                                     // H(i, end - 1)
            {
                const int le = ((end - 1) / LC) & 63, sl = (end - 1) % LC;
                int v = 0;
#pragma unroll
                for (int r = 0; r < NS; ++r) if ((sl >> 1) == r) v = wv::bcast(hcur[r], le);
                h_last = (sl & 1) ? pk::hi(v) : pk::lo(v);
            }
            // eh[j+1].h = H(i,j): the row one column up -- inside the lane, and the lane's first column from the lane below's last
            wv::Lane<int> rot = hcur[NS - 1], LB;
            wv::ror1(rot);
            WAVE_FOR(l) {
                int bits = 0;
#pragma unroll
                for (int r = 0; r < NS; ++r) {
                    const int below = r > 0 ? hcur[r > 0 ? r - 1 : 0][l] : rot[l];
                    const int hsh = pk::shift_up(hcur[r][l], below);
                    const int in = INB[r][l];
                    const int upd = ~pk::neg_mask(hsh);                    // beg < j <= end (columns the row did not compute arrive as -1)
                    int hs = Hs[r][l], es = Es[r][l];
                    hs = pk::sel(upd, hsh, hs);
                    hs = pk::sel(in & ~upd, H1, hs);                       // j == beg
                    es &= ~(upd & ~in);                                    // eh[end].e = 0, :758
                    Hs[r][l] = hs; Es[r][l] = es;
                    const int nzh = pk::min_u((hs | es) & (in | upd), 0x00010001);      // eh[j] not zero, j in [beg, end]: one bit per half
                    bits |= ((nzh | (nzh >> 15)) & 3) << (2 * r);
                }
                LB[l] = bits;
            }
            if (end == qlen) {                                             // :759-762
                max_ie = gscore > h_last ? max_ie : i;
                gscore = gscore > h_last ? gscore : h_last;
            }
            if (mrow == 0) { stop_rows = true; break; }                    // :763
            if (mrow > max) { max = mrow; max_i = i; max_j = mj; }
            else if (zdrop > 0) {                                          // :767-773
                if (i - max_i > mj - max_j) { if (max - mrow - ((i - max_i) - (mj - max_j)) * e_del > zdrop) { stop_rows = true; break; } }
                else { if (max - mrow - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) { stop_rows = true; break; } }
            }
            // shrink the band for the next row, :775-778: the first non-zero eh[j] below `end`, the last one from there up to `end`
            {
                const unsigned long long any = wv::ballot(LB);
                int nb = end, jl = end - 1;
                if (any) {
                    const unsigned long long rt = l0 ? ((any >> l0) | (any << (64 - l0))) : any;          // bit d: the lane d lanes after l0
                    const int df = __builtin_ctzll(rt), dl = 63 - __builtin_clzll(rt);
                    const int bf = wv::bcast(LB, (l0 + df) & 63), bl = wv::bcast(LB, (l0 + dl) & 63);
                    const int bb = beg / LC;
                    const int jf = (bb + df) * LC + __builtin_ctz((unsigned)bf), jx = (bb + dl) * LC + (31 - __builtin_clz((unsigned)bl));
                    if (jf < end) nb = jf;
                    jl = jx;                                               // (index `end` alone: nb = end, jl = end)
                }
                beg = nb;
                end = jl + 2 < qlen ? jl + 2 : qlen;
            }
        }
    }
#undef HP_PKB_LOAD
#undef HP_PKB_EH0
    int i, k;
    if (gscore <= 0 || gscore <= max - end_bonus) { i = max_i; k = max_j; }   // :785-789
    else { i = max_ie; k = qlen - 1; }
    er.qle = k + 1; er.tle = i + 1; er.score = max;
    if (out) { wv::sync(); HP_T0(tb0_); dp_backtrack(cx, nullptr, z, rowb, n_col, w, i, k, *out, zs); HP_TADD(cx, 28, tb0_); }
    cx.n_cells += cells_;
    arena_release(cx.tmp, mark);
    HP_TADD(cx, 26, te0_);
    return er;
}

// ksw_global2 with two columns per lane, scores as int16 pairs: the layout of round 3's extension routine (column j = half j & 1 of lane (j >> 1) & 63 of register set j >> 7) without the row maximum and the
// band logic.  MINUS_INF becomes -12 000 and the scan's identity -24 000: every cell is MINUS_INF or zero plus a sum of scores and
// penalties that pk_global_ok keeps below 4 000 in size, so the three classes never meet and every comparison of the reference -- all
// between sums of the same terms -- comes out as it does in 32 bits.
#define HP_PKG_NEG (-12000)
#define HP_PKG_IDENT (-24000)
HP_INL bool pk_global_ok(const lamsa_hp_para *P, int qlen, int tlen, int o_del, int e_del, int o_ins, int e_ins)
{
    const int mx = P->match > P->mis ? P->match : P->mis;
    const long long o = o_del > o_ins ? o_del : o_ins, e = e_del > e_ins ? e_del : e_ins;
    return mx > 0 && mx < 256 && o_del >= 0 && e_del >= 0 && o_ins >= 0 && e_ins >= 0 && o + e * ((long long)qlen + tlen + 4) + (long long)mx * (qlen + 2) < 4000;
}
template <int NS>
HP_NOINL int ksw_global_pk(Ctx &cx, int qlen, Seq q, int tlen, Seq t, int o_del, int e_del, int o_ins, int e_ins, int w, CigV *out)
{
    long long cells_ = 0;                                                  // in a register: a counter in cx would be a memory round trip per row
    HP_T0(tg0_);
    qlen = wv::uni(qlen); tlen = wv::uni(tlen); w = wv::uni(w);
    o_del = wv::uni(o_del); e_del = wv::uni(e_del); o_ins = wv::uni(o_ins); e_ins = wv::uni(e_ins);
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;                 // :559
    const int zs = 64 * NS;                                                // bytes of a row of the direction matrix
    const size_t mark = arena_mark(cx.tmp);
    const bool zl = (size_t)zs * (size_t)tlen <= z_cap_pk(cx);            // the matrix in LDS when the wave's share holds it
    if (out && zl) ++cx.lds_epoch;
    uint8_t *z = (out && !zl) ? (uint8_t *)arena_alloc(cx, (size_t)zs * tlen + 1) : nullptr;
    if (out && !zl && !z) { arena_release(cx.tmp, mark); return 0; }
    HP_L uint8_t *LZ = (HP_L uint8_t *)cx.lds;
    HP_G uint8_t *gz = (HP_G uint8_t *)wv::uni64((long long)z);
    const int sc_match = wv::uni(cx.P->match), sc_mis = -wv::uni(cx.P->mis);
    const HP_G uint8_t *gq = (const HP_G uint8_t *)wv::uni64((long long)q.p); const int qs = wv::uni(q.stride);
    const HP_G uint8_t *gt = (const HP_G uint8_t *)wv::uni64((long long)t.p); const int ts = wv::uni(t.stride);
    const int OEI = pk::rep(oe_ins), OED = pk::rep(oe_del), EI = pk::rep(e_ins), ED = pk::rep(e_del);
    const int DSC = pk::rep(sc_match - sc_mis), MIS = pk::rep(sc_mis), IDENT = pk::rep(HP_PKG_IDENT), NEG = pk::rep(HP_PKG_NEG);
    wv::Lane<int> Hs[NS], Es[NS], qoh[NS], qN[NS], hcur[NS], jp0, jep0, tl;
    WAVE_FOR(l) { jp0[l] = pk::pack(2 * l, 2 * l + 1); jep0[l] = pk::pack(2 * l * e_ins, (2 * l + 1) * e_ins); tl[l] = 4; }
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        WAVE_FOR(l) {                                                      // first row, :569-572
            int hv[2], oh = 0, nn = 0;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int j = 128 * c + 2 * l + b;
                hv[b] = j == 0 ? 0 : (j <= w ? -(o_ins + e_ins * j) : HP_PKG_NEG);
                const int code = j < qlen ? (int)gq[(long)j * qs] : 4;
                if (code < 4) oh |= 1 << (code + 16 * b); else nn |= (int)(0xffffu << (16 * b));
            }
            Hs[c][l] = pk::pack(hv[0], hv[1]); Es[c][l] = NEG;
            qoh[c][l] = oh; qN[c][l] = nn;
        }
    }
    for (int ib = 0; ib < tlen; ib += 64) {
        { WAVE_FOR(l) { const int ii = ib + l; tl[l] = ii < tlen ? gt[(long)ii * ts] : 4; } }
        const int ti_first = wv::bcast(tl, 0);
        const int ie = ib + 64 < tlen ? ib + 64 : tlen;
        for (int i = ib; i < ie; ++i) {
            const int ti = i == ib ? ti_first : wv::bcast(tl, i & 63);
            const int beg = i > w ? i - w : 0;
            const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
            cells_ += end > beg ? end - beg : 0;
            const int h1_init = beg == 0 ? -(o_del + e_del * (i + 1)) : HP_PKG_NEG;   // :579
            const int tsh = ti & 3, tN = ti > 3 ? -1 : 0;
            const int BEG = pk::rep(beg), END = pk::rep(end), FB = pk::rep(HP_PKG_NEG + beg * e_ins), H1 = pk::rep(h1_init);
            int carry = HP_PKG_IDENT;
            wv::Lane<int> inb[NS];
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                wv::Lane<int> key, klo, M;
                WAVE_FOR(l) {
                    const int jp = pk::add(jp0[l], pk::rep(128 * c));
                    inb[c][l] = pk::neg_mask(pk::sub(jp, END)) & ~pk::neg_mask(pk::sub(jp, BEG));      // beg <= j < end, per half
                    const int eq = (qoh[c][l] >> tsh) & 0x00010001;
                    const int S = pk::add(pk::mul(eq, DSC), MIS) | qN[c][l] | tN;                       // HP_SUB(ti, qb)
                    const int m = pk::add(Hs[c][l], S);
                    const int jep = pk::add(jep0[l], pk::rep(128 * c * e_ins));
                    const int k = pk::sel(inb[c][l], pk::add(pk::sub(m, OEI), jep), IDENT);
                    M[l] = m;
                    klo[l] = pk::lo(k);
                    const int kh = pk::hi(k);
                    key[l] = klo[l] > kh ? klo[l] : kh;
                }
                const int top = wv::scan_max_excl_top(key, HP_PKG_IDENT);
                WAVE_FOR(l) {
                    const int jep = pk::add(jep0[l], pk::rep(128 * c * e_ins));
                    const int s0 = key[l] > carry ? key[l] : carry, s1 = s0 > klo[l] ? s0 : klo[l];
                    const int pre = pk::pack(s0, s1);
                    int f = pk::max(pk::add(pk::sub(pre, jep), EI), pk::sub(FB, jep));                   // F(i,beg) = MINUS_INF carried along the row
                    const int m = M[l];
                    int ee = Es[c][l];
                    int d = pk::neg_mask(pk::sub(m, ee)) & 0x00010001;                                  // ties: M over E   :598-599
                    int h = pk::max(m, ee);
                    d = pk::sel(pk::neg_mask(pk::sub(h, f)), 0x00020002, d);                            //       then over F :600-601
                    h = pk::max(h, f);
                    const int tD = pk::sub(m, OED);
                    ee = pk::sub(ee, ED);
                    d |= pk::neg_mask(pk::sub(tD, ee)) & 0x00040004;                                    // :603-607
                    ee = pk::max(ee, tD);
                    f = pk::sub(f, EI);
                    d |= pk::neg_mask(pk::sub(pk::sub(m, OEI), f)) & 0x00080008;                        // :608-611
                    const int in = inb[c][l];
                    Es[c][l] = pk::sel(in, ee, Es[c][l]);
                    hcur[c][l] = h;
                    if (out && zl) LZ[i * zs + 64 * c + l] = (uint8_t)((d | (d >> 12)) & 0xff);     // (the traceback of a global alignment stays inside the band)
                    else if (out && in) gz[(long)i * zs + 64 * c + l] = (uint8_t)((d | (d >> 12)) & 0xff);
                }
                carry = top > carry ? top : carry;
            }
            // eh[j+1].h = H(i,j) for the row's columns (the band mask moves up with the values), eh[beg].h = H(i,-1), eh[end].e = MINUS_INF
#pragma unroll
            for (int c = NS - 1; c >= 0; --c) {
                const int below = c > 0 ? wv::bcast(hcur[c > 0 ? c - 1 : 0], 63) : 0, below_in = c > 0 ? wv::bcast(inb[c > 0 ? c - 1 : 0], 63) : 0;
                wv::Lane<int> dn = hcur[c], dm = inb[c];
                wv::shr1(dn, below); wv::shr1(dm, below_in);
                WAVE_FOR(l) {
                    const int hsh = pk::shift_up(hcur[c][l], dn[l]), upd = pk::shift_up(inb[c][l], dm[l]);      // beg < j <= end
                    const int in = inb[c][l];
                    int hs = pk::sel(upd, hsh, Hs[c][l]);
                    hs = pk::sel(in & ~upd, H1, hs);                       // j == beg
                    Hs[c][l] = hs;
                    Es[c][l] = pk::sel(upd & ~in, NEG, Es[c][l]);          // j == end, :632
                }
            }
        }
    }
    int score = 0;
#pragma unroll
    for (int c = 0; c < NS; ++c) if ((qlen >> 7) == c) { const int v = wv::bcast(Hs[c], (qlen >> 1) & 63); score = (qlen & 1) ? pk::hi(v) : pk::lo(v); }
    if (score <= HP_PKG_NEG / 2) score = HP_NEG_INF + (score - HP_PKG_NEG);           // (a cell the band never reached, as the reference would return it)
    if (out) {
        const int i = tlen - 1;
        const int k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;           // :638
        wv::sync();
        HP_T0(tb0_);
        dp_backtrack(cx, zl ? LZ : nullptr, z, nullptr, n_col, w, i, k, *out, zs);
        HP_TADD(cx, 28, tb0_);
    }
    cx.n_cells += cells_;                                                  // accounting: DP cell updates (bench.py: GCUPS), once per call
    arena_release(cx.tmp, mark);
    HP_TADD(cx, 24, tg0_);
    return score;
}

// ---- ksw_global2 (src/ksw.c:543-653).  out may be nullptr (score only). ----
HP_INL int ksw_global(Ctx &cx, int qlen, Seq q, int tlen, Seq t,
                      int o_del, int e_del, int o_ins, int e_ins, int w, CigV *out)
{
    if (out) out->n = 0;
    if (qlen < 0 || tlen < 0) { cx.status |= ST_REFEXIT; return 0; }      // reference: exit(-1), :547
    { const int d = iabs(qlen - tlen) + 3; if (w < d) w = d; }             // :549
    HP_DPLOG(1, qlen, tlen, w, 0);
    if (qlen <= HP_REG_QMAX) return ksw_global_reg(cx, qlen, q, tlen, t, o_del, e_del, o_ins, e_ins, w, out);
    if (HP_PK_RT && qlen <= HP_PK_QMAX(2) && pk_global_ok(cx.P, qlen, tlen, o_del, e_del, o_ins, e_ins)) {
        HP_STAT(18);
        return qlen <= HP_PK_QMAX(1) ? ksw_global_pk<1>(cx, qlen, q, tlen, t, o_del, e_del, o_ins, e_ins, w, out) : ksw_global_pk<2>(cx, qlen, q, tlen, t, o_del, e_del, o_ins, e_ins, w, out);
    }
    HP_STAT(19);
    if (2 * w + 4 + 64 <= HP_LDS_CELLS) return ksw_global_lds(cx, qlen, q, tlen, t, o_del, e_del, o_ins, e_ins, w, out);
    return ksw_global_wide(cx, qlen, q, tlen, t, o_del, e_del, o_ins, e_ins, w, out);
}

HP_NOINL ExtRes ksw_extend_lds(Ctx &cx, int qlen, Seq q, int tlen, Seq t, int w, int h0, CigV *out)
{
    ++cx.lds_epoch;                                                         // the rows live in LDS
    long long cells_ = 0;                                                  // in a register: a counter in cx would be a memory round trip per row
    ExtRes er; er.score = 0; er.qle = 0; er.tle = 0;
    HP_T0(te0_);
    // Arguments of a non-inlined device function arrive in vector registers: tell the compiler that they are
    // wave-uniform, so that loop counters, band limits and branches live in scalar registers.
    qlen = wv::uni(qlen); tlen = wv::uni(tlen); w = wv::uni(w); h0 = wv::uni(h0);
    const lamsa_hp_para *P = cx.P;
    const int o_ins = wv::uni(P->ins_ext_o), e_ins = wv::uni(P->ins_ext_e), o_del = wv::uni(P->del_ext_o), e_del = wv::uni(P->del_ext_e);
    const int end_bonus = wv::uni(P->end_bonus), zdrop = wv::uni(P->zdrop);
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    const size_t mark = arena_mark(cx.tmp);
    const bool zl = HP_ZFITS(n_col, tlen);
    uint8_t *z = zl ? nullptr : (uint8_t *)arena_alloc(cx, (size_t)n_col * tlen + 1);
    int32_t *rowb = zl ? nullptr : (int32_t *)arena_alloc(cx, sizeof(int32_t) * 2 * ((size_t)tlen + 1));
    if (!zl && (!z || !rowb)) { arena_release(cx.tmp, mark); return er; }
#ifdef HP_PROF
    if (!zl && cx.prof) { cx.prof[52] += (long long)n_col * tlen; cx.prof[53] += 1; }
#endif
    HP_L int32_t *LH = cx.lds, *LE = cx.lds + HP_LDS_CELLS;
    HP_L uint8_t *LQ = (HP_L uint8_t *)(cx.lds + 2 * HP_LDS_CELLS), *LZ = LQ + HP_LDS_CELLS;
    HP_G uint8_t *gz = (HP_G uint8_t *)wv::uni64((long long)z);
    HP_G int32_t *growb = (HP_G int32_t *)wv::uni64((long long)rowb);
    const int sc_match = wv::uni(P->match), sc_mis = -wv::uni(P->mis);
    const HP_G uint8_t *gq = (const HP_G uint8_t *)wv::uni64((long long)q.p); const int qs = wv::uni(q.stride);
    const HP_G uint8_t *gt = (const HP_G uint8_t *)wv::uni64((long long)t.p); const int ts = wv::uni(t.stride);
    // first row, :692-694: h0, h0-oe_ins, then -e_ins per column while the previous cell is > e_ins
    const int h1v = h0 > oe_ins ? h0 - oe_ins : 0;
#define HP_EH0(j) ((j) == 0 ? h0 : ((j) == 1 ? h1v : (((j) <= qlen && h1v - ((j) - 2) * e_ins > e_ins) ? h1v - ((j) - 1) * e_ins : 0)))
    int hw = -1, qw = -1;
    wv::Lane<int> tl;
    WAVE_FOR(l) { tl[l] = 4; }
    int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1;
    int beg = 0, end = qlen;
    // Rows in blocks of 64: the block's target bases are loaded (and waited for) once per block, so that the row loop
    // itself never waits on the vector-memory counter -- which would also wait for the direction-matrix stores of the
    // previous rows whenever that matrix lives in HBM.
    bool stop_rows = false;
    for (int ib = 0; ib < tlen && !stop_rows; ib += 64) {
    { WAVE_FOR(l) { const int ii = ib + l; tl[l] = ii < tlen ? gt[(long)ii * ts] : 4; } }
    const int ti_first = wv::bcast(tl, 0);                                // consumed here: the wait for the load stays outside the row loop
    const int ie = ib + 64 < tlen ? ib + 64 : tlen;
    for (int i = ib; i < ie; ++i) {
        const int ti = i == ib ? ti_first : wv::bcast(tl, i & 63);
        const int d_beg = i > w ? i - w : 0;
        if (beg < i - w) beg = i - w;                                      // :718-720
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        cells_ += end > beg ? end - beg : 0;
        if (end > hw) {                                                    // columns entering the window
            for (int j0 = hw + 1; j0 <= end; j0 += 64) { WAVE_FOR(l) { const int j = j0 + l; if (j <= end) { LH[j & HP_LDS_MASK] = HP_EH0(j); LE[j & HP_LDS_MASK] = 0; } } }
            hw = end;
        }
        while (end - 1 > qw) { WAVE_FOR(l) { const int j = qw + 1 + l; if (j < qlen) LQ[j & HP_LDS_MASK] = gq[(long)j * qs]; } qw += 64; }
        if (zl) z_row_clear(LZ, i, n_col);                                  // cells the band does not reach stay "never written"
        else { growb[2 * i] = beg; growb[2 * i + 1] = end; }
        wv::sync();
        int h1_init;
        if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
        else h1_init = 0;
        int carryH = wv::uni(LH[beg & HP_LDS_MASK]);
        int Fin = 0;
        long long best = -1;            // (h << 32 | j): row maximum, last j among equals (:743-744)
        int h_last = h1_init;           // H(i,end-1), or the first-column value when the row is empty
        // band shrink bookkeeping (:775-778): nz(j) = eh[j].h != 0 || eh[j].e != 0 after this row
        int first_nz = -1, last_nz = -1, prev_h_nz = h1_init != 0;
        wv::sync();
        if (beg < end) LH[beg & HP_LDS_MASK] = h1_init; else LH[end & HP_LDS_MASK] = h1_init;   // eh[end].h = h1 when the row is empty (:758)
        for (int j0 = beg; j0 < end; j0 += 64) {
            const int nxt = j0 + 64;
            const int carry_next = wv::uni(LH[nxt & HP_LDS_MASK]);
            wv::Lane<int> m, e, key;
            WAVE_FOR(l) {
                const int j = j0 + l;
                if (j < end) {
                    const int hm = l == 0 ? carryH : LH[j & HP_LDS_MASK];
                    const int qb = LQ[j & HP_LDS_MASK];
                    const int M = hm ? hm + HP_SUB(ti, qb) : 0;             // :737
                    int tt = M - oe_ins; tt = tt > 0 ? tt : 0;
                    m[l] = M; e[l] = LE[j & HP_LDS_MASK]; key[l] = tt + j * e_ins;
                } else { m[l] = 0; e[l] = 0; key[l] = HP_SCAN_IDENT; }
            }
            wv::scan_max_excl(key, HP_SCAN_IDENT);
            wv::sync();
            wv::Lane<int> fnext, hnz, enz, hh;
            WAVE_FOR(l) {
                const int j = j0 + l;
                fnext[l] = 0; hnz[l] = 0; enz[l] = 0; hh[l] = -1;
                if (j < end) {
                    int f = Fin - l * e_ins;
                    if (l > 0) { const int g = key[l] - (j - 1) * e_ins; f = g > f ? g : f; }
                    int M = m[l], ee = e[l], h, tt;
                    int dir = M > ee ? 0 : 1; h = M > ee ? M : ee;          // ties: E over M   :738-739
                    dir = h > f ? dir : 2;    h = h > f ? h : f;            //       F over both :740-741
                    tt = M - oe_del; tt = tt > 0 ? tt : 0; ee -= e_del;
                    if (ee > tt) dir |= 1 << 2; else ee = tt;               // :745-750
                    tt = M - oe_ins; tt = tt > 0 ? tt : 0; f -= e_ins;
                    if (f > tt) dir |= 2 << 4; else f = tt;                 // :751-755
                    LE[j & HP_LDS_MASK] = ee;
                    LH[(j + 1) & HP_LDS_MASK] = h;
                    if (zl) z_put(LZ, i, n_col, j - d_beg, dir); else gz[(long)i * n_col + (j - d_beg)] = (uint8_t)dir;
                    fnext[l] = f; hnz[l] = h != 0; enz[l] = ee != 0;
                    hh[l] = h;                                               // scores of this routine are never negative
                }
            }
            Fin = wv::bcast(fnext, 63);
            carryH = carry_next;
            {
                const int hmax = wv::reduce_max(hh);                         // tile maximum; the last lane holding it is the last j
                if (hmax >= 0) {
                    wv::Lane<int> eq;
                    WAVE_FOR(l) eq[l] = hh[l] == hmax;
                    const long long b = ((long long)hmax << 32) | (unsigned)(j0 + 63 - __builtin_clzll(wv::ballot(eq)));
                    if (b > best) best = b;
                }
                const int cnt = end - j0 < 64 ? end - j0 : 64;            // active lanes
                const unsigned long long bh = wv::ballot(hnz), be = wv::ballot(enz);
                // nz for index j0+l: E bit l | H bit (l-1); index j0+cnt (== end on the last chunk) gets H bit cnt-1
                const unsigned long long nzm = be | (bh << 1) | (unsigned long long)(prev_h_nz ? 1 : 0);
                const unsigned long long inrow = cnt == 64 ? ~0ull : ((1ull << cnt) - 1);
                if (first_nz < 0 && (nzm & inrow)) first_nz = j0 + __builtin_ctzll(nzm & inrow);
                if (nzm & inrow) last_nz = j0 + 63 - __builtin_clzll(nzm & inrow);
                prev_h_nz = (int)((bh >> (cnt - 1)) & 1);
            }
        }
        // H(i,end-1): value of the last computed cell (needed for gscore / eh[end].h)
        wv::sync();
        if (beg < end) h_last = wv::uni(LH[end & HP_LDS_MASK]);
        LE[end & HP_LDS_MASK] = 0;                                         // :758
        if (end > hw) hw = end;
        const int jj = beg < end ? end : beg;                              // loop variable j after the row
        if (jj == qlen) {                                                  // :759-762
            max_ie = gscore > h_last ? max_ie : i;
            gscore = gscore > h_last ? gscore : h_last;
        }
        int mrow = 0, mj = -1;
        if (best >= 0) { mrow = (int)(best >> 32); mj = (int)(best & 0xffffffffll); }
        if (mrow == 0) { stop_rows = true; break; }                                              // :763
        if (mrow > max) { max = mrow; max_i = i; max_j = mj; }
        else if (zdrop > 0) {                                              // :767-773
            if (i - max_i > mj - max_j) { if (max - mrow - ((i - max_i) - (mj - max_j)) * e_del > zdrop) { stop_rows = true; break; } }
            else { if (max - mrow - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) { stop_rows = true; break; } }
        }
        // shrink the band for the next row, :775-778.  index `end` itself: eh[end].h = h_last, eh[end].e = 0
        {
            const int nb = first_nz >= 0 ? first_nz : end;                 // first j in [beg,end) that is non-zero
            int jl;                                                        // last j in [nb,end] that is non-zero, else nb-1
            if (h_last != 0 && end >= nb) jl = end;
            else if (last_nz >= nb && last_nz >= 0) jl = last_nz;
            else jl = nb - 1;
            beg = nb;
            end = jl + 2 < qlen ? jl + 2 : qlen;
        }
    }
    }
#undef HP_EH0
    int i, k;
    if (gscore <= 0 || gscore <= max - end_bonus) { i = max_i; k = max_j; }   // :785-789
    else { i = max_ie; k = qlen - 1; }
    er.qle = k + 1; er.tle = i + 1; er.score = max;
    if (out) { wv::sync(); HP_T0(tb0_); dp_backtrack(cx, zl ? LZ : nullptr, z, rowb, n_col, w, i, k, *out); HP_TADD(cx, 28, tb0_); }
    cx.n_cells += cells_;                                                  // accounting: DP cell updates (bench.py: GCUPS), once per call
    arena_release(cx.tmp, mark);
    HP_TADD(cx, 26, te0_);
    return er;
}

// ---- ksw_extend_core (src/ksw.c:667-807).  Uses the *extension* gap penalties. ----
HP_INL int ksw_extend(Ctx &cx, int qlen, Seq q, int tlen, Seq t, int w, int h0, int *qle, int *tle, CigV *out)
{
    if (out) out->n = 0;
    if (qle) *qle = 0;
    if (tle) *tle = 0;
    if (qlen < 0 || tlen < 0 || h0 <= 0) { cx.status |= ST_REFEXIT; return 0; }   // :672, assert :682
    {   // adjust w, :696-704 (double arithmetic, truncation toward zero as in the reference)
        const lamsa_hp_para *P = cx.P;
        int mx = P->match > 0 ? P->match : 0;               // max over the 5x5 matrix (N cells are -1)
        if (-P->mis > mx) mx = -P->mis;
        int max_ins = (int)((double)(qlen * mx + P->end_bonus - P->ins_ext_o) / P->ins_ext_e + 1.);
        max_ins = max_ins > 1 ? max_ins : 1;
        w = w < max_ins ? w : max_ins;
        int max_del = (int)((double)(qlen * mx + P->end_bonus - P->del_ext_o) / P->del_ext_e + 1.);
        max_del = max_del > 1 ? max_del : 1;
        w = w < max_del ? w : max_del;
    }
    const long long cells0_ = cx.n_cells; (void)cells0_;
    ExtRes er;
    int cls_ = 0; (void)cls_;          // diagnostic: which routine ran
    HP_T0(tcls_);
    if (qlen <= HP_REG_QMAX) er = ksw_extend_reg(cx, qlen, q, tlen, t, w, h0, out);
    else if (HP_PK_RT && pkb_extend_ok(cx.P, qlen, h0, pkb_sets_q(qlen, w))) {
        // every longer query: the window routine.  (Round 3's lane-per-column-pair routine for 63 .. 254 bases, ksw_extend_pk, cost more per row
        // for half the columns -- 5.8 k wave cycles against 4.8 k -- and is gone: profiles/r04_dp_routines.txt.)
        HP_STAT(22);
        const int ns = pkb_sets_q(qlen, w);
        cls_ = ns == 1 ? 3 : (ns == 2 ? 4 : 5);
        er = ns == 1 ? ksw_extend_band<1>(cx, qlen, q, tlen, t, w, h0, out) : (ns == 2 ? ksw_extend_band<2>(cx, qlen, q, tlen, t, w, h0, out) : ksw_extend_band<4>(cx, qlen, q, tlen, t, w, h0, out));
    }
    else if (qlen <= HP_REGN_QMAX(2)) { HP_STAT(17); cls_ = 6; er = ksw_extend_regn<2>(cx, qlen, q, tlen, t, w, h0, out); }
    else if (qlen <= HP_REGN_QMAX(3) && HP_REGN_SETS >= 3) { HP_STAT(17); cls_ = 6; er = ksw_extend_regn<3>(cx, qlen, q, tlen, t, w, h0, out); }
    else if (qlen <= HP_REGN_QMAX(4) && HP_REGN_SETS >= 4) { HP_STAT(17); cls_ = 6; er = ksw_extend_regn<4>(cx, qlen, q, tlen, t, w, h0, out); }
    else { cls_ = 7; er = 2 * w + 4 + 64 <= HP_LDS_CELLS ? ksw_extend_lds(cx, qlen, q, tlen, t, w, h0, out) : ksw_extend_wide(cx, qlen, q, tlen, t, w, h0, out); }
#ifdef HP_PROF
    if (cx.prof_dp) { cx.prof_dp[4 * cls_] += wv::clock() - tcls_; cx.prof_dp[4 * cls_ + 1] += cx.n_cells - cells0_; cx.prof_dp[4 * cls_ + 2] += 1; cx.prof_dp[4 * cls_ + 3] += er.tle; }
#endif
    HP_DPLOG(0, qlen, tlen, w, cx.n_cells - cells0_);
    if (qle) *qle = er.qle;
    if (tle) *tle = er.tle;
    return er.score;
}

// ksw_extend_c (src/ksw.c:809): 0 query-to-end, 1 target-to-end, 2 neither
HP_INL int ksw_extend_c(Ctx &cx, int qlen, Seq q, int tlen, Seq t, int w, int h0, int *qle, int *tle, CigV *out)
{
    ksw_extend(cx, qlen, q, tlen, t, w, h0, qle, tle, out);
    if (*qle == qlen) return 0;
    if (*tle == tlen) return 1;
    return 2;
}
// ksw_extend_r (src/ksw.c:820): same on the reversed sequences (views, nothing is copied)
HP_INL int ksw_extend_r(Ctx &cx, int qlen, Seq q, int tlen, Seq t, int w, int h0, int *qre, int *tre, CigV *out)
{
    Seq rq, rt;
    rq.p = q.p + (long)(qlen > 0 ? qlen - 1 : 0) * q.stride; rq.stride = -q.stride;
    rt.p = t.p + (long)(tlen > 0 ? tlen - 1 : 0) * t.stride; rt.stride = -t.stride;
    ksw_extend(cx, qlen, rq, tlen, rt, w, h0, qre, tre, out);
    if (*qre == qlen) return 0;
    if (*tre == tlen) return 1;
    return 2;
}

// sw_mid_fix (src/ksw.c:841-860): appends to out
HP_INL void sw_mid_fix(Ctx &cx, CigV &out, const cig_t *lc, int ln, const cig_t *rc, int rn,
                      int qlen, Seq q, int lqe, int rqe, int tlen, Seq t, int lte, int rte)
{
    const lamsa_hp_para *P = cx.P;
    int Sn = qlen - lqe - rqe, Hn = tlen - lte - rte, half = P->split_len / 2;
    if (iabs(Sn) >= half || iabs(Hn) >= half || iabs(Sn - Hn) >= half || tlen < 0 || qlen < 0) {
        cig_pushv(cx, out, lc, ln);
        cig_push0(cx, out, (cig_t)((uint32_t)Sn << 4) | C_S);          // the counts can be negative here, as in the reference (src/ksw.c:855): same bits, no UB
        cig_push0(cx, out, (cig_t)((uint32_t)Hn << 4) | C_H);
        cig_pushv(cx, out, rc, rn);
    } else {
        const size_t mark = arena_mark(cx.tmp);
        CigV g;
        if (cig_alloc(cx, g, qlen + tlen + 2)) {
            ksw_global(cx, qlen, q, tlen, t, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, P->band_w, &g);
            cig_pushv(cx, out, g.c, g.n);
        }
        arena_release(cx.tmp, mark);
    }
}

// the single-precision test of src/ksw.c:881,900 -- evaluated without contraction
HP_INL bool bi_near_diag(const lamsa_hp_para *P, int qlen, int tlen)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float a = __fmul_rn((float)tlen, P->id_rate);
    float b = __fmul_rn(a, (float)(P->aln_mode & 2));
    float c = __fadd_rn((float)P->split_len, b);
#else
    volatile float a = (float)tlen * P->id_rate;
    volatile float b = a * (float)(P->aln_mode & 2);
    volatile float c = (float)P->split_len + b;
#endif
    return (float)iabs(qlen - tlen) < c;
}

// ksw_bi_extend (src/ksw.c:862-926): result replaces out; returns the "gap exists" flag
HP_NOINL int ksw_bi_extend(Ctx &cx_, int qlen, Seq q, int tlen, Seq t, int lh0, int rh0, CigV &out_)
{
    // Arguments of a non-inlined device function arrive in vector registers; everything here is wave-uniform, and what lives across the
    // calls below should sit in scalar registers (the DP routines use up to 55 of the 64 vector registers the fill kernel has per lane:
    // what the caller keeps in vector registers goes to scratch and back around every call).
    Ctx &cx = *(Ctx *)wv::uni64((long long)&cx_);
    CigV &out = *(CigV *)wv::uni64((long long)&out_);
    qlen = wv::uni(qlen); tlen = wv::uni(tlen); lh0 = wv::uni(lh0); rh0 = wv::uni(rh0);
    q.p = (const uint8_t *)wv::uni64((long long)q.p); q.stride = wv::uni(q.stride);
    t.p = (const uint8_t *)wv::uni64((long long)t.p); t.stride = wv::uni(t.stride);
    const lamsa_hp_para *P = (const lamsa_hp_para *)wv::uni64((long long)cx.P);
    out.n = 0;
    if (qlen < 0 || tlen < 0) { cx.status |= ST_REFEXIT; return 0; }
    if (qlen == 0 && lh0 > 0) {
        // Nothing to align (adjacent seeds that overlap, the most frequent call): the left extension ends at the query
        // end immediately, whatever row it stops in (:785-789 gives tle 0 or 1 and a "1D" or empty traceback), and :875-880
        // completes it to a deletion of the whole target.
        if (tlen > 0) cig_raw_push(cx, out, (tlen << 4) | C_D);
        return 0;
    }
    HP_T0(tbi0_);
    const size_t mark = arena_mark(cx.tmp);
    int ret = 0, res, lqe, lte, rqe, rte;
    CigV L, R;
    if (!cig_alloc(cx, L, qlen + tlen + 4) || !cig_alloc(cx, R, qlen + tlen + 4)) { arena_release(cx.tmp, mark); return 0; }
    L.c = (cig_t *)wv::uni64((long long)L.c); R.c = (cig_t *)wv::uni64((long long)R.c); L.cap = wv::uni(L.cap); R.cap = wv::uni(R.cap);
    const int band_w = wv::uni(P->band_w);
    const int w = iabs(qlen - tlen) + 3 > band_w ? iabs(qlen - tlen) + 3 : band_w;         // :873
    res = ksw_extend_c(cx, qlen, q, tlen, t, w, lh0, &lqe, &lte, &L);
    res = wv::uni(res); lqe = wv::uni(lqe); lte = wv::uni(lte);
    HP_T0(tbi1_);
    if (res < 2) {                                                                          // :875-880
        cig_pushv(cx, out, L.c, L.n);
        cig_push1(cx, out, res == 0 ? ((tlen - lte) << 4) | C_D : ((qlen - lqe) << 4) | C_I);
    } else if (bi_near_diag(P, qlen, tlen) && ((lqe << 1 > qlen) || (lte << 1 > tlen))) {   // :881-887
        ksw_global(cx, qlen, q, tlen, t, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, P->band_w, &out);
    } else {
        res = ksw_extend_r(cx, qlen, q, tlen, t, w, rh0, &rqe, &rte, &R);
        res = wv::uni(res); rqe = wv::uni(rqe); rte = wv::uni(rte);
        if (res < 2) {                                                                      // :892-899
            cig_push1(cx, R, res == 0 ? ((tlen - rte) << 4) | C_D : ((qlen - rqe) << 4) | C_I);
            cig_invert(R.c, R.n);
            cig_pushv(cx, out, R.c, R.n);
        } else if (bi_near_diag(P, qlen, tlen) && ((rqe << 1 > qlen) || (rte << 1 > tlen))) { // :900-906
            ksw_global(cx, qlen, q, tlen, t, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, P->band_w, &out);
        } else {
            cig_invert(R.c, R.n);
            sw_mid_fix(cx, out, L.c, L.n, R.c, R.n, qlen, q, lqe, rqe, tlen, t, lte, rte);
            ret = (qlen - lqe - rqe) >= P->split_len ? 1 : 0;                                // :924
        }
    }
    arena_release(cx.tmp, mark);
    HP_TADD(cx, 50, tbi1_);
    HP_TADD(cx, 48, tbi0_);
    return ret;
}

}  // namespace hp
