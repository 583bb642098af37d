// hp_fill.h -- lines of fragments -> base-level alignment records on one wavefront
// (reference src/frag_check.c, src/bntseq.c:465).
//
//   ref_fetch      <- pac2fa_core      bntseq.c:465  2-bit window unpacked by the 64 lanes
//   merge_cigar    <- merge_cigar      frag_check.c:251
//   frag_extend    <- frag_extend      :332
//   split_mapping  <- split_mapping    :416
//   head_fix/tail_fix <- frag_head_bound_fix / frag_tail_bound_fix  :576,:656
//   res_split      <- lamsa_res_split  :712
//   res_aux        <- lamsa_res_aux    :793  mismatch counting spread over the lanes
//   fill_lines     <- frag_check       :856
// Control flow is wave-uniform; sequences are never copied or reversed, DP routines read
// them through strided views.  Where the reference exit(1)s the read is flagged
// ST_REFEXIT and abandoned.
#pragma once
#include "hp_split.h"

namespace hp {

#define HP_REC_MAX 64            // records per line (split alignments); more flags ST_OVERFLOW

struct Rec {                     // res_t, frag_check.h:46-59
    int64_t offset, refend; int32_t chr, nstrand, readend, score, NM; CigV cig;
};
struct LineRes {                 // line_aln_res, frag_check.h:61-73
    int line_score, tol_score, tol_NM, cur_res_n;
    Rec rec[HP_REC_MAX];
};

// ---------------------------------------------------------------- pac2fa_core, bntseq.c:465-477
HP_FN bool ref_fetch(ReadCtx &r, int chr, int64_t start0, int32_t *len, uint8_t *dst)
{
    HP_T0(tr0_);
    const int32_t clen = r.ref.seq_len[chr - 1];
    if (start0 > clen || start0 < 0) { r.cx.status |= ST_REFEXIT; return false; }      // exit(1), :469-472
    if (start0 + *len > clen) *len = (int32_t)(clen - start0);                          // :474
    const int64_t k0 = r.ref.seq_off[chr - 1] + start0;
    const int32_t n = *len;
    r.t_bases += n > 0 ? n : 0;
    const uint8_t *pac = r.ref.pac;
    if (n >= 256 && ((uintptr_t)dst & 3) == 0) {
        // a long window (the NM / AS pass over a whole record, an end extension): four bases per lane -- the two packed bytes they lie in, one
        // 32-bit store -- and two such passes requested together; a base per lane was 157 dependent load-store rounds for a 10-kbp record
        const HP_G uint8_t *gp = (const HP_G uint8_t *)pac;
        const int sh0 = (int)(k0 & 3) << 1;
        int32_t b = 0;
        for (; b + 512 <= n; b += 512) {
            wv::Lane<int> w0, w1;
            WAVE_FOR(l) {
                const int64_t ka = (k0 + b + 4 * l) >> 2, kb = (k0 + b + 256 + 4 * l) >> 2;
                w0[l] = ((int)gp[ka] << 8) | (sh0 ? (int)gp[ka + 1] : 0);
                w1[l] = ((int)gp[kb] << 8) | (sh0 ? (int)gp[kb + 1] : 0);
            }
            WAVE_FOR(l) {
                const int x = w0[l] >> (8 - sh0), y = w1[l] >> (8 - sh0);        // the four bases, first one in bits 7..6
                *(HP_G uint32_t *)((HP_G uint8_t *)dst + b + 4 * l) = (uint32_t)((x >> 6 & 3) | (x >> 4 & 3) << 8 | (x >> 2 & 3) << 16 | (x & 3) << 24);
                *(HP_G uint32_t *)((HP_G uint8_t *)dst + b + 256 + 4 * l) = (uint32_t)((y >> 6 & 3) | (y >> 4 & 3) << 8 | (y >> 2 & 3) << 16 | (y & 3) << 24);
            }
        }
        for (; b < n; b += 64) {
            WAVE_FOR(l) {
                const int32_t i = b + l;
                if (i < n) { const int64_t k = k0 + i; dst[i] = pac[k >> 2] >> ((~k & 3) << 1) & 3; }
            }
        }
    } else
    for (int32_t b = 0; b < n; b += 64) {
        WAVE_FOR(l) {
            const int32_t i = b + l;
            if (i < n) { const int64_t k = k0 + i; dst[i] = pac[k >> 2] >> ((~k & 3) << 1) & 3; }   // _get_pac, :242
        }
    }
    wv::sync();
    HP_TADD(r.cx, 30, tr0_);
    return true;
}

// ---------------------------------------------------------------- merge_cigar, frag_check.c:251-328
// _push_cigar (frag_check.h:158-184) with what the caller already holds in registers: `last` = the vector's last element (used when it
// is not empty), `c0` = the first source element (which may differ from c[0] in memory: the repair below shortens it).  Returns the
// vector's new last element.  Nothing is loaded but the body of the copy.
// dst / vn / cap: the vector's buffer, length (updated) and capacity as the caller holds them -- through the vector they would be loads from memory
// that wait behind every store.
HP_FN int cig_pushv_known(Ctx &cx, HP_G cig_t *dst, int &vn_io, int cap, int last, const cig_t *c, int n, int c0)
{
    if (n == 0) return last;
    const HP_G cig_t *src = (const HP_G cig_t *)c;
    const int vn = vn_io;
    int j = 0;
    if (vn > 0) {
        if ((last & 0xf) == (c0 & 0xf)) { last = last + ((c0 >> 4) << 4); dst[vn - 1] = last; j = 1; }
        else if (((last & 0xf) == C_I && (c0 & 0xf) == C_S) || ((last & 0xf) == C_S && (c0 & 0xf) == C_I)) { last = (((last >> 4) + (c0 >> 4)) << 4) | C_S; dst[vn - 1] = last; j = 1; }
    }
    const int m = n - j;
    if (vn + m > cap) { cx.status |= ST_OVERFLOW; return last; }
    if (m > 0) {
        wv::Lane<int> w;
        WAVE_FOR(l) { w[l] = 0; }
        for (int b0 = 0; b0 < m; b0 += 64) { WAVE_FOR(l) { const int i = b0 + l; if (i < m) { w[l] = j + i == 0 ? c0 : (int)src[j + i]; dst[vn + i] = w[l]; } } }
        last = wv::bcast(w, (m - 1) & 63);
    }
    vn_io = vn + m;
    wv::sync();
    return last;
}

// The boundary repair (:264-322): both CIGARs are shortened from the junction by whole elements, up to five matched bases into the
// first long match on either side, the stretch between is aligned again, and if that alignment still begins or ends with a gap the
// flanks grow once more.  The elements it walks over -- the tail of c1, the head of c2 -- are loaded once, 64 of each, one per lane,
// and the walk reads lanes; the element a side stops in is shortened in a register and stored when the walk is over.  (Walked one at
// a time out of HBM this routine was a quarter of the fill kernel: ~16 calls per line, each a chain of ~40 dependent round trips.)
HP_NOINL bool merge_cigar_full(ReadCtx &r, CigV &c1, int64_t *c1_refend, int *c1_readend, int chr,
                               const cig_t *_c2, int c2_n, int c2_reflen, int c2_readlen)
{
    if (c2_n == 0) return true;
    Ctx &cx = r.cx;
    HP_T0(tmf_);
    const lamsa_hp_para *P = cx.P;
    // what is needed of the vector and of the record's ends, read once (they lie in memory: every later use would be a load behind the stores)
    HP_G cig_t *const c1c = (HP_G cig_t *)wv::uni64((long long)c1.c);
    const int n1_0 = wv::uni(c1.n), c1cap = wv::uni(c1.cap);
    const int64_t refend0 = *c1_refend; const int readend0 = wv::uni(*c1_readend), read_L = wv::uni(r.L);
    const uint8_t *const cur_read = (const uint8_t *)wv::uni64((long long)r.cur_read);
    const HP_G cig_t *g1 = c1c, *g2 = (const HP_G cig_t *)_c2;
    int vn = n1_0;
    wv::Lane<int> T1, T2;                                     // T1[l] = c1[n1_0 - 1 - l] (the top first), T2[l] = c2[l]
    WAVE_FOR(l) { T1[l] = l < n1_0 ? (int)g1[n1_0 - 1 - l] : 0; T2[l] = l < c2_n ? (int)g2[l] : 0; }
    bool repair = false;
    if (n1_0 > 1) {
        const int t = wv::bcast(T1, 0), h = wv::bcast(T2, 0);
        const int top = t & 0xf, hop = h & 0xf;
        if ((((top == C_I || top == C_D) && (t >> 4) <= 3) && hop != C_S && hop != C_H) ||
            (((hop == C_I || hop == C_D) && (h >> 4) <= 3) && top != C_S && top != C_H)) repair = true;
    }
    if (!repair) cig_pushv_known(cx, c1c, vn, c1cap, wv::bcast(T1, 0), _c2, c2_n, wv::bcast(T2, 0));
    else {
        const size_t mark = arena_mark(cx.tmp);
        int len1, len11 = 0, len2, len21 = 0, len22 = 0, len_dif1 = 0, len_dif2 = 0;
        int b = 0, min_b, ci = 0, left = 1, right = 1;
        const int md = 5;
        int64_t ref_start = 0; int read_start = 0;
        CigV bd; bd.c = nullptr; bd.n = 0; bd.cap = 0;
        int n1 = n1_0;                                        // c1.n while the walk goes on
        // the element on top of c1 / at ci of c2 as the walk has left it (a long match is shortened where the walk stops)
#define HP_MF_E1(k) ((k) <= 0 ? 0 : (n1_0 - (k) < 64 ? wv::bcast(T1, n1_0 - (k)) : (int)g1[(k) - 1]))          /* element k-1 of c1, untouched */
#define HP_MF_E2(k) ((k) >= c2_n ? 0 : ((k) < 64 ? wv::bcast(T2, (k)) : (int)g2[(k)]))                       /* element k of c2, untouched */
        int e1 = HP_MF_E1(n1), e2 = HP_MF_E2(0);
        int bd_first = 0, bd_last = 0;
        bool ok = true;
        for (;;) {
            if (left) {
                while (n1 >= 1) {
                    const int op = e1 & 0xf, l = e1 >> 4;
                    if (op == C_M && l > md) { e1 -= md << 4; len21 += md; break; }
                    else if (op == C_M) { len21 += l; --n1; e1 = HP_MF_E1(n1); }
                    else if (op == C_I) { len21 += l; len_dif1 -= l; b += l; --n1; e1 = HP_MF_E1(n1); }
                    else if (op == C_D) { len_dif1 += l; b += l; --n1; e1 = HP_MF_E1(n1); }
                    else { left = -1; break; }
                }
                len11 = len21 + len_dif1;
                read_start = readend0 - len21 + 1;
                ref_start = refend0 - len11 + 1;
            }
            if (right) {
                while (ci < c2_n) {
                    const int op = e2 & 0xf, l = e2 >> 4;
                    if (op == C_M && l > md) { e2 -= md << 4; len22 += md; break; }
                    else if (op == C_M) { len22 += l; ci++; e2 = HP_MF_E2(ci); }
                    else if (op == C_I) { len22 += l; len_dif2 -= l; b += l; ++ci; e2 = HP_MF_E2(ci); }
                    else if (op == C_D) { len_dif2 += l; b += l; ++ci; e2 = HP_MF_E2(ci); }
                    else { right = -1; break; }
                }
            }
            len2 = len21 + len22; len1 = len2 + len_dif1 + len_dif2;
            min_b = iabs(len_dif1 + len_dif2) + md; b = b > min_b ? b : min_b;
            const size_t m2 = arena_mark(cx.tmp);
            uint8_t *seq1 = (uint8_t *)arena_alloc(cx, (size_t)(len1 > 0 ? len1 : 0) + 16);
            int32_t l1 = len1;
            if (!seq1 || len2 < 0 || read_start < 1 || read_start - 1 + len2 > read_L || !ref_fetch(r, chr, ref_start - 1, &l1, seq1)) { cx.status |= seq1 ? ST_REFEXIT : ST_OVERFLOW; ok = false; break; }
            len1 = l1;
            if (!cig_alloc(cx, bd, len1 + len2 + 8)) { ok = false; break; }
            ksw_global(cx, len2, seq_fwd(cur_read + read_start - 1), len1, seq_fwd(seq1), P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, b, &bd);
            if (bd.n == 0) { cx.status |= ST_REFEXIT; ok = false; break; }    // the reference dereferences an empty CIGAR here
            {   // its first and last element, in one trip
                wv::Lane<int> fl;
                WAVE_FOR(l) { fl[l] = l < 2 ? (int)((const HP_G cig_t *)bd.c)[l ? bd.n - 1 : 0] : 0; }
                bd_first = wv::bcast(fl, 0); bd_last = wv::bcast(fl, 1);
            }
            bool stop = false;
            if ((bd_first & 0xf) == C_M) left = 0;
            else if (n1 == 0 || left < 0) stop = true;
            if (!stop) {
                if ((bd_last & 0xf) == C_M) right = 0;
                else if (ci == c2_n || right < 0) stop = true;
            }
            if (stop || left + right == 0) break;
            arena_release(cx.tmp, m2);           // drop this round's window + CIGAR, try a longer flank
        }
#undef HP_MF_E1
#undef HP_MF_E2
        if (ok) {
            vn = n1;
            if (n1 > 0) c1c[n1 - 1] = e1;                                       // the top of c1 as the walk left it (the push may merge into it and store it again)
            const int last = cig_pushv_known(cx, c1c, vn, c1cap, e1, bd.c, bd.n, bd_first);
            cig_pushv_known(cx, c1c, vn, c1cap, last, _c2 + ci, c2_n - ci, e2);
        }
        arena_release(cx.tmp, mark);
        if (!ok) return false;
    }
    c1.n = vn;
    *c1_refend = refend0 + c2_reflen;
    *c1_readend = readend0 + c2_readlen;
    HP_TADD_FILL(cx, 16, tmf_);
    return true;
}

// The junction test of merge_cigar (:256-263) decides between a plain append -- by far the common case, done here
// without a call -- and the boundary repair, which is the only part worth a call frame.  _c2 must lie in HBM.
HP_INL bool merge_cigar(ReadCtx &r, CigV &c1, int64_t *c1_refend, int *c1_readend, int chr,
                        const cig_t *_c2, int c2_n, int c2_reflen, int c2_readlen)
{
    if (c2_n == 0) return true;
    const int n1 = c1.n;
    if (n1 > 1) {
        const cig_t t = ((const HP_G cig_t *)c1.c)[n1 - 1], h = ((const HP_G cig_t *)_c2)[0];
        const int top = t & 0xf, hop = h & 0xf;
        if ((((top == C_I || top == C_D) && (t >> 4) <= 3) && hop != C_S && hop != C_H) ||
            (((hop == C_I || hop == C_D) && (h >> 4) <= 3) && top != C_S && top != C_H))
            return merge_cigar_full(r, c1, c1_refend, c1_readend, chr, _c2, c2_n, c2_reflen, c2_readlen);
    }
    cig_pushv(r.cx, c1, _c2, c2_n);
    *c1_refend += c2_reflen;
    *c1_readend += c2_readlen;
    return true;
}

// read interval between two chained seeds (get_read_intv, :116): pointer into the strand-appropriate read
HP_INL int read_gap(const ReadCtx &r, int s1, int s2, const uint8_t **p)
{
    const lamsa_hp_para *P = r.cx.P;
    int i, e;
    if (r.h_strand[s1] == 1) { i = sid(r, r.n_seed[s1]) * P->seed_step - P->seed_inv; e = (sid(r, r.n_seed[s2]) - 1) * P->seed_step; }
    else { i = r.last_len + sid(r, r.n_seed[s1]) * P->seed_step - P->seed_inv; e = r.last_len + (sid(r, r.n_seed[s2]) - 1) * P->seed_step; }
    *p = r.cur_read + i;
    return e > i ? e - i : 0;
}

// ---------------------------------------------------------------- frag_extend, :332-410
#ifndef HP_FRAG_BLOCK_MIN
#define HP_FRAG_BLOCK_MIN 3                  // fewer steps than this are walked one by one (the tests' CPU build sets it to 1 and to a large number)
#endif
HP_INL int seed_at_after_block(const HP_G int32_t *g_seed, int i, int step, int nb) { return g_seed[i + step * (nb - 1)]; }

// One seed step of the loop :360-400, as the reference runs it: the gap between seed `last` and seed s (its CIGAR computed ahead by the lane-DP
// launch, or ksw_global2 here), then seed s's own CIGAR, each through merge_cigar.
HP_INL bool frag_step(ReadCtx &r, const FLines &F, int frag, int i, int chr, CigV &fc, int64_t &ref_end, int &re, int &last)
{
    Ctx &cx = r.cx;
    const lamsa_hp_para *P = cx.P;
    const int32_t *seed = F.fr_seed + F.fr_seed_off[frag];
    const int s = seed[i];
    {   const int32_t *gt = F.gt ? F.gt + 4 * (F.fr_seed_off[frag] + i) : nullptr;
        if (gt && gt[3] && F.jarena) {                                          // the gap's CIGAR was computed ahead (hp_lanedp.h)
            const uint8_t *qp0; const int len1p = read_gap(r, last, s, &qp0);
            const bool ok = merge_cigar(r, fc, &ref_end, &re, chr, F.jarena + gt[0], gt[1], gt[2], len1p) &&
                            (r.cs_words += r.h_cig_n[s], merge_cigar(r, fc, &ref_end, &re, chr, r.cig + r.h_cig_off[s], r.h_cig_n[s], P->seed_len + r.h_len_dif[s], P->seed_len));
            last = s;
            return ok;
        }
    }
    const size_t m2 = arena_mark(cx.tmp);
    // get_ref_intv, :98
    const int64_t start = r.h_pos[last] + P->seed_len - 1 + r.h_len_dif[last];
    int32_t len2 = (int32_t)(r.h_pos[s] - 1 - start);
    uint8_t *tb = nullptr;
    if (len2 <= 0) len2 = 0;
    else {
        tb = (uint8_t *)arena_alloc(cx, (size_t)len2 + 16);
        if (!tb || !ref_fetch(r, r.h_chr[last], start, &len2, tb)) return false;
    }
    const uint8_t *qp; const int len1 = read_gap(r, last, s, &qp);
    CigV g;
    if (!cig_alloc(cx, g, len1 + len2 + 8)) return false;
    ksw_global(cx, len1, seq_fwd(qp), len2, seq_fwd(tb ? tb : qp), P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, P->band_w, &g);
    const bool ok = merge_cigar(r, fc, &ref_end, &re, chr, g.c, g.n, len2, len1) &&
                    (r.cs_words += r.h_cig_n[s], merge_cigar(r, fc, &ref_end, &re, chr, r.cig + r.h_cig_off[s], r.h_cig_n[s], P->seed_len + r.h_len_dif[s], P->seed_len));
    last = s;
    arena_release(cx.tmp, m2);
    return ok;
}

// Up to 32 seed steps of that loop at once, a PIECE (a gap's CIGAR, a seed's CIGAR) per lane, when every piece is there, is not empty and both
// begins and ends with a match, and so does what has been merged so far: merge_cigar then never repairs a boundary (:256-263 look for a short
// indel on either side of it), every piece's first element is added to the element on top (_push_cigar) and the rest is appended as it is.
// That is a concatenation: piece k's elements go to where the pieces before it end (a prefix sum of n - 1), the lengths of the first elements
// are summed over each run of pieces up to the next one that has more than one element, and the sum goes into the element on top of that run.
// A fragment of an error-poor read is ~50 seeds of "50M" with "50M" between them -- walked one after the other that was ~8 dependent round
// trips per seed and most of the fill kernel's time on the 1 %-error workloads (profiles/r04_fill_glue.txt).  Returns the steps taken (0: not
// this way); `top` = the element on top of fc, kept by the caller.
HP_INL int frag_steps_block(ReadCtx &r, const FLines &F, int frag, int i, int step, int n_steps, CigV &fc, int64_t &ref_end, int &re, int &last, cig_t &top)
{
    Ctx &cx = r.cx;
    const lamsa_hp_para *P = cx.P;
    if (!F.gt || !F.jarena || fc.n < 1 || (top & 0xf) != C_M) return 0;
    const int fro = F.fr_seed_off[frag];
    const HP_G int32_t *g_seed = (const HP_G int32_t *)(F.fr_seed + fro);
    const HP_G int32_t *g_gt = (const HP_G int32_t *)(F.gt + 4 * (size_t)fro);
    const HP_G cig_t *g_ja = (const HP_G cig_t *)F.jarena, *g_cig = (const HP_G cig_t *)r.cig;
    const int nb = n_steps < 32 ? n_steps : 32, K = 2 * nb;
    const int seed_len = P->seed_len, seed_step = P->seed_step, seed_inv = P->seed_inv;
    const int plus = r.h_strand[last] == 1, base_off = plus ? 0 : r.last_len;
    wv::Lane<int> n, wf, wl, rl, ql, cw; wv::Lane<long long> src;
    WAVE_FOR(l) {
        n[l] = 0; wf[l] = 0; wl[l] = 0; rl[l] = 0; ql[l] = 0; cw[l] = 0; src[l] = 0;
        if (l < K) {
            const int j = l >> 1, idx = i + step * j;
            const int s = g_seed[idx], prev = j == 0 ? last : g_seed[idx - step];
            if (!(l & 1)) {                                                     // the gap in front of seed s
                int g[4]; hp_load16(g_gt + 4 * (size_t)idx, g);
                if (g[3]) {
                    const int a = base_off + sid(r, r.n_seed[prev]) * seed_step - seed_inv, e = base_off + (sid(r, r.n_seed[s]) - 1) * seed_step;      // read_gap
                    n[l] = g[1]; src[l] = (long long)(g_ja + g[0]); rl[l] = g[2]; ql[l] = e > a ? e - a : 0;
                } else n[l] = -1;                                               // not computed ahead
            } else {                                                            // seed s itself
                n[l] = r.h_cig_n[s]; src[l] = (long long)(g_cig + r.h_cig_off[s]); rl[l] = seed_len + r.h_len_dif[s]; ql[l] = seed_len; cw[l] = n[l];
            }
            if (n[l] > 0) { const HP_G cig_t *w = (const HP_G cig_t *)src[l]; wf[l] = w[0]; wl[l] = w[n[l] - 1]; }
        }
    }
    {
        wv::Lane<int> bad;
        WAVE_FOR(l) { bad[l] = l < K && (n[l] <= 0 || (wf[l] & 0xf) != C_M || (wl[l] & 0xf) != C_M); }
        if (wv::ballot(bad)) return 0;
    }
    wv::Lane<int> adv, S, big;
    WAVE_FOR(l) { adv[l] = l < K ? n[l] - 1 : 0; S[l] = l < K ? (wf[l] >> 4) : 0; big[l] = l < K && n[l] >= 2; }
    const int n_new = wv::reduce_sum(adv);
    if (fc.n + n_new > fc.cap) return 0;                                        // (the walk one by one flags it)
    wv::Lane<int> base = adv;
    wv::scan_add_excl(base);                                                    // elements appended by the pieces before this one
    { wv::Lane<int> h = S; wv::scan_add_excl(h); WAVE_FOR(l) { S[l] += h[l]; } }    // inclusive sum of the first elements' lengths
    const unsigned long long mb = wv::ballot(big);
    // the run behind piece k: up to and with the next piece of several elements (or the block's last piece)
    wv::Lane<int> e_of;
    WAVE_FOR(l) { const unsigned long long above = l < 63 ? (mb >> (l + 1)) << (l + 1) : 0ull; e_of[l] = above ? __builtin_ctzll(above) : K - 1; }
    const wv::Lane<int> S_e = wv::gather(S, e_of);
    HP_G cig_t *out = (HP_G cig_t *)fc.c;
    const int B0 = fc.n - 1;
    const int e0 = mb ? __builtin_ctzll(mb) : K - 1;
    const cig_t top_new = top + (cig_t)(wv::bcast(S, e0) << 4);
    wv::Lane<int> adj;
    WAVE_FOR(l) {
        adj[l] = 0;
        if (l == 0) out[B0] = top_new;
        if (big[l]) {
            adj[l] = wl[l] + ((S_e[l] - S[l]) << 4);
            const HP_G cig_t *w = (const HP_G cig_t *)src[l];
            HP_G cig_t *o = out + B0 + base[l];
            for (int t = 1; t < n[l] - 1; ++t) o[t] = w[t];
            o[n[l] - 1] = adj[l];
        }
    }
    top = mb ? (cig_t)wv::bcast(adj, 63 - __builtin_clzll(mb)) : top_new;
    fc.n += n_new;
    ref_end += wv::reduce_sum(rl); re += wv::reduce_sum(ql); r.cs_words += wv::reduce_sum(cw);
    last = seed_at_after_block(g_seed, i, step, nb);
    wv::sync();
    return nb;
}

HP_NOINL bool frag_extend_multi(ReadCtx &r, const FLines &F, int frag, Rec &res)
{
    Ctx &cx = r.cx;
    const lamsa_hp_para *P = cx.P;
    const int32_t *seed = F.fr_seed + F.fr_seed_off[frag];
    const int seed_n = F.fr_seed_off[frag + 1] - F.fr_seed_off[frag];
    const int strand = r.h_strand[seed[0]], chr = r.h_chr[seed[0]];
    const size_t mark = arena_mark(cx.tmp);
    CigV fc;
    int cap = 0;                                          // seed CIGARs + gap CIGARs (gap = seed_step - seed_len bases + indels)
    {   const HP_G int32_t *g_seed = (const HP_G int32_t *)seed;
        for (int b0 = 0; b0 < seed_n; b0 += 64) { wv::Lane<int> c; WAVE_FOR(l) { c[l] = b0 + l < seed_n ? (int)r.h_cig_n[g_seed[b0 + l]] : 0; } cap += wv::reduce_sum(c); }
        cap += seed_n * (2 * iabs(P->seed_step) + 64);
    }
    if (!cig_alloc(cx, fc, cap + 16)) return false;
    int i, rs, re, last;
    if (strand == 1) { i = seed_n - 1; last = seed[i]; rs = (sid(r, r.n_seed[last]) - 1) * P->seed_step + 1; }
    else { i = 0; last = seed[0]; rs = r.last_len + (sid(r, r.n_seed[last]) - 1) * P->seed_step + 1; }
    re = rs - 1 + P->seed_len;
    cig_pushv(cx, fc, r.cig + r.h_cig_off[last], r.h_cig_n[last]); r.cs_words += r.h_cig_n[last];
    const int64_t ref_start = r.h_pos[last];
    int64_t ref_end = r.h_pos[last] + P->seed_len - 1 + r.h_len_dif[last];
    const int step = strand == 1 ? -1 : 1;
    bool ok = true;
    i += step;
    while (i >= 0 && i < seed_n && ok) {
        const int left = step == 1 ? seed_n - i : i + 1;                         // steps to go
        if (left >= HP_FRAG_BLOCK_MIN && fc.n > 0) {
            cig_t top = ((const HP_G cig_t *)fc.c)[fc.n - 1];
            const int took = frag_steps_block(r, F, frag, i, step, left, fc, ref_end, re, last, top);
            if (took) { HP_STAT(24); i += step * took; continue; }
        }
        const int one_by_one = left < 32 ? left : 32;                            // this stretch as the reference walks it
        HP_STAT(25);
        for (int t = 0; t < one_by_one && ok; ++t) { ok = frag_step(r, F, frag, i, chr, fc, ref_end, re, last); i += step; }
    }
    if (ok) ok = merge_cigar(r, res.cig, &res.refend, &res.readend, chr, fc.c, fc.n, (int)(ref_end - ref_start + 1), re - rs + 1);
    arena_release(cx.tmp, mark);
    return ok && !(cx.status & (ST_REFEXIT | ST_OVERFLOW));
}

// A fragment of one seed (every fragment of a noisy read whose neighbours are not exactly colinear): the loop of
// :360-400 does not run and the seed's own CIGAR is merged as it is (:402-405).
HP_INL bool frag_extend(ReadCtx &r, const FLines &F, int frag, Rec &res)
{
    const int o0 = F.fr_seed_off[frag], o1 = F.fr_seed_off[frag + 1];
    if (o1 - o0 != 1) return frag_extend_multi(r, F, frag, res);
    const lamsa_hp_para *P = r.cx.P;
    const int s = F.fr_seed[o0];
    r.cs_words += r.h_cig_n[s];
    const bool ok = merge_cigar(r, res.cig, &res.refend, &res.readend, r.h_chr[s], r.cig + r.h_cig_off[s], r.h_cig_n[s],
                                P->seed_len + r.h_len_dif[s], P->seed_len);
    return ok && !(r.cx.status & (ST_REFEXIT | ST_OVERFLOW));
}

// ---------------------------------------------------------------- split_mapping, :416-564
// Geometry of the junction between two fragments (:424-470), computed once and handed to whichever branch applies.
struct SplitGeo {
    const uint8_t *qp; int64_t at1_off, at2_off; int at1_ld, at1_chr, at2_chr, did, s_qlen, dis, match_dis;
    const int32_t *jt; const int32_t *jarena;        // this junction's slot of FLines::jt (or nullptr) and the arena its offset refers to
};

// DEL / INS / DUP branches (:475-546): the structural-variant cases, rare on ordinary reads
HP_NOINL bool split_sv(ReadCtx &r, const SplitGeo &g, Rec &res)
{
    Ctx &cx = r.cx;
    const lamsa_hp_para *P = cx.P;
    const int hash_len = P->hash_len, s_qlen = g.s_qlen, dis = g.dis;
    const uint8_t *qp = g.qp;
    const int64_t at1_off = g.at1_off, at2_off = g.at2_off;
    const int at1_ld = g.at1_ld, at1_chr = g.at1_chr, at2_chr = g.at2_chr;
    const int gh0 = hash_len * P->match;
    const size_t mark = arena_mark(cx.tmp);
    int s_tlen = 0;
    int32_t tl;
    CigV sc;
    bool ok = true;
    if (dis > g.match_dis) {                                      // DEL, :475-489
        s_tlen = s_qlen + dis; tl = s_tlen;
        uint8_t *tb = (uint8_t *)arena_alloc(cx, (size_t)(s_tlen > 0 ? s_tlen : 0) + 16);
        ok = tb && cig_alloc(cx, sc, s_qlen + s_tlen + 64) && ref_fetch(r, at1_chr, at1_off + P->seed_len + at1_ld - 1, &tl, tb);
        if (ok) {
            s_tlen = tl;
            if (s_qlen < hash_len) ksw_bi_extend(cx, s_qlen, seq_fwd(qp), s_tlen, seq_fwd(tb), gh0, gh0, sc);
            else split_indel_map(cx, sc, qp, s_qlen, tb, s_tlen, 0);
        }
    } else {                                                      // INS, :490-546
        s_tlen = s_qlen + dis;
        if (s_tlen < 2 * P->hash_step) {                          // overlapped insertion: extend from both sides
            int32_t _s_tlen = s_qlen + hash_len;
            int lqe, lte, rqe, rte;
            uint8_t *tb1 = (uint8_t *)arena_alloc(cx, (size_t)_s_tlen + 16), *tb2 = (uint8_t *)arena_alloc(cx, (size_t)_s_tlen + 16);
            CigV lc, rcg;
            ok = tb1 && tb2 && cig_alloc(cx, sc, 2 * (s_qlen + _s_tlen) + 64) && cig_alloc(cx, lc, s_qlen + _s_tlen + 8) && cig_alloc(cx, rcg, s_qlen + _s_tlen + 8);
            ok = ok && ref_fetch(r, at1_chr, at1_off + P->seed_len + at1_ld - 1, &_s_tlen, tb1);
            if (ok) {
                ksw_extend(cx, s_qlen, seq_fwd(qp), _s_tlen, seq_fwd(tb1), P->band_w, gh0, &lqe, &lte, &lc);
                ok = ref_fetch(r, at2_chr, at2_off - _s_tlen - 1, &_s_tlen, tb2);
            }
            if (ok) {
                const Seq rq = seq_rev(qp, s_qlen), rt = seq_rev(tb2, _s_tlen);
                ksw_extend(cx, s_qlen, rq, _s_tlen, rt, P->band_w, gh0, &rqe, &rte, &rcg);
                cig_invert(rcg.c, rcg.n);
                // the reference hands the still-reversed buffers to sw_mid_fix (:527)
                sw_mid_fix(cx, sc, lc.c, lc.n, rcg.c, rcg.n, s_qlen, rq, lqe, rqe, s_qlen + dis, rt, lte, rte);
            }
        } else {                                                  // DUP, :529-545
            s_tlen += 2 * (hash_len - dis); tl = s_tlen;
            uint8_t *tb = (uint8_t *)arena_alloc(cx, (size_t)s_tlen + 16);
            ok = tb && cig_alloc(cx, sc, s_qlen + s_tlen + 64) && ref_fetch(r, at1_chr, at1_off + P->seed_len + at1_ld + dis - hash_len - 1, &tl, tb);
            if (ok) {
                s_tlen = tl;
                const int off_dis = (s_tlen != s_qlen - dis + 2 * hash_len) ? 0 : -dis;
                s_tlen = s_qlen + dis;
                if (s_tlen < hash_len) ksw_bi_extend(cx, s_qlen, seq_fwd(qp), s_tlen, seq_fwd(tb + hash_len - dis), gh0, gh0, sc);
                else split_indel_map(cx, sc, qp, s_qlen, tb + hash_len - dis, s_tlen, off_dis);
            }
        }
    }
    if (ok) ok = merge_cigar(r, res.cig, &res.refend, &res.readend, at1_chr, sc.c, sc.n, s_tlen, s_qlen);
    arena_release(cx.tmp, mark);
    return ok && !(cx.status & (ST_REFEXIT | ST_OVERFLOW));
}

// mismatch class with read bases between the seeds (:547-559): two-sided extension
HP_NOINL bool split_mismatch(ReadCtx &r, const SplitGeo &g, Rec &res)
{
    Ctx &cx = r.cx;
    const lamsa_hp_para *P = cx.P;
    int s_tlen = g.s_qlen + g.dis;
    if (s_tlen < 0) { cx.status |= ST_REFEXIT; return false; }                 // ksw_extend_core exit(-1), ksw.c:672
    if (g.jt && g.jt[3] && g.jarena) {                                          // its CIGAR was computed ahead (hp_lanedp.h)
        return merge_cigar(r, res.cig, &res.refend, &res.readend, g.at1_chr, g.jarena + g.jt[0], g.jt[1], g.jt[2], g.s_qlen) && !(cx.status & (ST_REFEXIT | ST_OVERFLOW));
    }
    const size_t mark = arena_mark(cx.tmp);
    int32_t tl = s_tlen;
    CigV sc;
    uint8_t *tb = (uint8_t *)arena_alloc(cx, (size_t)s_tlen + 16);
    bool ok = tb && cig_alloc(cx, sc, g.s_qlen + s_tlen + 64) && ref_fetch(r, g.at1_chr, g.at1_off + P->seed_len + g.at1_ld - 1, &tl, tb);
    if (ok) {
        s_tlen = tl;
        ksw_bi_extend(cx, g.s_qlen, seq_fwd(g.qp), s_tlen, seq_fwd(tb), 100, 100, sc);
        ok = merge_cigar(r, res.cig, &res.refend, &res.readend, g.at1_chr, sc.c, sc.n, s_tlen, g.s_qlen);
    }
    arena_release(cx.tmp, mark);
    return ok && !(cx.status & (ST_REFEXIT | ST_OVERFLOW));
}

HP_INL bool split_mapping(ReadCtx &r, const FLines &F, int f1, int f2, Rec &res)
{
    Ctx &cx = r.cx;
    const lamsa_hp_para *P = cx.P;
    const int32_t *sd1 = F.fr_seed + F.fr_seed_off[f1], *sd2 = F.fr_seed + F.fr_seed_off[f2];
    const int n1 = F.fr_seed_off[f1 + 1] - F.fr_seed_off[f1], n2 = F.fr_seed_off[f2 + 1] - F.fr_seed_off[f2];
    int s1, s2;
    if (r.h_strand[sd1[0]] == 1) { s1 = sd1[0]; s2 = sd2[n2 - 1]; }
    else { s1 = sd1[n1 - 1]; s2 = sd2[0]; }
    SplitGeo g;
    g.at1_off = r.h_pos[s1]; g.at2_off = r.h_pos[s2];
    g.at1_ld = r.h_len_dif[s1]; g.at1_chr = r.h_chr[s1]; g.at2_chr = r.h_chr[s2];
    g.did = sid(r, r.n_seed[s2]) - sid(r, r.n_seed[s1]);
    g.s_qlen = g.did * P->seed_step - P->seed_len;
    if (g.s_qlen < 0) { cx.status |= ST_REFEXIT; return false; }
    read_gap(r, s1, s2, &g.qp);
    const int64_t exp = g.at1_off + g.at1_ld + (int64_t)(g.did * P->seed_step);
    g.dis = (int)(g.at2_off - exp);
    g.match_dis = P->match_dis * ((P->aln_mode & 2) ? g.did : 1);
    g.jt = F.jt ? F.jt + 4 * (f1 < f2 ? f1 : f2) : nullptr; g.jarena = F.jarena;
    if (g.dis > g.match_dis || g.dis < -g.match_dis) return split_sv(r, g, res);
    if (g.s_qlen > 0) return split_mismatch(r, g, res);
    // Mismatch class with no read base between the seeds (neighbouring seeds overlap by design, so this is most
    // junctions of a noisy read): ksw_bi_extend has nothing to align and returns "delete the whole target"
    // (see the shortcut in ksw_bi_extend).  The reference still cuts the window at the contig end (bntseq.c:469-474),
    // which is reproduced here; the window itself is not needed.
    int s_tlen = g.dis;
    if (s_tlen < 0) { cx.status |= ST_REFEXIT; return false; }                 // ksw_extend_core exit(-1), ksw.c:672
    {
        const int64_t start0 = g.at1_off + P->seed_len + g.at1_ld - 1;
        const int32_t clen = r.ref.seq_len[g.at1_chr - 1];
        if (start0 > clen || start0 < 0) { cx.status |= ST_REFEXIT; return false; }
        if (start0 + s_tlen > clen) s_tlen = (int)(clen - start0);
    }
    if (s_tlen <= 0) return !(cx.status & (ST_REFEXIT | ST_OVERFLOW));          // empty CIGAR: merge_cigar returns at once (:254)
    const size_t mark = arena_mark(cx.tmp);
    cig_t *w = (cig_t *)arena_alloc(cx, sizeof(cig_t));
    bool ok = w != nullptr;
    if (ok) {
        ((HP_G cig_t *)w)[0] = (s_tlen << 4) | C_D;
        wv::sync();
        ok = merge_cigar(r, res.cig, &res.refend, &res.readend, g.at1_chr, w, 1, s_tlen, 0);
    }
    arena_release(cx.tmp, mark);
    return ok && !(cx.status & (ST_REFEXIT | ST_OVERFLOW));
}


// ---------------------------------------------------------------- the junctions and single-seed fragments of a line without a memory
// round trip each.  frag_check (:886-955) alternates frag_extend (:332) and split_mapping (:416) along the line; on a noisy read
// nearly every fragment is one seed (its CIGAR is appended as it is) and nearly every junction is either "neighbouring seeds, no
// read base in between" (a deletion of the reference bases between them) or a small DP whose CIGAR the lane-per-job launch has
// left in the job arena.  Each of these steps is a merge_cigar (:251) whose inputs -- which seed, which CIGAR, its first and
// last element, how far it advances -- do not depend on the steps before it.  They are worked out for 64 steps at a time, one
// step per lane (the dependent loads of 64 steps overlap), and the sequential part keeps the last element of the growing CIGAR in
// a register: appending is then a store, never a load.  Everything else (fragments of several seeds, SV junctions, DPs that were
// not computed ahead, the boundary repair of merge_cigar) goes through the general routines above.
struct JGeo { int s1, s2, at1_ld, at1_chr, did, s_qlen, dis, match_dis, cls, tl; int64_t at1_off, start0; };   // cls 0: general routine, 1: mismatch class with read bases, 2: without
HP_INL void junction_geo(const ReadCtx &r, const FLines &F, int f1, int f2, JGeo &G)
{   // split_mapping :424-470 and the window of its mismatch branch (:547-559, pac2fa_core bntseq.c:469-474)
    const lamsa_hp_para *P = r.cx.P;
    const int32_t *sd1 = F.fr_seed + F.fr_seed_off[f1], *sd2 = F.fr_seed + F.fr_seed_off[f2];
    const int n1 = F.fr_seed_off[f1 + 1] - F.fr_seed_off[f1], n2 = F.fr_seed_off[f2 + 1] - F.fr_seed_off[f2];
    if (r.h_strand[sd1[0]] == 1) { G.s1 = sd1[0]; G.s2 = sd2[n2 - 1]; } else { G.s1 = sd1[n1 - 1]; G.s2 = sd2[0]; }
    G.at1_off = r.h_pos[G.s1]; G.at1_ld = r.h_len_dif[G.s1]; G.at1_chr = r.h_chr[G.s1];
    G.did = sid(r, r.n_seed[G.s2]) - sid(r, r.n_seed[G.s1]);
    G.s_qlen = G.did * P->seed_step - P->seed_len;
    const int64_t exp = G.at1_off + G.at1_ld + (int64_t)(G.did * P->seed_step);
    G.dis = (int)(r.h_pos[G.s2] - exp);
    G.match_dis = P->match_dis * ((P->aln_mode & 2) ? G.did : 1);
    G.cls = 0; G.tl = 0; G.start0 = 0;
    if (G.s_qlen < 0 || G.dis > G.match_dis || G.dis < -G.match_dis || G.s_qlen + G.dis < 0) return;
    G.start0 = G.at1_off + P->seed_len + G.at1_ld - 1;
    const int32_t clen = r.ref.seq_len[G.at1_chr - 1];
    if (G.start0 > clen || G.start0 < 0) return;
    G.tl = G.s_qlen + G.dis;
    if (G.start0 + G.tl > clen) G.tl = (int)(clen - G.start0);
    G.cls = G.s_qlen > 0 ? 1 : 2;
}

// p == nullptr: a one-element CIGAR, the element in `first`; ls: the CIGAR's words staged in the wave's LDS (or nullptr -- NOT tested: `staged` says)
struct MergeSrc { const cig_t *p; int n, first, last, reflen, readlen; const HP_L int32_t *ls; bool staged; };
// Short CIGARs of a 64-step block of frags_merge are staged in the wave's LDS when the block's plan is made (all of them requested at once, one
// step per lane): an append then copies from LDS instead of waiting ~2 us for its own load, 300 times per line one after the other.  Per step
// FM_SW words of its seed CIGAR and FM_JW of its junction's; longer ones are loaded when their turn comes, as before.  The staging lives in the
// DP rows' part of the LDS; a general routine that may run a DP with LDS rows invalidates it for the rest of the block.
enum { FM_SW = 8, FM_JW = 10, FM_W = FM_SW + FM_JW };      // 64 * 18 = 1152 words = the rows' part of HP_LDS_WORDS

// merge_cigar (:251-328) with _push_cigar (frag_check.h:158-184) for the common cases; `tail` = the record's last CIGAR element (valid when n > 0).
// It works on the record's running state held in registers (frags_merge): the length of the record's CIGAR, its reference and read
// ends live in locals across the ~300 steps of a line and are written to `res` only around the calls of the general routines -- every
// use of a field of `res` is a load, a wait and a store otherwise.  ovf: set instead of cx.status when the CIGAR buffer is full.
struct MergeLoc { int n, readend; int64_t refend; };
HP_INL void mloc_out(const MergeLoc &m, Rec &res) { res.cig.n = m.n; res.refend = m.refend; res.readend = m.readend; }
HP_INL void mloc_in(MergeLoc &m, const Rec &res) { m.n = res.cig.n; m.refend = res.refend; m.readend = res.readend; }
// dst / cap: the record's CIGAR buffer and its capacity, read once by the caller (through `res` they are a load from memory in front of every
// store: the compiler cannot keep them across the stores of the copy)
HP_INL bool merge_fast_loc(ReadCtx &r, Rec &res, MergeLoc &m, int &tail, bool &ovf, int chr, const MergeSrc &S, HP_G cig_t *dst, int cap)
{
    if (S.n == 0) return true;
    Ctx &cx = r.cx;
    const int n1 = m.n;
    if (n1 > 1) {
        const int top = tail & 0xf, hop = S.first & 0xf;
        if ((((top == C_I || top == C_D) && (tail >> 4) <= 3) && hop != C_S && hop != C_H) ||
            (((hop == C_I || hop == C_D) && (S.first >> 4) <= 3) && top != C_S && top != C_H)) {           // boundary repair: the general routine
            mloc_out(m, res);
            wv::sync();
            bool ok;
            if (S.p) ok = merge_cigar_full(r, res.cig, &res.refend, &res.readend, chr, S.p, S.n, S.reflen, S.readlen);
            else {
                const size_t mark = arena_mark(cx.tmp);
                cig_t *w = (cig_t *)arena_alloc(cx, sizeof(cig_t));
                ok = w != nullptr;
                if (ok) { ((HP_G cig_t *)w)[0] = S.first; wv::sync(); ok = merge_cigar_full(r, res.cig, &res.refend, &res.readend, chr, w, 1, S.reflen, S.readlen); }
                arena_release(cx.tmp, mark);
            }
            wv::sync();
            mloc_in(m, res);
            tail = m.n > 0 ? (int)dst[m.n - 1] : 0;
            return ok && !(cx.status & (ST_REFEXIT | ST_OVERFLOW));
        }
    }
    int j = 0;
    if (n1 > 0) {
        if ((tail & 0xf) == (S.first & 0xf)) { tail = tail + ((S.first >> 4) << 4); dst[n1 - 1] = tail; j = 1; }
        else if (((tail & 0xf) == C_I && (S.first & 0xf) == C_S) || ((tail & 0xf) == C_S && (S.first & 0xf) == C_I)) { tail = (((tail >> 4) + (S.first >> 4)) << 4) | C_S; dst[n1 - 1] = tail; j = 1; }
    }
    const int mm = S.n - j;
    if (n1 + mm > cap) ovf = true;
    else if (mm > 0) {
        if (S.staged) { WAVE_FOR(l) { if (l < mm) dst[n1 + l] = S.ls[j + l]; } }
        else if (S.p) { const HP_G cig_t *src = (const HP_G cig_t *)S.p; for (int b0 = 0; b0 < mm; b0 += 64) { WAVE_FOR(l) { const int i = b0 + l; if (i < mm) dst[n1 + i] = src[j + i]; } } }
        else dst[n1] = S.first;
        m.n = n1 + mm; tail = S.last;
    }
    m.refend += S.reflen;
    m.readend += S.readlen;
    return !ovf;
}

HP_NOINL bool frags_merge(ReadCtx &r, const FLines &F, int f0, int nfr, int strand, Rec &res)
{
    Ctx &cx = r.cx;
    const lamsa_hp_para *P = cx.P;
    const size_t mark = arena_mark(cx.tmp);
    int32_t *pl = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 16 * 64);
    if (!pl) return false;
    HP_G int32_t *g_pl = (HP_G int32_t *)pl;
    wv::sync();
    bool ok = true, ovf = false;
    long long cs_ = 0;                               // seed-CIGAR words appended (accounting, flushed once)
    MergeLoc ml; mloc_in(ml, res);                   // the record's running state, in registers
    // what the sequential loop below needs of the structures it is handed, read once into (scalar) registers
    HP_G cig_t *const res_c = (HP_G cig_t *)wv::uni64((long long)res.cig.c); const int res_cap = wv::uni(res.cig.cap);
    const cig_t *const seed_cigs = (const cig_t *)wv::uni64((long long)r.cig); const cig_t *const job_cigs = (const cig_t *)wv::uni64((long long)F.jarena);
    const int seed_len = wv::uni(P->seed_len);
    int tail = ml.n > 0 ? (int)res_c[ml.n - 1] : 0;
    HP_L int32_t *const stg = cx.lds;
    const bool stage = cx.lds_words >= 64 * FM_W;
    for (int t0 = 0; t0 < nfr && ok; t0 += 64) {
        HP_T0(tg64_);
        bool staged_ok = stage;                      // the block's staged CIGARs are intact
        const int epoch0 = cx.lds_epoch;             // (a boundary repair's DP may have been large enough to use the LDS rows: checked after each)
        wv::sync();                                  // (the previous block's are no longer read)
        // ---- 64 steps, one per lane: the fragment's seed CIGAR and the junction to the next fragment
        WAVE_FOR(l) {
            const int t = t0 + l;
            int v[16];
            for (int k = 0; k < 16; ++k) v[k] = 0;
            if (t < nfr) {
                const int f = strand == 1 ? f0 + nfr - 1 - t : f0 + t;
                const int o0 = F.fr_seed_off[f], o1 = F.fr_seed_off[f + 1];
                v[0] = o1 - o0 == 1;
                if (v[0]) {
                    const int s = F.fr_seed[o0];
                    const int64_t co = r.h_cig_off[s]; const int cn = r.h_cig_n[s];
                    v[1] = (int)(co & 0xffffffffll); v[2] = (int)(co >> 32); v[3] = cn; v[6] = P->seed_len + r.h_len_dif[s]; v[7] = r.h_chr[s];
                    if (cn > 0) { v[4] = r.cig[co]; v[5] = r.cig[co + cn - 1]; }
                    if (stage && cn > 0 && cn <= FM_SW) {
                        int w[FM_SW];
#pragma unroll
                        for (int k = 0; k < FM_SW; ++k) w[k] = k < cn ? (int)r.cig[co + k] : 0;
#pragma unroll
                        for (int k = 0; k < FM_SW; ++k) stg[l * FM_W + k] = w[k];
                    }
                }
                if (t < nfr - 1) {
                    const int f2 = strand == 1 ? f - 1 : f + 1;
                    JGeo G; junction_geo(r, F, f, f2, G);
                    v[8] = 3; v[15] = G.at1_chr;                                   // 3: the general routine
                    if (G.cls == 2) { v[8] = G.tl > 0 ? 1 : 0; v[10] = 1; v[11] = v[12] = (G.tl << 4) | C_D; v[13] = G.tl; v[14] = 0; }       // 1: deletion of the bases between the seeds, 0: nothing
                    else if (G.cls == 1) {
                        const int32_t *jt = F.jt ? F.jt + 4 * (f < f2 ? f : f2) : nullptr;
                        if (jt && jt[3] && F.jarena) {                                  // 2: computed ahead
                            v[8] = 2; v[9] = jt[0]; v[10] = jt[1]; v[13] = jt[2]; v[14] = G.s_qlen; if (jt[1] > 0) { v[11] = F.jarena[jt[0]]; v[12] = F.jarena[jt[0] + jt[1] - 1]; }
                            if (stage && jt[1] > 0 && jt[1] <= FM_JW) {
                                int w[FM_JW];
#pragma unroll
                                for (int k = 0; k < FM_JW; ++k) w[k] = k < jt[1] ? (int)F.jarena[jt[0] + k] : 0;
#pragma unroll
                                for (int k = 0; k < FM_JW; ++k) stg[l * FM_W + FM_SW + k] = w[k];
                            }
                        }
                    }
                }
            }
            for (int k = 0; k < 16; ++k) g_pl[k * 64 + l] = v[k];
        }
        wv::sync();
        wv::Lane<int> V[16];
        WAVE_FOR(l) { for (int k = 0; k < 16; ++k) V[k][l] = g_pl[k * 64 + l]; }
        const int cnt = nfr - t0 < 64 ? nfr - t0 : 64;
        HP_TADD_FILL(cx, 22, tg64_);
        for (int q = 0; q < cnt && ok; ++q) {
            const int t = t0 + q, f = strand == 1 ? f0 + nfr - 1 - t : f0 + t;
            // frag_extend, :332-410
            if (wv::bcast(V[0], q)) {
#if defined(HP_PROF) && defined(HP_PROF_FILL)
                const long long tpa_ = wv::clock();
#endif
                MergeSrc S;
                S.p = seed_cigs + (((int64_t)wv::bcast(V[2], q) << 32) | (unsigned)wv::bcast(V[1], q)); S.n = wv::bcast(V[3], q); S.first = wv::bcast(V[4], q); S.last = wv::bcast(V[5], q);
                S.reflen = wv::bcast(V[6], q); S.readlen = seed_len;
                if (cx.lds_epoch != epoch0) staged_ok = false;
                S.ls = stg + q * FM_W; S.staged = staged_ok && S.n <= FM_SW;
                cs_ += S.n;
#if defined(HP_PROF) && defined(HP_PROF_FILL)
                const long long tpb_ = wv::clock();
#endif
                ok = merge_fast_loc(r, res, ml, tail, ovf, wv::bcast(V[7], q), S, res_c, res_cap);
#if defined(HP_PROF) && defined(HP_PROF_FILL)
                if (cx.prof) { cx.prof[13] += tpb_ - tpa_; cx.prof[11] += wv::clock() - tpb_; }
#endif
            } else {
                mloc_out(ml, res);
                wv::sync();
                HP_T0(tfm_);
                staged_ok = false;                   // (its DPs may use the LDS rows)
                ok = frag_extend_multi(r, F, f, res);
                HP_TADD_FILL(cx, 20, tfm_);
                wv::sync();
                mloc_in(ml, res);
                tail = ml.n > 0 ? (int)res_c[ml.n - 1] : 0;
            }
            if (!ok || t == nfr - 1) continue;
            // split_mapping, :416-564
            const int kind = wv::bcast(V[8], q);
            if (kind == 3) {
                mloc_out(ml, res);
                wv::sync();
                HP_T0(tsm_);
                staged_ok = false;
                ok = split_mapping(r, F, f, strand == 1 ? f - 1 : f + 1, res);
                HP_TADD_FILL(cx, 18, tsm_);
                wv::sync();
                mloc_in(ml, res);
                tail = ml.n > 0 ? (int)res_c[ml.n - 1] : 0;
            } else if (kind != 0) {
#if defined(HP_PROF) && defined(HP_PROF_FILL)
                const long long tpc_ = wv::clock();
#endif
                MergeSrc S;
                S.p = kind == 2 ? job_cigs + wv::bcast(V[9], q) : nullptr; S.n = wv::bcast(V[10], q); S.first = wv::bcast(V[11], q); S.last = wv::bcast(V[12], q);
                S.reflen = wv::bcast(V[13], q); S.readlen = wv::bcast(V[14], q);
                if (cx.lds_epoch != epoch0) staged_ok = false;
                S.ls = stg + q * FM_W + FM_SW; S.staged = staged_ok && kind == 2 && S.n <= FM_JW;
                ok = merge_fast_loc(r, res, ml, tail, ovf, wv::bcast(V[15], q), S, res_c, res_cap);
#if defined(HP_PROF) && defined(HP_PROF_FILL)
                if (cx.prof) cx.prof[12] += wv::clock() - tpc_;
#endif
            }
        }
    }
    mloc_out(ml, res);
    r.cs_words += cs_;
    if (ovf) cx.status |= ST_OVERFLOW;
    wv::sync();
    arena_release(cx.tmp, mark);
    return ok && !(cx.status & (ST_REFEXIT | ST_OVERFLOW));
}

// ---------------------------------------------------------------- frag_head_bound_fix, :576-654
HP_NOINL bool head_fix(ReadCtx &r, const FLines &F, int line, Rec &res)
{
    Ctx &cx = r.cx;
    const lamsa_hp_para *P = cx.P;
    const int left_bound = F.left_bound[line];
    const int f0 = F.frag_off[line], fl = F.frag_off[line + 1] - 1;
    int s, read_len, read_start;
    if (r.h_strand[F.fr_seed[F.fr_seed_off[f0]]] == 1) {
        s = F.fr_seed[F.fr_seed_off[fl + 1] - 1];                       // last seed of the last fragment
        if (sid(r, r.n_seed[s]) != 1) {
            read_len = (left_bound == 0 ? 0 : P->seed_inv) + (sid(r, r.n_seed[s]) - left_bound - 1) * P->seed_step;
            if (read_len < 0) { cx.status |= ST_REFEXIT; return false; }
            read_start = left_bound == 0 ? 0 : left_bound * P->seed_step - P->seed_inv;
        } else { res.offset = r.h_pos[s]; res.refend = res.offset - 1; res.cig.n = 0; return true; }
    } else {
        s = F.fr_seed[F.fr_seed_off[f0]];
        read_len = (left_bound == 0 ? r.last_len : P->seed_inv) + (sid(r, r.n_seed[s]) - 1 - left_bound) * P->seed_step;
        if (read_len == 0) { res.offset = r.h_pos[s]; res.refend = res.offset - 1; res.cig.n = 0; return true; }
        if (read_len < 0) { cx.status |= ST_REFEXIT; return false; }
        read_start = left_bound == 0 ? 0 : r.last_len + left_bound * P->seed_step - P->seed_inv;
    }
    res.offset = r.h_pos[s];
    {   const int32_t *ht = F.ht ? F.ht + 16 * line : nullptr;
        if (ht && ht[4] && F.jarena) {                                      // the extension was computed ahead (hp_wavejob.h): its CIGAR, clipped and turned round
            HP_STAT(20);
            res.offset -= ht[2];
            res.refend = res.offset - 1;
            cig_pushv(cx, res.cig, F.jarena + ht[0], ht[1]);                // _push_cigar_e, frag_check.h:193
            res.refend += ht[2];
            res.readend += ht[3];
            return !(cx.status & (ST_REFEXIT | ST_OVERFLOW));
        }
    }
    int32_t ref_len = read_len + P->hash_step * 2;
    int64_t ref_start = r.h_pos[s] - ref_len;
    if (ref_start < 1) { ref_start = 1; ref_len = (int32_t)(r.h_pos[s] - 1); }
    const size_t mark = arena_mark(cx.tmp);
    uint8_t *tb = (uint8_t *)arena_alloc(cx, (size_t)(ref_len > 0 ? ref_len : 0) + 16);
    CigV c;
    bool ok = tb && cig_alloc(cx, c, read_len + ref_len + 16) && ref_fetch(r, r.h_chr[s], ref_start - 1, &ref_len, tb);
    if (ok) {
        int qre, tre;
        const int rr = ksw_extend_r(cx, read_len, seq_fwd(r.cur_read + read_start), ref_len, seq_fwd(tb), P->band_w, P->seed_len * P->match, &qre, &tre, &c);
        if (rr != 0) cig_push1(cx, c, ((read_len - qre) << 4) | C_S);
        cig_invert(c.c, c.n);
        res.offset -= cig_reflen(c.c, c.n);
        res.refend = res.offset - 1;
        cig_pushv(cx, res.cig, c.c, c.n);                              // _push_cigar_e, frag_check.h:193
        res.refend += cig_reflen(c.c, c.n);
        res.readend += cig_readlen(c.c, c.n);
    }
    arena_release(cx.tmp, mark);
    return ok && !(cx.status & (ST_REFEXIT | ST_OVERFLOW));
}

// ---------------------------------------------------------------- frag_tail_bound_fix, :656-707
HP_NOINL bool tail_fix(ReadCtx &r, const FLines &F, int line, Rec &res)
{
    Ctx &cx = r.cx;
    const lamsa_hp_para *P = cx.P;
    const int right_bound = F.right_bound[line];
    const int f0 = F.frag_off[line], fl = F.frag_off[line + 1] - 1;
    int s, read_len, read_start;
    if (r.h_strand[F.fr_seed[F.fr_seed_off[f0]]] == 1) {
        s = F.fr_seed[F.fr_seed_off[f0]];
        read_start = sid(r, r.n_seed[s]) * P->seed_step - P->seed_inv;
        read_len = (right_bound == r.seed_all + 1 ? r.last_len : P->seed_inv) + (right_bound - 1 - sid(r, r.n_seed[s])) * P->seed_step;
        if (read_len == 0) return true;
        if (read_len < 0) { cx.status |= ST_REFEXIT; return false; }
    } else {
        s = F.fr_seed[F.fr_seed_off[fl + 1] - 1];
        if (sid(r, r.n_seed[s]) == r.seed_all) return true;
        read_start = sid(r, r.n_seed[s]) * P->seed_step - P->seed_inv + r.last_len;
        read_len = (right_bound == r.seed_all + 1 ? 0 : P->seed_inv) + (right_bound - 1 - sid(r, r.n_seed[s])) * P->seed_step;
        if (read_len < 0) { cx.status |= ST_REFEXIT; return false; }
    }
    {   const int32_t *ht = F.ht ? F.ht + 16 * line + 8 : nullptr;
        if (ht && ht[4] && F.jarena) {                                      // computed ahead (hp_wavejob.h), clipped
            HP_STAT(21);
            return merge_cigar(r, res.cig, &res.refend, &res.readend, r.h_chr[s], F.jarena + ht[0], ht[1], ht[2], ht[3]) && !(cx.status & (ST_REFEXIT | ST_OVERFLOW));
        }
    }
    int32_t ref_len = read_len + P->hash_step * 2;
    const int64_t ref_start = r.h_pos[s] + P->seed_len + r.h_len_dif[s];
    const size_t mark = arena_mark(cx.tmp);
    uint8_t *tb = (uint8_t *)arena_alloc(cx, (size_t)ref_len + 16);
    CigV c;
    bool ok = tb && cig_alloc(cx, c, read_len + ref_len + 16) && ref_fetch(r, r.h_chr[s], ref_start - 1, &ref_len, tb);
    if (ok) {
        int qle, tle;
        const int rr = ksw_extend_c(cx, read_len, seq_fwd(r.cur_read + read_start), ref_len, seq_fwd(tb), P->band_w, P->seed_len * P->match, &qle, &tle, &c);
        if (rr != 0) cig_push1(cx, c, ((read_len - qle) << 4) | C_S);
        ok = merge_cigar(r, res.cig, &res.refend, &res.readend, r.h_chr[s], c.c, c.n, cig_reflen(c.c, c.n), cig_readlen(c.c, c.n));
    }
    arena_release(cx.tmp, mark);
    return ok && !(cx.status & (ST_REFEXIT | ST_OVERFLOW));
}

// ---------------------------------------------------------------- lamsa_res_split, :712-776
// The whole-line CIGAR in rec[0] (in its own buffer) is cut into records whose CIGARs live back to back in `buf`.
// Sequential by nature (every element is pushed with _push_cigar1's merge rule), so it is organised to touch memory
// as little as possible: 64 input words per coalesced load, handed out by readlane; the run being built, the read
// bases consumed so far (what the reference recomputes with readInCigar at every cut, :738,:748,:762) and the
// reference bases of the current record (refInCigar in push_res, :228) stay in registers; outputs are only stored.
HP_NOINL bool res_split(ReadCtx &r, LineRes &la, cig_t *buf, int buf_cap)
{
    Ctx &cx = r.cx;
    const lamsa_hp_para *P = cx.P;
    const int read_len = r.L, split_len = P->split_len;
    const int n = la.rec[0].cig.n;
    const HP_G cig_t *src = (const HP_G cig_t *)la.rec[0].cig.c;
    if (la.rec[0].cig.c == buf) { cx.status |= ST_REFEXIT; return false; }           // in and out must be different buffers
    HP_G cig_t *oc = (HP_G cig_t *)buf;
    {   // Most lines have no cut point and a CIGAR that _push_cigar1 would copy unchanged (no empty element, no two
        // neighbours of the same kind, only M/I/D/S): checked by the lanes, then copied by the lanes.
        bool plain = n <= buf_cap;
        for (int j0 = 0; j0 < n && plain; j0 += 64) {
            wv::Lane<int> bad;
            WAVE_FOR(l) {
                const int j = j0 + l;
                int b = 0;
                if (j < n) {
                    const int w = src[j], op = w & 0xf, len = w >> 4;
                    const int nxt = j + 1 < n ? (int)src[j + 1] : -1;
                    b = len == 0 || (op != C_M && op != C_I && op != C_D && op != C_S) || ((op == C_I || op == C_D) && len >= split_len) ||
                        (nxt >= 0 && (nxt & 0xf) == op) || (op == C_S && j > 0 && j < n - 1 && nxt >= 0 && (nxt & 0xf) == C_H);
                }
                bad[l] = b;
            }
            if (wv::ballot(bad) != 0) plain = false;
        }
        if (plain) {
            for (int j0 = 0; j0 < n; j0 += 64) { WAVE_FOR(l) { const int j = j0 + l; if (j < n) oc[j] = src[j]; } }
            cig_bind(la.rec[0].cig, buf, buf_cap); la.rec[0].cig.n = n;
            wv::sync();
            return !(cx.status & ST_OVERFLOW);
        }
    }
    int res_n = 0, used = 0;                  // current record, words of `buf` taken by finished records
    int wn = 0, pend = 0; bool have = false;  // words stored for the current record, run being built
    int rd_tot = 0, ref_rec = 0;              // read bases consumed by all elements so far; reference bases of the current record
#define HP_RS_PUSH(w_) do { const int v_ = (w_); if ((v_ >> 4) != 0) { \
        if (have && (pend & 0xf) == (v_ & 0xf)) pend += (v_ >> 4) << 4; \
        else { if (have) { if (used + wn < buf_cap) oc[used + wn++] = pend; else cx.status |= ST_OVERFLOW; } pend = v_; have = true; } } } while (0)
#define HP_RS_NEW_REC(extra_ref) do { \
        if (have) { if (used + wn < buf_cap) oc[used + wn++] = pend; else cx.status |= ST_OVERFLOW; have = false; } \
        cig_bind(la.rec[res_n].cig, buf + used, wn); la.rec[res_n].cig.n = wn; \
        used += wn; wn = 0; \
        if (res_n + 1 >= HP_REC_MAX) { cx.status |= ST_OVERFLOW; return false; } \
        ++res_n; ++la.cur_res_n; \
        la.rec[res_n].chr = la.rec[res_n - 1].chr; la.rec[res_n].nstrand = la.rec[res_n - 1].nstrand;        /* push_res, :228 */ \
        la.rec[res_n].offset = la.rec[res_n - 1].offset + ref_rec + (extra_ref); ref_rec = 0; } while (0)
    bool skip = false;
    for (int j0 = 0; j0 < n; j0 += 64) {
        wv::Lane<int> W;
        WAVE_FOR(l) { const int j = j0 + l; W[l] = j < n ? src[j] : 0; }
        const int peek = j0 + 64 < n ? (int)src[j0 + 64] : 0;
        const int cnt = n - j0 < 64 ? n - j0 : 64;
        for (int jj = 0; jj < cnt; ++jj) {
            if (skip) { skip = false; continue; }
            const int j = j0 + jj;
            const int cw = wv::bcast(W, jj), nw = jj < 63 ? wv::bcast(W, jj + 1) : peek;
            const int op = cw & 0xf, len = cw >> 4;
            if (op == C_M) { HP_RS_PUSH(cw); rd_tot += len; if (len != 0) ref_rec += len; }
            else if (op == C_I && len >= split_len) {
                HP_RS_PUSH(((read_len - rd_tot) << 4) | C_S);
                HP_RS_NEW_REC(0);
                HP_RS_PUSH(((len + rd_tot) << 4) | C_S);
                rd_tot += len;
            } else if (op == C_D && len >= split_len) {
                HP_RS_PUSH(((read_len - rd_tot) << 4) | C_S);
                HP_RS_NEW_REC(len);
                HP_RS_PUSH((rd_tot << 4) | C_S);
            } else if (op == C_I) { HP_RS_PUSH(cw); rd_tot += len; }
            else if (op == C_D) { HP_RS_PUSH(cw); ref_rec += len; }
            else if (op == C_S) {
                if (j > 0 && j < n - 1 && (nw & 0xf) == C_H) {
                    const int Sn = len, Hn = nw >> 4;
                    HP_RS_PUSH(((read_len - rd_tot) << 4) | C_S);
                    HP_RS_NEW_REC(Hn);
                    HP_RS_PUSH(((rd_tot + Sn) << 4) | C_S);
                    rd_tot += Sn;
                    skip = true;
                } else { HP_RS_PUSH(cw); rd_tot += len; }
            } else if (op != C_H) { cx.status |= ST_REFEXIT; return false; }
        }
    }
    if (have) { if (used + wn < buf_cap) oc[used + wn++] = pend; else cx.status |= ST_OVERFLOW; }
    cig_bind(la.rec[res_n].cig, buf + used, buf_cap - used); la.rec[res_n].cig.n = wn;
#undef HP_RS_PUSH
#undef HP_RS_NEW_REC
    wv::sync();
    return !(cx.status & ST_OVERFLOW);
}

// ---------------------------------------------------------------- lamsa_res_aux, :793-853
HP_NOINL bool res_aux(ReadCtx &r, LineRes &la)
{
    Ctx &cx = r.cx;
    const lamsa_hp_para *P = cx.P;
    for (int m = 0; m <= la.cur_res_n; ++m) {
        Rec &rec = la.rec[m];
        const size_t mark = arena_mark(cx.tmp);
        int32_t ref_len = cig_reflen(rec.cig.c, rec.cig.n);
        uint8_t *ref = (uint8_t *)arena_alloc(cx, (size_t)(ref_len > 0 ? ref_len : 0) + 16);
        if (!ref || !ref_fetch(r, rec.chr, rec.offset - 1, &ref_len, ref)) { arena_release(cx.tmp, mark); return false; }
        int ref_i = 0, read_i = 0, n_mm = 0, n_m = 0, n_io = 0, n_ie = 0, n_do = 0, n_de = 0;
        bool bad = false;
        // 64 CIGAR elements at a time, one per lane: where each starts on the read and on the reference is a prefix sum
        // over the lanes; every lane then counts the mismatches of its own M run (:806-834).  Any inconsistency makes
        // the reference exit (:834), in whatever order it is found.
        const HP_G cig_t *gc = (const HP_G cig_t *)rec.cig.c;
        const HP_G uint8_t *gread = (const HP_G uint8_t *)r.cur_read, *gref = (const HP_G uint8_t *)ref;
        const int cn = rec.cig.n;
        for (int c0 = 0; c0 < cn && !bad; c0 += 64) {
            wv::Lane<int> rinc, finc, opl, lenl, isbad;
            WAVE_FOR(l) {
                const int i = c0 + l;
                int op = -1, len = 0;
                if (i < cn) { const cig_t w = gc[i]; op = w & 0xf; len = w >> 4; }
                opl[l] = op; lenl[l] = len;
                rinc[l] = (op == C_M || op == C_I || op == C_S) ? len : 0;
                finc[l] = (op == C_M || op == C_D) ? len : 0;
                isbad[l] = i < cn && op != C_M && op != C_I && op != C_D && op != C_S;
            }
            const int r_tot = wv::reduce_sum(rinc), f_tot = wv::reduce_sum(finc);
            wv::Lane<int> rs = rinc, fs = finc;
            wv::scan_add_excl(rs); wv::scan_add_excl(fs);
            wv::Lane<int> mlen, mm, il, dl, io, dq;
            WAVE_FOR(l) {
                const int op = opl[l], len = lenl[l];
                int ml = 0;
                if (op == C_M) {
                    if (len < 0 || read_i + rs[l] + len > r.L || ref_i + fs[l] + len > ref_len) isbad[l] = 1;     // lengths cannot match any more: exit(1) at :834
                    else ml = len;
                }
                mlen[l] = ml; mm[l] = 0;
                il[l] = op == C_I ? len : 0; dl[l] = op == C_D ? len : 0; io[l] = op == C_I; dq[l] = op == C_D;
            }
            if (wv::ballot(isbad) != 0) { bad = true; break; }
            // The mismatches of the M runs, 64 read bases at a time, one per lane: a lane finds the element its base lies in by a binary
            // search over the elements' read ends (lane gathers, no memory), and with it the reference base that faces it.  (One lane
            // per M run, a base per trip, was as many dependent loads as the longest run of the 64 has bases.)
            {
                wv::Lane<int> rend, delta, ism;
                WAVE_FOR(l) { rend[l] = rs[l] + rinc[l]; delta[l] = fs[l] - rs[l]; ism[l] = mlen[l] > 0; }
                // four passes of 64 bases in flight: their searches are independent (lane gathers pipeline) and their eight loads are requested
                // together -- one pass at a time waited ~2 us for its two bytes, 157 times per 10-kbp record
                enum { RA_U = 4 };
                for (int q0 = 0; q0 < r_tot; q0 += 64 * RA_U) {
                    wv::Lane<int> e[RA_U];
#pragma unroll
                    for (int u = 0; u < RA_U; ++u) { WAVE_FOR(l) { e[u][l] = 0; } }
#pragma unroll
                    for (int step = 32; step >= 1; step >>= 1) {
#pragma unroll
                        for (int u = 0; u < RA_U; ++u) {
                            wv::Lane<int> probe;
                            WAVE_FOR(l) { probe[l] = e[u][l] + step - 1; }
                            const wv::Lane<int> v = wv::gather(rend, probe);
                            WAVE_FOR(l) { if (v[l] <= q0 + 64 * u + l) e[u][l] += step; }        // e = elements that end at or before this base
                        }
                    }
                    wv::Lane<int> a[RA_U], b[RA_U];
#pragma unroll
                    for (int u = 0; u < RA_U; ++u) {
                        const wv::Lane<int> dv = wv::gather(delta, e[u]), mv = wv::gather(ism, e[u]);
                        WAVE_FOR(l) { const int q = q0 + 64 * u + l; a[u][l] = 0; b[u][l] = 0; if (q < r_tot && mv[l]) { a[u][l] = gread[read_i + q]; b[u][l] = gref[ref_i + q + dv[l]] + 256; } }
                    }
#pragma unroll
                    for (int u = 0; u < RA_U; ++u) { WAVE_FOR(l) { mm[l] += b[u][l] != 0 && a[u][l] != b[u][l] - 256; } }
                }
            }
            const int mms = wv::reduce_sum(mm);
            n_mm += mms; n_m += wv::reduce_sum(mlen) - mms;
            n_ie += wv::reduce_sum(il); n_de += wv::reduce_sum(dl);
            n_io += __builtin_popcountll(wv::ballot(io)); n_do += __builtin_popcountll(wv::ballot(dq));
            read_i += r_tot; ref_i += f_tot;
        }
        arena_release(cx.tmp, mark);
        if (bad || read_i != r.L || ref_i != ref_len) { cx.status |= ST_REFEXIT; return false; }
        rec.NM = n_mm + n_ie + n_de;
        rec.score = n_m * P->match - n_mm * P->mis - n_io * P->ins_gapo - n_ie * P->ins_gape - n_do * P->del_gapo - n_de * P->del_gape;
        if (rec.score < 0) {                                          // record deleted, :839-844 (CIGARs are views: no copy needed)
            for (int i = m + 1; i <= la.cur_res_n; ++i) la.rec[i - 1] = la.rec[i];
            --m; --la.cur_res_n;
        } else { la.tol_score += rec.score; la.tol_NM += rec.NM; }
    }
    if (la.cur_res_n < 0) la.tol_score = -1;
    else la.tol_score -= la.cur_res_n * P->split_pen;
    return true;
}

// ---------------------------------------------------------------- one line of frag_check, :886-955

#ifdef HP_PROF
#define HP_TIMED(slot, call) ([&]() { HP_T0(t_); const bool ok_ = (call); HP_TADD(r.cx, slot, t_); return ok_; }())
#else
#define HP_TIMED(slot, call) (call)
#endif
HP_NOINL bool fill_line(ReadCtx &r, FLines &F, int line, LineRes &la, cig_t *cur_buf, int cur_cap, cig_t *rec_buf, int rec_cap)
{
    Ctx &cx = r.cx;
    const lamsa_hp_para *P = cx.P;
    const int f0 = F.frag_off[line], nfr = F.frag_off[line + 1] - f0;
    const int first_seed = F.fr_seed[F.fr_seed_off[f0]];
    const int strand = r.h_strand[first_seed];
    la.line_score = F.line_score[line]; la.cur_res_n = 0; la.tol_score = la.tol_NM = 0;
    Rec &r0 = la.rec[0];
    cig_bind(r0.cig, cur_buf, cur_cap);
    r0.nstrand = strand == 1 ? 1 : 0; r0.chr = r.h_chr[first_seed]; r0.readend = 0; r0.refend = 0; r0.offset = 0;
    bool ok = true;
    if (strand == 1) {
        r.cur_read = r.read; r.flip = false;
        if (F.left_bound[line] > 0) { const cig_t w = ((F.left_bound[line] * P->seed_step - P->seed_inv) << 4) | C_S; cig_push1(cx, r0.cig, w); r0.readend += (int)(w >> 4); }
        ok = HP_TIMED(32, head_fix(r, F, line, r0));
        ok = ok && HP_TIMED(34, frags_merge(r, F, f0, nfr, strand, r0)) && HP_TIMED(38, tail_fix(r, F, line, r0));
        if (ok && F.right_bound[line] <= r.seed_all) { const cig_t w = ((r.L - (F.right_bound[line] - 1) * P->seed_step) << 4) | C_S; cig_push1(cx, r0.cig, w); r0.readend += (int)(w >> 4); }
    } else {
        if (!r.rc_ready) {                                            // :922-925 (buffer reserved when the read was set up)
            for (int b = 0; b < r.L; b += 64) { WAVE_FOR(l) { const int i = b + l; if (i < r.L) { const int c = r.read[r.L - 1 - i]; r.rc_read[i] = c < 4 ? 3 - c : 4; } } }
            wv::sync();
            r.rc_ready = true;
        }
        r.cur_read = r.rc_read; r.flip = true;                        // :926
        const int tmp = F.left_bound[line];
        F.left_bound[line] = r.seed_all + 1 - F.right_bound[line]; F.right_bound[line] = r.seed_all + 1 - tmp;
        if (F.left_bound[line] > 0) { const cig_t w = ((F.left_bound[line] * P->seed_step - P->seed_inv + r.last_len) << 4) | C_S; cig_push1(cx, r0.cig, w); r0.readend += (int)(w >> 4); }
        ok = HP_TIMED(32, head_fix(r, F, line, r0));
        ok = ok && HP_TIMED(34, frags_merge(r, F, f0, nfr, strand, r0)) && HP_TIMED(38, tail_fix(r, F, line, r0));
        if (ok && F.right_bound[line] <= r.seed_all) { const cig_t w = (((r.seed_all - F.right_bound[line] + 1) * P->seed_step - P->seed_inv) << 4) | C_S; cig_push1(cx, r0.cig, w); r0.readend += (int)(w >> 4); }
    }
    ok = ok && HP_TIMED(40, res_split(r, la, rec_buf, rec_cap)) && HP_TIMED(42, res_aux(r, la));
    r.flip = false;                                                   // :953
    return ok && !(cx.status & (ST_REFEXIT | ST_OVERFLOW));
}

}  // namespace hp
