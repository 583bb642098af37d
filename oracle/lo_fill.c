/* lo_fill.c -- lines of fragments -> base-level alignments (oracle; see lo.h).
 *
 * Restates the reference's src/frag_check.c:
 *   merge_cigar       :251   boundary repair when CIGARs are concatenated
 *   frag_extend       :332   global alignment of the gaps between seeds of a fragment
 *   split_mapping     :416   DEL / INS / DUP / mismatch handling between fragments
 *   frag_head/tail_bound_fix :576,:656   extension to the read ends
 *   lamsa_res_split   :712   split at long I/D and nS mH pairs
 *   lamsa_res_aux     :793   NM / AS, drop records with AS < 0
 *   frag_check        :856   driver
 * Where the reference exit(1)s, functions here return -1 and the read is reported as such.
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include "lo_read.h"

#define CIG_M 1000

typedef struct {
    lo_seeds *S; const lo_ref *R; const lo_para *P;
    const uint8_t *read;        /* strand-appropriate read (forward or reverse complement) */
    int read_len;
} fctx;

#define FH(f, s) ((f)->S->hit[(f)->S->hit_off[(s).x] + (s).y])
#define SID(f, s) ((f)->S->seed_id[(s).x])

/* ---------------------------------------------------------------- result containers */
static void res_init(lo_res *r) { memset(r, 0, sizeof(*r)); lo_cigv_init(&r->cig); }
static void lres_init(lo_lres *l, int xa_m)
{
    memset(l, 0, sizeof(*l));
    l->res_m = 10; l->res = (lo_res*)malloc(10 * sizeof(lo_res));
    for (int j = 0; j < 10; ++j) res_init(&l->res[j]);
    l->XA_m = xa_m; l->XA_stage = (int*)malloc(sizeof(int) * (size_t)(xa_m + 1)); l->XA_line = (int*)malloc(sizeof(int) * (size_t)(xa_m + 1)); l->XA_res = (int*)malloc(sizeof(int) * (size_t)(xa_m + 1));
}
void lo_ares_init(lo_ares *a, int res_mul_max)
{   /* aln_init_res, lamsa_aln.c:369 */
    memset(a, 0, sizeof(*a));
    a->l_m = 1; a->la = (lo_lres*)malloc(sizeof(lo_lres));
    lres_init(&a->la[0], res_mul_max);
}
void lo_ares_reset(lo_ares *a, int read_len)
{   /* aln_reset_res, lamsa_aln.c:396 */
    a->l_n = 0; a->read_len = read_len;
    for (int i = 0; i < a->l_m; ++i) {
        a->la[i].cur_res_n = 0;
        for (int j = 0; j < a->la[i].res_m; ++j) a->la[i].res[j].cig.n = 0;
        a->la[i].tol_score = a->la[i].tol_NM = 0; a->la[i].XA_n = 0;
    }
}
void lo_ares_free(lo_ares *a)
{
    for (int i = 0; i < a->l_m; ++i) {
        for (int j = 0; j < a->la[i].res_m; ++j) lo_cigv_free(&a->la[i].res[j].cig);
        free(a->la[i].res); free(a->la[i].XA_stage); free(a->la[i].XA_line); free(a->la[i].XA_res);
    }
    free(a->la); memset(a, 0, sizeof(*a));
}
static void ares_room(lo_ares *a, int line_n, int xa_m)
{   /* aln_reloc_res, lamsa_aln.c:415 */
    if (line_n <= a->l_m) return;
    a->la = (lo_lres*)realloc(a->la, sizeof(lo_lres) * (size_t)line_n);
    for (int i = a->l_m; i < line_n; ++i) lres_init(&a->la[i], xa_m);
    a->l_m = line_n;
}
static void push_res(lo_lres *la)
{   /* frag_check.c:228-247 */
    if (la->cur_res_n == la->res_m - 1) {
        int m = la->res_m << 1;
        la->res = (lo_res*)realloc(la->res, sizeof(lo_res) * (size_t)m);
        for (int i = la->res_m; i < m; ++i) res_init(&la->res[i]);
        la->res_m = m;
    }
    ++la->cur_res_n;
    la->res[la->cur_res_n].chr = la->res[la->cur_res_n - 1].chr;
    la->res[la->cur_res_n].nstrand = la->res[la->cur_res_n - 1].nstrand;
}

/* ---------------------------------------------------------------- merge_cigar, :251-328 */
static int merge_cigar(fctx *f, lo_cigv *c1, int64_t *c1_refend, int *c1_readend, int chr,
                       const lo_cig *_c2, int c2_n, int c2_reflen, int c2_readlen)
{
    if (c2_n == 0) return 0;
    const lo_para *P = f->P;
    int repair = 0;
    if (c1->n > 1) {
        lo_cig t = c1->c[c1->n - 1], h = _c2[0];
        int top = t & 0xf, hop = h & 0xf;
        if ((((top == LO_I || top == LO_D) && (t >> 4) <= 3) && hop != LO_S && hop != LO_H) ||
            (((hop == LO_I || hop == LO_D) && (h >> 4) <= 3) && top != LO_S && top != LO_H)) repair = 1;
    }
    if (!repair) lo_cig_pushv(c1, _c2, c2_n);
    else {
        int len1, len11 = 0, len2, len21 = 0, len22 = 0, len_dif1 = 0, len_dif2 = 0;
        int b = 0, min_b, ci = 0, left = 1, right = 1;
        const int md = 5;
        int64_t ref_start = 0; int read_start = 0;
        lo_cig *c2 = (lo_cig*)malloc(sizeof(lo_cig) * (size_t)c2_n);
        memcpy(c2, _c2, sizeof(lo_cig) * (size_t)c2_n);
        lo_cigv bd; lo_cigv_init(&bd);
        for (;;) {
            if (left) {
                while (c1->n >= 1) {
                    lo_cig w = c1->c[c1->n - 1]; int op = w & 0xf, l = w >> 4;
                    if (op == LO_M && l > md) { c1->c[c1->n - 1] -= md << 4; len21 += md; break; }
                    else if (op == LO_M) { len21 += l; --c1->n; }
                    else if (op == LO_I) { len21 += l; len_dif1 -= l; b += l; --c1->n; }
                    else if (op == LO_D) { len_dif1 += l; b += l; --c1->n; }
                    else { left = -1; break; }
                }
                len11 = len21 + len_dif1;
                read_start = *c1_readend - len21 + 1;
                ref_start = *c1_refend - len11 + 1;
            }
            if (right) {
                while (ci < c2_n) {
                    int op = c2[ci] & 0xf, l = c2[ci] >> 4;
                    if (op == LO_M && l > md) { c2[ci] -= md << 4; len22 += md; break; }
                    else if (op == LO_M) { len22 += l; ci++; }
                    else if (op == LO_I) { len22 += l; len_dif2 -= l; b += l; ++ci; }
                    else if (op == LO_D) { len_dif2 += l; b += l; ++ci; }
                    else { right = -1; break; }
                }
            }
            len2 = len21 + len22; len1 = len2 + len_dif1 + len_dif2;
            min_b = abs(len_dif1 + len_dif2) + md; b = b > min_b ? b : min_b;
            uint8_t *seq1 = (uint8_t*)malloc((size_t)(len1 > 0 ? len1 : 0) + 1), *seq2 = (uint8_t*)malloc((size_t)(len2 > 0 ? len2 : 0) + 1);
            int32_t l1 = len1;
            if (lo_pac_fetch(f->R, chr, ref_start - 1, &l1, seq1) < 0) { free(seq1); free(seq2); free(c2); lo_cigv_free(&bd); return -1; }
            len1 = l1;
            for (int i = 0, j = read_start - 1; j < read_start + len2 - 1; ++i, ++j) seq2[i] = f->read[j];
            lo_ksw_global(len2, seq2, len1, seq1, P->sc_mat, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, b, &bd);
            free(seq1); free(seq2);
            if (bd.n == 0) break;                       /* (the reference would read bd_cigar[0] of an empty CIGAR) */
            if ((bd.c[0] & 0xf) == LO_M) left = 0;
            else if (c1->n == 0 || left < 0) break;
            if ((bd.c[bd.n - 1] & 0xf) == LO_M) right = 0;
            else if (ci == c2_n || right < 0) break;
            if (left + right == 0) break;
        }
        lo_cig_pushv(c1, bd.c, bd.n);
        lo_cig_pushv(c1, c2 + ci, c2_n - ci);
        lo_cigv_free(&bd); free(c2);
    }
    *c1_refend += c2_reflen;
    *c1_readend += c2_readlen;
    return 0;
}

/* get_ref_intv :98 / get_read_intv :116 */
static int ref_gap(fctx *f, lo_xy s1, lo_xy s2, uint8_t **buf, int *cap, int *err)
{
    const lo_hit *h1 = &FH(f, s1), *h2 = &FH(f, s2);
    int64_t start = h1->offset + f->P->seed_len - 1 + h1->len_dif;
    int32_t len = (int32_t)(h2->offset - 1 - start);
    if (len <= 0) return 0;
    if (len > *cap) { *buf = (uint8_t*)realloc(*buf, (size_t)len); *cap = len; }
    if (lo_pac_fetch(f->R, h1->chr, start, &len, *buf) < 0) { *err = 1; return 0; }
    return (int)len;
}
static int read_gap(fctx *f, lo_xy s1, lo_xy s2, uint8_t *dst)
{
    const lo_para *P = f->P;
    int j = 0, i, e;
    if (FH(f, s1).strand == 1) { i = SID(f, s1) * P->seed_step - P->seed_inv; e = (SID(f, s2) - 1) * P->seed_step; }
    else { i = f->S->last_len + SID(f, s1) * P->seed_step - P->seed_inv; e = f->S->last_len + (SID(f, s2) - 1) * P->seed_step; }
    for (; i < e; ++j, ++i) dst[j] = f->read[i];
    return j;
}

/* ---------------------------------------------------------------- frag_extend, :332-410 */
static int frag_extend(fctx *f, const lo_frag *fr, lo_res *res, uint8_t *bseq1, uint8_t **bseq2, int *cap2)
{
    const lo_para *P = f->P;
    lo_cigv fc, g; lo_cigv_init(&fc); lo_cigv_init(&g);
    lo_xy last;
    int i, rs, re, err = 0, rc = 0;
    if (fr->strand == 1) { i = fr->seed_n - 1; last = fr->seed[i]; rs = (SID(f, last) - 1) * P->seed_step + 1; }
    else { i = 0; last = fr->seed[0]; rs = f->S->last_len + (SID(f, last) - 1) * P->seed_step + 1; }
    re = rs - 1 + P->seed_len;
    const lo_hit *hl = &FH(f, last);
    lo_cig_pushv(&fc, f->S->cig + hl->cig_off, hl->cig_n);
    int64_t ref_start = hl->offset, ref_end = hl->offset + P->seed_len - 1 + hl->len_dif;
    const int step = fr->strand == 1 ? -1 : 1;
    for (i += step; i >= 0 && i < fr->seed_n; i += step) {
        lo_xy s = fr->seed[i];
        int len2 = ref_gap(f, last, s, bseq2, cap2, &err);
        if (err) { rc = -1; break; }
        int len1 = read_gap(f, last, s, bseq1);
        lo_ksw_global(len1, bseq1, len2, *bseq2, P->sc_mat, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, P->band_w, &g);
        if (merge_cigar(f, &fc, &ref_end, &re, fr->chr, g.c, g.n, len2, len1) < 0) { rc = -1; break; }
        const lo_hit *hs = &FH(f, s);
        if (merge_cigar(f, &fc, &ref_end, &re, fr->chr, f->S->cig + hs->cig_off, hs->cig_n, P->seed_len + hs->len_dif, P->seed_len) < 0) { rc = -1; break; }
        last = s;
    }
    if (rc == 0) rc = merge_cigar(f, &res->cig, &res->refend, &res->readend, fr->chr, fc.c, fc.n, (int)(ref_end - ref_start + 1), re - rs + 1);
    lo_cigv_free(&fc); lo_cigv_free(&g);
    return rc;
}

/* ---------------------------------------------------------------- split_mapping, :416-564 */
static int split_mapping(fctx *f, const lo_fline *fl, int f1_i, int f2_i, lo_res *res)
{
    const lo_para *P = f->P;
    const lo_frag *fa1 = &fl->frag[f1_i], *fa2 = &fl->frag[f2_i];
    lo_xy s1, s2;
    if (fa1->strand == 1) { s1 = fa1->seed[0]; s2 = fa2->seed[fa2->seed_n - 1]; }
    else { s1 = fa1->seed[fa1->seed_n - 1]; s2 = fa2->seed[0]; }
    const lo_hit at1 = FH(f, s1), at2 = FH(f, s2);
    const int hash_len = P->hash_len, did = SID(f, s2) - SID(f, s1);
    int s_qlen = did * P->seed_step - P->seed_len, s_tlen = 0, rc = 0;
    if (s_qlen < 0) return -1;                                   /* malloc(negative) in the reference */
    uint8_t *s_qseq = (uint8_t*)malloc((size_t)s_qlen + 1), *s_tseq = NULL;
    lo_cigv sc, tmp; lo_cigv_init(&sc); lo_cigv_init(&tmp);
    read_gap(f, s1, s2, s_qseq);
    int64_t exp = at1.offset + at1.len_dif + (int64_t)(did * P->seed_step);
    int dis = (int)(at2.offset - exp);
    int match_dis = P->match_dis * ((P->aln_mode & 2) ? did : 1);
    int64_t ref_offset;
    int32_t tl;
    if (dis > match_dis) {                                       /* DEL, :475-489 */
        s_tlen = s_qlen + dis; tl = s_tlen;
        s_tseq = (uint8_t*)malloc((size_t)(s_tlen > 0 ? s_tlen : 0) + 1);
        ref_offset = at1.offset + P->seed_len + at1.len_dif;
        if (lo_pac_fetch(f->R, at1.chr, ref_offset - 1, &tl, s_tseq) < 0) { rc = -1; goto end; }
        s_tlen = tl;
        if (s_qlen < hash_len) { lo_ksw_bi_extend(s_qlen, s_qseq, s_tlen, s_tseq, hash_len * P->match, hash_len * P->match, P, &tmp); lo_cig_pushv(&sc, tmp.c, tmp.n); }
        else lo_split_indel_map(&sc, s_qseq, s_qlen, s_tseq, s_tlen, 0, P);
    } else if (dis < -match_dis) {                               /* INS, :490-546 */
        s_tlen = s_qlen + dis;
        if (s_tlen < 2 * P->hash_step) {
            int32_t _s_tlen = s_qlen + hash_len;
            int lqe, lte, rqe, rte;
            s_tseq = (uint8_t*)malloc((size_t)_s_tlen + 1);
            lo_cigv lc, rcg; lo_cigv_init(&lc); lo_cigv_init(&rcg);
            ref_offset = at1.offset + P->seed_len + at1.len_dif;
            if (lo_pac_fetch(f->R, at1.chr, ref_offset - 1, &_s_tlen, s_tseq) < 0) { rc = -1; lo_cigv_free(&lc); lo_cigv_free(&rcg); goto end; }
            lo_ksw_extend(s_qlen, s_qseq, _s_tlen, s_tseq, P->sc_mat, P->band_w, hash_len * P->match, P, &lqe, &lte, &lc);
            ref_offset = at2.offset - _s_tlen;
            if (lo_pac_fetch(f->R, at2.chr, ref_offset - 1, &_s_tlen, s_tseq) < 0) { rc = -1; lo_cigv_free(&lc); lo_cigv_free(&rcg); goto end; }
            for (int i = 0; i < s_qlen >> 1; ++i) { uint8_t t = s_qseq[s_qlen-1-i]; s_qseq[s_qlen-1-i] = s_qseq[i]; s_qseq[i] = t; }
            for (int i = 0; i < _s_tlen >> 1; ++i) { uint8_t t = s_tseq[_s_tlen-1-i]; s_tseq[_s_tlen-1-i] = s_tseq[i]; s_tseq[i] = t; }
            lo_ksw_extend(s_qlen, s_qseq, _s_tlen, s_tseq, P->sc_mat, P->band_w, hash_len * P->match, P, &rqe, &rte, &rcg);
            lo_cig_invert(rcg.c, rcg.n);
            /* note: query/target are still reversed here, exactly as in the reference (:527) */
            lo_sw_mid_fix(&sc, lc.c, lc.n, rcg.c, rcg.n, s_qseq, s_qlen, lqe, rqe, s_tseq, s_qlen + dis, lte, rte, P);
            lo_cigv_free(&lc); lo_cigv_free(&rcg);
        } else {                                                 /* DUP, :529-545 */
            s_tlen += 2 * (hash_len - dis); tl = s_tlen;
            s_tseq = (uint8_t*)malloc((size_t)s_tlen + 1);
            ref_offset = at1.offset + P->seed_len + at1.len_dif + dis - hash_len;
            if (lo_pac_fetch(f->R, at1.chr, ref_offset - 1, &tl, s_tseq) < 0) { rc = -1; goto end; }
            s_tlen = tl;
            int off_dis = (s_tlen != s_qlen - dis + 2 * hash_len) ? 0 : -dis;
            s_tlen = s_qlen + dis;
            if (s_tlen < hash_len) { lo_ksw_bi_extend(s_qlen, s_qseq, s_tlen, s_tseq + hash_len - dis, hash_len * P->match, hash_len * P->match, P, &tmp); lo_cig_pushv(&sc, tmp.c, tmp.n); }
            else lo_split_indel_map(&sc, s_qseq, s_qlen, s_tseq + hash_len - dis, s_tlen, off_dis, P);
        }
    } else {                                                     /* mismatch class, :547-559 */
        s_tlen = s_qlen + dis; tl = s_tlen;
        if (s_tlen < 0) { rc = -1; goto end; }                   /* ksw_extend_core exit(-1), ksw.c:672 */
        s_tseq = (uint8_t*)malloc((size_t)s_tlen + 1);
        ref_offset = at1.offset + P->seed_len + at1.len_dif;
        if (lo_pac_fetch(f->R, at1.chr, ref_offset - 1, &tl, s_tseq) < 0) { rc = -1; goto end; }
        s_tlen = tl;
        lo_ksw_bi_extend(s_qlen, s_qseq, s_tlen, s_tseq, 100, 100, P, &tmp);
        lo_cig_pushv(&sc, tmp.c, tmp.n);
    }
    rc = merge_cigar(f, &res->cig, &res->refend, &res->readend, at1.chr, sc.c, sc.n, s_tlen, s_qlen);
end:
    free(s_qseq); free(s_tseq); lo_cigv_free(&sc); lo_cigv_free(&tmp);
    return rc;
}

/* ---------------------------------------------------------------- boundary fixes, :576-707 */
static int head_fix(fctx *f, const lo_fline *fl, lo_res *res)
{
    const lo_para *P = f->P; const lo_seeds *S = f->S;
    const int left_bound = fl->left_bound;
    lo_xy s; int read_len, read_start;
    if (fl->frag[0].strand == 1) {
        const lo_frag *fr = &fl->frag[fl->frag_n - 1];
        s = fr->seed[fr->seed_n - 1];
        if (SID(f, s) != 1) {
            read_len = (left_bound == 0 ? 0 : P->seed_inv) + (SID(f, s) - left_bound - 1) * P->seed_step;
            if (read_len < 0) return -1;
            read_start = left_bound == 0 ? 0 : left_bound * P->seed_step - P->seed_inv;
        } else { res->offset = FH(f, s).offset; res->refend = res->offset - 1; res->cig.n = 0; return 0; }
    } else {
        s = fl->frag[0].seed[0];
        read_len = (left_bound == 0 ? S->last_len : P->seed_inv) + (SID(f, s) - 1 - left_bound) * P->seed_step;
        if (read_len == 0) { res->offset = FH(f, s).offset; res->refend = res->offset - 1; res->cig.n = 0; return 0; }
        if (read_len < 0) return -1;
        read_start = left_bound == 0 ? 0 : S->last_len + left_bound * P->seed_step - P->seed_inv;
    }
    const lo_hit *h = &FH(f, s);
    res->offset = h->offset;
    int32_t ref_len = read_len + P->hash_step * 2;
    int64_t ref_start = h->offset - ref_len;
    if (ref_start < 1) { ref_start = 1; ref_len = (int32_t)(h->offset - 1); }
    uint8_t *b2 = (uint8_t*)calloc((size_t)(ref_len > 0 ? ref_len : 0) + 1, 1);
    if (lo_pac_fetch(f->R, h->chr, ref_start - 1, &ref_len, b2) < 0) { free(b2); return -1; }
    lo_cigv c; lo_cigv_init(&c);
    int qre, tre;
    int r = lo_ksw_extend_r(read_len, f->read + read_start, ref_len, b2, P->sc_mat, P->band_w, P->seed_len * P->match, P, &qre, &tre, &c);
    if (r != 0) lo_cig_push1(&c, ((read_len - qre) << 4) | LO_S);
    lo_cig_invert(c.c, c.n);
    res->offset -= lo_cig_reflen(c.c, c.n);
    res->refend = res->offset - 1;
    lo_cig_pushv(&res->cig, c.c, c.n);                              /* _push_cigar_e, frag_check.h:193 */
    res->refend += lo_cig_reflen(c.c, c.n);
    res->readend += lo_cig_readlen(c.c, c.n);
    lo_cigv_free(&c); free(b2);
    return 0;
}

static int tail_fix(fctx *f, const lo_fline *fl, lo_res *res)
{
    const lo_para *P = f->P; const lo_seeds *S = f->S;
    const int right_bound = fl->right_bound;
    lo_xy s; int read_len, read_start;
    if (fl->frag[0].strand == 1) {
        s = fl->frag[0].seed[0];
        read_start = SID(f, s) * P->seed_step - P->seed_inv;
        read_len = (right_bound == S->seed_all + 1 ? S->last_len : P->seed_inv) + (right_bound - 1 - SID(f, s)) * P->seed_step;
        if (read_len == 0) return 0;
        if (read_len < 0) return -1;
    } else {
        const lo_frag *fr = &fl->frag[fl->frag_n - 1];
        s = fr->seed[fr->seed_n - 1];
        if (SID(f, s) == S->seed_all) return 0;
        read_start = SID(f, s) * P->seed_step - P->seed_inv + S->last_len;
        read_len = (right_bound == S->seed_all + 1 ? 0 : P->seed_inv) + (right_bound - 1 - SID(f, s)) * P->seed_step;
        if (read_len < 0) return -1;
    }
    const lo_hit *h = &FH(f, s);
    int32_t ref_len = read_len + P->hash_step * 2;
    int64_t ref_start = h->offset + P->seed_len + h->len_dif;
    uint8_t *b2 = (uint8_t*)calloc((size_t)ref_len + 1, 1);
    if (lo_pac_fetch(f->R, h->chr, ref_start - 1, &ref_len, b2) < 0) { free(b2); return -1; }
    lo_cigv c; lo_cigv_init(&c);
    int qle, tle;
    int r = lo_ksw_extend_c(read_len, f->read + read_start, ref_len, b2, P->sc_mat, P->band_w, P->seed_len * P->match, P, &qle, &tle, &c);
    if (r != 0) lo_cig_push1(&c, ((read_len - qle) << 4) | LO_S);
    int rc = merge_cigar(f, &res->cig, &res->refend, &res->readend, h->chr, c.c, c.n, lo_cig_reflen(c.c, c.n), lo_cig_readlen(c.c, c.n));
    lo_cigv_free(&c); free(b2);
    return rc;
}

/* ---------------------------------------------------------------- lamsa_res_split, :712-776 */
static int res_split(lo_lres *la, int read_len, const lo_para *P)
{
    int n = la->res[0].cig.n, res_n = 0;
    lo_cig *cg = (lo_cig*)malloc(sizeof(lo_cig) * (size_t)(n + 1));
    memcpy(cg, la->res[0].cig.c, sizeof(lo_cig) * (size_t)n);
    la->res[0].cig.n = 0;
    for (int j = 0; j < n; ++j) {
        int op = cg[j] & 0xf, len = cg[j] >> 4, len1;
#define CUR (&la->res[res_n].cig)
        if (op == LO_M) lo_cig_push1(CUR, cg[j]);
        else if (op == LO_I && len >= P->split_len) {
            len1 = lo_cig_readlen(CUR->c, CUR->n);
            lo_cig_push1(CUR, ((read_len - len1) << 4) | LO_S);
            push_res(la); ++res_n;
            la->res[res_n].offset = la->res[res_n-1].offset + lo_cig_reflen(la->res[res_n-1].cig.c, la->res[res_n-1].cig.n);
            lo_cig_push1(CUR, ((len + len1) << 4) | LO_S);
        } else if (op == LO_D && len >= P->split_len) {
            len1 = lo_cig_readlen(CUR->c, CUR->n);
            lo_cig_push1(CUR, ((read_len - len1) << 4) | LO_S);
            ++res_n; push_res(la);
            la->res[res_n].offset = la->res[res_n-1].offset + lo_cig_reflen(la->res[res_n-1].cig.c, la->res[res_n-1].cig.n) + len;
            lo_cig_push1(CUR, (len1 << 4) | LO_S);
        } else if (op == LO_I || op == LO_D) lo_cig_push1(CUR, cg[j]);
        else if (op == LO_S) {
            if (j > 0 && j < n - 1 && (cg[j+1] & 0xf) == LO_H) {
                int Sn = cg[j] >> 4, Hn = cg[j+1] >> 4;
                len1 = lo_cig_readlen(CUR->c, CUR->n);
                lo_cig_push1(CUR, ((read_len - len1) << 4) | LO_S);
                ++res_n; push_res(la);
                la->res[res_n].offset = la->res[res_n-1].offset + lo_cig_reflen(la->res[res_n-1].cig.c, la->res[res_n-1].cig.n) + Hn;
                lo_cig_push1(CUR, ((len1 + Sn) << 4) | LO_S);
                j += 1;
            } else lo_cig_push1(CUR, cg[j]);
        } else if (op != LO_H) { free(cg); return -1; }
#undef CUR
    }
    free(cg);
    return 0;
}

/* ---------------------------------------------------------------- lamsa_res_aux, :793-853 */
static int res_aux(fctx *f, lo_lres *la)
{
    const lo_para *P = f->P;
    uint8_t *ref = NULL;
    for (int m = 0; m <= la->cur_res_n; ++m) {
        lo_res *r = &la->res[m];
        int32_t ref_len = lo_cig_reflen(r->cig.c, r->cig.n);
        int want = ref_len;
        ref = (uint8_t*)realloc(ref, (size_t)(ref_len > 0 ? ref_len : 0) + 1);
        if (lo_pac_fetch(f->R, r->chr, r->offset - 1, &ref_len, ref) < 0) { free(ref); return -1; }
        (void)want;
        int ref_i = 0, read_i = 0, n_mm = 0, n_m = 0, n_io = 0, n_ie = 0, n_do = 0, n_de = 0;
        for (int i = 0; i < r->cig.n; ++i) {
            int op = r->cig.c[i] & 0xf, len = r->cig.c[i] >> 4;
            if (op == LO_M) {
                int mm = 0;
                for (int j = 0; j < len; ++j) { if (read_i < f->read_len && ref_i < ref_len) { if (f->read[read_i] != ref[ref_i]) ++mm; } else ++mm; ++read_i; ++ref_i; }
                n_m += len - mm; n_mm += mm;
            } else if (op == LO_I) { read_i += len; n_ie += len; ++n_io; }
            else if (op == LO_D) { ref_i += len; n_de += len; ++n_do; }
            else if (op == LO_S) read_i += len;
            else { free(ref); return -1; }
        }
        if (read_i != f->read_len || ref_i != ref_len) { free(ref); return -1; }        /* exit(1), :834-835 */
        r->NM = n_mm + n_ie + n_de;
        r->score = n_m * P->match - n_mm * P->mis - n_io * P->ins_gapo - n_ie * P->ins_gape - n_do * P->del_gapo - n_de * P->del_gape;
        if (r->score < 0) {                                      /* delete the record, :839-844 */
            for (int i = m + 1; i <= la->cur_res_n; ++i) {
                lo_res *t = &la->res[i-1], *s = &la->res[i];
                t->offset = s->offset; t->chr = s->chr; t->nstrand = s->nstrand;
                t->cig.n = 0; lo_cig_pushv(&t->cig, s->cig.c, s->cig.n);
                t->refend = s->refend; t->readend = s->readend; t->score = s->score; t->NM = s->NM;
            }
            m--; la->cur_res_n--;
        } else { la->tol_score += r->score; la->tol_NM += r->NM; }
    }
    if (la->cur_res_n < 0) la->tol_score = -1;
    else la->tol_score -= la->cur_res_n * P->split_pen;
    free(ref);
    return 0;
}

/* ---------------------------------------------------------------- frag_check, :856-961 */
int lo_frag_check(lo_seeds *S, lo_fline *lines, int line_n, lo_ares *a_res, const lo_ref *R,
                  const uint8_t *read, uint8_t **rc_read, const lo_para *P)
{
    const int read_len = S->read_len;
    fctx fc = { S, R, P, read, read_len }, *f = &fc;
    uint8_t *bseq1 = (uint8_t*)malloc((size_t)read_len + 1), *bseq2 = (uint8_t*)malloc((size_t)read_len + 1);
    int cap2 = read_len + 1, rc = 0;
    ares_room(a_res, line_n, P->res_mul_max);
    a_res->l_n = line_n;
    for (int i = 0; i < line_n; ++i) {
        lo_lres *la = &a_res->la[i];
        la->line_score = lines[i].line_score; la->cur_res_n = 0; la->tol_score = la->tol_NM = 0;
        for (int j = 0; j < la->res_m; ++j) la->res[j].cig.n = 0;
    }
    for (int j = 0; j < line_n && rc == 0; ++j) {
        lo_lres *la = &a_res->la[j];
        lo_fline *fl = &lines[j];
        lo_res *r0 = &la->res[0];
        r0->cig.n = 0; r0->nstrand = fl->frag[0].strand == 1 ? 1 : 0; r0->chr = fl->frag[0].chr; r0->readend = 0;
        if (fl->frag[0].strand == 1) {
            f->read = read;
            if (fl->left_bound > 0) { lo_cig w = ((fl->left_bound * P->seed_step - P->seed_inv) << 4) | LO_S; lo_cig_push1(&r0->cig, w); r0->readend += lo_cig_readlen(&w, 1); }
            if ((rc = head_fix(f, fl, r0)) < 0) break;
            int i;
            for (i = fl->frag_n - 1; i > 0 && rc == 0; --i) {
                if ((rc = frag_extend(f, &fl->frag[i], r0, bseq1, &bseq2, &cap2)) < 0) break;
                rc = split_mapping(f, fl, i, i - 1, r0);
            }
            if (rc < 0) break;
            if ((rc = frag_extend(f, &fl->frag[0], r0, bseq1, &bseq2, &cap2)) < 0) break;
            if ((rc = tail_fix(f, fl, r0)) < 0) break;
            if (fl->right_bound <= S->seed_all) { lo_cig w = ((read_len - (fl->right_bound - 1) * P->seed_step) << 4) | LO_S; lo_cig_push1(&r0->cig, w); r0->readend += lo_cig_readlen(&w, 1); }
            if ((rc = res_split(la, read_len, P)) < 0) break;
            rc = res_aux(f, la);
        } else {
            if (*rc_read == NULL) {
                *rc_read = (uint8_t*)calloc((size_t)read_len + 1, 1);
                for (int i = 0; i < read_len; ++i) (*rc_read)[i] = read[read_len-1-i] < 4 ? 3 - read[read_len-1-i] : 4;
            }
            f->read = *rc_read;
            for (int i = 0; i < S->seed_out; ++i) S->seed_id[i] = S->seed_all + 1 - S->seed_id[i];   /* :926 */
            int tmp = fl->left_bound;
            fl->left_bound = S->seed_all + 1 - fl->right_bound; fl->right_bound = S->seed_all + 1 - tmp;
            do {
                if (fl->left_bound > 0) { lo_cig w = ((fl->left_bound * P->seed_step - P->seed_inv + S->last_len) << 4) | LO_S; lo_cig_push1(&r0->cig, w); r0->readend += lo_cig_readlen(&w, 1); }
                if ((rc = head_fix(f, fl, r0)) < 0) break;
                int i;
                for (i = 0; i < fl->frag_n - 1 && rc == 0; ++i) {
                    if ((rc = frag_extend(f, &fl->frag[i], r0, bseq1, &bseq2, &cap2)) < 0) break;
                    rc = split_mapping(f, fl, i, i + 1, r0);
                }
                if (rc < 0) break;
                if ((rc = frag_extend(f, &fl->frag[fl->frag_n - 1], r0, bseq1, &bseq2, &cap2)) < 0) break;
                if ((rc = tail_fix(f, fl, r0)) < 0) break;
                if (fl->right_bound <= S->seed_all) { lo_cig w = (((S->seed_all - fl->right_bound + 1) * P->seed_step - P->seed_inv) << 4) | LO_S; lo_cig_push1(&r0->cig, w); r0->readend += lo_cig_readlen(&w, 1); }
                if ((rc = res_split(la, read_len, P)) < 0) break;
                rc = res_aux(f, la);
            } while (0);
            for (int i = 0; i < S->seed_out; ++i) S->seed_id[i] = S->seed_all + 1 - S->seed_id[i];   /* :953 */
        }
    }
    free(bseq1); free(bseq2);
    return rc;
}
