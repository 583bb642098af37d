/* lo_ksw.c -- banded affine-gap DP primitives (oracle; see lo.h header note).
 *
 * Restates, bit for bit, the live routines of the reference's src/ksw.c:
 *   lo_ksw_global    <- ksw_global2      src/ksw.c:543-653
 *   lo_ksw_extend    <- ksw_extend_core  src/ksw.c:667-807
 *   lo_ksw_extend_c/r<- ksw_extend_c/_r  src/ksw.c:809-836
 *   lo_sw_mid_fix    <- sw_mid_fix       src/ksw.c:841-860
 *   lo_ksw_bi_extend <- ksw_bi_extend    src/ksw.c:862-926
 *
 * Layout differs from the reference (separate H/E rows, no query profile), the
 * arithmetic, tie rules and -- importantly -- the "stale cell" behaviour of the
 * single in-place row do not: cells outside the current band keep whatever an
 * earlier row (or the initial row) left there, exactly like eh[] in the reference.
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include "lo.h"

#define NEG_INF (-0x40000000)            /* MINUS_INF, src/ksw.c:504 */

static void die(const char *msg) { fprintf(stderr, "[lo_ksw] %s\n", msg); exit(1); }

/* traceback shared by both routines (src/ksw.c:638-649, 792-801) */
static void backtrack(const uint8_t *z, int n_col, int w, int i, int k, lo_cigv *out)
{
    int which = 0;
    lo_cigv_clear(out);
    while (i >= 0 && k >= 0) {
        int off = i > w ? i - w : 0;
        which = z[(long)i * n_col + (k - off)] >> (which << 1) & 3;
        if (which == 0) { lo_cig_push0(out, 1 << 4 | LO_M); --i; --k; }
        else if (which == 1) { lo_cig_push0(out, 1 << 4 | LO_D); --i; }
        else { lo_cig_push0(out, 1 << 4 | LO_I); --k; }
    }
    if (i >= 0) lo_cig_push0(out, (i + 1) << 4 | LO_D);
    if (k >= 0) lo_cig_push0(out, (k + 1) << 4 | LO_I);
    lo_cig_invert(out->c, out->n);
}

int lo_ksw_global(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                  const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins,
                  int w, lo_cigv *out)
{
    if (qlen < 0 || tlen < 0) die("global: negative length (reference exits here, src/ksw.c:547)");
    int d = abs(qlen - tlen) + 3;
    if (w < d) w = d;                                        /* :549 */
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;   /* :559 */
    int32_t *H = (int32_t*)malloc(sizeof(int32_t) * (size_t)(qlen + 1));
    int32_t *E = (int32_t*)malloc(sizeof(int32_t) * (size_t)(qlen + 1));
    uint8_t *z = out ? (uint8_t*)malloc((size_t)n_col * tlen + 1) : 0;
    int i, j;
    H[0] = 0; E[0] = NEG_INF;                                /* :569-572 */
    for (j = 1; j <= qlen && j <= w; ++j) { H[j] = -(o_ins + e_ins * j); E[j] = NEG_INF; }
    for (; j <= qlen; ++j) H[j] = E[j] = NEG_INF;
    for (i = 0; i < tlen; ++i) {
        int32_t f = NEG_INF, h1;
        const int8_t *srow = mat + target[i] * 5;
        int beg = i > w ? i - w : 0;
        int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : NEG_INF; /* :579 */
        for (j = beg; j < end; ++j) {
            int32_t m = H[j], e = E[j], h, t;
            uint8_t dir;
            H[j] = h1;
            m += srow[query[j]];
            dir = m >= e ? 0 : 1;  h = m >= e ? m : e;        /* ties: M over E */
            dir = h >= f ? dir : 2; h = h >= f ? h : f;       /*       then over F */
            h1 = h;
            t = m - oe_del; e -= e_del;
            if (e > t) dir |= 1 << 2; else e = t;
            E[j] = e;
            t = m - oe_ins; f -= e_ins;
            if (f > t) dir |= 2 << 4; else f = t;
            if (z) z[(long)i * n_col + (j - beg)] = dir;
        }
        H[end] = h1; E[end] = NEG_INF;                        /* :632 */
    }
    int score = H[qlen];
    if (out) {
        i = tlen - 1;
        int k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;    /* :638 */
        backtrack(z, n_col, w, i, k, out);
    }
    free(H); free(E); free(z);
    return score;
}

int lo_ksw_extend(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                  const int8_t *mat, int w, int h0, const lo_para *P,
                  int *qle, int *tle, lo_cigv *out)
{
    if (qlen < 0 || tlen < 0) die("extend: negative length (reference exits here, src/ksw.c:672)");
    if (h0 <= 0) die("extend: h0 must be positive (assert, src/ksw.c:682)");
    const int o_ins = P->ins_ext_o, e_ins = P->ins_ext_e, o_del = P->del_ext_o, e_del = P->del_ext_e;
    const int end_bonus = P->end_bonus, zdrop = P->zdrop;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int i, j, k, beg, end, max, max_i, max_j, max_ins, max_del, max_ie, gscore;
    int32_t *H = (int32_t*)calloc((size_t)qlen + 2, sizeof(int32_t));
    int32_t *E = (int32_t*)calloc((size_t)qlen + 2, sizeof(int32_t));
    H[0] = h0; H[1] = h0 > oe_ins ? h0 - oe_ins : 0;          /* :692-694 */
    for (j = 2; j <= qlen && H[j-1] > e_ins; ++j) H[j] = H[j-1] - e_ins;
    for (i = 0, max = 0; i < 25; ++i) max = max > mat[i] ? max : mat[i];
    max_ins = (int)((double)(qlen * max + end_bonus - o_ins) / e_ins + 1.);   /* :699-704 */
    max_ins = max_ins > 1 ? max_ins : 1;
    w = w < max_ins ? w : max_ins;
    max_del = (int)((double)(qlen * max + end_bonus - o_del) / e_del + 1.);
    max_del = max_del > 1 ? max_del : 1;
    w = w < max_del ? w : max_del;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    uint8_t *z = (uint8_t*)malloc((size_t)n_col * tlen + 1);
    memset(z, 255, (size_t)n_col * tlen);                     /* :707 */
    max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1;
    beg = 0; end = qlen;
    for (i = 0; i < tlen; ++i) {
        int32_t t, f = 0, h1, m = 0;
        int mj = -1;
        const int8_t *srow = mat + target[i] * 5;
        const int d_beg = i > w ? i - w : 0;
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
        else h1 = 0;
        for (j = beg; j < end; ++j) {
            int32_t M = H[j], e = E[j], h;
            uint8_t dir;
            H[j] = h1;
            M = M ? M + srow[query[j]] : 0;                   /* :737 */
            dir = M > e ? 0 : 1;  h = M > e ? M : e;           /* ties: E over M */
            dir = h > f ? dir : 2; h = h > f ? h : f;          /*       F over both */
            h1 = h;
            mj = m > h ? mj : j;                               /* last j among equals */
            m = m > h ? m : h;
            t = M - oe_del; t = t > 0 ? t : 0; e -= e_del;
            if (e > t) dir |= 1 << 2; else e = t;
            E[j] = e;
            t = M - oe_ins; t = t > 0 ? t : 0; f -= e_ins;
            if (f > t) dir |= 2 << 4; else f = t;
            z[(long)i * n_col + (j - d_beg)] = dir;
        }
        H[end] = h1; E[end] = 0;                               /* :758 */
        if (j == qlen) {                                       /* :759-762 */
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        if (m == 0) break;
        if (m > max) { max = m; max_i = i; max_j = mj; }
        else if (zdrop > 0) {                                  /* :767-773 */
            if (i - max_i > mj - max_j) { if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
            else { if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
        }
        for (j = beg; j < end && H[j] == 0 && E[j] == 0; ++j);  /* :775-778 */
        beg = j;
        for (j = end; j >= beg && H[j] == 0 && E[j] == 0; --j);
        end = j + 2 < qlen ? j + 2 : qlen;
    }
    if (gscore <= 0 || gscore <= max - end_bonus) { i = max_i; k = max_j; }   /* :785-789 */
    else { i = max_ie; k = qlen - 1; }
    if (qle) *qle = k + 1;
    if (tle) *tle = i + 1;
    if (out) backtrack(z, n_col, w, i, k, out);
    free(H); free(E); free(z);
    return max;
}

int lo_ksw_extend_c(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                    const int8_t *mat, int w, int h0, const lo_para *P, int *qle, int *tle, lo_cigv *out)
{
    lo_ksw_extend(qlen, query, tlen, target, mat, w, h0, P, qle, tle, out);
    if (*qle == qlen) return 0;
    if (*tle == tlen) return 1;
    return 2;
}

int lo_ksw_extend_r(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                    const int8_t *mat, int w, int h0, const lo_para *P, int *qre, int *tre, lo_cigv *out)
{
    if (qlen < 0 || tlen < 0) die("extend_r: negative length");
    uint8_t *rq = (uint8_t*)malloc((size_t)qlen + 1), *rt = (uint8_t*)malloc((size_t)tlen + 1);
    for (int i = 0; i < qlen; ++i) rq[qlen-1-i] = query[i];
    for (int i = 0; i < tlen; ++i) rt[tlen-1-i] = target[i];
    lo_ksw_extend(qlen, rq, tlen, rt, mat, w, h0, P, qre, tre, out);
    free(rq); free(rt);
    if (*qre == qlen) return 0;
    if (*tre == tlen) return 1;
    return 2;
}

void lo_sw_mid_fix(lo_cigv *out, const lo_cig *lc, int ln, const lo_cig *rc, int rn,
                   const uint8_t *query, int qlen, int lqe, int rqe,
                   const uint8_t *target, int tlen, int lte, int rte, const lo_para *P)
{
    int Sn = qlen - lqe - rqe, Hn = tlen - lte - rte, half = P->split_len / 2;
    if (abs(Sn) >= half || abs(Hn) >= half || abs(Sn - Hn) >= half || tlen < 0 || qlen < 0) {
        lo_cig_pushv(out, lc, ln);
        lo_cig_push0(out, (Sn << 4) | LO_S);     /* may be negative or zero; kept as is */
        lo_cig_push0(out, (Hn << 4) | LO_H);
        lo_cig_pushv(out, rc, rn);
    } else {
        lo_cigv g; lo_cigv_init(&g);
        lo_ksw_global(qlen, query, tlen, target, P->sc_mat, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, P->band_w, &g);
        lo_cig_pushv(out, g.c, g.n);
        lo_cigv_free(&g);
    }
}

/* the float expression of src/ksw.c:881,900 evaluated in single precision, no contraction */
static int near_diag(int qlen, int tlen, const lo_para *P)
{
    volatile float a = (float)tlen * P->id_rate;
    volatile float b = a * (float)(P->aln_mode & 2);
    volatile float c = (float)P->split_len + b;
    return (float)abs(qlen - tlen) < c;
}

int lo_ksw_bi_extend(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                     int lh0, int rh0, const lo_para *P, lo_cigv *out)
{
    int res, lqe, lte, rqe, rte;
    lo_cigv L, R; lo_cigv_init(&L); lo_cigv_init(&R);
    lo_cigv_clear(out);
    int w = abs(qlen - tlen) + 3 > P->band_w ? abs(qlen - tlen) + 3 : P->band_w;     /* :873 */
    res = lo_ksw_extend_c(qlen, query, tlen, target, P->sc_mat, w, lh0, P, &lqe, &lte, &L);
    if (res < 2) {                                                                    /* :875-880 */
        lo_cig_pushv(out, L.c, L.n);
        lo_cig_push1(out, res == 0 ? ((tlen - lte) << 4) | LO_D : ((qlen - lqe) << 4) | LO_I);
        goto done0;
    } else if (near_diag(qlen, tlen, P) && ((lqe << 1 > qlen) || (lte << 1 > tlen))) { /* :881-887 */
        lo_ksw_global(qlen, query, tlen, target, P->sc_mat, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, P->band_w, out);
        goto done0;
    }
    res = lo_ksw_extend_r(qlen, query, tlen, target, P->sc_mat, w, rh0, P, &rqe, &rte, &R);
    if (res < 2) {                                                                    /* :892-899 */
        lo_cig_push1(&R, res == 0 ? ((tlen - rte) << 4) | LO_D : ((qlen - rqe) << 4) | LO_I);
        lo_cig_invert(R.c, R.n);
        lo_cig_pushv(out, R.c, R.n);
        goto done0;
    } else if (near_diag(qlen, tlen, P) && ((rqe << 1 > qlen) || (rte << 1 > tlen))) { /* :900-906 */
        lo_ksw_global(qlen, query, tlen, target, P->sc_mat, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, P->band_w, out);
        goto done0;
    }
    lo_cig_invert(R.c, R.n);
    lo_sw_mid_fix(out, L.c, L.n, R.c, R.n, query, qlen, lqe, rqe, target, tlen, lte, rte, P);
    lo_cigv_free(&L); lo_cigv_free(&R);
    return (qlen - lqe - rqe) >= P->split_len ? 1 : 0;                                 /* :924 */
done0:
    lo_cigv_free(&L); lo_cigv_free(&R);
    return 0;
}
