/* lo_reg.c -- covered read intervals, uncovered regions, result ranking (oracle; see lo.h).
 * Restates src/lamsa_aln.c:454-724 (aln_reg helpers, get_remain_reg, push_reg_res, get_reg,
 * get_cov_f, get_cover_res, rearr_aln_res).  Sorts are stable (glibc merge-sort goldens). */
#include <stdlib.h>
#include <string.h>
#include "lo_read.h"

static void reg_alloc(lo_reg *r)
{
    r->beg_n = r->end_n = 0; r->beg_m = r->end_m = 10;
    r->ref_beg = (lo_regb*)malloc(10 * sizeof(lo_regb)); r->ref_end = (lo_regb*)malloc(10 * sizeof(lo_regb));
}
lo_areg *lo_areg_new(int read_len)
{
    lo_areg *a = (lo_areg*)malloc(sizeof(lo_areg));
    a->reg_n = 0; a->reg_m = 1; a->read_len = read_len;
    a->reg = (lo_reg*)malloc(sizeof(lo_reg));
    reg_alloc(a->reg);
    return a;
}
void lo_areg_free(lo_areg *a)
{
    for (int i = 0; i < a->reg_m; ++i) { free(a->reg[i].ref_beg); free(a->reg[i].ref_end); }
    free(a->reg); free(a);
}
static void areg_room(lo_areg *a)
{
    if (a->reg_n < a->reg_m) return;
    int m = a->reg_m << 1;
    a->reg = (lo_reg*)realloc(a->reg, sizeof(lo_reg) * (size_t)m);
    for (int i = a->reg_m; i < m; ++i) reg_alloc(&a->reg[i]);
    a->reg_m = m;
}
static void push_b(lo_reg *r, int beg_n, const lo_regb *beg, int end_n, const lo_regb *end)
{   /* push_reg_b, :479 */
    for (int i = 0; i < beg_n; ++i) {
        if (r->beg_n == r->beg_m) { r->beg_m <<= 1; r->ref_beg = (lo_regb*)realloc(r->ref_beg, sizeof(lo_regb) * (size_t)r->beg_m); }
        r->ref_beg[r->beg_n++] = beg[i];
    }
    for (int i = 0; i < end_n; ++i) {
        if (r->end_n == r->end_m) { r->end_m <<= 1; r->ref_end = (lo_regb*)realloc(r->ref_end, sizeof(lo_regb) * (size_t)r->end_m); }
        r->ref_end[r->end_n++] = end[i];
    }
}
static void push_reg(lo_areg *a, int beg, int end, int beg_n, const lo_regb *rb, int end_n, const lo_regb *re)
{   /* :531 */
    areg_room(a);
    lo_reg *r = &a->reg[a->reg_n];
    r->beg = beg; r->end = end; r->beg_n = r->end_n = 0;
    push_b(r, beg_n, rb, end_n, re);
    a->reg_n++;
}
static void sort_reg(lo_areg *a)
{   /* aln_sort_reg, :477: by beg ascending, stable */
    for (int i = 1; i < a->reg_n; ++i) {
        lo_reg t = a->reg[i]; int k = i - 1;
        while (k >= 0 && a->reg[k].beg > t.beg) { a->reg[k + 1] = a->reg[k]; --k; }
        a->reg[k + 1] = t;
    }
}
static void merge_reg(lo_areg *a, int thd)
{   /* aln_merg_reg, :499 */
    int cur = 0;
    for (int i = 1; i < a->reg_n; ++i) {
        if (a->reg[i].beg - a->reg[cur].end - 1 < thd) {
            if (a->reg[i].end > a->reg[cur].end) a->reg[cur].end = a->reg[i].end;
            push_b(&a->reg[cur], a->reg[i].beg_n, a->reg[i].ref_beg, a->reg[i].end_n, a->reg[i].ref_end);
        } else {
            cur++;
            if (cur != i) {
                a->reg[cur].beg = a->reg[i].beg; a->reg[cur].end = a->reg[i].end;
                a->reg[cur].beg_n = a->reg[cur].end_n = 0;
                push_b(&a->reg[cur], a->reg[i].beg_n, a->reg[i].ref_beg, a->reg[i].end_n, a->reg[i].ref_end);
            }
        }
    }
    a->reg_n = cur + 1;
}

int lo_get_remain_reg(lo_areg *a, lo_areg *re, const lo_para *P, int min_thd, int max_thd)
{   /* :550-569 */
    if (a->reg_n == 0) {
        if (min_thd < a->read_len && a->read_len <= max_thd) { push_reg(re, 1, a->read_len, 0, 0, 0, 0); return 1; }
        return 0;
    }
    sort_reg(a); merge_reg(a, P->bwt_seed_len);
    int i;
    if (a->reg[0].beg > min_thd && a->reg[0].beg - 1 <= max_thd)
        push_reg(re, 1, a->reg[0].beg - 1, 0, 0, a->reg[0].beg_n, a->reg[0].ref_beg);
    for (i = 1; i < a->reg_n; ++i)
        if (a->reg[i].beg - a->reg[i-1].end > min_thd && a->reg[i].beg - 1 - a->reg[i-1].end <= max_thd)
            push_reg(re, a->reg[i-1].end + 1, a->reg[i].beg - 1, a->reg[i-1].end_n, a->reg[i-1].ref_end, a->reg[i].beg_n, a->reg[i].ref_beg);
    if (a->read_len - a->reg[i-1].end > min_thd && a->read_len - a->reg[i-1].end <= max_thd)
        push_reg(re, a->reg[i-1].end + 1, a->read_len, a->reg[i-1].end_n, a->reg[i-1].ref_end, 0, 0);
    return re->reg_n;
}

static void push_reg_res(lo_areg *a, lo_res *r)
{   /* :571-595 */
    areg_room(a);
    lo_reg *g = &a->reg[a->reg_n];
    const lo_cig *c = r->cig.c; const int n = r->cig.n;
    g->ref_beg[0].chr = g->ref_end[0].chr = r->chr;
    g->ref_beg[0].is_rev = g->ref_end[0].is_rev = 1 - r->nstrand;
    if (r->nstrand == 1) {
        g->beg = (c[0] & 0xf) == LO_S ? (c[0] >> 4) + 1 : 1;
        g->end = (c[n-1] & 0xf) == LO_S ? a->read_len - (c[n-1] >> 4) : a->read_len;
        g->ref_beg[0].ref_pos = r->offset;
        g->ref_end[0].ref_pos = r->offset + lo_cig_reflen(c, n) - 1;
    } else {
        g->beg = (c[n-1] & 0xf) == LO_S ? (c[n-1] >> 4) + 1 : 1;
        g->end = (c[0] & 0xf) == LO_S ? a->read_len - (c[0] >> 4) : a->read_len;
        g->ref_end[0].ref_pos = r->offset;
        g->ref_beg[0].ref_pos = r->offset + lo_cig_reflen(c, n) - 1;
    }
    r->reg_beg = g->beg; r->reg_end = g->end;
    g->beg_n = g->end_n = 1;
    a->reg_n++;
}

void lo_get_reg(lo_ares *res, lo_areg *reg)
{   /* :597-605 */
    for (int i = 0; i < res->l_n; ++i) {
        if (res->la[i].tol_score < 0) continue;
        for (int j = 0; j <= res->la[i].cur_res_n; ++j) push_reg_res(reg, &res->la[i].res[j]);
    }
}

float lo_get_cov_f(lo_ares *res3, lo_areg *reg)
{   /* :639-651 */
    int cov = 0;
    reg->reg_n = 0;
    for (int i = 0; i < 3; ++i) lo_get_reg(res3 + i, reg);
    sort_reg(reg); merge_reg(reg, 0);
    for (int i = 0; i < reg->reg_n; ++i) cov += reg->reg[i].end - reg->reg[i].beg + 1;
    return (float)((cov + 0.0) / reg->read_len);
}

static float cover_rate(int s1, int e1, int s2, int e2)
{   /* lamsa_dp_con.c:61-67 */
    int s = s2 > s1 ? s2 : s1, e = e2 < e1 ? e2 : e1;
    float rat1 = (float)((e - s + 1 + 0.0) / (e1 - s1 + 1 + 0.0));
    float rat2 = (float)((e - s + 1 + 0.0) / (e2 - s2 + 1 + 0.0));
    return rat1 > rat2 ? rat1 : rat2;
}

typedef struct { int x, y, a, b; } qua_t;

static int covered_by(lo_areg *reg, lo_ares *res, int qi, int *cov_qi, qua_t *qua, const int *head, int head_n, float ovlp_r)
{   /* get_cover_res, :607-629 */
    lo_lres *nl = &res[qua[qi].x].la[qua[qi].y];
    for (int r = 0; r <= nl->cur_res_n; ++r) {
        lo_res *nr = &nl->res[r];
        int reg_i = 0;
        for (int i = 0; i < head_n; ++i) {
            lo_lres *hl = &res[qua[head[i]].x].la[qua[head[i]].y];
            for (int j = 0; j <= hl->cur_res_n; ++j) {
                if (cover_rate(reg->reg[reg_i].beg, reg->reg[reg_i].end, nr->reg_beg, nr->reg_end) >= ovlp_r) { *cov_qi = head[i]; return 1; }
                reg_i++;
            }
        }
    }
    return 0;
}

void lo_rearr(lo_ares *res, int n, float ovlp_r)
{   /* rearr_aln_res, :654-724 */
    int qua_n = 0, qua_m = 16;
    qua_t *qua = (qua_t*)malloc(sizeof(qua_t) * (size_t)qua_m);
    for (int a = 0; a < n; ++a)
        for (int i = 0; i < res[a].l_n; ++i) {
            lo_lres *l = &res[a].la[i];
            if (l->tol_score < 0) { l->merg_x = 0; l->merg_y = -1; continue; }
            if (qua_n == qua_m) { qua_m <<= 1; qua = (qua_t*)realloc(qua, sizeof(qua_t) * (size_t)qua_m); }
            qua[qua_n].x = a; qua[qua_n].y = i; qua[qua_n].a = l->tol_score; qua[qua_n].b = l->line_score; qua_n++;
        }
    if (qua_n == 0) { free(qua); return; }
    for (int i = 1; i < qua_n; ++i) {                 /* res_comp, :632: (a, b) descending, stable */
        qua_t t = qua[i]; int k = i - 1;
        while (k >= 0 && (qua[k].a < t.a || (qua[k].a == t.a && qua[k].b < t.b))) { qua[k + 1] = qua[k]; --k; }
        qua[k + 1] = t;
    }
    int *head = (int*)malloc(sizeof(int) * (size_t)qua_n), head_n = 0;
    lo_areg *reg = lo_areg_new(res->read_len);
#define LQ(i) (res[qua[i].x].la[qua[i].y])
    for (int i = 0; i <= LQ(0).cur_res_n; ++i) push_reg_res(reg, &LQ(0).res[i]);
    head[head_n++] = 0;
    LQ(0).merg_x = 1; LQ(0).merg_y = 0;
    for (int i = 0; i < qua_n; ++i) LQ(i).mapQ = 255;
    int mapq_max = (int)(254 * res->cov_f);            /* MAPQ_MAX * cov_f, float product truncated, :686 */
    int cov_qi = 0;
    for (int i = 1; i < qua_n; ++i) {
        if (!covered_by(reg, res, i, &cov_qi, qua, head, head_n, ovlp_r)) {
            for (int j = 0; j <= LQ(i).cur_res_n; ++j) push_reg_res(reg, &LQ(i).res[j]);
            head[head_n++] = i;
            LQ(i).merg_x = 1; LQ(i).merg_y = 0;
        } else if (qua[i].a > qua[cov_qi].a / 2 && LQ(cov_qi).XA_n + LQ(i).cur_res_n < LQ(cov_qi).XA_m) {
            for (int j = 0; j <= LQ(i).cur_res_n; ++j) {
                lo_lres *h = &LQ(cov_qi);
                h->XA_stage[h->XA_n] = qua[i].x; h->XA_line[h->XA_n] = qua[i].y; h->XA_res[h->XA_n] = j; h->XA_n++;
            }
            LQ(cov_qi).merg_y = 1;
            uint8_t tmpQ = (uint8_t)(mapq_max * (qua[cov_qi].a - qua[i].a) / qua[cov_qi].a);   /* :705 */
            if (tmpQ < LQ(cov_qi).mapQ) LQ(cov_qi).mapQ = tmpQ;
            LQ(i).merg_x = 2; LQ(i).merg_y = 0;
        } else { LQ(i).merg_x = 0; LQ(i).merg_y = -1; }
    }
    for (int i = 0; i < qua_n; ++i) {                  /* :717-722 */
        if (LQ(i).merg_x != 1) continue;
        if (LQ(i).mapQ == 255) LQ(i).mapQ = (uint8_t)(mapq_max / head_n);
        else LQ(i).mapQ /= head_n;
    }
#undef LQ
    free(qua); free(head); lo_areg_free(reg);
}
