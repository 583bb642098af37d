/* lo_chain.c -- sparse-DP chaining of seed hits into lines ("skeletons") (oracle; see lo.h).
 *
 * Restates the live part of the reference's src/lamsa_dp_con.c and src/lamsa_heap.c:
 *   lo_edge_flag     <- get_fseed_dis              lamsa_dp_con.c:596
 *   dp_update        <- frag_dp_update             :701      (order dependent, son_flag side effect)
 *   track / cut      <- branch_track_new, cut_branch, get_max_son   :873,:831,:808
 *   mini_line        <- frag_mini_dp_line          :1068
 *   multi_line       <- frag_mini_dp_multi_line    :923
 *   set_bound        <- line_set_bound / line_set_bound1 (:425,:496) minus the extended
 *                       bounds E_LB/E_RB, which nothing ever reads
 *   build_flines     <- frag_dp_path               :1152 (+ line_filter_overlap :568)
 *   lo_chain_first   <- frag_line_BCC              :1305
 *   lo_chain_remain  <- frag_line_remain           :1252
 * Sorting uses a stable insertion sort: the goldens were produced with glibc 2.35 qsort
 * (merge sort, stable) -- SURVEY.md section 7.3 h2.
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include "lo_read.h"

static const int SCORE_TABLE[10] = { 1, 1, 1, 1, -3, -3, -3, -3, -6, -6 };   /* f_BCC_score_table, lamsa_aln.c:177 */
static const lo_xy START = { -1, 0 };

/* merge-flag bits of a line, lamsa_aln.h:132-136 */
enum { L_MERGB = 0, L_NMERG = 1, L_MERGH = 2, L_INTER = 4, L_DUMP = 8 };

typedef struct { int start, len; int lb, rb, mf, mh, ls, bs, nm; } line_t;
typedef struct { lo_xy n1, n2; } trig_t;
typedef struct { lo_xy *node; int *score, *NM; int min_score_thd, max_n, node_n, cap; } nscore_t;

typedef struct {
    lo_seeds *S; const lo_para *P; lo_node *nodes;
} cctx;

#define MAPN(c, x)   ((c)->S->hit_off[(x) + 1] - (c)->S->hit_off[x])
#define HIT(c, x, y) ((c)->S->hit[(c)->S->hit_off[x] + (y)])
#define ND(c, x, y)  ((c)->nodes[(c)->S->hit_off[x] + (y)])
#define NDX(c, n)    ND(c, (n).x, (n).y)

static void bug(const char *m) { fprintf(stderr, "[lo_chain] BUG: %s\n", m); exit(1); }

/* ---------------------------------------------------------------- edge classification */
int lo_edge_flag(const lo_seeds *S, const lo_para *P, int pre, int pre_a, int i, int j)
{
    if (pre == -1 || i == -1) return LO_F_MATCH;
    if (pre == i) return pre_a == j ? LO_F_MATCH : LO_F_UNCONNECT;
    const lo_hit *hp = &S->hit[S->hit_off[pre] + pre_a], *hc = &S->hit[S->hit_off[i] + j];
    if (hc->chr != hp->chr || hc->strand != hp->strand) return LO_F_CHR_DIF;
    const int idp = S->seed_id[pre], idc = S->seed_id[i], did = abs(idp - idc);
    if (did * P->seed_step < P->seed_len) return LO_F_UNCONNECT;                 /* overlapping seeds, :613 */
    int64_t exp = hp->offset + (int64_t)(hp->strand * (idc - idp) * P->seed_step);
    int64_t act = hc->offset;
    int dis = (int)((int64_t)hp->strand * ((idp < idc) ? (act - exp) : (exp - act))
                    - ((hp->strand * (idp - idc) < 0) ? hp->len_dif : hc->len_dif));     /* :619 */
    int mat_dis = P->match_dis * ((P->aln_mode & 2) ? did : 1);
    if (dis <= mat_dis && dis >= -mat_dis) {
        if (did == 1) return LO_F_MATCH;
        if (did <= 3 * P->mismatch_thd) return LO_F_MISMATCH;
        return LO_F_LONG_MISMATCH;
    }
    if (dis > mat_dis && dis < P->SV_len_thd) return LO_F_DELETE;
    if ((dis < -mat_dis && dis >= 0 - (did * P->seed_step - P->seed_len)) ||
        (dis < -(P->split_len / 2) && dis >= -P->SV_len_thd)) return LO_F_INSERT;
    return LO_F_UNCONNECT;
}

/* ---------------------------------------------------------------- node helpers */
static void node_set(cctx *c, int x, int y, lo_xy from, int score, int NM, int match_flag, int dp_flag)
{   /* fnode_set, :636 */
    lo_node *n = &ND(c, x, y);
    n->son_flag = LO_F_INIT; n->from = from; n->score = score; n->tol_NM = NM;
    n->match_flag = match_flag; n->dp_flag = dp_flag;
    n->node_n = 1; n->in_de = 0; n->son_n = 0;
    n->max_score = score; n->max_NM = NM; n->max_node.x = x; n->max_node.y = y;
}

static void node_per_init(cctx *c, int x, int y, lo_xy from, int dp_flag)
{   /* frag_dp_per_init, :766 */
    if (from.x == -1) { node_set(c, x, y, from, 1, HIT(c, x, y).NM, LO_F_MATCH, dp_flag); return; }
    int flag = lo_edge_flag(c->S, c->P, from.x, from.y, x, y);
    if (flag != LO_F_UNCONNECT && flag != LO_F_CHR_DIF)
        node_set(c, x, y, from, 2 + SCORE_TABLE[flag], HIT(c, x, y).NM + HIT(c, from.x, from.y).NM, flag, dp_flag);
    else ND(c, x, y).dp_flag = 0 - dp_flag;
}

static void add_son(cctx *c, lo_xy fa, lo_xy son)
{   /* fnode_add_son, :683 */
    lo_node *f = &NDX(c, fa);
    ++f->in_de;
    if (f->son_n == f->son_m) { f->son_m = f->son_m ? f->son_m << 1 : 4; f->son = (lo_xy*)realloc(f->son, sizeof(lo_xy) * (size_t)f->son_m); }
    f->son[f->son_n++] = son;
}

static void dp_update(cctx *c, int x, int y, int start, int dp_flag)
{   /* frag_dp_update, :701-764 */
    lo_node *t = &ND(c, x, y);
    lo_xy max_from = t->from;
    int max_score = t->score, max_NM = t->tol_NM, max_flag = t->dp_flag;
    for (int i = x - 1; i >= start; --i) {
        int n = MAPN(c, i);
        for (int j = 0; j < n; ++j) {
            lo_node *p = &ND(c, i, j);
            if (p->dp_flag != dp_flag) continue;
            if (HIT(c, i, j).strand == 1 && p->son_flag <= LO_F_MATCH_THD) continue;         /* '+': already has a match son */
            int flag = lo_edge_flag(c->S, c->P, i, j, x, y);
            if (flag == LO_F_UNCONNECT || flag == LO_F_CHR_DIF) continue;
            int cand = p->score + 1 + SCORE_TABLE[flag];
            if (HIT(c, i, j).strand == -1 && flag <= LO_F_MATCH_THD) {                         /* '-': first match precursor wins */
                max_from.x = i; max_from.y = j; max_score = cand; max_flag = flag; max_NM = p->tol_NM + t->tol_NM;
                goto UPDATE;
            }
            if (cand > max_score || (cand == max_score && t->tol_NM + p->tol_NM < max_NM)) {
                max_from.x = i; max_from.y = j; max_score = cand; max_flag = flag; max_NM = p->tol_NM + t->tol_NM;
            }
        }
    }
UPDATE:
    if (max_from.x != t->from.x || max_from.y != t->from.y) {
        NDX(c, max_from).son_flag = max_flag;
        t->from = max_from; t->score = max_score; t->tol_NM = max_NM; t->match_flag = max_flag;
        t->node_n = NDX(c, max_from).node_n + 1;
        lo_xy me = { x, y };
        add_son(c, max_from, me);
    }
}

/* ---------------------------------------------------------------- end-node stack / heaps (lamsa_heap.c) */
static nscore_t *ns_new(int max_n)
{
    nscore_t *ns = (nscore_t*)calloc(1, sizeof(nscore_t));
    ns->max_n = max_n; ns->cap = 16;
    ns->node = (lo_xy*)malloc(sizeof(lo_xy) * 16); ns->score = (int*)malloc(sizeof(int) * 16); ns->NM = (int*)malloc(sizeof(int) * 16);
    return ns;
}
static void ns_free(nscore_t *ns) { free(ns->node); free(ns->score); free(ns->NM); free(ns); }
static void ns_room(nscore_t *ns)
{
    if (ns->node_n < ns->cap) return;
    ns->cap <<= 1;
    ns->node = (lo_xy*)realloc(ns->node, sizeof(lo_xy) * (size_t)ns->cap);
    ns->score = (int*)realloc(ns->score, sizeof(int) * (size_t)ns->cap);
    ns->NM = (int*)realloc(ns->NM, sizeof(int) * (size_t)ns->cap);
}
static lo_xy ns_pop(nscore_t *ns, int *score, int *NM)
{   /* node_pop: LIFO, lamsa_heap.c:5 */
    if (ns->node_n < 1) return START;
    lo_xy n = ns->node[--ns->node_n];
    *score = ns->score[ns->node_n]; *NM = ns->NM[ns->node_n];
    return n;
}
static void ns_swap(nscore_t *ns, int a, int b)
{
    lo_xy t = ns->node[a]; ns->node[a] = ns->node[b]; ns->node[b] = t;
    int s = ns->score[a]; ns->score[a] = ns->score[b]; ns->score[b] = s;
    s = ns->NM[a]; ns->NM[a] = ns->NM[b]; ns->NM[b] = s;
}
static void ns_min_sift(nscore_t *ns, int i)
{   /* node_min_heap: score ascending, NM descending, lamsa_heap.c:151 */
    for (;;) {
        int l = 2 * i + 1, r = 2 * (i + 1), m = i;
        if (l < ns->node_n && (ns->score[l] < ns->score[i] || (ns->score[l] == ns->score[i] && ns->NM[l] > ns->NM[i]))) m = l;
        if (r < ns->node_n && (ns->score[r] < ns->score[m] || (ns->score[r] == ns->score[m] && ns->NM[r] > ns->NM[m]))) m = r;
        if (m == i) return;
        ns_swap(ns, i, m); i = m;
    }
}
static void ns_minpos_sift(nscore_t *ns, int i)
{   /* node_minpos_heap: node.x ascending, lamsa_heap.c:100 */
    for (;;) {
        int l = 2 * i + 1, r = 2 * (i + 1), m = i;
        if (l < ns->node_n && ns->node[l].x < ns->node[i].x) m = l;
        if (r < ns->node_n && ns->node[r].x < ns->node[m].x) m = r;
        if (m == i) return;
        ns_swap(ns, i, m); i = m;
    }
}
static lo_xy ns_extract_minpos(nscore_t *ns)
{   /* lamsa_heap.c:126 */
    if (ns->node_n < 1) return START;
    lo_xy m = ns->node[0];
    --ns->node_n;
    ns->node[0] = ns->node[ns->node_n]; ns->score[0] = ns->score[ns->node_n]; ns->NM[0] = ns->NM[ns->node_n];
    ns_minpos_sift(ns, 0);
    return m;
}
/* heap_add_node, lamsa_dp_con.c:44: returns -1 stored, -2 rejected, else the line index that was evicted */
static int ns_add_bounded(nscore_t *ns, lo_xy node, int score, int NM)
{
    if (ns->node_n < ns->max_n) {
        ns_room(ns);
        ns->score[ns->node_n] = score; ns->NM[ns->node_n] = NM; ns->node[ns->node_n++] = node;
        if (ns->node_n == ns->max_n) for (int i = (ns->node_n - 1) / 2; i >= 0; --i) ns_min_sift(ns, i);
        return -1;
    }
    if (ns->score[0] < score || (ns->score[0] == score && ns->NM[0] > NM)) {    /* node_heap_update_min, lamsa_heap.c:190 */
        int ret = ns->node[0].x;
        ns->score[0] = score; ns->NM[0] = NM; ns->node[0] = node;
        ns_min_sift(ns, 0);
        return ret;
    }
    return -2;
}

static void ns_add_end(cctx *c, nscore_t *ns, int score, int NM, lo_xy node)
{   /* node_add_score, lamsa_dp_con.c:786: push and mark the whole path TRACKED */
    if (score < ns->min_score_thd) return;
    ns_room(ns);
    ns->score[ns->node_n] = score; ns->NM[ns->node_n] = NM; ns->node[ns->node_n++] = node;
    NDX(c, node).dp_flag = LO_TRACKED_FLAG;
    for (lo_xy t = NDX(c, node).from; t.x != -1; t = NDX(c, t).from) NDX(c, t).dp_flag = LO_TRACKED_FLAG;
}

/* ---------------------------------------------------------------- forest -> disjoint paths */
static lo_xy best_son(cctx *c, int x, int y)
{   /* get_max_son, :808 */
    lo_node *f = &ND(c, x, y);
    int max_score = 0, max_NM = 0, max_dis = 0, flag_thd = LO_F_INIT;
    lo_xy max = { -1, 0 };
    for (int i = 0; i < f->son_n; ++i) {
        lo_xy s = f->son[i]; lo_node *sn = &NDX(c, s);
        if (sn->match_flag <= flag_thd && (sn->max_score > max_score || (sn->max_score == max_score && (s.x - x < max_dis || sn->max_NM < max_NM)))) {
            max = s; max_score = sn->max_score; max_NM = sn->max_NM; max_dis = s.x - x;
            if (sn->match_flag <= LO_F_MATCH_THD) flag_thd = LO_F_MATCH_THD;
        }
    }
    if (max.x < 0) bug("best_son: no son selected");
    return max;
}

static void detach(cctx *c, lo_xy s, lo_xy max_node, nscore_t *ns)
{   /* make s the root of its own path; scores lose the cut prefix (:842-847, :851-857, :893-899) */
    lo_node *sn = &NDX(c, s);
    sn->from = START;
    sn->max_score -= (sn->score - 1);
    sn->max_NM -= (sn->tol_NM - HIT(c, s.x, s.y).NM);
    NDX(c, max_node).node_n -= (sn->node_n - 1);
    ns_add_end(c, ns, sn->max_score, sn->max_NM, max_node);
}

static void cut_branch(cctx *c, int x, int y, nscore_t *ns)
{   /* :831-870 */
    lo_node *f = &ND(c, x, y);
    lo_xy keep = best_son(c, x, y);
    for (int i = 0; i < f->son_n; ++i) {
        lo_xy s = f->son[i];
        if (s.x == keep.x && s.y == keep.y) continue;
        detach(c, s, NDX(c, s).max_node, ns);
    }
    lo_node *k = &NDX(c, keep);
    if (f->score > k->max_score) {                    /* negative edge */
        k->in_de = -1;
        detach(c, keep, k->max_node, ns);
        f->son_n = 0; f->max_node.x = x; f->max_node.y = y; f->max_score = f->score; f->max_NM = f->tol_NM;
    } else {
        f->son_n = 1; f->son[0] = keep;
        f->max_node = k->max_node; f->max_score = k->max_score; f->max_NM = k->max_NM;
    }
    f->in_de = 0;
}

static void branch_track(cctx *c, int x, int y, nscore_t *ns)
{   /* branch_track_new, :873-920 */
    lo_node *n = &ND(c, x, y);
    int max_score, max_NM; lo_xy max_node;
    n->in_de = -1;
    if (n->son_n == 0) { max_node.x = x; max_node.y = y; n->max_node = max_node; max_score = n->max_score = n->score; max_NM = n->max_NM = n->tol_NM; }
    else { max_node = n->max_node; max_score = n->max_score; max_NM = n->max_NM; }
    lo_xy fa = n->from;
    while (fa.x != -1) {
        lo_node *f = &NDX(c, fa);
        if (f->son_n == 1) {
            if (f->score > max_score) {               /* negative edge */
                lo_xy s = f->son[0];
                NDX(c, s).in_de = -1;
                detach(c, s, max_node, ns);
                f->son_n = 0;
                max_score = f->score; max_NM = f->tol_NM; max_node = fa;
            }
            f->max_score = max_score; f->max_NM = max_NM; f->max_node = max_node; f->in_de = -1;
            fa = f->from;
        } else {
            --f->in_de;
            if (f->in_de == 0) cut_branch(c, fa.x, fa.y, ns);
            return;
        }
    }
    ns_add_end(c, ns, max_score, max_NM, max_node);
}

/* ---------------------------------------------------------------- mini DP between two anchors */
static int mini_line(cctx *c, lo_xy left, lo_xy right, lo_xy *line, int *de_score, int *de_NM, int _head, int _tail)
{   /* frag_mini_dp_line, :1068-1150 */
    lo_xy head = _head ? left : START;
    int old_score, old_NM;
    const int left_NM = left.x == -1 ? 0 : HIT(c, left.x, left.y).NM;
    if (_tail == 0) { old_score = 1; old_NM = left_NM; }
    else { old_score = 2 + SCORE_TABLE[NDX(c, right).match_flag]; old_NM = left_NM + HIT(c, right.x, right.y).NM; }
    const int dp_flag = LO_MULTI_FLAG;
    for (int i = left.x + 1; i < right.x; ++i)
        for (int j = 0, n = MAPN(c, i); j < n; ++j)
            if (ND(c, i, j).dp_flag == dp_flag || ND(c, i, j).dp_flag == 0 - dp_flag) node_per_init(c, i, j, head, dp_flag);
    for (int i = left.x + 2; i < right.x; ++i)
        for (int j = 0, n = MAPN(c, i); j < n; ++j)
            if (ND(c, i, j).dp_flag == dp_flag) dp_update(c, i, j, left.x + 1, dp_flag);
    int max_score, max_NM = 0, max_n = 0;
    lo_xy max_node = head;
    if (_tail == 0) {
        max_score = old_score;
        for (int i = right.x - 1; i > left.x; --i)
            for (int j = 0, n = MAPN(c, i); j < n; ++j) {
                lo_node *p = &ND(c, i, j);
                if (p->dp_flag != dp_flag) continue;
                if (p->score > max_score || (p->score == max_score && p->tol_NM < max_NM)) {
                    max_score = p->score; max_NM = p->tol_NM; max_node.x = i; max_node.y = j; max_n = p->node_n;
                }
            }
    } else {
        lo_node *r = &NDX(c, right);
        r->from = head; r->score = old_score; r->tol_NM = old_NM; r->node_n = 1;
        dp_update(c, right.x, right.y, left.x + 1, dp_flag);
        max_score = r->score; max_NM = r->tol_NM; max_node = r->from; max_n = r->node_n - 1;
    }
    lo_xy cur = max_node;
    int node_i = max_n - 1;
    while (cur.x != head.x) {
        if (node_i < 0) bug("mini_line node_i 1");
        line[node_i--] = cur;
        cur = NDX(c, cur).from;
    }
    if (node_i >= 0) bug("mini_line node_i 2");
    *de_score += max_score - old_score;
    *de_NM += max_NM - old_NM;
    return max_n;
}

/* ---------------------------------------------------------------- line clustering */
typedef struct {
    lo_xy *pool; line_t *ln; int n;             /* all lines built so far */
    int *rank, *sel;                            /* line_rank / line_select_rank */
} lset;

static void sort_endpos(lset *L, int ls, int len)
{   /* line_sort_endpos, :12: end seed slot descending, stable */
    int *pos = (int*)malloc(sizeof(int) * (size_t)len), *li = (int*)malloc(sizeof(int) * (size_t)len);
    for (int i = 0; i < len; ++i) { li[i] = ls + i; pos[i] = L->pool[L->ln[ls + i].start + L->ln[ls + i].len - 1].x; }
    for (int i = 1; i < len; ++i) {
        int p = pos[i], l = li[i], k = i - 1;
        while (k >= 0 && pos[k] < p) { pos[k + 1] = pos[k]; li[k + 1] = li[k]; --k; }
        pos[k + 1] = p; li[k + 1] = l;
    }
    for (int i = 0; i < len; ++i) { L->rank[ls + i] = li[i]; L->sel[li[i]] = ls + i; }
    free(pos); free(li);
}

#define FIRSTX(L, l) ((L)->pool[(L)->ln[l].start].x)
#define LASTX(L, l)  ((L)->pool[(L)->ln[l].start + (L)->ln[l].len - 1].x)

static int line_merge(lset *L, int a, int b, float ovlp_r)
{   /* :69-112 */
    line_t *la = &L->ln[a], *lb = &L->ln[b], *lhi;
    int s1, e1, s2, e2, s, e, hi;
    s2 = FIRSTX(L, a); e2 = LASTX(L, a);
    if (lb->mf & L_NMERG) { hi = b; lhi = lb; s1 = FIRSTX(L, b); e1 = LASTX(L, b); }
    else { hi = lb->mh; lhi = &L->ln[hi]; s1 = lhi->lb; e1 = lhi->rb; }
    s = s2 > s1 ? s2 : s1; e = e2 < e1 ? e2 : e1;
    float rat1 = (float)((e - s + 1 + 0.0) / (e1 - s1 + 1 + 0.0));
    float rat2 = (float)((e - s + 1 + 0.0) / (e2 - s2 + 1 + 0.0));
    if (rat1 < ovlp_r && rat2 < ovlp_r) { la->mf = L_NMERG; return 0; }
    if (la->ls <= lb->ls / 2 || la->ls <= lb->bs / 2) {
        lhi->lb = s1; lhi->rb = e1; lhi->mf = L_MERGH;
        la->mf = L_MERGB; la->mh = hi; la->mf |= L_DUMP;
        return 1;
    }
    lhi->lb = s1 + s2 - s; lhi->rb = e1 + e2 - e; lhi->mf = L_MERGH;
    la->mf = L_MERGB; la->mh = hi;
    if (lb->bs > la->bs) la->bs = lb->bs;
    return 1;
}

typedef struct { int x, y, z; } tri_t;

/* cluster -> best + secondaries; shared by line_filter (:122) and line_filter1 (:321).
 * m_f/m_fn (winners per cluster) are only produced when m_f != NULL. */
static void pick_in_cluster(lset *L, tri_t *mb, int mbn, int per_max_multi, int *tri_n, int *mf, int *mfn)
{
    int b_score = 0, s_score = 0;
    for (int j = 0; j < mbn; ++j) {
        if (mb[j].y > b_score) { s_score = b_score; b_score = mb[j].y; }
        else if (mb[j].y > s_score) s_score = mb[j].y;
    }
    if (mfn) *mfn = 1;
    if (s_score >= b_score / 2) {
        nscore_t *ns = ns_new(per_max_multi);
        for (int j = 0; j < mbn; ++j) {
            if (mb[j].y >= b_score / 2) {
                lo_xy nd = { mb[j].x, -1 };
                int ret = ns_add_bounded(ns, nd, mb[j].y, mb[j].z);
                if (ret == -2) { L->ln[mb[j].x].mf |= L_DUMP; if (tri_n) tri_n[mb[j].x] = 0; }
                else if (ret != -1) { L->ln[ret].mf |= L_DUMP; if (tri_n) tri_n[ret] = 0; }
            } else { L->ln[mb[j].x].mf |= L_DUMP; if (tri_n) tri_n[mb[j].x] = 0; }
        }
        for (int i = (ns->node_n - 1) / 2; i >= 0; --i) ns_minpos_sift(ns, i);   /* build_node_minpos_heap */
        int m_head = ns_extract_minpos(ns).x;
        line_t *mh = &L->ln[m_head];
        mh->mf = L_MERGH;
        int min_l = FIRSTX(L, m_head), max_r = LASTX(L, m_head);
        if (mf) { if (mh->ls == b_score) mf[0] = m_head; mf[(*mfn)++] = m_head; }
        int body;
        while ((body = ns_extract_minpos(ns).x) != -1) {
            line_t *bd = &L->ln[body];
            bd->mf = L_MERGB; bd->mh = m_head;
            if (FIRSTX(L, body) < min_l) min_l = FIRSTX(L, body);
            if (LASTX(L, body) > max_r) max_r = LASTX(L, body);
            if (mf) { if (bd->ls == b_score) mf[0] = body; mf[(*mfn)++] = body; }
        }
        mh->lb = FIRSTX(L, m_head) < min_l ? FIRSTX(L, m_head) : min_l;
        mh->rb = LASTX(L, m_head) > max_r ? LASTX(L, m_head) : max_r;
        ns_free(ns);
    } else {
        for (int j = 0; j < mbn; ++j) {
            if (mb[j].y == b_score) { L->ln[mb[j].x].mf = L_NMERG; if (mf) { mf[0] = mb[j].x; mf[(*mfn)++] = mb[j].x; } }
            else { L->ln[mb[j].x].mf |= L_DUMP; if (tri_n) tri_n[mb[j].x] = 0; }
        }
    }
}

static void dump_edge_cluster(lset *L, int ls, int len, int *mf_row, int mfn_row)
{   /* :289-297 / :306-314 */
    for (int i = 1; i < mfn_row; ++i) {
        L->ln[mf_row[i]].mf = L_DUMP;
        for (int _j = ls; _j < ls + len; ++_j) {
            line_t *jl = &L->ln[L->rank[_j]];
            if (!(jl->mf & L_NMERG) && !(jl->mf & L_MERGH) && !(jl->mf & L_DUMP) && jl->mh == mf_row[i]) jl->mf = L_DUMP;
        }
    }
}

static void line_filter(cctx *c, lset *L, int ls, int len, trig_t **trg, int *tri_n, int per_max_multi)
{   /* :122-319 */
    tri_t **m_b = (tri_t**)malloc(sizeof(tri_t*) * (size_t)len);
    int **m_f = (int**)malloc(sizeof(int*) * (size_t)len);
    for (int i = 0; i < len; ++i) { m_b[i] = (tri_t*)malloc(sizeof(tri_t) * (size_t)len); m_f[i] = (int*)malloc(sizeof(int) * (size_t)(len + 1)); }
    int *m_bn = (int*)calloc((size_t)len, sizeof(int)), *m_fn = (int*)calloc((size_t)len, sizeof(int));
    int m_i = -1;
    for (int _i = ls; _i < ls + len; ++_i) {
        int i = L->rank[_i]; line_t *l = &L->ln[i];
        if (l->mf & L_DUMP) continue;
        if (l->mf & L_NMERG) { ++m_i; m_b[m_i][0].x = i; m_b[m_i][0].y = -2; m_bn[m_i] = 1; }
        else if (l->mf & L_MERGH) { ++m_i; m_b[m_i][0].x = i; m_b[m_i][0].y = l->ls; m_b[m_i][0].z = l->nm; m_bn[m_i] = 1; }
        else {
            if (m_i < 0) bug("line_filter: body before head");
            m_b[m_i][m_bn[m_i]].x = i; m_b[m_i][m_bn[m_i]].y = l->ls; m_b[m_i][m_bn[m_i]].z = l->nm; m_bn[m_i]++;
        }
    }
    for (int i = 0; i <= m_i; ++i) {
        if (m_b[i][0].y == -2) { m_f[i][0] = m_b[i][0].x; m_fn[i] = 1; continue; }
        pick_in_cluster(L, m_b[i], m_bn[i], per_max_multi, tri_n, m_f[i], &m_fn[i]);
        for (int ii = 1; ii < m_fn[i]; ++ii) {                       /* inter-lines (candidate inversions), :236-273 */
            int j = m_f[i][ii], _j = L->sel[j];
            for (int k = 0; k < tri_n[j]; ++k) {
                int head = -1;
                lo_xy n1 = trg[j][k].n1, n2 = trg[j][k].n2;
                for (int _l = _j + 1; _l < ls + len; ++_l) {
                    int l = L->rank[_l]; line_t *nl = &L->ln[l];
                    if ((nl->mf & 0x3) != 0) break;
                    if (FIRSTX(L, l) > n1.x && LASTX(L, l) < n2.x) {
                        int mfl = NDX(c, n2).match_flag;
                        if (mfl == LO_F_MISMATCH || mfl == LO_F_LONG_MISMATCH) {
                            lo_xy s = L->pool[nl->start], e = L->pool[nl->start + nl->len - 1];
                            const lo_hit *hs = &HIT(c, s.x, s.y), *he = &HIT(c, e.x, e.y), *h1 = &HIT(c, n1.x, n1.y), *h2 = &HIT(c, n2.x, n2.y);
                            int st = hs->strand;
                            if (st == h1->strand || hs->chr != h1->chr || st * hs->offset < st * h2->offset || st * he->offset > st * h1->offset) continue;
                            nl->mf = L_INTER;
                            if (head == -1) { nl->mf |= L_NMERG; head = l; }
                            else { nl->mf |= L_MERGB; nl->mh = head; L->ln[head].mf = L_INTER | L_MERGH; }
                        }
                    }
                }
            }
        }
    }
    if (m_i > 0) {                                                   /* :279-316 */
        int a = m_f[0][0], b = m_f[1][0];
        if (LASTX(L, a) - FIRSTX(L, a) < 2 && LASTX(L, b) - FIRSTX(L, b) >= 2) dump_edge_cluster(L, ls, len, m_f[0], m_fn[0]);
        a = m_f[m_i][0]; b = m_f[m_i - 1][0];
        if (LASTX(L, a) - FIRSTX(L, a) < 2 && LASTX(L, b) - FIRSTX(L, b) >= 2) dump_edge_cluster(L, ls, len, m_f[m_i], m_fn[m_i]);
    }
    for (int i = 0; i < len; ++i) { free(m_b[i]); free(m_f[i]); }
    free(m_b); free(m_f); free(m_bn); free(m_fn);
}

static void line_filter1(lset *L, int ls, int len, int per_max_multi)
{   /* :321-404 */
    tri_t **m_b = (tri_t**)malloc(sizeof(tri_t*) * (size_t)len);
    for (int i = 0; i < len; ++i) m_b[i] = (tri_t*)malloc(sizeof(tri_t) * (size_t)len);
    int *m_n = (int*)calloc((size_t)len, sizeof(int));
    int m_i = -1;
    for (int _i = ls; _i < ls + len; ++_i) {
        int i = L->rank[_i]; line_t *l = &L->ln[i];
        if ((l->mf & L_DUMP) || (l->mf & L_NMERG)) continue;
        if (l->mf & L_MERGH) { ++m_i; m_b[m_i][0].x = i; m_b[m_i][0].y = l->ls; m_b[m_i][0].z = l->nm; m_n[m_i] = 1; }
        else {
            if (m_i < 0) bug("line_filter1: body before head");
            m_b[m_i][m_n[m_i]].x = i; m_b[m_i][m_n[m_i]].y = l->ls; m_b[m_i][m_n[m_i]].z = l->nm; m_n[m_i]++;
        }
    }
    for (int i = 0; i <= m_i; ++i) pick_in_cluster(L, m_b[i], m_n[i], per_max_multi, NULL, NULL, NULL);
    for (int i = 0; i < len; ++i) free(m_b[i]);
    free(m_b); free(m_n);
}

static int line_remove(lset *L, int ls, int len)
{   /* :406-423 */
    int cur = ls;
    for (int _l = ls; _l < ls + len; ++_l) {
        int l = L->rank[_l];
        if (!(L->ln[l].mf & L_DUMP)) L->rank[cur++] = l;
    }
    return cur - ls;
}

/* line_set_bound (:425) / line_set_bound1 (:496) up to line_remove; with_inter selects line_filter vs line_filter1 */
static int set_bound(cctx *c, lset *L, int ls, int len, trig_t **trg, int *tri_n, int with_inter)
{
    if (len <= 0) return len;
    sort_endpos(L, ls, len);
    L->ln[L->rank[ls]].mf = L_NMERG;
    for (int i = 1; i < len; ++i) line_merge(L, L->rank[ls + i], L->rank[ls + i - 1], c->P->ovlp_rat);
    if (with_inter) line_filter(c, L, ls, len, trg, tri_n, c->P->ske_max);
    else line_filter1(L, ls, len, c->P->ske_max);
    return line_remove(L, ls, len);
}

/* ---------------------------------------------------------------- lines -> fragments */
static void frag_push_seed(lo_frag *f, lo_xy s)
{
    f->seed = (lo_xy*)realloc(f->seed, sizeof(lo_xy) * (size_t)(f->seed_n + 1));
    f->seed[f->seed_n++] = s;
}
static lo_frag *fline_new_frag(cctx *c, lo_fline *fl, lo_xy s)
{   /* frag_set_msg(FRAG_END), frag_check.c:58-81 */
    fl->frag = (lo_frag*)realloc(fl->frag, sizeof(lo_frag) * (size_t)(fl->frag_n + 1));
    lo_frag *f = &fl->frag[fl->frag_n];
    f->chr = HIT(c, s.x, s.y).chr; f->strand = HIT(c, s.x, s.y).strand; f->seed_n = 0; f->seed = NULL;
    frag_push_seed(f, s);
    return f;
}

static void filter_overlap(cctx *c, lset *L, int line_n)
{   /* line_filter_overlap, :568-594 */
    const int seed_len = c->P->seed_len;
    for (int _i = 0; _i < line_n; ++_i) {
        line_t *l = &L->ln[L->rank[_i]];
        lo_xy *ni = L->pool + l->start; int li = l->len, last_i = 0;
        for (int j = 1; j < li - 1; ++j) {
            const lo_hit *cur = &HIT(c, ni[j].x, ni[j].y), *pre = &HIT(c, ni[last_i].x, ni[last_i].y);
            if (seed_len + (cur->strand == 1 ? pre->len_dif : cur->len_dif) > cur->strand * (cur->offset - pre->offset)
                && NDX(c, ni[j]).match_flag != LO_F_INSERT) ni[j].x = -1;
            else last_i = j;
        }
        if (li - 1 != last_i) {
            const lo_hit *cur = &HIT(c, ni[li-1].x, ni[li-1].y), *pre = &HIT(c, ni[last_i].x, ni[last_i].y);
            if (seed_len + (cur->strand == 1 ? pre->len_dif : cur->len_dif) > cur->strand * (cur->offset - pre->offset)
                && NDX(c, ni[li-1]).match_flag != LO_F_INSERT) ni[last_i].x = -1;
        }
    }
}

static int build_flines(cctx *c, lset *L, int line_n, lo_fline **out)
{   /* frag_dp_path, :1152-1250 */
    *out = NULL;
    if (line_n == 0) return 0;
    lo_fline *fl = (lo_fline*)calloc((size_t)line_n, sizeof(lo_fline));
    if (c->P->aln_mode & 1) filter_overlap(c, L, line_n);
    for (int _l = 0; _l < line_n; ++_l) {
        line_t *l = &L->ln[L->rank[_l]];
        lo_xy *ln = L->pool + l->start; int ll = l->len;
        lo_fline *f = &fl[_l];
        lo_xy pre = ln[ll - 1], cur;
        lo_frag *fr = fline_new_frag(c, f, pre);
        f->right_bound = c->S->seed_all + 1;
        for (int i = ll - 1; i > 0; --i) {
            cur = pre;
            if (ln[i - 1].x < 0) continue;
            pre = ln[i - 1];
            int mf = NDX(c, cur).match_flag;
            if (mf == LO_F_INSERT || mf == LO_F_DELETE || mf == LO_F_MISMATCH || mf == LO_F_LONG_MISMATCH) {
                f->frag_n++;                                       /* FRAG_START closes the fragment, a new one opens */
                fr = fline_new_frag(c, f, pre);
            } else if (mf == LO_F_MATCH) frag_push_seed(fr, pre);
            else bug("build_flines: unknown flag");
        }
        f->frag_n++;
        f->left_bound = 0;
        f->line_score = l->ls;
    }
    *out = fl;
    return line_n;
}

void lo_flines_free(lo_fline *f, int n)
{
    if (!f) return;
    for (int i = 0; i < n; ++i) { for (int j = 0; j < f[i].frag_n; ++j) free(f[i].frag[j].seed); free(f[i].frag); }
    free(f);
}

static void lset_init(lset *L, int max_nodes, int max_lines)
{
    L->pool = (lo_xy*)malloc(sizeof(lo_xy) * (size_t)(max_nodes + 8));
    L->ln = (line_t*)calloc((size_t)max_lines + 1, sizeof(line_t));
    L->rank = (int*)malloc(sizeof(int) * (size_t)(max_lines + 1));
    L->sel = (int*)malloc(sizeof(int) * (size_t)(max_lines + 1));
    L->n = 0;
}
static void lset_free(lset *L) { free(L->pool); free(L->ln); free(L->rank); free(L->sel); }

/* ---------------------------------------------------------------- round 1: frag_line_BCC */
int lo_chain_first(lo_seeds *S, const lo_para *P, lo_node *nodes, lo_fline **out)
{
    cctx cc = { S, P, nodes }, *c = &cc;
    const int seed_out = S->seed_out, H = S->hit_off[seed_out];
    int min_n = P->first_loci_thd, min_exist = 0, min_num = 0;
    *out = NULL;
    for (int i = 0; i < seed_out; ++i) {                                      /* dp init, :1315-1323 */
        int n = MAPN(c, i), flag = LO_MULTI_FLAG;
        if (n <= min_n) { flag = LO_MIN_FLAG; min_exist = 1; ++min_num; }
        for (int j = 0; j < n; ++j) node_set(c, i, j, START, 1, HIT(c, i, j).NM, LO_F_MATCH, flag);
    }
    if (!min_exist || min_num * 3 < seed_out) {                               /* :1324-1331 */
        for (int k = 0; k < H; ++k) nodes[k].dp_flag = LO_MIN_FLAG;
        min_n = P->per_aln_m; min_exist = 1;
    }
    if (min_n != P->per_aln_m) {                                              /* frag_min_extend, :1335-1343,:1031 */
        for (int i = 0; i < seed_out; ++i) {
            if (MAPN(c, i) > min_n) continue;
            for (int j = 0, n = MAPN(c, i); j < n; ++j) {
                for (int k = i - 1; k >= 0; --k) {
                    if (MAPN(c, k) <= min_n) continue;
                    for (int a = 0, m = MAPN(c, k); a < m; ++a) {
                        int f = lo_edge_flag(S, P, k, a, i, j);
                        if (f == LO_F_MATCH || f == LO_F_MISMATCH || f == LO_F_LONG_MISMATCH) { ND(c, k, a).dp_flag = LO_MIN_FLAG; break; }
                    }
                }
                for (int k = i + 1; k < seed_out; ++k) {
                    if (MAPN(c, k) <= min_n) continue;
                    for (int a = 0, m = MAPN(c, k); a < m; ++a) {
                        int f = lo_edge_flag(S, P, i, j, k, a);
                        if (f == LO_F_MATCH || f == LO_F_MISMATCH || f == LO_F_LONG_MISMATCH) { ND(c, k, a).dp_flag = LO_MIN_FLAG; break; }
                    }
                }
            }
        }
    }
    for (int i = 1; i < seed_out; ++i)                                        /* main pass, :1345-1350 */
        for (int j = 0, n = MAPN(c, i); j < n; ++j)
            if (ND(c, i, j).dp_flag == LO_MIN_FLAG) dp_update(c, i, j, 0, LO_MIN_FLAG);

    nscore_t *ns = ns_new(0);
    ns->min_score_thd = 2;
    for (int i = seed_out - 1; i >= 0; --i)                                   /* :1356-1361 */
        for (int j = 0, n = MAPN(c, i); j < n; ++j)
            if (ND(c, i, j).dp_flag == LO_MIN_FLAG && ND(c, i, j).in_de == 0) branch_track(c, i, j, ns);

    const int o_l = ns->node_n;
    lset L; lset_init(&L, 2 * H + 6 * o_l + 16, o_l);
    trig_t **trg = (trig_t**)calloc((size_t)o_l + 1, sizeof(trig_t*));
    int *tri_n = (int*)calloc((size_t)o_l + 1, sizeof(int));
    lo_xy *_line = (lo_xy*)malloc(sizeof(lo_xy) * (size_t)(seed_out + 2));
    int l_i = 0, next_start = 0, line_score, line_NM;
    for (;;) {                                                                /* :1370-1432 */
        lo_xy max_node = ns_pop(ns, &line_score, &line_NM);
        if (max_node.x == -1) break;
        lo_xy *ln = L.pool + next_start, last_n, right, left;
        int node_i = 0, mini_len;
        trg[l_i] = (trig_t*)malloc(sizeof(trig_t) * (size_t)(seed_out + 1));
        tri_n[l_i] = 0;
        if (max_node.x < seed_out - 1) {                                      /* beyond the chain end */
            lo_xy virt = { seed_out, 0 };
            mini_len = mini_line(c, max_node, virt, _line, &line_score, &line_NM, 1, 0);
            for (int k = mini_len - 1; k >= 0; --k) { ln[node_i++] = _line[k]; NDX(c, _line[k]).dp_flag = LO_TRACKED_FLAG; }
            ln[node_i] = max_node;
            last_n = ln[0];
            for (int k = mini_len - 1; k >= 0; --k) {
                if (last_n.x - ln[node_i - k].x > 2) { trg[l_i][tri_n[l_i]].n1 = ln[node_i - k]; trg[l_i][tri_n[l_i]].n2 = last_n; tri_n[l_i]++; }
                last_n = ln[node_i - k];
            }
        }
        right = max_node;
        while (right.x != -1) {                                               /* gaps between anchors */
            ln[node_i++] = right;
            left = NDX(c, right).from;
            if (left.x < right.x - 1) {
                mini_len = mini_line(c, left, right, _line, &line_score, &line_NM, 1, 1);
                for (int k = mini_len - 1; k >= 0; --k) { ln[node_i++] = _line[k]; NDX(c, _line[k]).dp_flag = LO_TRACKED_FLAG; }
                ln[node_i] = left;
                last_n = right;
                for (int k = mini_len; k >= 0; --k) {
                    if (last_n.x - ln[node_i - k].x > 2) {
                        if (ln[node_i - k].x == -1) continue;
                        trg[l_i][tri_n[l_i]].n1 = ln[node_i - k]; trg[l_i][tri_n[l_i]].n2 = last_n; tri_n[l_i]++;
                    }
                    last_n = ln[node_i - k];
                }
            }
            right = left;
        }
        for (int k = 0; k < node_i / 2; ++k) { lo_xy t = ln[k]; ln[k] = ln[node_i - k - 1]; ln[node_i - k - 1] = t; }
        line_t *l = &L.ln[l_i];
        l->start = next_start; l->len = node_i; l->ls = l->bs = line_score; l->nm = line_NM; l->mf = 0; l->mh = 0; l->lb = l->rb = 0;
        l_i++; next_start += node_i + 5;
    }
    L.n = l_i;
    int line_n = set_bound(c, &L, 0, o_l, trg, tri_n, 1);                     /* :1435 (min_l == ns->node_n == l_i) */
    for (int i = 0; i < o_l; ++i) free(trg[i]);
    free(trg); free(tri_n); free(_line); ns_free(ns);
    line_n = build_flines(c, &L, line_n, out);
    lset_free(&L);
    return line_n;
}

/* ---------------------------------------------------------------- round 2: frag_line_remain */
static int multi_line(cctx *c, int left_b, int right_b, const lo_reg *trg_reg, lset *L)
{   /* frag_mini_dp_multi_line, :923-1017 */
    if (left_b + 1 >= right_b) return 0;
    const int start = left_b + 1, end = right_b - 1, dp_flag = LO_WHOLE_FLAG;
    for (int i = start; i <= end; ++i)
        for (int j = 0, n = MAPN(c, i); j < n; ++j)
            if (ND(c, i, j).dp_flag != LO_TRACKED_FLAG) node_per_init(c, i, j, START, dp_flag);
    for (int i = start + 1; i <= end; ++i)
        for (int j = 0, n = MAPN(c, i); j < n; ++j)
            if (ND(c, i, j).dp_flag == dp_flag) dp_update(c, i, j, start, dp_flag);
    nscore_t *ns = ns_new(0);
    ns->min_score_thd = 0;
    for (int i = end; i >= start; --i)
        for (int j = 0, n = MAPN(c, i); j < n; ++j)
            if (ND(c, i, j).dp_flag == dp_flag && ND(c, i, j).in_de == 0) branch_track(c, i, j, ns);
    int l_i = 0, next_start = 0, score, NM;
    for (;;) {
        lo_xy r = ns_pop(ns, &score, &NM);
        if (r.x == -1) break;
        int node_i = NDX(c, r).node_n - 1;
        line_t *l = &L->ln[l_i];
        l->start = next_start; l->len = node_i + 1; l->mf = 0; l->mh = 0; l->lb = l->rb = 0;
        next_start += node_i + 1 + 5;
        const lo_hit *h = &HIT(c, r.x, r.y);
        int hit = 0;
        for (int i = 0; i < trg_reg->beg_n && !hit; ++i)                     /* proximity bonus, :979-996 */
            if (h->chr == trg_reg->ref_beg[i].chr &&
                labs((long)((h->offset - trg_reg->ref_beg[i].ref_pos) - (int64_t)((r.x - left_b) * c->P->seed_step))) < c->P->SV_len_thd) hit = 1;
        for (int i = 0; i < trg_reg->end_n && !hit; ++i)
            if (h->chr == trg_reg->ref_end[i].chr &&
                labs((long)((h->offset - trg_reg->ref_end[i].ref_pos) - (int64_t)((r.x - left_b) * c->P->seed_step))) < c->P->SV_len_thd) hit = 1;
        if (hit) { if (score > 1) score += score / 2; else score++; }
        l->ls = l->bs = score; l->nm = NM;
        lo_xy *node = L->pool + l->start;
        while (r.x != -1) {
            if (node_i < 0) bug("multi_line node_i 1");
            node[node_i--] = r;
            r = NDX(c, r).from;
        }
        if (node_i >= 0) bug("multi_line node_i 2");
        ++l_i;
    }
    ns_free(ns);
    return l_i;
}

int lo_chain_remain(lo_areg *a_reg, lo_seeds *S, const lo_para *P, lo_node *nodes, lo_fline **out)
{   /* frag_line_remain, :1252-1302 */
    cctx cc = { S, P, nodes }, *c = &cc;
    const int seed_out = S->seed_out, H = S->hit_off[seed_out];
    *out = NULL;
    lo_areg *re = lo_areg_new(S->read_len);
    int l_n = 0;
    lset L; lset_init(&L, 7 * H + 64, H + 1);       /* final lines */
    lset T; lset_init(&T, 7 * H + 64, H + 1);       /* per-region scratch (_line, _lsl, _line_rank) */
    int next_start = 0;
    if (lo_get_remain_reg(a_reg, re, P, P->seed_len, S->read_len) != 0) {
        for (int i = 0; i < re->reg_n; ++i) {
            int left_id = (re->reg[i].beg + P->seed_inv - 1) / P->seed_step + 1;
            int right_id = (re->reg[i].end - 1) / P->seed_step + 1;
            if (right_id > S->seed_all) right_id -= 1;
            int left = -2, right = -2;
            for (int j = 0; j < seed_out; ++j) if (S->seed_id[j] >= left_id) { left = j - 1; break; }
            if (left == -2) continue;
            for (int j = seed_out - 1; j >= 0; --j) if (S->seed_id[j] <= right_id) { right = j + 1; break; }
            if (right == -2) continue;
            int l = multi_line(c, left, right, &re->reg[i], &T);             /* trg_dp_line, :1019 */
            T.n = l;
            l = set_bound(c, &T, 0, l, NULL, NULL, 0);
            for (int _j = 0; _j < l; ++_j) {                                 /* :1288-1295 */
                int j = T.rank[_j];
                line_t *d = &L.ln[l_n + _j];
                *d = T.ln[j];
                d->start = next_start;
                memcpy(L.pool + next_start, T.pool + T.ln[j].start, sizeof(lo_xy) * (size_t)T.ln[j].len);
                next_start += T.ln[j].len + 5;
                L.rank[l_n + _j] = l_n + _j;
            }
            l_n += l;
        }
    }
    lo_areg_free(re);
    L.n = l_n;
    l_n = build_flines(c, &L, l_n, out);
    lset_free(&L); lset_free(&T);
    return l_n;
}
