/* lo_split.c -- k-mer hash split mapping for large DEL / DUP gaps (oracle; see lo.h).
 *
 * Restates the live part of the reference's src/split_mapping.c for the only way it is
 * reached on the path, split_indel_map (:829) = init_hash + hash_split_map(_head=1,_tail=1):
 *   build_index     <- init_hash, init_hash_pos_num, hash_calcu_pos_start, init_hash_core  :99-208
 *   node_dis        <- hash_main_dis      :218
 *   dp_update       <- hash_dp_update     :341  (limited branch only: both anchors are set)
 *   mini_main_line  <- mini_hash_main_line:444
 *   main_line       <- hash_main_line     :492
 *   split_map       <- hash_split_map     :634  (note: its `split_len` is AP->split_pen, :640)
 * The reference keeps per-key sorted arrays of 64-bit {k-mer, start} entries with a side
 * count table; here the index is (k-mer -> ascending position list), which answers the
 * same two questions: does this k-mer occur, and at which positions in ascending order.
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include "lo_read.h"

static const int8_t NT4_HASH[5] = { 0, 1, 2, 3, 2 };     /* hash_nt4_table, bntseq.c:78 (N hashed as G) */
#define HASH_SV_PEN 2                                    /* split_mapping.h:72 */
#define HASH_MAX_HITS 50                                 /* split_mapping.c:669 */

typedef struct { lo_xy from; int read_i, offset, score, node_n, match_flag, dp_flag; } hnode;

typedef struct {
    const lo_para *P; int ref_len, read_len, ref_offset;
    hnode **h; int *len_a;
} hctx;

static unsigned kmer_code(const uint8_t *s, int hash_len)
{   /* hash_calcu (:81): key = first key_len bases, k-mer = the rest; one combined code suffices */
    unsigned v = 0;
    for (int i = 0; i < hash_len; ++i) v = v << 2 | (unsigned)NT4_HASH[s[i]];
    return v;
}

/* index: counting sort of positions by k-mer code */
typedef struct { int *start, *cnt, *pos; int n_codes; } kidx;
static void build_index(kidx *K, const uint8_t *ref, int ref_len, int hash_len)
{
    K->n_codes = 1 << (2 * hash_len);
    K->start = (int*)calloc((size_t)K->n_codes + 1, sizeof(int));
    K->cnt = (int*)calloc((size_t)K->n_codes, sizeof(int));
    int n = ref_len - hash_len + 1; if (n < 0) n = 0;
    K->pos = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    for (int i = 0; i < n; ++i) K->cnt[kmer_code(ref + i, hash_len)]++;
    for (int c = 0; c < K->n_codes; ++c) K->start[c + 1] = K->start[c] + K->cnt[c];
    int *fill = (int*)calloc((size_t)K->n_codes, sizeof(int));
    for (int i = 0; i < n; ++i) { unsigned c = kmer_code(ref + i, hash_len); K->pos[K->start[c] + fill[c]++] = i; }
    free(fill);
}
static void free_index(kidx *K) { free(K->start); free(K->cnt); free(K->pos); }

static int node_dis(const hctx *c, int a_i, int a_offset, int b_i, int b_offset, int *con_flag)
{   /* hash_main_dis, :218-261 */
    const int hash_len = c->P->hash_len, hash_step = c->P->hash_step;
    const int ref_len = c->ref_len, read_len = c->read_len, ref_offset = c->ref_offset;
    int dis = a_i > b_i ? a_offset - b_offset : b_offset - a_offset;
    int gap = abs(b_i - a_i);
    if (dis == 0) {
        if (gap < hash_len + 2 * hash_step) *con_flag = LO_F_MATCH;
        else if (gap < hash_len + 6 * hash_step) *con_flag = LO_F_MISMATCH;
        else *con_flag = LO_F_LONG_MISMATCH;
    } else if (dis > 0) *con_flag = LO_F_DELETE;
    else if (dis >= -(gap - hash_len)) *con_flag = LO_F_INSERT;
    else if (dis <= -(c->P->split_len / 2)) {
        if (ref_offset > 0) {
            if (b_i > a_i) *con_flag = (read_len - ref_len + b_offset >= -(a_i + hash_len - 1) && read_len - a_offset >= b_i) ? LO_F_INSERT : LO_F_UNCONNECT;
            else *con_flag = (read_len - ref_len + a_offset >= -(b_i + hash_len - 1) && read_len - b_offset >= a_i) ? LO_F_INSERT : LO_F_UNCONNECT;
        } else {
            if (b_i > a_i) *con_flag = (b_offset >= -(a_i - 1) && ref_len - a_offset >= b_i) ? LO_F_INSERT : LO_F_UNCONNECT;
            else *con_flag = (a_offset >= -(b_i - 1) && ref_len - b_offset >= a_i) ? LO_F_INSERT : LO_F_UNCONNECT;
        }
    } else *con_flag = LO_F_UNCONNECT;
    return abs(dis);
}

static void node_init_from(hctx *c, int node_i, lo_xy head, int dp_flag)
{   /* hash_dp_init / hash_mini_dp_init (:264,:399) for a limited head; read_i/offset already set */
    hnode *hd = &c->h[head.x][head.y];
    for (int i = 0; i < c->len_a[node_i]; ++i) {
        hnode *n = &c->h[node_i][i];
        int flag;
        node_dis(c, hd->read_i, hd->offset, n->read_i, n->offset, &flag);
        if (flag == LO_F_UNCONNECT) { n->from.x = -1; n->from.y = 0; n->score = 0; n->node_n = 0; n->match_flag = flag; n->dp_flag = 0 - dp_flag; }
        else { n->from = head; n->score = 2 - (flag <= LO_F_MATCH_THD ? 0 : HASH_SV_PEN); n->node_n = 1; n->match_flag = flag; n->dp_flag = dp_flag; }
    }
}

static void dp_update(hctx *c, int x, int y, int start, int dp_flag)
{   /* hash_dp_update, :341-377 (h_node[x][y].dp_flag is never UNLIMITED here) */
    hnode *t = &c->h[x][y];
    lo_xy max_from = t->from;
    int max_score = t->score, max_flag = 0, flag;
    for (int i = x - 1; i >= start; --i)
        for (int j = 0; j < c->len_a[i]; ++j) {
            hnode *p = &c->h[i][j];
            if (p->dp_flag != dp_flag) continue;
            node_dis(c, p->read_i, p->offset, t->read_i, t->offset, &flag);
            if (flag == LO_F_UNCONNECT) continue;
            int cand = p->score + 1 - (flag <= LO_F_MATCH_THD ? 0 : HASH_SV_PEN);
            if (cand > max_score) { max_score = cand; max_from.x = i; max_from.y = j; max_flag = flag; }
        }
    if (max_from.x != t->from.x || max_from.y != t->from.y) {
        t->score = max_score; t->from = max_from; t->match_flag = max_flag;
        if (max_flag == LO_F_MATCH) c->h[max_from.x][max_from.y].dp_flag = 0 - dp_flag;
        t->node_n += c->h[max_from.x][max_from.y].node_n;
    }
}

static int mini_main_line(hctx *c, lo_xy head, lo_xy tail, lo_xy *line)
{   /* mini_hash_main_line, :444-488 */
    const int flag = LO_MULTI_FLAG;
    for (int i = head.x + 1; i < tail.x; ++i) node_init_from(c, i, head, flag);
    hnode *tl = &c->h[tail.x][tail.y];
    tl->from = head; tl->score = 0; tl->node_n = 0; tl->dp_flag = flag;
    for (int i = head.x + 2; i < tail.x; ++i)
        for (int j = 0; j < c->len_a[i]; ++j)
            if (c->h[i][j].dp_flag == flag) dp_update(c, i, j, head.x + 1, flag);
    dp_update(c, tail.x, tail.y, head.x + 1, flag);
    int node_i = tl->node_n - 1;
    lo_xy cur = tl->from;
    while (cur.x != head.x) {
        if (node_i < 0) break;                        /* the reference only prints a message here */
        line[node_i--] = cur;
        cur = c->h[cur.x][cur.y].from;
    }
    return tl->node_n;
}

static int main_line(hctx *c, const int *hash_pos, const int *start_a, int hash_seed_n, lo_xy *line)
{   /* hash_main_line with _head = _tail = 1, :492-602 */
    const lo_para *P = c->P;
    int min_exist = 0, node_i;
    lo_xy head = { 0, 0 }, tail = { hash_seed_n + 1, 0 };
    hnode *hd = &c->h[0][0], *tl = &c->h[hash_seed_n + 1][0];
    hd->from.x = -1; hd->from.y = 0; hd->read_i = 0 - P->hash_len; hd->offset = 0; hd->score = 0; hd->node_n = 0; hd->match_flag = LO_F_MATCH; hd->dp_flag = LO_MIN_FLAG;
    tl->from = head; tl->read_i = c->read_len; tl->offset = c->ref_len - c->read_len; tl->score = 0; tl->node_n = 0; tl->match_flag = LO_F_UNMATCH; tl->dp_flag = LO_MIN_FLAG;
    for (int i = 1; i <= hash_seed_n; ++i) {
        int read_i = (i - 1) * P->hash_step;
        for (int k = 0; k < c->len_a[i]; ++k) { c->h[i][k].read_i = read_i; c->h[i][k].offset = hash_pos[start_a[i] + k] - read_i; }
        if (c->len_a[i] == 1) { node_init_from(c, i, head, LO_MIN_FLAG); min_exist = 1; }
        else node_init_from(c, i, head, LO_MULTI_FLAG);
    }
    if (min_exist) {
        for (int i = 1; i <= hash_seed_n; ++i) {                 /* hash_min_extend, :312-338 */
            if (c->len_a[i] <= 1) continue;
            for (int a = 0; a < c->len_a[i]; ++a) {
                int done = 0;
                for (int j = 0; j < hash_seed_n + 2 && !done; ++j) {
                    if (c->len_a[j] != 1) continue;
                    for (int k = 0; k < c->len_a[j]; ++k) {
                        if (c->h[i][a].dp_flag < 0) continue;
                        if (c->h[i][a].offset == c->h[j][k].offset) { c->h[i][a].dp_flag = LO_MIN_FLAG; done = 1; break; }
                    }
                }
            }
        }
        for (int i = 2; i <= hash_seed_n; ++i)
            for (int j = 0; j < c->len_a[i]; ++j)
                if (c->h[i][j].dp_flag == LO_MIN_FLAG) dp_update(c, i, j, 1, LO_MIN_FLAG);
        dp_update(c, tail.x, tail.y, 1, LO_MIN_FLAG);
        lo_xy *_line = (lo_xy*)malloc(sizeof(lo_xy) * (size_t)(hash_seed_n + 1));
        lo_xy right = tail, left = c->h[tail.x][tail.y].from;
        node_i = 0;
        for (;;) {
            if (c->h[right.x][right.y].match_flag != LO_F_MATCH && left.x < right.x - 1) {
                int mini_len = mini_main_line(c, left, right, _line);
                for (int i = mini_len - 1; i >= 0; --i) line[node_i++] = _line[i];
            }
            if (left.x == head.x) break;
            line[node_i++] = left;
            right = left;
            left = c->h[right.x][right.y].from;
        }
        free(_line);
        for (int i = 0; i < node_i / 2; ++i) { lo_xy t = line[i]; line[i] = line[node_i - 1 - i]; line[node_i - 1 - i] = t; }
        return node_i;
    }
    for (int i = 2; i <= hash_seed_n; ++i)
        for (int j = 0; j < c->len_a[i]; ++j)
            if (c->h[i][j].dp_flag == LO_MULTI_FLAG) dp_update(c, i, j, 1, LO_MULTI_FLAG);
    dp_update(c, tail.x, tail.y, 1, LO_MULTI_FLAG);
    node_i = tl->node_n - 1;
    lo_xy cur = tl->from;
    while (cur.x != head.x) {
        if (node_i < 0) { fprintf(stderr, "[lo_split] main_line node_i < 0\n"); exit(1); }
        line[node_i--] = cur;
        cur = c->h[cur.x][cur.y].from;
    }
    if (node_i >= 0) { fprintf(stderr, "[lo_split] main_line node_i >= 0\n"); exit(1); }
    return tl->node_n;
}

static int indel_cigar(int ref_left, int read_left, int ref_right, int read_right, lo_cig *cg, int *clen, int split_len, int *split_flag)
{   /* make_indel_cigar, :606-632 */
    int dlen = ref_left - ref_right + 1, ilen = read_left - read_right + 1;
    if (dlen < 0 && ilen < 0) { fprintf(stderr, "[lo_split] indel_cigar error\n"); exit(1); }
    int len = ilen - dlen;
    if (len > 0) { *clen = 1; cg[0] = (len << 4) + LO_D; if (len >= split_len) *split_flag |= 2; }
    else if (len < 0) { *clen = 1; cg[0] = ((0 - len) << 4) + LO_I; if (-len >= split_len) *split_flag |= 2; }
    else *clen = 0;
    return dlen > ilen ? dlen : ilen;
}

int lo_split_indel_map(lo_cigv *out, const uint8_t *read_seq, int read_len, const uint8_t *ref_seq, int ref_len,
                       int ref_offset, const lo_para *P)
{   /* split_indel_map (:829) -> hash_split_map (:634) with _head = _tail = 1 */
    const int hash_len = P->hash_len, hash_step = P->hash_step, split_len = P->split_pen;   /* sic, :640 */
    int res = 0;
    kidx K; build_index(&K, ref_seq, ref_len, hash_len);
    const int hash_seed_n = (read_len - hash_len) / hash_step + 1;
    int *start_a = (int*)calloc((size_t)hash_seed_n + 3, sizeof(int)), *len_a = (int*)calloc((size_t)hash_seed_n + 3, sizeof(int));
    lo_cigv_clear(out);
    len_a[0] = 1;
    int i;
    for (i = 0; i <= read_len - hash_len; i += hash_step) {
        unsigned code = kmer_code(read_seq + i, hash_len);
        int slot = i / hash_step + 1;
        if (K.cnt[code] > 0) { start_a[slot] = K.start[code]; len_a[slot] = K.cnt[code] > HASH_MAX_HITS ? 0 : K.cnt[code]; }
        else len_a[slot] = 0;
    }
    len_a[i / hash_step + 1] = 1;                                      /* tail node, :674 */
    hctx cx = { P, ref_len, read_len, ref_offset, NULL, len_a };
    lo_xy *line = (lo_xy*)malloc(sizeof(lo_xy) * (size_t)(hash_seed_n + 1));
    cx.h = (hnode**)malloc(sizeof(hnode*) * (size_t)(hash_seed_n + 2));
    for (i = 0; i < hash_seed_n + 2; ++i) cx.h[i] = (hnode*)calloc((size_t)len_a[i] + 1, sizeof(hnode));
    const int m_len = main_line(&cx, K.pos, start_a, hash_seed_n, line);
    hnode **h = cx.h;
#define HN(k) (h[line[k].x][line[k].y])
    lo_cigv tmp; lo_cigv_init(&tmp);
    lo_cig g[1]; int _clen = 0, _q_len, _t_len;
    const int tail_in = hash_len / 2, head_in = (hash_len + 1) / 2;
    if (m_len > 0) {
        int _refi = HN(0).read_i + HN(0).offset, _readi = HN(0).read_i;
        _q_len = _readi + tail_in; _t_len = _refi + tail_in;
        if (_readi != 0 && _refi != 0) {                               /* 1. left blank, :700-715 */
            if (_t_len < P->split_len && _q_len < P->split_len) lo_ksw_global(_q_len, read_seq, _t_len, ref_seq, P->sc_mat, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, P->band_w, &tmp);
            else res |= lo_ksw_bi_extend(_q_len, read_seq, _t_len, ref_seq, hash_len * P->match, hash_len * P->match, P, &tmp);
            lo_cig_pushv(out, tmp.c, tmp.n);
        } else {
            indel_cigar(-1, -1, _refi, _readi, g, &_clen, split_len, &res);
            lo_cig_pushv(out, g, _clen);
            lo_cig_push1(out, (tail_in << 4) | LO_M);
        }
        int start_i = 0, overlap = 0;                                  /* 2. between anchors, :718-784 */
        for (i = 0; i < m_len; ++i) {
            if (!(i == m_len - 1 || HN(i + 1).match_flag >= LO_F_MATCH_THD)) continue;
            lo_cig_push1(out, ((HN(i).read_i - HN(start_i).read_i + hash_len - tail_in - head_in - overlap) << 4) | LO_M);
            if (i == m_len - 1) break;
            int l_readi = HN(i).read_i + hash_len - 1, r_readi = HN(i + 1).read_i;
            int l_refi = HN(i).read_i + hash_len + HN(i).offset - 1, r_refi = HN(i + 1).read_i + HN(i + 1).offset;
            int l_offset = HN(i).offset, r_offset = HN(i + 1).offset;
            if (l_readi + 1 < r_readi && l_refi + 1 < r_refi) {
                _q_len = r_readi - (l_readi + 1) + head_in + tail_in;
                _t_len = _q_len + r_offset - l_offset;
                if (_q_len < P->split_len && _t_len < P->split_len)
                    lo_ksw_global(_q_len, read_seq + l_readi + 1 - head_in, _t_len, ref_seq + l_refi + 1 - head_in, P->sc_mat, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, P->band_w, &tmp);
                else res |= lo_ksw_bi_extend(_q_len, read_seq + l_readi + 1 - head_in, _t_len, ref_seq + l_refi + 1 - head_in, hash_len * P->match, hash_len * P->match, P, &tmp);
                lo_cig_pushv(out, tmp.c, tmp.n);
                overlap = 0;
            } else if (l_refi >= r_refi) {                             /* overlap on the reference, :746-775 */
                int lqe, lte, rqe, rte;
                _q_len = r_readi - (l_readi + 1) + head_in;
                _t_len = _q_len + (ref_offset > 0 ? hash_len : 0);
                lo_ksw_extend(_q_len, read_seq + l_readi + 1 - head_in, _t_len, ref_seq + l_refi + 1 - head_in, P->sc_mat, P->band_w, hash_len * P->match, P, &lqe, &lte, &tmp);
                lo_cig_pushv(out, tmp.c, tmp.n);
                _q_len = r_readi - (l_readi + 1) + tail_in;
                _t_len = _q_len + (ref_offset > 0 ? hash_len : 0);
                if (r_readi + tail_in - _q_len < 0 || r_refi + tail_in - _t_len < -ref_offset - (ref_offset > 0 ? hash_len : 0)) { fprintf(stderr, "[lo_split] BUG (reference exits, :760)\n"); exit(1); }
                uint8_t *rq = (uint8_t*)malloc((size_t)_q_len + 1), *rt = (uint8_t*)malloc((size_t)_t_len + 1);
                for (int j = 0; j < _q_len; ++j) rq[j] = read_seq[r_readi + tail_in - 1 - j];
                for (int j = 0; j < _t_len; ++j) rt[j] = ref_seq[r_refi + tail_in - 1 - j];
                lo_ksw_extend(_q_len, rq, _t_len, rt, P->sc_mat, P->band_w, hash_len * P->match, P, &rqe, &rte, &tmp);
                lo_cig_invert(tmp.c, tmp.n);
                free(rq); free(rt);
                int Sn = _q_len + head_in - lqe - rqe, Hn = r_refi + head_in + tail_in - l_refi - 1 - lte - rte;
                lo_cig_push0(out, (Sn << 4) | LO_S);
                lo_cig_push0(out, (Hn << 4) | LO_H);
                lo_cig_pushv(out, tmp.c, tmp.n);
                overlap = 0;
            } else {
                lo_cig_push1(out, (head_in << 4) | LO_M);
                overlap = indel_cigar(l_refi, l_readi, r_refi, r_readi, g, &_clen, split_len, &res);
                lo_cig_pushv(out, g, _clen);
                lo_cig_push1(out, (tail_in << 4) | LO_M);
            }
            start_i = i + 1;
        }
        _readi = HN(m_len - 1).read_i + hash_len - 1;                 /* 3. right blank, :786-805 */
        _refi = HN(m_len - 1).read_i + HN(m_len - 1).offset + hash_len - 1;
        _q_len = read_len - (_readi + 1) + head_in; _t_len = ref_len - (_refi + 1) + head_in;
        if (_readi + 1 < read_len && _refi + 1 < ref_len) {
            if (_q_len < P->split_len && _t_len < P->split_len)
                lo_ksw_global(_q_len, read_seq + _readi + 1 - head_in, _t_len, ref_seq + _refi + 1 - head_in, P->sc_mat, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, P->band_w, &tmp);
            else res |= lo_ksw_bi_extend(_q_len, read_seq + _readi + 1 - head_in, _t_len, ref_seq + _refi + 1 - head_in, hash_len * P->match, hash_len * P->match, P, &tmp);
            lo_cig_pushv(out, tmp.c, tmp.n);
        } else {
            lo_cig_push1(out, (head_in << 4) | LO_M);
            indel_cigar(_refi, _readi, ref_len, read_len, g, &_clen, split_len, &res);
            lo_cig_pushv(out, g, _clen);
        }
    } else {                                                          /* no anchors, :807-819 */
        _t_len = ref_len; _q_len = read_len;
        if (_t_len < P->split_len && _q_len < P->split_len) lo_ksw_global(_q_len, read_seq, _t_len, ref_seq, P->sc_mat, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, P->band_w, &tmp);
        else res |= lo_ksw_bi_extend(_q_len, read_seq, _t_len, ref_seq, hash_len * P->match, hash_len * P->match, P, &tmp);
        lo_cig_pushv(out, tmp.c, tmp.n);
    }
#undef HN
    lo_cigv_free(&tmp);
    for (i = 0; i < hash_seed_n + 2; ++i) free(cx.h[i]);
    free(cx.h); free(line); free(start_a); free(len_a); free_index(&K);
    return res;
}
