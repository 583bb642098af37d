// hp_split.h -- k-mer split mapping of large DEL / DUP gaps on one wavefront
// (reference src/split_mapping.c, reached only through
//  split_indel_map :829 = init_hash :181 + hash_split_map :634 with _head = _tail = 1).
//
// The reference indexes every k-mer of the reference window in 16 sorted buckets and looks
// the read's k-mers up one by one.  Here the window's k-mer codes are computed once by the
// 64 lanes, and each read k-mer is matched against all of them with ballots -- no sort, no
// atomics, positions come out ascending exactly like the reference's position lists.
// The mini chaining over k-mer hits and the gap filling are wave-uniform.
#pragma once
#include "hp_chain.h"
#include "hp_ksw.h"

namespace hp {

#define HP_HASH_SV_PEN 2         // split_mapping.h:72
#define HP_HASH_MAX_HITS 50      // split_mapping.c:669

HP_INL unsigned kmer_code(const uint8_t *s, int hash_len)
{   // hash_calcu (:81) with hash_nt4_table = {0,1,2,3,2} (bntseq.c:78): N is hashed as G
    unsigned v = 0;
    for (int i = 0; i < hash_len; ++i) { unsigned b = s[i]; v = v << 2 | (b > 3 ? 2u : b); }
    return v;
}

struct HCtx {
    const lamsa_hp_para *P; int ref_len, read_len, ref_offset;
    int32_t *nstart, *len_a;                                  // per slot: first node, number of nodes
    int32_t *h_from, *h_read_i, *h_offset, *h_score, *h_node_n, *h_slot; int8_t *h_match, *h_dp;
};

HP_FN int hnode_dis(const HCtx &c, int a_i, int a_offset, int b_i, int b_offset)
{   // hash_main_dis, :218-261; returns the connect flag
    const int hash_len = c.P->hash_len, hash_step = c.P->hash_step;
    const int ref_len = c.ref_len, read_len = c.read_len, ref_offset = c.ref_offset;
    const int dis = a_i > b_i ? a_offset - b_offset : b_offset - a_offset;
    const int gap = iabs(b_i - a_i);
    if (dis == 0) {
        if (gap < hash_len + 2 * hash_step) return F_MATCH;
        if (gap < hash_len + 6 * hash_step) return F_MISMATCH;
        return F_LONG_MISMATCH;
    }
    if (dis > 0) return F_DELETE;
    if (dis >= -(gap - hash_len)) return F_INSERT;
    if (dis <= -(c.P->split_len / 2)) {
        if (ref_offset > 0) {
            if (b_i > a_i) return (read_len - ref_len + b_offset >= -(a_i + hash_len - 1) && read_len - a_offset >= b_i) ? F_INSERT : F_UNCONNECT;
            HP_STAT(11);
            return (read_len - ref_len + a_offset >= -(b_i + hash_len - 1) && read_len - b_offset >= a_i) ? F_INSERT : F_UNCONNECT;
        }
        if (b_i > a_i) return (b_offset >= -(a_i - 1) && ref_len - a_offset >= b_i) ? F_INSERT : F_UNCONNECT;
        HP_STAT(12);
        return (a_offset >= -(b_i - 1) && ref_len - b_offset >= a_i) ? F_INSERT : F_UNCONNECT;
    }
    return F_UNCONNECT;
}

HP_FN void hnode_init_from(HCtx &c, int slot, int head, int dp_flag)
{   // hash_dp_init / hash_mini_dp_init (:264,:399), limited head
    for (int k = c.nstart[slot], e = c.nstart[slot] + c.len_a[slot]; k < e; ++k) {
        const int flag = hnode_dis(c, c.h_read_i[head], c.h_offset[head], c.h_read_i[k], c.h_offset[k]);
        if (flag == F_UNCONNECT) { c.h_from[k] = -1; c.h_score[k] = 0; c.h_node_n[k] = 0; c.h_match[k] = (int8_t)flag; c.h_dp[k] = (int8_t)(0 - dp_flag); }
        else { c.h_from[k] = head; c.h_score[k] = 2 - (flag <= F_MATCH_THD ? 0 : HP_HASH_SV_PEN); c.h_node_n[k] = 1; c.h_match[k] = (int8_t)flag; c.h_dp[k] = (int8_t)dp_flag; }
    }
}

HP_FN void hdp_update(HCtx &c, int t, int start_slot, int dp_flag)
{   // hash_dp_update, :341-377 (limited branch)
    int max_from = c.h_from[t], max_score = c.h_score[t], max_flag = 0;
    const int x = c.h_slot[t];
    for (int i = x - 1; i >= start_slot; --i)
        for (int p = c.nstart[i], e = c.nstart[i] + c.len_a[i]; p < e; ++p) {
            if (c.h_dp[p] != dp_flag) continue;
            const int flag = hnode_dis(c, c.h_read_i[p], c.h_offset[p], c.h_read_i[t], c.h_offset[t]);
            if (flag == F_UNCONNECT) continue;
            const int cand = c.h_score[p] + 1 - (flag <= F_MATCH_THD ? 0 : HP_HASH_SV_PEN);
            if (cand > max_score) { max_score = cand; max_from = p; max_flag = flag; }
        }
    if (max_from != c.h_from[t]) {
        c.h_score[t] = max_score; c.h_from[t] = max_from; c.h_match[t] = (int8_t)max_flag;
        if (max_flag == F_MATCH) c.h_dp[max_from] = (int8_t)(0 - dp_flag);
        c.h_node_n[t] += c.h_node_n[max_from];
    }
}

HP_INL int hslot_of(const HCtx &c, int node) { return node < 0 ? -1 : c.h_slot[node]; }

HP_FN int hmini_main_line(HCtx &c, int head, int tail, int32_t *line)
{   // mini_hash_main_line, :444-488
    const int flag = MULTI_FLAG, hx = c.h_slot[head], tx = c.h_slot[tail];
    for (int i = hx + 1; i < tx; ++i) hnode_init_from(c, i, head, flag);
    c.h_from[tail] = head; c.h_score[tail] = 0; c.h_node_n[tail] = 0; c.h_dp[tail] = (int8_t)flag;
    for (int i = hx + 2; i < tx; ++i)
        for (int k = c.nstart[i], e = c.nstart[i] + c.len_a[i]; k < e; ++k)
            if (c.h_dp[k] == flag) hdp_update(c, k, hx + 1, flag);
    hdp_update(c, tail, hx + 1, flag);
    int node_i = c.h_node_n[tail] - 1, cur = c.h_from[tail];
    while (hslot_of(c, cur) != hx) {
        if (node_i < 0 || cur < 0) break;
        line[node_i--] = cur;
        cur = c.h_from[cur];
    }
    return c.h_node_n[tail];
}

HP_FN int hmain_line(Ctx &cx, HCtx &c, int hash_seed_n, int32_t *line)
{   // hash_main_line with _head = _tail = 1, :492-602
    const lamsa_hp_para *P = c.P;
    const int head = c.nstart[0], tail = c.nstart[hash_seed_n + 1];
    int min_exist = 0, node_i;
    c.h_from[head] = -1; c.h_read_i[head] = 0 - P->hash_len; c.h_offset[head] = 0; c.h_score[head] = 0; c.h_node_n[head] = 0; c.h_match[head] = F_MATCH; c.h_dp[head] = MIN_FLAG;
    c.h_from[tail] = head; c.h_read_i[tail] = c.read_len; c.h_offset[tail] = c.ref_len - c.read_len; c.h_score[tail] = 0; c.h_node_n[tail] = 0; c.h_match[tail] = F_UNMATCH; c.h_dp[tail] = MIN_FLAG;
    for (int i = 1; i <= hash_seed_n; ++i) {
        if (c.len_a[i] == 1) { hnode_init_from(c, i, head, MIN_FLAG); min_exist = 1; }
        else hnode_init_from(c, i, head, MULTI_FLAG);
    }
    if (min_exist) {
        for (int i = 1; i <= hash_seed_n; ++i) {                     // hash_min_extend, :312-338
            if (c.len_a[i] <= 1) continue;
            for (int a = c.nstart[i], ae = c.nstart[i] + c.len_a[i]; a < ae; ++a) {
                if (c.h_dp[a] < 0) continue;
                for (int j = 0; j < hash_seed_n + 2; ++j) {
                    if (c.len_a[j] != 1) continue;
                    if (c.h_offset[a] == c.h_offset[c.nstart[j]]) { c.h_dp[a] = MIN_FLAG; break; }
                }
            }
        }
        for (int i = 2; i <= hash_seed_n; ++i)
            for (int k = c.nstart[i], e = c.nstart[i] + c.len_a[i]; k < e; ++k)
                if (c.h_dp[k] == MIN_FLAG) hdp_update(c, k, 1, MIN_FLAG);
        hdp_update(c, tail, 1, MIN_FLAG);
        const size_t mark = arena_mark(cx.tmp);
        int32_t *_line = (int32_t *)arena_alloc(cx, sizeof(int32_t) * (size_t)(hash_seed_n + 2));
        if (!_line) return 0;
        int right = tail, left = c.h_from[tail];
        node_i = 0;
        for (int guard = 0; guard < hash_seed_n + 4; ++guard) {
            if (c.h_match[right] != F_MATCH && hslot_of(c, left) < c.h_slot[right] - 1) {
                const int mini_len = hmini_main_line(c, left, right, _line);
                for (int i = mini_len - 1; i >= 0 && node_i < hash_seed_n; --i) line[node_i++] = _line[i];
            }
            if (hslot_of(c, left) == 0) break;
            if (left < 0 || node_i >= hash_seed_n) break;
            line[node_i++] = left;
            right = left;
            left = c.h_from[right];
        }
        arena_release(cx.tmp, mark);
        for (int i = 0; i < node_i / 2; ++i) { int t = line[i]; line[i] = line[node_i - 1 - i]; line[node_i - 1 - i] = t; }
        return node_i;
    }
    for (int i = 2; i <= hash_seed_n; ++i)
        for (int k = c.nstart[i], e = c.nstart[i] + c.len_a[i]; k < e; ++k)
            if (c.h_dp[k] == MULTI_FLAG) { HP_STAT(13); hdp_update(c, k, 1, MULTI_FLAG); }
    hdp_update(c, tail, 1, MULTI_FLAG);
    node_i = c.h_node_n[tail] - 1;
    int cur = c.h_from[tail];
    while (hslot_of(c, cur) != 0) {
        if (node_i < 0 || cur < 0) { cx.status |= ST_REFEXIT; return 0; }      // "[hash main line] bug" exit, :589
        line[node_i--] = cur;
        cur = c.h_from[cur];
    }
    if (node_i >= 0) { cx.status |= ST_REFEXIT; return 0; }
    return c.h_node_n[tail];
}

HP_FN int indel_cigar(Ctx &cx, int ref_left, int read_left, int ref_right, int read_right, cig_t *cg, int *clen, int split_len, int *split_flag)
{   // make_indel_cigar, :606-632
    const int dlen = ref_left - ref_right + 1, ilen = read_left - read_right + 1;
    if (dlen < 0 && ilen < 0) { cx.status |= ST_REFEXIT; *clen = 0; return 0; }
    const int len = ilen - dlen;
    if (len > 0) { *clen = 1; cg[0] = (len << 4) + C_D; if (len >= split_len) *split_flag |= 2; }
    else if (len < 0) { *clen = 1; cg[0] = ((0 - len) << 4) + C_I; if (-len >= split_len) *split_flag |= 2; }
    else *clen = 0;
    return dlen > ilen ? dlen : ilen;
}

// split_indel_map (:829).  read/ref are plain byte sequences in HBM (forward views).  Appends nothing on
// failure; `out` is cleared first like hash_split_map does (:652).
HP_NOINL int split_indel_map(Ctx &cx, CigV &out, const uint8_t *read_seq, int read_len, const uint8_t *ref_seq, int ref_len, int ref_offset)
{
    const lamsa_hp_para *P = cx.P;
    const int hash_len = P->hash_len, hash_step = P->hash_step, split_len = P->split_pen;   // sic, :640
    int res = 0;
    out.n = 0;
    if (read_len < hash_len) { cx.status |= ST_REFEXIT; return 0; }
    const size_t mark = arena_mark(cx.tmp);
    const int n_codes = ref_len - hash_len + 1 > 0 ? ref_len - hash_len + 1 : 0;
    const int hash_seed_n = (read_len - hash_len) / hash_step + 1;
    uint32_t *rcode = (uint32_t *)arena_alloc(cx, sizeof(uint32_t) * (size_t)(n_codes + 1));
    int32_t *nstart = (int32_t *)arena_alloc(cx, sizeof(int32_t) * (size_t)(hash_seed_n + 3));
    int32_t *len_a = (int32_t *)arena_alloc(cx, sizeof(int32_t) * (size_t)(hash_seed_n + 3));
    int32_t *line = (int32_t *)arena_alloc(cx, sizeof(int32_t) * (size_t)(hash_seed_n + 2));
    if (!rcode || !nstart || !len_a || !line) { arena_release(cx.tmp, mark); return 0; }
    for (int b0 = 0; b0 < n_codes; b0 += 64) {                          // k-mer code of every window position
        WAVE_FOR(l) { int i = b0 + l; if (i < n_codes) rcode[i] = kmer_code(ref_seq + i, hash_len); }
    }
    wv::sync();
    // pass 1: number of window positions matching each read k-mer (more than 50: ignored, :669)
    int nn = 1;
    len_a[0] = 1;
    for (int s = 1; s <= hash_seed_n; ++s) {
        const unsigned qc = kmer_code(read_seq + (s - 1) * hash_step, hash_len);
        int cnt = 0;
        for (int b0 = 0; b0 < n_codes && cnt <= HP_HASH_MAX_HITS; b0 += 64) {
            wv::Lane<int> eq;
            WAVE_FOR(l) { int i = b0 + l; eq[l] = (i < n_codes && rcode[i] == qc); }
            cnt += __builtin_popcountll(wv::ballot(eq));
        }
        len_a[s] = cnt > HP_HASH_MAX_HITS ? 0 : cnt;
        nn += len_a[s];
    }
    len_a[hash_seed_n + 1] = 1; ++nn;
    const int node_cap = nn + 1;
    int32_t *nm = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 6 * (size_t)node_cap);
    int8_t *nb = (int8_t *)arena_alloc(cx, 2 * (size_t)node_cap);
    if (!nm || !nb) { arena_release(cx.tmp, mark); return 0; }
    HCtx c;
    c.P = P; c.ref_len = ref_len; c.read_len = read_len; c.ref_offset = ref_offset; c.nstart = nstart; c.len_a = len_a;
    c.h_from = nm; c.h_read_i = nm + node_cap; c.h_offset = nm + 2 * node_cap; c.h_score = nm + 3 * node_cap; c.h_node_n = nm + 4 * node_cap; c.h_slot = nm + 5 * node_cap;
    c.h_match = nb; c.h_dp = nb + node_cap;
    // pass 2: the matching positions, ascending, become the DP nodes of the slot
    nn = 0;
    nstart[0] = nn; c.h_slot[nn] = 0; ++nn;                             // head node
    for (int s = 1; s <= hash_seed_n; ++s) {
        const int read_i = (s - 1) * hash_step;
        nstart[s] = nn;
        if (len_a[s] == 0) continue;
        const unsigned qc = kmer_code(read_seq + read_i, hash_len);
        int cnt = 0;
        for (int b0 = 0; b0 < n_codes && cnt < len_a[s]; b0 += 64) {
            wv::Lane<int> eq;
            WAVE_FOR(l) { int i = b0 + l; eq[l] = (i < n_codes && rcode[i] == qc); }
            unsigned long long m = wv::ballot(eq);
            while (m && cnt < len_a[s]) {
                const int bit = __builtin_ctzll(m); m &= m - 1;
                const int k = nn + cnt; c.h_slot[k] = s; c.h_read_i[k] = read_i; c.h_offset[k] = (b0 + bit) - read_i;
                ++cnt;
            }
        }
        nn += len_a[s];
    }
    nstart[hash_seed_n + 1] = nn; c.h_slot[nn] = hash_seed_n + 1; ++nn;   // tail node
    wv::sync();
    const int m_len = hmain_line(cx, c, hash_seed_n, line);
    CigV tmp;
    if (!cig_alloc(cx, tmp, read_len + ref_len + 16) || (cx.status & ST_REFEXIT)) { arena_release(cx.tmp, mark); return 0; }
    cig_t g[1]; int _clen = 0, _q_len, _t_len;
    const int tail_in = hash_len / 2, head_in = (hash_len + 1) / 2;
    const int gh0 = hash_len * P->match;
#define HN_RI(k) (c.h_read_i[line[k]])
#define HN_OF(k) (c.h_offset[line[k]])
#define HP_GLOBAL(ql, qp, tl, tp) ksw_global(cx, (ql), seq_fwd(qp), (tl), seq_fwd(tp), P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, P->band_w, &tmp)
    if (m_len > 0) {
        int _refi = HN_RI(0) + HN_OF(0), _readi = HN_RI(0);
        _q_len = _readi + tail_in; _t_len = _refi + tail_in;
        if (_readi != 0 && _refi != 0) {                                  // 1. left blank, :700-715
            if (_t_len < P->split_len && _q_len < P->split_len) HP_GLOBAL(_q_len, read_seq, _t_len, ref_seq);
            else res |= ksw_bi_extend(cx, _q_len, seq_fwd(read_seq), _t_len, seq_fwd(ref_seq), gh0, gh0, tmp);
            cig_pushv(cx, out, tmp.c, tmp.n);
        } else {
            indel_cigar(cx, -1, -1, _refi, _readi, g, &_clen, split_len, &res);
            if (_clen) cig_pushw(cx, out, g[0]);
            cig_push1(cx, out, (tail_in << 4) | C_M);
        }
        int start_i = 0, overlap = 0;                                     // 2. between anchors, :718-784
        for (int i = 0; i < m_len; ++i) {
            if (!(i == m_len - 1 || c.h_match[line[i + 1]] >= F_MATCH_THD)) continue;
            cig_push1(cx, out, ((HN_RI(i) - HN_RI(start_i) + hash_len - tail_in - head_in - overlap) << 4) | C_M);
            if (i == m_len - 1) break;
            const int l_readi = HN_RI(i) + hash_len - 1, r_readi = HN_RI(i + 1);
            const int l_refi = HN_RI(i) + hash_len + HN_OF(i) - 1, r_refi = HN_RI(i + 1) + HN_OF(i + 1);
            const int l_offset = HN_OF(i), r_offset = HN_OF(i + 1);
            if (l_readi + 1 < r_readi && l_refi + 1 < r_refi) {
                _q_len = r_readi - (l_readi + 1) + head_in + tail_in;
                _t_len = _q_len + r_offset - l_offset;
                if (_q_len < P->split_len && _t_len < P->split_len) HP_GLOBAL(_q_len, read_seq + l_readi + 1 - head_in, _t_len, ref_seq + l_refi + 1 - head_in);
                else res |= ksw_bi_extend(cx, _q_len, seq_fwd(read_seq + l_readi + 1 - head_in), _t_len, seq_fwd(ref_seq + l_refi + 1 - head_in), gh0, gh0, tmp);
                cig_pushv(cx, out, tmp.c, tmp.n);
                overlap = 0;
            } else if (l_refi >= r_refi) {                                // overlap on the reference, :746-775
                int lqe, lte, rqe, rte;
                _q_len = r_readi - (l_readi + 1) + head_in;
                _t_len = _q_len + (ref_offset > 0 ? hash_len : 0);
                ksw_extend(cx, _q_len, seq_fwd(read_seq + l_readi + 1 - head_in), _t_len, seq_fwd(ref_seq + l_refi + 1 - head_in), P->band_w, gh0, &lqe, &lte, &tmp);
                cig_pushv(cx, out, tmp.c, tmp.n);
                _q_len = r_readi - (l_readi + 1) + tail_in;
                _t_len = _q_len + (ref_offset > 0 ? hash_len : 0);
                if (r_readi + tail_in - _q_len < 0 || r_refi + tail_in - _t_len < -ref_offset - (ref_offset > 0 ? hash_len : 0)) { cx.status |= ST_REFEXIT; break; }   // "[hash_split_map] BUG" exit, :760
                Seq rq, rt;                                               // reversed views ending at r_readi+tail_in-1 / r_refi+tail_in-1
                rq.p = read_seq + (r_readi + tail_in - 1); rq.stride = -1;
                rt.p = ref_seq + (r_refi + tail_in - 1); rt.stride = -1;
                ksw_extend(cx, _q_len, rq, _t_len, rt, P->band_w, gh0, &rqe, &rte, &tmp);
                cig_invert(tmp.c, tmp.n);
                const int Sn = _q_len + head_in - lqe - rqe, Hn = r_refi + head_in + tail_in - l_refi - 1 - lte - rte;
                cig_push0(cx, out, (Sn << 4) | C_S);
                cig_push0(cx, out, (Hn << 4) | C_H);
                cig_pushv(cx, out, tmp.c, tmp.n);
                overlap = 0;
            } else {
                cig_push1(cx, out, (head_in << 4) | C_M);
                overlap = indel_cigar(cx, l_refi, l_readi, r_refi, r_readi, g, &_clen, split_len, &res);
                if (_clen) cig_pushw(cx, out, g[0]);
                cig_push1(cx, out, (tail_in << 4) | C_M);
            }
            start_i = i + 1;
        }
        _readi = HN_RI(m_len - 1) + hash_len - 1;                         // 3. right blank, :786-805
        _refi = HN_RI(m_len - 1) + HN_OF(m_len - 1) + hash_len - 1;
        _q_len = read_len - (_readi + 1) + head_in; _t_len = ref_len - (_refi + 1) + head_in;
        if (_readi + 1 < read_len && _refi + 1 < ref_len) {
            if (_q_len < P->split_len && _t_len < P->split_len) HP_GLOBAL(_q_len, read_seq + _readi + 1 - head_in, _t_len, ref_seq + _refi + 1 - head_in);
            else res |= ksw_bi_extend(cx, _q_len, seq_fwd(read_seq + _readi + 1 - head_in), _t_len, seq_fwd(ref_seq + _refi + 1 - head_in), gh0, gh0, tmp);
            cig_pushv(cx, out, tmp.c, tmp.n);
        } else {
            cig_push1(cx, out, (head_in << 4) | C_M);
            indel_cigar(cx, _refi, _readi, ref_len, read_len, g, &_clen, split_len, &res);
            if (_clen) cig_pushw(cx, out, g[0]);
        }
    } else {                                                              // no anchors, :807-819
        _t_len = ref_len; _q_len = read_len;
        if (_t_len < P->split_len && _q_len < P->split_len) HP_GLOBAL(_q_len, read_seq, _t_len, ref_seq);
        else res |= ksw_bi_extend(cx, _q_len, seq_fwd(read_seq), _t_len, seq_fwd(ref_seq), gh0, gh0, tmp);
        cig_pushv(cx, out, tmp.c, tmp.n);
    }
#undef HN_RI
#undef HN_OF
#undef HP_GLOBAL
    arena_release(cx.tmp, mark);
    return res;
}

}  // namespace hp
