// hp_chain.h -- sparse-DP chaining of seed hits into lines on one wavefront
// (reference src/lamsa_dp_con.c + src/lamsa_heap.c).
//
//   edge_flag_packed  <- get_fseed_dis          lamsa_dp_con.c:596-634 (gap_edge in hp_cluster.h: the same on 32-bit relative positions)
//   dp_update_range    <- frag_dp_update        :701   targets in order; candidates = the target's neighbours in the
//                                                      (contig, strand, position) order, spread over the 64 lanes
//   min_extend_all <- frag_min_extend           :1031  all MIN hits x all hits, blocked, records in registers
//   branch_track / cut_branch / best_son <- :873,:831,:808   (pointer chasing, wave-uniform)
//   reach_run      (no counterpart)             the hits that can be connected to an anchor at all: an exact run of the sorted order
//   mini_line      <- frag_mini_dp_line         :1068  mini_line_regs / mini_line_sets: the whole pass on registers;
//                                                      mini_line_mem: through memory, for passes with more than 256 listed hits
//   multi_line     <- frag_mini_dp_multi_line   :923
//   set_bound      <- line_set_bound(1)         :425,:496 (minus E_LB/E_RB, which nothing reads)
//   build_flines   <- frag_dp_path              :1152 (+ line_filter_overlap :568)
//   chain_first    <- frag_line_BCC             :1305
//   chain_remain   <- frag_line_remain          :1252
//
// Targets are processed strictly in the reference's order (the son_flag side effect of a
// chosen predecessor is visible to later targets, :718-720/:754), only the scan over the
// predecessors of one target is parallel: lanes evaluate 64 predecessors at a time and a
// wave reduction picks the winner by (score desc, NM asc, scan order asc), with the
// "first '-' strand match precursor wins outright" rule (:726-733) as a second reduction.
// Nodes are struct-of-arrays in the wave's HBM slab; sons are intrusive linked lists.
#pragma once
#include "hp_batch.h"
#include "hp_sort.h"

namespace hp {

enum { F_MATCH = 0, F_SPLIT_MATCH = 1, F_MISMATCH = 2, F_MATCH_THD = 2, F_LONG_MISMATCH = 3, F_INSERT = 4, F_DELETE = 5,
       F_CHR_DIF = 6, F_REVERSE = 7, F_UNCONNECT = 8, F_UNMATCH = 9, F_INIT = 20 };
enum { MIN_FLAG = 1, MULTI_FLAG = 2, UNLIMITED_FLAG = 3, WHOLE_FLAG = 4, TRACKED_FLAG = 5 };
enum { L_MERGB = 0, L_NMERG = 1, L_MERGH = 2, L_INTER = 4, L_DUMP = 8 };

HP_INL int score_table(int flag) { return flag <= 3 ? 1 : (flag <= 7 ? -3 : -6); }   // f_BCC_score_table, lamsa_aln.c:177

// limits of the packed reduction key in dp_update (checked on the host before launch)
#define HP_MAX_SLOTS 16383
#define HP_MAX_HITS_PER_SEED 16383

// Everything the predecessor scan needs about one seed hit, static and dynamic, in ONE 32-byte record: a candidate
// costs a single 32-B sector of HBM/L2 traffic and two 16-B loads instead of six scattered cache lines.
struct NodeS {
    int64_t pos;                // 1-based leftmost reference coordinate
    int32_t chr;                // contig id
    int32_t slot_j;             // seed slot << 14 | hit index within the seed
    int16_t sid;                // 1-based seed id (unflipped: chaining never runs while a '-' line is being filled)
    int8_t strand, len_dif8;    // +1/-1; len_dif (|len_dif| <= 127 always holds for 50-bp seeds)
    int8_t dp_flag;             // pass flag (MIN / MULTI / WHOLE / TRACKED, negative: unreachable)   -- dynamic
    uint8_t son_flag;           // edge class of the last chosen son (F_INIT: none)                   -- dynamic
    uint8_t match_flag;         // edge class to `from`                                                -- dynamic
    uint8_t pad_;
    int32_t score, NM;          // chain score / total NM up to this node                             -- dynamic
};

// one read being aligned by this wave
struct ReadCtx {
    Ctx cx;
    RefView ref;
    int L, seed_all, seed_out, last_len, H;
    const uint8_t *read;        // forward read, codes 0..4
    uint8_t *rc_read; bool rc_ready;   // reverse complement, filled on first use (frag_check.c:922-925)
    const uint8_t *cur_read;    // strand-appropriate read of the line being filled
    long long t_bases;          // reference bases fetched for this read (sum of DP target / NM window lengths): roofline accounting
    long long n_pairs;          // edge classifications (get_fseed_dis evaluations) executed for this read: accounting
    long long cs_words;         // seed-CIGAR words the fill has read (the hits on the read's lines): roofline accounting
    bool flip;                  // seed ids flipped (k -> seed_all+1-k) while a '-' line is filled (frag_check.c:926,953)
    const int32_t *seed_id;     // [seed_out]
    const int64_t *hit_off;     // [seed_out+1], global; local hit index = global - hb
    int64_t hb;
    const int64_t *h_pos, *h_cig_off; const int32_t *h_chr; const int16_t *h_nm, *h_len_dif;
    const int8_t *h_strand; const uint8_t *h_cig_n; const int32_t *cig;
    // chaining DP cells (frag_dp_node, lamsa_aln.h:352-373), indexed by local hit index
    int32_t *n_from, *n_in_de, *n_son_n, *n_first, *n_last, *n_next;
    int32_t *n_max_score, *n_max_NM, *n_max_node, *n_node_n, *n_seed;
    NodeS *nd;                  // 32-byte hot record per hit (static facts + score/NM/flags)
    const int32_t *srt, *rnk;   // hits sorted by (contig, strand, position) and the inverse permutation (local indices)
    bool nodes_ready;           // nodes_fill has written the initial chaining state of every hit (hp_align.h); chain_first need not
    HP_L int32_t *leaf_bits; bool leaf_on;   // while track_leaves runs (leaf_on): one bit per seed slot that may hold a hit to start a track from, in LDS
                                // (a flag of its own: the wave's LDS starts at offset 0, which is what a null LDS pointer compares equal to)
    long long *prof;            // diagnostic build only
};

HP_INL int hoff(const ReadCtx &r, int x) { return (int)(r.hit_off[x] - r.hb); }
HP_INL int mapn(const ReadCtx &r, int x) { return (int)(r.hit_off[x + 1] - r.hit_off[x]); }
HP_INL int sid(const ReadCtx &r, int x) { int s = r.seed_id[x]; return r.flip ? r.seed_all + 1 - s : s; }
HP_INL int nx(const ReadCtx &r, int node) { return node < 0 ? -1 : r.n_seed[node]; }

// ---------------------------------------------------------------- node helpers
HP_INL void node_set(ReadCtx &r, int n, int from, int score, int NM, int match_flag, int dp_flag)
{   // fnode_set, :636
    r.nd[n].son_flag = F_INIT; r.n_from[n] = from; r.nd[n].score = score; r.nd[n].NM = NM;
    r.nd[n].match_flag = (uint8_t)match_flag; r.nd[n].dp_flag = (int8_t)dp_flag;
    r.n_node_n[n] = 1; r.n_in_de[n] = 0; r.n_son_n[n] = 0; r.n_first[n] = -1; r.n_last[n] = -1;
    r.n_max_score[n] = score; r.n_max_NM[n] = NM; r.n_max_node[n] = n;
}
HP_INL void add_son(ReadCtx &r, int fa, int son)
{   // fnode_add_son, :683 (append keeps insertion order, which get_max_son depends on)
    ++r.n_in_de[fa];
    r.n_next[son] = -1;
    if (r.n_son_n[fa] == 0) r.n_first[fa] = son; else r.n_next[r.n_last[fa]] = son;
    r.n_last[fa] = son;
    ++r.n_son_n[fa];
}

HP_INL NodeS node_load(const HP_G NodeS *p)
{   // layout: pos[0..8) chr[8..12) slot_j[12..16) | sid[16..18) strand[18] len_dif8[19] dp_flag[20] son_flag[21] match_flag[22] pad[23] score[24..28) NM[28..32)
    int a[4], b[4];                         // two whole-vector loads of one 32-byte sector, see hp_load16
    hp_load16(p, a); hp_load16((const HP_G char *)p + 16, b);
    NodeS q;
    q.pos = (int64_t)(((unsigned long long)(unsigned)a[1] << 32) | (unsigned)a[0]); q.chr = a[2]; q.slot_j = a[3];
    q.sid = (int16_t)(b[0] & 0xffff); q.strand = (int8_t)((b[0] >> 16) & 0xff); q.len_dif8 = (int8_t)((b[0] >> 24) & 0xff);
    q.dp_flag = (int8_t)(b[1] & 0xff); q.son_flag = (uint8_t)((b[1] >> 8) & 0xff); q.match_flag = (uint8_t)((b[1] >> 16) & 0xff); q.pad_ = 0;
    q.score = b[2]; q.NM = b[3];
    return q;
}

// edge class from two packed records (same arithmetic as edge_flag; pre != cur, ids unflipped)
struct EdgeK { int seed_step, seed_len, match_dis, high_err, mis3, sv_len, half_split; };
HP_INL EdgeK edge_consts(const lamsa_hp_para *P)
{
    EdgeK k; k.seed_step = P->seed_step; k.seed_len = P->seed_len; k.match_dis = P->match_dis; k.high_err = P->aln_mode & 2;
    k.mis3 = 3 * P->mismatch_thd; k.sv_len = P->SV_len_thd; k.half_split = P->split_len / 2;
    return k;
}
HP_INL int edge_flag_packed(const EdgeK &k, const NodeS &pre, const NodeS &cur)
{
    if ((pre.slot_j >> 14) == (cur.slot_j >> 14)) return F_UNCONNECT;
    const int sp = pre.strand;
    if (cur.chr != pre.chr || cur.strand != sp) return F_CHR_DIF;
    const int idp = pre.sid, idc = cur.sid, did = iabs(idp - idc);
    if (did * k.seed_step < k.seed_len) return F_UNCONNECT;
    const int64_t exp = pre.pos + (int64_t)(sp * (idc - idp) * k.seed_step);
    const int64_t act = cur.pos;
    const int dis = (int)((int64_t)sp * ((idp < idc) ? (act - exp) : (exp - act)) - ((sp * (idp - idc) < 0) ? pre.len_dif8 : cur.len_dif8));
    const int mat_dis = k.match_dis * (k.high_err ? did : 1);
    if (dis <= mat_dis && dis >= -mat_dis) return did == 1 ? F_MATCH : (did <= k.mis3 ? F_MISMATCH : F_LONG_MISMATCH);
    if (dis > mat_dis && dis < k.sv_len) return F_DELETE;
    if ((dis < -mat_dis && dis >= 0 - (did * k.seed_step - k.seed_len)) || (dis < -k.half_split && dis >= -k.sv_len)) return F_INSERT;
    return F_UNCONNECT;
}

// frag_dp_per_init over a whole range of nodes, one node per lane.
// which == 0: nodes whose dp_flag is +-dp_flag (frag_mini_dp_line, :1086-1091); which == 1: every node that is not TRACKED (:946-951)
// rlo..rhi: when from < 0, only nodes whose rank in the (contig, strand, position) order lies in [rlo, rhi] become
// active; the others are marked unreachable exactly like nodes that cannot be connected to `from` (see mini_line).
HP_NOINL void nodes_per_init(ReadCtx &r, int k0, int k1, int from, int dp_flag, int which, int rlo = 0, int rhi = 0x7fffffff)
{
#ifdef HP_PROF
    if (r.prof) { r.prof[54] += k1 - k0; r.prof[55] += 1; }
#endif
#ifdef HP_PROF
    const long long t0_ = wv::clock();
#endif
    const HP_G NodeS *ns = (const HP_G NodeS *)r.nd;
    const EdgeK K = edge_consts(r.cx.P);
    NodeS F; F.pos = 0; F.chr = 0; F.slot_j = 0; F.sid = 0; F.strand = 0; F.len_dif8 = 0; F.pad_ = 0; F.dp_flag = 0; F.son_flag = 0; F.match_flag = 0; F.score = 0; F.NM = 0;
    int from_nm = 0;
    if (from >= 0) { F = node_load(ns + from); from_nm = r.h_nm[from]; r.n_pairs += k1 > k0 ? k1 - k0 : 0; }
    for (int b = k0; b < k1; b += 64) {
        WAVE_FOR(l) {
            const int k = b + l;
            if (k < k1) {
                const int df = r.nd[k].dp_flag;
                const bool take = which == 0 ? (df == dp_flag || df == 0 - dp_flag) : (df != TRACKED_FLAG);
                if (take) {
                    if (from < 0) {
                        const int rk = rhi == 0x7fffffff ? 0 : r.rnk[k];
                        if (rk >= rlo && rk <= rhi) node_set(r, k, from, 1, r.h_nm[k], F_MATCH, dp_flag);
                        else r.nd[k].dp_flag = (int8_t)(0 - dp_flag);
                    } else {
                        const NodeS Q = node_load(ns + k);
                        const int flag = k == from ? F_MATCH : edge_flag_packed(K, F, Q);
                        if (flag != F_UNCONNECT && flag != F_CHR_DIF) node_set(r, k, from, 2 + score_table(flag), r.h_nm[k] + from_nm, flag, dp_flag);
                        else r.nd[k].dp_flag = (int8_t)(0 - dp_flag);
                    }
                }
            }
        }
    }
    wv::sync();
#ifdef HP_PROF
    if (r.prof) r.prof[13] += wv::clock() - t0_;
#endif
}

// ---------------------------------------------------------------- frag_dp_update, :701-764, over a range of targets
// Targets k0..k1-1 are updated strictly in order (the son_flag side effect of a chosen predecessor is visible to
// the next target).  `force`: update node k0 whatever its dp_flag (the right anchor of a mini-DP, :1129); otherwise
// only nodes whose dp_flag equals `dp_flag`.
//
// Latency structure (this loop is where most of the read's time goes; every dependent HBM/L2 round trip counts):
//   * the 64 targets of a chunk load their own records once, one per lane; a target's record then comes out of
//     those registers by readlane (a target's record is never written by an earlier target of the pass);
//   * candidates are the target's neighbours in the (contig, strand, position) order (see below); the sort-index
//     entries of the NEXT target are fetched while the current one is evaluated, and the next target's candidate
//     records are requested right after the current target's own stores, so that they travel together with the
//     son-list bookkeeping loads of the current target;
//   * every lane remembers node id and edge class of its running best, so the winner is read out of a lane
//     instead of being re-fetched.
// One dependent round trip per target instead of seven.
// `sons`: maintain the son lists (fnode_add_son).  They are read only by branch tracking, which runs after the main
// pass of the first round and after the pass of frag_mini_dp_multi_line; both start from nodes that fnode_set has just
// reset, so the lists a frag_mini_dp_line pass would leave behind are never read and that pass skips them.
struct ScanT { NodeS T; int tkey, x, t_NM; long long Rw; };

HP_INL NodeS node_unpack(const int *a, const int *b)
{
    NodeS q;
    q.pos = (int64_t)(((unsigned long long)(unsigned)a[1] << 32) | (unsigned)a[0]); q.chr = a[2]; q.slot_j = a[3];
    q.sid = (int16_t)(b[0] & 0xffff); q.strand = (int8_t)((b[0] >> 16) & 0xff); q.len_dif8 = (int8_t)((b[0] >> 24) & 0xff);
    q.dp_flag = (int8_t)(b[1] & 0xff); q.son_flag = (uint8_t)((b[1] >> 8) & 0xff); q.match_flag = (uint8_t)((b[1] >> 16) & 0xff); q.pad_ = 0;
    q.score = b[2]; q.NM = b[3];
    return q;
}

// one candidate predecessor Q (node id p) against target T; per-lane running bests are updated in place
HP_INL void scan_eval(const EdgeK &K, const ScanT &S, const NodeS &Q, int p, int inb, int start_slot, int dp_flag,
                      long long &key, int &bp, int &bf, int &negp, int &n_p, int &n_f, int &n_c, int &n_n, int &outw, int &oka)
{
    const int POSMAX = (1 << 28) - 1;
    long long dp = Q.pos - S.T.pos; if (dp < 0) dp = -dp;
    const int inwin = inb & ((Q.chr * 2 + (Q.strand > 0 ? 1 : 0)) == S.tkey) & (dp <= S.Rw);
    outw = !inwin;
    const int qslot = Q.slot_j >> 14;
    const int flag = edge_flag_packed(K, Q, S.T);
    const int ok = inwin & (qslot >= start_slot) & (qslot < S.x) & (Q.dp_flag == dp_flag) & !((Q.strand == 1) & (Q.son_flag <= F_MATCH_THD)) &
                   (flag != F_UNCONNECT) & (flag != F_CHR_DIF);
    const int pos = (int)((unsigned)(S.x - 1 - qslot) << 14) | (Q.slot_j & 16383);            // scan order: seeds descending, hits ascending
    const int cand = Q.score + 1 + score_table(flag);
    const int nm = Q.NM + S.t_NM;
    const int isneg = ok & (Q.strand == -1) & (flag <= F_MATCH_THD);           // '-': first match precursor wins, :726-733
    const long long k = ((long long)(cand + 32768) << 47) | ((long long)(524287 - nm) << 28) | (long long)(POSMAX - pos);
    const long long kk = ok ? k : -1;
    const bool better = kk > key;
    key = better ? kk : key; bp = better ? p : bp; bf = better ? flag : bf;
    const int np = isneg ? -pos : -0x7fffffff;
    const bool nb = np > negp;
    negp = nb ? np : negp; n_p = nb ? p : n_p; n_f = nb ? flag : n_f; n_c = nb ? cand : n_c; n_n = nb ? nm : n_n;
    oka |= ok;
}

HP_NOINL void dp_update_range(ReadCtx &r, int k0, int k1, int start_slot, int dp_flag, bool force, bool sons, const uint8_t *only = nullptr)
{
    const HP_G uint8_t *g_only = (const HP_G uint8_t *)only;     // when given: only hits flagged here are targets (their clusters did not fit LDS, hp_cluster.h)
    const HP_G NodeS *ns = (const HP_G NodeS *)r.nd;
    HP_G NodeS *gd = (HP_G NodeS *)r.nd;
    HP_G int32_t *g_from = (HP_G int32_t *)r.n_from;
    HP_G int32_t *g_node_n = (HP_G int32_t *)r.n_node_n, *g_in_de = (HP_G int32_t *)r.n_in_de, *g_son_n = (HP_G int32_t *)r.n_son_n;
    HP_G int32_t *g_first = (HP_G int32_t *)r.n_first, *g_last = (HP_G int32_t *)r.n_last, *g_next = (HP_G int32_t *)r.n_next;
    const EdgeK K = edge_consts(r.cx.P);
    const HP_G int32_t *g_srt = (const HP_G int32_t *)r.srt, *g_rnk = (const HP_G int32_t *)r.rnk;
    const int sid_lo = start_slot < r.seed_out ? r.seed_id[start_slot] : 0;
    const int H = r.H;
#ifdef HP_PROF
    const long long tu0_ = wv::clock();
    if (r.prof) { r.prof[13] += 1; }
#endif
    for (int tb = k0; tb < k1; tb += 64) {
#ifdef HP_PROF
        const long long tp0_ = wv::clock();
#endif
        // ---- one target per lane: its record, rank and predecessor; is it due in this pass, and does it have any
        // neighbour (in sorted order) inside its window?  A target without one has no connectable predecessor at all
        // and keeps its state (exact skip).
        wv::Lane<int> Ta0, Ta1, Ta2, Ta3, Tb0, Tb1, Tb2, Tb3, Trk, Tfrom, c;
        WAVE_FOR(l) {
            const int k = tb + l, kk = k < k1 ? k : k0;
            int a[4], b[4];
            hp_load16(ns + kk, a); hp_load16((const HP_G char *)(ns + kk) + 16, b);
            const int rk = g_rnk[kk];
            Ta0[l] = a[0]; Ta1[l] = a[1]; Ta2[l] = a[2]; Ta3[l] = a[3]; Tb0[l] = b[0]; Tb1[l] = b[1]; Tb2[l] = b[2]; Tb3[l] = b[3];
            Trk[l] = rk; Tfrom[l] = g_from[kk];
            int due = 0;
            if (!force) {
                const NodeS Tk = node_unpack(a, b);
                if (k < k1 && Tk.dp_flag == dp_flag && g_only) due = g_only[kk];
                else if (k < k1 && Tk.dp_flag == dp_flag) {
                    const int dm = Tk.sid - sid_lo;
                    const int mdm_ = K.match_dis * (K.high_err ? dm : 1);
                    long long Rk = K.sv_len > dm * K.seed_step ? K.sv_len : dm * K.seed_step;
                    if (mdm_ + 1 > Rk) Rk = mdm_ + 1;
                    Rk += 128 + (long long)dm * K.seed_step;
                    const int tk_ = Tk.chr * 2 + (Tk.strand > 0 ? 1 : 0);
#pragma unroll
                    for (int d = -1; d <= 1; d += 2) {
                        const int i2 = rk + d;
                        if (i2 >= 0 && i2 < H) {
                            const NodeS Nb = node_load(ns + g_srt[i2]);
                            long long dp_ = Nb.pos - Tk.pos; if (dp_ < 0) dp_ = -dp_;
                            due |= ((Nb.chr * 2 + (Nb.strand > 0 ? 1 : 0)) == tk_) & (dp_ <= Rk);
                        }
                    }
                }
            }
            c[l] = due;
        }
        unsigned long long todo = force ? 1ull : wv::ballot(c);
#ifdef HP_PROF
        if (r.prof) { r.prof[13] += 1; }
#endif
        if (!todo) continue;
#ifdef HP_PROF
        const long long tq0_ = wv::clock();
#endif
        // ---- software pipeline over the due targets of the chunk
        int li = __builtin_ctzll(todo);
        wv::Lane<int> pn0, pn1;                          // sort-index entries of the current target, first trip
        wv::Lane<int> Qa[2][4], Qb[2][4];                // and the candidate records they name
        {
            const int rT = wv::bcast(Trk, li);
            WAVE_FOR(l) {
                const int i0 = rT - 1 - l, i1 = rT + 1 + l;
                pn0[l] = g_srt[i0 >= 0 ? i0 : rT]; pn1[l] = g_srt[i1 < H ? i1 : rT];
            }
            WAVE_FOR(l) {
                int a[4], b[4];
                hp_load16(ns + pn0[l], a); hp_load16((const HP_G char *)(ns + pn0[l]) + 16, b);
#pragma unroll
                for (int q = 0; q < 4; ++q) { Qa[0][q][l] = a[q]; Qb[0][q][l] = b[q]; }
                hp_load16(ns + pn1[l], a); hp_load16((const HP_G char *)(ns + pn1[l]) + 16, b);
#pragma unroll
                for (int q = 0; q < 4; ++q) { Qa[1][q][l] = a[q]; Qb[1][q][l] = b[q]; }
            }
        }
#ifdef HP_PROF
        if (r.prof) r.prof[13] += 1;
#endif
        for (;;) {
            const int t = tb + li;
#ifdef HP_PROF
            if (r.prof) r.prof[11] += 1;
#endif
            ScanT S;
            {
                int a[4] = { wv::bcast(Ta0, li), wv::bcast(Ta1, li), wv::bcast(Ta2, li), wv::bcast(Ta3, li) };
                int b[4] = { wv::bcast(Tb0, li), wv::bcast(Tb1, li), wv::bcast(Tb2, li), wv::bcast(Tb3, li) };
                S.T = node_unpack(a, b);
            }
            const int rT = wv::bcast(Trk, li), t_from = wv::bcast(Tfrom, li);
            S.x = S.T.slot_j >> 14; S.t_NM = S.T.NM; S.tkey = S.T.chr * 2 + (S.T.strand > 0 ? 1 : 0);
            // Predecessors that get_fseed_dis can connect at all lie on the same contig and strand within R bases of
            // the target (|dis| < max(SV_len_thd, did*step, mat_dis) and |act-exp| <= |dis|+|len_dif|, exp within
            // did*step of the predecessor).  In the (contig, strand, position) order they are the neighbours of the
            // target itself, so the scan walks outwards from the target's rank, 64 hits per trip and direction,
            // and stops at the first hit outside the window.  Everything else is F_CHR_DIF / F_UNCONNECT for the
            // reference too, hence the result is unchanged.
            {
                const int did_max = S.T.sid - sid_lo;
                const int mdm = K.match_dis * (K.high_err ? did_max : 1);
                long long Rw = K.sv_len > did_max * K.seed_step ? K.sv_len : did_max * K.seed_step;
                if (mdm + 1 > Rw) Rw = mdm + 1;
                S.Rw = Rw + 128 + (long long)did_max * K.seed_step;
            }
            todo &= todo - 1;
            const bool has_next = todo != 0;
            const int li2 = has_next ? __builtin_ctzll(todo) : li;
            wv::Lane<int> pm0, pm1;                      // prefetch: the next target's sort-index entries
            if (has_next) {
                const int rT2 = wv::bcast(Trk, li2);
                WAVE_FOR(l) {
                    const int i0 = rT2 - 1 - l, i1 = rT2 + 1 + l;
                    pm0[l] = g_srt[i0 >= 0 ? i0 : rT2]; pm1[l] = g_srt[i1 < H ? i1 : rT2];
                }
            }
            r.n_pairs += (rT < 64 ? rT : 64) + (H - 1 - rT < 64 ? H - 1 - rT : 64);
            wv::Lane<long long> key;
            wv::Lane<int> bp, bf, negp, n_p, n_f, n_c, n_n, out0, out1, okl;
            WAVE_FOR(l) {                                // first trip: the records are already here
#ifdef HP_PROF
                if (l == 0 && r.prof) r.prof[13] += 1;
#endif
                key[l] = -1; bp[l] = 0; bf[l] = 0; negp[l] = -0x7fffffff; n_p[l] = 0; n_f[l] = 0; n_c[l] = 0; n_n[l] = 0;
                int ow[2], oka = 0;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    int a[4], b[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { a[q] = Qa[u][q][l]; b[q] = Qb[u][q][l]; }
                    const NodeS Q = node_unpack(a, b);
                    const int idx = u ? rT + 1 + l : rT - 1 - l;
                    scan_eval(K, S, Q, u ? pn1[l] : pn0[l], idx >= 0 && idx < H, start_slot, dp_flag,
                              key[l], bp[l], bf[l], negp[l], n_p[l], n_f[l], n_c[l], n_n[l], ow[u], oka);
                }
                out0[l] = ow[0]; out1[l] = ow[1]; okl[l] = oka;
            }
            // sorted order: once a hit is outside the window, all farther ones on that side are
            bool live0 = wv::ballot(out0) == 0, live1 = wv::ballot(out1) == 0;
            int any_ok = wv::ballot(okl) != 0;
            for (int cc = 1; live0 || live1; ++cc) {     // wide windows (repeat clusters): further trips, fetched directly
#ifdef HP_PROF
                if (r.prof) r.prof[13] += 1;
#endif
                {   const int lo_left = rT - cc * 64, hi_left = H - 1 - rT - cc * 64;
                    r.n_pairs += (live0 ? (lo_left < 0 ? 0 : (lo_left < 64 ? lo_left : 64)) : 0) + (live1 ? (hi_left < 0 ? 0 : (hi_left < 64 ? hi_left : 64)) : 0); }
                WAVE_FOR(l) {
                    int idx[2], inb[2], pn[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        idx[u] = rT + (u ? 1 : -1) * (1 + cc * 64 + l);
                        inb[u] = (u ? live1 : live0) && idx[u] >= 0 && idx[u] < H;
                        pn[u] = g_srt[inb[u] ? idx[u] : rT];
                    }
                    NodeS Q[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) Q[u] = node_load(ns + pn[u]);
                    int ow[2], oka = 0;
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        scan_eval(K, S, Q[u], pn[u], inb[u], start_slot, dp_flag, key[l], bp[l], bf[l], negp[l], n_p[l], n_f[l], n_c[l], n_n[l], ow[u], oka);
                    out0[l] = ow[0]; out1[l] = ow[1]; okl[l] = oka;
                }
                if (live0 && wv::ballot(out0) != 0) live0 = false;
                if (live1 && wv::ballot(out1) != 0) live1 = false;
                any_ok |= wv::ballot(okl) != 0;
            }
            int max_from = t_from, max_score = S.T.score, max_NM = S.t_NM, max_flag = 0;
            bool changed = false;
            if (any_ok) {                                 // otherwise no connectable predecessor: the node keeps its state
                const int npos = wv::reduce_max(negp);
                if (npos != -0x7fffffff) {                // '-' strand: the first match precursor in scan order
                    wv::Lane<int> w;
                    WAVE_FOR(l) w[l] = negp[l] == npos;
                    const int wl = __builtin_ctzll(wv::ballot(w));
                    max_from = wv::bcast(n_p, wl); max_flag = wv::bcast(n_f, wl); max_score = wv::bcast(n_c, wl); max_NM = wv::bcast(n_n, wl);
                    changed = max_from != t_from;
                } else {
                    const long long best_key = wv::reduce_max64(key);
                    if (best_key >= 0) {
                        const int nm = 524287 - (int)((best_key >> 28) & 524287);
                        const int cand = (int)(best_key >> 47) - 32768;
                        if (cand > max_score || (cand == max_score && nm < max_NM)) {
                            wv::Lane<int> w;
                            WAVE_FOR(l) w[l] = key[l] == best_key;
                            const int wl = __builtin_ctzll(wv::ballot(w));
                            max_from = wv::bcast(bp, wl); max_flag = wv::bcast(bf, wl); max_score = cand; max_NM = nm;
                            changed = max_from != t_from;
                        }
                    }
                }
            }
            if (changed) {                               // wave-uniform stores (:753-761); every lane re-reads only what it wrote itself
                gd[max_from].son_flag = (uint8_t)max_flag;
                g_from[t] = max_from; gd[t].score = max_score; gd[t].NM = max_NM; gd[t].match_flag = (uint8_t)max_flag;
            }
            if (has_next) {                              // the next target's candidates, after the stores they may have to see
                WAVE_FOR(l) {
                    int a[4], b[4];
                    hp_load16(ns + pm0[l], a); hp_load16((const HP_G char *)(ns + pm0[l]) + 16, b);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { Qa[0][q][l] = a[q]; Qb[0][q][l] = b[q]; }
                    hp_load16(ns + pm1[l], a); hp_load16((const HP_G char *)(ns + pm1[l]) + 16, b);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { Qa[1][q][l] = a[q]; Qb[1][q][l] = b[q]; }
                    pn0[l] = pm0[l]; pn1[l] = pm1[l];
                }
            }
            if (changed) {
                g_node_n[t] = g_node_n[max_from] + 1;
            }
            if (changed && sons) {
                const int sn = g_son_n[max_from], la = g_last[max_from];      // fnode_add_son, :683
                g_in_de[max_from] = g_in_de[max_from] + 1;
                g_next[t] = -1;
                if (sn == 0) g_first[max_from] = t; else g_next[la] = t;
                g_last[max_from] = t;
                g_son_n[max_from] = sn + 1;
            }
            if (!has_next) break;
            li = li2;
        }
    }
#ifdef HP_PROF
    if (r.prof) r.prof[13] += 1;
#endif
}
HP_INL void dp_update(ReadCtx &r, int t, int start_slot, int dp_flag, bool sons) { dp_update_range(r, t, t + 1, start_slot, dp_flag, true, sons); }

// ---------------------------------------------------------------- frag_min_extend for every MIN hit, :1031-1066, :1335-1343
// For every MIN hit m (a hit of a seed with at most min_n hits) and every seed with more than min_n hits, the first
// hit (ascending) of that seed that is match-class colinear with m joins the MIN pass.  The result is a set union, so
// the order over MIN hits is free: the hits are walked in chunks of 64 (one coalesced load per chunk, records kept in
// registers), the MIN hits of a chunk are broadcast out of the lanes that hold them, and "first within its seed" is a
// segmented ballot.  No dependent memory round trip inside the pair loop.
HP_NOINL void min_extend_all(ReadCtx &r, int min_n)
{
    const HP_G NodeS *ns = (const HP_G NodeS *)r.nd;
    HP_G NodeS *gd = (HP_G NodeS *)r.nd;
    const EdgeK K = edge_consts(r.cx.P);
    const int H = r.H;
    // Which seeds have at most min_n hits is read off the pass flag the caller has just assigned (MIN / MULTI, :1315-1334);
    // the hits that join are collected in `mark` and flagged at the end, so that the flags stay what they were while the
    // pairs are evaluated.  A hit's index within its seed (slot_j) gives the seed's first hit without a table lookup.
    const size_t mark_ = arena_mark(r.cx.tmp);
    uint8_t *mk = (uint8_t *)arena_alloc(r.cx, (size_t)H + 64);
    if (!mk) return;
    HP_G uint8_t *gmk = (HP_G uint8_t *)mk;
    for (int b0 = 0; b0 < H; b0 += 64) { WAVE_FOR(l) { if (b0 + l < H) gmk[b0 + l] = 0; } }
    wv::sync();
    for (int mbase = 0; mbase < H; mbase += 64) {
        // the MIN hits of this chunk, one per lane
        wv::Lane<int> Ma0, Ma1, Ma2, Ma3, Mb0, ism;
        WAVE_FOR(l) {
            const int k = mbase + l, kk = k < H ? k : H - 1;
            int a[4], b[4];
            hp_load16(ns + kk, a); hp_load16((const HP_G char *)(ns + kk) + 16, b);
            Ma0[l] = a[0]; Ma1[l] = a[1]; Ma2[l] = a[2]; Ma3[l] = a[3]; Mb0[l] = b[0];
            ism[l] = k < H && (int)(int8_t)(b[1] & 0xff) == MIN_FLAG;
        }
        const unsigned long long mset = wv::ballot(ism);
        if (!mset) continue;
        unsigned long long carry = 0;                     // bit j: MIN hit j already found its hit in seed carry_seed
        int carry_seed = -1;
        for (int base = 0; base < H; base += 64) {
            wv::Lane<int> seg, sd, elig, hit, qkey, qsid, qld;
            wv::Lane<long long> qdiag;          // position minus the seed's offset on the read: colinear hits share it
            WAVE_FOR(l) {
                const int k = base + l, kk = k < H ? k : H - 1;
                int a[4], b[4];
                hp_load16(ns + kk, a); hp_load16((const HP_G char *)(ns + kk) + 16, b);
                const int s = a[3] >> 14;
                const int st = kk - (a[3] & 16383) - base;                 // first hit of this hit's seed, relative to the chunk
                seg[l] = st > 0 ? st : 0; sd[l] = k < H ? s : -1;
                elig[l] = k < H && (int)(int8_t)(b[1] & 0xff) == MULTI_FLAG;
                hit[l] = 0;
                const int sid_ = (int)(int16_t)(b[0] & 0xffff), st_ = (int)(int8_t)((b[0] >> 16) & 0xff);
                qkey[l] = a[2] * 2 + (st_ > 0 ? 1 : 0); qsid[l] = sid_; qld[l] = (int)(int8_t)((b[0] >> 24) & 0xff);
                qdiag[l] = (long long)(((unsigned long long)(unsigned)a[1] << 32) | (unsigned)a[0]) - (long long)(st_ * sid_ * K.seed_step);
            }
            const int last = H - 1 - base < 63 ? H - 1 - base : 63;
            r.n_pairs += (long long)__builtin_popcountll(mset) * (last + 1);
            const int s_last = wv::bcast(sd, last), st_last = wv::bcast(seg, last);
            unsigned long long next_carry = 0;
            for (unsigned long long mm = mset; mm; mm &= mm - 1) {
                const int j = __builtin_ctzll(mm);
                // M's record out of lane j; the edge class is evaluated directly on diagonals (position minus the seed's
                // offset on the read): for two hits on the same contig and strand, get_fseed_dis' act - exp is the
                // difference of their diagonals (:607-612), so "match class" (:614-619) is two subtractions and a compare.
                const int ma0 = wv::bcast(Ma0, j), ma1 = wv::bcast(Ma1, j), ma2 = wv::bcast(Ma2, j), ma3 = wv::bcast(Ma3, j), mb0 = wv::bcast(Mb0, j);
                const int msid = (int)(int16_t)(mb0 & 0xffff), mst = (int)(int8_t)((mb0 >> 16) & 0xff), mld = (int)(int8_t)((mb0 >> 24) & 0xff);
                const int xm = ma3 >> 14;
                const int mkey = ma2 * 2 + (mst > 0 ? 1 : 0);
                const long long mdiag = (long long)(((unsigned long long)(unsigned)ma1 << 32) | (unsigned)ma0) - (long long)(mst * msid * K.seed_step);
                wv::Lane<int> q;
                WAVE_FOR(l) {
                    int v = 0;
                    if (elig[l] && qkey[l] == mkey && sd[l] != xm) {
                        const bool q_first = sd[l] < xm;                     // the hit of the earlier seed is `pre`
                        const int did = iabs(qsid[l] - msid);
                        const long long D = q_first ? mdiag - qdiag[l] : qdiag[l] - mdiag;     // act - exp
                        const int ld = mst > 0 ? (q_first ? qld[l] : mld) : (q_first ? mld : qld[l]);
                        const int dis = (int)((long long)mst * D - (long long)ld);
                        const int mat_dis = K.match_dis * (K.high_err ? did : 1);
                        v = did * K.seed_step >= K.seed_len && dis <= mat_dis && dis >= -mat_dis;
                    }
                    q[l] = v;
                }
                const unsigned long long qb = wv::ballot(q);
                if (!qb) { if (carry_seed == s_last && ((carry >> j) & 1)) next_carry |= 1ull << j; continue; }
                const bool cj = (carry >> j) & 1;
                WAVE_FOR(l) {
                    if (q[l]) {
                        const unsigned long long earlier = qb & ((1ull << l) - 1) & ~((1ull << seg[l]) - 1);
                        if (earlier == 0 && !(sd[l] == carry_seed && cj)) hit[l] = 1;
                    }
                }
                if ((qb >> st_last) != 0 || (carry_seed == s_last && cj)) next_carry |= 1ull << j;
            }
            WAVE_FOR(l) { if (hit[l]) gmk[base + l] = 1; }
            carry = next_carry; carry_seed = s_last;
        }
    }
    wv::sync();
    for (int b0 = 0; b0 < H; b0 += 64) { WAVE_FOR(l) { const int k = b0 + l; if (k < H && gmk[k]) gd[k].dp_flag = MIN_FLAG; } }
    wv::sync();
    arena_release(r.cx.tmp, mark_);
}

// ---------------------------------------------------------------- end-node stack / bounded heaps (lamsa_heap.c)
struct NScore { int32_t *node, *score, *NM; int min_score_thd, max_n, node_n, cap; };

HP_INL bool ns_alloc(Ctx &cx, NScore &ns, int cap, int max_n)
{
    ns.node = (int32_t *)arena_alloc(cx, sizeof(int32_t) * (size_t)(cap + 1));
    ns.score = (int32_t *)arena_alloc(cx, sizeof(int32_t) * (size_t)(cap + 1));
    ns.NM = (int32_t *)arena_alloc(cx, sizeof(int32_t) * (size_t)(cap + 1));
    ns.cap = cap; ns.max_n = max_n; ns.node_n = 0; ns.min_score_thd = 0;
    return ns.node && ns.score && ns.NM;
}
HP_INL int ns_pop(NScore &ns, int *score, int *NM)
{   // node_pop (LIFO), lamsa_heap.c:5; returns -1 when empty
    if (ns.node_n < 1) return -1;
    --ns.node_n;
    *score = ns.score[ns.node_n]; *NM = ns.NM[ns.node_n];
    return ns.node[ns.node_n];
}
HP_INL void ns_swap(NScore &ns, int a, int b)
{
    int t = ns.node[a]; ns.node[a] = ns.node[b]; ns.node[b] = t;
    t = ns.score[a]; ns.score[a] = ns.score[b]; ns.score[b] = t;
    t = ns.NM[a]; ns.NM[a] = ns.NM[b]; ns.NM[b] = t;
}
HP_FN void ns_min_sift(NScore &ns, int i)
{   // node_min_heap: score ascending, NM descending, lamsa_heap.c:151
    for (;;) {
        int l = 2 * i + 1, rr = 2 * (i + 1), m = i;
        if (l < ns.node_n && (ns.score[l] < ns.score[i] || (ns.score[l] == ns.score[i] && ns.NM[l] > ns.NM[i]))) m = l;
        if (rr < ns.node_n && (ns.score[rr] < ns.score[m] || (ns.score[rr] == ns.score[m] && ns.NM[rr] > ns.NM[m]))) m = rr;
        if (m == i) return;
        ns_swap(ns, i, m); i = m;
    }
}
HP_FN void ns_minpos_sift(NScore &ns, int i)
{   // node_minpos_heap, lamsa_heap.c:100
    for (;;) {
        int l = 2 * i + 1, rr = 2 * (i + 1), m = i;
        if (l < ns.node_n && ns.node[l] < ns.node[i]) m = l;
        if (rr < ns.node_n && ns.node[rr] < ns.node[m]) m = rr;
        if (m == i) return;
        ns_swap(ns, i, m); i = m;
    }
}
HP_FN int ns_extract_minpos(NScore &ns)
{   // lamsa_heap.c:126
    if (ns.node_n < 1) return -1;
    int m = ns.node[0];
    --ns.node_n;
    ns.node[0] = ns.node[ns.node_n]; ns.score[0] = ns.score[ns.node_n]; ns.NM[0] = ns.NM[ns.node_n];
    ns_minpos_sift(ns, 0);
    return m;
}
// heap_add_node, lamsa_dp_con.c:44: -1 stored, -2 rejected, else the evicted line index
HP_FN int ns_add_bounded(NScore &ns, int node, int score, int NM)
{
    if (ns.node_n < ns.max_n) {
        ns.score[ns.node_n] = score; ns.NM[ns.node_n] = NM; ns.node[ns.node_n++] = node;
        if (ns.node_n == ns.max_n) for (int i = (ns.node_n - 1) / 2; i >= 0; --i) ns_min_sift(ns, i);
        return -1;
    }
    if (ns.score[0] < score || (ns.score[0] == score && ns.NM[0] > NM)) {
        int ret = ns.node[0];
        ns.score[0] = score; ns.NM[0] = NM; ns.node[0] = node;
        ns_min_sift(ns, 0);
        return ret;
    }
    return -2;
}
// path (may be null): the ancestors of `node` in lane order, n_path <= 64 of them, when the caller has just walked them -- they are
// then marked with one store instead of being chased through n_from again
HP_FN void ns_add_end(ReadCtx &r, NScore &ns, int score, int NM, int node, const wv::Lane<int> *path = nullptr, int n_path = 0)
{   // node_add_score, lamsa_dp_con.c:786
    if (score < ns.min_score_thd) return;
    if (ns.node_n >= ns.cap) { r.cx.status |= ST_OVERFLOW; return; }
    ns.score[ns.node_n] = score; ns.NM[ns.node_n] = NM; ns.node[ns.node_n++] = node;
    r.nd[node].dp_flag = TRACKED_FLAG;
    if (path) { HP_G NodeS *gd = (HP_G NodeS *)r.nd; WAVE_FOR(l) { if (l < n_path) gd[(*path)[l]].dp_flag = TRACKED_FLAG; } return; }
    for (int t = r.n_from[node]; t >= 0; t = r.n_from[t]) r.nd[t].dp_flag = TRACKED_FLAG;
}

// ---------------------------------------------------------------- forest -> disjoint paths
HP_FN int best_son(ReadCtx &r, int f)
{   // get_max_son, :808
    int max_score = 0, max_NM = 0, max_dis = 0, flag_thd = F_INIT, max = -1;
    const int x = r.n_seed[f];
    for (int s = r.n_first[f], c = 0; c < r.n_son_n[f] && s >= 0; s = r.n_next[s], ++c) {
        const int mf = r.nd[s].match_flag, sx = r.n_seed[s];
        if (mf <= flag_thd && (r.n_max_score[s] > max_score || (r.n_max_score[s] == max_score && (sx - x < max_dis || r.n_max_NM[s] < max_NM)))) {
            max = s; max_score = r.n_max_score[s]; max_NM = r.n_max_NM[s]; max_dis = sx - x;
            if (mf <= F_MATCH_THD) flag_thd = F_MATCH_THD;
        }
    }
    return max;
}
HP_FN void detach(ReadCtx &r, int s, int max_node, NScore &ns)
{   // :842-847 / :851-857 / :893-899
    r.n_from[s] = -1;
    r.n_max_score[s] -= (r.nd[s].score - 1);
    r.n_max_NM[s] -= (r.nd[s].NM - r.h_nm[s]);
    r.n_node_n[max_node] -= (r.n_node_n[s] - 1);
    ns_add_end(r, ns, r.n_max_score[s], r.n_max_NM[s], max_node);
}
HP_INL void leaf_mark(ReadCtx &r, int f)
{   // see track_leaves: the driver only visits seeds whose bit is set
    if (r.leaf_on) { const int x = r.n_seed[f]; r.leaf_bits[x >> 5] |= (int)(1u << (x & 31)); }
}
HP_FN void cut_branch(ReadCtx &r, int f, NScore &ns)
{   // :831-870
    const int keep = best_son(r, f);
    if (keep < 0) { r.cx.status |= ST_REFEXIT; r.n_in_de[f] = 0; leaf_mark(r, f); return; }
    for (int s = r.n_first[f], c = 0, nn = r.n_son_n[f]; c < nn && s >= 0; ++c) {
        const int nxt = r.n_next[s];
        if (s != keep) detach(r, s, r.n_max_node[s], ns);
        s = nxt;
    }
    if (r.nd[f].score > r.n_max_score[keep]) {        // negative edge
        r.n_in_de[keep] = -1;
        detach(r, keep, r.n_max_node[keep], ns);
        r.n_son_n[f] = 0; r.n_first[f] = r.n_last[f] = -1;
        r.n_max_node[f] = f; r.n_max_score[f] = r.nd[f].score; r.n_max_NM[f] = r.nd[f].NM;
    } else {
        r.n_son_n[f] = 1; r.n_first[f] = r.n_last[f] = keep; r.n_next[keep] = -1;
        r.n_max_node[f] = r.n_max_node[keep]; r.n_max_score[f] = r.n_max_score[keep]; r.n_max_NM[f] = r.n_max_NM[keep];
    }
    r.n_in_de[f] = 0;
    leaf_mark(r, f);                                  // f is complete: a track starts from it when its seed is reached
}
HP_HOT void branch_track(ReadCtx &r, int n, NScore &ns)
{   // branch_track_new, :873-920
    // The walk up a chain is a pointer chase through HBM: what a step needs (son count, score, predecessor) is requested together, one
    // memory round trip per step, and the nodes walked are kept in a lane register so that node_add_score need not chase them again.
    const HP_G int32_t *g_from = (const HP_G int32_t *)r.n_from, *g_son_n = (const HP_G int32_t *)r.n_son_n;
    const HP_G NodeS *g_nd = (const HP_G NodeS *)r.nd;
    HP_G int32_t *g_ms = (HP_G int32_t *)r.n_max_score, *g_mn = (HP_G int32_t *)r.n_max_NM, *g_mx = (HP_G int32_t *)r.n_max_node, *g_in_de = (HP_G int32_t *)r.n_in_de;
    int max_score, max_NM, max_node;
    g_in_de[n] = -1;
    const int n_sons = g_son_n[n], n_score = g_nd[n].score, n_NM = g_nd[n].NM;
    int fa = g_from[n];
    if (n_sons == 0) { max_node = n; max_score = n_score; max_NM = n_NM; }      // (stored below, when and where they are read again)
    else { max_node = g_mx[n]; max_score = g_ms[n]; max_NM = g_mn[n]; }
    wv::Lane<int> path;                               // the ancestors of max_node walked so far, while path_ok
    WAVE_FOR(l) { path[l] = 0; }
    int n_path = 0; bool path_ok = n_sons == 0;       // a leaf: max_node is n itself, its ancestors are exactly the nodes walked below
    // What the reference stores in every node it walks over (max_score, max_NM, max_node, in_de = -1; :885-910) is read again only for
    // the node right below a node with several sons (get_max_son / cut_branch look at their sons) or below a negative edge: the walk
    // keeps the three values in registers and stores them for that node alone -- four stores less per step.  in_de is only ever
    // compared with 0 (is the node a leaf?), and a walked node with one son keeps its 1.
    int prev = n;                                     // the node below fa
    while (fa >= 0) {
        const int fa_sons = g_son_n[fa], fa_score = g_nd[fa].score, fa_from = g_from[fa];
#ifdef HP_PROF_TRACK
        if (HP_PROF_CHAIN_ON && r.prof) r.prof[20] += 1;
#endif
        if (fa_sons == 1) {
            if (fa_score > max_score) {               // negative edge
                const int s = r.n_first[fa];
                g_ms[prev] = max_score; g_mn[prev] = max_NM; g_mx[prev] = max_node;       // s == prev: detach reads them
                wv::sync();
                r.n_in_de[s] = -1;
                detach(r, s, max_node, ns);
                r.n_son_n[fa] = 0; r.n_first[fa] = r.n_last[fa] = -1;
                max_score = r.nd[fa].score; max_NM = r.nd[fa].NM; max_node = fa;
                n_path = 0; path_ok = true;           // from here on the ancestors of max_node = fa are what is walked next
            } else if (path_ok) {
                if (n_path < 64) { WAVE_FOR(l) { if (l == n_path) path[l] = fa; } ++n_path; } else path_ok = false;
            }
            prev = fa;
            fa = fa_from;                             // detach() above changes n_from of the son only, never of fa
        } else {
            g_ms[prev] = max_score; g_mn[prev] = max_NM; g_mx[prev] = max_node;           // prev is a son of fa: what get_max_son / cut_branch read
            const int left_ = g_in_de[fa] - 1;
            g_in_de[fa] = left_;
#ifdef HP_PROF_TRACK
            if (HP_PROF_CHAIN_ON && r.prof) r.prof[21] += 1;
            const long long tcb_ = wv::clock();
#endif
            if (left_ == 0) { wv::sync(); cut_branch(r, fa, ns); }
#ifdef HP_PROF_TRACK
            if (HP_PROF_CHAIN_ON && r.prof) r.prof[22] += wv::clock() - tcb_;
#endif
            return;
        }
    }
    ns_add_end(r, ns, max_score, max_NM, max_node, path_ok ? &path : nullptr, n_path);
}

// ---------------------------------------------------------------- which hits can matter for the chain into `node`
// A predecessor that get_fseed_dis connects to a target lies on the target's contig and strand within R(target) bases
// of it, and R(target) <= Rcap for every target of a pass (Rcap = R of the pass's right anchor, which has the largest
// seed distance).  In the (contig, strand, position) order, the hits that can reach `node` through any number of such
// edges therefore lie in the run around `node` in which consecutive hits are at most Rcap apart: two hits within Rcap
// of each other have only gaps <= Rcap between them.  Hits outside that run never connect to a hit inside it, so the DP
// state of the run (predecessors, scores, the son_flag side effects) is the same whether or not they are processed.
struct RunR { int lo, hi; };     // results come back by value, in registers: no caller's local is handed by address to a non-inlined routine (hazards)
HP_NOINL RunR reach_run(ReadCtx &r, int node, long long Rcap)
{
    const HP_G NodeS *ns = (const HP_G NodeS *)r.nd;
    const HP_G int32_t *g_srt = (const HP_G int32_t *)r.srt;
    const int H = r.H, rT = r.rnk[node];
    const NodeS N = node_load(ns + node);
    const int key = N.chr * 2 + (N.strand > 0 ? 1 : 0);
    int lo = rT, hi = rT;
    for (int dir = -1; dir <= 1; dir += 2) {
        int edge = rT;                                   // last rank known to be in the run
        for (;;) {
            wv::Lane<int> bad;
            WAVE_FOR(l) {
                const int i = edge + dir * (1 + l), j = i - dir;           // j: the neighbour on the side of `node`
                int b = 1;
                if (i >= 0 && i < H) {
                    const NodeS A = node_load(ns + g_srt[i]), B = node_load(ns + g_srt[j]);
                    long long d = A.pos - B.pos; if (d < 0) d = -d;
                    b = !((A.chr * 2 + (A.strand > 0 ? 1 : 0)) == key && d <= Rcap);
                }
                bad[l] = b;
            }
            const unsigned long long m = wv::ballot(bad);
            if (m) { edge += dir * __builtin_ctzll(m); break; }
            edge += dir * 64;
        }
        if (dir < 0) lo = edge; else hi = edge;
    }
    RunR rr; rr.lo = lo; rr.hi = hi;
    return rr;
}

// The driver loop of branch tracking (:1356-1361, :961-966): seeds from last to first, hits of a seed in ascending
// order, every hit that is (still) a leaf of the pass starts a track.  The leaf test of a seed's hits is done by the
// lanes when the seed is reached: tracking a hit only changes hits of earlier seeds (its ancestors) and sons whose own
// tracks are complete, never another hit of the same seed that has not been visited yet.
// The reference looks at every seed; most have nothing to start from, and looking costs a memory round trip each.  Here one pass
// over all hits of the range marks the seeds that hold a leaf now (a bit per seed in LDS); a hit becomes a leaf later only
// through cut_branch, which marks its seed (always an earlier one than the track that completed it); the driver visits the
// marked seeds only, last to first, with the same test as before.
HP_INL void track_slot(ReadCtx &r, int h0, int h1, int dp_flag, bool skip_lone, NScore &ns)
{
    const HP_G NodeS *ns_ = (const HP_G NodeS *)r.nd;
    const HP_G int32_t *g_in_de = (const HP_G int32_t *)r.n_in_de, *g_from = (const HP_G int32_t *)r.n_from, *g_son_n = (const HP_G int32_t *)r.n_son_n;
    for (int b = h0; b < h1; b += 64) {
        wv::Lane<int> leaf;
        WAVE_FOR(l) {
            const int k = b + l;
            int v = 0;
            if (k < h1) {
                int q[4]; hp_load16((const HP_G char *)(ns_ + k) + 16, q); v = (int)(int8_t)(q[1] & 0xff) == dp_flag && g_in_de[k] == 0;
                // A hit without predecessor and without sons is a path of its own with score 1: node_add_score (:786) drops it when the
                // threshold is above that, and nothing reads what branch_track_new would leave in its own fields
                if (v && skip_lone && g_from[k] < 0 && g_son_n[k] == 0) v = 0;
            }
            leaf[l] = v;
        }
        for (unsigned long long m = wv::ballot(leaf); m; m &= m - 1) {
#ifdef HP_PROF_TRACK
            const long long tb_ = wv::clock();
#endif
            branch_track(r, b + __builtin_ctzll(m), ns);
#ifdef HP_PROF_TRACK
            if (HP_PROF_CHAIN_ON && r.prof) { r.prof[17] += wv::clock() - tb_; r.prof[18] += 1; }
#endif
        }
    }
}
HP_NOINL void track_leaves(ReadCtx &r, int first_slot, int last_slot, int dp_flag, NScore &ns)
{
    const HP_G NodeS *ns_ = (const HP_G NodeS *)r.nd;
    const HP_G int32_t *g_in_de = (const HP_G int32_t *)r.n_in_de, *g_from = (const HP_G int32_t *)r.n_from, *g_son_n = (const HP_G int32_t *)r.n_son_n;
    const HP_G int32_t *g_seed = (const HP_G int32_t *)r.n_seed;
    const HP_G int64_t *g_hoff = (const HP_G int64_t *)r.hit_off;
    const int64_t hb = r.hb;
    const bool skip_lone = ns.min_score_thd > 1;
    if (last_slot < first_slot) return;
    const int nw = (last_slot >> 5) + 1, w_lo = first_slot >> 5;
    if (nw > r.cx.lds_words) {                           // more seeds than this wave's LDS has bits for: every seed, as the reference
        for (int i = last_slot; i >= first_slot; --i) track_slot(r, (int)(g_hoff[i] - hb), (int)(g_hoff[i + 1] - hb), dp_flag, skip_lone, ns);
        return;
    }
#ifdef HP_PROF_TRACK
    const long long tt0_ = wv::clock();
#endif
    HP_L int32_t *bits = r.cx.lds;
    for (int w0 = w_lo; w0 < nw; w0 += 64) { WAVE_FOR(l) { if (w0 + l < nw) bits[w0 + l] = 0; } }
    wv::sync();
    {
        const int k_lo = (int)(g_hoff[first_slot] - hb), k_hi = (int)(g_hoff[last_slot + 1] - hb);
        for (int b = k_lo; b < k_hi; b += 64) {
            WAVE_FOR(l) {
                const int k = b + l;
                if (k < k_hi) {
                    int q[4]; hp_load16((const HP_G char *)(ns_ + k) + 16, q);
                    int v = (int)(int8_t)(q[1] & 0xff) == dp_flag && g_in_de[k] == 0;
                    if (v && skip_lone && g_from[k] < 0 && g_son_n[k] == 0) v = 0;
                    if (v) { const int x = g_seed[k]; wv::lds_or(bits + (x >> 5), (int)(1u << (x & 31))); }
                }
            }
        }
    }
    wv::sync();
#ifdef HP_PROF_TRACK
    if (HP_PROF_CHAIN_ON && r.prof) r.prof[16] += wv::clock() - tt0_;
#endif
    r.leaf_bits = bits; r.leaf_on = true;
    for (int w = nw - 1; w >= w_lo; --w) {
        for (;;) {
            wv::sync();
            unsigned v = (unsigned)wv::uni(bits[w]);
            if (w == w_lo) v &= ~0u << (first_slot & 31);
            if (w == nw - 1 && (last_slot & 31) != 31) v &= (2u << (last_slot & 31)) - 1;
            if (!v) break;
            const int bit = 31 - __builtin_clz(v);
            wv::sync();
            bits[w] = (int)((unsigned)wv::uni(bits[w]) & ~(1u << bit));
            wv::sync();
            const int i = w * 32 + bit;
#ifdef HP_PROF_TRACK
            if (HP_PROF_CHAIN_ON && r.prof) r.prof[19] += 1;
#endif
            track_slot(r, (int)(g_hoff[i] - hb), (int)(g_hoff[i + 1] - hb), dp_flag, skip_lone, ns);
        }
    }
    r.leaf_on = false;
}

// ---------------------------------------------------------------- frag_mini_dp_line with the whole pass in registers
// Most mini-DP passes involve few hits: either the seed range between the two anchors is short, or (pass from START)
// only the hits that can reach the right anchor matter (reach_run).  Up to HP_MS_SETS x 64 such hits are loaded once,
// one per lane and set, and frag_dp_per_init, frag_dp_update over the range, the forced update of the right anchor
// (or the choice of the best end node), the walk back along the chosen predecessors and the final state of every
// touched hit are computed on those registers; memory sees one gather at the start and one scatter at the end.
// Candidates of a target are then simply all loaded hits of earlier seeds -- what the reference scans -- so no
// window pruning is involved.  Hits of the range that are not loaded (pass from START, outside the run) keep their
// previous state: they can neither be candidates nor targets of this pass (see reach_run), and every later pass
// re-initialises the hits it uses (fnode_set) before reading them.
// Returns -1 when the hits do not fit; the caller then runs the pass through memory (mini_line).
#define HP_MS_MAX_SETS 4
template <int NS> HP_INL int ms_pick(const wv::Lane<int> *f, int c) {
    const int l = c & 63, j = c >> 6;
    int v = wv::bcast(f[0], l);
#pragma unroll
    for (int q = 1; q < NS; ++q) { const int u = wv::bcast(f[q], l); v = j == q ? u : v; }
    return v;
}
// ids: list ? list[idx] : k_lo + idx, idx < n_ids <= 64 * NS
struct MiniR { int n, d_score, d_NM; };     // nodes written to line[] (-1: the hits do not fit this routine), score and NM gained over the plain edge
template <int NS>
HP_NOINL MiniR mini_line_sets(ReadCtx &r, int left, int right, int right_x, int32_t *line, int _head, int _tail,
                              int n_ids, const int32_t *list)
{
    left = wv::uni(left); right = wv::uni(right); right_x = wv::uni(right_x); _head = wv::uni(_head); _tail = wv::uni(_tail);
    const int head = _head ? left : -1;
    const int left_x = nx(r, left);
    const int dp_flag = MULTI_FLAG;
    const int start_slot = left_x + 1;
    const int k_lo = wv::uni(hoff(r, start_slot)), k_t0 = wv::uni(hoff(r, left_x + 2));
    const HP_G NodeS *ns = (const HP_G NodeS *)r.nd;
    HP_G NodeS *gd = (HP_G NodeS *)r.nd;
    const HP_G int16_t *g_hnm = (const HP_G int16_t *)r.h_nm;
    HP_G int32_t *g_from = (HP_G int32_t *)r.n_from, *g_node_n = (HP_G int32_t *)r.n_node_n;
    const EdgeK K = edge_consts(r.cx.P);
    const int POSMAX = (1 << 28) - 1;
    n_ids = wv::uni(n_ids);
    const HP_G int32_t *g_list = (const HP_G int32_t *)wv::uni64((long long)list);
    const bool by_list = list != nullptr;
    // ---- anchors
    NodeS Fh; Fh.pos = 0; Fh.chr = 0; Fh.slot_j = 0; Fh.sid = 0; Fh.strand = 0; Fh.len_dif8 = 0; Fh.pad_ = 0; Fh.dp_flag = 0; Fh.son_flag = 0; Fh.match_flag = 0; Fh.score = 0; Fh.NM = 0;
    NodeS Rt = Fh;
    int head_nm = 0, right_nm = 0;
    if (head >= 0) { Fh = node_load(ns + head); head_nm = g_hnm[head]; }
    if (right >= 0) { Rt = node_load(ns + right); right_nm = g_hnm[right]; }
    const int left_NM = left < 0 ? 0 : (left == head ? head_nm : (int)g_hnm[left]);
    int old_score, old_NM;
    if (_tail == 0) { old_score = 1; old_NM = left_NM; }
    else { old_score = 2 + score_table(Rt.match_flag); old_NM = left_NM + right_nm; }
    // ---- gather + frag_dp_per_init (:766-784, :1086-1091)
    long long pairs_ = head >= 0 ? n_ids : 0;      // accounting, flushed once at the end
    wv::Lane<int> A0[NS], A1[NS], A2[NS], A3[NS], B0[NS];
    wv::Lane<int> Dpf[NS], Son[NS], Mf[NS], Sc[NS], Nm[NS], Fr[NS], Nn[NS], Cf[NS], Id[NS], Tk[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        WAVE_FOR(l) {
            const int idx = 64 * j + l;
            const bool valid = idx < n_ids;
            const int id = valid ? (by_list ? (int)g_list[idx] : k_lo + idx) : 0;
            int a[4], b[4];
            hp_load16(ns + id, a); hp_load16((const HP_G char *)(ns + id) + 16, b);
            const int nm0 = g_hnm[id];
            const int slot = a[3] >> 14;
            const int df = (int)(int8_t)(b[1] & 0xff);
            const bool take = valid && slot >= start_slot && slot < right_x && (df == dp_flag || df == 0 - dp_flag);
            int dpf = df, son = (b[1] >> 8) & 0xff, mf = (b[1] >> 16) & 0xff, sc = b[2], nm = b[3], fr = -1;
            if (take) {
                if (head < 0) { dpf = dp_flag; son = F_INIT; mf = F_MATCH; sc = 1; nm = nm0; fr = -1; }
                else {
                    const NodeS Q = node_unpack(a, b);
                    const int flag = edge_flag_packed(K, Fh, Q);
                    if (flag != F_UNCONNECT && flag != F_CHR_DIF) { dpf = dp_flag; son = F_INIT; mf = flag; sc = 2 + score_table(flag); nm = nm0 + head_nm; fr = head; }
                    else dpf = 0 - dp_flag;
                }
            }
            A0[j][l] = a[0]; A1[j][l] = a[1]; A2[j][l] = a[2]; A3[j][l] = a[3]; B0[j][l] = b[0];
            Dpf[j][l] = dpf; Son[j][l] = son; Mf[j][l] = mf; Sc[j][l] = sc; Nm[j][l] = nm; Fr[j][l] = fr; Nn[j][l] = 1; Cf[j][l] = -1;
            Id[j][l] = id; Tk[j][l] = take ? 1 : 0;
        }
    }
#define HP_MS_Q(j, l, Q) do { int a_[4] = { A0[j][l], A1[j][l], A2[j][l], A3[j][l] }; \
        int b_[4] = { B0[j][l], (Dpf[j][l] & 0xff) | (Son[j][l] << 8) | (Mf[j][l] << 16), Sc[j][l], Nm[j][l] }; Q = node_unpack(a_, b_); } while (0)
    // one target T against every loaded hit; returns the winner (cidx, or -1) and its edge class, score and NM
#define HP_MS_SCAN(S, t_from_id, w_c, w_flag, w_score, w_nm, changed_) do { \
        pairs_ += n_ids; \
        wv::Lane<long long> key; wv::Lane<int> bp, bf, negp, n_p, n_f, n_c, n_n, okl; \
        WAVE_FOR(l) { key[l] = -1; bp[l] = 0; bf[l] = 0; negp[l] = -0x7fffffff; n_p[l] = 0; n_f[l] = 0; n_c[l] = 0; n_n[l] = 0; okl[l] = 0; } \
        _Pragma("unroll") for (int j = 0; j < NS; ++j) { \
            WAVE_FOR(l) { NodeS Q; HP_MS_Q(j, l, Q); int ow_, oka_ = 0; \
                scan_eval(K, S, Q, 64 * j + l, Tk[j][l] & (Dpf[j][l] == dp_flag), start_slot, dp_flag, key[l], bp[l], bf[l], negp[l], n_p[l], n_f[l], n_c[l], n_n[l], ow_, oka_); \
                okl[l] |= oka_; } } \
        w_c = -1; changed_ = false; \
        if (wv::ballot(okl) != 0) { \
            const int npos = wv::reduce_max(negp); \
            if (npos != -0x7fffffff) { \
                wv::Lane<int> w_; WAVE_FOR(l) w_[l] = negp[l] == npos; \
                const int wl = __builtin_ctzll(wv::ballot(w_)); \
                w_c = wv::bcast(n_p, wl); w_flag = wv::bcast(n_f, wl); w_score = wv::bcast(n_c, wl); w_nm = wv::bcast(n_n, wl); \
                changed_ = ms_pick<NS>(Id, w_c) != (t_from_id); \
            } else { \
                const long long best_key = wv::reduce_max64(key); \
                if (best_key >= 0) { \
                    const int nm_ = 524287 - (int)((best_key >> 28) & 524287), cand_ = (int)(best_key >> 47) - 32768; \
                    if (cand_ > w_score || (cand_ == w_score && nm_ < w_nm)) { \
                        wv::Lane<int> w_; WAVE_FOR(l) w_[l] = key[l] == best_key; \
                        const int wl = __builtin_ctzll(wv::ballot(w_)); \
                        w_c = wv::bcast(bp, wl); w_flag = wv::bcast(bf, wl); w_score = cand_; w_nm = nm_; \
                        changed_ = ms_pick<NS>(Id, w_c) != (t_from_id); \
                    } } } } } while (0)
    // ---- frag_dp_update over the range (:701-764), targets in ascending hit order
    for (int k_next = k_t0;;) {
        wv::Lane<int> m;
        WAVE_FOR(l) {
            int v = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < NS; ++j) { const int id = Id[j][l]; if (Tk[j][l] && Dpf[j][l] == dp_flag && id >= k_next && id < v) v = id; }
            m[l] = 0 - v;
        }
        const int kmin = 0 - wv::reduce_max(m);
        if (kmin == 0x7fffffff) break;
        k_next = kmin + 1;
        int tc = 0;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            wv::Lane<int> e;
            WAVE_FOR(l) e[l] = Tk[j][l] && Id[j][l] == kmin;
            const unsigned long long bm = wv::ballot(e);
            if (bm) tc = 64 * j + __builtin_ctzll(bm);
        }
        ScanT S;
        {
            int a[4] = { ms_pick<NS>(A0, tc), ms_pick<NS>(A1, tc), ms_pick<NS>(A2, tc), ms_pick<NS>(A3, tc) };
            int b[4] = { ms_pick<NS>(B0, tc), 0, ms_pick<NS>(Sc, tc), ms_pick<NS>(Nm, tc) };
            S.T = node_unpack(a, b);
        }
        S.x = S.T.slot_j >> 14; S.t_NM = S.T.NM; S.tkey = S.T.chr * 2 + (S.T.strand > 0 ? 1 : 0); S.Rw = 0x7fffffffffffll;
        const int t_from = ms_pick<NS>(Fr, tc);
        int w_c, w_flag = 0, w_score = S.T.score, w_nm = S.t_NM; bool changed;
        HP_MS_SCAN(S, t_from, w_c, w_flag, w_score, w_nm, changed);
        if (changed) {
            const int from_id = ms_pick<NS>(Id, w_c), nn = ms_pick<NS>(Nn, w_c) + 1;
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                WAVE_FOR(l) {
                    const int c = 64 * j + l;
                    if (c == tc) { Fr[j][l] = from_id; Cf[j][l] = w_c; Sc[j][l] = w_score; Nm[j][l] = w_nm; Mf[j][l] = w_flag; Nn[j][l] = nn; }
                    if (c == w_c) Son[j][l] = w_flag;
                }
            }
        }
    }
    // ---- the end of the line: best end node (:1105-1123) or the forced update of the right anchor (:1125-1134)
    int max_score, max_NM = 0, max_n = 0, max_c = -1;
    if (_tail == 0) {
        max_score = old_score;
        wv::Lane<long long> key;
        WAVE_FOR(l) {
            long long kb = -1;
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                if (Tk[j][l] && Dpf[j][l] == dp_flag) {
                    const int sj = A3[j][l];
                    const int pos = ((right_x - 1 - (sj >> 14)) << 14) | (sj & 16383);
                    const long long kk = ((long long)(Sc[j][l] + 32768) << 47) | ((long long)(524287 - Nm[j][l]) << 28) | (long long)(POSMAX - pos);
                    kb = kk > kb ? kk : kb;
                }
            }
            key[l] = kb;
        }
        const long long bk = wv::reduce_max64(key);
        if (bk >= 0) {
            const int pos = POSMAX - (int)(bk & POSMAX);
            const int nm = 524287 - (int)((bk >> 28) & 524287), sc = (int)(bk >> 47) - 32768;
            if (sc > max_score || (sc == max_score && nm < max_NM)) {
                const int want_sj = ((right_x - 1 - (pos >> 14)) << 14) | (pos & 16383);
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    wv::Lane<int> e;
                    WAVE_FOR(l) e[l] = Tk[j][l] && Dpf[j][l] == dp_flag && A3[j][l] == want_sj;
                    const unsigned long long bm = wv::ballot(e);
                    if (bm) max_c = 64 * j + __builtin_ctzll(bm);
                }
                max_score = sc; max_NM = nm; max_n = ms_pick<NS>(Nn, max_c);
            }
        }
    } else {
        ScanT S;
        S.T = Rt; S.T.score = old_score; S.T.NM = old_NM;
        S.x = right_x; S.t_NM = old_NM; S.tkey = Rt.chr * 2 + (Rt.strand > 0 ? 1 : 0); S.Rw = 0x7fffffffffffll;
        int w_c, w_flag = 0, w_score = old_score, w_nm = old_NM; bool changed;
        HP_MS_SCAN(S, head, w_c, w_flag, w_score, w_nm, changed);
        int r_from = head, r_nn = 1;
        if (changed) {
            r_from = ms_pick<NS>(Id, w_c); r_nn = ms_pick<NS>(Nn, w_c) + 1; max_c = w_c;
#pragma unroll
            for (int j = 0; j < NS; ++j) { WAVE_FOR(l) { if (64 * j + l == w_c) Son[j][l] = w_flag; } }
            gd[right].match_flag = (uint8_t)w_flag;
        }
        g_from[right] = r_from; gd[right].score = w_score; gd[right].NM = w_nm; g_node_n[right] = r_nn;
        max_score = w_score; max_NM = w_nm; max_n = r_nn - 1;
    }
    // ---- walk back to the head (:1136-1147)
    bool bad = false;
    {
        int c = max_c, node_i = max_n - 1;
        while (c >= 0) {
            if (node_i < 0) { bad = true; break; }                            // "[frag mini dp] BUG" exit, :1140
            line[node_i--] = ms_pick<NS>(Id, c);
            c = ms_pick<NS>(Cf, c);
        }
        if (node_i >= 0) bad = true;
    }
    // ---- final state of every hit the pass touched
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        WAVE_FOR(l) {
            if (Tk[j][l]) {
                const int id = Id[j][l];
                hp_store16((HP_G char *)(gd + id) + 16, B0[j][l], (Dpf[j][l] & 0xff) | (Son[j][l] << 8) | (Mf[j][l] << 16), Sc[j][l], Nm[j][l]);
                if (Dpf[j][l] == dp_flag) { g_from[id] = Fr[j][l]; g_node_n[id] = Nn[j][l]; }
            }
        }
    }
    wv::sync();
#undef HP_MS_Q
#undef HP_MS_SCAN
    r.n_pairs += pairs_;
    MiniR R_; R_.n = 0; R_.d_score = 0; R_.d_NM = 0;
    if (bad) { r.cx.status |= ST_REFEXIT; return R_; }
    R_.n = max_n; R_.d_score = max_score - old_score; R_.d_NM = max_NM - old_NM;
    return R_;
}

HP_INL MiniR mini_line_regs(ReadCtx &r, int left, int right, int right_x, int32_t *line, int _head, int _tail)
{
    MiniR none; none.n = -1; none.d_score = 0; none.d_NM = 0;
    const int head = _head ? left : -1;
    const int left_x = nx(r, left);
    const int k_lo = hoff(r, left_x + 1), k_hi = hoff(r, right_x);
    if (k_hi - k_lo <= 64) return mini_line_sets<1>(r, left, right, right_x, line, _head, _tail, k_hi - k_lo, nullptr);
    // A longer seed range.  The hits that take part are those that can be connected to the pass's anchor: the head when
    // there is one (frag_dp_per_init keeps only hits that connect to it, :766-784), else the right anchor (pass from
    // START, see reach_run).  They lie in the anchor's run of the sorted order; the hits of that run that belong to the
    // seed range and to this kind of pass are listed, and if at most 256 remain the pass runs on registers.
    const int anchor = head >= 0 ? head : ((_tail != 0 && right >= 0) ? right : -1);
    if (anchor < 0) return k_hi - k_lo <= 64 * HP_MS_MAX_SETS ? mini_line_sets<HP_MS_MAX_SETS>(r, left, right, right_x, line, _head, _tail, k_hi - k_lo, nullptr) : none;
    const lamsa_hp_para *P = r.cx.P;
    const int sid_hi = right_x < r.seed_out ? r.seed_id[right_x] : r.seed_id[r.seed_out - 1];
    const int did_max = sid_hi - (head >= 0 ? r.seed_id[left_x] : r.seed_id[left_x + 1]);
    const int mdm = P->match_dis * ((P->aln_mode & 2) ? did_max : 1);
    long long Rw = P->SV_len_thd > did_max * P->seed_step ? P->SV_len_thd : did_max * P->seed_step;
    if (mdm + 1 > Rw) Rw = mdm + 1;
    Rw += 128 + (long long)did_max * P->seed_step;
    const RunR rr_ = reach_run(r, anchor, Rw);
    const int rlo = rr_.lo, rhi = rr_.hi;
    const int R = rhi - rlo + 1;
    if (R > 2048) return none;
    const size_t mark = arena_mark(r.cx.tmp);
    int32_t *list = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)(R + 64));
    if (!list) { arena_release(r.cx.tmp, mark); return none; }
    int n = 0;
    {
        const HP_G NodeS *ns = (const HP_G NodeS *)r.nd;
        const HP_G int32_t *g_srt = (const HP_G int32_t *)r.srt;
        HP_G int32_t *g_list = (HP_G int32_t *)list;
        const int start_slot = left_x + 1;
        for (int b0 = 0; b0 < R; b0 += 64) {
            wv::Lane<int> keep, idl;
            WAVE_FOR(l) {
                int ok = 0, id = 0;
                if (b0 + l < R) {
                    id = g_srt[rlo + b0 + l];
                    int b[4];
                    hp_load16((const HP_G char *)(ns + id) + 16, b);
                    const int slot = ns[id].slot_j >> 14, df = (int)(int8_t)(b[1] & 0xff);
                    ok = slot >= start_slot && slot < right_x && (df == MULTI_FLAG || df == 0 - MULTI_FLAG);
                }
                keep[l] = ok; idl[l] = id;
            }
            const unsigned long long m = wv::ballot(keep);
            WAVE_FOR(l) { if (keep[l]) g_list[n + __builtin_popcountll(m & ((1ull << l) - 1))] = idl[l]; }
            n += __builtin_popcountll(m);
        }
        wv::sync();
    }
    MiniR ret = none;
    if (n <= 64) ret = mini_line_sets<1>(r, left, right, right_x, line, _head, _tail, n, list);
    else if (n <= 64 * HP_MS_MAX_SETS) ret = mini_line_sets<HP_MS_MAX_SETS>(r, left, right, right_x, line, _head, _tail, n, list);
    arena_release(r.cx.tmp, mark);
    return ret;
}

// ---------------------------------------------------------------- frag_mini_dp_line, :1068-1150
// left / right are node indices (left may be -1 = START); right_x is right's slot, which may be the
// virtual slot seed_out (then right < 0 and _tail == 0).  Returns the number of nodes written to line[].
HP_NOINL MiniR mini_line_mem(ReadCtx &r, int left, int right, int right_x, int32_t *line, int _head, int _tail)
{
    MiniR R_; R_.n = 0; R_.d_score = 0; R_.d_NM = 0;
    const int head = _head ? left : -1;
    const int left_x = nx(r, left), head_x = nx(r, head);
    const int left_NM = left < 0 ? 0 : r.h_nm[left];
    int old_score, old_NM;
    if (_tail == 0) { old_score = 1; old_NM = left_NM; }
    else { old_score = 2 + score_table(r.nd[right].match_flag); old_NM = left_NM + r.h_nm[right]; }
    const int dp_flag = MULTI_FLAG;
    int rlo = 0, rhi = 0x7fffffff;
    if (head < 0 && _tail != 0 && right >= 0) {
        // From START every hit of the range would be active, but only the chain into `right` is used afterwards
        // (:1129-1147): restrict the pass to the hits that can reach `right` at all (exact, see reach_run).
        const lamsa_hp_para *P = r.cx.P;
        const int did_max = r.seed_id[right_x] - r.seed_id[left_x + 1];
        const int mdm = P->match_dis * ((P->aln_mode & 2) ? did_max : 1);
        long long Rw = P->SV_len_thd > did_max * P->seed_step ? P->SV_len_thd : did_max * P->seed_step;
        if (mdm + 1 > Rw) Rw = mdm + 1;
        Rw += 128 + (long long)did_max * P->seed_step;
        const RunR rr_ = reach_run(r, right, Rw); rlo = rr_.lo; rhi = rr_.hi;
    }
    nodes_per_init(r, hoff(r, left_x + 1), hoff(r, right_x), head, dp_flag, 0, rlo, rhi);
    dp_update_range(r, hoff(r, left_x + 2), hoff(r, right_x), left_x + 1, dp_flag, false, false);      // callers guarantee left_x + 2 <= right_x
#ifdef HP_PROF
    const long long tm0_ = wv::clock();
    if (r.prof) r.prof[13] += 1;
#endif
    int max_score, max_NM = 0, max_n = 0, max_node = head;
    if (_tail == 0) {
        // best end node: score desc, NM asc, then the reference's scan order (seeds descending, hits ascending), :1105-1123
        max_score = old_score;
        const HP_G NodeS *ns = (const HP_G NodeS *)r.nd;
        const int k0 = hoff(r, left_x + 1), k1 = hoff(r, right_x);
        wv::Lane<long long> key;
        WAVE_FOR(l) { key[l] = -1; }
        for (int b = k0; b < k1; b += 64) {
            WAVE_FOR(l) {
                const int k = b + l;
                if (k < k1 && r.nd[k].dp_flag == dp_flag) {
                    const int sj = ns[k].slot_j;
                    const int pos = ((right_x - 1 - (sj >> 14)) << 14) | (sj & 16383);
                    const long long kk = ((long long)(r.nd[k].score + 32768) << 47) | ((long long)(524287 - r.nd[k].NM) << 28) | (long long)(((1 << 28) - 1) - pos);
                    key[l] = kk > key[l] ? kk : key[l];
                }
            }
        }
        const long long bk = wv::reduce_max64(key);
        if (bk >= 0) {
            const int pos = ((1 << 28) - 1) - (int)(bk & ((1 << 28) - 1));
            const int nm = 524287 - (int)((bk >> 28) & 524287), sc = (int)(bk >> 47) - 32768;
            if (sc > max_score || (sc == max_score && nm < max_NM)) {
                const int k = hoff(r, right_x - 1 - (pos >> 14)) + (pos & 16383);
                max_score = sc; max_NM = nm; max_node = k; max_n = r.n_node_n[k];
            }
        }
    } else {
        r.n_from[right] = head; r.nd[right].score = old_score; r.nd[right].NM = old_NM; r.n_node_n[right] = 1;
        wv::sync();
        dp_update(r, right, left_x + 1, dp_flag, false);
        max_score = r.nd[right].score; max_NM = r.nd[right].NM; max_node = r.n_from[right]; max_n = r.n_node_n[right] - 1;
    }
    int cur = max_node, node_i = max_n - 1;
    while (nx(r, cur) != head_x) {
        if (node_i < 0) { r.cx.status |= ST_REFEXIT; return R_; }     // "[frag mini dp] BUG" exit, :1140
        line[node_i--] = cur;
        cur = r.n_from[cur];
    }
    if (node_i >= 0) { r.cx.status |= ST_REFEXIT; return R_; }
    R_.d_score = max_score - old_score; R_.d_NM = max_NM - old_NM;
#ifdef HP_PROF
    if (r.prof) r.prof[13] += 1;
#endif
    R_.n = max_n;
    return R_;
}

HP_INL MiniR mini_line(ReadCtx &r, int left, int right, int right_x, int32_t *line, int _head, int _tail)
{
#ifdef HP_PROF
    const long long t0_ = wv::clock();
#endif
    const MiniR m = mini_line_regs(r, left, right, right_x, line, _head, _tail);
#ifdef HP_PROF
    if (r.prof) { r.prof[44] += wv::clock() - t0_; r.prof[45] += 1; }
#endif
    return m.n >= 0 ? m : mini_line_mem(r, left, right, right_x, line, _head, _tail);
}

// ---------------------------------------------------------------- lines
struct LSet {
    int32_t *pool; int pool_cap;          // node indices of all lines, back to back
    int32_t *start, *len, *lb, *rb, *mf, *mh, *ls, *bs, *nm;   // per line (L_LB..L_NM, lamsa_aln.h:139-149)
    int32_t *rank, *sel;                  // line_rank / line_select_rank
    int n, cap;
    int32_t *fx, *lx;                     // first / last seed slot of every line, when staged (lset_stage); else nullptr
    int32_t *xtra; int xtra_n;            // staged: spare words of this wave's LDS for the temporaries of line_filter
};
HP_FN bool lset_alloc(Ctx &cx, LSet &L, int pool_cap, int line_cap)
{
    L.pool = (int32_t *)arena_alloc(cx, sizeof(int32_t) * (size_t)(pool_cap + 8));
    int32_t *m = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 11 * (size_t)(line_cap + 1));
    if (!L.pool || !m) return false;
    const int c = line_cap + 1;
    L.start = m; L.len = m + c; L.lb = m + 2 * c; L.rb = m + 3 * c; L.mf = m + 4 * c; L.mh = m + 5 * c;
    L.ls = m + 6 * c; L.bs = m + 7 * c; L.nm = m + 8 * c; L.rank = m + 9 * c; L.sel = m + 10 * c;
    L.pool_cap = pool_cap; L.cap = line_cap; L.n = 0;
    L.fx = L.lx = nullptr; L.xtra = nullptr; L.xtra_n = 0;
    return true;
}
HP_INL int firstx(const ReadCtx &r, const LSet &L, int l) { return L.fx ? L.fx[l] : r.n_seed[L.pool[L.start[l]]]; }
HP_INL int lastx(const ReadCtx &r, const LSet &L, int l) { return L.lx ? L.lx[l] : r.n_seed[L.pool[L.start[l] + L.len[l] - 1]]; }

// The per-line arrays of lines 0..n-1 copied into this wave's LDS (S points there; the node pool stays where it is), plus the first and
// last seed slot of every line: line_set_bound and frag_dp_path (:425, :1152) are a few hundred dependent look-ups of single words per
// read -- L.mf[L.rank[i]], the slot of a line's end node -- each a round trip to HBM otherwise.  False when they do not fit.
HP_INL bool lset_stage(const ReadCtx &r, const LSet &L, int n, LSet &S)
{
    const int c = n + 1;
    if (n <= 0 || 13 * c + 64 > r.cx.lds_words) return false;
    HP_L int32_t *w = r.cx.lds;
    wv::sync();                                                    // whatever used this LDS before is done
    const HP_G int32_t *src[11] = { (const HP_G int32_t *)L.start, (const HP_G int32_t *)L.len, (const HP_G int32_t *)L.lb, (const HP_G int32_t *)L.rb, (const HP_G int32_t *)L.mf,
                                    (const HP_G int32_t *)L.mh, (const HP_G int32_t *)L.ls, (const HP_G int32_t *)L.bs, (const HP_G int32_t *)L.nm, (const HP_G int32_t *)L.rank, (const HP_G int32_t *)L.sel };
    const HP_G int32_t *g_pool = (const HP_G int32_t *)L.pool, *g_seed = (const HP_G int32_t *)r.n_seed;
    for (int i0 = 0; i0 < n; i0 += 64) {
        WAVE_FOR(l) {
            const int i = i0 + l;
            if (i < n) {
#pragma unroll
                for (int a = 0; a < 11; ++a) w[a * c + i] = src[a][i];
                const int st = src[0][i], ln = src[1][i];
                w[11 * c + i] = ln > 0 ? g_seed[g_pool[st]] : 0; w[12 * c + i] = ln > 0 ? g_seed[g_pool[st + ln - 1]] : 0;
            }
        }
    }
    wv::sync();
    S = L;
    int32_t *m = (int32_t *)w;
    S.start = m; S.len = m + c; S.lb = m + 2 * c; S.rb = m + 3 * c; S.mf = m + 4 * c; S.mh = m + 5 * c;
    S.ls = m + 6 * c; S.bs = m + 7 * c; S.nm = m + 8 * c; S.rank = m + 9 * c; S.sel = m + 10 * c; S.fx = m + 11 * c; S.lx = m + 12 * c;
    S.xtra = m + 13 * c; S.xtra_n = r.cx.lds_words - 13 * c;
    return true;
}

HP_FN void sort_endpos(ReadCtx &r, LSet &L, int ls, int len, int32_t *tmp_pos)
{   // line_sort_endpos, :12 -- end slot descending, stable (the goldens come from glibc's merge sort).  Every line finds its own
    // place: the number of lines that end later, or as late and come first -- one line per lane, the keys of all lines streamed past
    int32_t *g_tmp = tmp_pos, *g_rank = L.rank, *g_sel = L.sel;          // generic pointers: the arrays may be staged in LDS (lset_stage)
    for (int i0 = 0; i0 < len; i0 += 64) { WAVE_FOR(l) { const int i = i0 + l; if (i < len) g_tmp[i] = lastx(r, L, ls + i); } }
    wv::sync();
    for (int i0 = 0; i0 < len; i0 += 64) {
        wv::Lane<int> key, place;
        WAVE_FOR(l) { const int i = i0 + l; key[l] = i < len ? g_tmp[i] : 0; place[l] = 0; }
        for (int j0 = 0; j0 < len; j0 += 64) {
            wv::Lane<int> kj;
            WAVE_FOR(l) { const int j = j0 + l; kj[l] = j < len ? g_tmp[j] : -0x7fffffff; }
            const int cnt = len - j0 < 64 ? len - j0 : 64;
            for (int q = 0; q < cnt; ++q) {
                const int k = wv::bcast(kj, q), j = j0 + q;
                WAVE_FOR(l) { const int i = i0 + l; place[l] += (k > key[l]) || (k == key[l] && j < i); }
            }
        }
        WAVE_FOR(l) { const int i = i0 + l; if (i < len) { g_rank[ls + place[l]] = ls + i; g_sel[ls + i] = ls + place[l]; } }
    }
    wv::sync();
}

HP_FN int line_merge(ReadCtx &r, LSet &L, int a, int b, float ovlp_r)
{   // :69-112
    int s1, e1, s2, e2, s, e, hi;
    s2 = firstx(r, L, a); e2 = lastx(r, L, a);
    if (L.mf[b] & L_NMERG) { hi = b; s1 = firstx(r, L, b); e1 = lastx(r, L, b); }
    else { hi = L.mh[b]; s1 = L.lb[hi]; e1 = L.rb[hi]; }
    s = s2 > s1 ? s2 : s1; e = e2 < e1 ? e2 : e1;
    const float rat1 = (float)((double)(e - s + 1) / (double)(e1 - s1 + 1));
    const float rat2 = (float)((double)(e - s + 1) / (double)(e2 - s2 + 1));
    if (rat1 < ovlp_r && rat2 < ovlp_r) { L.mf[a] = L_NMERG; return 0; }
    if (L.ls[a] <= L.ls[b] / 2 || L.ls[a] <= L.bs[b] / 2) {
        L.lb[hi] = s1; L.rb[hi] = e1; L.mf[hi] = L_MERGH;
        L.mf[a] = L_MERGB | L_DUMP; L.mh[a] = hi;
        return 1;
    }
    L.lb[hi] = s1 + s2 - s; L.rb[hi] = e1 + e2 - e; L.mf[hi] = L_MERGH;
    L.mf[a] = L_MERGB; L.mh[a] = hi;
    if (L.bs[b] > L.bs[a]) L.bs[a] = L.bs[b];
    return 1;
}

// best + secondaries of one cluster mb[0..mbn) (line indices); winners to mf[0..*mfn) when mf != nullptr
// returns the number of winners written to mf[] (mf may be null: nothing is written)
HP_NOINL int pick_in_cluster(ReadCtx &r, LSet &L, const int32_t *mb, int mbn, int per_max_multi, int32_t *tri_n, int32_t *mf, int32_t *spare = nullptr, int spare_n = 0)
{
    int mfn_ = 1; int *mfn = &mfn_;   // (a pointer to this routine's own local, never handed on) -- shared tail of line_filter (:166-235) and line_filter1 (:346-400)
    int b_score = 0, s_score = 0;
    for (int j = 0; j < mbn; ++j) {
        const int y = L.ls[mb[j]];
        if (y > b_score) { s_score = b_score; b_score = y; } else if (y > s_score) s_score = y;
    }
    if (s_score >= b_score / 2) {
        const size_t mark = arena_mark(r.cx.tmp);
        NScore ns;
        if (spare && 3 * (per_max_multi + 2) <= spare_n) {             // the heap in the spare words of the staged line set (LDS)
            ns.node = spare; ns.score = spare + (per_max_multi + 2); ns.NM = spare + 2 * (per_max_multi + 2);
            ns.cap = per_max_multi + 1; ns.max_n = per_max_multi; ns.node_n = 0; ns.min_score_thd = 0;
        } else if (!ns_alloc(r.cx, ns, per_max_multi + 1, per_max_multi)) { arena_release(r.cx.tmp, mark); return mfn_; }
        for (int j = 0; j < mbn; ++j) {
            const int li = mb[j];
            if (L.ls[li] >= b_score / 2) {
                int ret = ns_add_bounded(ns, li, L.ls[li], L.nm[li]);
                if (ret == -2) { L.mf[li] |= L_DUMP; if (tri_n) tri_n[li] = 0; }
                else if (ret != -1) { L.mf[ret] |= L_DUMP; if (tri_n) tri_n[ret] = 0; }
            } else { L.mf[li] |= L_DUMP; if (tri_n) tri_n[li] = 0; }
        }
        for (int i = (ns.node_n - 1) / 2; i >= 0; --i) ns_minpos_sift(ns, i);
        const int m_head = ns_extract_minpos(ns);
        L.mf[m_head] = L_MERGH;
        int min_l = firstx(r, L, m_head), max_r = lastx(r, L, m_head);
        if (mf) { if (L.ls[m_head] == b_score) mf[0] = m_head; mf[(*mfn)++] = m_head; }
        int body;
        while ((body = ns_extract_minpos(ns)) != -1) {
            L.mf[body] = L_MERGB; L.mh[body] = m_head;
            min_l = imin(min_l, firstx(r, L, body)); max_r = imax(max_r, lastx(r, L, body));
            if (mf) { if (L.ls[body] == b_score) mf[0] = body; mf[(*mfn)++] = body; }
        }
        L.lb[m_head] = imin(firstx(r, L, m_head), min_l);
        L.rb[m_head] = imax(lastx(r, L, m_head), max_r);
        arena_release(r.cx.tmp, mark);
    } else {
        for (int j = 0; j < mbn; ++j) {
            const int li = mb[j];
            if (L.ls[li] == b_score) { L.mf[li] = L_NMERG; if (mf) { mf[0] = li; mf[(*mfn)++] = li; } }
            else { L.mf[li] |= L_DUMP; if (tri_n) tri_n[li] = 0; }
        }
    }
    return mfn_;
}

struct Trig { int32_t *n1, *n2, *off, *cnt; int cap, used; };   // inter-triggers per line (trig_node, lamsa_aln.h:326)

HP_FN void dump_edge_cluster(LSet &L, int ls, int len, const int32_t *mf_row, int mfn_row)
{   // :289-297 / :306-314
    for (int i = 1; i < mfn_row; ++i) {
        L.mf[mf_row[i]] = L_DUMP; HP_STAT(15);
        for (int _j = ls; _j < ls + len; ++_j) {
            const int j = L.rank[_j];
            if (!(L.mf[j] & L_NMERG) && !(L.mf[j] & L_MERGH) && !(L.mf[j] & L_DUMP) && L.mh[j] == mf_row[i]) L.mf[j] = L_DUMP;
        }
    }
}

// line_filter (:122-319) when trg != nullptr, line_filter1 (:321-404) otherwise
HP_NOINL void line_filter(ReadCtx &r, LSet &L, int ls, int len, Trig *trg, int per_max_multi)
{
    const size_t mark = arena_mark(r.cx.tmp);
    // clusters are runs in rank order: cl_off[c] .. cl_off[c+1] index into mb[]; winners of cluster c at mfv[c*?]
    int32_t *mb, *cl_off, *cl_nm, *mfv, *mfn, *spare = nullptr; int spare_n = 0;
    if (L.xtra && 6 * len + 7 <= L.xtra_n) {                           // staged line set: the temporaries next to it in LDS
        mb = L.xtra; cl_off = mb + (len + 1); cl_nm = cl_off + (len + 2); mfv = cl_nm + (len + 1); mfn = mfv + (2 * len + 2);
        spare = mfn + (len + 1); spare_n = L.xtra_n - (6 * len + 7);
    } else {
        mb = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)(len + 1));
        cl_off = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)(len + 2));
        cl_nm = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)(len + 1));      // 1: "not merged" cluster (y == -2)
        mfv = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)(2 * len + 2));  // winners: cluster c at mfv[cl_off[c]+c ..]
        mfn = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)(len + 1));
    }
    if (!mb || !cl_off || !cl_nm || !mfv || !mfn) { arena_release(r.cx.tmp, mark); return; }
    int m_i = -1, nb = 0;
    for (int _i = ls; _i < ls + len; ++_i) {
        const int i = L.rank[_i];
        if (L.mf[i] & L_DUMP) continue;
        if (trg) {
            if (L.mf[i] & L_NMERG) { ++m_i; cl_off[m_i] = nb; cl_nm[m_i] = 1; mb[nb++] = i; }
            else if (L.mf[i] & L_MERGH) { ++m_i; cl_off[m_i] = nb; cl_nm[m_i] = 0; mb[nb++] = i; }
            else { if (m_i < 0) { r.cx.status |= ST_REFEXIT; arena_release(r.cx.tmp, mark); return; } mb[nb++] = i; }
        } else {
            if (L.mf[i] & L_NMERG) continue;
            if (L.mf[i] & L_MERGH) { ++m_i; cl_off[m_i] = nb; cl_nm[m_i] = 0; mb[nb++] = i; }
            else { if (m_i < 0) { r.cx.status |= ST_REFEXIT; arena_release(r.cx.tmp, mark); return; } mb[nb++] = i; }
        }
    }
    cl_off[m_i + 1] = nb;
    for (int c = 0; c <= m_i; ++c) {
        int32_t *mf = mfv + cl_off[c] + c;             // room for (cluster size + 1) entries
        const int mbn = cl_off[c + 1] - cl_off[c];
        if (!trg) { pick_in_cluster(r, L, mb + cl_off[c], mbn, per_max_multi, nullptr, nullptr, spare, spare_n); continue; }
        if (cl_nm[c]) { mf[0] = mb[cl_off[c]]; mfn[c] = 1; continue; }
        const int n = pick_in_cluster(r, L, mb + cl_off[c], mbn, per_max_multi, trg->cnt, mf, spare, spare_n);
        mfn[c] = n;
        for (int ii = 1; ii < n; ++ii) {               // inter-lines (candidate inversions), :236-273
            const int j = mf[ii], _j = L.sel[j];
            for (int k = 0; k < trg->cnt[j]; ++k) {
                int head = -1;
                const int n1 = trg->n1[trg->off[j] + k], n2 = trg->n2[trg->off[j] + k];
                for (int _l = _j + 1; _l < ls + len; ++_l) {
                    const int l = L.rank[_l];
                    if ((L.mf[l] & 0x3) != 0) break;
                    if (firstx(r, L, l) > r.n_seed[n1] && lastx(r, L, l) < r.n_seed[n2]) {
                        const int mfl = r.nd[n2].match_flag;
                        if (mfl == F_MISMATCH || mfl == F_LONG_MISMATCH) {
                            const int s = L.pool[L.start[l]], e = L.pool[L.start[l] + L.len[l] - 1];
                            const int st = r.h_strand[s];
                            if (st == r.h_strand[n1] || r.h_chr[s] != r.h_chr[n1] ||
                                st * r.h_pos[s] < st * r.h_pos[n2] || st * r.h_pos[e] > st * r.h_pos[n1]) continue;
                            L.mf[l] = L_INTER; HP_STAT(14);
                            if (head == -1) { L.mf[l] |= L_NMERG; head = l; }
                            else { L.mf[l] |= L_MERGB; L.mh[l] = head; L.mf[head] = L_INTER | L_MERGH; }
                        }
                    }
                }
            }
        }
    }
    if (trg && m_i > 0) {                              // :279-316
        int a = mfv[cl_off[0] + 0], b = mfv[cl_off[1] + 1];
        if (lastx(r, L, a) - firstx(r, L, a) < 2 && lastx(r, L, b) - firstx(r, L, b) >= 2) dump_edge_cluster(L, ls, len, mfv + cl_off[0], mfn[0]);
        a = mfv[cl_off[m_i] + m_i]; b = mfv[cl_off[m_i - 1] + m_i - 1];
        if (lastx(r, L, a) - firstx(r, L, a) < 2 && lastx(r, L, b) - firstx(r, L, b) >= 2) dump_edge_cluster(L, ls, len, mfv + cl_off[m_i] + m_i, mfn[m_i]);
    }
    arena_release(r.cx.tmp, mark);
}

// line_set_bound (:425) / line_set_bound1 (:496) up to line_remove (:406)
HP_NOINL int set_bound(ReadCtx &r, LSet &L, int ls, int len, Trig *trg)
{
    if (len <= 0) return len;
    const size_t mark = arena_mark(r.cx.tmp);
    int32_t *tmp = (L.xtra && len + 1 <= L.xtra_n) ? L.xtra : (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)(len + 1));
    if (!tmp) return 0;
    sort_endpos(r, L, ls, len, tmp);
    arena_release(r.cx.tmp, mark);
    L.mf[L.rank[ls]] = L_NMERG;
    for (int i = 1; i < len; ++i) line_merge(r, L, L.rank[ls + i], L.rank[ls + i - 1], r.cx.P->ovlp_rat);
    line_filter(r, L, ls, len, trg, r.cx.P->ske_max);
    int cur = ls;
    for (int _l = ls; _l < ls + len; ++_l) { const int l = L.rank[_l]; if (!(L.mf[l] & L_DUMP)) L.rank[cur++] = l; }
    return cur - ls;
}

// ---------------------------------------------------------------- lines -> fragments (frag_dp_path, :1152)
struct FLines {
    int n, nfrag;
    int32_t *line_score, *left_bound, *right_bound, *frag_off;   // per line (frag_off has n+1 entries)
    int32_t *fr_seed_off;                                         // per fragment (+1)
    int32_t *fr_seed;                                             // node indices, fragment seeds in the reference's order
    // CIGARs computed ahead of the fill by the lane-per-job kernel (hp_lanedp.h), four words per slot {offset into jarena,
    // words, target length after clipping at the contig end, 1 = present}: jt[4 * f] for the junction between fragments f and
    // f + 1, gt[4 * p] for the gap in front of the seed at position p of fr_seed
    int32_t *jt, *gt;
    // the same for the two end extensions of a line (frag_head_bound_fix / frag_tail_bound_fix), computed ahead by the job launch of
    // hp_wavejob.h: ht[16 * line] = head, ht[16 * line + 8] = tail, eight words per slot {offset into jarena, words, reference bases and
    // read bases the CIGAR covers, 1 = present, -, -, -}
    int32_t *ht;
    const int32_t *jarena;
};

// Where the fragments of a round are kept.  The one-kernel path keeps them in the wave's slab; the phased path (hp_phase.h)
// hands them from the chaining kernel to the fill kernel through an arena in HBM shared by the batch (`base`, bump
// cursor advanced once per read and round).
struct FlStore { int32_t *base; int64_t cap; unsigned long long *cursor; int64_t got_off; int32_t got_tot; };
HP_INL int flines_words(int line_n, int tot) { return 4 * (line_n + 1) + (tot + 2) + tot + 4 + 8 * (tot + 2) + 16 * (line_n + 1); }
HP_INL void flines_bind(FLines &F, int32_t *m, int line_n, int tot)
{
    F.line_score = m; F.left_bound = m + (line_n + 1); F.right_bound = m + 2 * (line_n + 1); F.frag_off = m + 3 * (line_n + 1);
    F.fr_seed_off = m + 4 * (line_n + 1); F.fr_seed = F.fr_seed_off + (tot + 2);
    F.jt = F.fr_seed + tot + 4; F.gt = F.jt + 4 * (tot + 2);          // fragments <= seeds = tot
    F.ht = F.gt + 4 * (tot + 2);
    F.jarena = nullptr;
}

HP_NOINL bool build_flines(ReadCtx &r, LSet &L, int line_n, FLines &F, FlStore *fs = nullptr)
{
    F.n = 0; F.nfrag = 0;
    if (line_n == 0) return true;
    int tot = 0;
    for (int _l = 0; _l < line_n; ++_l) tot += L.len[L.rank[_l]];
    int32_t *m = nullptr;
    if (fs) {
        const int words = (flines_words(line_n, tot) + 3) & ~3;
        unsigned long long off = 0;
        if (wv::leader()) off = atomicAdd(fs->cursor, (unsigned long long)words);
        off = (unsigned long long)wv::uni64((long long)off);
        if ((int64_t)(off + (unsigned long long)words) > fs->cap) { r.cx.status |= ST_OVERFLOW; return false; }
        m = fs->base + off; fs->got_off = (int64_t)off; fs->got_tot = tot;
    } else m = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)flines_words(line_n, tot));
    if (!m) return false;
    flines_bind(F, m, line_n, tot);
    { HP_G int32_t *gj = (HP_G int32_t *)F.jt; const int nw = 8 * (tot + 2) + 16 * (line_n + 1); for (int b0 = 0; b0 < nw; b0 += 64) { WAVE_FOR(l) { if (b0 + l < nw) gj[b0 + l] = 0; } } }      // nothing precomputed yet
    const lamsa_hp_para *P = r.cx.P;
    // Both passes below walk a line node by node with a decision that depends on the previous one, so they stay
    // sequential; what they read about a node is fetched 64 nodes at a time by the lanes and handed out by readlane.
    const HP_G NodeS *gns = (const HP_G NodeS *)r.nd;
    if (P->aln_mode & 1) {                              // line_filter_overlap, :568-594
        for (int _i = 0; _i < line_n; ++_i) {
            const int li = L.rank[_i];
            int32_t *ni = L.pool + L.start[li]; const int ll = L.len[li];
            HP_G int32_t *gni = (HP_G int32_t *)ni;
            int last_i = 0, st_p = 0, ld_p = 0; long long pos_p = 0;
            for (int j0 = 0; j0 < ll; j0 += 64) {
                wv::Lane<int> plo, phi, ldl, stl, mfl;
                WAVE_FOR(l) {
                    const int j = j0 + l;
                    int a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
                    if (j < ll) { const int c = gni[j]; hp_load16(gns + c, a); hp_load16((const HP_G char *)(gns + c) + 16, b); }
                    plo[l] = a[0]; phi[l] = a[1]; stl[l] = (int)(int8_t)((b[0] >> 16) & 0xff); ldl[l] = (int)(int8_t)((b[0] >> 24) & 0xff); mfl[l] = (b[1] >> 16) & 0xff;
                }
                const int cnt = ll - j0 < 64 ? ll - j0 : 64;
                for (int q = 0; q < cnt; ++q) {
                    const int j = j0 + q;
                    const long long pos_c = (long long)(((unsigned long long)(unsigned)wv::bcast(phi, q) << 32) | (unsigned)wv::bcast(plo, q));
                    const int st = wv::bcast(stl, q), ld_c = wv::bcast(ldl, q), mf_c = wv::bcast(mfl, q);
                    if (j == 0) { pos_p = pos_c; ld_p = ld_c; st_p = st; continue; }
                    const bool ovl = (long long)(P->seed_len + (st == 1 ? ld_p : ld_c)) > (long long)st * (pos_c - pos_p) && mf_c != F_INSERT;
                    if (j < ll - 1) {
                        if (ovl) gni[j] = -1; else { last_i = j; pos_p = pos_c; ld_p = ld_c; st_p = st; }
                    } else if (ll - 1 != last_i && ovl) gni[last_i] = -1;             // the last node stays, the one before it goes (:586-592)
                }
            }
            (void)st_p;
        }
        wv::sync();
    }
    int nf = 0, ns = 0;
    for (int _l = 0; _l < line_n; ++_l) {
        const int li = L.rank[_l];
        const int32_t *ln = L.pool + L.start[li]; const int ll = L.len[li];
        const HP_G int32_t *gln = (const HP_G int32_t *)ln;
        F.frag_off[_l] = nf;
        F.right_bound[_l] = r.seed_all + 1;
        int pre = -1, mf_pre = 0;
        // nodes from the last one backwards; lane q of a block holds node j1 - q
        for (int j1 = ll - 1; j1 >= 0; j1 -= 64) {
            wv::Lane<int> idl, mfl;
            WAVE_FOR(l) {
                const int j = j1 - l;
                int id = -1, mf = 0;
                if (j >= 0) { id = gln[j]; if (id >= 0) { int b[4]; hp_load16((const HP_G char *)(gns + id) + 16, b); mf = (b[1] >> 16) & 0xff; } }
                idl[l] = id; mfl[l] = mf;
            }
            const int cnt = j1 + 1 < 64 ? j1 + 1 : 64;
            for (int q = 0; q < cnt; ++q) {
                const int id = wv::bcast(idl, q);
                if (j1 - q == ll - 1) { pre = id; mf_pre = wv::bcast(mfl, q); F.fr_seed_off[nf] = ns; F.fr_seed[ns++] = pre; continue; }   // FRAG_END: a new fragment opens with its last seed
                if (id < 0) continue;                                      // dropped by the overlap filter
                const int mf = mf_pre;                                     // edge class of the node after this one (`cur`)
                pre = id; mf_pre = wv::bcast(mfl, q);
                if (mf == F_INSERT || mf == F_DELETE || mf == F_MISMATCH || mf == F_LONG_MISMATCH) { ++nf; F.fr_seed_off[nf] = ns; F.fr_seed[ns++] = pre; }
                else if (mf == F_MATCH) F.fr_seed[ns++] = pre;
                else { r.cx.status |= ST_REFEXIT; return false; }          // "[frag dp path] Error: Unknown flag", :1223
            }
        }
        ++nf;
        F.left_bound[_l] = 0;
        F.line_score[_l] = L.ls[li];
    }
    F.frag_off[line_n] = nf; F.fr_seed_off[nf] = ns;
    F.n = line_n; F.nfrag = nf;
    return true;
}

}  // namespace hp
#include "hp_cluster.h"
#include "hp_gaps.h"
namespace hp {

// The loop over the end nodes that branch tracking has left on the stack (:1370-1432): every one becomes a line -- its anchors, the mini
// DPs of its gaps (hp_gaps.h), the inter-line triggers.  A function of its own, called once per read: what it keeps in registers does
// not add to the pressure of the phases around it.  Returns the number of lines, or -1 (status flagged).
HP_NOINL int lines_pop(ReadCtx &r, NScore &ns, LSet &L, Trig &T, int32_t *_line, const Clusters *C)
{
    int l_i = 0, next_start = 0, line_score = 0, line_NM = 0;
    GapCache gc; gc.n = 0; gc.used = 0; gc.cap = 0; gc.ids = nullptr;
    if (C) gapcache_init(r, gc);
    for (;;) {
        int max_node = ns_pop(ns, &line_score, &line_NM);
        if (max_node < 0) break;
        // pool never overflows: every node joins at most one line (TRACKED), plus one slack slot per line
        int32_t *ln = L.pool + next_start;
        T.off[l_i] = T.used; T.cnt[l_i] = 0;
        const int node_i = line_build(r, max_node, ln, _line, &line_score, &line_NM, T, l_i, C, C ? &gc : nullptr);       // anchors, mini DPs of the gaps, triggers (hp_gaps.h)
        if (node_i < 0) return -1;
        { HP_G int32_t *g_ln = (HP_G int32_t *)ln; const int half = node_i / 2;                        // invert the line (:1419-1422), 64 pairs per step
          for (int k0 = 0; k0 < half; k0 += 64) { WAVE_FOR(l) { const int k = k0 + l; if (k < half) { const int t = g_ln[k], u = g_ln[node_i - k - 1]; g_ln[k] = u; g_ln[node_i - k - 1] = t; } } }
          wv::sync(); }
        L.start[l_i] = next_start; L.len[l_i] = node_i; L.ls[l_i] = L.bs[l_i] = line_score; L.nm[l_i] = line_NM;
        L.mf[l_i] = 0; L.mh[l_i] = 0; L.lb[l_i] = L.rb[l_i] = 0;
        ++l_i; next_start += node_i + 1;
        if (r.cx.status & ST_REFEXIT) return -1;
    }
    return l_i;
}

// ---------------------------------------------------------------- round 1: frag_line_BCC, :1305-1445
#ifdef HP_PROF
#define HP_CSTAMP(k) do { const long long now_ = wv::clock(); if (r.prof) r.prof[(k)] += now_ - tc_; tc_ = now_; } while (0)
#else
#define HP_CSTAMP(k) do { } while (0)
#endif

HP_NOINL bool chain_first(ReadCtx &r, FLines &F, FlStore *fs = nullptr)
{
#ifdef HP_PROF
    long long tc_ = wv::clock();
#endif
    const lamsa_hp_para *P = r.cx.P;
    const int seed_out = r.seed_out, H = r.H;
    F.n = 0; F.nfrag = 0;
    int min_n = P->first_loci_thd, min_exist = 0, min_num = 0;
    {                                                                                             // :1315-1323, 64 seeds per step
        const HP_G int64_t *g_hoff = (const HP_G int64_t *)r.hit_off;
        for (int i0 = 0; i0 < seed_out; i0 += 64) {
            wv::Lane<int> few;
            WAVE_FOR(l) { const int i = i0 + l; few[l] = i < seed_out && (int)(g_hoff[i + 1] - g_hoff[i]) <= min_n; }
            min_num += __builtin_popcountll(wv::ballot(few));
        }
        min_exist = min_num > 0;
    }
    const bool all_min = (!min_exist || min_num * 3 < seed_out);                                 // :1324-1331
#ifdef HP_PROF
    long long tq_ = wv::clock(); if (r.prof) r.prof[56] += tq_ - tc_;
#endif
    if (!r.nodes_ready) {
        for (int base = 0; base < H; base += 64) {
            WAVE_FOR(l) {
                const int k = base + l;
                if (k < H) node_set(r, k, -1, 1, r.h_nm[k], F_MATCH, (all_min || mapn(r, r.n_seed[k]) <= min_n) ? MIN_FLAG : MULTI_FLAG);
            }
        }
        wv::sync();
    }
    r.nodes_ready = false;
#ifdef HP_PROF
    { const long long t2_ = wv::clock(); if (r.prof) r.prof[57] += t2_ - tq_; tq_ = t2_; }
#endif
    if (all_min) min_n = P->per_aln_m;
    // the clusters of the read's hits (hp_cluster.h): what frag_min_extend and the main pass below work on
    const size_t cmark = arena_mark(r.cx.tmp);
    Clusters C;
    bool have_cl = false;
    if (seed_out > 1 && HP_CL_CAP_RT(1) > 0) {
        have_cl = clusters_build(r, C, (HP_L uint64_t *)r.cx.lds, r.cx.lds_words / 2);
        if (!have_cl && (r.cx.status & ST_OVERFLOW)) return false;
    }
    // frag_min_extend (:1335-1343): cluster by cluster, together with the main pass below; for the whole read when there are no clusters
    const bool do_me = min_n != P->per_aln_m && seed_out > 1;
    if (do_me && !have_cl) min_extend_all(r, min_n);
#ifdef HP_PROF
    { const long long t2_ = wv::clock(); if (r.prof) { r.prof[58] += t2_ - tq_; r.prof[59] += all_min ? 0 : 1; } }
#endif
    HP_CSTAMP(6);
#if defined(HP_CHAIN_STOP) && HP_CHAIN_STOP == 1
    return true;                 // traffic experiment (tools/chain_stops.sh): nothing after the clusters are cut (the MIN pass runs with the main pass below)
#endif
    if (seed_out > 1) {                                                                           // main pass, :1345-1350
        // cluster by cluster out of LDS; clusters that do not fit LDS through dp_update_range; then the son lists
        bool any_big = false;
        if (!have_cl) dp_update_range(r, hoff(r, 1), H, 0, MIN_FLAG, false, false);               // keys too wide for the packed sort: everything through HBM
        else {
            const HP_G int32_t *g_cs = (const HP_G int32_t *)C.cs, *g_srt = (const HP_G int32_t *)r.srt;
            HP_G uint8_t *g_big = (HP_G uint8_t *)C.big;
            for (int c0 = 0; c0 < C.n_cl; c0 += 63) {
                wv::Lane<int> csl;
                WAVE_FOR(l) { const int c = c0 + l; csl[l] = c <= C.n_cl ? g_cs[c] : H; }
                const int cn = C.n_cl - c0 < 63 ? C.n_cl - c0 : 63;
                wv::Lane<int> csn = csl;                                                          // the next cluster's first rank
                WAVE_FOR(l) { const int c = c0 + l + 1; csn[l] = c <= C.n_cl ? g_cs[c] : H; }
                const bool geo32 = (long long)(r.seed_id[seed_out - 1] - r.seed_id[0] + 1) * P->seed_step <= 0x3fffffffll;
                if (geo32 && HP_CL_CAP_RT(1 << 20) >= HP_CLL_MCAP) {                                   // clusters of two to six hits: one per lane
                    const EdgeK K = edge_consts(P);
                    wv::sync();
                    WAVE_FOR(l) { const int n = csn[l] - csl[l]; if (l < cn && n >= 2 && n <= HP_CLL_MCAP) cluster_lane(r, K, C, r.cx.lds + l, csl[l], n, do_me); }
                    wv::sync();
                }
                for (int q = 0; q < cn; ++q) {
                    const int lo = wv::bcast(csl, q), n = wv::bcast(csl, q + 1) - lo;
                    if (n < 2) continue;                                                          // a lone hit has no predecessor
                    if (geo32 && HP_CL_CAP_RT(1 << 20) >= HP_CLL_MCAP && n <= HP_CLL_MCAP) continue;   // done above
                    if (n <= HP_CL_CAP_RT(r.cx.lds_words / 5) && dp_cluster_lds(r, C, lo, n, do_me)) continue;
                    if (do_me) min_extend_clusters(r, C, lo, lo + n);                             // the cluster goes through HBM: so does its MIN extension
                    for (int i0 = 0; i0 < n; i0 += 64) { WAVE_FOR(l) { if (i0 + l < n) g_big[g_srt[lo + i0 + l]] = 1; } }
                    any_big = true;
                }
            }
            wv::sync();
            if (any_big) dp_update_range(r, hoff(r, 1), H, 0, MIN_FLAG, false, false, C.big);
        }
        if (!have_cl) arena_release(r.cx.tmp, cmark);                  // the clusters stay: the line loop below works on them (hp_gaps.h)
        if (!build_sons(r, (HP_L uint64_t *)r.cx.lds, r.cx.lds_words / 2)) return false;
    }

    HP_CSTAMP(7);
#if defined(HP_CHAIN_STOP) && HP_CHAIN_STOP == 2
    return true;
#endif
    NScore ns;
    if (!ns_alloc(r.cx, ns, H + 1, 0)) return false;
    ns.min_score_thd = 2;
    track_leaves(r, 0, seed_out - 1, MIN_FLAG, ns);                                                 // :1356-1361

    HP_CSTAMP(8);
#if defined(HP_CHAIN_STOP) && HP_CHAIN_STOP == 3
    return true;
#endif
    const int o_l = ns.node_n;
    LSet L;
    Trig T;
    if (!lset_alloc(r.cx, L, 2 * H + o_l + 16, o_l)) return false;
    T.cap = 2 * H + 2 * o_l + 16; T.used = 0;
    T.n1 = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)T.cap);
    T.n2 = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)T.cap);
    T.off = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)(o_l + 1));
    T.cnt = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)(o_l + 1));
    int32_t *_line = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)(H + 2));
    if (!T.n1 || !T.n2 || !T.off || !T.cnt || !_line) return false;
    const int l_i = lines_pop(r, ns, L, T, _line, have_cl ? &C : nullptr);                      // :1370-1432
    if (l_i < 0) return false;
    L.n = l_i;
    HP_CSTAMP(9);
#if defined(HP_CHAIN_STOP) && HP_CHAIN_STOP == 4
    return true;
#endif
    LSet S;
    const bool staged = lset_stage(r, L, l_i, S);                     // the per-line arrays in LDS from here on, when they fit
    LSet &LL = staged ? S : L;
    const int line_n = set_bound(r, LL, 0, l_i, &T);                  // :1435
    const bool okf = build_flines(r, LL, line_n, F, fs);
    HP_CSTAMP(10);
    if (r.prof) { r.prof[14] = o_l; r.prof[15] = H; }
    return okf;
}

// ---------------------------------------------------------------- uncovered regions of the read (aln_reg / get_remain_reg)
struct RegB { int32_t is_rev, chr; int64_t pos; };
struct Regs {                     // sorted covered intervals + the remain regions derived from them
    int n;                        // covered intervals (one per result record), in stable beg order
    int32_t *beg, *end; RegB *rb, *re;           // ref_beg / ref_end of each interval
    int m;                        // remain regions
    int32_t *r_beg, *r_end, *r_bs, *r_bn, *r_es, *r_en;   // read interval + (start,count) into re[] / rb[] lists
};

// ---------------------------------------------------------------- round 2: frag_mini_dp_multi_line, :923-1017
HP_NOINL int multi_line(ReadCtx &r, int left_b, int right_b, const Regs &G, int reg_i, LSet &L)
{
    if (left_b + 1 >= right_b) return 0;
    const lamsa_hp_para *P = r.cx.P;
    const int start = left_b + 1, end = right_b - 1, dp_flag = WHOLE_FLAG;
    nodes_per_init(r, hoff(r, start), hoff(r, end + 1), -1, dp_flag, 1);
    if (start + 1 <= end) dp_update_range(r, hoff(r, start + 1), hoff(r, end + 1), start, dp_flag, false, true);
    const size_t mark = arena_mark(r.cx.tmp);
    NScore ns;
    if (!ns_alloc(r.cx, ns, hoff(r, end + 1) - hoff(r, start) + 1, 0)) return 0;
    ns.min_score_thd = 0;
    track_leaves(r, start, end, dp_flag, ns);
    int l_i = 0, next_start = 0, score = 0, NM = 0;
    for (;;) {
        int rr = ns_pop(ns, &score, &NM);
        if (rr < 0) break;
        int node_i = r.n_node_n[rr] - 1;
        if (l_i >= L.cap || next_start + node_i + 1 > L.pool_cap) { r.cx.status |= ST_OVERFLOW; break; }
        L.start[l_i] = next_start; L.len[l_i] = node_i + 1; L.mf[l_i] = 0; L.mh[l_i] = 0; L.lb[l_i] = L.rb[l_i] = 0;
        next_start += node_i + 1;
        int hit = 0;                                                    // proximity bonus, :979-996
        const int64_t expect = (int64_t)((r.n_seed[rr] - left_b) * P->seed_step);
        for (int i = 0; i < G.r_bn[reg_i] && !hit; ++i) {
            const RegB &b = G.re[G.r_bs[reg_i] + i];
            int64_t d = (r.h_pos[rr] - b.pos) - expect; if (d < 0) d = -d;
            if (r.h_chr[rr] == b.chr && d < P->SV_len_thd) hit = 1;
        }
        for (int i = 0; i < G.r_en[reg_i] && !hit; ++i) {
            const RegB &b = G.rb[G.r_es[reg_i] + i];
            int64_t d = (r.h_pos[rr] - b.pos) - expect; if (d < 0) d = -d;
            if (r.h_chr[rr] == b.chr && d < P->SV_len_thd) hit = 1;
        }
        if (hit) { if (score > 1) score += score / 2; else score++; }
        L.ls[l_i] = L.bs[l_i] = score; L.nm[l_i] = NM;
        int32_t *node = L.pool + L.start[l_i];
        while (rr >= 0) {
            if (node_i < 0) { r.cx.status |= ST_REFEXIT; arena_release(r.cx.tmp, mark); return 0; }
            node[node_i--] = rr;
            rr = r.n_from[rr];
        }
        if (node_i >= 0) { r.cx.status |= ST_REFEXIT; arena_release(r.cx.tmp, mark); return 0; }
        ++l_i;
    }
    arena_release(r.cx.tmp, mark);
    return l_i;
}

// frag_line_remain, :1252-1302
HP_NOINL bool chain_remain(ReadCtx &r, const Regs &G, FLines &F, FlStore *fs = nullptr)
{
    const lamsa_hp_para *P = r.cx.P;
    const int seed_out = r.seed_out, H = r.H;
    F.n = 0; F.nfrag = 0;
    LSet L, T;
    if (!lset_alloc(r.cx, L, H + 16, H + 1) || !lset_alloc(r.cx, T, H + 16, H + 1)) return false;
    int l_n = 0, next_start = 0;
    for (int i = 0; i < G.m; ++i) {
        const int left_id = (G.r_beg[i] + P->seed_inv - 1) / P->seed_step + 1;
        int right_id = (G.r_end[i] - 1) / P->seed_step + 1;
        if (right_id > r.seed_all) right_id -= 1;
        int left = -2, right = -2;
        for (int j = 0; j < seed_out; ++j) if (r.seed_id[j] >= left_id) { left = j - 1; break; }
        if (left == -2) continue;
        for (int j = seed_out - 1; j >= 0; --j) if (r.seed_id[j] <= right_id) { right = j + 1; break; }
        if (right == -2) continue;
        int l = multi_line(r, left, right, G, i, T);                   // trg_dp_line, :1019
        if (r.cx.status & ST_REFEXIT) return false;
        T.n = l;
        l = set_bound(r, T, 0, l, nullptr);
        for (int _j = 0; _j < l; ++_j) {                               // :1288-1295
            const int j = T.rank[_j], d = l_n + _j;
            if (d >= L.cap || next_start + T.len[j] > L.pool_cap) { r.cx.status |= ST_OVERFLOW; return false; }
            L.start[d] = next_start; L.len[d] = T.len[j]; L.lb[d] = T.lb[j]; L.rb[d] = T.rb[j]; L.mf[d] = T.mf[j]; L.mh[d] = T.mh[j];
            L.ls[d] = T.ls[j]; L.bs[d] = T.bs[j]; L.nm[d] = T.nm[j];
            for (int k = 0; k < T.len[j]; ++k) L.pool[next_start + k] = T.pool[T.start[j] + k];
            next_start += T.len[j];
            L.rank[d] = d;
        }
        l_n += l;
    }
    L.n = l_n;
    return build_flines(r, L, l_n, F, fs);
}

}  // namespace hp
