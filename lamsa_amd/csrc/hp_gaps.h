// hp_gaps.h -- frag_mini_dp_line (src/lamsa_dp_con.c:1068-1150) for all gaps of one line at once, ONE GAP PER LANE.
// Included by hp_chain.h.
//
// After the main pass a line is a chain of MIN anchors (n_from links from its end node).  Wherever two consecutive anchors
// are more than one seed slot apart -- and beyond the end node -- frag_line_BCC (:1370-1432) runs a small DP over the hits
// of the repetitive (MULTI) seeds in between: initialised from the left anchor (frag_dp_per_init :766), updated in seed order
// (frag_dp_update :701), closed by a forced update of the right anchor (or by the best end node) and walked back.  A 10-kbp
// noisy read has ~300 such gaps, nearly all with a handful of hits that can be connected to the left anchor at all.  The
// wave-wide routine (mini_line_sets) spends ~1500 wave instructions and 6-8 dependent memory round trips per gap; but the
// gaps of a line do not depend on each other -- their seed ranges are disjoint, and what a pass leaves behind outside its own
// result (scores, predecessors, son flags of the hits it touched) is reset by fnode_set / frag_dp_per_init before anything reads
// it again -- so here every lane takes one gap: it scans the gap's hit range, keeps the (at most HP_GAP_MCAP) hits that
// frag_dp_per_init would activate in its own strip of LDS, runs the same recurrence with the same tie rules over them and
// returns the nodes of the mini line, their edge classes, the score / NM deltas and the right anchor's new state.  Gaps
// with a longer range or more active hits, and passes that start from START, go through mini_line as before.
#pragma once

namespace hp {

#define HP_GAP_MCAP (HP_LANE_STRIP_WORDS / 384 < 6 ? HP_LANE_STRIP_WORDS / 384 : 6)   // active hits of a gap a lane can hold: six words each in the lane's LDS strip, at most six
static_assert(HP_GAP_MCAP >= 2, "the chaining kernels need at least 768 words of LDS per wave");
#ifndef HP_GAP_RANGE
#define HP_GAP_RANGE 96           // hits in the seed range of a gap a lane will scan
#endif
#ifndef HP_GAP_MIN
#define HP_GAP_MIN 6              // lines with fewer gaps run them through the wave-wide routine
#endif
#ifndef HP_WALK_MIN
#define HP_WALK_MIN 16            // clusters of at least this many hits have their lines' anchors walked in LDS
#endif

struct GapOut {                   // per lane
    int n;                        // nodes of the mini line (ascending), or -1: not handled here (fallback), -2: the reference's BUG exit
    int ids[HP_GAP_MCAP], mfs[HP_GAP_MCAP];
    int d_score, d_NM;
    int r_from, r_score, r_NM, r_nn, r_mf;     // the right anchor after the forced update (_tail == 1)
};

// The DP of one gap on one lane over the m hits that frag_dp_per_init has activated (entries 0..m-1 of the lane's LDS strip, ascending hit
// order; word w of entry e at strip[(e * 6 + w) * 64]: 0 position relative to the gap's base | 1 slot_j | 2 sid, len_dif << 16 |
// 3 score << 16, NM | 4 hit | 5 from (0xff: the head) << 24, node_n << 16, son_flag << 8, match_flag): frag_dp_update over the range,
// the end of the line (best end node, or the forced update of the right anchor given by rrel / r_sid / r_ld / r_mf / right_nm when
// r_ok), the walk back.  left: the head (or -1 = START), left_NM its NM (0 for START).
HP_INL void gap_dp_core(const EdgeK &K, HP_L int32_t *strip, int m, int sp, int left, int left_x, int right_x, int tail, int left_NM,
                        bool r_ok, int rrel, int r_sid, int r_ld, int r_mf, int right_nm, GapOut &O)
{
#define GW(e, w) strip[((e) * 6 + (w)) * 64]
    // ---- frag_dp_update over the range (:701-764), targets in ascending hit order
    for (int a = 0; a < m; ++a) {
        const int tpos = GW(a, 0), tsj = GW(a, 1), t2 = GW(a, 2), t3 = GW(a, 3);
        const int tslot = tsj >> 14, tsid = (int)(short)(t2 & 0xffff), tld = (int)(int8_t)((t2 >> 16) & 0xff);
        if (tslot < left_x + 2) continue;                                                // targets are the hits of slots left_x + 2 .. (:1094)
        const int t_score = t3 >> 16, t_NM = t3 & 0xffff;
        int best_hi = -0x7fffffff, best_lo = -1, best_b = -1, best_f = 0, neg_p = -0x7fffffff, neg_b = -1, neg_f = 0, neg_hi = 0;
        for (int b = 0; b < a; ++b) {
            const int qsj = GW(b, 1), qslot = qsj >> 14;
            if (qslot >= tslot) continue;
            const int q5 = GW(b, 5);
            if (sp == 1 && ((q5 >> 8) & 0xff) <= F_MATCH_THD) continue;                 // '+': the candidate already has a match son, :718-720
            const int q2 = GW(b, 2), q3 = GW(b, 3);
            const int flag = gap_edge(K, sp, GW(b, 0), (int)(short)(q2 & 0xffff), (int)(int8_t)((q2 >> 16) & 0xff), tpos, tsid, tld);
            if (flag == F_UNCONNECT) continue;
            const int pos = ((tslot - 1 - qslot) << 14) | (qsj & 16383);                // scan order: seeds descending, hits ascending
            const int cand = (q3 >> 16) + 1 + score_table(flag), nm = (q3 & 0xffff) + t_NM;
            const int hi = (int)(((unsigned)cand << 16) | (unsigned)(65535 - nm)), lo = (1 << 28) - 1 - pos;
            if (hi > best_hi || (hi == best_hi && lo > best_lo)) { best_hi = hi; best_lo = lo; best_b = b; best_f = flag; }
            if (sp == -1 && flag <= F_MATCH_THD && 0 - pos > neg_p) { neg_p = 0 - pos; neg_b = b; neg_f = flag; neg_hi = hi; }   // '-': first match precursor, :726-733
        }
        int w_b = -1, w_f = 0, w_score = t_score, w_nm = t_NM;
        if (neg_b >= 0) { w_b = neg_b; w_f = neg_f; w_score = neg_hi >> 16; w_nm = 65535 - (neg_hi & 0xffff); }
        else if (best_b >= 0) {
            const int cand = best_hi >> 16, nm = 65535 - (best_hi & 0xffff);
            if (cand > w_score || (cand == w_score && nm < w_nm)) { w_b = best_b; w_f = best_f; w_score = cand; w_nm = nm; }
        }
        if (w_b >= 0) {                                                                  // :753-761
            const int q5 = GW(w_b, 5);
            GW(a, 3) = (int)(((unsigned)w_score << 16) | (unsigned)(w_nm & 0xffff));
            GW(a, 5) = (w_b << 24) | ((((q5 >> 16) & 0xff) + 1) << 16) | (GW(a, 5) & 0xff00) | w_f;
            GW(w_b, 5) = (q5 & ~0xff00) | (w_f << 8);
        }
    }
    // ---- the end of the line
    int max_c = -1, max_n = 0, max_score, max_NM = 0, old_score, old_NM;
    if (!tail) {                                                                         // best end node, :1105-1123
        old_score = 1; old_NM = left_NM;
        max_score = old_score;
        int bh = -0x7fffffff, bl = -1, bb = -1;
        for (int b = 0; b < m; ++b) {
            const int q3 = GW(b, 3), qsj = GW(b, 1);
            const int pos = ((right_x - 1 - (qsj >> 14)) << 14) | (qsj & 16383);
            const int hi = (int)(((unsigned)(q3 >> 16) << 16) | (unsigned)(65535 - (q3 & 0xffff))), lo = (1 << 28) - 1 - pos;
            if (hi > bh || (hi == bh && lo > bl)) { bh = hi; bl = lo; bb = b; }
        }
        if (bb >= 0) {
            const int sc = bh >> 16, nm = 65535 - (bh & 0xffff);
            if (sc > max_score || (sc == max_score && nm < max_NM)) { max_c = bb; max_score = sc; max_NM = nm; max_n = (GW(bb, 5) >> 16) & 0xff; }
        }
    } else {                                                                             // forced update of the right anchor, :1125-1134
        old_score = 2 + score_table(r_mf); old_NM = left_NM + right_nm;
        int best_hi = -0x7fffffff, best_lo = -1, best_b = -1, best_f = 0, neg_p = -0x7fffffff, neg_b = -1, neg_f = 0, neg_hi = 0;
        for (int b = 0; b < m && r_ok; ++b) {
            const int qsj = GW(b, 1), qslot = qsj >> 14, q5 = GW(b, 5);
            if (sp == 1 && ((q5 >> 8) & 0xff) <= F_MATCH_THD) continue;
            const int q2 = GW(b, 2), q3 = GW(b, 3);
            const int flag = gap_edge(K, sp, GW(b, 0), (int)(short)(q2 & 0xffff), (int)(int8_t)((q2 >> 16) & 0xff), rrel, r_sid, r_ld);
            if (flag == F_UNCONNECT) continue;
            const int pos = ((right_x - 1 - qslot) << 14) | (qsj & 16383);
            const int cand = (q3 >> 16) + 1 + score_table(flag), nm = (q3 & 0xffff) + old_NM;
            const int hi = (int)(((unsigned)cand << 16) | (unsigned)(65535 - nm)), lo = (1 << 28) - 1 - pos;
            if (hi > best_hi || (hi == best_hi && lo > best_lo)) { best_hi = hi; best_lo = lo; best_b = b; best_f = flag; }
            if (sp == -1 && flag <= F_MATCH_THD && 0 - pos > neg_p) { neg_p = 0 - pos; neg_b = b; neg_f = flag; neg_hi = hi; }
        }
        int w_b = -1, w_f = 0, w_score = old_score, w_nm = old_NM;
        if (neg_b >= 0) { w_b = neg_b; w_f = neg_f; w_score = neg_hi >> 16; w_nm = 65535 - (neg_hi & 0xffff); }
        else if (best_b >= 0) {
            const int cand = best_hi >> 16, nm = 65535 - (best_hi & 0xffff);
            if (cand > w_score || (cand == w_score && nm < w_nm)) { w_b = best_b; w_f = best_f; w_score = cand; w_nm = nm; }
        }
        O.r_mf = r_mf;
        if (w_b >= 0) { O.r_from = GW(w_b, 4); O.r_nn = ((GW(w_b, 5) >> 16) & 0xff) + 1; O.r_mf = w_f; max_c = w_b; }
        O.r_score = w_score; O.r_NM = w_nm;
        max_score = w_score; max_NM = w_nm; max_n = O.r_nn - 1;
    }
    // ---- walk back to the head (:1136-1147)
    {
        int c = max_c, node_i = max_n - 1;
        bool bad = max_n > HP_GAP_MCAP;
        while (c >= 0 && !bad) {
            if (node_i < 0) { bad = true; break; }
            O.ids[node_i] = GW(c, 4); O.mfs[node_i] = GW(c, 5) & 0xff; --node_i;
            const int f = (GW(c, 5) >> 24) & 0xff;
            c = f == 0xff ? -1 : f;
        }
        if (node_i >= 0) bad = true;
        if (bad) { O.n = -2; return; }
    }
#undef GW
    O.n = max_n; O.d_score = max_score - old_score; O.d_NM = max_NM - old_NM;
}

// One gap on one lane.  left >= 0 (the head), right >= 0 when tail, right_x = right's slot (or seed_out).  LDS strip: word w of
// entry e at strip[(e * 6 + w) * 64].
HP_INL void gap_lane(const ReadCtx &r, const EdgeK &K, HP_L int32_t *strip, int left, int right, int left_x, int right_x, int tail, GapOut &O)
{
    const HP_G NodeS *ns = (const HP_G NodeS *)r.nd;
    const HP_G int16_t *g_hnm = (const HP_G int16_t *)r.h_nm;
    const HP_G int64_t *g_hoff = (const HP_G int64_t *)r.hit_off;
    O.n = -1; O.d_score = 0; O.d_NM = 0; O.r_from = left; O.r_score = 0; O.r_NM = 0; O.r_nn = 1; O.r_mf = 0;
    const int k_lo = (int)(g_hoff[left_x + 1] - r.hb), k_hi = (int)(g_hoff[right_x] - r.hb);
#ifdef HP_PROF
    if (r.prof) atomicAdd((unsigned long long *)&r.prof[11], (unsigned long long)(k_hi - k_lo));
    if (k_hi - k_lo > HP_GAP_RANGE) { if (r.prof) atomicAdd((unsigned long long *)&r.prof[13], 1ull); return; }
#endif
    if (k_hi - k_lo > HP_GAP_RANGE) return;
    const NodeS Fh = node_load(ns + left);
    const int head_nm = g_hnm[left], sp = Fh.strand;
    if ((long long)(r.seed_id[r.seed_out - 1] - r.seed_id[0] + 1) * K.seed_step > 0x3fffffffll) return;
#define GW(e, w) strip[((e) * 6 + (w)) * 64]
    // ---- frag_dp_per_init over the range (:766-784, :1086-1091): the hits the head can be connected to
    int m = 0;
    // four records in flight per lane: the loop is a chain of dependent HBM round trips otherwise (a gap's range is ~40 hits)
    for (int k0 = k_lo; k0 < k_hi; k0 += 4) {
        NodeS Qs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) Qs[u] = node_load(ns + (k0 + u < k_hi ? k0 + u : k_hi - 1));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u;
            if (k >= k_hi) break;
            const NodeS &Q = Qs[u];
            const int df = Q.dp_flag;
            if (df != MULTI_FLAG && df != 0 - MULTI_FLAG) continue;
            if (Q.chr != Fh.chr || Q.strand != sp) continue;
            const long long rel = Q.pos - Fh.pos;
            if (rel > 0x3fffffffll || rel < -0x3fffffffll) continue;                   // far beyond any connectable distance
            const int flag = gap_edge(K, sp, 0, Fh.sid, Fh.len_dif8, (int)rel, Q.sid, Q.len_dif8);
            if (flag == F_UNCONNECT) continue;
#ifdef HP_PROF
            if (m >= HP_GAP_MCAP && r.prof) atomicAdd((unsigned long long *)&r.prof[13], 1ull);
#endif
            if (m >= HP_GAP_MCAP) return;                                              // too many for a lane
            GW(m, 0) = (int)rel; GW(m, 1) = Q.slot_j; GW(m, 2) = ((int)Q.sid & 0xffff) | (((int)Q.len_dif8 & 0xff) << 16);
            GW(m, 3) = (int)(((unsigned)(2 + score_table(flag)) << 16) | (unsigned)((g_hnm[k] + head_nm) & 0xffff));
            GW(m, 4) = k; GW(m, 5) = (0xff << 24) | (1 << 16) | (F_INIT << 8) | flag;   // from (0xff = the head) | node_n | son_flag | match_flag
            ++m;
        }
    }
    // ---- frag_dp_update over the range, the end of the line, the walk back
    bool r_ok = false; int rrel = 0, r_sid = 0, r_ld = 0, r_mf = 0, right_nm = 0;
    if (tail) {
        const NodeS Rt = node_load(ns + right);
        const long long rr = Rt.pos - Fh.pos;
        right_nm = g_hnm[right]; r_sid = Rt.sid; r_ld = Rt.len_dif8; r_mf = Rt.match_flag; rrel = (int)rr;
        r_ok = Rt.chr == Fh.chr && Rt.strand == sp && rr <= 0x3fffffffll && rr >= -0x3fffffffll;
    }
#undef GW
    gap_dp_core(K, strip, m, sp, left, left_x, right_x, tail, head_nm, r_ok, rrel, r_sid, r_ld, r_mf, right_nm, O);
}

// The gaps of a line that no lane has taken (o_lane == 0), one after the other through mini_line.  A function of its own on purpose:
// mini_line_sets needs ~100 VGPRs, so whatever the caller holds in registers at the call is saved to and restored from scratch
// around it -- 80 calls per read.  Inside line_build that was ~50 dwords per lane and call (12 KB per wave, most of the chaining
// kernel's HBM traffic: the scratch of 4 096 waves does not stay in L2); here it is the handful of values this loop needs.
// gp: the per-gap arrays of line_build (GA entries each, in the order left, right, lx, rx, tail, after, o_n, o_off, ..., o_lane at 15).
struct GapRest { int pool_n, d_score, d_NM; };
HP_NOINL MiniR mini_line_far(ReadCtx &r, int left, int right, int right_x, int32_t *line, int _head, int _tail)
{
    return mini_line(r, left, right, right_x, line, _head, _tail);
}
HP_NOINL GapRest gaps_one_by_one(ReadCtx &r, int G, int32_t *gp, int GA, int32_t *pool, int pool_n, int32_t *_line)
{
    Ctx &cx = r.cx;
    const int H = r.H;
    const int32_t *g_left = gp, *g_right = gp + GA, *g_rx = gp + 3 * GA, *g_tail = gp + 4 * GA, *o_lane = gp + 15 * GA;
    int32_t *o_n = gp + 6 * GA, *o_off = gp + 7 * GA;
    GapRest R; R.pool_n = pool_n; R.d_score = 0; R.d_NM = 0;
    for (int g0 = 0; g0 < G; g0 += 64) {
        wv::Lane<int> todo, lf, rt, rx, tl;                                             // their parameters by readlane, not by a load per gap
        WAVE_FOR(l) {
            const int g = g0 + l;
            todo[l] = g < G && !o_lane[g];
            lf[l] = g < G ? g_left[g] : -1; rt[l] = g < G ? g_right[g] : -1; rx[l] = g < G ? g_rx[g] : 0; tl[l] = g < G ? g_tail[g] : 0;
        }
        unsigned long long m = wv::ballot(todo);
        while (m) {
            const int q = __builtin_ctzll(m), g = g0 + q;
            m &= m - 1;
            // the usual case straight into the register routine; everything else (longer ranges: reach_run and the listing, the memory
            // version) behind a call of its own, so that this loop's frame stays small
            const int left = wv::bcast(lf, q), right = wv::bcast(rt, q), right_x = wv::bcast(rx, q), tail = wv::bcast(tl, q);
            const int k_n = hoff(r, right_x) - hoff(r, nx(r, left) + 1);
            MiniR mr; mr.n = -1; mr.d_score = 0; mr.d_NM = 0;
            if (k_n <= 64) mr = mini_line_sets<1>(r, left, right, right_x, _line, 1, tail, k_n, nullptr);
            if (mr.n < 0) mr = mini_line_far(r, left, right, right_x, _line, 1, tail);
            const int n = mr.n, ds = mr.d_score, dn = mr.d_NM;
            if (cx.status & ST_REFEXIT) { R.pool_n = -1; return R; }
            if (R.pool_n + n > H + 8) { cx.status |= ST_OVERFLOW; R.pool_n = -1; return R; }
            o_n[g] = n; o_off[g] = R.pool_n;
            wv::sync();
            for (int k0 = 0; k0 < n; k0 += 64) { WAVE_FOR(l) { if (k0 + l < n) pool[R.pool_n + k0 + l] = _line[k0 + l]; } }
            R.pool_n += n; R.d_score += ds; R.d_NM += dn;
            wv::sync();
        }
    }
    return R;
}

// ---------------------------------------------------------------- the gaps of a line from the point of view of its CLUSTER
// Every node of a line lies in one cluster of the read's hits (hp_cluster.h: no edge other than F_CHR_DIF / F_UNCONNECT joins two
// clusters), and so does every hit a gap's pass can put on the line: frag_dp_per_init keeps what the head connects to, a pass from
// START is only read along the chain into its right anchor.  So instead of every gap scanning the hits of its seed range (all ~40
// hits of the repetitive seeds in between, from every cluster, for each of a read's ~330 gaps), ONE pass over the hits of the
// line's cluster (C.csrt: ascending hit order) finds, for every hit that a mini DP may use at all (dp_flag +-MULTI), the gap its seed
// slot falls into -- a slot -> gap table in LDS -- and whether that gap's head connects to it; the survivors, in ascending hit order
// and therefore grouped by gap, are what the mini DPs run on: one gap per lane (gap_dp_core) for up to HP_GAP_MCAP hits, the
// wave-wide routine on the listed hits beyond that.  Gaps without a survivor -- nearly all of them at the read's true locus, where
// frag_min_extend has already promoted the colinear hit of every repetitive seed -- cost nothing: their mini line is empty and the
// forced update of their right anchor finds no candidate (what it would leave in the anchor's score / NM / node_n is never read
// again: the anchor is TRACKED, no later pass initialises, updates or scans it).
// gp: the per-gap arrays of line_build.  Returns pool_n >= 0, -1 (status flagged) or -2: not applicable (the caller scans by seed range).
#ifndef HP_GAPTAB_CAP_RT
#define HP_GAPTAB_CAP_RT(cap) (cap)          // slots the LDS table holds; the tests' CPU build lowers it
#endif
#ifndef HP_GAP_MCAP_RT
#define HP_GAP_MCAP_RT(cap) (cap)            // survivors of a gap a lane takes; the tests' CPU build lowers it
#endif
// HP_STAT slots: 0 lines by cluster, 1 lines by seed range, 2 gaps in lanes, 3 of them from START, 4 gaps through the wave-wide routine, 5 of them through memory,
// 6-8 lines in clusters of <= 6 / <= 16 / more hits, 9 wave jobs published (hp_phase.h), 10 an uncovered region at the read's end
// (hp_align.h), 11-13 the F_INSERT classes and the MULTI re-update of the k-mer split mapper (hp_split.h), 14 inter-lines, 15 dumped edge clusters
// The +-MULTI hits of the large clusters a read's lines have visited so far (ascending hit order, as in C.csrt): the ~20 lines of the read's
// true locus share one cluster of several hundred hits, of which a few dozen are left for the mini DPs -- the first line lists them, the
// others scan the list.  (A hit that a line has taken since is TRACKED; every scan tests the flag it loads anyway.)
#ifndef HP_GAPCACHE_MIN
#define HP_GAPCACHE_MIN 96           // clusters of fewer hits are scanned directly
#endif
struct GapCache { int n, used, cap; int lo[4], off[4], cnt[4]; int32_t *ids; };
HP_INL void gapcache_init(ReadCtx &r, GapCache &gc) { gc.n = 0; gc.used = 0; gc.cap = r.H; gc.ids = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)(r.H + 64)); if (!gc.ids) gc.cap = 0; }

HP_HOT GapRest gaps_by_cluster(ReadCtx &r, const Clusters &C, int max_node, int G, int32_t *gp, int GA, int32_t *pool, int32_t *pool_mf, int32_t *_line, GapCache *gc = nullptr, int cl_lo = -1, int cl_n = 0)
{
    Ctx &cx = r.cx;
    const int H = r.H, seed_out = r.seed_out;
    const EdgeK K = edge_consts(cx.P);
    GapRest R; R.pool_n = -2; R.d_score = 0; R.d_NM = 0;
    if (seed_out > HP_GAPTAB_CAP_RT(2 * cx.lds_words) || G > 32767 || HP_GAP_MCAP * 6 * 64 > cx.lds_words) return R;
    if ((long long)(r.seed_id[seed_out - 1] - r.seed_id[0] + 1) * K.seed_step > 0x3fffffffll) return R;           // 32-bit geometry below
    const HP_G NodeS *ns = (const HP_G NodeS *)r.nd;
    HP_G NodeS *gd = (HP_G NodeS *)r.nd;
    const HP_G int16_t *g_hnm = (const HP_G int16_t *)r.h_nm;
    const HP_G int32_t *g_csrt = (const HP_G int32_t *)C.csrt;
    HP_G int32_t *g_from = (HP_G int32_t *)r.n_from, *g_node_n = (HP_G int32_t *)r.n_node_n;
    const int lo = cl_lo >= 0 ? cl_lo : C.cl_lo_r[r.rnk[max_node]], n_c = cl_lo >= 0 ? cl_n : C.ce[lo] - lo;       // the line's cluster (line_build has looked it up already)
    if ((long long)(n_c - 1) * C.reach > 0x3fffffffll) return R;      // neighbours of a cluster are at most `reach` apart: its span fits 30 bits
    const int sp = r.h_strand[max_node];
    const HP_G int32_t *g_left = (const HP_G int32_t *)gp, *g_right = (const HP_G int32_t *)(gp + GA), *g_lx = (const HP_G int32_t *)(gp + 2 * GA),
                       *g_rx = (const HP_G int32_t *)(gp + 3 * GA), *g_tail = (const HP_G int32_t *)(gp + 4 * GA);
    HP_G int32_t *o_n = (HP_G int32_t *)(gp + 6 * GA), *o_off = (HP_G int32_t *)(gp + 7 * GA), *o_s0 = (HP_G int32_t *)(gp + 8 * GA), *o_m = (HP_G int32_t *)(gp + 9 * GA),
                 *o_lane = (HP_G int32_t *)(gp + 15 * GA);
    const size_t mark = arena_mark(cx.tmp);
    int32_t *sv = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 8 * (size_t)(n_c + 1));       // the survivors: six strip words, the gap, a spare
    int32_t *s_id = (int32_t *)arena_alloc(cx, sizeof(int32_t) * (size_t)(n_c + 64));          // their hits, as a plain list
    int32_t *rs = (int32_t *)arena_alloc(cx, sizeof(int32_t) * (size_t)(n_c + 2));             // first survivor of every run (= gap with survivors)
    if (!sv || !s_id || !rs) { arena_release(cx.tmp, mark); R.pool_n = -1; return R; }
    HP_G int32_t *g_sv = (HP_G int32_t *)sv, *g_sid = (HP_G int32_t *)s_id, *g_rs = (HP_G int32_t *)rs;
    // ---- slot -> gap (16 bits per slot, -1: the slot of an anchor, or outside every gap)
    HP_L int16_t *tab = (HP_L int16_t *)cx.lds;
    wv::sync();                                                    // whatever used this LDS before is done
    { const int nw = (seed_out + 1) / 2; for (int w0 = 0; w0 < nw; w0 += 64) { WAVE_FOR(l) { if (w0 + l < nw) cx.lds[w0 + l] = -1; } } }
    wv::sync();
    for (int g0 = 0; g0 < G; g0 += 64) {
        WAVE_FOR(l) {
            const int g = g0 + l;
            if (g < G) { o_n[g] = 0; o_off[g] = 0; o_lane[g] = 1; const int x1 = g_rx[g]; for (int x = g_lx[g] + 1; x < x1; ++x) tab[x] = (int16_t)g; }
        }
    }
    wv::sync();
    // ---- the hits to look at: the cluster's, or the +-MULTI ones among them listed by an earlier line of this read
    const HP_G int32_t *g_src = g_csrt + lo; int n_src = n_c;
    if (gc && n_c >= HP_GAPCACHE_MIN && gc->cap > 0) {
        int e = -1, e_off = 0, e_cnt = 0;                              // (constant indices only: the directory stays in registers)
#pragma unroll
        for (int q = 0; q < 4; ++q) if (q < gc->n && gc->lo[q] == lo) { e = q; e_off = gc->off[q]; e_cnt = gc->cnt[q]; }
        if (e < 0 && gc->n < 4 && gc->used + n_c <= gc->cap) {
            e = gc->n++; e_off = gc->used;
            HP_G int32_t *g_dst = (HP_G int32_t *)(gc->ids + gc->used);
            int cnt = 0;
            for (int i0 = 0; i0 < n_c; i0 += 64) {
                wv::Lane<int> isml, idl;
                WAVE_FOR(l) {
                    const int i = i0 + l;
                    int v = 0, id = 0;
                    if (i < n_c) { id = g_csrt[lo + i]; int b[4]; hp_load16((const HP_G char *)(ns + id) + 16, b); const int df = (int)(int8_t)(b[1] & 0xff); v = df == MULTI_FLAG || df == 0 - MULTI_FLAG; }
                    isml[l] = v; idl[l] = id;
                }
                const unsigned long long m = wv::ballot(isml);
                WAVE_FOR(l) { if (isml[l]) g_dst[cnt + __builtin_popcountll(m & ((1ull << l) - 1))] = idl[l]; }
                cnt += __builtin_popcountll(m);
            }
            e_cnt = cnt;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (q == e) { gc->lo[q] = lo; gc->off[q] = e_off; gc->cnt[q] = cnt; }
            gc->used += cnt;
            wv::sync();
        }
        if (e >= 0) { g_src = (const HP_G int32_t *)(gc->ids + e_off); n_src = e_cnt; }
    }
    // ---- one pass over them: frag_dp_per_init (:766-784, :1086-1091) for every gap at once
    int n_surv = 0, n_eval = 0;
    for (int i0 = 0; i0 < n_src; i0 += 64) {
        wv::Lane<int> keep, u0, u1, u2, u3, u4, u5, ug, evl;
        WAVE_FOR(l) {
            const int i = i0 + l;
            int ev = 0;
            int kp = 0, w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, wg = 0;
            if (i < n_src) {
                const int id = g_src[i];
                int a[4], b[4];
                hp_load16(ns + id, a); hp_load16((const HP_G char *)(ns + id) + 16, b);
                const int df = (int)(int8_t)(b[1] & 0xff);
                if (df == MULTI_FLAG || df == 0 - MULTI_FLAG) {
                    const int g = tab[a[3] >> 14];
                    if (g >= 0) {
                        const int left = g_left[g];
                        const NodeS F = node_load(ns + (left >= 0 ? left : g_right[g]));      // the gap's base: the head, or the right anchor of a pass from START
                        const long long qpos = (long long)(((unsigned long long)(unsigned)a[1] << 32) | (unsigned)a[0]);
                        const int rel = (int)(qpos - F.pos);
                        const int qsid = (int)(int16_t)(b[0] & 0xffff), qld = (int)(int8_t)((b[0] >> 24) & 0xff), nm = g_hnm[id];
                        w0 = rel; w1 = a[3]; w2 = (qsid & 0xffff) | ((qld & 0xff) << 16); w4 = id; wg = g;
                        if (left >= 0) {
                            const int flag = gap_edge(K, sp, 0, F.sid, F.len_dif8, rel, qsid, qld);
                            ev = 1;
                            if (flag != F_UNCONNECT) {
                                kp = 1;
                                w3 = (int)(((unsigned)(2 + score_table(flag)) << 16) | (unsigned)((nm + g_hnm[left]) & 0xffff));
                                w5 = (0xff << 24) | (1 << 16) | (F_INIT << 8) | flag;       // from (0xff = the head) | node_n | son_flag | match_flag
                            }
                        } else {                                                             // from START: fnode_set(START, 1, NM, F_MATCH), :771
                            kp = 1; w3 = (int)((1u << 16) | (unsigned)(nm & 0xffff)); w5 = (0xff << 24) | (1 << 16) | (F_INIT << 8) | F_MATCH;
                        }
                    }
                }
            }
            keep[l] = kp; u0[l] = w0; u1[l] = w1; u2[l] = w2; u3[l] = w3; u4[l] = w4; u5[l] = w5; ug[l] = wg; evl[l] = ev;
        }
        n_eval += __builtin_popcountll(wv::ballot(evl));
        const unsigned long long m = wv::ballot(keep);
        if (!m) continue;
        WAVE_FOR(l) {
            if (keep[l]) {
                const int at = n_surv + __builtin_popcountll(m & ((1ull << l) - 1));
                hp_store16(g_sv + 8 * (size_t)at, u0[l], u1[l], u2[l], u3[l]); hp_store16(g_sv + 8 * (size_t)at + 4, u4[l], u5[l], ug[l], 0);
                g_sid[at] = u4[l];
            }
        }
        n_surv += __builtin_popcountll(m);
    }
    r.n_pairs += n_eval;
    wv::sync();
    // ---- runs of equal gap (ascending hits = ascending slots = descending gaps: a gap's survivors are consecutive)
    int n_runs = 0;
    for (int j0 = 0; j0 < n_surv; j0 += 64) {
        wv::Lane<int> st;
        WAVE_FOR(l) { const int j = j0 + l; st[l] = j < n_surv && (j == 0 || g_sv[8 * (size_t)j + 6] != g_sv[8 * (size_t)(j - 1) + 6]); }
        const unsigned long long m = wv::ballot(st);
        WAVE_FOR(l) { if (st[l]) g_rs[n_runs + __builtin_popcountll(m & ((1ull << l) - 1))] = j0 + l; }
        n_runs += __builtin_popcountll(m);
    }
    WAVE_FOR(l) { if (l == 0) g_rs[n_runs] = n_surv; }
    wv::sync();
    // ---- the mini DPs, one gap per lane
    int pool_n = 0, d_score = 0, d_NM = 0;
    bool any_wide = false;
#define GW(e, w) strip[((e) * 6 + (w)) * 64]
    for (int q0 = 0; q0 < n_runs; q0 += 64) {
        wv::Lane<int> nn, ds, dn, gl;
        WAVE_FOR(l) {
            const int q = q0 + l;
            GapOut O; O.n = 0; O.d_score = 0; O.d_NM = 0; O.r_from = -1; O.r_score = 0; O.r_NM = 0; O.r_nn = 1; O.r_mf = 0;
            int g = -1;
            if (q < n_runs) {
                const int s0 = g_rs[q], m = g_rs[q + 1] - s0;
                g = g_sv[8 * (size_t)s0 + 6];
                if (m > HP_GAP_MCAP_RT(HP_GAP_MCAP)) { O.n = -1; o_lane[g] = 0; o_s0[g] = s0; o_m[g] = m; }
                else {
                    HP_L int32_t *strip = cx.lds + l;
                    HP_STAT(2);
                    for (int e = 0; e < m; ++e) {
                        int u[4], v[4];
                        hp_load16(g_sv + 8 * (size_t)(s0 + e), u); hp_load16(g_sv + 8 * (size_t)(s0 + e) + 4, v);
                        GW(e, 0) = u[0]; GW(e, 1) = u[1]; GW(e, 2) = u[2]; GW(e, 3) = u[3]; GW(e, 4) = v[0]; GW(e, 5) = v[1];
                    }
                    const int left = g_left[g], right = g_right[g], tail = g_tail[g];
                    long long base = 0; int left_NM = 0, rrel = 0, r_sid = 0, r_ld = 0, r_mf = 0, right_nm = 0;
                    if (left >= 0) { const NodeS Fh = node_load(ns + left); base = Fh.pos; left_NM = g_hnm[left]; }
                    if (tail) {
                        const NodeS Rt = node_load(ns + right);
                        if (left < 0) base = Rt.pos;
                        rrel = (int)(Rt.pos - base); r_sid = Rt.sid; r_ld = Rt.len_dif8; r_mf = Rt.match_flag; right_nm = g_hnm[right];
                    }
                    O.r_from = left;
                    if (left < 0) HP_STAT(3);
                    gap_dp_core(K, strip, m, sp, left, g_lx[g], g_rx[g], tail, left_NM, tail != 0, rrel, r_sid, r_ld, r_mf, right_nm, O);
                    if (O.n >= 0 && tail) {                        // the right anchor after its forced update (:1125-1134)
                        g_from[right] = O.r_from; gd[right].score = O.r_score; gd[right].NM = O.r_NM; g_node_n[right] = O.r_nn; gd[right].match_flag = (uint8_t)O.r_mf;
                    }
                }
            }
            nn[l] = O.n; ds[l] = O.n >= 0 ? O.d_score : 0; dn[l] = O.n >= 0 ? O.d_NM : 0; gl[l] = g;
            // the nodes of the lane's mini line, staged in its strip until the pool offsets of the group are known
            for (int k = 0; k < HP_GAP_MCAP; ++k) { if (k < O.n) { cx.lds[(k * 6 + 0) * 64 + l] = O.ids[k]; cx.lds[(k * 6 + 1) * 64 + l] = O.mfs[k]; } }
        }
        wv::Lane<int> bad, wide, cntl;
        WAVE_FOR(l) { bad[l] = nn[l] == -2; wide[l] = nn[l] == -1; cntl[l] = nn[l] > 0 ? nn[l] : 0; }
        if (wv::ballot(bad) != 0) { cx.status |= ST_REFEXIT; arena_release(cx.tmp, mark); R.pool_n = -1; return R; }    // "[frag mini dp] BUG" exit, :1140
        if (wv::ballot(wide) != 0) any_wide = true;
        wv::Lane<int> pre = cntl;
        wv::scan_add_excl(pre);
        const int tot = wv::reduce_sum(cntl);
        if (pool_n + tot > H + 8) { cx.status |= ST_OVERFLOW; arena_release(cx.tmp, mark); R.pool_n = -1; return R; }
        WAVE_FOR(l) {
            if (nn[l] > 0) {
                const int g = gl[l], off = pool_n + pre[l];
                o_n[g] = nn[l]; o_off[g] = off;
                for (int k = 0; k < nn[l]; ++k) {
                    const int id = cx.lds[(k * 6 + 0) * 64 + l], mf = cx.lds[(k * 6 + 1) * 64 + l];
                    pool[off + k] = id; pool_mf[off + k] = mf; gd[id].match_flag = (uint8_t)mf;
                }
            }
        }
        pool_n += tot;
        d_score += wv::reduce_sum(ds); d_NM += wv::reduce_sum(dn);
        wv::sync();
    }
#undef GW
    // ---- gaps with more survivors than a lane holds: the wave-wide routine on the listed hits, one gap after the other
    if (any_wide) {
        for (int g0 = 0; g0 < G; g0 += 64) {
            wv::Lane<int> todo, lf, rt, rx, tl, s0l, ml;
            WAVE_FOR(l) {
                const int g = g0 + l;
                todo[l] = g < G && !o_lane[g];
                lf[l] = g < G ? g_left[g] : -1; rt[l] = g < G ? g_right[g] : -1; rx[l] = g < G ? g_rx[g] : 0; tl[l] = g < G ? g_tail[g] : 0;
                s0l[l] = todo[l] ? o_s0[g] : 0; ml[l] = todo[l] ? o_m[g] : 0;
            }
            unsigned long long m = wv::ballot(todo);
            while (m) {
                const int q = __builtin_ctzll(m), g = g0 + q;
                m &= m - 1;
                const int left = wv::bcast(lf, q), right = wv::bcast(rt, q), right_x = wv::bcast(rx, q), tail = wv::bcast(tl, q), s0 = wv::bcast(s0l, q), cnt = wv::bcast(ml, q);
                MiniR mr; mr.n = -1; mr.d_score = 0; mr.d_NM = 0;
                if (cnt <= 64) mr = mini_line_sets<1>(r, left, right, right_x, _line, 1, tail, cnt, s_id + s0);
                else if (cnt <= 64 * HP_MS_MAX_SETS) mr = mini_line_sets<HP_MS_MAX_SETS>(r, left, right, right_x, _line, 1, tail, cnt, s_id + s0);
                HP_STAT(4);
                if (mr.n < 0) { HP_STAT(5); mr = mini_line_mem(r, left, right, right_x, _line, 1, tail); }
                const int n = mr.n, dsc = mr.d_score, dnm = mr.d_NM;
                if (cx.status & ST_REFEXIT) { arena_release(cx.tmp, mark); R.pool_n = -1; return R; }
                if (pool_n + n > H + 8) { cx.status |= ST_OVERFLOW; arena_release(cx.tmp, mark); R.pool_n = -1; return R; }
                o_n[g] = n; o_off[g] = pool_n;
                wv::sync();
                for (int k0 = 0; k0 < n; k0 += 64) { WAVE_FOR(l) { if (k0 + l < n) pool[pool_n + k0 + l] = _line[k0 + l]; } }
                pool_n += n; d_score += dsc; d_NM += dnm;
                wv::sync();
            }
        }
    }
    arena_release(cx.tmp, mark);
    R.pool_n = pool_n; R.d_score = d_score; R.d_NM = d_NM;
    return R;
}

// ---------------------------------------------------------------- one line of frag_line_BCC's loop (:1370-1432)
// The anchors of the line from its end node `max_node` back to START, the mini DPs of all its gaps (one per lane where
// possible, mini_line otherwise), the nodes in read order in ln[], the inter-line triggers (:1384-1386, :1404-1414).  Returns the
// number of nodes, or -1 (status flagged).  `_line`: scratch of H + 2 words for mini_line.
HP_HOT int line_build(ReadCtx &r, int max_node, int32_t *ln, int32_t *_line, int *line_score, int *line_NM, Trig &T, int l_i, const Clusters *C = nullptr, GapCache *gc = nullptr)
{
    Ctx &cx = r.cx;
    const int H = r.H, seed_out = r.seed_out;
    const EdgeK K = edge_consts(cx.P);
    const size_t mark = arena_mark(cx.tmp);
    HP_G NodeS *gd = (HP_G NodeS *)r.nd;
    HP_G int32_t *g_from = (HP_G int32_t *)r.n_from, *g_node_n = (HP_G int32_t *)r.n_node_n;
    const HP_G int32_t *g_seed = (const HP_G int32_t *)r.n_seed;
    // ---- the anchors, end node first (the chain the main pass and branch tracking left in n_from)
    int32_t *anc = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 2 * (size_t)(H + 2));
    if (!anc) { arena_release(cx.tmp, mark); return -1; }
    int32_t *anc_x = anc + (H + 2);
    int A = 0;
#if defined(HP_PROF) && !defined(HP_PROF_TRACK)
    long long tl_ = wv::clock();
#define HP_LSTAMP(k) do { const long long now_ = wv::clock(); if (r.prof) r.prof[(k)] += now_ - tl_; tl_ = now_; } while (0)
#else
#define HP_LSTAMP(k) do { } while (0)
#endif
    HP_G int32_t *g_anc = (HP_G int32_t *)anc, *g_ancx = (HP_G int32_t *)anc_x;
    bool walked = false;
    int cl_lo = -1, cl_n = 0;                                  // the line's cluster, when the read's hits have been clustered
    if (C) {
        // A line of a large cluster (the read's true locus: a few hundred anchors) is a pointer chase of that many dependent loads.  Its
        // cluster's predecessors are fetched into LDS first -- as places in the cluster's rank range, 64 hits per step -- and chased there.
        const HP_G int32_t *g_srt = (const HP_G int32_t *)r.srt, *g_rnk = (const HP_G int32_t *)r.rnk;
        const int rk = r.rnk[max_node], lo = C->cl_lo_r[rk], n_c = C->ce[lo] - lo;
        cl_lo = lo; cl_n = n_c;
        // (worth it for a long line only: the number of nodes the main pass counted up to the end node says how long)
        if (n_c <= 6) HP_STAT(6); else if (n_c <= 16) HP_STAT(7); else HP_STAT(8);
        if (n_c >= HP_WALK_MIN && n_c <= cx.lds_words && r.n_node_n[max_node] >= HP_WALK_MIN) {
            wv::sync();                                                // whatever used this LDS before is done
            for (int i0 = 0; i0 < n_c; i0 += 64) {
                WAVE_FOR(l) {
                    const int i = i0 + l;
                    if (i < n_c) { const int f = g_from[g_srt[lo + i]]; const int p = f >= 0 ? g_rnk[f] - lo : -1; cx.lds[i] = (unsigned)p < (unsigned)n_c ? p : -1; }
                }
            }
            wv::sync();
            int cur = rk - lo;
            while (cur >= 0 && A <= H) {
                wv::Lane<int> pl;
                WAVE_FOR(l) pl[l] = 0;
                int cnt = 0;
                while (cur >= 0 && cnt < 64 && A + cnt <= H) { WAVE_FOR(l) { if (l == cnt) pl[l] = cur; } ++cnt; cur = wv::uni(cx.lds[cur]); }
                WAVE_FOR(l) { if (l < cnt) g_anc[A + l] = g_srt[lo + pl[l]]; }
                A += cnt;
            }
            wv::sync();
            walked = true;
        }
    }
    if (!walked) for (int right = max_node; right >= 0 && A <= H; right = g_from[right]) anc[A++] = right;
    for (int i0 = 0; i0 < A; i0 += 64) { WAVE_FOR(l) { if (i0 + l < A) g_ancx[i0 + l] = g_seed[g_anc[i0 + l]]; } }
    wv::sync();
    HP_LSTAMP(16);
#if defined(HP_PROF) && !defined(HP_PROF_TRACK)
    if (HP_PROF_CHAIN_ON && r.prof) { r.prof[21] += 1; r.prof[12] += A; }
#endif
    // Most lines of a read against a repeat-rich genome are a few hits of neighbouring seed slots at some repeat copy: no gap at all.
    {
        int any_gap = anc_x[0] < seed_out - 1;
        for (int i0 = 0; i0 < A && !any_gap; i0 += 64) {
            wv::Lane<int> ex;
            WAVE_FOR(l) { const int i = i0 + l; ex[l] = i < A && (i + 1 < A ? g_ancx[i + 1] : -1) < g_ancx[i] - 1; }
            any_gap = wv::ballot(ex) != 0;
        }
        if (!any_gap) {
            HP_G int32_t *g_ln0 = (HP_G int32_t *)ln;
            for (int i0 = 0; i0 < A; i0 += 64) { WAVE_FOR(l) { if (i0 + l < A) g_ln0[i0 + l] = g_anc[i0 + l]; } }
            wv::sync();
            arena_release(cx.tmp, mark);
            HP_LSTAMP(17);
#if defined(HP_PROF) && !defined(HP_PROF_TRACK)
            if (HP_PROF_CHAIN_ON && r.prof) r.prof[22] += 1;
#endif
            return A;
        }
    }
    // ---- the gaps: [0] beyond the end node (when it is not of the last seed slot), then after anchor i when the next anchor (or
    // START) is more than one slot below.  Per gap: left, right, left_x, right_x, tail, the anchor it follows (-1: the first kind)
    const int GA = A + 2;
    int32_t *gp = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 16 * (size_t)GA);
    int32_t *pool = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 2 * (size_t)(H + 8));    // mini-line nodes of all gaps (ids, then edge classes)
    int32_t *posx = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 2 * (size_t)(H + A + 8));  // per position of ln: slot, "pair with the position before is checked"
    if (!gp || !pool || !posx) { arena_release(cx.tmp, mark); return -1; }
    int32_t *g_left = gp, *g_right = gp + GA, *g_lx = gp + 2 * GA, *g_rx = gp + 3 * GA, *g_tail = gp + 4 * GA, *g_after = gp + 5 * GA,
            *o_n = gp + 6 * GA, *o_off = gp + 7 * GA, *o_ds = gp + 8 * GA, *o_dn = gp + 9 * GA, *o_rf = gp + 10 * GA, *o_rs = gp + 11 * GA, *o_rn = gp + 12 * GA,
            *o_rnn = gp + 13 * GA, *o_rmf = gp + 14 * GA, *o_lane = gp + 15 * GA;
    int32_t *pool_mf = pool + (H + 8);
    int G = 0;
    if (anc_x[0] < seed_out - 1) { g_left[0] = max_node; g_right[0] = -1; g_lx[0] = anc_x[0]; g_rx[0] = seed_out; g_tail[0] = 0; g_after[0] = -1; G = 1; }
    for (int i0 = 0; i0 < A; i0 += 64) {
        wv::Lane<int> ex, lf, lx;
        WAVE_FOR(l) {
            const int i = i0 + l;
            int e = 0, left = -1, left_x = -1;
            if (i < A) { if (i + 1 < A) { left = g_anc[i + 1]; left_x = g_ancx[i + 1]; } e = left_x < g_ancx[i] - 1; }
            ex[l] = e; lf[l] = left; lx[l] = left_x;
        }
        const unsigned long long m = wv::ballot(ex);
        WAVE_FOR(l) {
            if (ex[l]) {
                const int g = G + __builtin_popcountll(m & ((1ull << l) - 1)), i = i0 + l;
                g_left[g] = lf[l]; g_right[g] = g_anc[i]; g_lx[g] = lx[l]; g_rx[g] = g_ancx[i]; g_tail[g] = 1; g_after[g] = i;
            }
        }
        G += __builtin_popcountll(m);
    }
    wv::sync();
    HP_LSTAMP(17);
#if defined(HP_PROF) && !defined(HP_PROF_TRACK)
    if (HP_PROF_CHAIN_ON && r.prof) { r.prof[23] += G; if (G < HP_GAP_MIN) r.prof[54] += G; r.prof[55] += 1; }
#endif
    // ---- the mini DPs.  By cluster when the read's hits have been clustered (gaps_by_cluster above); else every gap scans its own seed range:
    int pool_n = 0, d_score = 0, d_NM = 0;
    bool by_cluster = false;
    if (C) {
        const GapRest gr = gaps_by_cluster(r, *C, max_node, G, gp, GA, pool, pool_mf, _line, gc, cl_lo, cl_n);
        if (gr.pool_n == -1) { arena_release(cx.tmp, mark); return -1; }
        if (gr.pool_n >= 0) { by_cluster = true; pool_n = gr.pool_n; d_score = gr.d_score; d_NM = gr.d_NM; HP_STAT(0); }
        HP_LSTAMP(18);
    }
    if (!by_cluster) {
        HP_STAT(1);
        // ---- the mini DPs: one gap per lane (hp_gaps.h); what a lane cannot take goes through mini_line afterwards
        // (a lane walks the hit range of its gap by itself, one dependent load after the other: that pays when many gaps share the wait,
        // not for the two or three gaps of a short line)
        const bool use_lanes = HP_CL_CAP_RT(1) > 0 && G >= HP_GAP_MIN;
        for (int g0 = 0; g0 < G; g0 += 64) {
            wv::Lane<int> nn, ds, dn;
            WAVE_FOR(l) {
                const int g = g0 + l;
                GapOut O; O.n = -1; O.d_score = 0; O.d_NM = 0; O.r_from = -1; O.r_score = 0; O.r_NM = 0; O.r_nn = 1; O.r_mf = 0;
                if (g < G && g_left[g] >= 0 && use_lanes) gap_lane(r, K, cx.lds + l, g_left[g], g_right[g], g_lx[g], g_rx[g], g_tail[g], O);
                nn[l] = g < G ? O.n : 0; ds[l] = O.n >= 0 ? O.d_score : 0; dn[l] = O.n >= 0 ? O.d_NM : 0;
                if (g < G) {
                    o_n[g] = O.n; o_ds[g] = O.d_score; o_dn[g] = O.d_NM; o_rf[g] = O.r_from; o_rs[g] = O.r_score; o_rn[g] = O.r_NM; o_rnn[g] = O.r_nn; o_rmf[g] = O.r_mf; o_lane[g] = O.n >= 0;
                }
                // the nodes of the lane's mini line, staged in its strip until the pool offsets of the group are known
                for (int k = 0; k < HP_GAP_MCAP; ++k) { if (g < G && k < O.n) { cx.lds[(k * 6 + 0) * 64 + l] = O.ids[k]; cx.lds[(k * 6 + 1) * 64 + l] = O.mfs[k]; } }
            }
            wv::Lane<int> bad, cntl;
            WAVE_FOR(l) { bad[l] = nn[l] == -2; cntl[l] = nn[l] > 0 ? nn[l] : 0; }
            if (wv::ballot(bad) != 0) { cx.status |= ST_REFEXIT; arena_release(cx.tmp, mark); return -1; }    // "[frag mini dp] BUG" exit, :1140
            wv::Lane<int> pre = cntl;
            wv::scan_add_excl(pre);
            WAVE_FOR(l) {
                const int g = g0 + l;
                if (g < G && nn[l] >= 0) {
                    o_off[g] = pool_n + pre[l];
                    for (int k = 0; k < nn[l]; ++k) { pool[pool_n + pre[l] + k] = cx.lds[((k) * 6 + 0) * 64 + l]; pool_mf[pool_n + pre[l] + k] = cx.lds[((k) * 6 + 1) * 64 + l]; }
                }
            }
            pool_n += wv::reduce_sum(cntl);
            d_score += wv::reduce_sum(ds); d_NM += wv::reduce_sum(dn);
            wv::sync();
        }
        // right anchors and line nodes of the gaps the lanes have done
        for (int g0 = 0; g0 < G; g0 += 64) {
            WAVE_FOR(l) {
                const int g = g0 + l;
                if (g < G && o_lane[g]) {
                    if (g_tail[g]) {
                        const int right = g_right[g];
                        g_from[right] = o_rf[g]; gd[right].score = o_rs[g]; gd[right].NM = o_rn[g]; g_node_n[right] = o_rnn[g]; gd[right].match_flag = (uint8_t)o_rmf[g];
                    }
                    for (int k = 0; k < o_n[g]; ++k) gd[pool[o_off[g] + k]].match_flag = (uint8_t)pool_mf[o_off[g] + k];
                }
            }
        }
        wv::sync();
        HP_LSTAMP(18);
        {                                                                                   // the others, one at a time
            const GapRest gr = gaps_one_by_one(r, G, gp, GA, pool, pool_n, _line);
            if (gr.pool_n < 0) { arena_release(cx.tmp, mark); return -1; }
            pool_n = gr.pool_n; d_score += gr.d_score; d_NM += gr.d_NM;
        }
    }
    *line_score += d_score; *line_NM += d_NM;
    HP_LSTAMP(19);
    // ---- the line, end node first: [nodes beyond the end node], anchor 0, [nodes of the gap after it], anchor 1, ...
    int node_i = 0;
    if (pool_n == 0) {
        // No gap has put a node on the line (the usual case): the line is its anchors, and an inter-line trigger (:1384-1386, :1404-1414)
        // is a pair of consecutive anchors more than two slots apart -- the gap between them exists then, and START never takes part.
        HP_G int32_t *g_ln = (HP_G int32_t *)ln;
        for (int q0 = 0; q0 < A; q0 += 64) {
            wv::Lane<int> push, a0, a1;
            WAVE_FOR(l) {
                const int q = q0 + l;
                int p = 0, x0 = 0, x1 = 0;
                if (q < A) { x0 = g_anc[q]; g_ln[q] = x0; if (q > 0) { x1 = g_anc[q - 1]; p = g_ancx[q - 1] - g_ancx[q] > 2; } }
                push[l] = p; a0[l] = x0; a1[l] = x1;
            }
            const unsigned long long m = wv::ballot(push);
            const int cnt = __builtin_popcountll(m);
            if (!cnt) continue;
            if (T.used + cnt > T.cap) { cx.status |= ST_OVERFLOW; break; }
            WAVE_FOR(l) { if (push[l]) { const int at = T.used + __builtin_popcountll(m & ((1ull << l) - 1)); T.n1[at] = a0[l]; T.n2[at] = a1[l]; } }
            T.used += cnt; T.cnt[l_i] += cnt;
        }
        node_i = A;
        wv::sync();
    } else {
        HP_G int32_t *g_ln = (HP_G int32_t *)ln; HP_G int32_t *g_seg = (HP_G int32_t *)(posx + (H + A + 8));
        // Positions by prefix sums instead of one anchor after the other: an anchor sits at its index plus the nodes of all gaps before it.
        HP_G int32_t *g_na = g_ancx;                       // per anchor: nodes of the gap that follows it << 1 | 1, or 0 (the slots are not needed any more)
        HP_G int32_t *g_pa = (HP_G int32_t *)posx;         // per anchor: its position in the line
        const HP_G int32_t *gg_after = (const HP_G int32_t *)g_after, *gg_on = (const HP_G int32_t *)o_n, *gg_off = (const HP_G int32_t *)o_off;
        const HP_G int32_t *g_pool = (const HP_G int32_t *)pool;
        const int first = (G > 0 && g_after[0] == -1) ? 1 : 0;
        const int n_first = first ? o_n[0] : 0, off_first = first ? o_off[0] : 0;
        for (int i0 = 0; i0 < A; i0 += 64) { WAVE_FOR(l) { if (i0 + l < A) g_na[i0 + l] = 0; } }
        wv::sync();
        for (int g0 = first; g0 < G; g0 += 64) { WAVE_FOR(l) { const int g = g0 + l; if (g < G) g_na[gg_after[g]] = (gg_on[g] << 1) | 1; } }
        for (int k0 = 0; k0 < n_first; k0 += 64) { WAVE_FOR(l) { const int k = k0 + l; if (k < n_first) { g_ln[k] = g_pool[off_first + n_first - 1 - k]; g_seg[k] = k > 0; } } }
        wv::sync();
        int carry = n_first, carry_has = n_first > 0;      // anchor 0: pair (last node beyond the end, end node)
        for (int i0 = 0; i0 < A; i0 += 64) {
            wv::Lane<int> ex, has, pre;
            WAVE_FOR(l) { const int i = i0 + l; const int na = i < A ? g_na[i] : 0; ex[l] = na >> 1; has[l] = na & 1; }
            pre = ex;
            wv::scan_add_excl(pre);
            const int last_has = wv::bcast(has, 63);
            wv::shr1(has, carry_has);                      // pair (last node of the gap after the previous anchor or that anchor itself, this anchor)
            WAVE_FOR(l) {
                const int i = i0 + l;
                if (i < A) { const int p = i + carry + pre[l]; g_pa[i] = p; g_ln[p] = g_anc[i]; g_seg[p] = has[l]; }
            }
            carry += wv::reduce_sum(ex); carry_has = last_has;
        }
        wv::sync();
        for (int g0 = first; g0 < G; g0 += 64) {
            WAVE_FOR(l) {
                const int g = g0 + l;
                if (g < G) {
                    const int n = gg_on[g], off = gg_off[g], st = g_pa[gg_after[g]] + 1;
                    for (int k = 0; k < n; ++k) { g_ln[st + k] = g_pool[off + n - 1 - k]; g_seg[st + k] = 1; }
                }
            }
        }
        node_i = A + carry;
        wv::sync();
        // every node that came out of a mini DP is tracked now (:1376, :1398)
        for (int k0 = 0; k0 < pool_n; k0 += 64) { WAVE_FOR(l) { if (k0 + l < pool_n) gd[pool[k0 + l]].dp_flag = TRACKED_FLAG; } }
        // inter-line triggers: consecutive nodes of a gap's stretch more than two slots apart (:1384-1386, :1404-1414)
        HP_G int32_t *g_px = (HP_G int32_t *)posx;
        for (int q0 = 0; q0 < node_i; q0 += 64) { WAVE_FOR(l) { if (q0 + l < node_i) g_px[q0 + l] = g_seed[g_ln[q0 + l]]; } }
        wv::sync();
        for (int q0 = 1; q0 < node_i; q0 += 64) {
            wv::Lane<int> push;
            WAVE_FOR(l) { const int q = q0 + l; push[l] = q < node_i && g_seg[q] && g_px[q - 1] - g_px[q] > 2; }
            const unsigned long long m = wv::ballot(push);
            const int cnt = __builtin_popcountll(m);
            if (!cnt) continue;
            if (T.used + cnt > T.cap) { cx.status |= ST_OVERFLOW; break; }
            WAVE_FOR(l) { if (push[l]) { const int at = T.used + __builtin_popcountll(m & ((1ull << l) - 1)); T.n1[at] = g_ln[q0 + l]; T.n2[at] = g_ln[q0 + l - 1]; } }
            T.used += cnt; T.cnt[l_i] += cnt;
        }
        wv::sync();
    }
    arena_release(cx.tmp, mark);
    HP_LSTAMP(20);
#undef HP_LSTAMP
    return node_i;
}

}  // namespace hp
