// hp_cluster.h -- the main chaining pass of round 1 (frag_dp_update over all MIN hits, src/lamsa_dp_con.c:1345-1350) cluster by
// cluster, with the cluster's node state in LDS.  Included by hp_chain.h.
//
// get_fseed_dis (:596-634) connects two hits only if they lie on the same contig and strand within Rcl bases of each other,
// Rcl bounded by the read's seed span (see `cluster_reach`).  In the (contig, strand, position) order of hp_sort.h the
// hits of a read therefore fall into CLUSTERS -- maximal runs in which neighbours are at most Rcl apart -- and no edge of
// any class other than F_CHR_DIF / F_UNCONNECT ever joins two clusters.  frag_dp_update skips exactly those two classes
// (:722), so the DP over the read's hits is the union of independent DPs over its clusters, each visiting its own hits in
// the reference's order (seed slot ascending, hit index ascending).  A read against a repeat-rich genome has one cluster
// of a few hundred hits at its true locus and hundreds of tiny ones at repeat copies; the reference (and the HBM path of
// dp_update_range) scans every earlier hit of the big cluster for every one of its targets, O(n^2) loads of 32-byte
// records.  Here a cluster of up to lds_words / 5 hits is packed into this wave's LDS (20 bytes per hit, positions relative to
// the cluster's first hit, the dynamic DP fields included), the whole pass runs out of LDS -- every trip of the
// predecessor scan is five conflict-free ds_read_b32 -- and the cluster is written back once.  Clusters of one hit have no
// predecessor at all and are skipped; clusters that do not fit go through dp_update_range as before.
//
// The son lists (fnode_add_son, :683) that branch tracking walks are not maintained target by target any more: in this
// pass every target starts from START (fnode_set has just reset it) and changes its predecessor at most once, targets
// being visited in ascending hit order, so the list of a node is the set of hits whose final predecessor it is, in
// ascending order -- built for the whole read afterwards by one sort on (predecessor, hit) (build_sons).
#pragma once

namespace hp {

// hits of one cluster that fit this wave's LDS: five words per hit (cap = lds_words / 5); the tests' CPU build overrides this
// to send small clusters down the HBM path too
#ifndef HP_CL_CAP_RT
#define HP_CL_CAP_RT(cap) (cap)
#endif

struct Clusters {
    int n_cl;                    // number of clusters
    int32_t *cs;                 // [n_cl + 1] first rank of each cluster (cs[n_cl] = H)
    int32_t *cl_lo_r;            // [H] by rank: first rank of the rank's cluster
    int32_t *ce;                 // [H] by rank, valid at the first rank of a cluster: one past its last rank
    int32_t *csrt;               // [H] rank positions lo..hi-1 of a cluster hold its hits in ascending hit order
    uint8_t *big;                // [H] by hit: 1 = the hit's cluster did not fit LDS (dp_update_range handles its targets)
    long long reach;             // Rcl
};

// Upper bound, over every pair of seeds of the read, of the distance at which get_fseed_dis still connects two hits:
// |dis| < max(SV_len_thd, did*step, mat_dis+1), |act - exp| <= |dis| + |len_dif|, exp within did*step of the predecessor
// (the window of dp_update_range for the largest seed distance of the read); frag_min_extend's match-class test
// (|dis| <= match_dis * did) lies inside it.
HP_INL long long cluster_reach(const ReadCtx &r)
{
    const lamsa_hp_para *P = r.cx.P;
    const int did = r.seed_out > 0 ? r.seed_id[r.seed_out - 1] - r.seed_id[0] : 0;
    const int mdm = P->match_dis * ((P->aln_mode & 2) ? did : 1);
    long long R = P->SV_len_thd > did * P->seed_step ? P->SV_len_thd : (long long)did * P->seed_step;
    if (mdm + 1 > R) R = mdm + 1;
    R += 128 + (long long)did * P->seed_step;
    const long long R2 = (long long)did * (P->seed_step + P->match_dis) + 256;
    return R > R2 ? R : R2;
}

HP_NOINL bool clusters_build(ReadCtx &r, Clusters &C, HP_L uint64_t *lw, int lds_n)
{
    const int H = r.H;
    C.n_cl = 0; C.reach = cluster_reach(r);
    C.cs = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)(H + 2));
    C.cl_lo_r = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)(H + 1));
    C.csrt = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)(H + 1));
    C.ce = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)(H + 1));
    C.big = (uint8_t *)arena_alloc(r.cx, (size_t)H + 64);
    if (!C.cs || !C.cl_lo_r || !C.csrt || !C.ce || !C.big) return false;
    const HP_G NodeS *ns = (const HP_G NodeS *)r.nd;
    const HP_G int32_t *g_srt = (const HP_G int32_t *)r.srt, *g_rnk = (const HP_G int32_t *)r.rnk;
    HP_G int32_t *g_cs = (HP_G int32_t *)C.cs, *g_lo = (HP_G int32_t *)C.cl_lo_r;
    HP_G uint8_t *g_big = (HP_G uint8_t *)C.big;
    const long long R = C.reach;
    int n_cl = 0, run_start = 0;
    for (int i0 = 0; i0 < H; i0 += 64) {
        wv::Lane<int> b;
        WAVE_FOR(l) {
            const int i = i0 + l;
            int v = 0;
            if (i < H) {
                g_big[i] = 0;
                if (i == 0) v = 1;
                else {
                    const NodeS A = node_load(ns + g_srt[i]), B = node_load(ns + g_srt[i - 1]);
                    v = (A.chr != B.chr) | (A.strand != B.strand) | (A.pos - B.pos > R);
                }
            }
            b[l] = v;
        }
        const unsigned long long m = wv::ballot(b);
        WAVE_FOR(l) {
            const int i = i0 + l;
            if (i < H) {
                const unsigned long long below = m & ((2ull << l) - 1);          // starts at or below this lane
                g_lo[i] = below ? i0 + 63 - __builtin_clzll(below) : run_start;
                if (b[l]) g_cs[n_cl + __builtin_popcountll(m & ((1ull << l) - 1))] = i;
            }
        }
        n_cl += __builtin_popcountll(m);
        if (m) run_start = i0 + 63 - __builtin_clzll(m);
    }
    wv::sync();
    g_cs[n_cl] = H;
    C.n_cl = n_cl;
    wv::sync();
    { HP_G int32_t *g_ce = (HP_G int32_t *)C.ce; for (int c0 = 0; c0 < n_cl; c0 += 64) { WAVE_FOR(l) { const int c = c0 + l; if (c < n_cl) g_ce[g_cs[c]] = g_cs[c + 1]; } } }
    // the hits of every cluster in ascending hit order: one sort on (first rank of the cluster, hit index); the clusters keep
    // their rank ranges
    const size_t mark = arena_mark(r.cx.tmp);
    uint64_t *work = (uint64_t *)arena_alloc(r.cx, sizeof(uint64_t) * (size_t)(H + 1));
    if (!work) return false;
    const bool ok = sort_packed([&](int k) { return (uint64_t)(unsigned)g_lo[g_rnk[k]]; }, bits_of((unsigned)(H > 0 ? H : 1)), H,
                                (HP_G int32_t *)C.csrt, (HP_G int32_t *)nullptr, (HP_G uint64_t *)work, lw, lds_n);
    arena_release(r.cx.tmp, mark);
    return ok;
}


// ---------------------------------------------------------------- frag_min_extend for every MIN hit (:1031-1066, :1335-1343), cluster-wise
// For every MIN hit m and every seed with more than min_n hits, the first hit (ascending) of that seed that is match-class
// colinear with m joins the MIN pass.  Colinear hits share a cluster (cluster_reach), so m only has to be compared with
// the MULTI hits of its own cluster -- min_extend_all compares it with every hit of the read.  The hits are walked in
// the order of C.csrt (cluster by cluster, ascending hit index inside a cluster: the hits of one seed inside one cluster
// are consecutive and ascending), 64 MIN candidates per outer step against the 64-hit chunks that overlap their
// clusters; "first within its seed" is a segmented ballot over runs of equal (cluster, seed), carried across chunks.
// p_lo .. p_hi: the places in C.csrt to take MIN hits from (a cluster's rank range = its range of places): the whole read, or one
// cluster whose node state could not be packed into LDS (dp_cluster_lds does the same on the packed records otherwise).
HP_NOINL void min_extend_clusters(ReadCtx &r, const Clusters &C, int p_lo, int p_hi)
{
    const HP_G NodeS *ns = (const HP_G NodeS *)r.nd;
    HP_G NodeS *gd = (HP_G NodeS *)r.nd;
    const HP_G int32_t *g_csrt = (const HP_G int32_t *)C.csrt, *g_lo = (const HP_G int32_t *)C.cl_lo_r;
    const EdgeK K = edge_consts(r.cx.P);
    const int H = r.H;
    long long pairs_ = 0;                                      // accounting, flushed once at the end
    const size_t mark_ = arena_mark(r.cx.tmp);
    uint8_t *mk = (uint8_t *)arena_alloc(r.cx, (size_t)H + 64);
    if (!mk) return;
    HP_G uint8_t *gmk = (HP_G uint8_t *)mk;
    for (int b0 = p_lo; b0 < p_hi; b0 += 64) { WAVE_FOR(l) { if (b0 + l < p_hi) gmk[g_csrt[b0 + l]] = 0; } }
    wv::sync();
    for (int mbase = p_lo; mbase < p_hi; mbase += 64) {
        // the MIN hits of this chunk of the cluster order, one per lane; hits that are alone in their cluster have no partner
        wv::Lane<int> Ma0, Ma1, Ma2, Ma3, Mb0, ism, mcl;
        WAVE_FOR(l) {
            const int p = mbase + l, pp = p < H ? p : H - 1;
            const int id = g_csrt[pp];
            int a[4], b[4];
            hp_load16(ns + id, a); hp_load16((const HP_G char *)(ns + id) + 16, b);
            Ma0[l] = a[0]; Ma1[l] = a[1]; Ma2[l] = a[2]; Ma3[l] = a[3]; Mb0[l] = b[0];
            const int cl = g_lo[pp];
            const bool lone = cl == pp && (pp + 1 >= H || g_lo[pp + 1] == pp + 1);
            mcl[l] = cl;
            ism[l] = p < p_hi && !lone && (int)(int8_t)(b[1] & 0xff) == MIN_FLAG;
        }
        const unsigned long long mset = wv::ballot(ism);
        if (!mset) continue;
        const int cl_first = wv::bcast(mcl, __builtin_ctzll(mset)), cl_last = wv::bcast(mcl, 63 - __builtin_clzll(mset));
        unsigned long long carry = 0;                     // bit j: MIN hit j already found its hit in the run (carry_cl, carry_seed)
        int carry_seed = -1, carry_cl = -1;
        for (int base = cl_first; base < H; base += 64) {
            wv::Lane<int> sd, qcl, elig, hit, qsid, qld, idl;
            wv::Lane<long long> qdiag;          // position minus the seed's offset on the read: colinear hits share it
            WAVE_FOR(l) {
                const int p = base + l, pp = p < H ? p : H - 1;
                const int id = g_csrt[pp];
                int a[4], b[4];
                hp_load16(ns + id, a); hp_load16((const HP_G char *)(ns + id) + 16, b);
                sd[l] = p < H ? (a[3] >> 14) : -1; qcl[l] = p < H ? g_lo[pp] : 0x7fffffff; idl[l] = id;
                elig[l] = p < H && (int)(int8_t)(b[1] & 0xff) == MULTI_FLAG;
                hit[l] = 0;
                const int sid_ = (int)(int16_t)(b[0] & 0xffff), st_ = (int)(int8_t)((b[0] >> 16) & 0xff);
                qsid[l] = sid_; qld[l] = (int)(int8_t)((b[0] >> 24) & 0xff);
                qdiag[l] = (long long)(((unsigned long long)(unsigned)a[1] << 32) | (unsigned)a[0]) - (long long)(st_ * sid_ * K.seed_step);
            }
            if (wv::bcast(qcl, 0) > cl_last) break;                             // past the clusters of this chunk's MIN hits
            // runs of equal (cluster, seed): the lane where this lane's run starts inside the chunk
            wv::Lane<int> psd = sd, pcl = qcl, rs, seg;
            wv::shr1(psd, -2); wv::shr1(pcl, -2);
            WAVE_FOR(l) rs[l] = psd[l] != sd[l] || pcl[l] != qcl[l];
            const unsigned long long rsm = wv::ballot(rs) | 1ull;
            WAVE_FOR(l) seg[l] = 63 - __builtin_clzll(rsm & ((2ull << l) - 1));
            const int last = H - 1 - base < 63 ? H - 1 - base : 63;
            pairs_ += (long long)__builtin_popcountll(mset) * (last + 1);
            const int s_last = wv::bcast(sd, last), c_last = wv::bcast(qcl, last), st_last = wv::bcast(seg, last);
            unsigned long long next_carry = 0;
            for (unsigned long long mm = mset; mm; mm &= mm - 1) {
                const int j = __builtin_ctzll(mm);
                const int ma0 = wv::bcast(Ma0, j), ma1 = wv::bcast(Ma1, j), ma3 = wv::bcast(Ma3, j), mb0 = wv::bcast(Mb0, j), mc = wv::bcast(mcl, j);
                const int msid = (int)(int16_t)(mb0 & 0xffff), mst = (int)(int8_t)((mb0 >> 16) & 0xff), mld = (int)(int8_t)((mb0 >> 24) & 0xff);
                const int xm = ma3 >> 14;
                const long long mdiag = (long long)(((unsigned long long)(unsigned)ma1 << 32) | (unsigned)ma0) - (long long)(mst * msid * K.seed_step);
                wv::Lane<int> q;
                WAVE_FOR(l) {
                    int v = 0;
                    if (elig[l] && qcl[l] == mc && sd[l] != xm) {                 // same cluster: same contig and strand
                        const bool q_first = sd[l] < xm;                     // the hit of the earlier seed is `pre`
                        const int did = iabs(qsid[l] - msid);
                        const long long D = q_first ? mdiag - qdiag[l] : qdiag[l] - mdiag;     // act - exp
                        const int ld = mst > 0 ? (q_first ? qld[l] : mld) : (q_first ? mld : qld[l]);
                        const long long dis = (long long)mst * D - (long long)ld;
                        const int mat_dis = K.match_dis * (K.high_err ? did : 1);
                        v = did * K.seed_step >= K.seed_len && dis <= mat_dis && dis >= -mat_dis;
                    }
                    q[l] = v;
                }
                const unsigned long long qb = wv::ballot(q);
                const bool cj = (carry >> j) & 1, run_on = carry_seed == s_last && carry_cl == c_last;
                if (!qb) { if (run_on && cj) next_carry |= 1ull << j; continue; }
                WAVE_FOR(l) {
                    if (q[l]) {
                        const unsigned long long earlier = qb & ((1ull << l) - 1) & ~((1ull << seg[l]) - 1);
                        if (earlier == 0 && !(sd[l] == carry_seed && qcl[l] == carry_cl && cj)) hit[l] = 1;
                    }
                }
                if ((qb >> st_last) != 0 || (run_on && cj)) next_carry |= 1ull << j;
            }
            WAVE_FOR(l) { if (hit[l]) gmk[idl[l]] = 1; }
            carry = next_carry; carry_seed = s_last; carry_cl = c_last;
        }
    }
    wv::sync();
    r.n_pairs += pairs_;
    for (int b0 = p_lo; b0 < p_hi; b0 += 64) { WAVE_FOR(l) { if (b0 + l < p_hi) { const int k = g_csrt[b0 + l]; if (gmk[k]) gd[k].dp_flag = MIN_FLAG; } } }
    wv::sync();
    arena_release(r.cx.tmp, mark_);
}


// ---------------------------------------------------------------- small clusters, ONE CLUSTER PER LANE
// Most clusters of a read against a repeat-rich genome are tiny: two to six hits at some other copy of a repeat (on the 10-kbp
// ONT workload ~330 of the ~360 clusters with more than one hit).  Giving each of them the whole wave costs a gather, a sort-index
// look-up and a write-back round trip per cluster for a handful of comparisons.  Here 64 such clusters are handled at once, one
// per lane: the lane keeps the cluster's MIN hits (ascending hit order, from C.csrt) in its own strip of LDS -- six words each --
// and runs frag_dp_update (:701-764) over them exactly as dp_cluster_lds does.
#define HP_CLL_MCAP (HP_LANE_STRIP_WORDS / 384 < 6 ? HP_LANE_STRIP_WORDS / 384 : 6)
HP_INL int gap_edge(const EdgeK &K, int sp, int qpos, int qsid, int qld, int tpos, int tsid, int tld)
{   // get_fseed_dis (:596-634) for two hits of the same contig and strand, the first of an earlier seed (cf. dp_cluster_lds)
    const int dsid = tsid - qsid, span = dsid * K.seed_step;
    if (span < K.seed_len) return F_UNCONNECT;
    const int dis = sp * (tpos - qpos) - span - (sp > 0 ? qld : tld);
    const int mat_dis = K.match_dis * (K.high_err ? dsid : 1);
    if (dis <= mat_dis && dis >= -mat_dis) return dsid == 1 ? F_MATCH : (dsid <= K.mis3 ? F_MISMATCH : F_LONG_MISMATCH);
    if (dis > mat_dis && dis < K.sv_len) return F_DELETE;
    if ((dis < -mat_dis && dis >= 0 - (span - K.seed_len)) || (dis < -K.half_split && dis >= -K.sv_len)) return F_INSERT;
    return F_UNCONNECT;
}

// do_me: frag_min_extend (:1031-1066, :1335-1343) for the cluster first -- then every hit of the cluster is kept, and for every MIN hit and
// every repetitive seed the first hit (ascending) of that seed that is match-class colinear with it joins the MIN pass.
HP_INL void cluster_lane(ReadCtx &r, const EdgeK &K, const Clusters &C, HP_L int32_t *strip, int lo, int n, bool do_me)
{
    const HP_G NodeS *ns = (const HP_G NodeS *)r.nd;
    HP_G NodeS *gd = (HP_G NodeS *)r.nd;
    const HP_G int32_t *g_csrt = (const HP_G int32_t *)C.csrt;
    HP_G int32_t *g_from = (HP_G int32_t *)r.n_from, *g_node_n = (HP_G int32_t *)r.n_node_n;
#define CW(e, w) strip[((e) * 6 + (w)) * 64]
    int m = 0, sp = 0; int64_t pos0 = 0;
    unsigned is_min = 0, promoted = 0;                                                  // bit e: entry e is a MIN hit / has just become one
    for (int i0 = 0; i0 < n; i0 += 3) {                                                 // the cluster's hits (do_me) or its MIN hits, ascending hit order
      // three at a time: their ids, then their records, requested together (one after the other every hit is two dependent trips)
      int ids[3]; NodeS Qs[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) ids[u] = g_csrt[lo + (i0 + u < n ? i0 + u : n - 1)];
#pragma unroll
      for (int u = 0; u < 3; ++u) Qs[u] = node_load(ns + ids[u]);
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (i0 + u >= n) continue;
        const int id = ids[u];
        const NodeS Q = Qs[u];
        if (Q.dp_flag != MIN_FLAG && !(do_me && Q.dp_flag == MULTI_FLAG)) continue;
        if (m == 0) { pos0 = Q.pos; sp = Q.strand; }
        if (Q.dp_flag == MIN_FLAG) is_min |= 1u << m;
        CW(m, 0) = (int)(Q.pos - pos0); CW(m, 1) = Q.slot_j; CW(m, 2) = ((int)Q.sid & 0xffff) | (((int)Q.len_dif8 & 0xff) << 16);
        CW(m, 3) = (int)(((unsigned)Q.score << 16) | (unsigned)(Q.NM & 0xffff));
        CW(m, 4) = id; CW(m, 5) = (0xff << 24) | (1 << 16) | ((int)Q.son_flag << 8) | Q.match_flag;     // from (0xff = START) | node_n | son_flag | match_flag
        ++m;
      }
    }
    if (do_me && is_min != 0 && is_min != (1u << m) - 1) {
        for (int a = 0; a < m; ++a) {
            if (!((is_min >> a) & 1)) continue;
            const int apos = CW(a, 0), a2 = CW(a, 2), ax = CW(a, 1) >> 14;
            const int asid = (int)(short)(a2 & 0xffff), ald = (int)(int8_t)((a2 >> 16) & 0xff);
            int run_seed = -1; bool run_done = false;
            for (int q = 0; q < m; ++q) {
                if ((is_min >> q) & 1) continue;
                const int qx = CW(q, 1) >> 14;
                if (qx != run_seed) { run_seed = qx; run_done = false; }                 // the hits of one seed are consecutive
                if (run_done || qx == ax) continue;
                const int q2 = CW(q, 2), qpos = CW(q, 0);
                const int qsid = (int)(short)(q2 & 0xffff), qld = (int)(int8_t)((q2 >> 16) & 0xff);
                const int flag = qx < ax ? gap_edge(K, sp, qpos, qsid, qld, apos, asid, ald) : gap_edge(K, sp, apos, asid, ald, qpos, qsid, qld);     // the hit of the earlier seed is `pre`
                if (flag <= F_LONG_MISMATCH) { promoted |= 1u << q; run_done = true; }
            }
        }
    }
    const unsigned in_pass = is_min | promoted;                                          // the entries the MIN pass works on
    bool any = false;
    for (int a = 1; a < m; ++a) {
        if (!((in_pass >> a) & 1)) continue;
        const int tpos = CW(a, 0), tsj = CW(a, 1), t2 = CW(a, 2), t3 = CW(a, 3);
        const int tslot = tsj >> 14, tsid = (int)(short)(t2 & 0xffff), tld = (int)(int8_t)((t2 >> 16) & 0xff);
        const int t_score = t3 >> 16, t_NM = t3 & 0xffff;
        int best_hi = -0x7fffffff, best_lo = -1, best_b = -1, best_f = 0, neg_p = -0x7fffffff, neg_b = -1, neg_f = 0, neg_hi = 0;
        for (int b = 0; b < a; ++b) {
            if (!((in_pass >> b) & 1)) continue;
            const int qsj = CW(b, 1), qslot = qsj >> 14;
            if (qslot >= tslot) continue;
            const int q5 = CW(b, 5);
            if (sp == 1 && ((q5 >> 8) & 0xff) <= F_MATCH_THD) continue;                 // '+': the candidate already has a match son, :718-720
            const int q2 = CW(b, 2), q3 = CW(b, 3);
            const int flag = gap_edge(K, sp, CW(b, 0), (int)(short)(q2 & 0xffff), (int)(int8_t)((q2 >> 16) & 0xff), tpos, tsid, tld);
            if (flag == F_UNCONNECT) continue;
            const int pos = ((tslot - 1 - qslot) << 14) | (qsj & 16383);                // scan order: seeds descending, hits ascending
            const int cand = (q3 >> 16) + 1 + score_table(flag), nm = (q3 & 0xffff) + t_NM;
            const int hi = (int)(((unsigned)cand << 16) | (unsigned)(65535 - nm)), lo_ = (1 << 28) - 1 - pos;
            if (hi > best_hi || (hi == best_hi && lo_ > best_lo)) { best_hi = hi; best_lo = lo_; best_b = b; best_f = flag; }
            if (sp == -1 && flag <= F_MATCH_THD && 0 - pos > neg_p) { neg_p = 0 - pos; neg_b = b; neg_f = flag; neg_hi = hi; }   // '-': first match precursor, :726-733
        }
        int w_b = -1, w_f = 0, w_score = t_score, w_nm = t_NM;
        if (neg_b >= 0) { w_b = neg_b; w_f = neg_f; w_score = neg_hi >> 16; w_nm = 65535 - (neg_hi & 0xffff); }
        else if (best_b >= 0) {
            const int cand = best_hi >> 16, nm = 65535 - (best_hi & 0xffff);
            if (cand > w_score || (cand == w_score && nm < w_nm)) { w_b = best_b; w_f = best_f; w_score = cand; w_nm = nm; }
        }
        if (w_b >= 0) {                                                                  // :753-761
            const int q5 = CW(w_b, 5);
            CW(a, 3) = (int)(((unsigned)w_score << 16) | (unsigned)(w_nm & 0xffff));
            CW(a, 5) = (w_b << 24) | ((((q5 >> 16) & 0xff) + 1) << 16) | (CW(a, 5) & 0xff00) | w_f;
            CW(w_b, 5) = (q5 & ~0xff00) | (w_f << 8);
            any = true;
        }
    }
    if (any || promoted) {
        for (int a = 0; a < m; ++a) {
            if (!((in_pass >> a) & 1) || !(any || ((promoted >> a) & 1))) continue;
            const int id = CW(a, 4), w2 = CW(a, 2), w3 = CW(a, 3), w5 = CW(a, 5);
            const int b0 = (w2 & 0xffff) | ((sp & 0xff) << 16) | (((w2 >> 16) & 0xff) << 24);
            hp_store16((HP_G char *)(gd + id) + 16, b0, MIN_FLAG | (((w5 >> 8) & 0xff) << 8) | ((w5 & 0xff) << 16), w3 >> 16, w3 & 0xffff);
            const int f = (w5 >> 24) & 0xff;
            g_from[id] = f == 0xff ? -1 : CW(f, 4);
            g_node_n[id] = (w5 >> 16) & 0xff;
        }
    }
#undef CW
}

// ---------------------------------------------------------------- the cluster in LDS
// five arrays of `cap` words: W0 position relative to the cluster's first hit | W1 slot:14 j:14 dp_flag:4 |
// W2 sid:15 len_dif:8 son_flag:5 match_flag:4 | W3 score:16 NM:16 | W4 (predecessor's index in the cluster + 1):16 node_n:16
struct ClLds { HP_L int32_t *w0, *w1, *w2, *w3, *w4; };
HP_INL ClLds cl_lds(HP_L int32_t *lds, int cap) { ClLds c; c.w0 = lds; c.w1 = lds + cap; c.w2 = lds + 2 * cap; c.w3 = lds + 3 * cap; c.w4 = lds + 4 * cap; return c; }

HP_INL NodeS cl_unpack(int w0, int w1, int w2, int w3, int chr, int strand)
{
    NodeS q;
    q.pos = (int64_t)(unsigned)w0; q.chr = chr; q.slot_j = (int)((unsigned)w1 >> 4);
    q.sid = (int16_t)((unsigned)w2 >> 17); q.strand = (int8_t)strand; q.len_dif8 = (int8_t)((w2 >> 9) & 0xff);
    q.dp_flag = (int8_t)((int)((unsigned)w1 << 28) >> 28); q.son_flag = (uint8_t)((w2 >> 4) & 31); q.match_flag = (uint8_t)(w2 & 15); q.pad_ = 0;
    q.score = w3 >> 16; q.NM = w3 & 0xffff;
    return q;
}

// One cluster [lo, lo + n) (ranks), n <= lds_words / 5: frag_dp_update (:701-764) for its MIN hits, out of LDS.
// Returns false when the cluster cannot be packed (span or NM beyond the field widths): the caller marks it big.
// do_me: frag_min_extend (:1031-1066, :1335-1343) for the cluster first, on the packed records: the cluster's repetitive (MULTI) hits are
// listed in ascending hit order -- a few dozen at a read's true locus, where most hits are the only ones of their seeds -- and every MIN
// hit is compared with that list only; "first hit of its seed" is a segmented ballot over the list, carried across its 64-hit chunks.
HP_NOINL bool dp_cluster_lds(ReadCtx &r, const Clusters &C, int lo, int n, bool do_me)
{
    const HP_G NodeS *ns = (const HP_G NodeS *)r.nd;
    HP_G NodeS *gd = (HP_G NodeS *)r.nd;
    const HP_G int32_t *g_srt = (const HP_G int32_t *)r.srt, *g_rnk = (const HP_G int32_t *)r.rnk, *g_csrt = (const HP_G int32_t *)C.csrt;
    const HP_G int16_t *g_hnm = (const HP_G int16_t *)r.h_nm;
    HP_G int32_t *g_from = (HP_G int32_t *)r.n_from, *g_node_n = (HP_G int32_t *)r.n_node_n;
    const ClLds L = cl_lds(r.cx.lds, r.cx.lds_words / 5);
    const EdgeK K = edge_consts(r.cx.P);
    const int dp_flag = MIN_FLAG;
    if ((long long)(r.seed_id[r.seed_out - 1] - r.seed_id[0] + 1) * K.seed_step > 0x3fffffffll) return false;     // 32-bit geometry below
    // ---- load + pack; lane b of `bmin` keeps the smallest seed slot of block b (64 consecutive ranks)
    const NodeS first = node_load(ns + g_srt[lo]);
    const int64_t pos0 = first.pos; const int strand = first.strand;
    int nm_sum = 0; int bad = 0;
    wv::Lane<int> bmin;
    WAVE_FOR(l) bmin[l] = 0x7fffffff;
    wv::sync();                                                    // whatever used this LDS before is done
    for (int i0 = 0; i0 < n; i0 += 64) {
        wv::Lane<int> nml, badl, nsl;
        WAVE_FOR(l) {
            const int i = i0 + l;
            int nmv = 0, bv = 0, sv = -0x7fffffff;
            if (i < n) {
                const int id = g_srt[lo + i];
                const NodeS q = node_load(ns + id);
                const int64_t rel = q.pos - pos0;
                bv = rel < 0 || rel > 0x7ffffff0ll || q.score < -30000 || q.score > 30000 || q.NM < 0 || q.NM > 65535 || q.sid < 0;
                nmv = g_hnm[id]; sv = 0 - (q.slot_j >> 14);
                L.w0[i] = (int)(unsigned)rel;
                L.w1[i] = (int)(((unsigned)q.slot_j << 4) | ((unsigned)q.dp_flag & 15u));
                L.w2[i] = (int)(((unsigned)(unsigned short)q.sid << 17) | (((unsigned)q.len_dif8 & 255u) << 9) | (((unsigned)q.son_flag & 31u) << 4) | ((unsigned)q.match_flag & 15u));
                L.w3[i] = (int)(((unsigned)q.score << 16) | ((unsigned)q.NM & 0xffffu));
                L.w4[i] = 1;                                       // from = START, node_n = 1 (fnode_set has just run for every hit)
            }
            nml[l] = nmv; badl[l] = bv; nsl[l] = sv;
        }
        nm_sum += wv::reduce_sum(nml);
        if (wv::ballot(badl) != 0) bad = 1;
        const int mn = 0 - wv::reduce_max(nsl);
        WAVE_FOR(l) { if (l == (i0 >> 6)) bmin[l] = mn; }
    }
    if (bad || nm_sum > 65535 || n > 64 * 64) return false;       // a chain's NM is at most the sum over the cluster: 16 bits suffice
    wv::sync();
    const int sp = strand, POSMAX = (1 << 28) - 1, NEG = -0x7fffffff;
    long long pairs_ = 0;                                      // accounting, flushed once (a counter in r is a memory round trip per use)
    if (do_me) {
        const size_t mark = arena_mark(r.cx.tmp);
        int32_t *ml = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * 2 * (size_t)(n + 64));      // places of the MULTI hits, then one mark each
        if (!ml) return false;
        HP_G int32_t *g_ml = (HP_G int32_t *)ml, *g_mk = (HP_G int32_t *)(ml + n + 64);
        int nml = 0, n_min = 0;
        for (int o0 = 0; o0 < n; o0 += 64) {
            wv::Lane<int> ism, isu, tll;
            WAVE_FOR(l) {
                const int o = o0 + l; const int tl = o < n ? g_rnk[g_csrt[lo + o]] - lo : 0;
                const int dpf = (int)((unsigned)L.w1[tl] << 28) >> 28;
                tll[l] = tl; isu[l] = o < n && dpf == MULTI_FLAG; ism[l] = o < n && dpf == MIN_FLAG;
            }
            const unsigned long long mu = wv::ballot(isu);
            WAVE_FOR(l) { if (isu[l]) { const int at = nml + __builtin_popcountll(mu & ((1ull << l) - 1)); g_ml[at] = tll[l]; g_mk[at] = 0; } }
            nml += __builtin_popcountll(mu); n_min += __builtin_popcountll(wv::ballot(ism));
        }
        wv::sync();
        if (nml > 0 && n_min > 0) {
            for (int o0 = 0; o0 < n; o0 += 64) {
                // the MIN hits of this chunk of the ascending hit order
                wv::Lane<int> ism, T0, T1, T2;
                WAVE_FOR(l) {
                    const int o = o0 + l; const int tl = o < n ? g_rnk[g_csrt[lo + o]] - lo : 0;
                    T0[l] = L.w0[tl]; T1[l] = L.w1[tl]; T2[l] = L.w2[tl];
                    ism[l] = o < n && ((int)((unsigned)T1[l] << 28) >> 28) == MIN_FLAG;
                }
                const unsigned long long mset = wv::ballot(ism);
                if (!mset) continue;
                unsigned long long carry = 0;                     // bit j: MIN hit j already found its hit in the run of seed carry_seed
                int carry_seed = -1;
                for (int b0 = 0; b0 < nml; b0 += 64) {
                    wv::Lane<int> sd, q0, qsid, qld, hit;
                    WAVE_FOR(l) {
                        const int at = b0 + l;
                        int s_ = -1, p_ = 0, i_ = 0, d_ = 0;
                        if (at < nml) { const int tl = g_ml[at]; const int w1 = L.w1[tl], w2 = L.w2[tl]; s_ = (int)((unsigned)w1 >> 18); p_ = L.w0[tl]; i_ = (int)((unsigned)w2 >> 17); d_ = (int)(int8_t)((w2 >> 9) & 0xff); }
                        sd[l] = s_; q0[l] = p_; qsid[l] = i_; qld[l] = d_; hit[l] = 0;
                    }
                    wv::Lane<int> psd = sd, rs, seg;
                    wv::shr1(psd, -2);
                    WAVE_FOR(l) rs[l] = psd[l] != sd[l];
                    const unsigned long long rsm = wv::ballot(rs) | 1ull;
                    WAVE_FOR(l) seg[l] = 63 - __builtin_clzll(rsm & ((2ull << l) - 1));
                    const int last = nml - 1 - b0 < 63 ? nml - 1 - b0 : 63;
                    pairs_ += (long long)__builtin_popcountll(mset) * (last + 1);
                    const int s_last = wv::bcast(sd, last), st_last = wv::bcast(seg, last);
                    unsigned long long next_carry = 0;
                    for (unsigned long long mm = mset; mm; mm &= mm - 1) {
                        const int j = __builtin_ctzll(mm);
                        const int m0 = wv::bcast(T0, j), m1 = wv::bcast(T1, j), m2 = wv::bcast(T2, j);
                        const int xm = (int)((unsigned)m1 >> 18), msid = (int)((unsigned)m2 >> 17), mld = (int)(int8_t)((m2 >> 9) & 0xff);
                        wv::Lane<int> q;
                        WAVE_FOR(l) {
                            int v = 0;
                            if (sd[l] >= 0 && sd[l] != xm) {
                                const bool q_first = sd[l] < xm;                       // the hit of the earlier seed is `pre` (get_fseed_dis :607-619)
                                const int dsid = q_first ? msid - qsid[l] : qsid[l] - msid, span = dsid * K.seed_step;
                                const int dis = (q_first ? sp * (m0 - q0[l]) : sp * (q0[l] - m0)) - span - (sp > 0 ? (q_first ? qld[l] : mld) : (q_first ? mld : qld[l]));
                                const int mat_dis = K.match_dis * (K.high_err ? dsid : 1);
                                v = span >= K.seed_len && dis <= mat_dis && dis >= -mat_dis;
                            }
                            q[l] = v;
                        }
                        const unsigned long long qb = wv::ballot(q);
                        const bool cj = (carry >> j) & 1, run_on = carry_seed == s_last;
                        if (!qb) { if (run_on && cj) next_carry |= 1ull << j; continue; }
                        WAVE_FOR(l) {
                            if (q[l]) {
                                const unsigned long long earlier = qb & ((1ull << l) - 1) & ~((1ull << seg[l]) - 1);
                                if (earlier == 0 && !(sd[l] == carry_seed && cj)) hit[l] = 1;
                            }
                        }
                        if ((qb >> st_last) != 0 || (run_on && cj)) next_carry |= 1ull << j;
                    }
                    WAVE_FOR(l) { if (hit[l]) g_mk[b0 + l] = 1; }
                    carry = next_carry; carry_seed = s_last;
                }
            }
            wv::sync();
            for (int b0 = 0; b0 < nml; b0 += 64) {
                WAVE_FOR(l) { const int at = b0 + l; if (at < nml && g_mk[at]) { const int tl = g_ml[at]; L.w1[tl] = (L.w1[tl] & ~15) | MIN_FLAG; } }
            }
            wv::sync();
        }
        arena_release(r.cx.tmp, mark);
    }
    // ---- targets in ascending hit order (C.csrt), 64 at a time: their places in the cluster and their records.  A target's
    // record is still what was loaded when its turn comes: only later targets (higher hit index) can choose it as their
    // predecessor and touch its son_flag.
    for (int o0 = 0; o0 < n; o0 += 64) {
        wv::Lane<int> tloc, T0, T1, T2, T3;
        WAVE_FOR(l) {
            const int o = o0 + l; const int tl = o < n ? g_rnk[g_csrt[lo + o]] - lo : 0;
            tloc[l] = tl; T0[l] = L.w0[tl]; T1[l] = L.w1[tl]; T2[l] = L.w2[tl]; T3[l] = L.w3[tl];
        }
        const int cnt = n - o0 < 64 ? n - o0 : 64;
        for (int q = 0; q < cnt; ++q) {
            const int t1 = wv::bcast(T1, q);
            if ((int)((unsigned)t1 << 28) >> 28 != dp_flag) continue;
            const int x = (int)((unsigned)t1 >> 18);
            if (x == 0) continue;                                  // a hit of the first seed slot has no predecessor
            const int ct = wv::bcast(tloc, q), t0 = wv::bcast(T0, q), t2 = wv::bcast(T2, q), t3 = wv::bcast(T3, q);
            const int tsid = (int)((unsigned)t2 >> 17), tld = (int)(int8_t)((t2 >> 9) & 0xff), t_score = t3 >> 16, t_NM = t3 & 0xffff;
            wv::Lane<int> bhi, blo, bi, bf, negp, n_i, n_f, n_hi, okl;
            WAVE_FOR(l) { bhi[l] = NEG; blo[l] = -1; bi[l] = 0; bf[l] = 0; negp[l] = NEG; n_i[l] = 0; n_f[l] = 0; n_hi[l] = 0; okl[l] = 0; }
            for (int i0 = 0; i0 < n; i0 += 64) {
                if (wv::bcast(bmin, i0 >> 6) >= x) continue;       // no hit of an earlier seed in this block
                pairs_ += n - i0 < 64 ? n - i0 : 64;
                WAVE_FOR(l) {
                    const int i = i0 + l, ii = i < n ? i : 0;
                    const int q0 = L.w0[ii], q1 = L.w1[ii], q2 = L.w2[ii], q3 = L.w3[ii];
                    const int qslot = (int)((unsigned)q1 >> 18), qdpf = (int)((unsigned)q1 << 28) >> 28;
                    const int dsid = tsid - (int)((unsigned)q2 >> 17);
                    const int span = dsid * K.seed_step;
                    const int qld = (int)(int8_t)((q2 >> 9) & 0xff);
                    const int dis = sp * (t0 - q0) - span - (sp > 0 ? qld : tld);               // get_fseed_dis :607-619 for a predecessor of an earlier seed
                    const int mat_dis = K.match_dis * (K.high_err ? dsid : 1);
                    int flag = F_UNCONNECT;
                    if (span >= K.seed_len) {
                        if (dis <= mat_dis && dis >= -mat_dis) flag = dsid == 1 ? F_MATCH : (dsid <= K.mis3 ? F_MISMATCH : F_LONG_MISMATCH);
                        else if (dis > mat_dis && dis < K.sv_len) flag = F_DELETE;
                        else if ((dis < -mat_dis && dis >= 0 - (span - K.seed_len)) || (dis < -K.half_split && dis >= -K.sv_len)) flag = F_INSERT;
                    }
                    const int ok = (i < n) & (qslot < x) & (qdpf == dp_flag) & !((sp == 1) & (((q2 >> 4) & 31) <= F_MATCH_THD)) & (flag != F_UNCONNECT);
                    const int pos = ((x - 1 - qslot) << 14) | ((q1 >> 4) & 16383);                  // scan order: seeds descending, hits ascending
                    const int cand = (q3 >> 16) + 1 + score_table(flag);
                    const int hi = ok ? (int)(((unsigned)cand << 16) | (unsigned)(65535 - ((q3 & 0xffff) + t_NM))) : NEG;   // score desc, then NM asc
                    const int lo_ = POSMAX - pos;
                    const bool better = hi > bhi[l] || (hi == bhi[l] && ok && lo_ > blo[l]);
                    bhi[l] = better ? hi : bhi[l]; blo[l] = better ? lo_ : blo[l]; bi[l] = better ? i : bi[l]; bf[l] = better ? flag : bf[l];
                    const int np = (ok & (sp == -1) & (flag <= F_MATCH_THD)) ? -pos : NEG;          // '-': first match precursor wins, :726-733
                    const bool nb = np > negp[l];
                    negp[l] = nb ? np : negp[l]; n_i[l] = nb ? i : n_i[l]; n_f[l] = nb ? flag : n_f[l]; n_hi[l] = nb ? hi : n_hi[l];
                    okl[l] |= ok;
                }
            }
            if (wv::ballot(okl) == 0) continue;                    // no connectable predecessor: the node keeps its state
            int max_from = -1, max_score = t_score, max_NM = t_NM, max_flag = 0;
            bool changed = false;
            const int npos = wv::reduce_max(negp);
            if (npos != NEG) {
                wv::Lane<int> w;
                WAVE_FOR(l) w[l] = negp[l] == npos;
                const int wl = __builtin_ctzll(wv::ballot(w));
                const int hi = wv::bcast(n_hi, wl);
                max_from = wv::bcast(n_i, wl); max_flag = wv::bcast(n_f, wl); max_score = hi >> 16; max_NM = 65535 - (hi & 0xffff);
                changed = true;
            } else {
                const int mh = wv::reduce_max(bhi);
                if (mh != NEG) {
                    const int cand = mh >> 16, nm = 65535 - (mh & 0xffff);
                    if (cand > max_score || (cand == max_score && nm < max_NM)) {
                        wv::Lane<int> l2;
                        WAVE_FOR(l) l2[l] = bhi[l] == mh ? blo[l] : -1;
                        const int ml = wv::reduce_max(l2);
                        wv::Lane<int> w;
                        WAVE_FOR(l) w[l] = bhi[l] == mh && blo[l] == ml;
                        const int wl = __builtin_ctzll(wv::ballot(w));
                        max_from = wv::bcast(bi, wl); max_flag = wv::bcast(bf, wl); max_score = cand; max_NM = nm;
                        changed = true;
                    }
                }
            }
            if (changed) {                                         // :753-761; the LDS copy is the node state until the write-back
                const int f2 = wv::uni(L.w2[max_from]), f4 = wv::uni(L.w4[max_from]);
                L.w2[max_from] = (f2 & ~(31 << 4)) | ((max_flag & 31) << 4);
                L.w2[ct] = (t2 & ~15) | (max_flag & 15);
                L.w3[ct] = (int)(((unsigned)max_score << 16) | ((unsigned)max_NM & 0xffffu));
                L.w4[ct] = ((max_from + 1) << 16) | (((f4 & 0xffff) + 1) & 0xffff);
                wv::sync();
            }
        }
    }
    r.n_pairs += pairs_;
    // ---- write back: the dynamic half of the record, predecessor and node count
    wv::sync();
    for (int i0 = 0; i0 < n; i0 += 64) {
        WAVE_FOR(l) {
            const int i = i0 + l;
            if (i < n) {
                const int id = g_srt[lo + i];
                const int w1 = L.w1[i], w2 = L.w2[i], w3 = L.w3[i], w4 = L.w4[i];
                const int b0 = (int)(((unsigned)w2 >> 17) & 0xffffu) | ((strand & 0xff) << 16) | (((w2 >> 9) & 0xff) << 24);
                const int dpf = (int)((unsigned)w1 << 28) >> 28;
                hp_store16((HP_G char *)(gd + id) + 16, b0, (dpf & 0xff) | (((w2 >> 4) & 31) << 8) | ((w2 & 15) << 16), w3 >> 16, w3 & 0xffff);
                const int fl = (int)((unsigned)w4 >> 16);
                g_from[id] = fl ? g_srt[lo + fl - 1] : -1;
                g_node_n[id] = w4 & 0xffff;
            }
        }
    }
    wv::sync();
    return true;
}

// fnode_add_son (:683) for every hit whose predecessor the pass has set, all at once: hits ordered by (predecessor, hit
// index); a run of equal predecessors is that node's son list in insertion order (see the header comment).
HP_NOINL bool build_sons(ReadCtx &r, HP_L uint64_t *lw, int lds_n)
{
    const int H = r.H;
    if (H == 0) return true;
    const size_t mark = arena_mark(r.cx.tmp);
    int32_t *ps = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)(H + 1));
    uint64_t *work = (uint64_t *)arena_alloc(r.cx, sizeof(uint64_t) * (size_t)(H + 1));
    if (!ps || !work) { arena_release(r.cx.tmp, mark); return false; }
    const HP_G int32_t *g_from = (const HP_G int32_t *)r.n_from;
    HP_G int32_t *g_ps = (HP_G int32_t *)ps;
    HP_G int32_t *g_in_de = (HP_G int32_t *)r.n_in_de, *g_son_n = (HP_G int32_t *)r.n_son_n;
    HP_G int32_t *g_first = (HP_G int32_t *)r.n_first, *g_last = (HP_G int32_t *)r.n_last, *g_next = (HP_G int32_t *)r.n_next;
    const bool ok = sort_packed([&](int k) { const int f = g_from[k]; return (uint64_t)(unsigned)(f >= 0 ? f : H); }, bits_of((unsigned)H) , H,
                                g_ps, (HP_G int32_t *)nullptr, (HP_G uint64_t *)work, lw, lds_n);
    if (!ok) { arena_release(r.cx.tmp, mark); return false; }
    int run_start = 0;                                             // position where the run reaching into this chunk began
    for (int i0 = 0; i0 < H; i0 += 64) {
        wv::Lane<int> st, tl, fl, fnl;
        WAVE_FOR(l) {
            const int i = i0 + l;
            int t = -1, f = -1, fp = -2, fn = -2;
            if (i < H) {
                t = g_ps[i]; f = g_from[t];
                fp = i > 0 ? g_from[g_ps[i - 1]] : -2;
                fn = i + 1 < H ? g_from[g_ps[i + 1]] : -2;
            }
            tl[l] = t; fl[l] = f; fnl[l] = fn;
            st[l] = i < H && f >= 0 && fp != f;
        }
        const unsigned long long m = wv::ballot(st);
        WAVE_FOR(l) {
            const int i = i0 + l, t = tl[l], f = fl[l];
            if (i < H && f >= 0) {
                const unsigned long long below = m & ((2ull << l) - 1);
                const int rs = below ? i0 + 63 - __builtin_clzll(below) : run_start;
                if (st[l]) g_first[f] = t;
                if (fnl[l] == f) g_next[t] = g_ps[i + 1];
                else { g_next[t] = -1; g_last[f] = t; g_son_n[f] = i - rs + 1; g_in_de[f] = i - rs + 1; }
            }
        }
        if (m) run_start = i0 + 63 - __builtin_clzll(m);
    }
    wv::sync();
    arena_release(r.cx.tmp, mark);
    return true;
}

}  // namespace hp
