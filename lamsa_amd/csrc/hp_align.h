// hp_align.h -- one read on one wavefront: stages (2) chaining, (3) gap-fill / extension, and the
// second round (2')/(3') on the uncovered read regions -- the body of the reference's per-read
// worker lamsa_main_aln (src/lamsa_aln.c:857-871), minus stage (4)/(5)/(6) which stay on the host.
#pragma once
#include "hp_fill.h"
#include "hp_sort.h"

namespace hp {

// ---- covered intervals of the first round and the regions they leave uncovered ----
// push_reg_res (lamsa_aln.c:571), aln_sort_reg (:477), aln_merg_reg (:499), get_remain_reg (:550)
HP_FN void regs_push(ReadCtx &r, Regs &G, int cap, const Rec &rec)
{
    if (G.n >= cap) { r.cx.status |= ST_OVERFLOW; return; }
    const cig_t *c = rec.cig.c; const int n = rec.cig.n, k = G.n;
    const int reflen = cig_reflen(c, n);
    G.rb[k].chr = G.re[k].chr = rec.chr; G.rb[k].is_rev = G.re[k].is_rev = 1 - rec.nstrand;
    if (rec.nstrand == 1) {
        G.beg[k] = (c[0] & 0xf) == C_S ? (c[0] >> 4) + 1 : 1;
        G.end[k] = (c[n - 1] & 0xf) == C_S ? r.L - (c[n - 1] >> 4) : r.L;
        G.rb[k].pos = rec.offset; G.re[k].pos = rec.offset + reflen - 1;
    } else {
        G.beg[k] = (c[n - 1] & 0xf) == C_S ? (c[n - 1] >> 4) + 1 : 1;
        G.end[k] = (c[0] & 0xf) == C_S ? r.L - (c[0] >> 4) : r.L;
        G.re[k].pos = rec.offset; G.rb[k].pos = rec.offset + reflen - 1;
    }
    ++G.n;
}

HP_NOINL void regs_remain(ReadCtx &r, Regs &G, int min_thd, int max_thd)
{
    const lamsa_hp_para *P = r.cx.P;
    G.m = 0;
    if (G.n == 0) {
        if (min_thd < r.L && r.L <= max_thd) { G.r_beg[0] = 1; G.r_end[0] = r.L; G.r_bs[0] = G.r_bn[0] = G.r_es[0] = G.r_en[0] = 0; G.m = 1; }
        return;
    }
    for (int i = 1; i < G.n; ++i) {                        // stable sort by beg
        const int b = G.beg[i], e = G.end[i]; const RegB rb = G.rb[i], re = G.re[i];
        int k = i - 1;
        while (k >= 0 && G.beg[k] > b) { G.beg[k + 1] = G.beg[k]; G.end[k + 1] = G.end[k]; G.rb[k + 1] = G.rb[k]; G.re[k + 1] = G.re[k]; --k; }
        G.beg[k + 1] = b; G.end[k + 1] = e; G.rb[k + 1] = rb; G.re[k + 1] = re;
    }
    // merge neighbours closer than bwt_seed_len; a merged group is the index range [gs, ge) of the sorted intervals,
    // its ref_beg / ref_end lists are exactly rb[gs..ge) / re[gs..ge)
    int gs = 0, g_beg = G.beg[0], g_end = G.end[0];
    int prev_gs = -1, prev_ge = -1, prev_end = 0, first = 1;
    for (int i = 1; i <= G.n; ++i) {
        if (i < G.n && G.beg[i] - g_end - 1 < P->bwt_seed_len) { if (G.end[i] > g_end) g_end = G.end[i]; continue; }
        // group [gs, i) is final
        if (first) {
            if (g_beg > min_thd && g_beg - 1 <= max_thd) { const int m = G.m++; G.r_beg[m] = 1; G.r_end[m] = g_beg - 1; G.r_bs[m] = 0; G.r_bn[m] = 0; G.r_es[m] = gs; G.r_en[m] = i - gs; }
            first = 0;
        } else if (g_beg - prev_end > min_thd && g_beg - 1 - prev_end <= max_thd) {
            const int m = G.m++; G.r_beg[m] = prev_end + 1; G.r_end[m] = g_beg - 1; G.r_bs[m] = prev_gs; G.r_bn[m] = prev_ge - prev_gs; G.r_es[m] = gs; G.r_en[m] = i - gs;
        }
        prev_gs = gs; prev_ge = i; prev_end = g_end;
        if (i < G.n) { gs = i; g_beg = G.beg[i]; g_end = G.end[i]; }
    }
    if (r.L - prev_end > min_thd && r.L - prev_end <= max_thd) {
        HP_STAT(10);
        const int m = G.m++; G.r_beg[m] = prev_end + 1; G.r_end[m] = r.L; G.r_bs[m] = prev_gs; G.r_bn[m] = prev_ge - prev_gs; G.r_es[m] = 0; G.r_en[m] = 0;
    }
}

// ---- result stream (hp_batch.h) ----
struct OutBuf { int32_t *w; int n, cap; };
HP_INL void out_put(Ctx &cx, OutBuf &o, int32_t v) { if (o.n < o.cap) o.w[o.n++] = v; else cx.status |= ST_OVERFLOW; }

HP_FN void out_line(Ctx &cx, OutBuf &o, const LineRes &la)
{
    out_put(cx, o, la.line_score); out_put(cx, o, la.tol_score); out_put(cx, o, la.tol_NM); out_put(cx, o, la.cur_res_n + 1);
    for (int j = 0; j <= la.cur_res_n; ++j) {
        const Rec &rec = la.rec[j];
        out_put(cx, o, (int32_t)(rec.offset & 0xffffffffll)); out_put(cx, o, (int32_t)(rec.offset >> 32));
        out_put(cx, o, rec.chr); out_put(cx, o, rec.nstrand); out_put(cx, o, rec.score); out_put(cx, o, rec.NM); out_put(cx, o, rec.cig.n);
        if (o.n + rec.cig.n <= o.cap) {                               // the record's CIGAR, copied by the lanes (word by word it was 2 500 dependent trips per line: 14 % of the fill kernel)
            HP_G int32_t *dst = (HP_G int32_t *)(o.w + o.n);
            const HP_G cig_t *src = (const HP_G cig_t *)rec.cig.c;
            const int cn = rec.cig.n;
            wv::sync();
            for (int b0 = 0; b0 < cn; b0 += 64) { WAVE_FOR(l) { const int k = b0 + l; if (k < cn) dst[k] = (int32_t)src[k]; } }
            o.n += cn;
            wv::sync();
        }
        else cx.status |= ST_OVERFLOW;
    }
}

// frag_check over all lines of one round (frag_check.c:886-955) + get_reg (lamsa_aln.c:597) for round 1
HP_NOINL void fill_round(ReadCtx &r, FLines &F, OutBuf &o, Regs *G, int reg_cap, int scale)
{
    Ctx &cx = r.cx;
    const size_t mark = arena_mark(cx.tmp);
    const int cur_cap = (2 * r.L + 512) * scale;
    LineRes *la = (LineRes *)arena_alloc(cx, sizeof(LineRes));
    cig_t *cur_buf = (cig_t *)arena_alloc(cx, sizeof(cig_t) * (size_t)cur_cap);
    cig_t *rec_buf = (cig_t *)arena_alloc(cx, sizeof(cig_t) * (size_t)(cur_cap + 4 * HP_REC_MAX));
    if (!la || !cur_buf || !rec_buf) { arena_release(cx.tmp, mark); return; }
    for (int j = 0; j < F.n; ++j) {
        if (!fill_line(r, F, j, *la, cur_buf, cur_cap, rec_buf, cur_cap + 4 * HP_REC_MAX)) break;
        out_line(cx, o, *la);
        if (G && la->tol_score >= 0) for (int k = 0; k <= la->cur_res_n; ++k) regs_push(r, *G, reg_cap, la->rec[k]);
    }
    arena_release(cx.tmp, mark);
}

// ---- one read's view of the batch (hp_batch.h) and its scratch ----
HP_INL void read_bind(ReadCtx &r, const lamsa_hp_para &P, const RefView &ref, const BatchIn &in, int rd, char *slab, size_t slab_bytes, HP_L int32_t *lds, long long *prof, int lds_words = HP_BOTH_LDS_WORDS)
{
    r.cx.P = &P; r.cx.status = 0; r.cx.n_cells = 0; r.cx.lds_epoch = 0; r.cx.prof_dp = nullptr; r.n_pairs = 0; r.cx.lds = lds; r.cx.lds_words = lds_words; r.cx.prof = prof ? prof + (size_t)rd * 64 : nullptr;
    arena_init(r.cx.tmp, slab, slab_bytes);
    r.ref = ref;
    r.L = (int)(in.read_off[rd + 1] - in.read_off[rd]);
    r.read = in.read_seq + in.read_off[rd];
    r.seed_all = in.seed_all[rd]; r.last_len = in.last_len[rd];
    const int64_t s0 = in.seed_off[rd];
    r.seed_out = (int)(in.seed_off[rd + 1] - s0);
    r.seed_id = in.seed_id + s0; r.hit_off = in.hit_off + s0;
    r.hb = r.hit_off[0];
    r.H = (int)(r.hit_off[r.seed_out] - r.hb);
    r.h_pos = in.h_pos + r.hb; r.h_chr = in.h_chr + r.hb; r.h_cig_off = in.h_cig_off + r.hb; r.h_nm = in.h_nm + r.hb;
    r.h_len_dif = in.h_len_dif + r.hb; r.h_strand = in.h_strand + r.hb; r.h_cig_n = in.h_cig_n + r.hb; r.cig = in.cig;
    r.flip = false; r.cur_read = r.read; r.rc_ready = false; r.rc_read = nullptr; r.t_bases = 0; r.cs_words = 0;
    r.prof = r.cx.prof; r.leaf_bits = nullptr; r.leaf_on = false; r.nodes_ready = false;
}

// The packed 32-byte record, the seed slot and the initial chaining state of every hit, one hit per lane.  A hit's slot comes from a
// binary search in the read's hit offsets, which are staged in LDS first (nine dependent look-ups per hit otherwise).  The state is what
// frag_line_BCC starts from (:1315-1334: fnode_set(START, 1, NM, F_MATCH) with the pass flag MIN for the hits of seeds with at most
// first_loci_thd hits -- or for every hit when fewer than a third of the seeds are such), written here once instead of a second pass
// over the records; chain_first finds r.nodes_ready set.  The side arrays (aux_bind) must be bound.
HP_FN void nodes_fill(ReadCtx &r)
{
    const HP_G int64_t *g_hoff = (const HP_G int64_t *)r.hit_off;
    const HP_G int32_t *g_sid = (const HP_G int32_t *)r.seed_id;
    const HP_G int16_t *g_hnm = (const HP_G int16_t *)r.h_nm;
    const int64_t hb = r.hb;
    const int H = r.H, S = r.seed_out;
    const lamsa_hp_para *P = r.cx.P;
    HP_L int32_t *lo_ = r.cx.lds;
    const bool in_lds = S + 1 <= r.cx.lds_words;
    int min_num = 0;
    if (in_lds) wv::sync();                                        // whatever used this LDS before is done
    for (int s0 = 0; s0 <= S; s0 += 64) {
        wv::Lane<int> few;
        WAVE_FOR(l) {
            const int s = s0 + l;
            int f = 0;
            if (s <= S) { const int o = (int)(g_hoff[s] - hb); if (in_lds) lo_[s] = o; if (s < S) f = (int)(g_hoff[s + 1] - hb) - o <= P->first_loci_thd; }
            few[l] = f;
        }
        min_num += __builtin_popcountll(wv::ballot(few));
    }
    wv::sync();
    const bool all_min = min_num == 0 || min_num * 3 < S;          // :1324-1331
    HP_G int32_t *g_from = (HP_G int32_t *)r.n_from, *g_in_de = (HP_G int32_t *)r.n_in_de, *g_son_n = (HP_G int32_t *)r.n_son_n, *g_first = (HP_G int32_t *)r.n_first,
                 *g_last = (HP_G int32_t *)r.n_last, *g_ms = (HP_G int32_t *)r.n_max_score, *g_mn = (HP_G int32_t *)r.n_max_NM, *g_mx = (HP_G int32_t *)r.n_max_node,
                 *g_nn = (HP_G int32_t *)r.n_node_n, *g_seed = (HP_G int32_t *)r.n_seed;
    HP_G NodeS *gd = (HP_G NodeS *)r.nd;
    for (int k0 = 0; k0 < H; k0 += 64) {
        WAVE_FOR(l) {
            const int k = k0 + l;
            if (k < H) {
                int lo = 0, hi = S;                          // largest slot s with hit_off[s] - hb <= k
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((in_lds ? lo_[mid] : (int)(g_hoff[mid] - hb)) <= k) lo = mid; else hi = mid; }
                const int s = lo, b = in_lds ? lo_[s] : (int)(g_hoff[s] - hb), e = in_lds ? lo_[s + 1] : (int)(g_hoff[s + 1] - hb);
                const int nm = g_hnm[k];
                g_seed[k] = s;
                const int64_t pos = r.h_pos[k];
                const int w4 = ((int)g_sid[s] & 0xffff) | (((int)r.h_strand[k] & 0xff) << 16) | (((int)r.h_len_dif[k] & 0xff) << 24);
                const int flag = (all_min || e - b <= P->first_loci_thd) ? MIN_FLAG : MULTI_FLAG;
                hp_store16(gd + k, (int)(pos & 0xffffffffll), (int)(pos >> 32), r.h_chr[k], (s << 14) | (k - b));
                hp_store16((HP_G char *)(gd + k) + 16, w4, (flag & 0xff) | (F_INIT << 8) | (F_MATCH << 16), 1, nm);       // dp_flag | son_flag | match_flag, score, NM
                g_from[k] = -1; g_nn[k] = 1; g_in_de[k] = 0; g_son_n[k] = 0; g_first[k] = -1; g_last[k] = -1; g_ms[k] = 1; g_mn[k] = nm; g_mx[k] = k;      // fnode_set, :636
            }
        }
    }
    r.nodes_ready = true;
    wv::sync();
}

HP_INL void aux_bind(ReadCtx &r, int32_t *nm)
{
    const int c = r.H + 1;
    r.n_from = nm; r.n_in_de = nm + c; r.n_son_n = nm + 2 * c; r.n_first = nm + 3 * c;
    r.n_last = nm + 4 * c; r.n_next = nm + 5 * c; r.n_max_score = nm + 6 * c; r.n_max_NM = nm + 7 * c; r.n_max_node = nm + 8 * c;
    r.n_node_n = nm + 9 * c;
}

// ---- the per-read entry point ----
#ifdef HP_PROF
#define HP_STAMP(k) do { const long long now_ = wv::clock(); if (a.prof) a.prof[(size_t)rd * 64 + (k)] += now_ - t_last_; t_last_ = now_; } while (0)
#else
#define HP_STAMP(k) do { } while (0)
#endif

HP_NOINL void align_read(const AlignArgs &a, int rd, int wave_slot, HP_L int32_t *lds)
{
#ifdef HP_PROF
    long long t_last_ = wv::clock();
#endif
    ReadCtx r;
    read_bind(r, a.P, a.ref, a.in, rd, a.slab + (size_t)wave_slot * a.slab_per_wave, a.slab_per_wave, lds, a.prof);
    Ctx &cx = r.cx;
    const bool skip = a.in.read_skip && a.in.read_skip[rd];        // refused by the batch check: an empty result with ST_UNSUPPORTED
    if (skip) { cx.status |= ST_UNSUPPORTED; r.H = 0; r.seed_out = 0; }
    const int H = r.H;
    // read-lifetime allocations
    const int out_cap = 64 + 12 * r.L * a.scale;
    OutBuf o; o.n = 0; o.cap = out_cap;
    o.w = (int32_t *)arena_alloc(cx, sizeof(int32_t) * (size_t)out_cap);
    r.rc_read = (uint8_t *)arena_alloc(cx, (size_t)r.L + 16);
    int32_t *nm = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 11 * (size_t)(H + 1));
    int32_t *sidx = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 2 * (size_t)(H + 1));       // the sort index of the hits (hp_sort.h)
    r.nd = (NodeS *)arena_alloc(cx, sizeof(NodeS) * (size_t)(H + 1));
    const int reg_cap = 256 * a.scale;
    Regs G; G.n = 0; G.m = 0;
    G.beg = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 2 * (size_t)reg_cap); G.end = G.beg + reg_cap;
    G.rb = (RegB *)arena_alloc(cx, sizeof(RegB) * 2 * (size_t)reg_cap); G.re = G.rb + reg_cap;
    G.r_beg = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 6 * (size_t)(reg_cap + 2));
    G.r_end = G.r_beg + (reg_cap + 2); G.r_bs = G.r_end + (reg_cap + 2); G.r_bn = G.r_bs + (reg_cap + 2); G.r_es = G.r_bn + (reg_cap + 2); G.r_en = G.r_es + (reg_cap + 2);
    int n0_pos = 1, n1_pos = 2;
    const size_t sort_mark = arena_mark(cx.tmp);
    uint64_t *sort_work = (uint64_t *)arena_alloc(cx, sizeof(uint64_t) * (size_t)(H + 1));      // released once the index is built
    if (!skip && o.w && r.rc_read && nm && sidx && sort_work && r.nd && G.beg && G.rb && G.r_beg) {
        const int c = H + 1;
        { HP_T0(t_sort_);
        sort_read_hits(r.h_pos, r.h_chr, r.h_strand, H, sidx, sidx + c, sort_work, (HP_L uint64_t *)lds, HP_BOTH_LDS_WORDS / 2, a.sort_pb, a.sort_cb);
        HP_TADD(cx, 46, t_sort_); }
        arena_release(cx.tmp, sort_mark);
        r.srt = sidx; r.rnk = sidx + c;
        aux_bind(r, nm); r.n_seed = nm + 10 * c;
        nodes_fill(r);
        out_put(cx, o, 0); out_put(cx, o, 0); out_put(cx, o, 0);
        HP_STAMP(0);
        // round 1: frag_line_BCC + frag_check + get_reg   (lamsa_aln.c:857-865)
        {
            const size_t mark = arena_mark(cx.tmp);
            FLines F;
            const bool ok1 = chain_first(r, F);
            HP_STAMP(1);
            if (ok1 && F.n > 0) { o.w[n0_pos] = F.n; fill_round(r, F, o, &G, reg_cap, a.scale); }
            HP_STAMP(2);
            arena_release(cx.tmp, mark);
        }
        // round 2: frag_line_remain + frag_check          (lamsa_aln.c:867-871)
        if (!(cx.status & ST_DEAD)) {
            const size_t mark = arena_mark(cx.tmp);
            regs_remain(r, G, a.P.seed_len, r.L);
            FLines F;
            const bool ok2 = chain_remain(r, G, F);
            HP_STAMP(3);
            if (ok2 && F.n > 0) { o.w[n1_pos] = F.n; fill_round(r, F, o, nullptr, 0, a.scale); }
            HP_STAMP(4);
            arena_release(cx.tmp, mark);
        }
    }
    // publish: reserve the exact size in the global arena, copy with all lanes
    const int st = cx.status;
    int n_words = (st & ST_DEAD) || !o.w ? 3 : o.n;
    unsigned long long off = 0;
    if (wv::leader()) off = atomicAdd(a.out.cursor, (unsigned long long)n_words);
    off = (unsigned long long)wv::uni64((long long)off);
    if ((int64_t)(off + (unsigned long long)n_words) <= a.out.stream_cap) {
        int32_t *dst = a.out.stream + off;
        if (n_words == 3 && (!o.w || (st & ST_DEAD))) { dst[0] = st; dst[1] = 0; dst[2] = 0; }
        else {
            o.w[0] = st;
            wv::sync();
            for (int b = 0; b < n_words; b += 64) { WAVE_FOR(l) { const int i = b + l; if (i < n_words) dst[i] = o.w[i]; } }
        }
        a.out.read_out_off[rd] = (int64_t)off; a.out.read_out_len[rd] = n_words;
    } else { a.out.read_out_off[rd] = -1; a.out.read_out_len[rd] = 0; }
    a.out.read_status[rd] = st;
    HP_STAMP(5);
    if (a.out.read_tbases) a.out.read_tbases[rd] = (int32_t)(r.t_bases > 0x7fffffffLL ? 0x7fffffffLL : r.t_bases);
    if (a.out.read_work) {
        a.out.read_work[4 * rd] = (int32_t)(cx.n_cells > 0x7fffffffLL ? 0x7fffffffLL : cx.n_cells);
        a.out.read_work[4 * rd + 1] = (int32_t)(r.n_pairs > 0x7fffffffLL ? 0x7fffffffLL : r.n_pairs);
        a.out.read_work[4 * rd + 2] = (int32_t)(r.cs_words > 0x7fffffffLL ? 0x7fffffffLL : r.cs_words);
        a.out.read_work[4 * rd + 3] = 0;
    }
}

}  // namespace hp
