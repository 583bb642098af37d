// hp_phase.h -- the per-read path of hp_align.h cut at frag_dp_path into separate launches:
//
//   chain1  (one read per wave)   sort index, frag_line_BCC                       lamsa_dp_con.c:1305-1445
//   list    (one LINE per wave)   the line's DP jobs into the queues of the batch (small ones for the lane-per-job launch, the others for the wave-per-job launch)
//   dp      (64 jobs / one job per wave)   hp_lanedp.h / hp_wavejob.h: the CIGARs into the job arena
//   fill    (one LINE per wave)   frag_check of one line of round 1 + its get_reg  frag_check.c:856-961, lamsa_aln.c:571-605
//   chain2  (one read per wave)   get_remain_reg, frag_line_remain                lamsa_aln.c:550-569, lamsa_dp_con.c:1252-1302
//   fill    (one LINE per wave)   frag_check of one line of round 2
//   publish (one read per wave)   the read's result stream (hp_batch.h) from its lines, in line order
//
// Why: (i) the work unit of the expensive half (banded DP, junctions) becomes a line instead of a read, and the lines of a
// batch are handed out costliest first by what chaining has found (extension lengths, seeds), so a launch no longer ends
// with a few heavy reads running alone; (ii) each kernel carries only its own code and registers; (iii) the chaining
// kernels need no DP rows in LDS and the fill kernels no node arrays in their slab.
// Between the launches a read's state lives in HBM: the 32-byte node records (their TRACKED flags are what round 2 needs),
// the seed slot of every hit, the sort index, and per round one block of fragments (FLines) in an arena shared by the
// batch.  Every line writes its serialised records (+ the covered read intervals of round 1) into a second arena; the
// publish launch gathers them per read.  Lines of a read are independent (frag_check.c:886-955 reads only the line's own
// fragments), and a line that fails flags the whole read exactly as the per-read loop does (fill_round in hp_align.h).
// A read that overflows a buffer in any phase is re-run by the one-kernel path with larger buffers (hp_align_api.hip).
#pragma once
#include "hp_align.h"
#include "hp_lanedp.h"
#include "hp_wavejob.h"

namespace hp {

struct RdMeta {                  // 72 bytes per read, zeroed before chain1
    int64_t fl_off[2];           // the FLines block of each round in the fragment arena (words)
    int32_t fl_n[2], fl_tot[2], fl_nfrag[2];
    int32_t unit_base[2];        // first fill unit of each round (units of a read and round are consecutive, in line order)
    int32_t status;              // ST_* bits, OR-ed in by every phase
    int32_t tbases;              // reference bases fetched by the read's DP jobs (roofline accounting)
    int32_t cells, pairs;        // DP cells updated / chaining edge classifications executed (accounting)
    int32_t cs_words, pad_;      // seed-CIGAR words the fill read (accounting)
};

struct UnitRec {                 // one line to fill
    int32_t read, line;
    int64_t out_off;             // its serialised result in the line arena (words): out_line words, then 10 words per region
    int32_t out_len, n_reg;
    int32_t pad[2];
};

enum { PH_NBUCKET = 8, PH_REG_WORDS = 10 };

// Laid out by 128-byte lines: a line that thousands of waves update with atomics must not also hold words the same waves READ once per
// job -- every such read then queues behind the atomics in the one L2 channel that owns the line.  (Round 4: wj_bytes / wj_cells shared a
// line with half of wj_bucket_n and the wave-per-job launch took 151 ms instead of 64.)
struct alignas(128) PhaseCtl {   // counters of one launch sequence, zeroed before chain1
    alignas(128) int32_t q_head[12];   // queue heads: 0 chain1, 1 fill(round 1), 2 chain2, 3 fill(round 2), 4 publish, 5 listing, 6 lane DP, 7 wave DP, 8 wave DP (jobs that need a big slab)
    alignas(128) int32_t n_units[2];   // fill units reserved by chain1 / chain2 (may exceed unit_cap: the excess is flagged, not stored)
    int32_t bucket_n[2][PH_NBUCKET];
    alignas(128) unsigned long long fl_cursor;
    alignas(128) unsigned long long line_cursor;
    alignas(128) unsigned long long job_cursor;
    alignas(128) int32_t lj_n[2]; int32_t lj_bucket_n[2][24];   // lane-per-job DP: jobs listed per round, and per queue (kind x query-length class; LJ_NBUCKET <= 24)
    alignas(128) int32_t wj_n[2]; int32_t wj_bucket_n[2][WJ_NBUCKET];      // wave-per-job DP (hp_wavejob.h): jobs listed per round, and per cost class
    alignas(128) unsigned long long wj_bytes;          // algorithmic bytes of the wave-per-job launches: query bases + 2-bit target bases read, CIGAR words + slots written
    alignas(128) unsigned long long wj_cells;          // DP cells they updated
    // when the first and the last wave of each of the four long launches found its queue empty (wall clock, 100 MHz; the first
    // one stored complemented so that zero-initialised words work with atomicMax): last - first is the time a launch spends
    // draining, i.e. with idle wave slots
    alignas(128) unsigned long long t_first_inv[4], t_last[4];
};

struct LjRec;
struct PhaseArgs {
    lamsa_hp_para P;
    RefView ref;
    BatchIn in;
    BatchOut out;
    char *slab; size_t slab_per_wave;
    int32_t sort_pb, sort_cb;
    const int32_t *order;        // reads, costliest chaining first (or nullptr)
    int32_t n_reads;
    long long *prof;
    // state of the batch between the launches; per-hit arrays are indexed by (global hit index + read index)
    NodeS *g_nd; int32_t *g_nseed; int32_t *g_sidx;
    RdMeta *meta;
    UnitRec *units; int32_t unit_cap;            // [2][unit_cap]
    int32_t *bucket_q;                           // [2][PH_NBUCKET][unit_cap] unit indices by cost class, costliest class first
    int32_t *fl_base; int64_t fl_cap;            // fragment arena (words)
    int32_t *line_base; int64_t line_cap;        // line arena (words)
    int32_t *job_base; int64_t job_cap;          // CIGARs of the lane-per-job DPs (words, < 2^31)
    struct LjRec *ljobs; int32_t *lj_bucket; int32_t lj_cap;     // [lj_cap] job records and [LJ_NBUCKET][lj_cap] queues of job indices of the round being filled
    WjRec *wjobs; int32_t *wj_bucket; int32_t wj_cap;            // the same for the wave-per-job launch: [wj_cap] records, [WJ_NBUCKET][wj_cap] queues; wj_cap 0: no such launch
    size_t slab_fill, slab_wj;                                   // scratch of a wave of the listing / lane-DP / fill launches and of the wave-per-job launch (slab_per_wave: the chaining launches)
    size_t slab_wjb, wjb_off; int32_t n_wjb;                     // the first n_wjb waves of the wave-per-job launch also own a big slab of slab_wjb bytes (at slab + wjb_off): they take the jobs that need one
    PhaseCtl *ctl;
};

HP_INL void drain_stamp(const PhaseArgs &a, int k)
{
    if (wv::leader()) { const unsigned long long now = wv::wall(); atomicMax(&a.ctl->t_first_inv[k], ~now); atomicMax(&a.ctl->t_last[k], now); }
}

HP_INL void meta_flag(const PhaseArgs &a, int rd, const ReadCtx &r)
{
    if (wv::leader()) {
        const long long tb = r.t_bases, nc = r.cx.n_cells, np = r.n_pairs, cw = r.cs_words;
        if (r.cx.status) atomicOr(&a.meta[rd].status, r.cx.status);
        if (tb > 0) atomicAdd(&a.meta[rd].tbases, (int)(tb > 0x3fffffffLL ? 0x3fffffffLL : tb));
        if (nc > 0) atomicAdd(&a.meta[rd].cells, (int)(nc > 0x3fffffffLL ? 0x3fffffffLL : nc));
        if (np > 0) atomicAdd(&a.meta[rd].pairs, (int)(np > 0x3fffffffLL ? 0x3fffffffLL : np));
        if (cw > 0) atomicAdd(&a.meta[rd].cs_words, (int)(cw > 0x3fffffffLL ? 0x3fffffffLL : cw));
    }
}

// hand the lines of one round to the fill launch: one unit per line, queued by cost class
HP_FN void units_push(const PhaseArgs &a, ReadCtx &r, int rd, int round, const FLines &F, const FlStore &fs)
{
    RdMeta &M = a.meta[rd];
    int base = 0;
    if (wv::leader()) base = atomicAdd(&a.ctl->n_units[round], F.n);
    base = wv::uni(base);
    if (base + F.n > a.unit_cap) { r.cx.status |= ST_OVERFLOW; return; }
    M.fl_off[round] = fs.got_off; M.fl_n[round] = F.n; M.fl_tot[round] = fs.got_tot; M.fl_nfrag[round] = F.nfrag; M.unit_base[round] = base;
    const lamsa_hp_para *P = r.cx.P;
    const int tiles = (2 * P->band_w + 1 + 63) / 64;
    for (int j = 0; j < F.n; ++j) {
        // cost class of line j: the two end extensions run over the read bases outside the line (frag_check.c:576-707)
        // with a DP row per base, the junctions cost roughly per seed
        const int f0 = F.frag_off[j], f1 = F.frag_off[j + 1];
        const int n_seed = F.fr_seed_off[f1] - F.fr_seed_off[f0];
        const int sa = r.seed_id[r.n_seed[F.fr_seed[F.fr_seed_off[f0]]]], sb = r.seed_id[r.n_seed[F.fr_seed[F.fr_seed_off[f1] - 1]]];
        const int s_lo = sa < sb ? sa : sb, s_hi = sa < sb ? sb : sa;
        long long ext = (long long)(s_lo - 1 + r.seed_all - s_hi) * P->seed_step;
        if (ext < 0) ext = 0;
        const long long cost = ext * tiles * 8 + (long long)n_seed * 64;
        int b = 0;
        for (long long c = cost >> 10; c > 0 && b < PH_NBUCKET - 1; c >>= 1) ++b;
        const int bucket = PH_NBUCKET - 1 - b;                        // bucket 0 = costliest
        UnitRec &U = a.units[(size_t)round * a.unit_cap + base + j];
        U.read = rd; U.line = j; U.out_off = 0; U.out_len = 0; U.n_reg = 0;
        int at = 0;
        if (wv::leader()) at = atomicAdd(&a.ctl->bucket_n[round][bucket], 1);
        at = wv::uni(at);
        a.bucket_q[((size_t)round * PH_NBUCKET + bucket) * a.unit_cap + at] = base + j;
    }
}

HP_INL void pers_bind(ReadCtx &r, const PhaseArgs &a, int rd)
{
    const int64_t pb = r.hb + rd;
    r.nd = a.g_nd + pb; r.n_seed = a.g_nseed + pb;
    r.srt = a.g_sidx + 2 * pb; r.rnk = a.g_sidx + 2 * pb + (r.H + 1);
}

// ---------------------------------------------------------------- chain1: one read
#ifdef HP_PROF
#define PH_T0() const long long ph_t0_ = wv::clock()
#define PH_TADD(k) do { if (r.prof) r.prof[k] += wv::clock() - ph_t0_; } while (0)
#define PH_TMID(k, v) do { if (r.prof) { const long long now_ = wv::clock(); r.prof[k] += now_ - v; v = now_; } } while (0)
#else
#define PH_T0() do { } while (0)
#define PH_TADD(k) do { } while (0)
#define PH_TMID(k, v) do { } while (0)
#endif

HP_NOINL void phase_chain1(const PhaseArgs &a, int rd, int wave_slot, HP_L int32_t *lds, int lds_words = HP_CHAIN_LDS_WORDS)
{
#ifdef HP_PROF
    long long ph_t_ = wv::clock();
#endif
    if (a.in.read_skip && a.in.read_skip[rd]) { if (wv::leader()) atomicOr(&a.meta[rd].status, ST_UNSUPPORTED); return; }       // refused by the batch check: no result
    ReadCtx r;
    read_bind(r, a.P, a.ref, a.in, rd, a.slab + (size_t)wave_slot * a.slab_per_wave, a.slab_per_wave, lds, a.prof, lds_words);
    pers_bind(r, a, rd);
    Ctx &cx = r.cx;
    const int H = r.H, c = H + 1;
    int32_t *nm = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 10 * (size_t)c);
    const size_t sort_mark = arena_mark(cx.tmp);
    uint64_t *sort_work = (uint64_t *)arena_alloc(cx, sizeof(uint64_t) * (size_t)c);
    if (nm && sort_work) {
        int32_t *sidx = a.g_sidx + 2 * (r.hb + rd);
        { HP_T0(t_sort_);
        sort_read_hits(r.h_pos, r.h_chr, r.h_strand, H, sidx, sidx + c, sort_work, (HP_L uint64_t *)lds, lds_words / 2, a.sort_pb, a.sort_cb);
        HP_TADD(cx, 46, t_sort_); }
        arena_release(cx.tmp, sort_mark);
        aux_bind(r, nm);
        nodes_fill(r);
        PH_TMID(0, ph_t_);
        FLines F;
        FlStore fs; fs.base = a.fl_base; fs.cap = a.fl_cap; fs.cursor = &a.ctl->fl_cursor; fs.got_off = 0; fs.got_tot = 0;
#if defined(HP_CHAIN_STOP) && HP_CHAIN_STOP == 0
        const bool ok1 = true; F.n = 0; F.nfrag = 0;       // traffic experiment (tools/chain_stops.sh): sort index and node records only
#else
        const bool ok1 = chain_first(r, F, &fs);
#endif
        PH_TMID(1, ph_t_);
        if (ok1 && F.n > 0) units_push(a, r, rd, 0, F, fs);
        else if (!ok1 && !(cx.status & (ST_REFEXIT | ST_OVERFLOW))) cx.status |= ST_OVERFLOW;
    }
    meta_flag(a, rd, r);
}

// ---------------------------------------------------------------- fill: one line of one read
HP_NOINL void phase_fill(const PhaseArgs &a, int round, int u, int wave_slot, HP_L int32_t *lds)
{
    UnitRec &U = a.units[(size_t)round * a.unit_cap + u];
    const int rd = U.read, line = U.line;
    RdMeta &M = a.meta[rd];
    if (*(volatile int32_t *)&M.status & ST_DEAD) return;       // the read is lost already (another line or phase failed)
    PH_T0();
    ReadCtx r;
    read_bind(r, a.P, a.ref, a.in, rd, a.slab + (size_t)wave_slot * a.slab_fill, a.slab_fill, lds, a.prof, HP_LDS_WORDS);
    pers_bind(r, a, rd);
    Ctx &cx = r.cx;
    FLines F;
    F.n = M.fl_n[round]; F.nfrag = M.fl_nfrag[round];
    flines_bind(F, a.fl_base + M.fl_off[round], F.n, M.fl_tot[round]);
    F.jarena = a.job_base;
    const int cur_cap = 2 * r.L + 512;
    const int out_cap = 64 + 12 * r.L + PH_REG_WORDS * HP_REC_MAX;
    OutBuf o; o.n = 0; o.cap = out_cap;
    o.w = (int32_t *)arena_alloc(cx, sizeof(int32_t) * (size_t)out_cap);
    r.rc_read = (uint8_t *)arena_alloc(cx, (size_t)r.L + 16);
    LineRes *la = (LineRes *)arena_alloc(cx, sizeof(LineRes));
    cig_t *cur_buf = (cig_t *)arena_alloc(cx, sizeof(cig_t) * (size_t)cur_cap);
    cig_t *rec_buf = (cig_t *)arena_alloc(cx, sizeof(cig_t) * (size_t)(cur_cap + 4 * HP_REC_MAX));
    int n_reg = 0;
    if (o.w && r.rc_read && la && cur_buf && rec_buf) {
        const bool ok = fill_line(r, F, line, *la, cur_buf, cur_cap, rec_buf, cur_cap + 4 * HP_REC_MAX);
        if (!ok && !(cx.status & (ST_REFEXIT | ST_OVERFLOW))) cx.status |= ST_OVERFLOW;
        if (ok) {
            out_line(cx, o, *la);
            if (round == 0 && la->tol_score >= 0) {                                // get_reg, lamsa_aln.c:597-605
                Regs G; G.n = 0; G.m = 0;
                const size_t mark = arena_mark(cx.tmp);
                G.beg = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 2 * HP_REC_MAX); G.end = G.beg ? G.beg + HP_REC_MAX : nullptr;
                G.rb = (RegB *)arena_alloc(cx, sizeof(RegB) * 2 * HP_REC_MAX); G.re = G.rb ? G.rb + HP_REC_MAX : nullptr;
                if (G.beg && G.rb) {
                    for (int k = 0; k <= la->cur_res_n; ++k) regs_push(r, G, HP_REC_MAX, la->rec[k]);
                    for (int k = 0; k < G.n; ++k) {
                        out_put(cx, o, G.beg[k]); out_put(cx, o, G.end[k]);
                        out_put(cx, o, G.rb[k].is_rev); out_put(cx, o, G.rb[k].chr); out_put(cx, o, (int32_t)(G.rb[k].pos & 0xffffffffll)); out_put(cx, o, (int32_t)(G.rb[k].pos >> 32));
                        out_put(cx, o, G.re[k].is_rev); out_put(cx, o, G.re[k].chr); out_put(cx, o, (int32_t)(G.re[k].pos & 0xffffffffll)); out_put(cx, o, (int32_t)(G.re[k].pos >> 32));
                    }
                    n_reg = G.n;
                }
                arena_release(cx.tmp, mark);
            }
        }
    }
    if (!(cx.status & (ST_REFEXIT | ST_OVERFLOW))) {
        unsigned long long off = 0;
        if (wv::leader()) off = atomicAdd(&a.ctl->line_cursor, (unsigned long long)o.n);
        off = (unsigned long long)wv::uni64((long long)off);
        if ((int64_t)(off + (unsigned long long)o.n) <= a.line_cap) {
            HP_G int32_t *dst = (HP_G int32_t *)(a.line_base + off);
            const HP_G int32_t *src = (const HP_G int32_t *)o.w;
            wv::sync();
            for (int b = 0; b < o.n; b += 64) { WAVE_FOR(l) { const int i = b + l; if (i < o.n) dst[i] = src[i]; } }
            U.out_off = (int64_t)off; U.out_len = o.n; U.n_reg = n_reg;
        } else cx.status |= ST_OVERFLOW;
    }
    PH_TADD(round == 0 ? 2 : 4);
    meta_flag(a, rd, r);
}


// ---------------------------------------------------------------- fill, step 1: the DP jobs of the lines, computed ahead of the fill.
// phase_filllist (one line per wave) lists the line's junctions of the mismatch class with read bases in between (split_mapping,
// frag_check.c:547-559), the gaps between neighbouring seeds of its fragments (frag_extend, :360-400) and its two end extensions
// (frag_head_bound_fix :576-654, frag_tail_bound_fix :656-707), with the geometry the fill would compute, into job queues of the whole batch:
// jobs with queries of up to HP_LJ_QSMALL bases by kind and query length for phase_filldp (64 jobs per wave, one per LANE, hp_lanedp.h:
// measured on the MI355X they are cheaper that way than one per wave, profiles/r02_*), everything else by cost for phase_wavejob (one job per
// WAVE, hp_wavejob.h).  Both leave the CIGARs in the job arena and their slots in FLines::jt / gt / ht, where the fill finds them.
struct LjRec { int64_t qaddr, tk, slot; int32_t rd; uint16_t tlen; uint8_t qlen; int8_t type_comp; };      // type_comp: type (1: ksw_bi_extend(100, 100), 2: ksw_global2) | complement << 4
// queues of the lane jobs: kind 1 longest first, kind 2 longest first
enum { LJ_NCLS = HP_LJ_QSMALL / 16, LJ_NBUCKET = 2 * LJ_NCLS };
HP_INL int lj_bucket_of(int type, int qlen)
{
    const int cls = (qlen > 0 ? qlen - 1 : 0) >> 4;                            // 0 .. HP_LJ_QSMALL / 16 - 1
    return (type == 1 ? 0 : LJ_NCLS) + (LJ_NCLS - 1 - cls);
}

// end extension of line `line` as frag_head_bound_fix (head) / frag_tail_bound_fix computes it, with the line's bounds as the fill will see
// them (a '-' line's are flipped, frag_check.c:926-930; r.flip must be set accordingly).  false: there is no DP to run ahead (no read base
// to extend over, or a geometry the reference exits on, which is left to the fill to flag).
struct EndGeo { int s, read_start, read_len, chr; int64_t start0; int32_t ref_len; };
HP_INL bool end_geo(const ReadCtx &r, const FLines &F, int line, int strand, bool head, EndGeo &G)
{
    const lamsa_hp_para *P = r.cx.P;
    const int f0 = F.frag_off[line], fl = F.frag_off[line + 1] - 1;
    const int lb = strand == 1 ? F.left_bound[line] : r.seed_all + 1 - F.right_bound[line];
    const int rb = strand == 1 ? F.right_bound[line] : r.seed_all + 1 - F.left_bound[line];
    const int s_first = F.fr_seed[F.fr_seed_off[f0]], s_last = F.fr_seed[F.fr_seed_off[fl + 1] - 1];
    int64_t ref_start;
    if (head) {                                                                  // :592-640
        if (strand == 1) {
            G.s = s_last;
            const int id = sid(r, r.n_seed[G.s]);
            if (id == 1) return false;
            G.read_len = (lb == 0 ? 0 : P->seed_inv) + (id - lb - 1) * P->seed_step;
            G.read_start = lb == 0 ? 0 : lb * P->seed_step - P->seed_inv;
        } else {
            G.s = s_first;
            G.read_len = (lb == 0 ? r.last_len : P->seed_inv) + (sid(r, r.n_seed[G.s]) - 1 - lb) * P->seed_step;
            G.read_start = lb == 0 ? 0 : r.last_len + lb * P->seed_step - P->seed_inv;
        }
        if (G.read_len <= 0) return false;
        G.ref_len = G.read_len + P->hash_step * 2;
        ref_start = r.h_pos[G.s] - G.ref_len;
        if (ref_start < 1) { ref_start = 1; G.ref_len = (int32_t)(r.h_pos[G.s] - 1); }
    } else {                                                                     // :670-699
        if (strand == 1) {
            G.s = s_first;
            const int id = sid(r, r.n_seed[G.s]);
            G.read_start = id * P->seed_step - P->seed_inv;
            G.read_len = (rb == r.seed_all + 1 ? r.last_len : P->seed_inv) + (rb - 1 - id) * P->seed_step;
        } else {
            G.s = s_last;
            const int id = sid(r, r.n_seed[G.s]);
            if (id == r.seed_all) return false;
            G.read_start = id * P->seed_step - P->seed_inv + r.last_len;
            G.read_len = (rb == r.seed_all + 1 ? 0 : P->seed_inv) + (rb - 1 - id) * P->seed_step;
        }
        if (G.read_len <= 0) return false;
        G.ref_len = G.read_len + P->hash_step * 2;
        ref_start = r.h_pos[G.s] + P->seed_len + r.h_len_dif[G.s];
    }
    G.chr = r.h_chr[G.s];
    G.start0 = ref_start - 1;                                                    // pac2fa_core, bntseq.c:469-474
    const int32_t clen = r.ref.seq_len[G.chr - 1];
    if (G.start0 > clen || G.start0 < 0) return false;
    if (G.start0 + G.ref_len > clen) G.ref_len = (int32_t)(clen - G.start0);
    return G.ref_len > 0 && G.read_start >= 0 && G.read_start + G.read_len <= r.L;
}

HP_NOINL void phase_filllist(const PhaseArgs &a, int round, int u, int wave_slot, HP_L int32_t *lds)
{
    UnitRec &U = a.units[(size_t)round * a.unit_cap + u];
    const int rd = U.read, line = U.line;
    RdMeta &M = a.meta[rd];
    if (*(volatile int32_t *)&M.status & ST_DEAD) return;
    ReadCtx r;
    read_bind(r, a.P, a.ref, a.in, rd, a.slab + (size_t)wave_slot * a.slab_fill, a.slab_fill, lds, a.prof, 0);
    pers_bind(r, a, rd);
    const lamsa_hp_para *P = r.cx.P;
    const bool lane_ok = lj_params_ok(P) && a.lj_cap > 0, wave_ok = a.wj_cap > 0;
    if (!lane_ok && !wave_ok) return;
    FLines F;
    F.n = M.fl_n[round]; F.nfrag = M.fl_nfrag[round];
    flines_bind(F, a.fl_base + M.fl_off[round], F.n, M.fl_tot[round]);
    const int f0 = F.frag_off[line], nfr = F.frag_off[line + 1] - f0;
    const int p0 = F.fr_seed_off[f0], np = F.fr_seed_off[f0 + nfr] - p0;       // the line's seeds in fr_seed
    const int strand = r.h_strand[F.fr_seed[p0]];
    const int n_junc = (nfr - 1) + np, n_cand = n_junc + 2;                    // junctions, seed gaps, then the head and the tail extension
    r.flip = strand != 1;                                                      // seed ids as a '-' line sees them (frag_check.c:926)
    const int64_t rbase = a.in.read_off[rd];
    long long tb = 0;
    // the fragment of every seed of the line, written once fragment by fragment (a binary search per seed is eight dependent loads)
    int32_t *frag_of = (int32_t *)arena_alloc(r.cx, sizeof(int32_t) * (size_t)(np + 1));
    if (!frag_of) { r.flip = false; return; }
    {
        HP_G int32_t *g_fo = (HP_G int32_t *)frag_of;
        const HP_G int32_t *g_so = (const HP_G int32_t *)F.fr_seed_off;
        for (int b0 = 0; b0 < nfr; b0 += 64) { WAVE_FOR(l) { const int f = f0 + b0 + l; if (b0 + l < nfr) { for (int q = g_so[f]; q < g_so[f + 1]; ++q) g_fo[q - p0] = f; } } }
        wv::sync();
    }
    long long *cand = (long long *)arena_alloc(r.cx, sizeof(long long) * 4 * (size_t)(n_cand + 64));       // what the first pass finds out about every candidate: 32 bytes each
    if (!cand) { r.flip = false; return; }
    int n_lb[LJ_NBUCKET], n_wb[WJ_NBUCKET];
#pragma unroll
    for (int b = 0; b < LJ_NBUCKET; ++b) n_lb[b] = 0;
#pragma unroll
    for (int b = 0; b < WJ_NBUCKET; ++b) n_wb[b] = 0;
    for (int c0 = 0; c0 < n_cand; c0 += 64) {
        // ty: 0 none, 1 / 2 junction / seed gap (WJ_BI / WJ_GLOBAL), 3 / 4 head / tail; qo: where the query begins in the strand-appropriate read
        wv::Lane<int> ty, qo, ql, tl, qrev; wv::Lane<long long> tk, sl;
        WAVE_FOR(l) {
            const int c = c0 + l;
            int type = 0, qoff = 0, qlen = 0, tlen = 0, rev = 0; long long k0 = 0, slot = 0;
            if (c < nfr - 1) {                                                  // junction between fragments jf and jf + 1, split_mapping :416-470
                const int jf = f0 + c;
                const int f1 = strand == 1 ? jf + 1 : jf, f2 = strand == 1 ? jf : jf + 1;
                const int32_t *sd1 = F.fr_seed + F.fr_seed_off[f1], *sd2 = F.fr_seed + F.fr_seed_off[f2];
                const int n1 = F.fr_seed_off[f1 + 1] - F.fr_seed_off[f1], n2 = F.fr_seed_off[f2 + 1] - F.fr_seed_off[f2];
                int s1, s2;
                if (r.h_strand[sd1[0]] == 1) { s1 = sd1[0]; s2 = sd2[n2 - 1]; } else { s1 = sd1[n1 - 1]; s2 = sd2[0]; }
                const int64_t at1_off = r.h_pos[s1], at2_off = r.h_pos[s2];
                const int at1_ld = r.h_len_dif[s1], at1_chr = r.h_chr[s1];
                const int id1 = sid(r, r.n_seed[s1]), did = sid(r, r.n_seed[s2]) - id1;
                const int s_qlen = did * P->seed_step - P->seed_len;
                const int64_t exp = at1_off + at1_ld + (int64_t)(did * P->seed_step);
                const int dis = (int)(at2_off - exp);
                const int match_dis = P->match_dis * ((P->aln_mode & 2) ? did : 1);
                if (s_qlen > 0 && dis <= match_dis && dis >= -match_dis && s_qlen + dis >= 0) {
                    const int64_t start0 = at1_off + P->seed_len + at1_ld - 1;
                    const int32_t clen = r.ref.seq_len[at1_chr - 1];
                    if (start0 <= clen && start0 >= 0) {                        // pac2fa_core, bntseq.c:469-474
                        int tl_ = s_qlen + dis;
                        if (start0 + tl_ > clen) tl_ = (int)(clen - start0);
                        type = 1; qlen = s_qlen; tlen = tl_; slot = (F.jt + 4 * jf) - a.fl_base; k0 = r.ref.seq_off[at1_chr - 1] + start0;
                        qoff = (strand == 1 ? 0 : r.last_len) + id1 * P->seed_step - P->seed_inv;      // get_read_intv, :116
                    }
                }
            } else if (c < n_junc) {                                            // gap in front of the seed at position p, frag_extend :360-385
                const int p = p0 + (c - (nfr - 1));
                const int lo = ((const HP_G int32_t *)frag_of)[p - p0];        // its fragment
                const int fb = F.fr_seed_off[lo], fe = F.fr_seed_off[lo + 1], i = p - fb, seed_n = fe - fb;
                const int ip = strand == 1 ? i + 1 : i - 1;                     // the seed walked before it
                if (seed_n > 1 && ip >= 0 && ip < seed_n) {
                    const int s = F.fr_seed[p], last = F.fr_seed[fb + ip];
                    const int64_t start = r.h_pos[last] + P->seed_len - 1 + r.h_len_dif[last];
                    int len2 = (int)(r.h_pos[s] - 1 - start);
                    bool ok = true;
                    const int32_t clen = r.ref.seq_len[r.h_chr[last] - 1];
                    if (len2 <= 0) len2 = 0;
                    else if (start > clen || start < 0) ok = false;
                    else if (start + len2 > clen) len2 = (int)(clen - start);
                    const int idl = sid(r, r.n_seed[last]), ids = sid(r, r.n_seed[s]);
                    const int qi = (strand == 1 ? 0 : r.last_len) + idl * P->seed_step - P->seed_inv, qe = (strand == 1 ? 0 : r.last_len) + (ids - 1) * P->seed_step;
                    const int len1 = qe > qi ? qe - qi : 0;
                    if (ok) { type = 2; qlen = len1; tlen = len2; slot = (F.gt + 4 * p) - a.fl_base; qoff = qi; k0 = r.ref.seq_off[r.h_chr[last] - 1] + start; }
                }
            } else if (c < n_cand && wave_ok) {                                 // the end extensions, frag_head_bound_fix :576 / frag_tail_bound_fix :656
                const bool head = c == n_junc;
                EndGeo G;
                if (end_geo(r, F, line, strand, head, G)) {
                    type = head ? 3 : 4; qlen = G.read_len; tlen = G.ref_len; slot = (F.ht + 16 * line + (head ? 0 : 8)) - a.fl_base;
                    // the head runs on both sequences reversed (ksw_extend_r, src/ksw.c:820): query base j = base read_len - 1 - j of the interval
                    qoff = head ? G.read_start + G.read_len - 1 : G.read_start; rev = head;
                    k0 = r.ref.seq_off[G.chr - 1] + G.start0 + (head ? G.ref_len - 1 : 0);
                }
            }
            // which launch takes it
            const bool small = type != 0 && type <= 2 && qlen <= HP_LJ_QSMALL && tlen <= HP_LJ_TSMALL;
            if (type != 0 && !(small ? lane_ok : wave_ok)) type = 0;
            if (type != 0 && !small) type |= 16;                                // bit 4: a job of the wave-per-job launch
            ty[l] = type; qo[l] = qoff; ql[l] = qlen; tl[l] = tlen; sl[l] = slot; tk[l] = k0; qrev[l] = rev;
        }
        // noted in the wave's slab; the queues are entered once per LINE (below): a reservation per 64 candidates made every wave wait for
        // ~10 atomics on the same few counters, 8 192 waves at a time (the listing launch was 98 % waiting: profiles/r04_ont10k_pmc.json)
        WAVE_FOR(l) {
            const int c = c0 + l;
            if (c < n_cand) { HP_G int32_t *cw = (HP_G int32_t *)cand + 8 * (size_t)c; cw[0] = ty[l]; cw[1] = qo[l]; cw[2] = ql[l]; cw[3] = tl[l] | (qrev[l] << 30); *(HP_G long long *)(cw + 4) = tk[l]; *(HP_G long long *)(cw + 6) = sl[l]; }
        }
#pragma unroll
        for (int b = 0; b < LJ_NBUCKET; ++b) { wv::Lane<int> inb; WAVE_FOR(l) inb[l] = ty[l] != 0 && !(ty[l] & 16) && lj_bucket_of(ty[l], ql[l]) == b; n_lb[b] += __builtin_popcountll(wv::ballot(inb)); }
        { wv::Lane<int> wb; WAVE_FOR(l) wb[l] = (ty[l] & 16) ? wj_bucket_of(P, ty[l] & 15, ql[l], tl[l], wj_need(P, ty[l] & 15, ql[l], tl[l]) > (long long)a.slab_wj) : -1;
#pragma unroll
          for (int b = 0; b < WJ_NBUCKET; ++b) { wv::Lane<int> inb; WAVE_FOR(l) inb[l] = wb[l] == b; n_wb[b] += __builtin_popcountll(wv::ballot(inb)); } }
    }
    // ---- one reservation per queue for the whole line: lanes 0 .. LJ_NBUCKET - 1 the lane-job queues, the next WJ_NBUCKET the wave-job queues, then the two record arrays
    int n_l = 0, n_w = 0;
#pragma unroll
    for (int b = 0; b < LJ_NBUCKET; ++b) n_l += n_lb[b];
#pragma unroll
    for (int b = 0; b < WJ_NBUCKET; ++b) n_w += n_wb[b];
    if (n_l + n_w > 0) {
        wv::Lane<int> got;
        WAVE_FOR(l) {
            int v = 0, add = 0; int32_t *ctr = nullptr;
#pragma unroll
            for (int b = 0; b < LJ_NBUCKET; ++b) if (l == b) { add = n_lb[b]; ctr = &a.ctl->lj_bucket_n[round][b]; }
#pragma unroll
            for (int b = 0; b < WJ_NBUCKET; ++b) if (l == LJ_NBUCKET + b) { add = n_wb[b]; ctr = &a.ctl->wj_bucket_n[round][b]; }
            if (l == LJ_NBUCKET + WJ_NBUCKET) { add = n_l; ctr = &a.ctl->lj_n[round]; }
            if (l == LJ_NBUCKET + WJ_NBUCKET + 1) { add = n_w; ctr = &a.ctl->wj_n[round]; }
            if (ctr && add > 0) v = atomicAdd(ctr, add);
            got[l] = v;
        }
        int base_l = wv::bcast(got, LJ_NBUCKET + WJ_NBUCKET), base_w = wv::bcast(got, LJ_NBUCKET + WJ_NBUCKET + 1);
        // a queue or a record array that is full: the line's jobs of that launch stay with the fill (the counters have moved on: the launches
        // clamp what they read to the capacities)
        bool ok_l = n_l > 0 && base_l + n_l <= a.lj_cap, ok_w = n_w > 0 && base_w + n_w <= a.wj_cap;
        int pos_lb[LJ_NBUCKET], pos_wb[WJ_NBUCKET];
#pragma unroll
        for (int b = 0; b < LJ_NBUCKET; ++b) { pos_lb[b] = wv::bcast(got, b); if (n_lb[b] > 0 && pos_lb[b] + n_lb[b] > a.lj_cap) ok_l = false; }
#pragma unroll
        for (int b = 0; b < WJ_NBUCKET; ++b) { pos_wb[b] = wv::bcast(got, LJ_NBUCKET + b); if (n_wb[b] > 0 && pos_wb[b] + n_wb[b] > a.wj_cap) ok_w = false; }
        // (a reservation that does not fit leaves its queue slots marked empty: the DP launches skip them)
#pragma unroll
        for (int b = 0; b < LJ_NBUCKET; ++b) if (!ok_l && n_lb[b] > 0) { for (int k0 = 0; k0 < n_lb[b]; k0 += 64) { WAVE_FOR(l) { const int k = pos_lb[b] + k0 + l; if (k0 + l < n_lb[b] && k < a.lj_cap) a.lj_bucket[(size_t)b * a.lj_cap + k] = -1; } } }
#pragma unroll
        for (int b = 0; b < WJ_NBUCKET; ++b) if (!ok_w && n_wb[b] > 0) { for (int k0 = 0; k0 < n_wb[b]; k0 += 64) { WAVE_FOR(l) { const int k = pos_wb[b] + k0 + l; if (k0 + l < n_wb[b] && k < a.wj_cap) a.wj_bucket[(size_t)b * a.wj_cap + k] = -1; } } }
        wv::sync();
        for (int c0 = 0; c0 < n_cand && (ok_l || ok_w); c0 += 64) {
            wv::Lane<int> ty, qo, ql, tl, qrev; wv::Lane<long long> tk, sl;
            WAVE_FOR(l) {
                const int c = c0 + l;
                ty[l] = 0; qo[l] = 0; ql[l] = 0; tl[l] = 0; qrev[l] = 0; tk[l] = 0; sl[l] = 0;
                if (c < n_cand) { const HP_G int32_t *cw = (const HP_G int32_t *)cand + 8 * (size_t)c; ty[l] = cw[0]; qo[l] = cw[1]; ql[l] = cw[2]; tl[l] = cw[3] & 0x3fffffff; qrev[l] = (cw[3] >> 30) & 1; tk[l] = *(const HP_G long long *)(cw + 4); sl[l] = *(const HP_G long long *)(cw + 6); }
            }
            wv::Lane<int> has, big, tbl;
            WAVE_FOR(l) { has[l] = ok_l && ty[l] != 0 && !(ty[l] & 16); big[l] = ok_w && (ty[l] & 16) != 0; tbl[l] = (has[l] || big[l]) ? tl[l] : 0; }
            const unsigned long long m = wv::ballot(has), mw = wv::ballot(big);
            if (m) {                                                            // ---- the lane jobs
                wv::Lane<int> at;
                WAVE_FOR(l) {
                    at[l] = base_l + __builtin_popcountll(m & ((1ull << l) - 1));
                    if (has[l]) {
                        LjRec &J = a.ljobs[at[l]];
                        // the query in the read as stored: a '-' line reads the reverse complement, base j of it is the complement of base L-1-j
                        J.qaddr = strand == 1 ? rbase + qo[l] : rbase + (r.L - 1 - qo[l]);
                        J.tk = tk[l]; J.slot = sl[l]; J.rd = rd; J.qlen = (uint8_t)ql[l]; J.tlen = (uint16_t)tl[l]; J.type_comp = (int8_t)(ty[l] | (strand == 1 ? 0 : 16));
                    }
                }
                base_l += __builtin_popcountll(m);
#pragma unroll
                for (int b = 0; b < LJ_NBUCKET; ++b) {                          // into the queue of its kind and length class
                    wv::Lane<int> inb;
                    WAVE_FOR(l) inb[l] = has[l] && lj_bucket_of(ty[l], ql[l]) == b;
                    const unsigned long long mb = wv::ballot(inb);
                    if (!mb) continue;
                    WAVE_FOR(l) { if (inb[l]) a.lj_bucket[(size_t)b * a.lj_cap + pos_lb[b] + __builtin_popcountll(mb & ((1ull << l) - 1))] = at[l]; }
                    pos_lb[b] += __builtin_popcountll(mb);
                }
            }
            if (mw) {                                                           // ---- the wave jobs
                wv::Lane<int> at, bk;
                WAVE_FOR(l) {
                    at[l] = base_w + __builtin_popcountll(mw & ((1ull << l) - 1)); bk[l] = -1;
                    if (big[l]) {
                        const int type = ty[l] & 15;
                        WjRec &J = a.wjobs[at[l]];
                        // query base j of the job = base qo + j (qo - j when walked backwards) of the strand-appropriate read; in the read as
                        // stored a '-' line's base x is the complement of base L - 1 - x, so its walk runs the other way
                        const int back = qrev[l] ? 1 : 0, comp = strand == 1 ? 0 : 1;
                        J.qaddr = strand == 1 ? rbase + qo[l] : rbase + (r.L - 1 - qo[l]);
                        J.tk = tk[l]; J.slot = sl[l]; J.rd = rd; J.qlen = ql[l]; J.tlen = tl[l];
                        J.type_comp = type | (comp << 4) | ((back ^ comp) << 5) | (back << 6);
                        bk[l] = wj_bucket_of(P, type, ql[l], tl[l], wj_need(P, type, ql[l], tl[l]) > (long long)a.slab_wj);
                    }
                }
                base_w += __builtin_popcountll(mw);
#pragma unroll
                for (int b = 0; b < WJ_NBUCKET; ++b) {
                    wv::Lane<int> inb;
                    WAVE_FOR(l) inb[l] = bk[l] == b;
                    const unsigned long long mb = wv::ballot(inb);
                    if (!mb) continue;
                    WAVE_FOR(l) { if (inb[l]) a.wj_bucket[(size_t)b * a.wj_cap + pos_wb[b] + __builtin_popcountll(mb & ((1ull << l) - 1))] = at[l]; }
                    pos_wb[b] += __builtin_popcountll(mb);
                }
            }
            tb += wv::reduce_sum(tbl);
        }
    }
    r.flip = false;
    r.t_bases = tb;
    meta_flag(a, rd, r);
}

// group g of 64 jobs of the round's queues (the caller maps g to a queue and an offset)
HP_INL void phase_filldp(const PhaseArgs &a, int round, int bucket, int off, int wave_slot, HP_L int32_t *lds, int qcap)
{
    const lamsa_hp_para *P = &a.P;
    const int n_in = a.ctl->lj_bucket_n[round][bucket] < a.lj_cap ? a.ctl->lj_bucket_n[round][bucket] : a.lj_cap;
    const int cnt = n_in - off < 64 ? n_in - off : 64;
    char *slab = a.slab + (size_t)wave_slot * a.slab_fill;
    cig_t *cbuf = (cig_t *)slab;                                               // per lane three CIGAR buffers
    uint8_t *zbuf = (uint8_t *)(slab + sizeof(cig_t) * 3 * HP_LJ_CIG * 64);     // the lane-interleaved direction matrices
    if (sizeof(cig_t) * 3 * HP_LJ_CIG * 64 + (size_t)qcap * HP_LJ_TSMALL * 64 + 64 > a.slab_fill) return;
    const int32_t *bq = a.lj_bucket + (size_t)bucket * a.lj_cap + off;
    wv::Lane<int> nw, rdl, cel; wv::Lane<long long> slotl;
    wv::sync();
    WAVE_FOR(l) {
        nw[l] = 0; rdl[l] = -1; cel[l] = 0; slotl[l] = 0;
        if (l < cnt && bq[l] >= 0) {
            const LjRec R = a.ljobs[bq[l]];
            LaneJob J;
            const int comp = (R.type_comp >> 4) & 1, type = R.type_comp & 15;
            J.q = (const HP_G uint8_t *)(a.in.read_seq + R.qaddr); J.qs = comp ? -1 : 1; J.qcomp = comp; J.qlen = R.qlen; J.pac = (const HP_G uint8_t *)a.ref.pac; J.tk = R.tk; J.ts = 1; J.tlen = R.tlen;
            J.z = (HP_G uint8_t *)zbuf; J.zl = l; J.zs = qcap; J.cells = 0; J.row = lds + l; J.qrow = (HP_L uint8_t *)(lds + (qcap + 2) * 64) + l; J.rev = 0;
            lj_stage_query(J);
            LCig out, Lc, Rc;
            out.c = cbuf + (size_t)l * 3 * HP_LJ_CIG; out.n = 0; Lc.c = out.c + HP_LJ_CIG; Lc.n = 0; Rc.c = Lc.c + HP_LJ_CIG; Rc.n = 0;
            if (type == 1) lj_bi_extend(P, J, 100, 100, Lc, Rc, out);
            else lj_global(P, J, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, P->band_w, &out);
            nw[l] = out.n; rdl[l] = R.rd; cel[l] = (int)J.cells; slotl[l] = R.slot | ((long long)R.tlen << 48);
        }
    }
    // publish: one reservation in the job arena per group
    wv::Lane<int> pre = nw;
    wv::scan_add_excl(pre);
    const int total = wv::reduce_sum(nw);
    unsigned long long base = 0;
    if (wv::leader()) base = atomicAdd(&a.ctl->job_cursor, (unsigned long long)total);
    base = (unsigned long long)wv::uni64((long long)base);
    if ((int64_t)(base + (unsigned long long)total) > a.job_cap) return;       // arena full: these stay with the fill
    WAVE_FOR(l) {
        if (l < cnt && bq[l] >= 0) {
            const cig_t *src = cbuf + (size_t)l * 3 * HP_LJ_CIG;
            int32_t *dst = a.job_base + base + pre[l];
            for (int k = 0; k < nw[l]; ++k) dst[k] = src[k];
            int32_t *slot = a.fl_base + (slotl[l] & 0xffffffffffffll);
            slot[0] = (int32_t)(base + pre[l]); slot[1] = nw[l]; slot[2] = (int32_t)(slotl[l] >> 48); slot[3] = 1;
            if (cel[l] > 0) atomicAdd(&a.meta[rdl[l]].cells, cel[l]);
        }
    }
    wv::sync();
}

// job `g` of the round's wave-job queues b0 .. b1 - 1, costliest class first (hp_wavejob.h); big: the jobs that need a big slab (this wave owns one)
#ifdef HP_WJ_NOINL
HP_NOINL
#else
HP_INL
#endif
void phase_wavejob(const PhaseArgs &a, int round, int g, bool big, int wave_slot, HP_L int32_t *lds)
{
    const int b1 = big ? WJ_NBIG : WJ_NBUCKET;
    int b = big ? 0 : WJ_NBIG;
    for (; b < b1 - 1; ++b) { const int nb = a.ctl->wj_bucket_n[round][b] < a.wj_cap ? a.ctl->wj_bucket_n[round][b] : a.wj_cap; if (g < nb) break; g -= nb; }
    const int ji = wv::uni(a.wj_bucket[(size_t)b * a.wj_cap + g]);
    if (ji < 0) return;                                                        // (a slot of a reservation that did not fit)
    const WjRec R = a.wjobs[ji];
    const int rd = wv::uni(R.rd), type = wv::uni(R.type_comp) & 15, comp = (wv::uni(R.type_comp) >> 4) & 1;
    const int qs = (wv::uni(R.type_comp) >> 5) & 1 ? -1 : 1, ts = (wv::uni(R.type_comp) >> 6) & 1 ? -1 : 1;
    const int qlen = wv::uni(R.qlen), tlen = wv::uni(R.tlen);
    if (*(volatile int32_t *)&a.meta[rd].status & ST_DEAD) return;
    Ctx cx;
    cx.P = &a.P; cx.lds = lds; cx.lds_words = HP_WJ_LDS_WORDS; cx.status = 0; cx.n_cells = 0; cx.lds_epoch = 0; cx.prof = a.prof ? a.prof + (size_t)rd * 64 : nullptr;
    cx.prof_dp = a.prof ? a.prof + ((size_t)a.n_reads + 1 + rd) * 64 : nullptr;          // (the second half of the diagnostic buffer)
    if (big) arena_init(cx.tmp, a.slab + a.wjb_off + (size_t)wave_slot * a.slab_wjb, a.slab_wjb);
    else arena_init(cx.tmp, a.slab + (size_t)wave_slot * a.slab_wj, a.slab_wj);
    CigV out;
    if (!cig_alloc(cx, out, qlen + tlen + 64)) return;
    WjOut o;
    const lamsa_hp_para *P = &a.P;
    const bool ok = wj_run(cx, a.in.read_seq, a.ref.pac, type, comp, wv::uni64(R.qaddr), qs, qlen, wv::uni64(R.tk), ts, tlen,
                           P->band_w, type == WJ_BI ? 100 : P->seed_len * P->match, out, o);
    if (cx.n_cells > 0 && wv::leader()) { atomicAdd(&a.meta[rd].cells, (int)(cx.n_cells > 0x3fffffffLL ? 0x3fffffffLL : cx.n_cells)); atomicAdd(&a.ctl->wj_cells, (unsigned long long)cx.n_cells); }
    if (!ok) return;                                                           // left to the fill, which runs the job itself and flags what there is to flag
    unsigned long long base = 0;
    if (wv::leader()) base = atomicAdd(&a.ctl->job_cursor, (unsigned long long)out.n);
    base = (unsigned long long)wv::uni64((long long)base);
    if ((int64_t)(base + (unsigned long long)out.n) > a.job_cap) return;       // arena full: stays with the fill
    {
        HP_G int32_t *dst = (HP_G int32_t *)(a.job_base + base);
        const HP_G cig_t *src = (const HP_G cig_t *)out.c;
        for (int b0 = 0; b0 < out.n; b0 += 64) { WAVE_FOR(l) { const int i = b0 + l; if (i < out.n) dst[i] = src[i]; } }
        HP_G int32_t *slot = (HP_G int32_t *)(a.fl_base + wv::uni64(R.slot));
        WAVE_FOR(l) {
            if (type == WJ_BI || type == WJ_GLOBAL) { if (l < 4) slot[l] = l == 0 ? (int32_t)base : (l == 1 ? out.n : (l == 2 ? tlen : 1)); }
            else if (l < 8) slot[l] = l == 0 ? (int32_t)base : (l == 1 ? out.n : (l == 2 ? o.reflen : (l == 3 ? o.readlen : (l == 4 ? 1 : 0))));
        }
    }
    if (wv::leader()) atomicAdd(&a.ctl->wj_bytes, (unsigned long long)(qlen + (tlen + 3) / 4 + 4 * out.n + 32));
    HP_STAT(9);
    wv::sync();
}

// ---------------------------------------------------------------- chain2: one read
HP_NOINL void phase_chain2(const PhaseArgs &a, int rd, int wave_slot, HP_L int32_t *lds, int lds_words = HP_CHAIN_LDS_WORDS)
{
    RdMeta &M = a.meta[rd];
    if (M.status & ST_DEAD) return;
    PH_T0();
    ReadCtx r;
    read_bind(r, a.P, a.ref, a.in, rd, a.slab + (size_t)wave_slot * a.slab_per_wave, a.slab_per_wave, lds, a.prof, lds_words);
    pers_bind(r, a, rd);
    Ctx &cx = r.cx;
    const int H = r.H, c = H + 1;
    int32_t *nm = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 10 * (size_t)c);
    const int reg_cap = 256;
    Regs G; G.n = 0; G.m = 0;
    G.beg = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 2 * (size_t)reg_cap); G.end = G.beg + reg_cap;
    G.rb = (RegB *)arena_alloc(cx, sizeof(RegB) * 2 * (size_t)reg_cap); G.re = G.rb + reg_cap;
    G.r_beg = (int32_t *)arena_alloc(cx, sizeof(int32_t) * 6 * (size_t)(reg_cap + 2));
    G.r_end = G.r_beg + (reg_cap + 2); G.r_bs = G.r_end + (reg_cap + 2); G.r_bn = G.r_bs + (reg_cap + 2); G.r_es = G.r_bn + (reg_cap + 2); G.r_en = G.r_es + (reg_cap + 2);
    if (nm && G.beg && G.rb && G.r_beg) {
        aux_bind(r, nm);
        // the covered read intervals of round 1, in line and record order (push_reg_res, lamsa_aln.c:571-595)
        for (int j = 0; j < M.fl_n[0]; ++j) {
            const UnitRec &U = a.units[M.unit_base[0] + j];
            const int32_t *w = a.line_base + U.out_off + (U.out_len - PH_REG_WORDS * U.n_reg);
            for (int k = 0; k < U.n_reg; ++k, w += PH_REG_WORDS) {
                if (G.n >= reg_cap) { cx.status |= ST_OVERFLOW; break; }
                const int g = G.n++;
                G.beg[g] = w[0]; G.end[g] = w[1];
                G.rb[g].is_rev = w[2]; G.rb[g].chr = w[3]; G.rb[g].pos = (int64_t)(((unsigned long long)(unsigned)w[5] << 32) | (unsigned)w[4]);
                G.re[g].is_rev = w[6]; G.re[g].chr = w[7]; G.re[g].pos = (int64_t)(((unsigned long long)(unsigned)w[9] << 32) | (unsigned)w[8]);
            }
        }
        wv::sync();
        if (!(cx.status & ST_OVERFLOW)) {
            regs_remain(r, G, a.P.seed_len, r.L);
            FLines F;
            FlStore fs; fs.base = a.fl_base; fs.cap = a.fl_cap; fs.cursor = &a.ctl->fl_cursor; fs.got_off = 0; fs.got_tot = 0;
            const bool ok2 = chain_remain(r, G, F, &fs);
            if (ok2 && F.n > 0) units_push(a, r, rd, 1, F, fs);
            else if (!ok2 && !(cx.status & (ST_REFEXIT | ST_OVERFLOW))) cx.status |= ST_OVERFLOW;
        }
    }
    PH_TADD(3);
    meta_flag(a, rd, r);
}

// ---------------------------------------------------------------- publish: one read's result stream
HP_NOINL void phase_publish(const PhaseArgs &a, int rd)
{
    const RdMeta &M = a.meta[rd];
    const int st = M.status;
    const bool dead = (st & ST_DEAD) != 0;
    int n_words = 3;
    if (!dead)
        for (int round = 0; round < 2; ++round)
            for (int j = 0; j < M.fl_n[round]; ++j) { const UnitRec &U = a.units[(size_t)round * a.unit_cap + M.unit_base[round] + j]; n_words += U.out_len - PH_REG_WORDS * U.n_reg; }
    unsigned long long off = 0;
    if (wv::leader()) off = atomicAdd(a.out.cursor, (unsigned long long)n_words);
    off = (unsigned long long)wv::uni64((long long)off);
    if ((int64_t)(off + (unsigned long long)n_words) <= a.out.stream_cap) {
        HP_G int32_t *dst = (HP_G int32_t *)(a.out.stream + off);
        dst[0] = st; dst[1] = dead ? 0 : M.fl_n[0]; dst[2] = dead ? 0 : M.fl_n[1];
        int at = 3;
        if (!dead)
            for (int round = 0; round < 2; ++round)
                for (int j = 0; j < M.fl_n[round]; ++j) {
                    const UnitRec &U = a.units[(size_t)round * a.unit_cap + M.unit_base[round] + j];
                    const int n = U.out_len - PH_REG_WORDS * U.n_reg;
                    const HP_G int32_t *src = (const HP_G int32_t *)(a.line_base + U.out_off);
                    for (int b = 0; b < n; b += 64) { WAVE_FOR(l) { const int i = b + l; if (i < n) dst[at + i] = src[i]; } }
                    at += n;
                }
        a.out.read_out_off[rd] = (int64_t)off; a.out.read_out_len[rd] = n_words;
    } else { a.out.read_out_off[rd] = -1; a.out.read_out_len[rd] = 0; }
    a.out.read_status[rd] = st;
    if (a.out.read_tbases) a.out.read_tbases[rd] = M.tbases;
    if (a.out.read_work) { a.out.read_work[4 * rd] = M.cells; a.out.read_work[4 * rd + 1] = M.pairs; a.out.read_work[4 * rd + 2] = M.cs_words; a.out.read_work[4 * rd + 3] = 0; }
}

// launch accounting for the host: the earlier launches are complete when publish starts
HP_INL void publish_diag(const PhaseArgs &a)
{
    if (a.out.diag && wv::leader()) {
        for (int k = 0; k < 4; ++k) { a.out.diag[2 * k] = ~a.ctl->t_first_inv[k]; a.out.diag[2 * k + 1] = a.ctl->t_last[k]; }
        a.out.diag[8] = (unsigned long long)a.ctl->n_units[0]; a.out.diag[9] = (unsigned long long)a.ctl->n_units[1];
        a.out.diag[10] = a.ctl->fl_cursor; a.out.diag[11] = a.ctl->line_cursor;
        a.out.diag[12] = (unsigned long long)(a.ctl->wj_n[0] + a.ctl->wj_n[1]); a.out.diag[13] = a.ctl->wj_bytes; a.out.diag[14] = (unsigned long long)(a.ctl->lj_n[0] + a.ctl->lj_n[1]);
        a.out.diag[15] = a.ctl->job_cursor; a.out.diag[16] = a.ctl->wj_cells;
    }
}

}  // namespace hp
