// emu_api.cpp -- TEST INFRASTRUCTURE: compiles the device sources (lamsa_amd/csrc/hp_*.h)
// against the CPU lane emulation (tests/emu/hp/wave.h) so that `-m "not gpu"` tests and
// sanitizers can run the kernel algorithms on the CPU.  Not part of the product library.
#include <stdlib.h>
#include <string.h>
#include <vector>
int g_emu_cl_cap = 1 << 30;            // tests: clusters larger than this take the HBM path of the main chaining pass (hp_cluster.h)
#define HP_CL_CAP_RT(cap) (g_emu_cl_cap < (cap) ? g_emu_cl_cap : (cap))
int g_emu_gaptab_cap = 1 << 30;        // tests: reads with more seed slots than this scan their gaps by seed range (hp_gaps.h)
#define HP_GAPTAB_CAP_RT(cap) (g_emu_gaptab_cap < (cap) ? g_emu_gaptab_cap : (cap))
int g_emu_gap_mcap = 1 << 30;          // tests: gaps with more survivors than this take the wave-wide routine (hp_gaps.h)
#define HP_GAP_MCAP_RT(cap) (g_emu_gap_mcap < (cap) ? g_emu_gap_mcap : (cap))
int g_emu_wj_small = 0;                // tests: the ordinary slab of a wave job in bytes (0: as large as a big one), so that small inputs have jobs that need a big slab
int g_emu_wave_jobs = 1;               // tests: 0 = no wave-per-job launch (hp_wavejob.h): the fill runs the junctions beyond a lane job and the end extensions itself
int g_emu_frag_block_min = 3;      // tests: fragments with fewer seed steps to go than this are merged step by step (hp_fill.h: frag_steps_block); 1 << 30: never in blocks
#define HP_FRAG_BLOCK_MIN g_emu_frag_block_min
int g_emu_pk = 1;                      // tests: 0 = extensions of 63 .. 254 query bases take the int32 register sets instead of the packed int16 routine (hp_ksw.h)
#define HP_PK_RT g_emu_pk
#include <vector>
std::vector<long long> g_emu_dplog;    // one record per DP call of the fill: kind (0 extension, 1 global), query, target, band, cells
int g_emu_dplog_on = 0;
#define HP_DPLOG(kind, qlen, tlen, w, cells) do { if (g_emu_dplog_on) { g_emu_dplog.push_back(kind); g_emu_dplog.push_back(qlen); g_emu_dplog.push_back(tlen); g_emu_dplog.push_back(w); g_emu_dplog.push_back(cells); } } while (0)
long long g_emu_stat[32];              // path counters (HP_STAT slots of the device sources)
#define HP_STAT(i) (++g_emu_stat[i])
#include "hp_dp_batch.h"

using namespace hp;

extern "C" int emu_dp_batch(const lamsa_hp_para *P, int n, const uint8_t *seq,
                            const int64_t *q_off, const int32_t *qlen, const int64_t *t_off, const int32_t *tlen,
                            const int32_t *kind, const int32_t *w, const int32_t *h0,
                            int32_t *score, int32_t *qle, int32_t *tle, int32_t *status,
                            int32_t *cig_n, const int64_t *cig_cap_off, int32_t *cig, size_t slab_bytes)
{
    DpBatchArgs a;
    a.P = *P; a.n_jobs = n; a.seq = seq; a.q_off = q_off; a.t_off = t_off; a.qlen = qlen; a.tlen = tlen;
    a.kind = kind; a.w = w; a.h0 = h0; a.score = score; a.qle = qle; a.tle = tle; a.status = status; a.cig_n = cig_n;
    a.cig_cap_off = cig_cap_off; a.cig = cig;
    std::vector<char> slab(slab_bytes);
    a.slab = slab.data(); a.slab_per_wave = slab_bytes; a.counter = nullptr;
    static thread_local int32_t lds[HP_BOTH_LDS_WORDS];
    for (int j = 0; j < n; ++j) dp_run_job(a, j, 0, lds);
    return 0;
}

// ---------------------------------------------------------------- whole per-read path, emulated
#include <algorithm>
#include "hp_phase.h"

// key field widths of a batch, as the product's validation pass derives them
static void emu_sort_widths(const BatchIn &in, int n_reads, int &pb, int &cb)
{
    const int64_t n_hits = n_reads ? in.hit_off[in.seed_off[n_reads]] : 0;
    long long mp = 0, mc = 0;
    for (int64_t k = 0; k < n_hits; ++k) { mp = in.h_pos[k] > mp ? in.h_pos[k] : mp; mc = in.h_chr[k] > mc ? in.h_chr[k] : mc; }
    pb = bits_of((unsigned long long)mp); cb = bits_of((unsigned long long)(2 * mc + 1));
}

// the sort index of a whole batch through the device code (hp_sort.h) under the lane emulation
static void emu_sort_index(const BatchIn &in, int n_reads, std::vector<int32_t> &srt, std::vector<int32_t> &rnk)
{
    const int64_t n_hits = n_reads ? in.hit_off[in.seed_off[n_reads]] : 0;
    srt.assign((size_t)n_hits + 1, 0); rnk.assign((size_t)n_hits + 1, 0);
    std::vector<uint64_t> keys((size_t)n_hits + 1, 0);
    static thread_local uint64_t lw[HP_BOTH_LDS_WORDS / 2];
    int pb, cb; emu_sort_widths(in, n_reads, pb, cb);
    for (int r = 0; r < n_reads; ++r) {
        const int64_t hb = in.hit_off[in.seed_off[r]]; const int H = (int)(in.hit_off[in.seed_off[r + 1]] - hb);
        sort_read_hits(in.h_pos + hb, in.h_chr + hb, in.h_strand + hb, H, srt.data() + hb, rnk.data() + hb, keys.data() + hb, lw, HP_BOTH_LDS_WORDS / 2, pb, cb);
    }
}

// checker: the same index by std::stable_sort; returns the number of entries that differ
extern "C" int64_t emu_sort_check(int n_reads, const int64_t *seed_off, const int64_t *hit_off, const int64_t *h_pos, const int32_t *h_chr, const int8_t *h_strand)
{
    BatchIn in; memset(&in, 0, sizeof in);
    in.n_reads = n_reads; in.seed_off = seed_off; in.hit_off = hit_off; in.h_pos = h_pos; in.h_chr = h_chr; in.h_strand = h_strand;
    std::vector<int32_t> srt, rnk;
    emu_sort_index(in, n_reads, srt, rnk);
    int64_t bad = 0;
    for (int r = 0; r < n_reads; ++r) {
        const int64_t hb = hit_off[seed_off[r]], he = hit_off[seed_off[r + 1]];
        const int H = (int)(he - hb);
        std::vector<int> idx((size_t)H);
        for (int k = 0; k < H; ++k) idx[k] = k;
        auto key = [&](int k) { return ((uint64_t)((uint32_t)h_chr[hb + k] * 2u + (h_strand[hb + k] > 0 ? 1u : 0u)) << 40) | ((uint64_t)h_pos[hb + k] & ((1ull << 40) - 1)); };
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return key(a) < key(b); });
        for (int i = 0; i < H; ++i) bad += (srt[hb + i] != idx[i]) + (rnk[hb + idx[i]] != i);
    }
    return bad;
}

// 1 (default): scale-1 batches take the phased path of hp_phase.h, like the product's main pass; 0: the one-kernel path
static int g_emu_phased = 1, g_emu_unit_cap = 0, g_emu_lane_dp = 1;
static long long g_emu_job_words = 0;
extern "C" long long emu_last_job_words() { return g_emu_job_words; }      // CIGAR words the lane-per-job DP left for the fill in the last batch
extern "C" void emu_set_lane_dp(int on) { g_emu_lane_dp = on; }           // 0: the fill runs every DP itself (one job per wave)
extern "C" void emu_set_phased(int on) { g_emu_phased = on; }
extern "C" void emu_set_cl_cap(int cap) { g_emu_cl_cap = cap > 0 ? cap : (cap < 0 ? 0 : 1 << 30); }      // < 0: no clusters at all (the whole-read HBM paths)
extern "C" void emu_set_gap_caps(int tab_cap, int mcap) { g_emu_gaptab_cap = tab_cap > 0 ? tab_cap : (tab_cap < 0 ? 0 : 1 << 30); g_emu_gap_mcap = mcap > 0 ? mcap : (mcap < 0 ? 0 : 1 << 30); }
extern "C" void emu_set_wave_jobs(int on) { g_emu_wave_jobs = on; }
extern "C" void emu_set_wj_small(int bytes) { g_emu_wj_small = bytes; }
extern "C" void emu_set_pk(int on) { g_emu_pk = on; }
extern "C" void emu_set_frag_block_min(int n) { g_emu_frag_block_min = n > 0 ? n : 3; }
extern "C" long long emu_stat(int i) { return g_emu_stat[i & 31]; }
extern "C" void emu_dplog_on(int on) { g_emu_dplog_on = on; g_emu_dplog.clear(); }
extern "C" long long emu_dplog(long long *buf, long long cap) { long long n = (long long)g_emu_dplog.size(); for (long long i = 0; i < n && i < cap; ++i) buf[i] = g_emu_dplog[i]; return n; }
extern "C" void emu_stat_reset() { memset(g_emu_stat, 0, sizeof g_emu_stat); }
extern "C" void emu_set_unit_cap(int cap) { g_emu_unit_cap = cap; }      // tests: force the "too many lines" overflow

extern "C" int emu_align_batch(const lamsa_hp_para *P, const lamsa_hp_ref *ref, const lamsa_hp_batch *B, int scale, size_t slab_bytes,
                               int32_t *stream, int64_t stream_cap, int64_t *n_words, int64_t *read_off, int32_t *read_len, int32_t *status)
{
    AlignArgs a;
    a.P = *P;
    a.ref.pac = ref->pac; a.ref.l_pac = ref->l_pac; a.ref.n_seqs = ref->n_seqs; a.ref.seq_off = ref->seq_offset; a.ref.seq_len = ref->seq_len;
    a.in.n_reads = B->n_reads; a.in.read_skip = nullptr; a.in.read_off = B->read_off; a.in.read_seq = B->read_seq; a.in.seed_all = B->seed_all; a.in.last_len = B->last_len;
    a.in.seed_off = B->seed_off; a.in.seed_id = B->seed_id; a.in.hit_off = B->hit_off; a.in.h_pos = B->h_pos; a.in.h_chr = B->h_chr;
    // the boundary's two CIGAR forms -> what the kernels read (words, 64-bit offsets), as the product does on the device
    const int64_t n_hits_all = B->n_reads ? B->hit_off[B->seed_off[B->n_reads]] : 0;
    std::vector<int64_t> off64((size_t)n_hits_all + 1, 0); std::vector<int32_t> words;
    { int64_t run = 0; for (int64_t k = 0; k < n_hits_all; ++k) { off64[k] = B->h_cig_off ? (int64_t)B->h_cig_off[k] : run; run += B->h_cig_n[k]; } }
    if (B->cig8) { words.resize((size_t)B->n_cig + 1); for (int64_t i = 0; i < B->n_cig; ++i) words[i] = ((B->cig8[i] & 63) << 4) | (B->cig8[i] >> 6); }
    a.in.h_cig_off = off64.data(); a.in.h_nm = B->h_nm; a.in.h_len_dif = B->h_len_dif; a.in.h_strand = B->h_strand; a.in.h_cig_n = B->h_cig_n; a.in.cig = B->cig8 ? words.data() : B->cig;
    emu_sort_widths(a.in, B->n_reads, a.sort_pb, a.sort_cb);
    unsigned long long cursor = 0;
    a.out.stream = stream; a.out.stream_cap = stream_cap; a.out.cursor = &cursor;
    a.out.read_out_off = read_off; a.out.read_out_len = read_len; a.out.read_status = status; a.out.read_tbases = nullptr; a.out.read_work = nullptr; a.out.diag = nullptr;
    std::vector<char> slab(slab_bytes);
    static thread_local int32_t lds[HP_BOTH_LDS_WORDS];
    if (scale == 1 && g_emu_phased) {
        // the product's main pass: chain1 -> fill -> chain2 -> fill -> publish, every phase over the whole batch before the next starts
        PhaseArgs p;
        p.P = a.P; p.ref = a.ref; p.in = a.in; p.out = a.out; p.slab = slab.data(); p.slab_per_wave = slab_bytes; p.slab_fill = slab_bytes; p.slab_wj = g_emu_wj_small > 0 ? (size_t)g_emu_wj_small : slab_bytes; p.slab_wjb = slab_bytes; p.wjb_off = 0; p.n_wjb = 1;
        p.sort_pb = a.sort_pb; p.sort_cb = a.sort_cb; p.order = nullptr; p.n_reads = B->n_reads; p.prof = nullptr;
        const int n = B->n_reads;
        const int64_t n_hits = n ? B->hit_off[B->seed_off[n]] : 0, n_bases = n ? B->read_off[n] : 0;
        std::vector<NodeS> nd((size_t)(n_hits + n) + 1); std::vector<int32_t> nseed((size_t)(n_hits + n) + 1), sidx(2 * (size_t)(n_hits + n) + 2);
        std::vector<RdMeta> meta((size_t)n + 1); memset(meta.data(), 0, sizeof(RdMeta) * meta.size());
        p.unit_cap = g_emu_unit_cap > 0 ? g_emu_unit_cap : 8 * n + 64;
        std::vector<UnitRec> units(2 * (size_t)p.unit_cap); std::vector<int32_t> bq(2 * (size_t)PH_NBUCKET * p.unit_cap);
        p.fl_cap = 16 * (n_hits + n) + 1024 * (int64_t)n + 1024; p.line_cap = stream_cap + 16 * 2 * (int64_t)p.unit_cap;
        p.fl_cap = 32 * (n_hits + n) + 4096 * (int64_t)n + 4096;
        p.job_cap = (g_emu_lane_dp || g_emu_wave_jobs) ? 4096 + 1024 * (int64_t)n + 8 * n_bases : 0;
        std::vector<int32_t> fl((size_t)p.fl_cap), lines((size_t)p.line_cap), jobsv((size_t)p.job_cap + 4);
        p.job_base = jobsv.data();
        p.lj_cap = g_emu_lane_dp ? (int)(1024 + 64 * (int64_t)n + n_bases / 8) : 0;
        std::vector<LjRec> ljv((size_t)p.lj_cap + 1); std::vector<int32_t> ljq((size_t)LJ_NBUCKET * p.lj_cap + 1);
        p.ljobs = ljv.data(); p.lj_bucket = ljq.data();
        p.wj_cap = g_emu_wave_jobs ? (int)(1024 + 64 * (int64_t)n + n_bases / 8) : 0;
        std::vector<WjRec> wjv((size_t)p.wj_cap + 1); std::vector<int32_t> wjq((size_t)WJ_NBUCKET * p.wj_cap + 1);
        p.wjobs = wjv.data(); p.wj_bucket = wjq.data();
        static thread_local int32_t lds_lj[HP_LJ_LDS_WORDS(HP_LJ_QCAP)];
        static thread_local int32_t lds_wj[HP_WJ_LDS_WORDS];
        PhaseCtl ctl; memset(&ctl, 0, sizeof ctl);
        p.g_nd = nd.data(); p.g_nseed = nseed.data(); p.g_sidx = sidx.data(); p.meta = meta.data(); p.units = units.data(); p.bucket_q = bq.data();
        p.fl_base = fl.data(); p.line_base = lines.data(); p.ctl = &ctl;
        (void)n_bases;
        auto fill_all = [&](int round) {
            if (g_emu_lane_dp || g_emu_wave_jobs) {
                for (int b = 0; b < PH_NBUCKET; ++b)
                    for (int i = 0; i < ctl.bucket_n[round][b]; ++i) phase_filllist(p, round, bq[((size_t)round * PH_NBUCKET + b) * p.unit_cap + i], 0, lds);
                int nw = 0, nwb = 0;
                for (int b = 0; b < WJ_NBUCKET; ++b) { const int k = ctl.wj_bucket_n[round][b] < p.wj_cap ? ctl.wj_bucket_n[round][b] : p.wj_cap; if (b < WJ_NBIG) nwb += k; else nw += k; }
                for (int g = 0; g < nwb; ++g) { phase_wavejob(p, round, g, true, 0, lds_wj); ++g_emu_stat[23]; }      // the jobs that need a big slab
                for (int g = 0; g < nw; ++g) phase_wavejob(p, round, g, false, 0, lds_wj);
                for (int b = 0; b < LJ_NBUCKET; ++b)
                    for (int off = 0; off < (ctl.lj_bucket_n[round][b] < p.lj_cap ? ctl.lj_bucket_n[round][b] : p.lj_cap); off += 64) phase_filldp(p, round, b, off, 0, lds_lj, HP_LJ_QSMALL);
            }
            for (int b = 0; b < PH_NBUCKET; ++b)
                for (int i = 0; i < ctl.bucket_n[round][b]; ++i) phase_fill(p, round, bq[((size_t)round * PH_NBUCKET + b) * p.unit_cap + i], 0, lds);
        };
        for (int r = 0; r < n; ++r) phase_chain1(p, r, 0, lds);
        fill_all(0);
        for (int r = 0; r < n; ++r) phase_chain2(p, r, 0, lds);
        fill_all(1);
        for (int r = 0; r < n; ++r) phase_publish(p, r);
        g_emu_job_words = (long long)ctl.job_cursor;
        *n_words = (int64_t)cursor;
        return 0;
    }
    a.slab = slab.data(); a.slab_per_wave = slab_bytes; a.counter = nullptr; a.order = nullptr; a.n_units = B->n_reads; a.scale = scale; a.prof = nullptr;
    for (int r = 0; r < B->n_reads; ++r) align_read(a, r, 0, lds);
    *n_words = (int64_t)cursor;
    return 0;
}

// ---------------------------------------------------------------- the k-mer split mapper (hp_split.h) on one read gap / reference window
extern "C" int emu_split_indel_map(const lamsa_hp_para *P, const uint8_t *read, int read_len, const uint8_t *ref, int ref_len, int ref_offset,
                                   int32_t *cig, int cig_cap, int32_t *ret, int32_t *status)
{
    std::vector<char> slab((size_t)64 << 20);
    static thread_local int32_t lds[HP_BOTH_LDS_WORDS];
    Ctx cx; cx.P = P; cx.status = 0; cx.n_cells = 0; cx.lds_epoch = 0; cx.prof = nullptr; cx.lds = lds; cx.lds_words = HP_LDS_WORDS;
    arena_init(cx.tmp, slab.data(), slab.size());
    CigV out; cig_bind(out, cig, cig_cap);
    *ret = split_indel_map(cx, out, read, read_len, ref, ref_len, ref_offset);
    *status = cx.status;
    return out.n;
}

// ---------------------------------------------------------------- a job as the wave-per-job launch runs it (hp_wavejob.h: wj_run with that
// launch's LDS, sequences staged from the read bytes and the packed reference): type 1 = a junction's ksw_bi_extend(h0, h0), 2 = ksw_global2(w),
// 3 / 4 = a line's head / tail extension (both sequences walked backwards for the head; the rest of the query clipped, the head's CIGAR turned round)
extern "C" int emu_wave_job(const lamsa_hp_para *P, int n, const uint8_t *seq, const int64_t *q_off, const int32_t *qlen, const int64_t *t_off, const int32_t *tlen,
                            int type, int w, int h0, size_t slab_bytes, int32_t *score, int32_t *qle, int32_t *tle, int32_t *status, int32_t *cig_n, int32_t *cig, const int64_t *cig_off)
{
    int64_t tot = 0;
    for (int i = 0; i < n; ++i) tot += tlen[i];
    std::vector<uint8_t> pac((size_t)tot / 4 + 8, 0); std::vector<int64_t> tk((size_t)n + 1, 0);
    { int64_t k = 0; for (int i = 0; i < n; ++i) { tk[i] = k; for (int j = 0; j < tlen[i]; ++j, ++k) { if (seq[t_off[i] + j] > 3) return -1; pac[k >> 2] |= (uint8_t)((seq[t_off[i] + j] & 3) << ((~k & 3) << 1)); } } }
    std::vector<char> slab(slab_bytes);
    static thread_local int32_t lds_wj[HP_WJ_LDS_WORDS];
    for (int i = 0; i < n; ++i) {
        Ctx cx; cx.P = P; cx.lds = lds_wj; cx.lds_words = HP_WJ_LDS_WORDS; cx.status = 0; cx.n_cells = 0; cx.lds_epoch = 0; cx.prof = nullptr;
        arena_init(cx.tmp, slab.data(), slab.size());
        const bool back = type == WJ_HEAD;
        CigV out; cig_bind(out, cig + cig_off[i], (int)(cig_off[i + 1] - cig_off[i]));
        WjOut o;
        wj_run(cx, seq, pac.data(), type, 0, q_off[i] + (back && qlen[i] > 0 ? qlen[i] - 1 : 0), back ? -1 : 1, qlen[i], tk[i] + (back && tlen[i] > 0 ? tlen[i] - 1 : 0), back ? -1 : 1, tlen[i], w, h0, out, o);
        score[i] = o.score; qle[i] = o.qle; tle[i] = o.tle; status[i] = cx.status; cig_n[i] = out.n;
    }
    return 0;
}

// ---------------------------------------------------------------- the lane-per-job DP routines (hp_lanedp.h) on explicit jobs
// kind 0: ksw_global2(w), 1: ksw_extend_core(w, h0), 2: ksw_bi_extend(h0, h0).  Targets are packed 2 bits per base first, as the
// lanes read them from the reference; 64 jobs per group like the kernel.
extern "C" int emu_lane_dp(const lamsa_hp_para *P, int n, const uint8_t *seq, const int64_t *q_off, const int32_t *qlen, const int64_t *t_off, const int32_t *tlen,
                           int kind, int w, int h0, int32_t *score, int32_t *qle, int32_t *tle, int32_t *cig_n, int32_t *cig /* n * HP_LJ_CIG words */)
{
    int64_t tot = 0;
    for (int i = 0; i < n; ++i) tot += tlen[i];
    std::vector<uint8_t> pac((size_t)tot / 4 + 8, 0); std::vector<int64_t> tk((size_t)n + 1, 0);
    { int64_t k = 0; for (int i = 0; i < n; ++i) { tk[i] = k; for (int j = 0; j < tlen[i]; ++j, ++k) pac[k >> 2] |= (uint8_t)((seq[t_off[i] + j] & 3) << ((~k & 3) << 1)); } }
    std::vector<uint8_t> z((size_t)HP_LJ_QCAP * HP_LJ_TCAP * 64 + 64);
    std::vector<cig_t> cb((size_t)3 * HP_LJ_CIG * 64);
    static thread_local int32_t lds_lj[HP_LJ_LDS_WORDS(HP_LJ_QCAP)];
    if (!lj_params_ok(P)) return -2;
    for (int j0 = 0; j0 < n; j0 += 64) {
        for (int l = 0; l < 64 && j0 + l < n; ++l) {
            const int i = j0 + l;
            if (qlen[i] > HP_LJ_QCAP || tlen[i] > HP_LJ_TCAP) return -1;
            LaneJob J; J.q = seq + q_off[i]; J.qs = 1; J.qcomp = 0; J.qlen = qlen[i]; J.pac = pac.data(); J.tk = tk[i]; J.ts = 1; J.tlen = tlen[i]; J.z = z.data(); J.zl = l; J.zs = HP_LJ_QCAP; J.cells = 0;
            J.row = lds_lj + l; J.qrow = (uint8_t *)(lds_lj + (HP_LJ_QCAP + 2) * 64) + l; J.rev = 0;
            lj_stage_query(J);
            LCig out, Lc, Rc; out.c = cb.data() + (size_t)l * 3 * HP_LJ_CIG; out.n = 0; Lc.c = out.c + HP_LJ_CIG; Lc.n = 0; Rc.c = Lc.c + HP_LJ_CIG; Rc.n = 0;
            int a = 0, b = 0;
            if (kind == 0) score[i] = lj_global(P, J, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, w, &out);
            else if (kind == 1) score[i] = lj_extend(P, J, w, h0, &a, &b, &out);
            else score[i] = lj_bi_extend(P, J, h0, h0, Lc, Rc, out);
            qle[i] = a; tle[i] = b; cig_n[i] = out.n;
            memcpy(cig + (size_t)i * HP_LJ_CIG, out.c, sizeof(cig_t) * (size_t)out.n);
        }
    }
    return 0;
}
