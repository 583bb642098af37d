// hp_align_api.hip -- lamsa_hp_align_batch / upload_batch / run_uploaded (include/lamsa_hp.h):
// the batch form of the reference's per-read worker (src/lamsa_aln.c:857-871) on gfx950.
// One wavefront (= one 64-thread workgroup) per read, a persistent grid pulling reads from a
// queue head in costliest-first order, every wave with its own scratch slab in HBM; no
// collectives, no inter-workgroup communication except two atomics per read.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <map>
#include "hp_align.h"
#include "hp_handle.h"
#include "hp_hostprep.h"

using namespace hp;

#ifndef HP_WAVES_PER_SIMD
#define HP_WAVES_PER_SIMD 4          // final kernel, default workload: 3 -> 314 ms, 4 -> 280, 5 -> 270 (but 70 % more HBM traffic), 6 -> 295
#endif
__global__ __launch_bounds__(64, HP_WAVES_PER_SIMD) void k_align_batch(AlignArgs a)
{
    __shared__ int32_t lds[HP_LDS_WORDS];            // this wave's DP rows, query window and direction matrix (hp_ksw.h)
    const int slot = blockIdx.x;
    for (;;) {
        int u = 0;
        if (wv::leader()) u = atomicAdd(a.counter, 1);
        u = wv::uni(u);
        if (u >= a.n_units) break;          // every wave reaches this exit: the queue head only grows
        align_read(a, a.order ? a.order[u] : u, slot, (HP_L int32_t *)lds);
    }
}

struct OutDev {                   // result arrays of one launch: per-read offset / length / status + the stream arena
    DevBuf buf; int64_t stream_cap = 0;
    static size_t hdr(int n) { return al256(8 * (size_t)n) + 3 * al256(4 * (size_t)n); }
    int ensure(int n, int64_t cap) { stream_cap = cap; return buf.ensure(hdr(n) + 4 * (size_t)cap + 256); }
    int64_t *off() const { return (int64_t *)buf.p; }
    int32_t *len(int n) const { return (int32_t *)((char *)buf.p + al256(8 * (size_t)n)); }
    int32_t *st(int n) const { return (int32_t *)((char *)buf.p + al256(8 * (size_t)n) + al256(4 * (size_t)n)); }
    int32_t *tb(int n) const { return (int32_t *)((char *)buf.p + al256(8 * (size_t)n) + 2 * al256(4 * (size_t)n)); }
    int32_t *stream(int n) const { return (int32_t *)((char *)buf.p + hdr(n)); }
};

struct AlignState {
    DevBuf bin, slab, misc, retry_list;
    OutDev out1, out2;
    // the resident batch
    bool valid = false;
    int32_t n_reads = 0; int64_t n_bases = 0, n_cig = 0;
    BatchIn in; const int32_t *d_order = nullptr;
    std::vector<int32_t> order, h_len, h_H;
    int32_t max_L = 0, max_H = 0;
    // host copies of the results
    std::vector<int32_t> stream, r_len, r_st, r_tb; std::vector<int64_t> r_off;
};

static std::map<lamsa_hp_handle *, AlignState *> g_states;     // per-handle state of the align entry points
static AlignState *state_of(lamsa_hp_handle *h)
{
    auto it = g_states.find(h);
    if (it != g_states.end()) return it->second;
    return g_states[h] = new AlignState();
}
extern "C" void lamsa_hp_release_state_(lamsa_hp_handle *h)
{
    auto it = g_states.find(h);
    if (it == g_states.end()) return;
    AlignState *S = it->second;
    S->bin.release(); S->slab.release(); S->misc.release(); S->retry_list.release(); S->out1.buf.release(); S->out2.buf.release();
    delete S;
    g_states.erase(it);
}

static size_t slab_bytes_for(const lamsa_hp_para &P, int L, int H, int scale)
{   // per-wave scratch: node arrays + line sets (~400 B/hit), result + CIGAR buffers (~128 B/base), and the
    // direction matrix of the largest extension: (2w+1) columns x (L + 2*hash_step) rows.  Reads that need more
    // flag LAMSA_HP_ST_OVERFLOW and are re-run by the retry pass with `scale` = 8.
    const size_t z = (2 * (size_t)P.band_w + 128) * ((size_t)L + 256);
    return al256(((size_t)256 << 10) + (size_t)scale * 128 * (size_t)L + 400 * (size_t)H + z * (size_t)(scale > 1 ? 4 : 1));
}

extern "C" int lamsa_hp_upload_batch(lamsa_hp_handle *h, const lamsa_hp_batch *B)
{
    if (!h || !B || B->n_reads < 0) return LAMSA_HP_EINVAL;
    if (!h->d_pac) { h->err = "handle was created without a reference"; return LAMSA_HP_EINVAL; }
    HIPCHK(h, hipSetDevice(h->device), LAMSA_HP_ENODEV);
    AlignState *S = state_of(h);
    S->valid = false;
    const int n = B->n_reads;
    const int64_t n_slots = n ? B->seed_off[n] : 0, n_hits = n_slots ? B->hit_off[n_slots] : 0, n_bases = n ? B->read_off[n] : 0;
    // ---- validate everything the kernels index with, on the host, before anything is launched
    S->h_len.assign((size_t)n, 0); S->h_H.assign((size_t)n, 0); S->max_L = 0; S->max_H = 0;
    if (n && (B->seed_off[0] != 0 || B->read_off[0] != 0 || (n_slots && B->hit_off[0] != 0))) { h->err = "offsets must start at 0"; return LAMSA_HP_EINVAL; }
    {
        std::atomic<int> bad(0);                         // 1..6: which check failed (the first one reported wins)
        hp_parallel_blocks(n, [&](int r0, int r1) {
            for (int r = r0; r < r1 && !bad.load(std::memory_order_relaxed); ++r) {
                const int64_t L = B->read_off[r + 1] - B->read_off[r], ns = B->seed_off[r + 1] - B->seed_off[r];
                if (L < 0 || L > (1 << 24) || ns < 0 || ns > HP_MAX_SLOTS) { bad = 1; return; }
                int64_t H = 0;
                for (int64_t s = B->seed_off[r]; s < B->seed_off[r + 1]; ++s) {
                    const int64_t m = B->hit_off[s + 1] - B->hit_off[s];
                    if (m < 0 || m > HP_MAX_HITS_PER_SEED) { bad = 2; return; }
                    if (B->seed_id[s] < 1 || B->seed_id[s] > B->seed_all[r] || (s > B->seed_off[r] && B->seed_id[s] <= B->seed_id[s - 1])) { bad = 3; return; }
                    H += m;
                }
                if (H > (1 << 22)) { bad = 4; return; }
                for (int64_t i = B->read_off[r]; i < B->read_off[r + 1]; ++i) if (B->read_seq[i] > 4) { bad = 5; return; }
                for (int64_t k = B->hit_off[B->seed_off[r]]; k < B->hit_off[B->seed_off[r + 1]]; ++k)
                    if (B->h_chr[k] < 1 || B->h_chr[k] > h->n_seqs || (B->h_strand[k] != 1 && B->h_strand[k] != -1) ||
                        B->h_cig_off[k] < 0 || (int64_t)B->h_cig_off[k] + B->h_cig_n[k] > B->n_cig) { bad = 6; return; }
                S->h_len[r] = (int32_t)L; S->h_H[r] = (int32_t)H;
            }
        });
        static const char *why[] = {"", "read too long / too many seeds", "too many hits in one seed", "seed ids must be ascending in [1, seed_all]",
                                    "too many hits in one read", "read base code > 4", "bad hit record"};
        if (bad) { h->err = why[bad.load()]; return LAMSA_HP_EINVAL; }
        for (int r = 0; r < n; ++r) { S->max_L = std::max(S->max_L, S->h_len[r]); S->max_H = std::max(S->max_H, S->h_H[r]); }
    }
    // ---- processing order: costliest first (chaining ~ H^2/64 lane steps, extension ~ L * band)
    S->order.resize((size_t)n);
    for (int r = 0; r < n; ++r) S->order[r] = r;
    {
        std::vector<double> cost((size_t)n);
        for (int r = 0; r < n; ++r) cost[r] = (double)S->h_H[r] * S->h_H[r] / 64.0 + (double)S->h_len[r] * (2.0 * h->para.band_w + 1) / 16.0;
        std::stable_sort(S->order.begin(), S->order.end(), [&](int x, int y) { return cost[x] > cost[y]; });
    }
    // ---- one packed upload
    size_t off = 0;
    auto place = [&](size_t bytes) { size_t o = off; off = al256(off + bytes + 16); return o; };
    const size_t o_roff = place(8 * ((size_t)n + 1)), o_rseq = place((size_t)n_bases), o_sall = place(4 * (size_t)n), o_last = place(4 * (size_t)n),
                 o_soff = place(8 * ((size_t)n + 1)), o_sid = place(4 * (size_t)n_slots), o_hoff = place(8 * ((size_t)n_slots + 1)),
                 o_pos = place(8 * (size_t)n_hits), o_chr = place(4 * (size_t)n_hits), o_coff = place(4 * (size_t)n_hits), o_nm = place(2 * (size_t)n_hits),
                 o_ld = place(2 * (size_t)n_hits), o_st = place((size_t)n_hits), o_cn = place((size_t)n_hits), o_cig = place(4 * (size_t)B->n_cig), o_ord = place(4 * (size_t)n),
                 o_srt = place(4 * (size_t)n_hits), o_rnk = place(4 * (size_t)n_hits);
    std::vector<int32_t> srt, rnk;
    hp_build_sort_index(n, B->seed_off, B->hit_off, B->h_pos, B->h_chr, B->h_strand, srt, rnk);
    if (S->bin.ensure(off)) { h->err = "hipMalloc(batch)"; return LAMSA_HP_ENOMEM; }
    char *d = (char *)S->bin.p;
    hipStream_t s = h->stream;
    static const int64_t zero64 = 0;
#define UP(o, src, bytes) do { if ((bytes) > 0) HIPCHK(h, hipMemcpyAsync(d + (o), (src), (bytes), hipMemcpyHostToDevice, s), LAMSA_HP_EKERNEL); } while (0)
    if (n) { UP(o_roff, B->read_off, 8 * ((size_t)n + 1)); UP(o_soff, B->seed_off, 8 * ((size_t)n + 1)); }
    else { UP(o_roff, &zero64, 8); UP(o_soff, &zero64, 8); }
    UP(o_rseq, B->read_seq, (size_t)n_bases); UP(o_sall, B->seed_all, 4 * (size_t)n); UP(o_last, B->last_len, 4 * (size_t)n);
    UP(o_sid, B->seed_id, 4 * (size_t)n_slots);
    if (n_slots) UP(o_hoff, B->hit_off, 8 * ((size_t)n_slots + 1)); else UP(o_hoff, &zero64, 8);
    UP(o_pos, B->h_pos, 8 * (size_t)n_hits); UP(o_chr, B->h_chr, 4 * (size_t)n_hits); UP(o_coff, B->h_cig_off, 4 * (size_t)n_hits);
    UP(o_nm, B->h_nm, 2 * (size_t)n_hits); UP(o_ld, B->h_len_dif, 2 * (size_t)n_hits); UP(o_st, B->h_strand, (size_t)n_hits); UP(o_cn, B->h_cig_n, (size_t)n_hits);
    UP(o_cig, B->cig, 4 * (size_t)B->n_cig); UP(o_ord, S->order.data(), 4 * (size_t)n);
    UP(o_srt, srt.data(), 4 * (size_t)n_hits); UP(o_rnk, rnk.data(), 4 * (size_t)n_hits);
#undef UP
    HIPCHK(h, hipStreamSynchronize(s), LAMSA_HP_EKERNEL);
    BatchIn &in = S->in;
    in.n_reads = n; in.read_off = (const int64_t *)(d + o_roff); in.read_seq = (const uint8_t *)(d + o_rseq);
    in.seed_all = (const int32_t *)(d + o_sall); in.last_len = (const int32_t *)(d + o_last); in.seed_off = (const int64_t *)(d + o_soff);
    in.seed_id = (const int32_t *)(d + o_sid); in.hit_off = (const int64_t *)(d + o_hoff); in.h_pos = (const int64_t *)(d + o_pos);
    in.h_chr = (const int32_t *)(d + o_chr); in.h_cig_off = (const int32_t *)(d + o_coff); in.h_nm = (const int16_t *)(d + o_nm);
    in.h_len_dif = (const int16_t *)(d + o_ld); in.h_strand = (const int8_t *)(d + o_st); in.h_cig_n = (const uint8_t *)(d + o_cn);
    in.cig = (const int32_t *)(d + o_cig);
    in.h_sort = (const int32_t *)(d + o_srt); in.h_rank = (const int32_t *)(d + o_rnk);
    S->d_order = (const int32_t *)(d + o_ord);
    S->n_reads = n; S->n_bases = n_bases; S->n_cig = B->n_cig;
    S->valid = true;
    return LAMSA_HP_OK;
}

// one launch over `n_units` reads (order list on the device); results into `O`
static int launch_align(lamsa_hp_handle *h, AlignState *S, OutDev &O, const int32_t *d_order, int n_units, int scale, int max_L, int max_H, float *ms)
{
    size_t slab_per_wave = slab_bytes_for(h->para, max_L, max_H, scale);
    if (scale == 1 && h->scratch_limit && slab_per_wave > h->scratch_limit) slab_per_wave = al256(h->scratch_limit);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_align_batch, 64, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    int n_waves = h->n_cu * per_cu;
    if (n_waves > n_units) n_waves = n_units;
    while (n_waves > 1 && slab_per_wave * (size_t)n_waves > ((size_t)160 << 30)) n_waves /= 2;
    if (S->slab.ensure(slab_per_wave * (size_t)n_waves) || S->misc.ensure(256)) { h->err = "hipMalloc(slab)"; return LAMSA_HP_ENOMEM; }
    const int n = S->n_reads;
    AlignArgs a;
    a.P = h->para;
    a.ref.pac = h->d_pac; a.ref.l_pac = h->l_pac; a.ref.n_seqs = h->n_seqs; a.ref.seq_off = h->d_seq_off; a.ref.seq_len = h->d_seq_len;
    a.in = S->in;
    a.out.read_out_off = O.off(); a.out.read_out_len = O.len(n); a.out.read_status = O.st(n); a.out.read_tbases = O.tb(n); a.out.stream = O.stream(n);
    a.out.stream_cap = O.stream_cap; a.out.cursor = (unsigned long long *)((char *)S->misc.p + 64);
    a.slab = (char *)S->slab.p; a.slab_per_wave = slab_per_wave; a.counter = (int32_t *)S->misc.p;
    a.order = d_order; a.n_units = n_units; a.scale = scale; a.prof = nullptr;
#ifdef HP_PROF
    static DevBuf profbuf;
    if (profbuf.ensure(sizeof(long long) * 64 * (size_t)n + 64) == 0) { hipMemset(profbuf.p, 0, sizeof(long long) * 64 * (size_t)n); a.prof = (long long *)profbuf.p; }
#endif
    hipStream_t s = h->stream;
    HIPCHK(h, hipMemsetAsync(S->misc.p, 0, 128, s), LAMSA_HP_EKERNEL);
    HIPCHK(h, hipEventRecord(h->ev0, s), LAMSA_HP_EKERNEL);
    hipLaunchKernelGGL(k_align_batch, dim3(n_waves), dim3(64), 0, s, a);
    HIPCHK(h, hipGetLastError(), LAMSA_HP_EKERNEL);
    HIPCHK(h, hipEventRecord(h->ev1, s), LAMSA_HP_EKERNEL);
    HIPCHK(h, hipStreamSynchronize(s), LAMSA_HP_EKERNEL);
    if (ms) hipEventElapsedTime(ms, h->ev0, h->ev1);
#ifdef HP_PROF
    if (a.prof) {
        std::vector<long long> pr((size_t)n * 64);
        hipMemcpy(pr.data(), a.prof, sizeof(long long) * pr.size(), hipMemcpyDeviceToHost);
        std::vector<int> idx((size_t)n); for (int i = 0; i < n; ++i) idx[i] = i;
        auto tot = [&](int r) { long long t = 0; for (int k = 0; k < 6; ++k) t += pr[(size_t)r * 64 + k]; return t; };
        std::sort(idx.begin(), idx.end(), [&](int x, int y) { return tot(x) > tot(y); });
        long long sum[64] = {0}; for (int r = 0; r < n; ++r) for (int k = 0; k < 64; ++k) sum[k] += pr[(size_t)r * 64 + k];
        fprintf(stderr, "[HP_PROF] cycles: setup chain1 fill1 chain2 fill2 publish | in chain1: init+minext mainscan track pop-loop bound+flines | o_l H\n");
        fprintf(stderr, "[HP_PROF] SUM  "); for (int k = 0; k < 11; ++k) fprintf(stderr, " %lld", sum[k] / 1000000); fprintf(stderr, " (Mcycles) targets %lld trips %lld init_Mcyc %lld\n", sum[11], sum[12], sum[13] / 1000000);
        fprintf(stderr, "[HP_PROF] update_range: calls %lld total %lld prefilter %lld prologue %lld (Mcyc) chunks %lld | mini_line calls %lld tail+walk %lld Mcyc | forced %lld\n", sum[19], sum[16] / 1000000, sum[17] / 1000000, sum[18] / 1000000, sum[22], sum[21], sum[20] / 1000000, sum[23]);
        { const char *nm[] = {"ksw_global", "ksw_extend", "backtrack", "ref_fetch", "head_fix", "frag_extend", "split_mapping", "tail_fix", "res_split", "res_aux", "mini_line_regs", "(calls = fallbacks)"};
          for (int k = 0; k < 12; ++k) fprintf(stderr, "[HP_PROF] %-16s %8lld Mcyc %10lld calls\n", nm[k], sum[24 + 2 * k] / 1000000, sum[25 + 2 * k]);
          fprintf(stderr, "[HP_PROF] direction-matrix bytes in HBM %lld (%lld jobs); hits passed through nodes_per_init %lld (%lld calls)\n", sum[52], sum[53], sum[54], sum[55]);
          fprintf(stderr, "[HP_PROF] chain_first start: seed loop %lld, node_set loop %lld, min_extend %lld Mcyc (%lld reads with a MIN pass)\n", sum[56] / 1000000, sum[57] / 1000000, sum[58] / 1000000, sum[59]);
          fprintf(stderr, "[HP_PROF] query <= 62: ksw_extend %lld Mcyc %lld calls, ksw_global %lld Mcyc %lld calls\n", sum[60] / 1000000, sum[61], sum[62] / 1000000, sum[63]);
          const char *bn[] = {"[bi_extend total]", "[bi_extend after left ext]", "17-32", "33-64", "65-128", "129-256", "257-512", ">512"};
          for (int k = 0; k < 2; ++k) fprintf(stderr, "[HP_PROF] ksw_extend qlen %-8s %8lld Mcyc %10lld calls\n", bn[k], sum[48 + 2 * k] / 1000000, sum[49 + 2 * k]); }
        for (int q = 0; q < 8 && q < n; ++q) { int r = idx[q]; fprintf(stderr, "[HP_PROF] read %d L=%d:", r, S->h_len[r]); for (int k = 0; k < 11; ++k) fprintf(stderr, " %lld", pr[(size_t)r * 64 + k] / 1000000); fprintf(stderr, " | o_l %lld H %lld | targets %lld trips %lld init_Mcyc %lld\n", pr[(size_t)r * 64 + 14], pr[(size_t)r * 64 + 15], pr[(size_t)r * 64 + 11], pr[(size_t)r * 64 + 12], pr[(size_t)r * 64 + 13] / 1000000); }
    }
#endif
    return LAMSA_HP_OK;
}

extern "C" int lamsa_hp_run_uploaded(lamsa_hp_handle *h, lamsa_hp_result *R)
{
    if (!h) return LAMSA_HP_EINVAL;
    AlignState *S = state_of(h);
    if (!S->valid) { h->err = "no batch uploaded"; return LAMSA_HP_EINVAL; }
    HIPCHK(h, hipSetDevice(h->device), LAMSA_HP_ENODEV);
    const int n = S->n_reads;
    h->kernel_ms[0] = h->kernel_ms[1] = 0;
    S->r_st.assign((size_t)n + 1, 0); S->r_off.assign((size_t)n + 1, 0); S->r_len.assign((size_t)n + 1, 0); S->r_tb.assign((size_t)n + 1, 0);   // never empty: pointers stay valid
    if (n == 0) {
        S->stream.assign(4, 0);
        if (R) { R->stream = S->stream.data(); R->stream_words = 0; R->read_off = S->r_off.data(); R->read_len = S->r_len.data(); R->read_status = S->r_st.data(); R->read_tbases = S->r_tb.data(); }
        return LAMSA_HP_OK;
    }
    // ---- main pass: every read, costliest first
    if (S->out1.ensure(n, 1024 + (int64_t)n * 256 + 4 * S->n_bases)) { h->err = "hipMalloc(out)"; return LAMSA_HP_ENOMEM; }
    int rc = launch_align(h, S, S->out1, S->d_order, n, 1, S->max_L, S->max_H, &h->kernel_ms[0]);
    if (rc) return rc;
    unsigned long long used1 = 0;
    HIPCHK(h, hipMemcpy(&used1, (char *)S->misc.p + 64, 8, hipMemcpyDeviceToHost), LAMSA_HP_EKERNEL);
    if ((int64_t)used1 > S->out1.stream_cap) used1 = (unsigned long long)S->out1.stream_cap;
    HIPCHK(h, hipMemcpy(S->r_st.data(), S->out1.st(n), 4 * (size_t)n, hipMemcpyDeviceToHost), LAMSA_HP_EKERNEL);
    HIPCHK(h, hipMemcpy(S->r_off.data(), S->out1.off(), 8 * (size_t)n, hipMemcpyDeviceToHost), LAMSA_HP_EKERNEL);
    HIPCHK(h, hipMemcpy(S->r_len.data(), S->out1.len(n), 4 * (size_t)n, hipMemcpyDeviceToHost), LAMSA_HP_EKERNEL);
    HIPCHK(h, hipMemcpy(S->r_tb.data(), S->out1.tb(n), 4 * (size_t)n, hipMemcpyDeviceToHost), LAMSA_HP_EKERNEL);
    // ---- retry pass: reads whose work buffers (or the stream arena) were too small -- outliers; 8x capacities
    std::vector<int32_t> again;
    for (int r = 0; r < n; ++r) if ((S->r_st[r] & LAMSA_HP_ST_OVERFLOW) || S->r_off[r] < 0) again.push_back(r);
    unsigned long long used2 = 0;
    if (!again.empty()) {
        int mL = 0, mH = 0; int64_t cap2 = 1024;
        for (int r : again) { mL = std::max(mL, S->h_len[r]); mH = std::max(mH, S->h_H[r]); cap2 += 64 + 12LL * 8 * S->h_len[r]; }
        if (S->out2.ensure(n, cap2) || S->retry_list.ensure(4 * again.size())) { h->err = "hipMalloc(retry)"; return LAMSA_HP_ENOMEM; }
        HIPCHK(h, hipMemcpy(S->retry_list.p, again.data(), 4 * again.size(), hipMemcpyHostToDevice), LAMSA_HP_EKERNEL);
        rc = launch_align(h, S, S->out2, (const int32_t *)S->retry_list.p, (int)again.size(), 8, mL, mH, &h->kernel_ms[1]);
        if (rc) return rc;
        HIPCHK(h, hipMemcpy(&used2, (char *)S->misc.p + 64, 8, hipMemcpyDeviceToHost), LAMSA_HP_EKERNEL);
        if ((int64_t)used2 > cap2) used2 = (unsigned long long)cap2;
        std::vector<int64_t> off2((size_t)n); std::vector<int32_t> len2((size_t)n), st2((size_t)n), tb2((size_t)n);
        HIPCHK(h, hipMemcpy(st2.data(), S->out2.st(n), 4 * (size_t)n, hipMemcpyDeviceToHost), LAMSA_HP_EKERNEL);
        HIPCHK(h, hipMemcpy(off2.data(), S->out2.off(), 8 * (size_t)n, hipMemcpyDeviceToHost), LAMSA_HP_EKERNEL);
        HIPCHK(h, hipMemcpy(len2.data(), S->out2.len(n), 4 * (size_t)n, hipMemcpyDeviceToHost), LAMSA_HP_EKERNEL);
        HIPCHK(h, hipMemcpy(tb2.data(), S->out2.tb(n), 4 * (size_t)n, hipMemcpyDeviceToHost), LAMSA_HP_EKERNEL);
        for (int r : again) { S->r_st[r] = st2[r]; S->r_len[r] = len2[r]; S->r_tb[r] = tb2[r]; S->r_off[r] = off2[r] < 0 ? -1 : (int64_t)used1 + off2[r]; }
    }
    for (int r = 0; r < n; ++r) if (S->r_off[r] < 0) { S->r_off[r] = 0; S->r_len[r] = 0; S->r_st[r] |= LAMSA_HP_ST_OVERFLOW; }
    if (!R) return LAMSA_HP_OK;
    S->stream.resize((size_t)(used1 + used2) + 4);
    if (used1) HIPCHK(h, hipMemcpy(S->stream.data(), S->out1.stream(n), 4 * (size_t)used1, hipMemcpyDeviceToHost), LAMSA_HP_EKERNEL);
    if (used2) HIPCHK(h, hipMemcpy(S->stream.data() + used1, S->out2.stream(n), 4 * (size_t)used2, hipMemcpyDeviceToHost), LAMSA_HP_EKERNEL);
    R->stream = S->stream.data(); R->stream_words = (int64_t)(used1 + used2); R->read_off = S->r_off.data(); R->read_len = S->r_len.data(); R->read_status = S->r_st.data(); R->read_tbases = S->r_tb.data();
    return LAMSA_HP_OK;
}

extern "C" int lamsa_hp_set_scratch_limit(lamsa_hp_handle *h, size_t bytes)
{
    if (!h || (bytes && bytes < ((size_t)64 << 10))) return LAMSA_HP_EINVAL;
    h->scratch_limit = bytes;
    return LAMSA_HP_OK;
}

extern "C" int lamsa_hp_align_batch(lamsa_hp_handle *h, const lamsa_hp_batch *B, lamsa_hp_result *R)
{
    int rc = lamsa_hp_upload_batch(h, B);
    if (rc) return rc;
    return lamsa_hp_run_uploaded(h, R);
}
