// hp_align_api.hip -- lamsa_hp_align_batch / upload_batch / run_uploaded (include/lamsa_hp.h):
// the batch form of the reference's per-read worker (src/lamsa_aln.c:857-871) on gfx950.
// One wavefront (= one 64-thread workgroup) per read, a persistent grid pulling reads from a
// queue head in costliest-first order, every wave with its own scratch slab in HBM; no
// collectives, no inter-workgroup communication except two atomics per read.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "hp_phase.h"
#include "hp_handle.h"
#include "hp_hostprep.h"

using namespace hp;

#ifndef HP_WAVES_PER_SIMD
#define HP_WAVES_PER_SIMD 4          // final kernel, default workload: 3 -> 314 ms, 4 -> 280, 5 -> 270 (but 70 % more HBM traffic), 6 -> 295
#endif
__global__ __launch_bounds__(64, HP_WAVES_PER_SIMD) void k_align_batch(AlignArgs a)
{
    __shared__ int32_t lds[HP_BOTH_LDS_WORDS];       // chaining state or DP rows, query window and direction matrix (hp_ksw.h)
    const int slot = blockIdx.x;
    for (;;) {
        int u = 0;
        if (wv::leader()) u = atomicAdd(a.counter, 1);
        u = wv::uni(u);
        if (u >= a.n_units) break;          // every wave reaches this exit: the queue head only grows
        align_read(a, a.order ? a.order[u] : u, slot, (HP_L int32_t *)lds);
    }
}

// The main pass as five launches (hp_phase.h).  Every kernel is a persistent grid of single-wave workgroups pulling
// its units from a queue head; every wave reaches the exit test (the heads only grow, the unit counts are final when
// the launch starts: they were written by the previous launch of the same stream).
#ifndef HP_CHAIN_WAVES_PER_SIMD
#define HP_CHAIN_WAVES_PER_SIMD 4
#endif
#ifndef HP_FILL_WAVES_PER_SIMD
#define HP_FILL_WAVES_PER_SIMD 7          // round 4, the fill without its DPs (profiles/r04_overlap.txt; ms of k_fill, ont10k): 8 waves per SIMD (64 VGPRs) 44.6, 7 (72) 34.2, 6 (80) 34.7, 5 (96) 40.8, 4 (128) 41.3 --
#endif                                    // with 64 registers the append loop of frags_merge reloads spilled values behind its own stores (loads and stores share vmcnt on gfx9: a reload waits for every store before it)
// The chaining kernels in three shapes, chosen per batch (chain_shape): waves per SIMD against LDS words per wave.  A cluster of up to LDS / 5
// hits runs out of LDS (hp_cluster.h), a larger one through the HBM scan; the line sets and the gap tables are bounded by the LDS likewise.
// Measured (profiles/r04_chain_lds.txt, ms of k_chain1, one step at a time): 10-kbp ONT reads (400 seeds) 70 / 81 / 103 at 4 / 3 / 2 waves,
// 20-kbp PacBio reads (800 seeds: the cluster at the read's true locus no longer fits 486 hits) 130 / 83 / 106.
template <int WORDS, int WPS>
__global__ __launch_bounds__(64, WPS) void k_chain1(const PhaseArgs *ap)
{
    const PhaseArgs &a = *ap;        // in device memory: scalar loads, no private copy of the argument block
    __shared__ int32_t lds[WORDS];   // the hit sort's blocks (hp_sort.h), then the node state of one cluster at a time (hp_cluster.h)
    for (;;) {
        int u = 0;
        if (wv::leader()) u = atomicAdd(&a.ctl->q_head[0], 1);
        u = wv::uni(u);
        if (u >= a.n_reads) break;
        phase_chain1(a, a.order ? a.order[u] : u, blockIdx.x, (HP_L int32_t *)lds, WORDS);
    }
    drain_stamp(a, 0);
}
template <int WORDS, int WPS>
__global__ __launch_bounds__(64, WPS) void k_chain2(const PhaseArgs *ap)
{
    const PhaseArgs &a = *ap;
    __shared__ int32_t lds[WORDS];
    for (;;) {
        int u = 0;
        if (wv::leader()) u = atomicAdd(&a.ctl->q_head[2], 1);
        u = wv::uni(u);
        if (u >= a.n_reads) break;
        phase_chain2(a, a.order ? a.order[u] : u, blockIdx.x, (HP_L int32_t *)lds, WORDS);
    }
    drain_stamp(a, 2);
}
typedef void (*ChainKernel)(const PhaseArgs *);
struct ChainShape { ChainKernel k1, k2; int lds_words, waves_per_simd; };
static const ChainShape g_chain_shapes[3] = {
    {k_chain1<HP_CHAIN_LDS_WORDS, HP_CHAIN_WAVES_PER_SIMD>, k_chain2<HP_CHAIN_LDS_WORDS, HP_CHAIN_WAVES_PER_SIMD>, HP_CHAIN_LDS_WORDS, HP_CHAIN_WAVES_PER_SIMD},
    {k_chain1<3392, 3>, k_chain2<3392, 3>, 3392, 3},
    {k_chain1<5120, 2>, k_chain2<5120, 2>, 5120, 2},
};
__global__ __launch_bounds__(64, HP_FILL_WAVES_PER_SIMD) void k_fill(const PhaseArgs *ap, int round)
{
    const PhaseArgs &a = *ap;
    __shared__ int32_t lds[HP_LDS_WORDS];            // this wave's DP rows, query window and direction matrix (hp_ksw.h)
    int n = 0;
    for (int b = 0; b < PH_NBUCKET; ++b) n += a.ctl->bucket_n[round][b];
    n = wv::uni(n);
    for (;;) {
        int g = 0;
        if (wv::leader()) g = atomicAdd(&a.ctl->q_head[1 + 2 * round], 1);
        g = wv::uni(g);
        if (g >= n) break;
        int b = 0;
        while (b < PH_NBUCKET - 1 && g >= a.ctl->bucket_n[round][b]) { g -= a.ctl->bucket_n[round][b]; ++b; }     // costliest class first
        const int u = wv::uni(a.bucket_q[((size_t)round * PH_NBUCKET + b) * a.unit_cap + g]);
        phase_fill(a, round, u, blockIdx.x, (HP_L int32_t *)lds);
    }
    drain_stamp(a, 1 + 2 * round);
}
#ifndef HP_FILLDP_WAVES_PER_SIMD
#define HP_FILLDP_WAVES_PER_SIMD 2
#endif
#ifndef HP_LIST_WAVES_PER_SIMD
#define HP_LIST_WAVES_PER_SIMD 4
#endif
__global__ __launch_bounds__(64, HP_LIST_WAVES_PER_SIMD) void k_filllist(const PhaseArgs *ap, int round)
{
    const PhaseArgs &a = *ap;
    int n = 0;
    for (int b = 0; b < PH_NBUCKET; ++b) n += a.ctl->bucket_n[round][b];
    n = wv::uni(n);
    for (;;) {
        int g = 0;
        if (wv::leader()) g = atomicAdd(&a.ctl->q_head[5], 1);
        g = wv::uni(g);
        if (g >= n) break;
        int b = 0;
        while (b < PH_NBUCKET - 1 && g >= a.ctl->bucket_n[round][b]) { g -= a.ctl->bucket_n[round][b]; ++b; }
        const int u = wv::uni(a.bucket_q[((size_t)round * PH_NBUCKET + b) * a.unit_cap + g]);
        phase_filllist(a, round, u, blockIdx.x, (HP_L int32_t *)nullptr);
    }
}
// the lane-per-job DP over the round's queues (hp_lanedp.h): 64 jobs per wave, rows of HP_LJ_QSMALL cells per lane in LDS
__global__ __launch_bounds__(64, 2) void k_filldp_small(const PhaseArgs *ap, int round)
{
    __shared__ int32_t lds[HP_LJ_LDS_WORDS(HP_LJ_QSMALL)];
    const PhaseArgs &a = *ap;
    int n = 0;
    for (int b = 0; b < LJ_NBUCKET; ++b) n += ((a.ctl->lj_bucket_n[round][b] < a.lj_cap ? a.ctl->lj_bucket_n[round][b] : a.lj_cap) + 63) >> 6;
    n = wv::uni(n);
    for (;;) {
        int g = 0;
        if (wv::leader()) g = atomicAdd(&a.ctl->q_head[6], 1);
        g = wv::uni(g);
        if (g >= n) break;
        int b = 0;
        for (; b < LJ_NBUCKET - 1; ++b) { const int gb = ((a.ctl->lj_bucket_n[round][b] < a.lj_cap ? a.ctl->lj_bucket_n[round][b] : a.lj_cap) + 63) >> 6; if (g < gb) break; g -= gb; }
        phase_filldp(a, round, b, g * 64, blockIdx.x, (HP_L int32_t *)lds, HP_LJ_QSMALL);
    }
}
// the wave-per-job DP (hp_wavejob.h): the junctions beyond a lane job, the end extensions of every line; costliest class first.
__global__ __launch_bounds__(64, HP_WJ_WAVES_PER_SIMD) void k_filldp_wave(const PhaseArgs *ap, int round)
{
    __shared__ int32_t lds[HP_WJ_LDS_WORDS];
    const PhaseArgs &a = *ap;
    int n = 0, n_big = 0;
    for (int b = 0; b < WJ_NBUCKET; ++b) { const int k = a.ctl->wj_bucket_n[round][b] < a.wj_cap ? a.ctl->wj_bucket_n[round][b] : a.wj_cap; if (b < WJ_NBIG) n_big += k; else n += k; }
    n = wv::uni(n); n_big = wv::uni(n_big);
    bool own_big = (int)blockIdx.x < a.n_wjb;           // this wave owns a big slab: the jobs that need one first (they are the costliest)
    for (;;) {
        int g = 0;
        const bool big = own_big;
        if (wv::leader()) g = atomicAdd(&a.ctl->q_head[big ? 8 : 7], 1);
        g = wv::uni(g);
        if (g >= (big ? n_big : n)) { if (!big) break; own_big = false; continue; }
        phase_wavejob(a, round, g, big, blockIdx.x, (HP_L int32_t *)lds);
    }
}
__global__ __launch_bounds__(64) void k_publish(const PhaseArgs *ap)
{
    const PhaseArgs &a = *ap;
    if (blockIdx.x == 0) publish_diag(a);
    for (;;) {
        int u = 0;
        if (wv::leader()) u = atomicAdd(&a.ctl->q_head[4], 1);
        u = wv::uni(u);
        if (u >= a.n_reads) break;
        phase_publish(a, u);
    }
}

// ---- the seed CIGARs on their way in: one byte per element and no offsets over PCIe, words and 64-bit offsets in HBM
__global__ void k_cig8_expand(const uint8_t *src, int32_t *dst, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { const int b = src[i]; dst[i] = ((b & 63) << 4) | (b >> 6); }
}
__global__ void k_off_widen(const int32_t *src, int64_t *dst, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// exclusive prefix sum of n bytes into 64-bit offsets, blocks of 4096 elements: per-block sums, their scan by one workgroup,
// then the local scan of every block on top of its base
#define HP_SCAN_BLOCK 4096
__global__ __launch_bounds__(256) void k_scan_sums(const uint8_t *src, int64_t n, int64_t *sums)
{
    __shared__ int part[256];
    const int64_t b0 = (int64_t)blockIdx.x * HP_SCAN_BLOCK;
    int s = 0;
    for (int i = threadIdx.x; i < HP_SCAN_BLOCK; i += 256) { const int64_t k = b0 + i; s += k < n ? src[k] : 0; }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0];
}
__global__ __launch_bounds__(256) void k_scan_bases(int64_t *sums, int64_t nb)
{   // one workgroup: every thread owns a contiguous slice of the block sums
    __shared__ long long tot[256];
    const int64_t per = (nb + 255) / 256, a = (int64_t)threadIdx.x * per, b = a + per < nb ? a + per : nb;
    long long s = 0;
    for (int64_t i = a; i < b; ++i) s += sums[i];
    tot[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { long long run = 0; for (int t = 0; t < 256; ++t) { const long long v = tot[t]; tot[t] = run; run += v; } }
    __syncthreads();
    long long run = tot[threadIdx.x];
    for (int64_t i = a; i < b; ++i) { const long long v = sums[i]; sums[i] = run; run += v; }
}
__global__ __launch_bounds__(256) void k_scan_apply(const uint8_t *src, int64_t n, const int64_t *bases, int64_t *dst)
{
    __shared__ int part[256];
    const int64_t b0 = (int64_t)blockIdx.x * HP_SCAN_BLOCK + (int64_t)threadIdx.x * 16;
    int v[16], s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int64_t k = b0 + i; v[i] = k < n ? src[k] : 0; s += v[i]; }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { int run = 0; for (int t = 0; t < 256; ++t) { const int x = part[t]; part[t] = run; run += x; } }
    __syncthreads();
    long long run = bases[blockIdx.x] + part[threadIdx.x];
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int64_t k = b0 + i; if (k < n) dst[k] = run; run += v[i]; }
}

static double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static const bool g_trace = getenv("LAMSA_HP_TRACE") != nullptr;      // phase times of the host side on stderr
static const bool g_noshare = getenv("LAMSA_HP_FULL_GRIDS") != nullptr;   // diagnostics: full-size grids also when two batches are in flight (slab_plan)
static const bool g_nowave = getenv("LAMSA_HP_NO_WAVE_JOBS") != nullptr; // diagnostics: skip the wave-per-job DP launch (the fill then runs the junctions beyond a lane job and the end extensions itself)
static const bool g_nolane = getenv("LAMSA_HP_NO_LANE_DP") != nullptr;  // diagnostics: skip the lane-per-job DP launches (the fill then runs every DP itself, one job per wave)
static const bool g_mono = getenv("LAMSA_HP_ONE_KERNEL") != nullptr;  // diagnostics: the main pass through k_align_batch (the retry pass's kernel) instead of the phased launches

struct HostBuf {                  // page-locked host memory, mapped into the device's address space
    void *p = nullptr, *dev = nullptr; size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) hipHostFree(p);
        p = dev = nullptr; cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        if (hipHostMalloc(&p, want, hipHostMallocMapped) != hipSuccess) { p = nullptr; return -1; }
        if (hipHostGetDevicePointer(&dev, p, 0) != hipSuccess) { hipHostFree(p); p = nullptr; return -1; }
        cap = want;
        return 0;
    }
    void release() { if (p) hipHostFree(p); p = dev = nullptr; cap = 0; }
};

// Result arrays of one launch.  The stream arena is in HBM and fetched with one large copy.  The per-read offset /
// length / status / bases arrays (20 B per read) are written by the kernel straight into mapped host memory: a small
// device-to-host copy is a runtime kernel launch that, measured, does not start while another batch's persistent
// grid is running (a 0.6 GB copy took 10 ms beside it, five 0.5 MB ones 380 ms).
struct OutDev {
    DevBuf buf; HostBuf host; int64_t stream_cap = 0; int n_cap = 0;
    static size_t hdr(int n) { return al256(8 * (size_t)n) + 3 * al256(4 * (size_t)n) + 256 + al256(16 * (size_t)n); }      // offsets, three int arrays, 32 words of launch accounting, four accounting words per read
    int ensure(int n, int64_t cap) { stream_cap = cap; n_cap = n; return buf.ensure(4 * (size_t)cap + 256) || host.ensure(hdr(n) + 256); }
    // device-visible addresses (kernel arguments)
    int64_t *off() const { return (int64_t *)host.dev; }
    int32_t *len(int n) const { return (int32_t *)((char *)host.dev + al256(8 * (size_t)n)); }
    int32_t *st(int n) const { return (int32_t *)((char *)host.dev + al256(8 * (size_t)n) + al256(4 * (size_t)n)); }
    int32_t *tb(int n) const { return (int32_t *)((char *)host.dev + al256(8 * (size_t)n) + 2 * al256(4 * (size_t)n)); }
    unsigned long long *diag(int n) const { return (unsigned long long *)((char *)host.dev + al256(8 * (size_t)n) + 3 * al256(4 * (size_t)n)); }
    const unsigned long long *h_diag(int n) const { return (const unsigned long long *)((const char *)host.p + al256(8 * (size_t)n) + 3 * al256(4 * (size_t)n)); }
    int32_t *work(int n) const { return (int32_t *)((char *)host.dev + al256(8 * (size_t)n) + 3 * al256(4 * (size_t)n) + 256); }
    const int32_t *h_work(int n) const { return (const int32_t *)((const char *)host.p + al256(8 * (size_t)n) + 3 * al256(4 * (size_t)n) + 256); }
    // the same arrays as the host sees them (valid once the launch has completed)
    const int64_t *h_off() const { return (const int64_t *)host.p; }
    const int32_t *h_len(int n) const { return (const int32_t *)((const char *)host.p + al256(8 * (size_t)n)); }
    const int32_t *h_st(int n) const { return (const int32_t *)((const char *)host.p + al256(8 * (size_t)n) + al256(4 * (size_t)n)); }
    const int32_t *h_tb(int n) const { return (const int32_t *)((const char *)host.p + al256(8 * (size_t)n) + 2 * al256(4 * (size_t)n)); }
    int32_t *stream(int) const { return (int32_t *)buf.p; }
    void release() { buf.release(); host.release(); }
};

struct Slot {                     // one batch on the device: its inputs, the outputs of its main pass, its launch state
    DevBuf bin, misc, slab, pers, prof; OutDev out1;
    HostBuf args_host;            // page-locked copy of the launch sequence's argument block: the asynchronous copy to the device reads it after launch_phased has returned
    hipEvent_t ep[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // between the phases of the main pass (of round 1: ep[5] after the listing, ep[6] after the wave-per-job DP, ep[4] after the lane DP)
    hipStream_t cs = nullptr;     // the compute stream of this slot: the two slots' kernels run on different streams, so
                                  // that the waves of the next batch fill the SIMDs the tail of the previous one leaves idle
    bool valid = false;           // a batch is resident
    bool phased = false;          // the launch in flight on this slot's resources is the phased main pass
    int32_t n_reads = 0; int64_t n_bases = 0, n_cig = 0, n_hits = 0;
    BatchIn in; const int32_t *d_order = nullptr;
    std::vector<int32_t> order, h_len, h_H; std::vector<uint8_t> h_skip;      // h_skip: reads the batch check found beyond the device's field widths
    int32_t max_L = 0, max_H = 0;
    int32_t sort_pb = 40, sort_cb = 24;        // key field widths for the in-kernel sort of the hits (hp_sort.h)
    hipEvent_t e0 = nullptr, e1 = nullptr;     // around the main-pass kernel, on the compute stream
};

struct AlignState {
    Slot slot[2];                 // two batches: one computing, one being uploaded (lamsa_hp_submit_batch)
    int fifo[2] = {0, 0}, n_fifo = 0;          // submitted and not yet collected, oldest first
    int res_fifo[2] = {0, 0}, n_res = 0;       // runs of the resident batch (slot 0) started and not yet finished: the lane each uses
    DevBuf retry_list;            // second passes run one at a time (inside collect / run_uploaded)
    OutDev out2;
    // host copies of the results
    HostBuf stream;               // page-locked: the result stream
    std::vector<int32_t> r_len, r_st, r_tb, r_work; std::vector<int64_t> r_off;
};

static std::map<lamsa_hp_handle *, AlignState *> g_states;     // per-handle state of the align entry points
static std::mutex g_states_lock;                                // a handle is single-threaded, different handles may live on different threads
static AlignState *state_of(lamsa_hp_handle *h)
{
    std::lock_guard<std::mutex> guard(g_states_lock);
    auto it = g_states.find(h);
    if (it != g_states.end()) return it->second;
    return g_states[h] = new AlignState();
}
extern "C" void lamsa_hp_release_state_(lamsa_hp_handle *h)
{
    AlignState *S = nullptr;
    {
        std::lock_guard<std::mutex> guard(g_states_lock);
        auto it = g_states.find(h);
        if (it == g_states.end()) return;
        S = it->second;
        g_states.erase(it);
    }
    if (h->stream) hipStreamSynchronize(h->stream);      // batches submitted and never collected
    if (h->stream_b) hipStreamSynchronize(h->stream_b);
    for (Slot &T : S->slot) { T.bin.release(); T.misc.release(); T.slab.release(); T.pers.release(); T.prof.release(); T.args_host.release(); T.out1.release(); for (hipEvent_t e : {T.e0, T.e1, T.ep[0], T.ep[1], T.ep[2], T.ep[3], T.ep[4], T.ep[5], T.ep[6]}) if (e) hipEventDestroy(e); }
    S->retry_list.release(); S->out2.release(); S->stream.release();
    delete S;
}

// grow a device buffer that a queued kernel may still be using: drain both compute streams first
static int grow(lamsa_hp_handle *h, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return 0;
    if (hipStreamSynchronize(h->stream) != hipSuccess || hipStreamSynchronize(h->stream_b) != hipSuccess) return -1;
    return b.ensure(bytes);
}

static size_t slab_bytes_for(const lamsa_hp_para &P, int L, int H, int scale)
{   // per-wave scratch: node arrays, sort index + line sets (~424 B/hit), result + CIGAR buffers (~128 B/base), and the
    // direction matrix of the largest extension: (2w+1) columns x (L + 2*hash_step) rows.  Reads that need more
    // flag LAMSA_HP_ST_OVERFLOW and are re-run by the retry pass with `scale` = 8.
    const size_t z = (2 * (size_t)P.band_w + 128) * ((size_t)L + 256);
    return al256(((size_t)256 << 10) + (size_t)scale * 128 * (size_t)L + 424 * (size_t)H + z * (size_t)(scale > 1 ? 4 : 1));
}

// validate `B`, build its processing order and sort index, copy it into slot `T` on the copy stream

static int upload_into(lamsa_hp_handle *h, Slot *S, const lamsa_hp_batch *B)
{
    S->valid = false;
    const double t_0 = now_s();
    const int n = B->n_reads;
    const int64_t n_slots = n ? B->seed_off[n] : 0, n_hits = n_slots ? B->hit_off[n_slots] : 0, n_bases = n ? B->read_off[n] : 0;
    // ---- validate everything the kernels index with, on the host, before anything is launched
    S->h_len.assign((size_t)n, 0); S->h_H.assign((size_t)n, 0); S->h_skip.assign((size_t)n + 1, 0); S->max_L = 0; S->max_H = 0;
    if (n && (B->seed_off[0] != 0 || B->read_off[0] != 0 || (n_slots && B->hit_off[0] != 0))) { h->err = "offsets must start at 0"; return LAMSA_HP_EINVAL; }
    const bool packed_off = B->h_cig_off == nullptr, cig_bytes = B->cig8 != nullptr;
    if (B->n_cig < 0 || (!packed_off && B->n_cig > 0x7fffffffll)) { h->err = "more than 2^31-1 seed CIGAR elements with 32-bit h_cig_off: pass h_cig_off = NULL (CIGARs back to back in hit order) or split the batch"; return LAMSA_HP_EINVAL; }
    if (!cig_bytes && !B->cig && B->n_cig > 0) { h->err = "neither cig nor cig8 given"; return LAMSA_HP_EINVAL; }
    {
        std::atomic<int> bad(0);                         // 1..9: which check failed (the first one reported wins)
        const int64_t sl = h->para.seed_len, ss = h->para.seed_step > 0 ? h->para.seed_step : 1;
        std::atomic<long long> max_pos(0), sum_cig(0);
        hp_parallel_blocks(n, [&](int r0, int r1) {
            long long mp = 0, sc = 0;
            for (int r = r0; r < r1 && !bad.load(std::memory_order_relaxed); ++r) {
                const int64_t L = B->read_off[r + 1] - B->read_off[r], ns = B->seed_off[r + 1] - B->seed_off[r];
                if (L < 0 || L > (1 << 24) || ns < 0) { bad = 1; return; }
                if (ns > HP_MAX_SLOTS) S->h_skip[r] = 1;                   // more seed slots than the packed keys hold: the read is not aligned (ST_UNSUPPORTED)
                {   // the seed geometry the kernels derive read windows from (lamsa_aln.c:251-253,281) must be the read's own
                    const int64_t sa = L < sl ? 0 : 1 + (L - sl) / ss;
                    if (B->seed_all[r] != sa || B->last_len[r] != L - sl - (sa - 1) * ss) { bad = 7; return; }
                    if (sa > 32767) S->h_skip[r] = 1;                      // seed ids are kept in 16 bits on the device (NodeS::sid): the read is not aligned (ST_UNSUPPORTED)
                }
                int64_t H = 0;
                for (int64_t s = B->seed_off[r]; s < B->seed_off[r + 1]; ++s) {
                    const int64_t m = B->hit_off[s + 1] - B->hit_off[s];
                    if (m < 0 || (m > HP_MAX_HITS_PER_SEED && !S->h_skip[r])) { bad = 2; return; }       // (a read the kernels skip anyway is not held to the limits of their indexing: it only must not break the host's)
                    if (B->seed_id[s] < 1 || B->seed_id[s] > B->seed_all[r] || (s > B->seed_off[r] && B->seed_id[s] <= B->seed_id[s - 1])) { bad = 3; return; }
                    H += m;
                }
                if (H > (1 << 22) && !S->h_skip[r]) { bad = 4; return; }
                for (int64_t i = B->read_off[r]; i < B->read_off[r + 1]; ++i) if (B->read_seq[i] > 4) { bad = 5; return; }
                for (int64_t k = B->hit_off[B->seed_off[r]]; k < B->hit_off[B->seed_off[r + 1]]; ++k) {
                    mp = B->h_pos[k] > mp ? B->h_pos[k] : mp;
                    if (B->h_len_dif[k] < -127 || B->h_len_dif[k] > 127) S->h_skip[r] = 1;        // kept in 8 bits on the device (NodeS::len_dif8): the read is not aligned
                    if (B->h_chr[k] < 1 || B->h_chr[k] > h->n_seqs || (B->h_strand[k] != 1 && B->h_strand[k] != -1) || B->h_pos[k] < 0 || B->h_pos[k] >= (1ll << 40) ||
                        (!packed_off && (B->h_cig_off[k] < 0 || (int64_t)B->h_cig_off[k] + B->h_cig_n[k] > B->n_cig))) { bad = 6; return; }
                    sc += B->h_cig_n[k];
                }
                S->h_len[r] = (int32_t)L; S->h_H[r] = (int32_t)H;
            }
            long long seen = max_pos.load();
            while (mp > seen && !max_pos.compare_exchange_weak(seen, mp)) { }
            sum_cig += sc;
        });
        static const char *why[] = {"", "read longer than 2^24 bases", "too many hits in one seed", "seed ids must be ascending in [1, seed_all]",
                                    "too many hits in one read", "read base code > 4",
                                    "bad hit record (contig id, strand, position or seed CIGAR range)",
                                    "seed_all / last_len do not match the read length and the handle's seed length and step"};
        if (bad) { h->err = why[bad.load()]; return LAMSA_HP_EINVAL; }
        if (packed_off && sum_cig.load() != B->n_cig) { h->err = "h_cig_off is NULL but the h_cig_n do not add up to n_cig"; return LAMSA_HP_EINVAL; }
        for (int r = 0; r < n; ++r) { if (S->h_skip[r]) { S->h_len[r] = 0; S->h_H[r] = 0; } S->max_L = std::max(S->max_L, S->h_len[r]); S->max_H = std::max(S->max_H, S->h_H[r]); }
        auto bits = [](unsigned long long x) { int b = 0; while (x) { ++b; x >>= 1; } return b; };
        S->sort_pb = bits((unsigned long long)max_pos.load()); S->sort_cb = bits((unsigned long long)(2 * (long long)h->n_seqs + 1));
    }
    const double t_1 = now_s();
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
                 o_pos = place(8 * (size_t)n_hits), o_chr = place(4 * (size_t)n_hits), o_coff = place(8 * (size_t)n_hits), o_nm = place(2 * (size_t)n_hits),
                 o_ld = place(2 * (size_t)n_hits), o_st = place((size_t)n_hits), o_cn = place((size_t)n_hits), o_cig = place(4 * (size_t)B->n_cig), o_ord = place(4 * (size_t)n),
                 o_stage = place(packed_off ? 8 * ((size_t)n_hits / HP_SCAN_BLOCK + 2) : 4 * (size_t)n_hits), o_cig8 = place(cig_bytes ? (size_t)B->n_cig : 0), o_skip = place((size_t)n);
    const double t_3 = now_s();
    if (S->bin.ensure(off)) { h->err = "hipMalloc(batch)"; return LAMSA_HP_ENOMEM; }
    char *d = (char *)S->bin.p;
    hipStream_t s = h->copy_stream;
    static const int64_t zero64 = 0;
#define UP(o, src, bytes) do { if ((bytes) > 0) HIPCHK(h, hipMemcpyAsync(d + (o), (src), (bytes), hipMemcpyHostToDevice, s), LAMSA_HP_EKERNEL); } while (0)
    if (n) { UP(o_roff, B->read_off, 8 * ((size_t)n + 1)); UP(o_soff, B->seed_off, 8 * ((size_t)n + 1)); }
    else { UP(o_roff, &zero64, 8); UP(o_soff, &zero64, 8); }
    UP(o_rseq, B->read_seq, (size_t)n_bases); UP(o_sall, B->seed_all, 4 * (size_t)n); UP(o_last, B->last_len, 4 * (size_t)n);
    UP(o_sid, B->seed_id, 4 * (size_t)n_slots);
    if (n_slots) UP(o_hoff, B->hit_off, 8 * ((size_t)n_slots + 1)); else UP(o_hoff, &zero64, 8);
    UP(o_pos, B->h_pos, 8 * (size_t)n_hits); UP(o_chr, B->h_chr, 4 * (size_t)n_hits);
    if (!packed_off) UP(o_stage, B->h_cig_off, 4 * (size_t)n_hits);
    UP(o_nm, B->h_nm, 2 * (size_t)n_hits); UP(o_ld, B->h_len_dif, 2 * (size_t)n_hits); UP(o_st, B->h_strand, (size_t)n_hits); UP(o_cn, B->h_cig_n, (size_t)n_hits);
    if (cig_bytes) UP(o_cig8, B->cig8, (size_t)B->n_cig); else UP(o_cig, B->cig, 4 * (size_t)B->n_cig);
    UP(o_ord, S->order.data(), 4 * (size_t)n); UP(o_skip, S->h_skip.data(), (size_t)n);
#undef UP
    // on the device: CIGAR bytes -> words, offsets widened or summed up from the lengths (same stream, behind the copies)
    if (cig_bytes && B->n_cig > 0) hipLaunchKernelGGL(k_cig8_expand, dim3(2048), dim3(256), 0, s, (const uint8_t *)(d + o_cig8), (int32_t *)(d + o_cig), (int64_t)B->n_cig);
    if (n_hits > 0) {
        if (!packed_off) hipLaunchKernelGGL(k_off_widen, dim3(2048), dim3(256), 0, s, (const int32_t *)(d + o_stage), (int64_t *)(d + o_coff), (int64_t)n_hits);
        else {
            const int64_t nb = (n_hits + HP_SCAN_BLOCK - 1) / HP_SCAN_BLOCK;
            hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nb), dim3(256), 0, s, (const uint8_t *)(d + o_cn), (int64_t)n_hits, (int64_t *)(d + o_stage));
            hipLaunchKernelGGL(k_scan_bases, dim3(1), dim3(256), 0, s, (int64_t *)(d + o_stage), nb);
            hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, s, (const uint8_t *)(d + o_cn), (int64_t)n_hits, (const int64_t *)(d + o_stage), (int64_t *)(d + o_coff));
        }
    }
    HIPCHK(h, hipGetLastError(), LAMSA_HP_EKERNEL);
    HIPCHK(h, hipStreamSynchronize(s), LAMSA_HP_EKERNEL);
    if (g_trace) fprintf(stderr, "[lamsa_hp] upload: validate %.1f ms, order %.1f ms, copy %.1f ms (%.2f GB on the device)\n", 1e3 * (t_1 - t_0), 1e3 * (t_3 - t_1), 1e3 * (now_s() - t_3), off / 1e9);
    BatchIn &in = S->in;
    in.n_reads = n; in.read_skip = (const uint8_t *)(d + o_skip); in.read_off = (const int64_t *)(d + o_roff); in.read_seq = (const uint8_t *)(d + o_rseq);
    in.seed_all = (const int32_t *)(d + o_sall); in.last_len = (const int32_t *)(d + o_last); in.seed_off = (const int64_t *)(d + o_soff);
    in.seed_id = (const int32_t *)(d + o_sid); in.hit_off = (const int64_t *)(d + o_hoff); in.h_pos = (const int64_t *)(d + o_pos);
    in.h_chr = (const int32_t *)(d + o_chr); in.h_cig_off = (const int64_t *)(d + o_coff); in.h_nm = (const int16_t *)(d + o_nm);
    in.h_len_dif = (const int16_t *)(d + o_ld); in.h_strand = (const int8_t *)(d + o_st); in.h_cig_n = (const uint8_t *)(d + o_cn);
    in.cig = (const int32_t *)(d + o_cig);
    S->d_order = (const int32_t *)(d + o_ord);
    S->n_reads = n; S->n_bases = n_bases; S->n_cig = B->n_cig; S->n_hits = n_hits;
    S->valid = true;
    return LAMSA_HP_OK;
}

// Waves whose scratch slabs fit the device: two batches can be in flight, each with its own slabs, beside the inputs, the
// inter-launch state and the outputs of both -- 72 GB of slabs per batch in flight on a 288-GB device (100 GB each did not fit beside
// the rest for 20-kbp reads).  Whole CUs' worth of waves where that is possible (20-kbp reads at -w 200 need 17 MB per wave: 4 096
// waves instead of the fill kernel's 8 192).
static int cap_waves(int n_waves, size_t slab_per_wave, int n_cu)
{
    // a quarter of the device's memory per batch in flight (72 GB of an MI355X's 288 GB), from the device itself -- and no more than a third of
    // what was FREE when this process first asked (another process may hold part of the device: one shard per process and several on a GPU)
    static const size_t budget = []() { size_t f = 0, t = 0; if (hipMemGetInfo(&f, &t) != hipSuccess || t == 0) { t = (size_t)288 << 30; f = t; } return std::min(t / 4, f / 3); }();
    if (slab_per_wave * (size_t)n_waves <= budget) return n_waves;
    size_t fit = budget / (slab_per_wave ? slab_per_wave : 1);
    if (n_cu > 0 && fit >= (size_t)n_cu) fit -= fit % (size_t)n_cu;
    return fit < 1 ? 1 : (int)fit;
}

#ifdef HP_PROF
// diagnostic builds (-DHP_PROF): per-read cycle counters kept by the kernels, summed and printed after a launch
static void prof_report(Slot &T, const long long *d_prof, int n)
{
    if (!d_prof) return;
    {
        std::vector<long long> pr((size_t)n * 64);
        hipMemcpy(pr.data(), d_prof, sizeof(long long) * pr.size(), hipMemcpyDeviceToHost);
        if (const char *dump = getenv("LAMSA_HP_PROF_DUMP")) {          // per read: H, L, then the 64 counters
            if (FILE *fp = fopen(dump, "wb")) {
                for (int r = 0; r < n; ++r) { long long hl[2] = {T.h_H[r], T.h_len[r]}; fwrite(hl, 8, 2, fp); fwrite(&pr[(size_t)r * 64], 8, 64, fp); }
                fclose(fp);
            }
        }
        std::vector<int> idx((size_t)n); for (int i = 0; i < n; ++i) idx[i] = i;
        auto tot = [&](int r) { long long t = 0; for (int k = 0; k < 6; ++k) t += pr[(size_t)r * 64 + k]; return t; };
        std::sort(idx.begin(), idx.end(), [&](int x, int y) { return tot(x) > tot(y); });
        long long sum[64] = {0}; for (int r = 0; r < n; ++r) for (int k = 0; k < 64; ++k) sum[k] += pr[(size_t)r * 64 + k];
        fprintf(stderr, "[HP_PROF] cycles: setup chain1 fill1 chain2 fill2 publish | in chain1: init+minext mainscan track pop-loop bound+flines | o_l H\n");
        fprintf(stderr, "[HP_PROF] SUM  "); for (int k = 0; k < 11; ++k) fprintf(stderr, " %lld", sum[k] / 1000000); fprintf(stderr, " (Mcycles) targets %lld trips %lld init_Mcyc %lld\n", sum[11], sum[12], sum[13] / 1000000);
#ifdef HP_PROF_TRACK
        fprintf(stderr, "[HP_PROF] branch tracking (-DHP_PROF_TRACK: the line_build stamps below are off): marking pass %lld Mcyc; %lld seeds visited; %lld tracks, %lld Mcyc in them, %lld steps up; "
                        "%lld arrivals at a node with several sons, %lld Mcyc in cut_branch\n", sum[16] / 1000000, sum[19], sum[18], sum[17] / 1000000, sum[20], sum[21], sum[22] / 1000000);
#endif
        fprintf(stderr, "[HP_PROF] line_build: %lld lines (%lld without a gap), %lld anchors, %lld gaps | Mcyc: anchor walk %lld, gap list %lld, gaps in lanes %lld, gaps one by one %lld, assembly %lld\n",
                sum[21], sum[22], sum[12], sum[23], sum[16] / 1000000, sum[17] / 1000000, sum[18] / 1000000, sum[19] / 1000000, sum[20] / 1000000);
        fprintf(stderr, "[HP_PROF] gaps: %lld in lines with fewer than HP_GAP_MIN gaps (%lld lines with gaps); lane gaps: %lld hits scanned, %lld refused (range too long or too many active hits)\n", sum[54], sum[55], sum[11], sum[13]);
        { const char *nm[] = {"ksw_global", "ksw_extend", "backtrack", "ref_fetch", "head_fix", "frag_extend", "split_mapping", "tail_fix", "res_split", "res_aux", "mini_line_regs", "sort index"};
          for (int k = 0; k < 12; ++k) fprintf(stderr, "[HP_PROF] %-16s %8lld Mcyc %10lld calls\n", nm[k], sum[24 + 2 * k] / 1000000, sum[25 + 2 * k]);
#ifdef HP_PROF_FILL
          fprintf(stderr, "[HP_PROF] in frags_merge: boundary repair (merge_cigar_full) %lld Mcyc %lld calls; junctions through split_mapping %lld Mcyc %lld calls; fragments of several seeds %lld Mcyc %lld calls; 64-step gathers %lld Mcyc %lld\n", sum[16] / 1000000, sum[17], sum[18] / 1000000, sum[19], sum[20] / 1000000, sum[21], sum[22] / 1000000, sum[23]);
#endif
          fprintf(stderr, "[HP_PROF] direction-matrix bytes in HBM %lld (%lld jobs); hits passed through nodes_per_init %lld (%lld calls)\n", sum[52], sum[53], sum[54], sum[55]);
          fprintf(stderr, "[HP_PROF] chain_first start: seed loop %lld, node_set loop %lld, min_extend %lld Mcyc (%lld reads with a MIN pass)\n", sum[56] / 1000000, sum[57] / 1000000, sum[58] / 1000000, sum[59]);
          fprintf(stderr, "[HP_PROF] query <= 62: ksw_extend %lld Mcyc %lld calls, ksw_global %lld Mcyc %lld calls\n", sum[60] / 1000000, sum[61], sum[62] / 1000000, sum[63]);
          const char *bn[] = {"[bi_extend total]", "[bi_extend after left ext]", "17-32", "33-64", "65-128", "129-256", "257-512", ">512"};
          for (int k = 0; k < 2; ++k) fprintf(stderr, "[HP_PROF] ksw_extend qlen %-8s %8lld Mcyc %10lld calls\n", bn[k], sum[48 + 2 * k] / 1000000, sum[49 + 2 * k]); }
        {   // the wave-per-job launch's ksw_extend calls by routine (second half of the buffer)
            std::vector<long long> pd((size_t)n * 64);
            hipMemcpy(pd.data(), d_prof + ((size_t)n + 1) * 64, sizeof(long long) * pd.size(), hipMemcpyDeviceToHost);
            long long sd[64] = {0}; for (int r = 0; r < n; ++r) for (int k = 0; k < 64; ++k) sd[k] += pd[(size_t)r * 64 + k];
            const char *cn[] = {"one set, q <= 62", "packed, one set", "packed, two sets", "band 1 set", "band 2 sets", "band 4 sets", "int32 sets", "LDS / HBM rows"};
            for (int c = 0; c < 8; ++c) fprintf(stderr, "[HP_PROF] wave jobs, ksw_extend %-18s %8lld Mcyc %10lld Mcells %9lld calls %10lld rows\n", cn[c], sd[4 * c] / 1000000, sd[4 * c + 1] / 1000000, sd[4 * c + 2], sd[4 * c + 3]);
        }
        for (int q = 0; q < 8 && q < n; ++q) { int r = idx[q]; fprintf(stderr, "[HP_PROF] read %d L=%d:", r, T.h_len[r]); for (int k = 0; k < 11; ++k) fprintf(stderr, " %lld", pr[(size_t)r * 64 + k] / 1000000); fprintf(stderr, " | o_l %lld H %lld | targets %lld trips %lld init_Mcyc %lld\n", pr[(size_t)r * 64 + 14], pr[(size_t)r * 64 + 15], pr[(size_t)r * 64 + 11], pr[(size_t)r * 64 + 12], pr[(size_t)r * 64 + 13] / 1000000); }
    }
}
#endif

// state of a batch between the launches (hp_phase.h) in one device buffer: per-hit arrays indexed by global hit index + read index,
// the fragment, line and job arenas, the fill units and their cost-class queues, the job queues of the lane DP
struct PhasedLayout { size_t bytes, o[15]; int unit_cap, lj_cap, wj_cap; int64_t fl_cap, line_cap, job_cap; };
static PhasedLayout phased_layout(int n, int64_t n_hits, int64_t n_bases, int64_t stream_cap)
{
    PhasedLayout Y;
    const size_t n_ent = (size_t)n_hits + (size_t)n + 1;
    Y.unit_cap = 8 * n + 1024;
    Y.fl_cap = 4 * n_hits + 2304 * (int64_t)n + 4096; Y.line_cap = stream_cap + 64 * (int64_t)Y.unit_cap;
    Y.job_cap = std::min<int64_t>(0x7fffff00ll, 4096 + 256 * (int64_t)n + 2 * n_bases);      // CIGARs of the DP jobs computed ahead (~ 0.4 words per read base)
    Y.lj_cap = (int)std::min<int64_t>(0x3fffffffll, 1024 + 16 * (int64_t)n + n_bases / 100);   // lane DP jobs of a round (~ one per 150 read bases)
    Y.wj_cap = (int)std::min<int64_t>(0x3fffffffll, 1024 + 32 * (int64_t)n + n_bases / 200);   // wave DP jobs of a round (~ one per 600 read bases, and two per line)
    size_t off = 0;
    auto place = [&](size_t bytes) { size_t o = off; off = al256(off + bytes + 16); return o; };
    const size_t sz[15] = {sizeof(PhaseArgs), sizeof(PhaseCtl), sizeof(RdMeta) * ((size_t)n + 1), sizeof(NodeS) * n_ent, 4 * n_ent, 8 * n_ent, sizeof(UnitRec) * 2 * (size_t)Y.unit_cap,
                           4 * 2 * (size_t)PH_NBUCKET * Y.unit_cap, 4 * (size_t)Y.fl_cap, 4 * (size_t)Y.line_cap, 4 * (size_t)Y.job_cap, sizeof(LjRec) * (size_t)Y.lj_cap, 4 * (size_t)LJ_NBUCKET * Y.lj_cap,
                           sizeof(WjRec) * (size_t)Y.wj_cap, 4 * (size_t)WJ_NBUCKET * Y.wj_cap};
    for (int k = 0; k < 15; ++k) Y.o[k] = place(sz[k]);
    Y.bytes = off;
    return Y;
}
static int64_t main_stream_cap(int n, int64_t n_bases) { return 1024 + (int64_t)n * 256 + 4 * n_bases; }

// The main pass of the batch in slot `T` as the five launches of hp_phase.h, with the launch resources of slot `Ln`.
// scratch of a wave of each kind of launch (all launches of a batch share one allocation, one after the other): the chaining launches keep
// per-hit arrays, the listing / lane-DP / fill launches result and CIGAR buffers and the small DPs the fill still runs itself, the
// wave-per-job launch the direction matrix of the longest end extension
struct SlabPlan { size_t chain, fill, wj, wjb, wjb_off; int w_chain, w_fill, w_dp, w_wj, n_wjb, shape; size_t bytes; };
// shared: another batch's launches are in flight on the handle's other stream.  The launches are persistent grids; at full size the earlier
// batch's grid owns every wave slot and the later one only gets what its tail leaves.  The DP launch is bound by instruction issue (VALU port
// 78 % busy, profiles/r04_ont10k_pmc.json) and the chaining / fill launches by memory latency (wait 74-89 %, VALU 26-38 %): with every grid
// capped at half a CU's slots the launches of the two batches run side by side on the same CUs, one filling the issue slots the other leaves
// idle -- measured 357 k reads/s against 345 k with full grids (profiles/r04_overlap.txt).  A batch that runs alone gets the whole CU.
// which shape of the chaining kernels suits the batch: the hits at a read's true locus are at most one per seed -- about two in three of
// the seeds of a noisy read have one -- and that cluster should run out of LDS (LDS words / 5 hits)
static int chain_shape(const lamsa_hp_para &P, int64_t n_reads, int64_t n_bases)
{
    static const int force = getenv("LAMSA_HP_CHAIN_SHAPE") ? atoi(getenv("LAMSA_HP_CHAIN_SHAPE")) : -1;          // diagnostic
    if (force >= 0 && force < 3) return force;
    const int64_t mean_L = n_reads > 0 ? n_bases / n_reads : 0, ss = P.seed_step > 0 ? P.seed_step : 1;
    const int64_t est = mean_L / ss * 2 / 3;
    for (int k = 0; k < 2; ++k) if (est <= g_chain_shapes[k].lds_words / 5) return k;
    return 2;
}
static SlabPlan slab_plan(lamsa_hp_handle *h, int max_L, int max_H, bool shared = false, int shape = 0)
{
    SlabPlan Q;
    Q.shape = shape;
    const lamsa_hp_para &P = h->para;
    // chaining: per-hit arrays, line sets, fragments.  Listing / lane DP / fill: result and CIGAR buffers, the small DPs the fill still runs
    // itself, a lane-DP group's buffers.  Wave jobs: an ordinary slab takes the junctions and the end extensions of a few thousand rows; the
    // direction matrix of the longest end extension (the whole read long) lives in one of the big slabs that only the first waves own.
    Q.chain = al256(((size_t)256 << 10) + 128 * (size_t)max_L + 424 * (size_t)max_H);
    Q.fill = al256(((size_t)256 << 10) + 128 * (size_t)max_L + sizeof(cig_t) * 3 * HP_LJ_CIG * 64 + (size_t)HP_LJ_QSMALL * HP_LJ_TSMALL * 64 + 64);
    { static const int kb = getenv("LAMSA_HP_WJ_SLAB_KB") ? atoi(getenv("LAMSA_HP_WJ_SLAB_KB")) : 0;    // diagnostic
      Q.wj = ((size_t)(kb > 0 ? kb : 1024) << 10) + 33 * 256; }
    Q.wjb = al256((size_t)wj_need(&P, WJ_HEAD, max_L, max_L + 2 * P.hash_step + 64) + ((size_t)64 << 10)) + 33 * 256;
    if (Q.wjb < Q.wj) Q.wjb = Q.wj;
    if (g_nowave) Q.fill = std::max(Q.fill, slab_bytes_for(P, max_L, max_H, 1));            // (diagnostics: the fill runs every DP itself)
    if (h->scratch_limit) { const size_t lim = al256(h->scratch_limit); Q.chain = std::min(Q.chain, lim); Q.fill = std::min(Q.fill, lim); Q.wj = std::min(Q.wj, lim); Q.wjb = std::min(Q.wjb, lim); }
    int pc = 0, pf = 0, pd = 0, pw = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, g_chain_shapes[shape].k1, 64, 0) != hipSuccess || pc < 1) pc = 4 * g_chain_shapes[shape].waves_per_simd;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pf, k_fill, 64, 0) != hipSuccess || pf < 1) pf = 4;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pd, k_filldp_small, 64, 0) != hipSuccess || pd < 1) pd = 4;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pw, k_filldp_wave, 64, 0) != hipSuccess || pw < 1) pw = 4;
    if (shared && !g_noshare) { pc = std::min(pc, 8); pf = std::min(pf, 16); pw = std::min(pw, 16); }
    // diagnostic: LAMSA_HP_FILL_PER_CU / LAMSA_HP_CHAIN_PER_CU / LAMSA_HP_WJ_PER_CU set the grids' waves per CU
    { static const int cf = getenv("LAMSA_HP_FILL_PER_CU") ? atoi(getenv("LAMSA_HP_FILL_PER_CU")) : 0, cc = getenv("LAMSA_HP_CHAIN_PER_CU") ? atoi(getenv("LAMSA_HP_CHAIN_PER_CU")) : 0,
                       cw = getenv("LAMSA_HP_WJ_PER_CU") ? atoi(getenv("LAMSA_HP_WJ_PER_CU")) : 0;
      if (cf > 0 && cf < pf) pf = cf;
      if (cc > 0 && cc < pc) pc = cc;
      if (cw > 0 && cw < pw) pw = cw; }
    Q.w_chain = cap_waves(h->n_cu * pc, Q.chain, h->n_cu); Q.w_fill = cap_waves(h->n_cu * pf, Q.fill, h->n_cu);
    Q.w_dp = std::min(h->n_cu * pd, Q.w_fill); Q.w_wj = cap_waves(h->n_cu * pw, Q.wj, h->n_cu);
    // big slabs: 12 GB of them, at most one per wave of the launch and at least one per CU's worth of waves where that fits
    { const size_t budget_big = (size_t)12 << 30; size_t nb = budget_big / Q.wjb; if (nb < 1) nb = 1; Q.n_wjb = (int)std::min<size_t>(nb, (size_t)Q.w_wj); }
    Q.wjb_off = al256(Q.wj * (size_t)Q.w_wj);
    const size_t wj_bytes = Q.wjb_off + Q.wjb * (size_t)Q.n_wjb;
    Q.bytes = std::max(std::max(Q.chain * (size_t)Q.w_chain, Q.fill * (size_t)Q.w_fill), wj_bytes);
    return Q;
}

// The main pass of the batch in slot `T` as the launch sequence of hp_phase.h, with the launch resources of slot `Ln`.
static int launch_phased(lamsa_hp_handle *h, AlignState *S, Slot &T, Slot &Ln, OutDev &O, hipEvent_t e0, hipEvent_t e1)
{
    const int n = T.n_reads;
    const int64_t n_hits = T.n_hits;
    const SlabPlan Q = slab_plan(h, T.max_L, T.max_H, S->n_fifo + S->n_res > 0, chain_shape(h->para, T.n_reads, T.n_bases));
    const int w_chain = Q.w_chain, w_fill = Q.w_fill, w_dp = Q.w_dp, w_wj = Q.w_wj;
    const PhasedLayout Y = phased_layout(n, n_hits, T.n_bases, O.stream_cap);
    const int unit_cap = Y.unit_cap, lj_cap = Y.lj_cap; const int64_t fl_cap = Y.fl_cap, line_cap = Y.line_cap, job_cap = Y.job_cap;
    const size_t off = Y.bytes, o_args = Y.o[0], o_ctl = Y.o[1], o_meta = Y.o[2], o_nd = Y.o[3], o_ns = Y.o[4], o_sx = Y.o[5], o_un = Y.o[6], o_bq = Y.o[7], o_fl = Y.o[8], o_ln = Y.o[9],
                 o_jb = Y.o[10], o_lj = Y.o[11], o_lq = Y.o[12], o_wj = Y.o[13], o_wq = Y.o[14];
    if (grow(h, Ln.slab, Q.bytes) || grow(h, Ln.pers, off) || Ln.misc.ensure(256)) { h->err = "hipMalloc(slab)"; return LAMSA_HP_ENOMEM; }
    char *d = (char *)Ln.pers.p;
    PhaseArgs a;
    a.P = h->para;
    a.ref.pac = h->d_pac; a.ref.l_pac = h->l_pac; a.ref.n_seqs = h->n_seqs; a.ref.seq_off = h->d_seq_off; a.ref.seq_len = h->d_seq_len;
    a.in = T.in;
    a.out.read_out_off = O.off(); a.out.read_out_len = O.len(n); a.out.read_status = O.st(n); a.out.read_tbases = O.tb(n); a.out.read_work = O.work(n); a.out.stream = O.stream(n);
    a.out.stream_cap = O.stream_cap; a.out.cursor = (unsigned long long *)((char *)Ln.misc.p + 64); a.out.diag = O.diag(n);
    a.slab = (char *)Ln.slab.p; a.slab_per_wave = Q.chain; a.slab_fill = Q.fill; a.slab_wj = Q.wj; a.slab_wjb = Q.wjb; a.wjb_off = Q.wjb_off; a.n_wjb = Q.n_wjb; a.sort_pb = T.sort_pb; a.sort_cb = T.sort_cb;
    a.order = T.d_order; a.n_reads = n; a.prof = nullptr;
#ifdef HP_PROF
    if (Ln.prof.ensure(sizeof(long long) * 64 * 2 * ((size_t)n + 1))) { h->err = "hipMalloc(prof)"; return LAMSA_HP_ENOMEM; }
    a.prof = (long long *)Ln.prof.p;
    HIPCHK(h, hipMemsetAsync(Ln.prof.p, 0, sizeof(long long) * 64 * 2 * ((size_t)n + 1), Ln.cs), LAMSA_HP_EKERNEL);
#endif
    a.g_nd = (NodeS *)(d + o_nd); a.g_nseed = (int32_t *)(d + o_ns); a.g_sidx = (int32_t *)(d + o_sx); a.meta = (RdMeta *)(d + o_meta);
    a.units = (UnitRec *)(d + o_un); a.unit_cap = unit_cap; a.bucket_q = (int32_t *)(d + o_bq);
    a.fl_base = (int32_t *)(d + o_fl); a.fl_cap = fl_cap; a.line_base = (int32_t *)(d + o_ln); a.line_cap = line_cap; a.ctl = (PhaseCtl *)(d + o_ctl);
    a.job_base = (int32_t *)(d + o_jb); a.job_cap = job_cap;
    a.ljobs = (LjRec *)(d + o_lj); a.lj_bucket = (int32_t *)(d + o_lq); a.lj_cap = g_nolane ? 0 : lj_cap;
    a.wjobs = (WjRec *)(d + o_wj); a.wj_bucket = (int32_t *)(d + o_wq); a.wj_cap = g_nowave ? 0 : Y.wj_cap;
    const bool list = !g_nolane || !g_nowave;
    hipStream_t s = Ln.cs;
    HIPCHK(h, hipMemsetAsync(Ln.misc.p, 0, 128, s), LAMSA_HP_EKERNEL);
    HIPCHK(h, hipMemsetAsync(d + o_ctl, 0, (o_meta - o_ctl) + sizeof(RdMeta) * ((size_t)n + 1), s), LAMSA_HP_EKERNEL);      // counters + per-read state
    if (Ln.args_host.ensure(sizeof a)) { h->err = "hipHostMalloc(args)"; return LAMSA_HP_ENOMEM; }
    memcpy(Ln.args_host.p, &a, sizeof a);                   // the slot's launch resources are not reused before this launch has been collected
    HIPCHK(h, hipMemcpyAsync(d + o_args, Ln.args_host.p, sizeof a, hipMemcpyHostToDevice, s), LAMSA_HP_EKERNEL);
    const PhaseArgs *da = (const PhaseArgs *)(d + o_args);
    HIPCHK(h, hipEventRecord(e0, s), LAMSA_HP_EKERNEL);
    hipLaunchKernelGGL(g_chain_shapes[Q.shape].k1, dim3(std::min(w_chain, n)), dim3(64), 0, s, da);
    HIPCHK(h, hipEventRecord(Ln.ep[0], s), LAMSA_HP_EKERNEL);
    if (list) {
        hipLaunchKernelGGL(k_filllist, dim3(w_fill), dim3(64), 0, s, da, 0);
        HIPCHK(h, hipEventRecord(Ln.ep[5], s), LAMSA_HP_EKERNEL);
        if (!g_nowave) hipLaunchKernelGGL(k_filldp_wave, dim3(w_wj), dim3(64), 0, s, da, 0);       // the long jobs first: the short ones fill the SIMDs its tail leaves idle
        HIPCHK(h, hipEventRecord(Ln.ep[6], s), LAMSA_HP_EKERNEL);
        if (!g_nolane) hipLaunchKernelGGL(k_filldp_small, dim3(w_dp), dim3(64), 0, s, da, 0);
    } else { HIPCHK(h, hipEventRecord(Ln.ep[5], s), LAMSA_HP_EKERNEL); HIPCHK(h, hipEventRecord(Ln.ep[6], s), LAMSA_HP_EKERNEL); }
    HIPCHK(h, hipEventRecord(Ln.ep[4], s), LAMSA_HP_EKERNEL);
    hipLaunchKernelGGL(k_fill, dim3(w_fill), dim3(64), 0, s, da, 0);
    HIPCHK(h, hipEventRecord(Ln.ep[1], s), LAMSA_HP_EKERNEL);
    hipLaunchKernelGGL(g_chain_shapes[Q.shape].k2, dim3(std::min(w_chain, n)), dim3(64), 0, s, da);
    HIPCHK(h, hipEventRecord(Ln.ep[2], s), LAMSA_HP_EKERNEL);
    if (list) {
        HIPCHK(h, hipMemsetAsync(&((PhaseCtl *)(d + o_ctl))->q_head[5], 0, 16, s), LAMSA_HP_EKERNEL);          // the four queue heads of the listing and DP launches
        hipLaunchKernelGGL(k_filllist, dim3(w_fill), dim3(64), 0, s, da, 1);
        if (!g_nowave) hipLaunchKernelGGL(k_filldp_wave, dim3(w_wj), dim3(64), 0, s, da, 1);
        if (!g_nolane) hipLaunchKernelGGL(k_filldp_small, dim3(w_dp), dim3(64), 0, s, da, 1);
    }
    hipLaunchKernelGGL(k_fill, dim3(w_fill), dim3(64), 0, s, da, 1);
    HIPCHK(h, hipEventRecord(Ln.ep[3], s), LAMSA_HP_EKERNEL);
    hipLaunchKernelGGL(k_publish, dim3(std::min(h->n_cu * 8, n)), dim3(64), 0, s, da);
    HIPCHK(h, hipGetLastError(), LAMSA_HP_EKERNEL);
    HIPCHK(h, hipEventRecord(e1, s), LAMSA_HP_EKERNEL);
    Ln.phased = true;
    return LAMSA_HP_OK;
}

// one launch over `n_units` reads of the batch in slot `T` (order list on the device) with the launch resources
// (scratch slab, queue head, stream, events) of slot `Ln`; results into `O`.  The kernel
// is queued on the compute stream between the events e0/e1; `wait` blocks until it has finished.
static int launch_align(lamsa_hp_handle *h, AlignState *S, Slot &T, Slot &Ln, OutDev &O, const int32_t *d_order, int n_units, int scale, int max_L, int max_H,
                        hipEvent_t e0, hipEvent_t e1, bool wait)
{
    size_t slab_per_wave = slab_bytes_for(h->para, max_L, max_H, scale);
    if (scale == 1 && h->scratch_limit && slab_per_wave > h->scratch_limit) slab_per_wave = al256(h->scratch_limit);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_align_batch, 64, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    int n_waves = h->n_cu * per_cu;
    if (n_waves > n_units) n_waves = n_units;
    n_waves = cap_waves(n_waves, slab_per_wave, h->n_cu);
    if (grow(h, Ln.slab, slab_per_wave * (size_t)n_waves) || Ln.misc.ensure(256)) { h->err = "hipMalloc(slab)"; return LAMSA_HP_ENOMEM; }
    const int n = T.n_reads;
    AlignArgs a;
    a.P = h->para;
    a.ref.pac = h->d_pac; a.ref.l_pac = h->l_pac; a.ref.n_seqs = h->n_seqs; a.ref.seq_off = h->d_seq_off; a.ref.seq_len = h->d_seq_len;
    a.in = T.in;
    a.out.read_out_off = O.off(); a.out.read_out_len = O.len(n); a.out.read_status = O.st(n); a.out.read_tbases = O.tb(n); a.out.read_work = O.work(n); a.out.stream = O.stream(n);
    a.out.stream_cap = O.stream_cap; a.out.cursor = (unsigned long long *)((char *)Ln.misc.p + 64); a.out.diag = nullptr;
    a.slab = (char *)Ln.slab.p; a.slab_per_wave = slab_per_wave; a.counter = (int32_t *)Ln.misc.p;
    a.order = d_order; a.n_units = n_units; a.scale = scale; a.prof = nullptr; a.sort_pb = T.sort_pb; a.sort_cb = T.sort_cb;
    // Two main passes may be queued at a time, one per slot and stream.  The earlier one holds every wave slot until its
    // read queue runs empty; the later one's workgroups are dispatched as those slots fall free, i.e. it takes over the
    // SIMDs exactly as the earlier one's tail leaves them.  (Holding the later launch back explicitly, with
    // hipStreamWaitValue32 on a flag the earlier kernel sets, measured the same and could hang under tools that
    // serialise dispatches, so it is not done.)
    hipStream_t s = Ln.cs;
    HIPCHK(h, hipMemsetAsync(Ln.misc.p, 0, 128, s), LAMSA_HP_EKERNEL);
    HIPCHK(h, hipEventRecord(e0, s), LAMSA_HP_EKERNEL);
    hipLaunchKernelGGL(k_align_batch, dim3(n_waves), dim3(64), 0, s, a);
    HIPCHK(h, hipGetLastError(), LAMSA_HP_EKERNEL);
    HIPCHK(h, hipEventRecord(e1, s), LAMSA_HP_EKERNEL);
    if (wait) HIPCHK(h, hipStreamSynchronize(s), LAMSA_HP_EKERNEL);
#ifdef HP_PROF
    prof_report(T, a.prof, n);
#endif
    return LAMSA_HP_OK;
}

static int slot_events(lamsa_hp_handle *h, Slot &Ln)
{
    Ln.cs = &Ln == &state_of(h)->slot[1] ? h->stream_b : h->stream;
    if (!Ln.e0) HIPCHK(h, hipEventCreate(&Ln.e0), LAMSA_HP_EKERNEL);
    if (!Ln.e1) HIPCHK(h, hipEventCreate(&Ln.e1), LAMSA_HP_EKERNEL);
    for (hipEvent_t &e : Ln.ep) if (!e) HIPCHK(h, hipEventCreate(&e), LAMSA_HP_EKERNEL);
    return LAMSA_HP_OK;
}

// queue the main pass of the batch in slot `T`: every read, costliest first
static int start_main(lamsa_hp_handle *h, AlignState *S, Slot &T, Slot &Ln)
{
    int rc = slot_events(h, Ln);
    if (rc) return rc;
    const int n = T.n_reads;
    if (n == 0) return LAMSA_HP_OK;
    if (Ln.out1.ensure(n, main_stream_cap(n, T.n_bases))) { h->err = "hipMalloc(out)"; return LAMSA_HP_ENOMEM; }
    Ln.phased = false;
    if (!g_mono) return launch_phased(h, S, T, Ln, Ln.out1, Ln.e0, Ln.e1);
    return launch_align(h, S, T, Ln, Ln.out1, T.d_order, n, 1, T.max_L, T.max_H, Ln.e0, Ln.e1, false);
}

#define DL(dst, src, bytes) HIPCHK(h, hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, h->copy_stream), LAMSA_HP_EKERNEL)
#define DLSYNC() HIPCHK(h, hipStreamSynchronize(h->copy_stream), LAMSA_HP_EKERNEL)

// wait for the main pass of slot `T`, re-run the reads that overflowed, fetch the results
static int finish_main(lamsa_hp_handle *h, AlignState *S, Slot &T, Slot &Ln, lamsa_hp_result *R)
{
    const int n = T.n_reads;
    for (float &v : h->kernel_ms) v = 0;
    S->r_st.assign((size_t)n + 1, 0); S->r_off.assign((size_t)n + 1, 0); S->r_len.assign((size_t)n + 1, 0); S->r_tb.assign((size_t)n + 1, 0); S->r_work.assign(4 * (size_t)n + 4, 0);   // never empty: pointers stay valid
    if (S->stream.ensure(64)) { h->err = "hipHostMalloc(results)"; return LAMSA_HP_ENOMEM; }
    if (n == 0) {
        if (R) { R->stream = (const int32_t *)S->stream.p; R->stream_words = 0; R->read_off = S->r_off.data(); R->read_len = S->r_len.data(); R->read_status = S->r_st.data(); R->read_tbases = S->r_tb.data(); R->read_work = S->r_work.data(); }
        return LAMSA_HP_OK;
    }
    const double t_0 = now_s();
    HIPCHK(h, hipEventSynchronize(Ln.e1), LAMSA_HP_EKERNEL);
    const double t_1 = now_s();
    hipEventElapsedTime(&h->kernel_ms[0], Ln.e0, Ln.e1);
    if (Ln.phased) {                                     // the five launches of the main pass: chain1, fill, chain2, fill, publish
        hipEvent_t ev[6] = {Ln.e0, Ln.ep[0], Ln.ep[1], Ln.ep[2], Ln.ep[3], Ln.e1};
        for (int k = 0; k < 5; ++k) hipEventElapsedTime(&h->kernel_ms[2 + k], ev[k], ev[k + 1]);
        const unsigned long long *dg = Ln.out1.h_diag(n);          // drain of the four long launches: first wave out -> last wave out (100 MHz ticks)
        for (int k = 0; k < 4; ++k) h->kernel_ms[7 + k] = dg[2 * k + 1] >= dg[2 * k] && dg[2 * k + 1] ? (float)((double)(dg[2 * k + 1] - dg[2 * k]) * 1e-5) : 0.f;
        h->kernel_ms[11] = (float)dg[8]; h->kernel_ms[12] = (float)dg[9];                // fill units of the two rounds
        hipEventElapsedTime(&h->kernel_ms[13], Ln.ep[0], Ln.ep[4]);                         // of "fill1" (kernel_ms[3]): listing + the two DP launches
        hipEventElapsedTime(&h->kernel_ms[14], Ln.ep[0], Ln.ep[5]);                         // ... the listing
        hipEventElapsedTime(&h->kernel_ms[15], Ln.ep[5], Ln.ep[6]);                         // ... the wave-per-job DP launch
        h->kernel_ms[16] = (float)dg[12]; h->kernel_ms[17] = (float)((double)dg[13] * 1e-6); h->kernel_ms[18] = (float)dg[14]; h->kernel_ms[19] = (float)((double)dg[15] * 4e-6); h->kernel_ms[20] = (float)((double)dg[16] * 1e-6);   // wave jobs, their algorithmic MB, lane jobs, MB of CIGARs computed ahead
#ifdef HP_PROF
        prof_report(T, (const long long *)Ln.prof.p, n);
#endif
    }
    // the per-read arrays are in mapped host memory already (OutDev); the slot may be reused while the caller still
    // reads the results, so they are copied out
    memcpy(S->r_off.data(), Ln.out1.h_off(), 8 * (size_t)n); memcpy(S->r_len.data(), Ln.out1.h_len(n), 4 * (size_t)n);
    memcpy(S->r_st.data(), Ln.out1.h_st(n), 4 * (size_t)n); memcpy(S->r_tb.data(), Ln.out1.h_tb(n), 4 * (size_t)n); memcpy(S->r_work.data(), Ln.out1.h_work(n), 16 * (size_t)n);
    unsigned long long used1 = 0;                        // the arena is handed out front to back: its fill is the largest end
    for (int r = 0; r < n; ++r) if (S->r_off[r] >= 0 && (unsigned long long)(S->r_off[r] + S->r_len[r]) > used1) used1 = (unsigned long long)(S->r_off[r] + S->r_len[r]);
    if ((int64_t)used1 > Ln.out1.stream_cap) used1 = (unsigned long long)Ln.out1.stream_cap;
    // ---- retry pass: reads whose work buffers (or the stream arena) were too small -- outliers; 8x capacities.
    // On this slot's own stream: it runs beside the other slot's main pass when streaming.
    std::vector<int32_t> again;
    for (int r = 0; r < n; ++r) if ((S->r_st[r] & LAMSA_HP_ST_OVERFLOW) || S->r_off[r] < 0) again.push_back(r);
    unsigned long long used2 = 0;
    if (!again.empty()) {
        int mL = 0, mH = 0; int64_t cap2 = 1024;
        for (int r : again) { mL = std::max(mL, T.h_len[r]); mH = std::max(mH, T.h_H[r]); cap2 += 64 + 12LL * 8 * T.h_len[r]; }
        if (grow(h, S->out2.buf, 4 * (size_t)cap2 + 256) || S->out2.host.ensure(OutDev::hdr(n) + 256) || grow(h, S->retry_list, 4 * again.size())) { h->err = "hipMalloc(retry)"; return LAMSA_HP_ENOMEM; }
        S->out2.stream_cap = cap2;
        HIPCHK(h, hipMemcpyAsync(S->retry_list.p, again.data(), 4 * again.size(), hipMemcpyHostToDevice, Ln.cs), LAMSA_HP_EKERNEL);
        int rc = launch_align(h, S, T, Ln, S->out2, (const int32_t *)S->retry_list.p, (int)again.size(), 8, mL, mH, h->ev0, h->ev1, true);
        if (rc) return rc;
        hipEventElapsedTime(&h->kernel_ms[1], h->ev0, h->ev1);
        const int64_t *off2 = S->out2.h_off(); const int32_t *len2 = S->out2.h_len(n), *st2 = S->out2.h_st(n), *tb2 = S->out2.h_tb(n), *wk2 = S->out2.h_work(n);
        for (int r : again) if (off2[r] >= 0 && (unsigned long long)(off2[r] + len2[r]) > used2) used2 = (unsigned long long)(off2[r] + len2[r]);
        if ((int64_t)used2 > cap2) used2 = (unsigned long long)cap2;
        for (int r : again) { S->r_st[r] = st2[r]; S->r_len[r] = len2[r]; S->r_tb[r] = tb2[r]; for (int q = 0; q < 4; ++q) S->r_work[4 * r + q] = wk2[4 * r + q]; S->r_off[r] = off2[r] < 0 ? -1 : (int64_t)used1 + off2[r]; }
    }
    for (int r = 0; r < n; ++r) if (S->r_off[r] < 0) { S->r_off[r] = 0; S->r_len[r] = 0; S->r_st[r] |= LAMSA_HP_ST_OVERFLOW; }
    if (!R) return LAMSA_HP_OK;
    if (S->stream.ensure(4 * (size_t)(used1 + used2) + 64)) { h->err = "hipHostMalloc(results)"; return LAMSA_HP_ENOMEM; }
    int32_t *hs = (int32_t *)S->stream.p;
    if (used1) DL(hs, Ln.out1.stream(n), 4 * (size_t)used1);
    if (used2) DL(hs + used1, S->out2.stream(n), 4 * (size_t)used2);
    DLSYNC();
    if (g_trace) fprintf(stderr, "[lamsa_hp] collect: waited %.1f ms for the kernel (%.1f ms), results %.1f ms (%.2f GB)\n", 1e3 * (t_1 - t_0), h->kernel_ms[0], 1e3 * (now_s() - t_1), 4e-9 * (double)(used1 + used2));
    R->stream = hs; R->stream_words = (int64_t)(used1 + used2); R->read_off = S->r_off.data(); R->read_len = S->r_len.data(); R->read_status = S->r_st.data(); R->read_tbases = S->r_tb.data(); R->read_work = S->r_work.data();
    return LAMSA_HP_OK;
}
#undef DL
#undef DLSYNC

static int check_batch_args(lamsa_hp_handle *h, const lamsa_hp_batch *B)
{
    if (!h || !B || B->n_reads < 0) return LAMSA_HP_EINVAL;
    if (!h->d_pac) { h->err = "handle was created without a reference"; return LAMSA_HP_EINVAL; }
    HIPCHK(h, hipSetDevice(h->device), LAMSA_HP_ENODEV);
    return LAMSA_HP_OK;
}

extern "C" int lamsa_hp_upload_batch(lamsa_hp_handle *h, const lamsa_hp_batch *B)
{
    int rc = check_batch_args(h, B);
    if (rc) return rc;
    AlignState *S = state_of(h);
    if (S->n_fifo || S->n_res) { h->err = "batches are in flight: collect them first"; return LAMSA_HP_EINVAL; }
    return upload_into(h, &S->slot[0], B);
}

extern "C" int lamsa_hp_run_uploaded(lamsa_hp_handle *h, lamsa_hp_result *R)
{
    if (!h) return LAMSA_HP_EINVAL;
    AlignState *S = state_of(h);
    if (S->n_fifo) { h->err = "submitted batches are in flight: collect them first"; return LAMSA_HP_EINVAL; }
    if (!S->slot[0].valid) { h->err = "no batch uploaded"; return LAMSA_HP_EINVAL; }
    HIPCHK(h, hipSetDevice(h->device), LAMSA_HP_ENODEV);
    if (S->n_res) { h->err = "runs of the resident batch are in flight: finish them first"; return LAMSA_HP_EINVAL; }
    int rc = start_main(h, S, S->slot[0], S->slot[0]);
    if (rc) return rc;
    return finish_main(h, S, S->slot[0], S->slot[0], R);
}

// Runs of the resident batch, two deep: a run uses the inputs of slot 0 and the launch resources of either slot.
extern "C" int lamsa_hp_start_uploaded(lamsa_hp_handle *h)
{
    if (!h) return LAMSA_HP_EINVAL;
    AlignState *S = state_of(h);
    if (S->n_fifo) { h->err = "submitted batches are in flight: collect them first"; return LAMSA_HP_EINVAL; }
    if (!S->slot[0].valid) { h->err = "no batch uploaded"; return LAMSA_HP_EINVAL; }
    if (S->n_res >= 2) { h->err = "two runs are already in flight: finish one first"; return LAMSA_HP_EINVAL; }
    HIPCHK(h, hipSetDevice(h->device), LAMSA_HP_ENODEV);
    const int k = S->n_res == 1 ? 1 - S->res_fifo[0] : 0;
    S->slot[1].valid = false;                                  // its buffers serve as a lane now
    const int rc = start_main(h, S, S->slot[0], S->slot[k]);
    if (rc) return rc;
    S->res_fifo[S->n_res++] = k;
    return LAMSA_HP_OK;
}

extern "C" int lamsa_hp_finish_uploaded(lamsa_hp_handle *h, lamsa_hp_result *R)
{
    if (!h) return LAMSA_HP_EINVAL;
    AlignState *S = state_of(h);
    if (S->n_res == 0) { h->err = "no run in flight"; return LAMSA_HP_EINVAL; }
    HIPCHK(h, hipSetDevice(h->device), LAMSA_HP_ENODEV);
    const int k = S->res_fifo[0];
    S->res_fifo[0] = S->res_fifo[1]; --S->n_res;
    return finish_main(h, S, S->slot[0], S->slot[k], R);
}

extern "C" int lamsa_hp_align_batch(lamsa_hp_handle *h, const lamsa_hp_batch *B, lamsa_hp_result *R)
{
    int rc = lamsa_hp_upload_batch(h, B);
    if (rc) return rc;
    return lamsa_hp_run_uploaded(h, R);
}

extern "C" int lamsa_hp_submit_batch(lamsa_hp_handle *h, const lamsa_hp_batch *B)
{
    int rc = check_batch_args(h, B);
    if (rc) return rc;
    AlignState *S = state_of(h);
    if (S->n_res) { h->err = "runs of the resident batch are in flight: finish them first"; return LAMSA_HP_EINVAL; }
    if (S->n_fifo >= 2) { h->err = "two batches are already in flight: collect one first"; return LAMSA_HP_EINVAL; }
    const int k = S->n_fifo == 1 ? 1 - S->fifo[0] : 0;         // the slot no queued kernel reads
    rc = upload_into(h, &S->slot[k], B);                         // overlaps the kernel of the other slot
    if (rc) return rc;
    const double t_s = now_s();
    rc = start_main(h, S, S->slot[k], S->slot[k]);
    if (g_trace) fprintf(stderr, "[lamsa_hp] submit: launch queued in %.1f ms (scratch and output buffers are allocated on first use)\n", 1e3 * (now_s() - t_s));
    if (rc) { S->slot[k].valid = false; return rc; }
    S->fifo[S->n_fifo++] = k;
    return LAMSA_HP_OK;
}

extern "C" int lamsa_hp_collect_batch(lamsa_hp_handle *h, lamsa_hp_result *R)
{
    if (!h) return LAMSA_HP_EINVAL;
    AlignState *S = state_of(h);
    if (S->n_fifo == 0) { h->err = "no batch in flight"; return LAMSA_HP_EINVAL; }
    HIPCHK(h, hipSetDevice(h->device), LAMSA_HP_ENODEV);
    const int k = S->fifo[0];
    S->fifo[0] = S->fifo[1]; --S->n_fifo;
    const int rc = finish_main(h, S, S->slot[k], S->slot[k], R);
    S->slot[k].valid = false;
    return rc;
}

extern "C" void *lamsa_hp_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
extern "C" void lamsa_hp_host_free(void *p) { if (p) hipHostFree(p); }

// Allocate, ahead of the first batch, what batches of up to n_reads reads / n_bases bases / n_hits hits / n_cig seed-CIGAR elements
// with reads of up to max_read_len bases need on the device, for both batches that can be in flight: device allocations of tens of
// GB take seconds, and without this call they happen inside the first submit / align call.  Later batches within these bounds
// allocate nothing.
extern "C" int lamsa_hp_reserve(lamsa_hp_handle *h, int32_t n_reads, int64_t n_bases, int64_t n_hits, int64_t n_cig, int32_t max_read_len, int32_t max_hits_per_read)
{
    if (!h || n_reads < 0 || n_bases < 0 || n_hits < 0 || n_cig < 0) return LAMSA_HP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device), LAMSA_HP_ENODEV);
    AlignState *S = state_of(h);
    if (S->n_fifo || S->n_res) { h->err = "batches are in flight"; return LAMSA_HP_EINVAL; }
    const SlabPlan Q = slab_plan(h, max_read_len, max_hits_per_read, false, chain_shape(h->para, n_reads, n_bases));
    const int64_t cap = main_stream_cap(n_reads, n_bases);
    const PhasedLayout Y = phased_layout(n_reads, n_hits, n_bases, cap);
    // the packed input of a batch (upload_into): every array plus its 256-byte alignment, both CIGAR forms' staging included
    const size_t in_bytes = (size_t)n_bases + 16 * (size_t)n_reads * 4 + 32 * ((size_t)n_hits + n_reads) + 8 * (size_t)n_hits + 5 * (size_t)n_cig + 64 * 256 + 4096;
    for (Slot &T : S->slot) {
        int rc = slot_events(h, T);
        if (rc) return rc;
        if (T.slab.ensure(Q.bytes) || T.pers.ensure(Y.bytes) || T.misc.ensure(256) || T.bin.ensure(in_bytes) || T.out1.ensure(n_reads, cap)) {
            for (Slot &U : S->slot) { U.slab.release(); U.pers.release(); U.bin.release(); U.out1.release(); }       // nothing half reserved stays behind: the first batch sizes its own buffers
            h->err = "hipMalloc(reserve)"; return LAMSA_HP_ENOMEM;
        }
    }
    return LAMSA_HP_OK;
}

extern "C" int lamsa_hp_set_scratch_limit(lamsa_hp_handle *h, size_t bytes)
{
    if (!h || (bytes && bytes < ((size_t)64 << 10))) return LAMSA_HP_EINVAL;
    h->scratch_limit = bytes;
    return LAMSA_HP_OK;
}
