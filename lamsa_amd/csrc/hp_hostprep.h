// hp_hostprep.h -- host-side preparation of a batch (plain C++, no HIP): for every read the permutation that
// sorts its seed hits by (contig, strand, reference position) and its inverse.  The chaining kernels use it
// to visit only the predecessors that can be connected at all (same contig and strand, within the SV / read
// span window) instead of every earlier hit -- an exact pruning of frag_dp_update's scan (src/lamsa_dp_con.c:713-751):
// a hit outside that window is F_CHR_DIF or F_UNCONNECT for get_fseed_dis (:607,:613-633) and is skipped there too.
#pragma once
#include <stdint.h>
#include <algorithm>
#include <thread>
#include <vector>

// fn(first, last) over [0, n) split into contiguous blocks, one per host thread (reads are independent)
template <class F> static inline void hp_parallel_blocks(int n, F fn)
{
    unsigned hw = std::thread::hardware_concurrency();
    int T = (int)(hw ? hw : 1); if (T > 32) T = 32;
    if (n < 256 || T < 2) { fn(0, n); return; }
    std::vector<std::thread> th;
    const int per = (n + T - 1) / T;
    for (int t = 0; t < T; ++t) { const int a = t * per, b = std::min(n, a + per); if (a < b) th.emplace_back([=]() { fn(a, b); }); }
    for (auto &x : th) x.join();
}

static inline void hp_build_sort_index(int n_reads, const int64_t *seed_off, const int64_t *hit_off,
                                       const int64_t *h_pos, const int32_t *h_chr, const int8_t *h_strand,
                                       std::vector<int32_t> &srt, std::vector<int32_t> &rnk)
{
    const int64_t n_hits = n_reads ? hit_off[seed_off[n_reads]] : 0;
    srt.assign((size_t)n_hits + 1, 0); rnk.assign((size_t)n_hits + 1, 0);
    hp_parallel_blocks(n_reads, [&](int r0, int r1) {
    std::vector<uint64_t> key;
    std::vector<int32_t> idx;
    for (int r = r0; r < r1; ++r) {
        const int64_t hb = hit_off[seed_off[r]], he = hit_off[seed_off[r + 1]];
        const int H = (int)(he - hb);
        key.resize((size_t)H); idx.resize((size_t)H);
        for (int k = 0; k < H; ++k) {
            key[k] = ((uint64_t)((uint32_t)h_chr[hb + k] * 2u + (h_strand[hb + k] > 0 ? 1u : 0u)) << 40) | ((uint64_t)h_pos[hb + k] & ((1ull << 40) - 1));
            idx[k] = k;
        }
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return key[a] < key[b]; });
        for (int i = 0; i < H; ++i) { srt[hb + i] = idx[i]; rnk[hb + idx[i]] = i; }
    }
    });
}
