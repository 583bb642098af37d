// hp_sort.h -- the per-read sort index of the seed hits, built by the read's own wave before it chains.
//
// srt[i] = local index of the hit that is i-th in the order (contig, strand, reference position, hit index), rnk =
// the inverse permutation.  The chaining code uses it to visit only the predecessors that can be connected at all
// (same contig and strand, within the SV / read-span window) instead of every earlier hit -- an exact pruning of
// frag_dp_update's scan (src/lamsa_dp_con.c:713-751): a hit outside that window is F_CHR_DIF or F_UNCONNECT for
// get_fseed_dis (:607,:613-633) and is skipped there too.  The reference has no such index.
//
// Bitonic network in which EVERY compare-exchange is ascending (the first step of each merge pairs i with its mirror
// image in the block, the remaining steps are half-cleaners), so that the virtual elements beyond the last hit
// (+infinity) never move and no padding to a power of two is needed.  A stage's pairs are disjoint; stages are
// separated by a wave-level fence.  The elements are single 64-bit words (contig/strand | position | hit index, the
// field widths chosen per batch and read).  Lists of up to HP_SORT_BLOCK hits sort entirely in LDS; longer ones live in the
// wave's HBM scratch and pass through LDS block by block, only the far stages of the late merges touch HBM directly
// (6 sweeps instead of 78 for 4 096 hits).  Keys too wide to pack sort as (key, index) pairs in HBM.
#pragma once
#include "hp_core.h"

namespace hp {

HP_INL uint64_t hit_sort_key(int32_t chr, int8_t strand, int64_t pos)
{
    return ((uint64_t)((uint32_t)chr * 2u + (strand > 0 ? 1u : 0u)) << 40) | ((uint64_t)pos & ((1ull << 40) - 1));
}

HP_INL int bits_of(unsigned long long x) { return x ? 64 - (int)__builtin_clzll(x) : 0; }

#ifndef HP_SORT_BLOCK
#define HP_SORT_BLOCK 1024               // words sorted / merged in LDS at a time (8 KB of the wave's 9.5 KB)
#endif

// one stage of the network on 64-bit words (WP: pointer into LDS or HBM): pair t has lo = t with a 0 bit inserted at
// bit log2(j) and hi = lo ^ flip; four pairs per lane in flight
template <class WP> HP_INL void bitonic_stage(WP w, int H, int j, int flip)
{
    for (int t0 = 0;; t0 += 4 * wv::W) {
        if ((((t0 & ~(j - 1)) << 1) | (t0 & (j - 1))) >= H) break;      // lo grows with t: nothing left in this stage
        WAVE_FOR(l) {
            uint64_t a[4], b[4]; int lo[4], hi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + u * wv::W + l;
                lo[u] = ((t & ~(j - 1)) << 1) | (t & (j - 1)); hi[u] = lo[u] ^ flip;
                const bool ok = hi[u] < H;           // hi > lo always; hi >= H is +infinity: in place already
                a[u] = ok ? w[lo[u]] : 0ull; b[u] = ok ? w[hi[u]] : ~0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (a[u] > b[u]) { w[lo[u]] = b[u]; w[hi[u]] = a[u]; }
        }
    }
    wv::sync();
}

// block [b0, b0 + Hb) of the packed words: HBM -> LDS, four loads per lane in flight
HP_INL void block_load(HP_L uint64_t *lw, const HP_G uint64_t *work, int b0, int Hb)
{
    for (int i = 0; i < Hb; i += 4 * wv::W) WAVE_FOR(l) {
        uint64_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int k = i + u * wv::W + l; v[u] = k < Hb ? work[b0 + k] : 0ull; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int k = i + u * wv::W + l; if (k < Hb) lw[k] = v[u]; }
    }
    wv::sync();
}
// the sorted block leaves LDS: as packed words back to HBM, or -- after the last merge -- as the index itself
HP_INL void block_store(const HP_L uint64_t *lw, HP_G uint64_t *work, int b0, int Hb, bool final, int ib, HP_G int32_t *srt, HP_G int32_t *rnk)
{
    for (int i = 0; i < Hb; i += wv::W) WAVE_FOR(l) {
        const int k = i + l;
        if (k < Hb) {
            const uint64_t v = lw[k];
            if (final) { const int s = (int)(v & ((1ull << ib) - 1)); srt[b0 + k] = s; if (rnk) rnk[s] = b0 + k; }
            else work[b0 + k] = v;
        }
    }
    wv::sync();
}

// the same network on (key, index) pairs in HBM: keys too wide to pack
HP_FN void bitonic_pairs(HP_G uint64_t *key, HP_G int32_t *srt, int H)
{
    for (int k = 2; (k >> 1) < H; k <<= 1) {
        for (int j = k >> 1; j >= 1; j >>= 1) {
            const int flip = j == (k >> 1) ? k - 1 : j;
            for (int t0 = 0;; t0 += wv::W) {
                if ((((t0 & ~(j - 1)) << 1) | (t0 & (j - 1))) >= H) break;
                WAVE_FOR(l) {
                    const int t = t0 + l, lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo ^ flip;
                    if (hi < H) {
                        const uint64_t ka = key[lo], kb = key[hi];
                        const int32_t ia = srt[lo], ib = srt[hi];
                        if (ka > kb || (ka == kb && ia > ib)) { key[lo] = kb; key[hi] = ka; srt[lo] = ib; srt[hi] = ia; }
                    }
                }
            }
            wv::sync();
        }
    }
}

// The network on packed words (major key << ib | element index), `major(k)` < 2^major_bits evaluated per lane.  srt[i] = the
// element that is i-th in (major, index) order; rnk (may be null) the inverse.  work: 8*H bytes of scratch in HBM, lw: lds_n
// 64-bit words of LDS owned by this wave.  Returns false when the words would not fit 64 bits (the caller then sorts pairs).
template <class MajorFn>
HP_INL bool sort_packed(MajorFn major, int major_bits, int H, HP_G int32_t *srt, HP_G int32_t *rnk, HP_G uint64_t *work, HP_L uint64_t *lw, int lds_n)
{
    const int ib = H > 1 ? bits_of((unsigned)(H - 1)) : 0, C = HP_SORT_BLOCK;
    if (!(major_bits + ib <= 64 && lds_n >= C)) return false;
    // every block: packed straight into LDS, sorted there
    const bool one = H <= C;
    for (int b0 = 0; b0 < H; b0 += C) {
        const int Hb = H - b0 < C ? H - b0 : C;
        for (int i = 0; i < Hb; i += 4 * wv::W) WAVE_FOR(l) {
            uint64_t kk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int k = b0 + i + u * wv::W + l; kk[u] = k < H ? major(k) : 0ull; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = i + u * wv::W + l;
                if (k < Hb) lw[k] = (kk[u] << ib) | (uint64_t)(b0 + k);
            }
        }
        wv::sync();
        for (int k = 2; (k >> 1) < Hb; k <<= 1)
            for (int j = k >> 1; j >= 1; j >>= 1) bitonic_stage(lw, Hb, j, j == (k >> 1) ? k - 1 : j);
        block_store(lw, work, b0, Hb, one, ib, srt, rnk);
    }
    // merges of blocks of k/2 into blocks of k: the far stages (distance >= C) in HBM, the rest block by block in LDS
    for (int k = 2 * C; (k >> 1) < H; k <<= 1) {
        for (int j = k >> 1; j >= C; j >>= 1) bitonic_stage(work, H, j, j == (k >> 1) ? k - 1 : j);
        for (int b0 = 0; b0 < H; b0 += C) {
            const int Hb = H - b0 < C ? H - b0 : C;
            block_load(lw, work, b0, Hb);
            for (int j = C >> 1; j >= 1; j >>= 1) bitonic_stage(lw, Hb, j, j);
            block_store(lw, work, b0, Hb, k >= H, ib, srt, rnk);
        }
    }
    return true;
}

// pos/chr/strand: the H hits of one read.  srt_/rnk_: H entries each.  work_: 8*H bytes of scratch in HBM.
// lw: lds_n 64-bit words of LDS owned by this wave.  pb / cb: bits of the largest position / of the largest
// contig*2+strand code in the batch (host, from the validation pass).
HP_NOINL void sort_read_hits(const int64_t *pos_, const int32_t *chr_, const int8_t *strand_, int H, int32_t *srt_, int32_t *rnk_, uint64_t *work_,
                             HP_L uint64_t *lw, int lds_n, int pb, int cb)
{
    H = wv::uni(H); pb = wv::uni(pb); cb = wv::uni(cb);
    HP_G uint64_t *work = (HP_G uint64_t *)work_;
    HP_G int32_t *srt = (HP_G int32_t *)srt_, *rnk = (HP_G int32_t *)rnk_;
    const HP_G int64_t *pos = (const HP_G int64_t *)pos_;
    const HP_G int32_t *chr = (const HP_G int32_t *)chr_;
    const HP_G int8_t *strand = (const HP_G int8_t *)strand_;
    const int ib = H > 1 ? bits_of((unsigned)(H - 1)) : 0;
    if (pb + ib <= 62 &&
        sort_packed([&](int k) { const uint64_t kk = hit_sort_key(chr[k], strand[k], pos[k]); return ((kk >> 40) << pb) | (kk & ((1ull << 40) - 1)); },
                    pb + cb, H, srt, rnk, work, lw, lds_n)) return;
    for (int b = 0; b < H; b += wv::W) WAVE_FOR(l) {
        const int k = b + l;
        if (k < H) { work[k] = hit_sort_key(chr[k], strand[k], pos[k]); srt[k] = k; }
    }
    wv::sync();
    bitonic_pairs(work, srt, H);
    for (int b = 0; b < H; b += wv::W) WAVE_FOR(l) {
        const int i = b + l;
        if (i < H) rnk[srt[i]] = i;
    }
    wv::sync();
}

}  // namespace hp
