// hp_core.h -- per-wave context, bump arena, bounded CIGAR vectors, sequence views.
// Device code (compiled by hipcc for gfx950; the tests also compile it with the CPU
// lane emulation of tests/emu/hp/wave.h).  Wave-uniform unless a comment says otherwise.
#pragma once
#include <hp/wave.h>
#include "../../include/lamsa_hp.h"

namespace hp {

typedef int32_t cig_t;                       // len<<4|op, signed like the reference (src/lamsa_aln.h:210)
enum { C_M = 0, C_I = 1, C_D = 2, C_N = 3, C_S = 4, C_H = 5 };
enum { ST_OVERFLOW = LAMSA_HP_ST_OVERFLOW, ST_REFEXIT = LAMSA_HP_ST_REFEXIT, ST_UNSUPPORTED = LAMSA_HP_ST_UNSUPPORTED,
       ST_DEAD = ST_OVERFLOW | ST_REFEXIT | ST_UNSUPPORTED };     // ST_DEAD: the read has no result

#define HP_NEG_INF (-0x40000000)             // MINUS_INF, src/ksw.c:504

// Bump allocator over a slab of HBM owned by this wave.  Stack discipline via mark/release.
struct Arena {
    char *base; size_t cap, top;
};
HP_INL void arena_init(Arena &a, void *base, size_t cap) { a.base = (char *)base; a.cap = cap; a.top = 0; }
HP_INL size_t arena_mark(const Arena &a) { return a.top; }
HP_INL void arena_release(Arena &a, size_t m) { a.top = m; }

// LDS of one wave (= one workgroup): circular H and E rows, staged query bases, direction matrix (see hp_ksw.h).
// Sized for the fill kernel's 8 waves per SIMD = 32 workgroups per CU (the hardware's limit): 32 x 5 KB of the CU's 160 KB.  The fill
// kernel is bound by the latency each wave exposes, and every step up in occupancy paid even though it costs VGPRs (64 at 8 waves) and
// LDS for the direction matrix (measured, ms per step of 65 536 reads: 4 waves / 5 KB matrix 326, 5 / 3.5 KB 322, 6 / 2.1 KB 307,
// 7 / 1.2 KB 295, 8 / 0.5 KB 294 -- profiles/r02_variants_occupancy*.txt).  Bands up to w = 222 use the
// LDS rows (the presets use 10..200); wider ones fall back to rows in HBM.
#ifndef HP_LDS_CELLS
#define HP_LDS_CELLS 512
#endif
#ifndef HP_LDS_Z_BYTES
#define HP_LDS_Z_BYTES 512
#endif
#define HP_LDS_WORDS (2 * HP_LDS_CELLS + HP_LDS_CELLS / 4 + HP_LDS_Z_BYTES / 4)
#ifndef HP_CHAIN_LDS_WORDS
#define HP_CHAIN_LDS_WORDS 2432             // LDS words of a wave of the chaining kernels (>= HP_SORT_BLOCK * 2 for the hit sort; 16 waves per CU)
#endif
// LDS words of a wave that runs BOTH halves (the one-kernel form of the retry pass, the CPU emulation): enough for either
#define HP_BOTH_LDS_WORDS (HP_CHAIN_LDS_WORDS > HP_LDS_WORDS ? HP_CHAIN_LDS_WORDS : HP_LDS_WORDS)
// the lane strips of hp_cluster.h / hp_gaps.h (six words per entry and lane) are cut from the chaining kernels' LDS
#define HP_LANE_STRIP_WORDS HP_CHAIN_LDS_WORDS

struct Ctx {
    const lamsa_hp_para *P;   // parameters (kernel argument copy)
    HP_L int32_t *lds;        // the LDS owned by this wave: at least HP_LDS_WORDS words (the DP rows of hp_ksw.h need those)
    int lds_words;            // its size (the chaining kernels give a wave more: a cluster's node state lives there, hp_cluster.h)
    Arena tmp;                // scratch slab of this wave
    int status;               // ST_* bits for the unit (read / job) being processed
    long long n_cells;        // DP cells updated (accounting: GCUPS)
    long long *prof;          // per-read cycle counters of diagnostic builds (-DHP_PROF), else nullptr
    long long *prof_dp = nullptr;   // -DHP_PROF, wave-per-job launch only: per-routine cycle / cell counters of ksw_extend (else nullptr)
    int lds_epoch;            // bumped by every DP routine that writes the rows' part of the LDS (hp_ksw.h): what a caller parked there (frags_merge's staged CIGARs) is gone
};
// -DHP_PROF_FILL: slots 16 .. 23 time parts of frags_merge (hp_fill.h) instead of parts of the chaining (hp_chain.h, hp_gaps.h)
#ifdef HP_PROF_FILL
#define HP_PROF_CHAIN_ON 0
#define HP_TADD_FILL(cx, slot, v) HP_TADD(cx, slot, v)
#else
#define HP_PROF_CHAIN_ON 1
#define HP_TADD_FILL(cx, slot, v) do { } while (0)
#endif
#ifdef HP_PROF
#define HP_T0(v) const long long v = wv::clock()
#define HP_TADD(cx, slot, v) do { if ((cx).prof) { (cx).prof[slot] += wv::clock() - (v); (cx).prof[(slot) + 1] += 1; } } while (0)
#else
#define HP_T0(v) do { } while (0)
#define HP_TADD(cx, slot, v) do { } while (0)
#endif

// path counters of the tests' CPU build (tests/emu/emu_api.cpp defines HP_STAT): which variant of a routine a test has really run
#ifndef HP_DPLOG
#define HP_DPLOG(kind, qlen, tlen, w, cells) do { } while (0)      // tests' CPU build: one record per DP call (tools/dp_shapes.py)
#endif
#ifndef HP_STAT
#define HP_STAT(i) do { } while (0)
#endif

// returns nullptr (and flags overflow) when the slab is exhausted; callers must cope
HP_INL void *arena_alloc(Ctx &cx, size_t bytes) {
    size_t b = (bytes + 15) & ~(size_t)15;
    if (cx.tmp.top + b > cx.tmp.cap) { cx.status |= ST_OVERFLOW; return nullptr; }
    void *p = cx.tmp.base + cx.tmp.top;
    cx.tmp.top += b;
    return p;
}

// bounded CIGAR vector (storage elsewhere)
struct CigV { cig_t *c; int n, cap; };
HP_INL void cig_bind(CigV &v, cig_t *mem, int cap) { v.c = mem; v.n = 0; v.cap = mem ? cap : 0; }
HP_INL bool cig_alloc(Ctx &cx, CigV &v, int cap) {
    cig_t *m = (cig_t *)arena_alloc(cx, (size_t)cap * sizeof(cig_t));
    cig_bind(v, m, cap);
    return m != nullptr;
}
HP_INL void cig_raw_push(Ctx &cx, CigV &v, cig_t w) {
    if (v.n >= v.cap) { cx.status |= ST_OVERFLOW; return; }
    v.c[v.n++] = w;
}
// _push_cigar0, src/frag_check.h:136-151
HP_INL void cig_push0(Ctx &cx, CigV &v, cig_t w) {
    if (v.n > 0 && (v.c[v.n - 1] & 0xf) == (w & 0xf)) { v.c[v.n - 1] += (w >> 4) << 4; return; }
    cig_raw_push(cx, v, w);
}
// _push_cigar1, src/frag_check.h:153-156
HP_INL void cig_push1(Ctx &cx, CigV &v, cig_t w) { if ((w >> 4) == 0) return; cig_push0(cx, v, w); }
// _push_cigar, src/frag_check.h:158-184 (first word merges; I+S and S+I fuse into S).  The body is copied by the
// lanes (one coalesced pass); source and destination never overlap.
HP_FN void cig_pushv(Ctx &cx, CigV &v, const cig_t *c, int n) {
    if (n == 0) return;
    const HP_G cig_t *src = (const HP_G cig_t *)c;
    HP_G cig_t *dst = (HP_G cig_t *)v.c;
    const int vn = v.n;
    int j = 0;
    if (vn > 0) {
        const cig_t last = dst[vn - 1], c0 = src[0];
        if ((last & 0xf) == (c0 & 0xf)) { dst[vn - 1] = last + ((c0 >> 4) << 4); j = 1; }
        else if (((last & 0xf) == C_I && (c0 & 0xf) == C_S) || ((last & 0xf) == C_S && (c0 & 0xf) == C_I)) {
            dst[vn - 1] = (((last >> 4) + (c0 >> 4)) << 4) | C_S; j = 1;
        }
    }
    const int m = n - j;
    if (vn + m > v.cap) { cx.status |= ST_OVERFLOW; return; }
    for (int b0 = 0; b0 < m; b0 += 64) { WAVE_FOR(l) { const int i = b0 + l; if (i < m) dst[vn + i] = src[j + i]; } }
    v.n = vn + m;
    wv::sync();
}
// _push_cigar with a one-word source held in a register
HP_FN void cig_pushw(Ctx &cx, CigV &v, cig_t w) {
    if (v.n > 0) {
        const cig_t last = v.c[v.n - 1];
        if ((last & 0xf) == (w & 0xf)) { v.c[v.n - 1] = last + ((w >> 4) << 4); return; }
        if (((last & 0xf) == C_I && (w & 0xf) == C_S) || ((last & 0xf) == C_S && (w & 0xf) == C_I)) { v.c[v.n - 1] = (((last >> 4) + (w >> 4)) << 4) | C_S; return; }
    }
    if (v.n + 1 > v.cap) { cx.status |= ST_OVERFLOW; return; }
    v.c[v.n++] = w;
}
// _invert_cigar, src/frag_check.h:124
HP_FN void cig_invert(cig_t *c, int n) {
    HP_G cig_t *g = (HP_G cig_t *)c;
    for (int b0 = 0; b0 < n / 2; b0 += 64) {
        WAVE_FOR(l) { const int a = b0 + l; if (a < n / 2) { const cig_t x = g[a], y = g[n - 1 - a]; g[a] = y; g[n - 1 - a] = x; } }
    }
    wv::sync();
}
// readInCigar (M,I,S) / refInCigar (M,D,H), src/frag_check.c:179,205
HP_FN int cig_readlen(const cig_t *c, int n) {
    const HP_G cig_t *g = (const HP_G cig_t *)c;
    int tot = 0;
    for (int b0 = 0; b0 < n; b0 += 64) {
        wv::Lane<int> x;
        WAVE_FOR(l) { const int i = b0 + l; int v = 0; if (i < n) { const cig_t w = g[i]; const int op = w & 0xf; if (op == C_M || op == C_I || op == C_S) v = w >> 4; } x[l] = v; }
        tot += wv::reduce_sum(x);
    }
    return tot;
}
HP_FN int cig_reflen(const cig_t *c, int n) {
    const HP_G cig_t *g = (const HP_G cig_t *)c;
    int tot = 0;
    for (int b0 = 0; b0 < n; b0 += 64) {
        wv::Lane<int> x;
        WAVE_FOR(l) { const int i = b0 + l; int v = 0; if (i < n) { const cig_t w = g[i]; const int op = w & 0xf; if (op == C_M || op == C_D || op == C_H) v = w >> 4; } x[l] = v; }
        tot += wv::reduce_sum(x);
    }
    return tot;
}

// strided view of a base sequence (stride -1 = reversed), 1 byte/base codes 0..4
struct Seq { const uint8_t *p; int stride; };
HP_INL int seq_at(const Seq &s, int i) { return s.p[(long)i * s.stride]; }
HP_INL Seq seq_fwd(const uint8_t *p) { Seq s; s.p = p; s.stride = 1; return s; }
HP_INL Seq seq_rev(const uint8_t *p, int len) { Seq s; s.p = p + (len > 0 ? len - 1 : 0); s.stride = -1; return s; }

// substitution score: lamsa_fill_mat, src/lamsa_aln.c:1331-1340 (N row/column = -1)
HP_INL int sub_score(const lamsa_hp_para *P, int t, int q) { return (t > 3 || q > 3) ? -1 : (t == q ? P->match : -P->mis); }

HP_INL int iabs(int x) { return x < 0 ? -x : x; }
HP_INL int imin(int a, int b) { return a < b ? a : b; }
HP_INL int imax(int a, int b) { return a > b ? a : b; }

}  // namespace hp
