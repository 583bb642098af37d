// wave.h -- CDNA4 (gfx950) wavefront primitives used by every LAMSA hot-path kernel.
//
// Execution model of this code base: ONE READ PER 64-LANE WAVEFRONT (= one workgroup).
// Control logic is "wave-uniform": all 64 lanes execute the same scalar statements on
// the same values (so uniform stores are redundant same-address stores, never guarded),
// and the data-parallel inner loops (DP band rows, chaining predecessor scans, 2-bit
// unpacking, CIGAR walks) are written as WAVE_FOR(l) { ... } blocks in which `l` is the
// lane.  Values that live across a cross-lane primitive are wv::Lane<T>.
// No workgroup barriers are needed anywhere: a workgroup is a single wave.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HP_FN  __device__
#define HP_INL __device__ __forceinline__
#define HP_NOINL __device__ __noinline__

// pointers that the hot loops dereference are cast to the global address space so that hipcc emits
// global_load/global_store instead of flat_* (which also wait on the LDS counter)
#define HP_G __attribute__((address_space(1)))

typedef int hp_v4i __attribute__((ext_vector_type(4)));
typedef int hp_v2i __attribute__((ext_vector_type(2)));
// whole-vector loads: the compiler cannot sink individual members of a record behind later branches
HP_INL void hp_load16(const HP_G void *p, int *o) { hp_v4i v = *(const HP_G hp_v4i *)p; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
HP_INL void hp_load8(const HP_G void *p, int *o) { hp_v2i v = *(const HP_G hp_v2i *)p; o[0] = v.x; o[1] = v.y; }

namespace wv {

constexpr int W = 64;

HP_INL int lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <class T> struct Lane {
    T v;
    HP_INL T &operator[](int) { return v; }
    HP_INL const T &operator[](int) const { return v; }
};

// body runs once per lane; `l` is the lane index
#define WAVE_FOR(l) for (int l = wv::lane(), _wv_once = 1; _wv_once; _wv_once = 0)

// make stores of one lane visible to loads of the other lanes of this wave
HP_INL void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
HP_INL bool leader() { return lane() == 0; }
HP_INL long long clock() { return (long long)__builtin_readcyclecounter(); }     // diagnostic builds only (-DHP_PROF)
HP_INL int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
HP_INL long long uni64(long long v) {
    int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
    int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((long long)hi << 32) | (unsigned)lo;
}

HP_INL int bcast(const Lane<int> &x, int src) { return __builtin_amdgcn_readlane(x.v, src); }

HP_INL unsigned long long ballot(const Lane<int> &p) { return __ballot(p.v != 0); }

HP_INL int reduce_max(const Lane<int> &x) {
    int v = x.v;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { int o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return uni(v);
}
HP_INL int reduce_sum(const Lane<int> &x) {
    int v = x.v;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return uni(v);
}
HP_INL long long reduce_max64(const Lane<long long> &x) {
    long long v = x.v;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { long long o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return uni64(v);
}
// exclusive prefix max over lanes; lane 0 receives `ident`
HP_INL void scan_max_excl(Lane<int> &x, int ident) {
    int v = x.v;
    const int l = lane();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(v, d, 64); if (l >= d) v = o > v ? o : v; }
    int e = __shfl_up(v, 1, 64);
    x.v = l == 0 ? ident : e;
}

}  // namespace wv
