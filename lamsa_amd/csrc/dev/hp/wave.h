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
#define HP_HD __host__ __device__ inline         // parameter checks that the host side of the API makes too
// routines called once per line / per track of a read (~100 calls per read) are inlined into the routine of their phase: a call costs its
// frame -- the callee-saved registers of the callee, the live ones of the caller -- through scratch on EVERY call (measured on k_chain1:
// 40 % of the kernel's HBM writes were such frames).  The phases themselves stay HP_NOINL, one call per read each: one function holding
// the whole chaining path spills in its hot loops instead (measured: 479 spilled VGPRs, 25 % slower).
#define HP_HOT __device__ __forceinline__

// pointers that the hot loops dereference are cast to the global address space so that hipcc emits
// global_load/global_store instead of flat_* (which also wait on the LDS counter)
#define HP_G __attribute__((address_space(1)))
#define HP_L __attribute__((address_space(3)))     // LDS

typedef int hp_v4i __attribute__((ext_vector_type(4)));
typedef int hp_v2i __attribute__((ext_vector_type(2)));
// whole-vector loads: the compiler cannot sink individual members of a record behind later branches
HP_INL void hp_load16(const HP_G void *p, int *o) { hp_v4i v = *(const HP_G hp_v4i *)p; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
HP_INL void hp_load8(const HP_G void *p, int *o) { hp_v2i v = *(const HP_G hp_v2i *)p; o[0] = v.x; o[1] = v.y; }
HP_INL void hp_store16(HP_G void *p, int a, int b, int c, int d) { hp_v4i v; v.x = a; v.y = b; v.z = c; v.w = d; *(HP_G hp_v4i *)p = v; }

namespace wv {

constexpr int W = 64;

// LDS read-modify-write without a returned value (ds_and_b32): lanes of one wave may target the same dword
HP_INL void lds_and(HP_L int *p, int mask) { (void)__hip_atomic_fetch_and(p, mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
HP_INL void lds_or(HP_L int *p, int mask) { (void)__hip_atomic_fetch_or(p, mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }

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
HP_INL unsigned long long wall() { return (unsigned long long)wall_clock64(); }       // constant-rate counter (100 MHz): launch drain accounting
HP_INL int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
HP_INL long long uni64(long long v) {
    int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
    int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((long long)hi << 32) | (unsigned)lo;
}

HP_INL int bcast(const Lane<int> &x, int src) { return __builtin_amdgcn_readlane(x.v, src); }
// every lane reads x of the lane it names (ds_bpermute_b32: the LDS crossbar, no memory)
HP_INL Lane<int> gather(const Lane<int> &x, const Lane<int> &from) { Lane<int> o; o.v = __builtin_amdgcn_ds_bpermute(from.v << 2, x.v); return o; }
// lane `dst` of x takes the wave-uniform value v
HP_INL void setlane(Lane<int> &x, int dst, int v) { x.v = lane() == dst ? v : x.v; }

HP_INL unsigned long long ballot(const Lane<int> &p) { return __ballot(p.v != 0); }

// Cross-lane reductions and the prefix scan use DPP row operations (no LDS round trip):
//   quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140,
//   row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143, wave_shr:1 = 0x138  (GFX9 DPP controls)
template <int CTRL, int ROW_MASK = 0xf>
HP_INL int dpp(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false); }

// v = max(v, v[source lane]) in ONE VALU instruction: v_max_i32 with a DPP source operand.  Lanes whose DPP source is
// invalid or masked off keep v.  The s_nop covers the "VALU write -> DPP read of the same VGPR" hazard (2 wait states),
// which the assembler does not insert inside inline asm.
#define HP_MAX_DPP(v, CTRL) asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 " CTRL : "+v"(v))

HP_INL int reduce_max(const Lane<int> &x) {
    int v = x.v;
    HP_MAX_DPP(v, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    HP_MAX_DPP(v, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    HP_MAX_DPP(v, "row_half_mirror row_mask:0xf bank_mask:0xf");
    HP_MAX_DPP(v, "row_mirror row_mask:0xf bank_mask:0xf");          // every lane of a 16-lane row now holds the row maximum
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const int ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
HP_INL int reduce_sum(const Lane<int> &x) {
    int v = x.v;
    v += dpp<0xB1>(v, v); v += dpp<0x4E>(v, v); v += dpp<0x141>(v, v); v += dpp<0x140>(v, v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
HP_INL long long reduce_max64(const Lane<long long> &x) {
    int hi = (int)(x.v >> 32); unsigned lo = (unsigned)(x.v & 0xffffffffll);
#define HP_STEP64(C) { int oh = dpp<C>(hi, hi); unsigned ol = (unsigned)dpp<C>((int)lo, (int)lo); const bool g = oh > hi || (oh == hi && ol > lo); hi = g ? oh : hi; lo = g ? ol : lo; }
    HP_STEP64(0xB1) HP_STEP64(0x4E) HP_STEP64(0x141) HP_STEP64(0x140)
#undef HP_STEP64
    long long best = ((long long)__builtin_amdgcn_readlane(hi, 0) << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const long long c = ((long long)__builtin_amdgcn_readlane(hi, r) << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, r);
        best = c > best ? c : best;
    }
    return best;
}
// exclusive prefix max over lanes; lane 0 receives `ident` (which must be <= every input)
HP_INL void scan_max_excl(Lane<int> &x, int ident) {
    int v = x.v;
    HP_MAX_DPP(v, "row_shr:1 row_mask:0xf bank_mask:0xf");           // Hillis-Steele inside each 16-lane row
    HP_MAX_DPP(v, "row_shr:2 row_mask:0xf bank_mask:0xf");
    HP_MAX_DPP(v, "row_shr:4 row_mask:0xf bank_mask:0xf");
    HP_MAX_DPP(v, "row_shr:8 row_mask:0xf bank_mask:0xf");
    HP_MAX_DPP(v, "row_bcast:15 row_mask:0xa bank_mask:0xf");        // row 0 -> row 1, row 2 -> row 3
    HP_MAX_DPP(v, "row_bcast:31 row_mask:0xc bank_mask:0xf");        // rows 0-1 -> rows 2,3
    x.v = dpp<0x138>(ident, v);                        // shift the inclusive scan right by one lane
}

// the same, and the maximum of all 64 inputs is returned (lane 63 of the inclusive scan)
HP_INL int scan_max_excl_top(Lane<int> &x, int ident) {
    int v = x.v;
    HP_MAX_DPP(v, "row_shr:1 row_mask:0xf bank_mask:0xf");
    HP_MAX_DPP(v, "row_shr:2 row_mask:0xf bank_mask:0xf");
    HP_MAX_DPP(v, "row_shr:4 row_mask:0xf bank_mask:0xf");
    HP_MAX_DPP(v, "row_shr:8 row_mask:0xf bank_mask:0xf");
    HP_MAX_DPP(v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    HP_MAX_DPP(v, "row_bcast:31 row_mask:0xc bank_mask:0xf");
    x.v = dpp<0x138>(ident, v);
    return __builtin_amdgcn_readlane(v, 63);
}

// every lane receives the value of the lane below it; lane 0 receives `fill`
HP_INL void shr1(Lane<int> &x, int fill) { x.v = dpp<0x138>(fill, x.v); }
// the same round the wave: lane 0 receives lane 63's (wave_ror:1)
HP_INL void ror1(Lane<int> &x) { x.v = dpp<0x13C>(x.v, x.v); }


// exclusive prefix sum over lanes (lane 0 receives 0)
HP_INL void scan_add_excl(Lane<int> &x) {
    int v = x.v;
    v += dpp<0x111>(0, v);                             // Hillis-Steele inside each 16-lane row
    v += dpp<0x112>(0, v);
    v += dpp<0x114>(0, v);
    v += dpp<0x118>(0, v);
    v += dpp<0x142, 0xA>(0, v);                        // row 0 -> row 1, row 2 -> row 3
    v += dpp<0x143, 0xC>(0, v);                        // rows 0-1 -> rows 2,3
    x.v = dpp<0x138>(0, v);                            // shift the inclusive scan right by one lane
}

}  // namespace wv

// ---- two int16 values per 32-bit register (VOP3P packed math: v_pk_add_i16, v_pk_max_i16, ...): the banded DP of hp_ksw.h keeps two
// columns per lane where its scores fit.  Arithmetic wraps like the hardware's; the callers stay far from the ends of the range.
namespace pk {
typedef short v2 __attribute__((ext_vector_type(2)));
typedef unsigned short u2 __attribute__((ext_vector_type(2)));
HP_INL v2 V(int x) { return __builtin_bit_cast(v2, x); }
HP_INL u2 U(int x) { return __builtin_bit_cast(u2, x); }
HP_INL int I(v2 x) { return __builtin_bit_cast(int, x); }
HP_INL int I(u2 x) { return __builtin_bit_cast(int, x); }
HP_INL int add(int a, int b) { return I(V(a) + V(b)); }
HP_INL int sub(int a, int b) { return I(V(a) - V(b)); }
HP_INL int max(int a, int b) { return I(__builtin_elementwise_max(V(a), V(b))); }
HP_INL int min_u(int a, int b) { return I(__builtin_elementwise_min(U(a), U(b))); }
HP_INL int mul(int a, int b) { return I(V(a) * V(b)); }                                  // low 16 bits of each product
HP_INL int neg_mask(int a) { return I(V(a) >> (v2){15, 15}); }                           // per half: 0xFFFF where negative, else 0
HP_INL int rep(int x) { return (x & 0xffff) | (int)((unsigned)x << 16); }                // both halves = (short)x
HP_INL int lo(int a) { return (int)(short)(a & 0xffff); }                                // sign-extended halves
HP_INL int hi(int a) { return a >> 16; }
HP_INL int pack(int l, int h) { return (l & 0xffff) | (int)((unsigned)h << 16); }
HP_INL int sel(int mask, int a, int b) { return (a & mask) | (b & ~mask); }              // bitwise: a where the mask is set
// columns shifted up by one: half 1 <- half 0, half 0 <- half 1 of the lane below (`below`: the register of lane l - 1)
HP_INL int shift_up(int x, int below) { return (int)(((unsigned)x << 16) | ((unsigned)below >> 16)); }
}  // namespace pk

