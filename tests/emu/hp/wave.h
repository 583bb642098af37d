// CPU lane-emulation twin of lamsa_amd/csrc/dev/hp/wave.h -- TEST INFRASTRUCTURE ONLY.
// It lets the *same kernel sources* (hp_*.h) be compiled by g++ so the `-m "not gpu"`
// tests and ASan/UBSan can exercise the device algorithms on the CPU.  It is never
// part of the shipped library: the C-ABI library is built from the HIP header only.
#pragma once
#include <stdint.h>
#include <string.h>

#define HP_FN  static
#define HP_INL static inline
#define HP_NOINL static
#define HP_HD static inline
#define HP_HOT static inline

#define HP_G
#define HP_L

struct hp_v2i { int x, y; };
struct hp_v4i { int x, y, z, w; };
HP_INL void hp_load16(const void *p, int *o) { memcpy(o, p, 16); }
HP_INL void hp_load8(const void *p, int *o) { memcpy(o, p, 8); }
HP_INL void hp_store16(void *p, int a, int b, int c, int d) { int v[4] = {a, b, c, d}; memcpy(p, v, 16); }

namespace wv {

constexpr int W = 64;

HP_INL void lds_and(int *p, int mask) { *p &= mask; }
HP_INL void lds_or(int *p, int mask) { *p |= mask; }

template <class T> struct Lane {
    T v[64];
    T &operator[](int l) { return v[l]; }
    const T &operator[](int l) const { return v[l]; }
};

#define WAVE_FOR(l) for (int l = 0; l < 64; ++l)

HP_INL void sync() {}
HP_INL bool leader() { return true; }
HP_INL long long clock() { return 0; }
HP_INL unsigned long long wall() { return 0; }
HP_INL int uni(int v) { return v; }
HP_INL long long uni64(long long v) { return v; }
HP_INL int bcast(const Lane<int> &x, int src) { return x.v[src]; }
HP_INL void setlane(Lane<int> &x, int dst, int v) { x.v[dst] = v; }
HP_INL Lane<int> gather(const Lane<int> &x, const Lane<int> &from) { Lane<int> o; for (int l = 0; l < 64; ++l) o.v[l] = x.v[from.v[l] & 63]; return o; }
HP_INL unsigned long long ballot(const Lane<int> &p) {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) if (p.v[l]) m |= 1ull << l;
    return m;
}
HP_INL int reduce_max(const Lane<int> &x) { int v = x.v[0]; for (int l = 1; l < 64; ++l) v = x.v[l] > v ? x.v[l] : v; return v; }
HP_INL int reduce_sum(const Lane<int> &x) { int v = 0; for (int l = 0; l < 64; ++l) v += x.v[l]; return v; }
HP_INL long long reduce_max64(const Lane<long long> &x) { long long v = x.v[0]; for (int l = 1; l < 64; ++l) v = x.v[l] > v ? x.v[l] : v; return v; }
// exclusive prefix max over lanes; lane 0 receives `ident` (not folded into the others)
HP_INL void scan_max_excl(Lane<int> &x, int ident) {
    int run = 0;
    for (int l = 0; l < 64; ++l) {
        int cur = x.v[l];
        x.v[l] = l == 0 ? ident : run;
        run = l == 0 ? cur : (cur > run ? cur : run);
    }
}
HP_INL int scan_max_excl_top(Lane<int> &x, int ident) {
    int run = 0;
    for (int l = 0; l < 64; ++l) {
        int cur = x.v[l];
        x.v[l] = l == 0 ? ident : run;
        run = l == 0 ? cur : (cur > run ? cur : run);
    }
    return run;
}

HP_INL void shr1(Lane<int> &x, int fill) { for (int l = 63; l > 0; --l) x.v[l] = x.v[l - 1]; x.v[0] = fill; }
HP_INL void ror1(Lane<int> &x) { const int top = x.v[63]; for (int l = 63; l > 0; --l) x.v[l] = x.v[l - 1]; x.v[0] = top; }
HP_INL void scan_add_excl(Lane<int> &x) {
    int run = 0;
    for (int l = 0; l < 64; ++l) { const int cur = x.v[l]; x.v[l] = run; run += cur; }
}

}  // namespace wv

// two int16 values per 32-bit word, as the packed math of the device build (VOP3P) computes them
namespace pk {
HP_INL int pack(int l, int h) { return (l & 0xffff) | (int)((unsigned)h << 16); }
HP_INL int lo(int a) { return (int)(short)(a & 0xffff); }
HP_INL int hi(int a) { return a >> 16; }
HP_INL int add(int a, int b) { return pack((short)(lo(a) + lo(b)), (short)(hi(a) + hi(b))); }
HP_INL int sub(int a, int b) { return pack((short)(lo(a) - lo(b)), (short)(hi(a) - hi(b))); }
HP_INL int max(int a, int b) { return pack(lo(a) > lo(b) ? lo(a) : lo(b), hi(a) > hi(b) ? hi(a) : hi(b)); }
HP_INL int min_u(int a, int b) { const unsigned al = a & 0xffff, bl = b & 0xffff, ah = (unsigned)a >> 16, bh = (unsigned)b >> 16; return (int)((al < bl ? al : bl) | ((ah < bh ? ah : bh) << 16)); }
HP_INL int mul(int a, int b) { return pack((short)(lo(a) * lo(b)), (short)(hi(a) * hi(b))); }
HP_INL int neg_mask(int a) { return pack(lo(a) < 0 ? -1 : 0, hi(a) < 0 ? -1 : 0); }
HP_INL int rep(int x) { return (x & 0xffff) | (int)((unsigned)x << 16); }
HP_INL int sel(int mask, int a, int b) { return (a & mask) | (b & ~mask); }
HP_INL int shift_up(int x, int below) { return (int)(((unsigned)x << 16) | ((unsigned)below >> 16)); }
}  // namespace pk


// single-threaded stand-ins for the two device atomics the kernels use
static inline int atomicAdd(int *p, int v) { int o = *p; *p += v; return o; }
static inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }
static inline int atomicOr(int *p, int v) { int o = *p; *p |= v; return o; }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }

