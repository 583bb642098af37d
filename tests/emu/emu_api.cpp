// emu_api.cpp -- TEST INFRASTRUCTURE: compiles the device sources (lamsa_amd/csrc/hp_*.h)
// against the CPU lane emulation (tests/emu/hp/wave.h) so that `-m "not gpu"` tests and
// sanitizers can run the kernel algorithms on the CPU.  Not part of the product library.
#include <stdlib.h>
#include <string.h>
#include <vector>
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
    for (int j = 0; j < n; ++j) dp_run_job(a, j, 0);
    return 0;
}
