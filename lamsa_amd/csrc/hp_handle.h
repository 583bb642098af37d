// hp_handle.h -- the handle behind the C-ABI (host side, HIP runtime).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "../../include/lamsa_hp.h"

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return -1; }
        cap = want;
        return 0;
    }
    void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
};

struct lamsa_hp_handle {
    int device = 0;
    lamsa_hp_para para;
    hipStream_t stream = nullptr;          // compute: the DP batches and the align kernels of batch slot 0, in order
    hipStream_t stream_b = nullptr;        // compute: the align kernels of batch slot 1 (lamsa_hp_submit_batch alternates the slots)
    hipStream_t copy_stream = nullptr;     // batch uploads and result downloads, concurrent with the compute stream
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int n_cu = 256;
    // reference in HBM
    uint8_t *d_pac = nullptr; int64_t l_pac = 0; int32_t n_seqs = 0;
    int64_t *d_seq_off = nullptr; int32_t *d_seq_len = nullptr;
    // reusable device buffers
    DevBuf in, out, slab, misc, pac2;      // pac2: the targets of a lane-kind DP batch, 2 bits per base
    // host-side result storage (callee-owned outputs)
    std::vector<int32_t> h_i32; std::vector<int64_t> h_i64; std::vector<int32_t> h_cig;
    std::vector<int32_t> h_score, h_qle, h_tle, h_status;
    float kernel_ms[24] = {0};
    size_t scratch_limit = 0;      // lamsa_hp_set_scratch_limit
    std::string err;
};

#define HIPCHK(h, call, code) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (h)->err = std::string(#call) + ": " + hipGetErrorString(e_); return (code); } } while (0)

