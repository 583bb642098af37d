// emu_capi.cpp -- TEST INFRASTRUCTURE: the C-ABI of include/lamsa_hp.h implemented on the CPU lane emulation,
// so that the host program (lamsa_amd/host/*.cpp) can be linked into tests/_build/lamsa_emu and its file IO, GEM
// parsing, ranking and SAM writer tested without a GPU.  Never linked into the product (lamsa_amd/bin/lamsa links
// liblamsa_hp.so and has no CPU path).
#include <deque>
#include <stdlib.h>
#include <string>
#include <vector>
#include "hp_para.h"

extern "C" int emu_align_batch(const lamsa_hp_para *P, const lamsa_hp_ref *ref, const lamsa_hp_batch *B, int scale, size_t slab_bytes,
                               int32_t *stream, int64_t stream_cap, int64_t *n_words, int64_t *read_off, int32_t *read_len, int32_t *status);

struct lamsa_hp_handle {
    lamsa_hp_para P; lamsa_hp_ref ref; std::string err;
    std::vector<int32_t> stream, len, status, tb; std::vector<int64_t> off;
    struct Done { std::vector<int32_t> stream, len, status, tb; std::vector<int64_t> off; };
    std::deque<Done> fifo;                 // lamsa_hp_submit_batch computes at once; lamsa_hp_collect_batch hands the oldest out
};

extern "C" int lamsa_hp_create(lamsa_hp_handle **out, const lamsa_hp_para *para, const lamsa_hp_ref *ref, int)
{
    lamsa_hp_handle *h = new lamsa_hp_handle; h->P = *para; h->ref = *ref; *out = h; return LAMSA_HP_OK;
}
extern "C" void lamsa_hp_destroy(lamsa_hp_handle *h) { delete h; }
extern "C" const char *lamsa_hp_last_error(const lamsa_hp_handle *h) { return h ? h->err.c_str() : "null handle"; }
extern "C" float lamsa_hp_last_kernel_ms(const lamsa_hp_handle *, int) { return 0.f; }

extern "C" int lamsa_hp_align_batch(lamsa_hp_handle *h, const lamsa_hp_batch *B, lamsa_hp_result *res)
{
    const int n = B->n_reads;
    h->off.assign(n + 1, 0); h->len.assign(n + 1, 0); h->status.assign(n + 1, 0); h->tb.assign(n + 1, 0);
    std::vector<std::vector<int32_t>> per(n);
    {   // first pass: the whole batch at scale 1
        const int64_t L = n ? B->read_off[n] : 0, nh = n ? B->hit_off[B->seed_off[n]] : 0;
        const int64_t cap = 4096 * (int64_t)(n + 1) + 64 * L + 512 * nh;
        std::vector<int32_t> s((size_t)cap); int64_t nw = 0;
        emu_align_batch(&h->P, &h->ref, B, 1, (size_t)96 << 20, s.data(), cap, &nw, h->off.data(), h->len.data(), h->status.data());
        for (int r = 0; r < n; ++r) per[r].assign(s.begin() + h->off[r], s.begin() + h->off[r] + h->len[r]);
    }
    for (int r = 0; r < n; ++r) {                                          // same retry rule as the product: overflowed reads again at scale 8
        if (!(h->status[r] & LAMSA_HP_ST_OVERFLOW)) continue;
        lamsa_hp_batch one = *B;                                           // a one-read view: the offset arrays shifted, the data arrays shared
        one.n_reads = 1; one.read_off = B->read_off + r; one.seed_all = B->seed_all + r; one.last_len = B->last_len + r; one.seed_off = B->seed_off + r;
        const int64_t L = B->read_off[r + 1] - B->read_off[r], nh = B->hit_off[B->seed_off[r + 1]] - B->hit_off[B->seed_off[r]];
        const int64_t cap = 4096 + 512 * L + 4096 * nh;
        std::vector<int32_t> s((size_t)cap); int64_t nw = 0, o = 0; int32_t ln = 0, st = 0;
        emu_align_batch(&h->P, &h->ref, &one, 8, (size_t)768 << 20, s.data(), cap, &nw, &o, &ln, &st);
        h->status[r] = st; per[r].assign(s.begin() + o, s.begin() + o + ln);
    }
    h->stream.clear();
    for (int r = 0; r < n; ++r) { h->off[r] = (int64_t)h->stream.size(); h->len[r] = (int32_t)per[r].size(); h->stream.insert(h->stream.end(), per[r].begin(), per[r].end()); }
    res->stream = h->stream.data(); res->stream_words = (int64_t)h->stream.size(); res->read_off = h->off.data(); res->read_len = h->len.data();
    res->read_status = h->status.data(); res->read_tbases = h->tb.data();
    return LAMSA_HP_OK;
}

// the streaming form: same two-deep queue discipline and error returns as the product
extern "C" int lamsa_hp_submit_batch(lamsa_hp_handle *h, const lamsa_hp_batch *B)
{
    if (h->fifo.size() >= 2) { h->err = "two batches are already in flight: collect one first"; return LAMSA_HP_EINVAL; }
    lamsa_hp_result r;
    const int rc = lamsa_hp_align_batch(h, B, &r);
    if (rc) return rc;
    h->fifo.emplace_back();
    lamsa_hp_handle::Done &d = h->fifo.back();
    d.stream.swap(h->stream); d.len.swap(h->len); d.status.swap(h->status); d.tb.swap(h->tb); d.off.swap(h->off);
    return LAMSA_HP_OK;
}
extern "C" int lamsa_hp_collect_batch(lamsa_hp_handle *h, lamsa_hp_result *res)
{
    if (h->fifo.empty()) { h->err = "no batch in flight"; return LAMSA_HP_EINVAL; }
    lamsa_hp_handle::Done &d = h->fifo.front();
    h->stream.swap(d.stream); h->len.swap(d.len); h->status.swap(d.status); h->tb.swap(d.tb); h->off.swap(d.off);
    h->fifo.pop_front();
    if (h->stream.empty()) h->stream.push_back(0);
    res->stream = h->stream.data(); res->stream_words = 0; for (size_t r = 0; r + 1 < h->len.size(); ++r) res->stream_words += h->len[r];
    res->read_off = h->off.data(); res->read_len = h->len.data(); res->read_status = h->status.data(); res->read_tbases = h->tb.data();
    return LAMSA_HP_OK;
}
extern "C" void *lamsa_hp_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
extern "C" void lamsa_hp_host_free(void *p) { free(p); }
