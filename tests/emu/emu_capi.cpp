// emu_capi.cpp -- TEST INFRASTRUCTURE: the C-ABI of include/lamsa_hp.h implemented on the CPU lane emulation,
// so that the host program (lamsa_amd/host/*.cpp) can be linked into tests/_build/lamsa_emu and its file IO, GEM
// parsing, ranking and SAM writer tested without a GPU.  Never linked into the product (lamsa_amd/bin/lamsa links
// liblamsa_hp.so and has no CPU path).
#include <deque>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "hp_para.h"

extern "C" int emu_align_batch(const lamsa_hp_para *P, const lamsa_hp_ref *ref, const lamsa_hp_batch *B, int scale, size_t slab_bytes,
                               int32_t *stream, int64_t stream_cap, int64_t *n_words, int64_t *read_off, int32_t *read_len, int32_t *status);

extern "C" int emu_dp_batch(const lamsa_hp_para *P, int n, const uint8_t *seq,
                            const int64_t *q_off, const int32_t *qlen, const int64_t *t_off, const int32_t *tlen,
                            const int32_t *kind, const int32_t *w, const int32_t *h0,
                            int32_t *score, int32_t *qle, int32_t *tle, int32_t *status,
                            int32_t *cig_n, const int64_t *cig_cap_off, int32_t *cig, size_t slab_bytes);

struct lamsa_hp_handle {
    lamsa_hp_para P; lamsa_hp_ref ref; std::string err;
    std::vector<int32_t> d_score, d_qle, d_tle, d_status, d_cig; std::vector<int64_t> d_off;      // lamsa_hp_dp_batch results
    std::vector<int32_t> stream, len, status, tb, work; std::vector<int64_t> off;
    struct Done { std::vector<int32_t> stream, len, status, tb; std::vector<int64_t> off; };
    std::deque<Done> fifo;                 // lamsa_hp_submit_batch computes at once; lamsa_hp_collect_batch hands the oldest out
};

extern "C" int lamsa_hp_create(lamsa_hp_handle **out, const lamsa_hp_para *para, const lamsa_hp_ref *ref, int)
{
    lamsa_hp_handle *h = new lamsa_hp_handle; h->P = *para;
    if (ref) h->ref = *ref; else memset(&h->ref, 0, sizeof h->ref);
    *out = h; return LAMSA_HP_OK;
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
    res->read_status = h->status.data(); res->read_tbases = h->tb.data(); h->work.assign(4 * h->tb.size() + 4, 0); res->read_work = h->work.data();
    return LAMSA_HP_OK;
}

// the streaming form: same two-deep queue discipline and error returns as the product
extern "C" int lamsa_hp_submit_batch(lamsa_hp_handle *h, const lamsa_hp_batch *B)
{
    if (h->fifo.size() >= 2) { h->err = "two batches are already in flight: collect one first"; return LAMSA_HP_EINVAL; }
    // the vectors handed out by the last collect stay untouched (the caller may still be reading them, as with the product):
    // they are parked while this batch is computed into fresh ones
    lamsa_hp_handle::Done keep;
    keep.stream.swap(h->stream); keep.len.swap(h->len); keep.status.swap(h->status); keep.tb.swap(h->tb); keep.off.swap(h->off);
    lamsa_hp_result r;
    const int rc = lamsa_hp_align_batch(h, B, &r);
    h->fifo.emplace_back();
    lamsa_hp_handle::Done &d = h->fifo.back();
    d.stream.swap(h->stream); d.len.swap(h->len); d.status.swap(h->status); d.tb.swap(h->tb); d.off.swap(h->off);
    h->stream.swap(keep.stream); h->len.swap(keep.len); h->status.swap(keep.status); h->tb.swap(keep.tb); h->off.swap(keep.off);
    if (rc) { h->fifo.pop_back(); return rc; }
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
    res->read_off = h->off.data(); res->read_len = h->len.data(); res->read_status = h->status.data(); res->read_tbases = h->tb.data(); h->work.assign(4 * h->tb.size() + 4, 0); res->read_work = h->work.data();
    return LAMSA_HP_OK;
}
extern "C" int lamsa_hp_reserve(lamsa_hp_handle *, int32_t, int64_t, int64_t, int64_t, int32_t, int32_t) { return LAMSA_HP_OK; }
extern "C" void *lamsa_hp_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
extern "C" void lamsa_hp_host_free(void *p) { free(p); }

extern "C" int lamsa_hp_dp_batch(lamsa_hp_handle *h, const lamsa_hp_dp_jobs *J, lamsa_hp_dp_out *O)
{
    const int n = J->n_jobs;
    std::vector<int64_t> cap((size_t)n + 1, 0);
    size_t need = 1 << 20;
    for (int i = 0; i < n; ++i) { cap[i + 1] = cap[i] + J->qlen[i] + J->tlen[i] + 8; const size_t z = (size_t)(J->qlen[i] + 64) * (size_t)(J->tlen[i] + 64) + (1 << 16); if (z > need) need = z; }
    std::vector<int32_t> cn((size_t)n + 1, 0), raw((size_t)cap[n] + 4, 0);
    h->d_score.assign((size_t)n + 1, 0); h->d_qle.assign((size_t)n + 1, 0); h->d_tle.assign((size_t)n + 1, 0); h->d_status.assign((size_t)n + 1, 0);
    emu_dp_batch(&h->P, n, J->seq, J->q_off, J->qlen, J->t_off, J->tlen, J->kind, J->w, J->h0, h->d_score.data(), h->d_qle.data(), h->d_tle.data(), h->d_status.data(),
                 cn.data(), cap.data(), raw.data(), need + (1 << 20));
    h->d_off.assign((size_t)n + 1, 0); h->d_cig.clear();
    for (int i = 0; i < n; ++i) { h->d_off[i] = (int64_t)h->d_cig.size(); h->d_cig.insert(h->d_cig.end(), raw.begin() + cap[i], raw.begin() + cap[i] + cn[i]); }
    h->d_off[n] = (int64_t)h->d_cig.size();
    if (h->d_cig.empty()) h->d_cig.push_back(0);
    O->score = h->d_score.data(); O->qle = h->d_qle.data(); O->tle = h->d_tle.data(); O->status = h->d_status.data(); O->cig_off = h->d_off.data(); O->cigar = h->d_cig.data();
    return LAMSA_HP_OK;
}
