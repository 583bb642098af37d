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
    static int32_t lds[HP_LDS_WORDS];
    for (int j = 0; j < n; ++j) dp_run_job(a, j, 0, lds);
    return 0;
}

// ---------------------------------------------------------------- whole per-read path, emulated
#include "hp_align.h"
#include "hp_hostprep.h"

extern "C" int emu_align_batch(const lamsa_hp_para *P, const lamsa_hp_ref *ref, const lamsa_hp_batch *B, int scale, size_t slab_bytes,
                               int32_t *stream, int64_t stream_cap, int64_t *n_words, int64_t *read_off, int32_t *read_len, int32_t *status)
{
    AlignArgs a;
    a.P = *P;
    a.ref.pac = ref->pac; a.ref.l_pac = ref->l_pac; a.ref.n_seqs = ref->n_seqs; a.ref.seq_off = ref->seq_offset; a.ref.seq_len = ref->seq_len;
    a.in.n_reads = B->n_reads; a.in.read_off = B->read_off; a.in.read_seq = B->read_seq; a.in.seed_all = B->seed_all; a.in.last_len = B->last_len;
    a.in.seed_off = B->seed_off; a.in.seed_id = B->seed_id; a.in.hit_off = B->hit_off; a.in.h_pos = B->h_pos; a.in.h_chr = B->h_chr;
    a.in.h_cig_off = B->h_cig_off; a.in.h_nm = B->h_nm; a.in.h_len_dif = B->h_len_dif; a.in.h_strand = B->h_strand; a.in.h_cig_n = B->h_cig_n; a.in.cig = B->cig;
    std::vector<int32_t> srt, rnk;
    hp_build_sort_index(B->n_reads, B->seed_off, B->hit_off, B->h_pos, B->h_chr, B->h_strand, srt, rnk);
    a.in.h_sort = srt.data(); a.in.h_rank = rnk.data();
    unsigned long long cursor = 0;
    a.out.stream = stream; a.out.stream_cap = stream_cap; a.out.cursor = &cursor;
    a.out.read_out_off = read_off; a.out.read_out_len = read_len; a.out.read_status = status; a.out.read_tbases = nullptr;
    std::vector<char> slab(slab_bytes);
    a.slab = slab.data(); a.slab_per_wave = slab_bytes; a.counter = nullptr; a.order = nullptr; a.n_units = B->n_reads; a.scale = scale; a.prof = nullptr;
    static int32_t lds[HP_LDS_WORDS];
    for (int r = 0; r < B->n_reads; ++r) align_read(a, r, 0, lds);
    *n_words = (int64_t)cursor;
    return 0;
}
