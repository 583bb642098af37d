// hp_dp_batch.h -- one DP job per wavefront: the unit of work behind lamsa_hp_dp_batch().
#pragma once
#include "hp_ksw.h"

namespace hp {

struct DpBatchArgs {
    lamsa_hp_para P;
    int32_t n_jobs;
    const uint8_t *seq;
    const int64_t *q_off, *t_off;
    const int32_t *qlen, *tlen, *kind, *w, *h0;
    int32_t *score, *qle, *tle, *status, *cig_n;
    const int64_t *cig_cap_off;   // [n_jobs+1] capacity prefix sums into cig
    cig_t *cig;
    char *slab; size_t slab_per_wave;
    int32_t *counter;             // dynamic job queue head
};

HP_FN void dp_run_job(const DpBatchArgs &a, int job, int wave_slot, HP_L int32_t *lds)
{
    Ctx cx;
    cx.P = &a.P;
    cx.status = 0; cx.n_cells = 0; cx.lds_epoch = 0; cx.prof = nullptr; cx.lds = lds; cx.lds_words = HP_LDS_WORDS;
    arena_init(cx.tmp, a.slab + (size_t)wave_slot * a.slab_per_wave, a.slab_per_wave);
    const int ql = a.qlen[job], tl = a.tlen[job];
    Seq q = seq_fwd(a.seq + a.q_off[job]), t = seq_fwd(a.seq + a.t_off[job]);
    CigV out;
    cig_bind(out, a.cig + a.cig_cap_off[job], (int)(a.cig_cap_off[job + 1] - a.cig_cap_off[job]));
    int sc = 0, qle = 0, tle = 0;
    const int kind = a.kind[job];
    if (kind == 0) {
        sc = ksw_global(cx, ql, q, tl, t, a.P.del_gapo, a.P.del_gape, a.P.ins_gapo, a.P.ins_gape, a.w[job], &out);
    } else if (kind == 1) {
        sc = ksw_extend(cx, ql, q, tl, t, a.w[job], a.h0[job], &qle, &tle, &out);
    } else {
        sc = ksw_bi_extend(cx, ql, q, tl, t, a.h0[job], a.h0[job], out);
    }
    a.score[job] = sc; a.qle[job] = qle; a.tle[job] = tle; a.status[job] = cx.status; a.cig_n[job] = out.n;
}

}  // namespace hp
