// hp_api.hip -- C-ABI of liblamsa_hp.so (include/lamsa_hp.h) and the gfx950 kernels behind it.
// HIP only: there is no CPU path in this library; every entry point fails with
// LAMSA_HP_ENODEV when no HIP device is usable.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <string>
#include "hp_dp_batch.h"
#include "hp_wavejob.h"

using namespace hp;

// ------------------------------------------------------------------ kernels
// One wavefront (64 threads) per workgroup; a persistent grid pulls units from a queue head.
__global__ __launch_bounds__(64) void k_dp_batch(DpBatchArgs a)
{
    __shared__ int32_t lds[HP_LDS_WORDS];
    const int slot = blockIdx.x;
    for (;;) {
        int job = 0;
        if (wv::leader()) job = atomicAdd(a.counter, 1);
        job = wv::uni(job);
        if (job >= a.n_jobs) break;
        dp_run_job(a, job, slot, (HP_L int32_t *)lds);
    }
}

// The same jobs by the routines the read path's lane-per-job launch uses: kinds 4 / 5 / 6 = ksw_global2 / ksw_extend_core / ksw_bi_extend one
// job per LANE (hp_lanedp.h, what k_filldp_small runs).  Targets come 2 bits per base (tk: first base of every job in `pac`), as those
// routines read the reference.
__global__ __launch_bounds__(64, 1) void k_dp_batch_lane(DpBatchArgs a, const uint8_t *pac, const int64_t *tk)
{
    __shared__ int32_t lds[HP_LJ_LDS_WORDS(HP_LJ_QCAP)];
    const lamsa_hp_para *P = &a.P;
    char *slab = a.slab + (size_t)blockIdx.x * a.slab_per_wave;
    cig_t *cbuf = (cig_t *)slab;
    uint8_t *zbuf = (uint8_t *)(slab + sizeof(cig_t) * 3 * HP_LJ_CIG * 64);
    for (;;) {
        int g = 0;
        if (wv::leader()) g = atomicAdd(a.counter, 1);
        g = wv::uni(g);
        if (g * 64 >= a.n_jobs) break;
        wv::sync();
        WAVE_FOR(l) {
            const int job = g * 64 + l;
            if (job < a.n_jobs) {
                LaneJob J;
                J.q = (const HP_G uint8_t *)(a.seq + a.q_off[job]); J.qs = 1; J.qcomp = 0; J.qlen = a.qlen[job]; J.pac = (const HP_G uint8_t *)pac; J.tk = tk[job]; J.ts = 1; J.tlen = a.tlen[job];
                J.z = (HP_G uint8_t *)zbuf; J.zl = l; J.zs = HP_LJ_QCAP; J.cells = 0; J.row = (HP_L int32_t *)lds + l; J.qrow = (HP_L uint8_t *)((HP_L int32_t *)lds + (HP_LJ_QCAP + 2) * 64) + l; J.rev = 0;
                lj_stage_query(J);
                LCig out, Lc, Rc;
                out.c = cbuf + (size_t)l * 3 * HP_LJ_CIG; out.n = 0; Lc.c = out.c + HP_LJ_CIG; Lc.n = 0; Rc.c = Lc.c + HP_LJ_CIG; Rc.n = 0;
                int sc = 0, qle = 0, tle = 0;
                const int kind = a.kind[job];
                if (kind == 4) sc = lj_global(P, J, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, a.w[job], &out);
                else if (kind == 5) sc = lj_extend(P, J, a.w[job], a.h0[job], &qle, &tle, &out);
                else sc = lj_bi_extend(P, J, a.h0[job], a.h0[job], Lc, Rc, out);
                a.score[job] = sc; a.qle[job] = qle; a.tle[job] = tle; a.status[job] = 0; a.cig_n[job] = out.n;
                cig_t *dst = a.cig + a.cig_cap_off[job];
                for (int k = 0; k < out.n; ++k) dst[k] = out.c[k];
            }
        }
        wv::sync();
    }
}
// Kinds 8 .. 11: a job as the wave-per-job launch of the read path runs it (hp_wavejob.h: k_filldp_wave's registers and LDS, sequences staged
// from the read bytes and the packed reference, the direction matrix in LDS where it fits): 8 = a junction's ksw_bi_extend(h0, h0), 9 = a
// seed gap's ksw_global2(w), 10 / 11 = a line's head / tail extension (ksw_extend_r / ksw_extend_c with (w, h0), the rest of the query
// clipped, the head's CIGAR turned round -- frag_check.c:640-648, :699-703).
__global__ __launch_bounds__(64, HP_WJ_WAVES_PER_SIMD) void k_dp_batch_wave(DpBatchArgs a, const uint8_t *pac, const int64_t *tk)
{
    __shared__ int32_t lds[HP_WJ_LDS_WORDS];
    for (;;) {
        int job = 0;
        if (wv::leader()) job = atomicAdd(a.counter, 1);
        job = wv::uni(job);
        if (job >= a.n_jobs) break;
        Ctx cx;
        cx.P = &a.P; cx.lds = (HP_L int32_t *)lds; cx.lds_words = HP_WJ_LDS_WORDS; cx.status = 0; cx.n_cells = 0; cx.lds_epoch = 0; cx.prof = nullptr;
        arena_init(cx.tmp, a.slab + (size_t)blockIdx.x * a.slab_per_wave, a.slab_per_wave);
        const int type = wv::uni(a.kind[job]) - 7, qlen = wv::uni(a.qlen[job]), tlen = wv::uni(a.tlen[job]);
        const bool back = type == WJ_HEAD;
        CigV out; cig_bind(out, a.cig + a.cig_cap_off[job], (int)(a.cig_cap_off[job + 1] - a.cig_cap_off[job]));
        WjOut o;
        wj_run(cx, a.seq, pac, type, 0, a.q_off[job] + (back && qlen > 0 ? qlen - 1 : 0), back ? -1 : 1, qlen, tk[job] + (back && tlen > 0 ? tlen - 1 : 0), back ? -1 : 1, tlen,
               wv::uni(a.w[job]), wv::uni(a.h0[job]), out, o);
        if (wv::leader()) { a.score[job] = o.score; a.qle[job] = o.qle; a.tle[job] = o.tle; a.status[job] = cx.status; a.cig_n[job] = out.n; }
        wv::sync();
    }
}

#include "hp_handle.h"

#include "hp_para.h"

extern "C" int lamsa_hp_create(lamsa_hp_handle **out, const lamsa_hp_para *para, const lamsa_hp_ref *ref, int device_id)
{
    if (!out || !para) return LAMSA_HP_EINVAL;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device_id < 0 || device_id >= n_dev) return LAMSA_HP_ENODEV;
    if (hipSetDevice(device_id) != hipSuccess) return LAMSA_HP_ENODEV;
    lamsa_hp_handle *h = new lamsa_hp_handle();
    h->device = device_id; h->para = *para;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) h->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&h->stream_b, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) { delete h; return LAMSA_HP_ENODEV; }
    if (ref && ref->pac) {
        size_t pb = (size_t)(ref->l_pac / 4 + 1);
        h->l_pac = ref->l_pac; h->n_seqs = ref->n_seqs;
        if (hipMalloc((void **)&h->d_pac, pb + 16) != hipSuccess ||
            hipMalloc((void **)&h->d_seq_off, sizeof(int64_t) * (size_t)(ref->n_seqs + 1)) != hipSuccess ||
            hipMalloc((void **)&h->d_seq_len, sizeof(int32_t) * (size_t)(ref->n_seqs + 1)) != hipSuccess) { lamsa_hp_destroy(h); return LAMSA_HP_ENOMEM; }
        hipMemcpy(h->d_pac, ref->pac, pb, hipMemcpyHostToDevice);
        hipMemcpy(h->d_seq_off, ref->seq_offset, sizeof(int64_t) * (size_t)ref->n_seqs, hipMemcpyHostToDevice);
        hipMemcpy(h->d_seq_len, ref->seq_len, sizeof(int32_t) * (size_t)ref->n_seqs, hipMemcpyHostToDevice);
    }
    *out = h;
    return LAMSA_HP_OK;
}

extern "C" void lamsa_hp_release_state_(lamsa_hp_handle *h);
extern "C" void lamsa_hp_destroy(lamsa_hp_handle *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    lamsa_hp_release_state_(h);
    if (h->d_pac) hipFree(h->d_pac);
    if (h->d_seq_off) hipFree(h->d_seq_off);
    if (h->d_seq_len) hipFree(h->d_seq_len);
    h->in.release(); h->out.release(); h->slab.release(); h->misc.release(); h->pac2.release();
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->stream) hipStreamDestroy(h->stream);
    if (h->copy_stream) hipStreamDestroy(h->copy_stream);
    if (h->stream_b) hipStreamDestroy(h->stream_b);
    delete h;
}

extern "C" const char *lamsa_hp_last_error(const lamsa_hp_handle *h) { return h ? h->err.c_str() : "null handle"; }
extern "C" float lamsa_hp_last_kernel_ms(const lamsa_hp_handle *h, int which) { return (h && which >= 0 && which < 24) ? h->kernel_ms[which] : -1.f; }


extern "C" int lamsa_hp_dp_batch(lamsa_hp_handle *h, const lamsa_hp_dp_jobs *J, lamsa_hp_dp_out *O)
{
    if (!h || !J || !O || J->n_jobs < 0) return LAMSA_HP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device), LAMSA_HP_ENODEV);
    const int n = J->n_jobs;
    // capacities: a DP CIGAR has at most qlen+tlen words (+ slack for the S/H pair of sw_mid_fix)
    std::vector<int64_t> cap_off((size_t)n + 1, 0);
    size_t z_need = 4096;
    for (int i = 0; i < n; ++i) {
        int ql = J->qlen[i], tl = J->tlen[i];
        if (ql < 0) ql = 0;
        if (tl < 0) tl = 0;
        if (J->q_off[i] < 0 || J->t_off[i] < 0 || J->q_off[i] + ql > J->seq_bytes || J->t_off[i] + tl > J->seq_bytes) { h->err = "dp job sequence out of range"; return LAMSA_HP_EINVAL; }
        cap_off[i + 1] = cap_off[i] + ql + tl + (J->kind[i] >= 8 ? 72 : 8);
        // worst-case scratch of one job: H,E rows + row bounds + direction matrix + 3 temporary CIGARs (bi-extend)
        size_t wmax = (size_t)(abs(ql - tl) + 3 > (J->kind[i] == 2 ? h->para.band_w : J->w[i]) ? abs(ql - tl) + 3 : (J->kind[i] == 2 ? h->para.band_w : J->w[i]));
        size_t ncol = (size_t)ql < 2 * wmax + 1 ? (size_t)ql : 2 * wmax + 1;
        if (ql <= HP_PK_QMAX(2) && ncol < 128) ncol = 128;          // the two-columns-per-lane extension keeps whole rows of its direction matrix (hp_ksw.h)
        size_t need = 2 * 4 * ((size_t)ql + 18) + 8 * ((size_t)tl + 17) + ncol * tl + 64 + 3 * 4 * ((size_t)ql + tl + 24) + 1024;
        if (need > z_need) z_need = need;
    }
    // kinds 4..6: the lane-per-job routines (they take jobs up to their buffers' sizes and, like the read path, only when the handle's penalties
    // keep their 16-bit cells exact -- else the jobs run on the wave routines, kind - 4); kinds 8..11: a job as the wave-per-job launch runs it.
    // One class per call.
    int cls = 0;                                          // 0: wave routines (k_dp_batch), 1: a job per lane, 2: the wave-per-job launch's way
    std::vector<int32_t> kind_v;
    std::vector<uint8_t> pac; std::vector<int64_t> tkv;
    if (n > 0 && J->kind[0] >= 4) {
        cls = J->kind[0] >= 8 ? 2 : 1;
        for (int i = 0; i < n; ++i) {
            const int k = J->kind[i];
            if (k < 4 || k == 7 || k > 11 || (k >= 8) != (cls == 2)) { h->err = "dp batch mixes job classes"; return LAMSA_HP_EINVAL; }
            if (cls == 1 && (J->qlen[i] < 0 || J->qlen[i] > HP_LJ_QCAP || J->tlen[i] < 0 || J->tlen[i] > HP_LJ_TCAP || (k != 4 && J->h0[i] <= 0))) { h->err = "dp job beyond the lane routines' buffers"; return LAMSA_HP_EINVAL; }
            if (cls == 2 && (J->qlen[i] < 0 || J->tlen[i] < 0 || (k != 9 && J->h0[i] <= 0))) { h->err = "dp job with a negative length or without a start score"; return LAMSA_HP_EINVAL; }
        }
        if (cls == 1 && !lj_params_ok(&h->para)) {          // as the read path does: these parameters stay on the wave routines
            kind_v.assign(J->kind, J->kind + n);
            for (int i = 0; i < n; ++i) kind_v[i] -= 4;
            cls = 0;
        } else {
            int64_t tot = 0;
            for (int i = 0; i < n; ++i) tot += J->tlen[i];
            pac.assign((size_t)tot / 4 + 16, 0); tkv.assign((size_t)n + 1, 0);
            int64_t k = 0;
            for (int i = 0; i < n; ++i) {
                tkv[i] = k;
                for (int j = 0; j < J->tlen[i]; ++j, ++k) {
                    const uint8_t b = J->seq[J->t_off[i] + j];
                    if (b > 3) { h->err = "these routines read the target 2 bits per base: no N"; return LAMSA_HP_EINVAL; }
                    pac[(size_t)(k >> 2)] |= (uint8_t)(b << ((~k & 3) << 1));
                }
            }
            if (cls == 1) z_need = sizeof(cig_t) * 3 * HP_LJ_CIG * 64 + (size_t)HP_LJ_QCAP * HP_LJ_TCAP * 64 + 64;
        }
    }
    const int32_t *kind_src = kind_v.empty() ? J->kind : kind_v.data();
    size_t slab_per_wave = al256(z_need);
    int n_waves = h->n_cu * (cls == 1 ? 3 : (cls == 2 ? 16 : 8));
    if (cls == 2) {   // every job's worst case: the wave routines' scratch (above: computed from the forward band) plus the staged sequences
        size_t zz = 4096;
        for (int i = 0; i < n; ++i) {
            const size_t ql = (size_t)J->qlen[i], tl = (size_t)J->tlen[i];
            const size_t wmax = std::max<size_t>((size_t)abs((int)ql - (int)tl) + 3, (size_t)std::max(J->w[i], h->para.band_w));
            const size_t ncol = std::max<size_t>(std::min(ql, 2 * wmax + 1), 256);
            zz = std::max(zz, 2 * (ql + tl + 64) + 2 * 4 * (ql + 18) + 8 * (tl + 17) + ncol * tl + 64 + 4 * 4 * (ql + tl + 72) + 4096);
        }
        slab_per_wave = al256(zz);
    }
    if (n_waves > n) n_waves = n > 0 ? n : 1;
    while (n_waves > 1 && slab_per_wave * (size_t)n_waves > ((size_t)64 << 30)) n_waves /= 2;

    // ---- pack inputs into one upload
    size_t o_seq = 0, o_qoff = al256((size_t)J->seq_bytes + 16), o_toff = o_qoff + al256(8 * (size_t)n), o_cap = o_toff + al256(8 * (size_t)n),
           o_ql = o_cap + al256(8 * ((size_t)n + 1)), o_tl = o_ql + al256(4 * (size_t)n), o_kind = o_tl + al256(4 * (size_t)n),
           o_w = o_kind + al256(4 * (size_t)n), o_h0 = o_w + al256(4 * (size_t)n), in_bytes = o_h0 + al256(4 * (size_t)n);
    if (h->in.ensure(in_bytes)) { h->err = "hipMalloc(in)"; return LAMSA_HP_ENOMEM; }
    size_t o_score = 0, o_qle = al256(4 * (size_t)n), o_tle = 2 * o_qle, o_st = 3 * o_qle, o_cn = 4 * o_qle, o_cig = 5 * o_qle, out_bytes = o_cig + al256(4 * (size_t)cap_off[n] + 16);
    if (h->out.ensure(out_bytes)) { h->err = "hipMalloc(out)"; return LAMSA_HP_ENOMEM; }
    if (h->slab.ensure(slab_per_wave * (size_t)n_waves)) { h->err = "hipMalloc(slab)"; return LAMSA_HP_ENOMEM; }
    if (h->misc.ensure(256)) { h->err = "hipMalloc(misc)"; return LAMSA_HP_ENOMEM; }
    char *din = (char *)h->in.p, *dout = (char *)h->out.p;
    hipStream_t s = h->stream;
    if (n > 0) {
        HIPCHK(h, hipMemcpyAsync(din + o_seq, J->seq, (size_t)J->seq_bytes, hipMemcpyHostToDevice, s), LAMSA_HP_EKERNEL);
        HIPCHK(h, hipMemcpyAsync(din + o_qoff, J->q_off, 8 * (size_t)n, hipMemcpyHostToDevice, s), LAMSA_HP_EKERNEL);
        HIPCHK(h, hipMemcpyAsync(din + o_toff, J->t_off, 8 * (size_t)n, hipMemcpyHostToDevice, s), LAMSA_HP_EKERNEL);
        HIPCHK(h, hipMemcpyAsync(din + o_cap, cap_off.data(), 8 * ((size_t)n + 1), hipMemcpyHostToDevice, s), LAMSA_HP_EKERNEL);
        HIPCHK(h, hipMemcpyAsync(din + o_ql, J->qlen, 4 * (size_t)n, hipMemcpyHostToDevice, s), LAMSA_HP_EKERNEL);
        HIPCHK(h, hipMemcpyAsync(din + o_tl, J->tlen, 4 * (size_t)n, hipMemcpyHostToDevice, s), LAMSA_HP_EKERNEL);
        HIPCHK(h, hipMemcpyAsync(din + o_kind, kind_src, 4 * (size_t)n, hipMemcpyHostToDevice, s), LAMSA_HP_EKERNEL);
        HIPCHK(h, hipMemcpyAsync(din + o_w, J->w, 4 * (size_t)n, hipMemcpyHostToDevice, s), LAMSA_HP_EKERNEL);
        HIPCHK(h, hipMemcpyAsync(din + o_h0, J->h0, 4 * (size_t)n, hipMemcpyHostToDevice, s), LAMSA_HP_EKERNEL);
    }
    HIPCHK(h, hipMemsetAsync(h->misc.p, 0, 256, s), LAMSA_HP_EKERNEL);

    DpBatchArgs a;
    a.P = h->para; a.n_jobs = n;
    a.seq = (const uint8_t *)(din + o_seq); a.q_off = (const int64_t *)(din + o_qoff); a.t_off = (const int64_t *)(din + o_toff);
    a.qlen = (const int32_t *)(din + o_ql); a.tlen = (const int32_t *)(din + o_tl); a.kind = (const int32_t *)(din + o_kind);
    a.w = (const int32_t *)(din + o_w); a.h0 = (const int32_t *)(din + o_h0);
    a.score = (int32_t *)(dout + o_score); a.qle = (int32_t *)(dout + o_qle); a.tle = (int32_t *)(dout + o_tle);
    a.status = (int32_t *)(dout + o_st); a.cig_n = (int32_t *)(dout + o_cn);
    a.cig_cap_off = (const int64_t *)(din + o_cap); a.cig = (cig_t *)(dout + o_cig);
    a.slab = (char *)h->slab.p; a.slab_per_wave = slab_per_wave; a.counter = (int32_t *)h->misc.p;

    HIPCHK(h, hipEventRecord(h->ev0, s), LAMSA_HP_EKERNEL);
    if (n > 0 && cls == 0) hipLaunchKernelGGL(k_dp_batch, dim3(n_waves), dim3(64), 0, s, a);
    else if (n > 0) {
        const size_t pb = al256(pac.size()), tb = al256(8 * ((size_t)n + 1));
        if (h->pac2.ensure(pb + tb)) { h->err = "hipMalloc(pac)"; return LAMSA_HP_ENOMEM; }
        HIPCHK(h, hipMemcpy(h->pac2.p, pac.data(), pac.size(), hipMemcpyHostToDevice), LAMSA_HP_EKERNEL);
        HIPCHK(h, hipMemcpy((char *)h->pac2.p + pb, tkv.data(), 8 * ((size_t)n + 1), hipMemcpyHostToDevice), LAMSA_HP_EKERNEL);
        if (cls == 1) hipLaunchKernelGGL(k_dp_batch_lane, dim3(n_waves), dim3(64), 0, s, a, (const uint8_t *)h->pac2.p, (const int64_t *)((char *)h->pac2.p + pb));
        else hipLaunchKernelGGL(k_dp_batch_wave, dim3(n_waves), dim3(64), 0, s, a, (const uint8_t *)h->pac2.p, (const int64_t *)((char *)h->pac2.p + pb));
    }
    HIPCHK(h, hipGetLastError(), LAMSA_HP_EKERNEL);
    HIPCHK(h, hipEventRecord(h->ev1, s), LAMSA_HP_EKERNEL);

    // ---- fetch results, compact the CIGARs
    h->h_score.assign((size_t)n, 0); h->h_qle.assign((size_t)n, 0); h->h_tle.assign((size_t)n, 0); h->h_status.assign((size_t)n, 0);
    std::vector<int32_t> cn((size_t)n, 0), raw((size_t)cap_off[n] + 4, 0);
    if (n > 0) {
        HIPCHK(h, hipMemcpyAsync(h->h_score.data(), dout + o_score, 4 * (size_t)n, hipMemcpyDeviceToHost, s), LAMSA_HP_EKERNEL);
        HIPCHK(h, hipMemcpyAsync(h->h_qle.data(), dout + o_qle, 4 * (size_t)n, hipMemcpyDeviceToHost, s), LAMSA_HP_EKERNEL);
        HIPCHK(h, hipMemcpyAsync(h->h_tle.data(), dout + o_tle, 4 * (size_t)n, hipMemcpyDeviceToHost, s), LAMSA_HP_EKERNEL);
        HIPCHK(h, hipMemcpyAsync(h->h_status.data(), dout + o_st, 4 * (size_t)n, hipMemcpyDeviceToHost, s), LAMSA_HP_EKERNEL);
        HIPCHK(h, hipMemcpyAsync(cn.data(), dout + o_cn, 4 * (size_t)n, hipMemcpyDeviceToHost, s), LAMSA_HP_EKERNEL);
        HIPCHK(h, hipMemcpyAsync(raw.data(), dout + o_cig, 4 * (size_t)cap_off[n], hipMemcpyDeviceToHost, s), LAMSA_HP_EKERNEL);
    }
    HIPCHK(h, hipStreamSynchronize(s), LAMSA_HP_EKERNEL);
    hipEventElapsedTime(&h->kernel_ms[0], h->ev0, h->ev1);
    h->h_i64.assign((size_t)n + 1, 0);
    h->h_cig.clear();
    for (int i = 0; i < n; ++i) {
        h->h_i64[i] = (int64_t)h->h_cig.size();
        h->h_cig.insert(h->h_cig.end(), raw.begin() + cap_off[i], raw.begin() + cap_off[i] + cn[i]);
    }
    h->h_i64[n] = (int64_t)h->h_cig.size();
    if (h->h_cig.empty()) h->h_cig.push_back(0);
    O->score = h->h_score.data(); O->qle = h->h_qle.data(); O->tle = h->h_tle.data(); O->status = h->h_status.data();
    O->cig_off = h->h_i64.data(); O->cigar = h->h_cig.data();
    return LAMSA_HP_OK;
}
