// hp_wavejob.h -- the DP jobs of a line that need a whole wavefront, as jobs of a launch of their own (k_filldp_wave, hp_align_api.hip).
//
// What a line's gap fill (frag_check, src/frag_check.c:886-955) spends on banded DP is known before the fill starts: the junctions of the
// mismatch class with more read bases than a lane job takes (split_mapping :547-559 -> ksw_bi_extend(100, 100), src/ksw.c:862-926), the seed
// gaps of that size (frag_extend :360-400 -> ksw_global2) and the two end extensions (frag_head_bound_fix :576-654 -> ksw_extend_r,
// frag_tail_bound_fix :656-707 -> ksw_extend_c, up to the whole read long).  None of them depends on the line's growing CIGAR -- only
// merge_cigar (:251), which joins their results, is sequential -- so the listing launch (phase_filllist, hp_phase.h) writes them as job
// records with the geometry the fill would compute, and this launch runs them one per wavefront, costliest first: a launch that is all
// instruction issue (VALU port 78 % busy) beside a fill launch that is all memory latency (wait 89 %), instead of one kernel that is both; the
// direction matrix of the two-columns-per-lane routines lies in the wave's LDS where it fits (a junction of up to 80 rows).  The fill finds the
// CIGARs in the job arena (FLines::jt / gt / ht) and goes on with merge_cigar; a job that was not listed, or whose buffers did not suffice, is
// run by the fill as before.
// The routines are those of hp_ksw.h: same recurrences, tie rules, band and z-drop logic, whatever launch calls them.
#pragma once
#include "hp_lanedp.h"

namespace hp {

enum { WJ_BI = 1, WJ_GLOBAL = 2, WJ_HEAD = 3, WJ_TAIL = 4 };
// type_comp: type | complement << 4 (a '-' line reads the reverse complement of the read) | query walked backwards << 5 | target walked backwards << 6
struct WjRec { int64_t qaddr, tk, slot; int32_t rd, qlen, tlen, type_comp; };
// Queues of the wave jobs, costliest first: WJ_NBIG classes of jobs whose scratch (above all the direction matrix of a long end extension)
// does not fit a wave's ordinary slab -- only the waves that own a big slab take those (the first n_wjb waves of the launch: a few hundred
// MB-sized slabs instead of one per wave, which is what capped the 20-kbp workload at 4 096 waves and 17 MB each) -- then WJ_NSMALL cost
// classes (powers of two of query length x band) that every wave takes.
enum { WJ_NBIG = 4, WJ_NSMALL = 12, WJ_NBUCKET = WJ_NBIG + WJ_NSMALL };

// what a job can ask of its slab at most: the staged sequences, the CIGARs of ksw_bi_extend (left, right, result, the global fallback's), the
// band limits of every row, HBM rows of the widest routine, and the direction matrix -- a row of it is the band's columns, or 64 / 128 / 256
// bytes of the register routines (hp_ksw.h)
HP_HD long long wj_need(const lamsa_hp_para *P, int type, int qlen, int tlen)
{
    const int d = qlen > tlen ? qlen - tlen : tlen - qlen;
    const int w = type == WJ_BI && d + 3 > P->band_w ? d + 3 : P->band_w;
    long long ncol = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    if (ncol < 256) ncol = 256;
    return ((long long)64 << 10) + 40ll * ((long long)qlen + tlen) + 16ll * qlen + ncol * ((long long)tlen + 16);
}

HP_INL int wj_bucket_of(const lamsa_hp_para *P, int type, int qlen, int tlen, bool big)
{
    const int band = 2 * P->band_w + 1, cols = qlen < band ? qlen : band;
    // rows: a junction's target; an end extension stops by z-drop somewhere along the query
    const long long cost = (long long)(type == WJ_HEAD || type == WJ_TAIL ? qlen : tlen) * (cols > 0 ? cols : 1);
    int b = 0;
    if (big) { for (long long c = cost >> 20; c > 0 && b < WJ_NBIG - 1; c >>= 1) ++b; return WJ_NBIG - 1 - b; }
    for (long long c = cost >> 11; c > 0 && b < WJ_NSMALL - 1; c >>= 1) ++b;
    return WJ_NBIG + WJ_NSMALL - 1 - b;
}

#ifndef HP_WJ_WAVES_PER_SIMD
#define HP_WJ_WAVES_PER_SIMD 8          // measured (profiles/r04_overlap.txt, ont10k, ms of the launch alone / reads/s with two batches in flight): 4 waves per SIMD and 9.5 KB of LDS 66.3 / 317 k, 6 and 6.5 KB 63.4 / 326 k, 8 and 5 KB 64.7 / 339 k
#endif
#ifndef HP_WJ_LDS_WORDS
#define HP_WJ_LDS_WORDS HP_LDS_WORDS             // 5 KB: thirty-two waves per CU (the launch is bound by instruction issue; more waves beat more LDS, hp_align_api.hip)
#endif

struct WjOut { int score, qle, tle, reflen, readlen; };

// One job on this wavefront.  The sequences are staged into the slab in the order the DP consumes them (query base j = the read's base
// qaddr + j * qs, complemented for a '-' line; target base i = the packed reference's base tk + i * ts), so every routine sees forward views.
// `out` must be bound to a buffer of at least qlen + tlen + 64 words.  false: a buffer did not suffice or the reference would exit -- cx.status says which.
HP_FN bool wj_run(Ctx &cx, const uint8_t *reads, const uint8_t *pac, int type, int comp, int64_t qaddr, int qs, int qlen, int64_t tk, int ts, int tlen,
                  int w, int h0, CigV &out, WjOut &o)
{
    const lamsa_hp_para *P = cx.P;
    o.score = 0; o.qle = 0; o.tle = 0; o.reflen = 0; o.readlen = 0;
    out.n = 0;
    uint8_t *qb = (uint8_t *)arena_alloc(cx, (size_t)(qlen > 0 ? qlen : 0) + 16), *tb = (uint8_t *)arena_alloc(cx, (size_t)(tlen > 0 ? tlen : 0) + 16);
    if (!qb || !tb) return false;
    {
        const HP_G uint8_t *gr = (const HP_G uint8_t *)reads + qaddr, *gp = (const HP_G uint8_t *)pac;
        HP_G uint8_t *gq = (HP_G uint8_t *)qb, *gt = (HP_G uint8_t *)tb;
        for (int b = 0; b < qlen; b += 64) { WAVE_FOR(l) { const int j = b + l; if (j < qlen) { const int c = gr[(long)j * qs]; gq[j] = (uint8_t)(comp ? (c < 4 ? 3 - c : 4) : c); } } }
        for (int b = 0; b < tlen; b += 64) { WAVE_FOR(l) { const int i = b + l; if (i < tlen) { const int64_t k = tk + (int64_t)i * ts; gt[i] = (uint8_t)(gp[k >> 2] >> ((~k & 3) << 1) & 3); } } }      // _get_pac, bntseq.c:242
        wv::sync();
    }
    const Seq q = seq_fwd(qb), t = seq_fwd(tb);
    if (type == WJ_BI) o.score = ksw_bi_extend(cx, qlen, q, tlen, t, h0, h0, out);      // (the "gap exists" flag)
    else if (type == WJ_GLOBAL) o.score = ksw_global(cx, qlen, q, tlen, t, P->del_gapo, P->del_gape, P->ins_gapo, P->ins_gape, w, &out);
    else {
        // ksw_extend_c / ksw_extend_r (src/ksw.c:809-830) and what frag_head_bound_fix / frag_tail_bound_fix do with the result (:640-648, :699-703):
        // the rest of the read is clipped, the head's CIGAR turned round
        o.score = ksw_extend(cx, qlen, q, tlen, t, w, h0, &o.qle, &o.tle, &out);
        const int rr = o.qle == qlen ? 0 : (o.tle == tlen ? 1 : 2);
        if (rr != 0) cig_push1(cx, out, ((qlen - o.qle) << 4) | C_S);
        if (type == WJ_HEAD) cig_invert(out.c, out.n);
    }
    wv::sync();
    if (cx.status & (ST_REFEXIT | ST_OVERFLOW)) return false;
    o.reflen = cig_reflen(out.c, out.n); o.readlen = cig_readlen(out.c, out.n);
    return true;
}

}  // namespace hp
