// rescue.h -- stage (4) of the per-read pipeline: the BWT rescue of short read regions that rounds (2)/(3) and
// (2')/(3') left uncovered (reference src/bwt_aln.c).  Host work around the GPU's DP batch:
//   plan    (host threads)  uncovered regions of a read -> exact 19-mer hits through the FM index of the reference's
//                           <ref>.bwt / <ref>.sa -> seed chains -> per chain one global and two extension DP jobs
//   DP      (GPU)           all jobs of a chunk in one lamsa_hp_dp_batch() call (ksw_global2 / ksw_extend_core)
//   finish  (host threads)  CIGAR assembly, NM / AS, the read-type filter -> lines of a_res[2]
#pragma once
#include <string>
#include <vector>
#include "lamsa_host.h"

namespace lamsa {

// FM index in the reference's on-disk format (a BWA 0.7 .bwt with occurrence checkpoints every 128 symbols interleaved,
// and a sampled suffix array); bwt_restore_bwt / bwt_restore_sa, src/bwt.c:421-462
struct FmIndex {
    uint64_t primary = 0, L2[5] = {0, 0, 0, 0, 0}, seq_len = 0;
    std::vector<uint32_t> bwt;
    uint64_t sa_intv = 0, n_sa = 0;
    std::vector<uint64_t> sa;
    bool load(const std::string &prefix, std::string &err);
    uint64_t occ(uint64_t k, int c) const;                                   // bwt_occ :107
    uint64_t sa_at(uint64_t k) const;                                        // bwt_sa :86
    uint64_t match(int len, const uint8_t *s, uint64_t *k, uint64_t *l) const;   // bwt_match_exact_alt :241
};

struct RescueBound { int read_pos; int64_t ref_pos; };

struct RescueLine {                  // one seed chain that goes to bwt_aln_res (:200-303)
    int ref_id = 0, is_rev = 0, reg_beg = 0, reg_len = 0;
    RescueBound left, right;
    int left_eta = 0, right_eta = 0, extra_beg = 0, extra_end = 0;
    int64_t ref_start = 0; int ref_len = 0;
    std::vector<uint8_t> query, target, lq, lt;      // window of the read (strand of the hit), of the reference; reversed left flanks
    int job_mid = -1, job_left = -1, job_right = -1; // indices into the plan's job list (-1: no such extension)
    int left_qlen = 0, right_qlen = 0;
};

struct RescueJobs {                  // the DP jobs of a chunk, in lamsa_hp_dp_batch form
    std::vector<uint8_t> seq; std::vector<int64_t> q_off, t_off; std::vector<int32_t> qlen, tlen, kind, w, h0;
    int add(const uint8_t *q, int ql, const uint8_t *t, int tl, int kind_, int w_, int h0_);
};

struct RescuePlan {                  // of one read
    std::vector<RescueLine> lines;
    std::vector<int> region_of;      // region index of every line (lines of one region are consecutive)
    std::vector<int> job_base;       // unused by callers; kept for debugging
};

// plan: `bseq` = the read's base codes (0..4).  Jobs are appended to `jobs` (one job list per host thread).
void rescue_plan(const ReadResult &R, const uint8_t *bseq, int read_len, const Index &ix, const FmIndex &fm, const lamsa_hp_para &P, RescuePlan &plan, RescueJobs &jobs);

struct DpResults { const int32_t *score, *qle, *tle; const int64_t *cig_off; const int32_t *cigar; int64_t base; };   // base: first job of this thread's list

// finish: DP results -> R.stage[2]
void rescue_finish(ReadResult &R, const uint8_t *bseq, int read_len, const Index &ix, const lamsa_hp_para &P, RescuePlan &plan, const DpResults &dp);

}  // namespace lamsa
