// hp_batch.h -- HBM layout of one batch of reads + seed hits, and of the result stream.
//
// Inputs are struct-of-arrays: the reference's map_t is 1064 B per
// hit because of a 1 KB contig name (src/lamsa_aln.h:230-239); here a hit is 22 B spread
// over coalescable arrays.  Index spaces:
//   read r      : [0, n_reads)
//   seed slot s : read r owns slots [seed_off[r], seed_off[r+1])  (only seeds that have a GEM
//                 map line get a slot, src/lamsa_aln.c:945-952; seed_id keeps the 1-based index)
//   hit h       : slot s owns hits [hit_off[s], hit_off[s+1])
#pragma once
#include "hp_core.h"

namespace hp {

struct RefView {                 // packed reference resident in HBM
    const uint8_t *pac; int64_t l_pac; int32_t n_seqs;
    const int64_t *seq_off; const int32_t *seq_len;
};

struct BatchIn {
    int32_t n_reads;
    const uint8_t *read_skip;    // [n_reads] or nullptr: 1 = the batch check found the read beyond the device's field widths (ST_UNSUPPORTED)
    const int64_t *read_off;     // [n_reads+1] into read_seq
    const uint8_t *read_seq;     // 1 byte/base, codes 0..4
    const int32_t *seed_all;     // [n_reads]  1+(L-seed_len)/seed_step (src/lamsa_aln.c:252-253)
    const int32_t *last_len;     // [n_reads]  (src/lamsa_aln.c:281)
    const int64_t *seed_off;     // [n_reads+1]
    const int32_t *seed_id;      // [n_slots]
    const int64_t *hit_off;      // [n_slots+1]
    const int64_t *h_pos;        // [n_hits] 1-based leftmost reference coordinate
    const int32_t *h_chr;        // 1-based contig id
    const int64_t *h_cig_off;    // into cig[] (the boundary takes 32-bit offsets or none at all; widened / summed up on the device)
    const int16_t *h_nm, *h_len_dif;
    const int8_t  *h_strand;     // +1 / -1
    const uint8_t *h_cig_n;
    const int32_t *cig;          // seed CIGAR words
};

// Result stream of one read (int32 words), serialised by the wave that aligned it:
//   [0] status bits  [1] n_lines(stage 0 = first round)  [2] n_lines(stage 1 = remain round)
//   per line : line_score, tol_score, tol_NM, n_res
//   per res  : offset_lo, offset_hi, chr, nstrand(1 '+', 0 '-'), score(AS), NM, cigar_n, cigar words...
struct BatchOut {
    int32_t *stream;             // global result arena
    int64_t stream_cap;          // words
    unsigned long long *cursor;  // bump pointer (words), atomically advanced once per read
    int64_t *read_out_off;       // [n_reads] start of each read's stream (-1: arena overflow)
    int32_t *read_out_len;       // [n_reads]
    int32_t *read_status;        // [n_reads]
    int32_t *read_tbases;        // [n_reads] reference bases fetched (2-bit windows), or nullptr
    int32_t *read_work;          // [4 * n_reads] per read: DP cells updated, chaining edge classifications executed, seed-CIGAR words read, 0; or nullptr
    unsigned long long *diag;    // 32 words of launch accounting (hp_phase.h), or nullptr
};

struct AlignArgs {
    lamsa_hp_para P;
    RefView ref;
    BatchIn in;
    BatchOut out;
    char *slab; size_t slab_per_wave;
    int32_t sort_pb, sort_cb;    // bits of the largest hit position / contig*2+strand code of the batch (hp_sort.h)
    int32_t *counter;            // dynamic read queue head
    const int32_t *order;        // processing order (costliest first) / retry list, or nullptr
    int32_t n_units;             // number of entries to process (n_reads, or the length of the retry list)
    long long *prof;             // diagnostic build (-DHP_PROF): per-read phase cycle sums, 16 per read; else nullptr
    int32_t scale;               // multiplier of the per-read output / CIGAR capacities (1; larger in the retry pass)
};

}  // namespace hp
