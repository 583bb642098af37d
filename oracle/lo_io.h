/* lo_io.h -- host-side file handling of the oracle's `lamsa aln -N -I` restatement (see lo.h). */
#ifndef LAMSA_ORACLE_LO_IO_H_
#define LAMSA_ORACLE_LO_IO_H_
#include <stdio.h>
#include "lo_read.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {            /* .ann + .pac of the reference index (src/bntseq.c:114-223) */
    lo_ref ref; char **name; int64_t *off; int32_t *len; uint8_t *pac;
} lo_index;
int  lo_index_load(lo_index *ix, const char *prefix);
void lo_index_free(lo_index *ix);

/* Parse one GEM hit list ("chr:+:pos:gigar,chr:-:pos:gigar,...") into hits; returns the number
 * of hits, 0 when the seed has >= max_n hits (src/gem_parse.c:230-286, md2cigar :74) */
int lo_parse_hits(const char *s, const lo_index *ix, int max_n, lo_hit **hits, int *hit_m, lo_cig **cig, int *cig_n, int *cig_m);

/* whole run: the reference's `lamsa aln -N -I [-R 0]` (stage 4 skipped), SAM to `out`.
 * n_threads > 1 processes reads in parallel, output order unchanged. */
int lo_run_aln(const char *ref_prefix, const char *reads, lo_para *P, FILE *out, const char *pg_line, int n_threads, long max_reads, double *aln_seconds, long *n_reads_out, long *n_bases_out);

#ifdef __cplusplus
}
#endif
#endif

/* ---- struct-of-arrays batches (the layout of include/lamsa_hp.h's lamsa_hp_batch), for the parity tests ---- */
typedef struct {
    int32_t n_reads; int64_t n_slots, n_hits, n_cig;
    int64_t *read_off; uint8_t *read_seq; int32_t *seed_all, *last_len;
    int64_t *seed_off; int32_t *seed_id; int64_t *hit_off;
    int64_t *h_pos; int32_t *h_chr; int8_t *h_strand; int16_t *h_nm, *h_len_dif; int32_t *h_cig_off; uint8_t *h_cig_n; int32_t *cig;
} lo_batch;
#ifdef __cplusplus
extern "C" {
#endif
/* parse <reads> + <reads>.seed.gem.map against the index `ix` into a batch (max_reads <= 0: all) */
int  lo_batch_load(const lo_index *ix, const char *reads, const lo_para *P, long max_reads, lo_batch *B);
void lo_batch_free(lo_batch *B);
/* run the oracle's per-read path over a batch and serialise every read in the result-stream format of
 * include/lamsa_hp.h.  *stream is malloc'ed; read_off/read_len/status have n_reads entries. */
int  lo_batch_align_stream(const lo_batch *B, const lo_ref *R, const lo_para *P, int n_threads,
                           int32_t **stream, int64_t *n_words, int64_t *read_off, int32_t *read_len, int32_t *status);
#ifdef __cplusplus
}
#endif
