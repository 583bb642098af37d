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
