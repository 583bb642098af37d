/* lo_read.h -- per-read data of the oracle (see lo.h header note: test infrastructure only). */
#ifndef LAMSA_ORACLE_LO_READ_H_
#define LAMSA_ORACLE_LO_READ_H_
#include "lo.h"

#ifdef __cplusplus
extern "C" {
#endif

/* packed reference: .pac + contig table (src/bntseq.h, src/bntseq.c:242,465) */
typedef struct {
    const uint8_t *pac; int64_t l_pac; int n_seqs;
    const int64_t *seq_offset; const int32_t *seq_len;
} lo_ref;

/* one seed hit: map_t without the 1 KB name (src/lamsa_aln.h:230-239) */
typedef struct {
    int64_t offset;          /* 1-based leftmost reference coordinate, forward strand */
    int32_t chr;             /* 1-based contig id (nchr) */
    int32_t strand;          /* +1 / -1 (nstrand) */
    int32_t NM, len_dif, bmax;
    int32_t cig_off, cig_n;  /* into lo_seeds.cig */
} lo_hit;

/* the seeds of one read that have a GEM map line: m_msg[0..seed_out) (src/lamsa_aln.c:945-952) */
typedef struct {
    int seed_all, seed_out, last_len, read_len;
    int32_t *seed_id;        /* [seed_out] 1-based original index (flipped in place for '-' lines, frag_check.c:926) */
    int32_t *hit_off;        /* [seed_out+1] */
    lo_hit  *hit;
    lo_cig  *cig;
} lo_seeds;

typedef struct { int x, y; } lo_xy;           /* (seed slot, hit index within the seed) = line_node */

/* chaining DP cell: frag_dp_node (src/lamsa_aln.h:352-373) */
typedef struct {
    int son_flag; lo_xy from;
    int in_de, son_n, son_m; lo_xy *son;
    int max_score, max_NM; lo_xy max_node;
    int score, tol_NM;
    int match_flag, dp_flag, node_n;
} lo_node;

/* one fragment of a line: frag_aln_msg (src/frag_check.h:13-34), seeds kept in the reference's order */
typedef struct { int chr, strand; int seed_n; lo_xy *seed; } lo_frag;
/* one line (skeleton): frag_msg (src/frag_check.h:36-44) */
typedef struct { int frag_n; lo_frag *frag; int line_score, left_bound, right_bound; } lo_fline;

/* alignment record: res_t (src/frag_check.h:46-59) */
typedef struct {
    int64_t offset; int chr; int nstrand;      /* nstrand: 1 '+', 0 '-' */
    lo_cigv cig; int64_t refend; int readend;
    int score, NM, reg_beg, reg_end;
} lo_res;
/* per-line results: line_aln_res (src/frag_check.h:61-73) */
typedef struct {
    int line_score; int merg_x, merg_y;
    int *XA_stage, *XA_line, *XA_res; int XA_n, XA_m;   /* (stage,line,record) triples instead of pointers */
    int res_m, cur_res_n; lo_res *res;
    int tol_score, tol_NM; uint8_t mapQ;
} lo_lres;
/* aln_res (src/frag_check.h:75-81) */
typedef struct { int l_n, l_m; lo_lres *la; int read_len; float cov_f; } lo_ares;

/* covered read intervals: reg_t / aln_reg (src/lamsa_aln.h:288-315) */
typedef struct { int is_rev, chr; int64_t ref_pos; } lo_regb;
typedef struct { lo_regb *ref_beg, *ref_end; int beg_n, end_n, beg_m, end_m; int beg, end; } lo_reg;
typedef struct { lo_reg *reg; int reg_n, reg_m, read_len; } lo_areg;

/* ---- lo_ref.c ---- */
/* pac2fa_core (src/bntseq.c:465-477): returns 0, or -1 where the reference exit(1)s; *len may shrink */
int lo_pac_fetch(const lo_ref *R, int chr, int64_t start0, int32_t *len, uint8_t *dst);

/* ---- lo_chain.c ---- */
int lo_edge_flag(const lo_seeds *S, const lo_para *P, int pre, int pre_a, int i, int j);   /* get_fseed_dis */
/* frag_line_BCC (src/lamsa_dp_con.c:1305): returns number of lines, *out malloc'ed */
int lo_chain_first(lo_seeds *S, const lo_para *P, lo_node *nodes, lo_fline **out);
/* frag_line_remain (src/lamsa_dp_con.c:1252) */
int lo_chain_remain(lo_areg *a_reg, lo_seeds *S, const lo_para *P, lo_node *nodes, lo_fline **out);
void lo_flines_free(lo_fline *f, int n);

/* ---- lo_reg.c ---- */
lo_areg *lo_areg_new(int read_len);
void lo_areg_free(lo_areg *a);
int  lo_get_remain_reg(lo_areg *a, lo_areg *remain, const lo_para *P, int min_thd, int max_thd);  /* src/lamsa_aln.c:550 */
void lo_get_reg(lo_ares *res, lo_areg *reg);                                                      /* :597 */
float lo_get_cov_f(lo_ares *res3, lo_areg *reg);                                                  /* :639 */
void lo_rearr(lo_ares *res3, int n, float ovlp_r);                                                /* :654 */

/* ---- lo_fill.c ---- */
void lo_ares_init(lo_ares *a, int res_mul_max);
void lo_ares_reset(lo_ares *a, int read_len);
void lo_ares_free(lo_ares *a);
/* frag_check (src/frag_check.c:856): consumes lines (does not free them). returns 0, or -1 on a reference exit */
int lo_frag_check(lo_seeds *S, lo_fline *lines, int line_n, lo_ares *a_res, const lo_ref *R,
                  const uint8_t *read, uint8_t **rc_read, const lo_para *P);

/* ---- lo_split.c ---- */
/* split_indel_map (src/split_mapping.c:829) */
int lo_split_indel_map(lo_cigv *out, const uint8_t *read, int read_len, const uint8_t *ref, int ref_len,
                       int ref_offset, const lo_para *P);

/* ---- lo_align.c ---- */
/* stages (2),(3),(2'),(3') of lamsa_main_aln (src/lamsa_aln.c:857-871); res3[0..1] filled, res3[2] left empty.
 * returns 0, or -1 when the reference would exit(1) on this read. */
int lo_align_read(lo_seeds *S, const uint8_t *read, const lo_ref *R, const lo_para *P, lo_ares *res3, lo_areg *a_reg);

#ifdef __cplusplus
}
#endif
#endif
