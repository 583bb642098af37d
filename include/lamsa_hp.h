/*
 * lamsa_hp.h -- C-ABI of the MI355X hot-path library (liblamsa_hp.so).
 *
 * This is the drop-in boundary for LAMSA's per-read seed-chain-extend path: the batch
 * form of the reference's per-read worker
 *     int lamsa_main_aln(thread_aux_t *aux)            reference src/lamsa_aln.c:825-891
 * which the reference calls once per thread over a chunk of reads
 * (src/lamsa_aln.c:1143-1162).  Stages (2) chaining, (3) gap-fill/extension and the
 * second round (2')/(3') run on the GPU behind lamsa_hp_align_batch(); seeding,
 * FASTA/FASTQ + GEM-map parsing, the BWT rescue (stage 4), result ranking and SAM
 * output stay with the host caller, exactly as SURVEY.md section 8(b) draws the line.
 *
 * Plain pointers and sizes only; every buffer handed in is caller-owned host memory,
 * every buffer handed out is callee-owned and valid until the next call on the same
 * handle (or lamsa_hp_destroy).  One handle per GPU; a handle is not thread-safe.
 * All entry points return 0 on success or a negative LAMSA_HP_E* code -- never exit(),
 * unlike the reference (src/frag_check.c:173, src/bntseq.c:471, ...).
 */
#ifndef LAMSA_HP_H
#define LAMSA_HP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LAMSA_HP_OK          0
#define LAMSA_HP_ENODEV     -1   /* no usable HIP device / HIP runtime error            */
#define LAMSA_HP_EINVAL     -2   /* malformed arguments                                  */
#define LAMSA_HP_ENOMEM     -3   /* device allocation failed                             */
#define LAMSA_HP_EKERNEL    -4   /* kernel launch / execution failure                    */

/* per-read / per-job status bits (out arrays) */
#define LAMSA_HP_ST_OVERFLOW   1  /* a device work buffer was too small; result invalid   */
#define LAMSA_HP_ST_REFEXIT    2  /* input on which the reference itself exit(1)s         */
#define LAMSA_HP_ST_UNSUPPORTED 4 /* the read is beyond what the device keeps (more than 32767 seeds, or a seed hit with
                                     |len_dif| > 127): not aligned, empty result; the rest of the batch is unaffected */

/* Alignment parameters after presets: the fields of the reference's lamsa_aln_para
 * (src/lamsa_aln.h:386-436) read on the hot path.  Fill with lamsa_hp_para_init() +
 * overrides + lamsa_hp_para_finish(), which mirror init_aln_para (src/lamsa_aln.c:1281),
 * lamsa_set_aln_mode (:1342) and lamsa_fill_mat (:1331). */
typedef struct lamsa_hp_para {
    int32_t seed_len, seed_step, seed_inv;
    int32_t per_aln_m, first_loci_thd;
    int32_t SV_len_thd, ske_max;
    float   ovlp_rat;
    int32_t bwt_seed_len, bwt_max_len, bwt_min_len;
    int32_t split_len, split_pen, res_mul_max;
    int32_t hash_len, hash_key_len, hash_step, hash_size;
    int32_t match_dis, mismatch_thd;
    int32_t ins_gapo, ins_gape, del_gapo, del_gape;      /* global DP   (ksw_global2)      */
    int32_t ins_ext_o, ins_ext_e, del_ext_o, del_ext_e;  /* extension DP (ksw_extend_core) */
    int32_t match, mis;
    int32_t band_w, end_bonus, zdrop;
    float   id_rate;
    int32_t read_type;   /* 0 default, 1 pacbio, 2 ont2d (-T) */
    int32_t aln_mode;    /* bit0 overlapping seeds, bit1 high indel error */
} lamsa_hp_para;

void lamsa_hp_para_init(lamsa_hp_para *p);
void lamsa_hp_para_finish(lamsa_hp_para *p);

/* Packed reference: the reference's .pac bytes (2 bit/base, base k at
 * pac[k>>2] >> ((~k&3)<<1) & 3, src/bntseq.c:242) plus the contig table of .ann
 * (bntann1_t.offset/.len, src/bntseq.h).  Replaces the (bntseq_t*, uint8_t *pac) pair
 * handed to every worker (src/lamsa_aln.c:1135). */
typedef struct lamsa_hp_ref {
    const uint8_t *pac;       /* l_pac/4+1 bytes */
    int64_t        l_pac;
    int32_t        n_seqs;
    const int64_t *seq_offset;/* [n_seqs] */
    const int32_t *seq_len;   /* [n_seqs] */
} lamsa_hp_ref;

typedef struct lamsa_hp_handle lamsa_hp_handle;

/* Replaces aux_dp_init + the index hand-over of lamsa_aln_core (src/lamsa_aln.c:969-984,
 * 1130-1137): uploads parameters and the packed reference to HBM of `device_id`. */
int  lamsa_hp_create(lamsa_hp_handle **out, const lamsa_hp_para *para, const lamsa_hp_ref *ref, int device_id);
void lamsa_hp_destroy(lamsa_hp_handle *h);
const char *lamsa_hp_last_error(const lamsa_hp_handle *h);

/* ------------------------------------------------------------------------------------
 * Batched banded affine-gap DP primitives:
 *   kind 0: ksw_global2      src/ksw.c:543   (global gap penalties, band w)
 *   kind 1: ksw_extend_core  src/ksw.c:667   (extension gap penalties, band w, h0)
 *   kind 2: ksw_bi_extend    src/ksw.c:862   (lh0 = rh0 = h0; w ignored)
 *   kind 4, 5, 6: the same three by the one-job-per-lane routines the read path runs its small jobs on (queries up to 160, targets up
 *           to 256 bases without N).  When the handle's penalties do not keep 16-bit cells exact these kinds run on the routines of
 *           kinds 0-2, as the read path does.
 *   kind 8 .. 11: a job as the read path's wave-per-job launch runs it (targets without N): 8 = a junction's ksw_bi_extend (kind 2),
 *           9 = a seed gap's ksw_global2 (kind 0), 10 / 11 = a line's head / tail extension -- ksw_extend_r (src/ksw.c:820) /
 *           ksw_extend_c (:809) with (w, h0); unless the query was consumed the rest of it is appended as a soft clip, and the head's
 *           CIGAR is turned round (frag_head_bound_fix / frag_tail_bound_fix, src/frag_check.c:640-648, :699-703); score / qle / tle
 *           are the extension's.
 *   One class (0-2, 4-6, 8-11) per call.
 * Sequences are 1 byte/base codes 0..4; job i uses query seq[q_off[i] .. q_off[i]+qlen[i])
 * and target seq[t_off[i] .. t_off[i]+tlen[i]).
 * ---------------------------------------------------------------------------------- */
typedef struct lamsa_hp_dp_jobs {
    int32_t        n_jobs;
    const uint8_t *seq;  int64_t seq_bytes;
    const int64_t *q_off; const int32_t *qlen;
    const int64_t *t_off; const int32_t *tlen;
    const int32_t *kind, *w, *h0;
} lamsa_hp_dp_jobs;

typedef struct lamsa_hp_dp_out {       /* callee-owned, valid until the next call */
    const int32_t *score;              /* kind 0/1 (4/5, 9-11): DP score; kind 2 (6, 8): return flag */
    const int32_t *qle, *tle;          /* kind 1 (5, 10, 11) only                 */
    const int32_t *status;             /* LAMSA_HP_ST_* bits                      */
    const int64_t *cig_off;            /* [n_jobs+1] into cigar[]                 */
    const int32_t *cigar;              /* len<<4|op words, ops MIDNSH as in SAM   */
} lamsa_hp_dp_out;

int lamsa_hp_dp_batch(lamsa_hp_handle *h, const lamsa_hp_dp_jobs *jobs, lamsa_hp_dp_out *out);


/* ------------------------------------------------------------------------------------
 * The hot path proper: stages (2) sparse-DP chaining (frag_line_BCC, src/lamsa_dp_con.c:1305),
 * (3) gap-fill / split extension (frag_check, src/frag_check.c:856), and the second round
 * (2')/(3') on the read regions left uncovered (frag_line_remain :1252), for a batch of reads.
 * Replaces the loop body of lamsa_main_aln (src/lamsa_aln.c:857-871) for every read of a
 * chunk; the inputs are what the reference keeps in lamsa_seq_t / map_msg / map_t
 * (src/lamsa_aln.c:782-795, src/lamsa_aln.h:230-245), laid out as struct-of-arrays.
 * ---------------------------------------------------------------------------------- */
typedef struct lamsa_hp_batch {
    int32_t        n_reads;
    const int64_t *read_off;   /* [n_reads+1] into read_seq                                  */
    const uint8_t *read_seq;   /* 1 byte/base codes 0..4 (nst_nt4_table, src/bntseq.c:20)    */
    const int32_t *seed_all;   /* [n_reads] number of seeds of the read (APP->seed_all)      */
    const int32_t *last_len;   /* [n_reads] APP->last_len                                    */
    const int64_t *seed_off;   /* [n_reads+1]: read r owns seed slots [seed_off[r], seed_off[r+1]) = m_msg[0..seed_out) */
    const int32_t *seed_id;    /* [n_slots] 1-based seed index (map_msg.seed_id)             */
    const int64_t *hit_off;    /* [n_slots+1]: slot s owns hits [hit_off[s], hit_off[s+1]) = map[0..map_n) */
    const int64_t *h_pos;      /* [n_hits] map_t.offset (1-based leftmost reference coord)   */
    const int32_t *h_chr;      /* map_t.nchr (1-based contig id)                             */
    const int8_t  *h_strand;   /* map_t.nstrand (+1 / -1)                                    */
    const int16_t *h_nm;       /* map_t.NM                                                   */
    const int16_t *h_len_dif;  /* map_t.len_dif                                              */
    const int32_t *h_cig_off;  /* start of the seed CIGAR in cig[] / cig8[]; NULL: the CIGARs lie back to back in hit order
                                  (offset = sum of h_cig_n of the hits before; no 2^31 limit on n_cig then)           */
    const uint8_t *h_cig_n;    /* its length in elements                                     */
    const int32_t *cig;        /* seed CIGAR words, len << 4 | op (already reversed for '-' hits, src/gem_parse.c:267) */
    int64_t        n_cig;      /* elements in cig[] / cig8[]                                 */
    const uint8_t *cig8;       /* optional, instead of cig[]: one BYTE per element, op << 6 | len, op 0 M / 1 I / 2 D, len <= 63
                                  (a seed is seed_len = 50 bases: src/lamsa_aln.h:44) -- a quarter of the bytes over PCIe */
} lamsa_hp_batch;

/* Results: one int32 stream per read (callee-owned, valid until the next call):
 *   [0] status bits (LAMSA_HP_ST_*)  [1] lines of round 1 (a_res[0].l_n)  [2] lines of round 2 (a_res[1].l_n)
 *   per line : line_score, tol_score, tol_NM, n_res (= cur_res_n+1, 0 when every record was dropped)
 *   per res  : offset_lo, offset_hi, chr, nstrand (1 '+', 0 '-'), score (AS), NM, cigar_n, cigar words...
 * i.e. the fields of line_aln_res / res_t (src/frag_check.h:46-73) that aln_res_output and
 * rearr_aln_res consume. */
typedef struct lamsa_hp_result {
    const int32_t *stream; int64_t stream_words;
    const int64_t *read_off;      /* [n_reads] start of read r's stream */
    const int32_t *read_len;      /* [n_reads] its length in words      */
    const int32_t *read_status;   /* [n_reads] LAMSA_HP_ST_* bits       */
    const int32_t *read_tbases;   /* [n_reads] reference bases the read's DP jobs fetched from the packed reference (accounting) */
    const int32_t *read_work;     /* [4*n_reads] per read: DP cells updated, chaining edge classes evaluated, seed-CIGAR words read by the gap fill, 0
                                     (accounting: GCUPS, pair evaluations/s, the algorithmic bytes of the fill launches) */
} lamsa_hp_result;

int lamsa_hp_align_batch(lamsa_hp_handle *h, const lamsa_hp_batch *batch, lamsa_hp_result *res);

/* The same in two steps, so that a caller can keep a batch resident in HBM and overlap or
 * repeat the compute: upload copies the batch to the device, run aligns the resident batch
 * (res may be NULL: results stay on the device and are not fetched). */
int lamsa_hp_upload_batch(lamsa_hp_handle *h, const lamsa_hp_batch *batch);
int lamsa_hp_run_uploaded(lamsa_hp_handle *h, lamsa_hp_result *res);
/* run_uploaded in two halves, two runs deep: start queues a run of the resident batch and returns; finish waits for
 * the oldest run and fetches its results (res may be NULL).  The second run's waves start as the first one's drain. */
int lamsa_hp_start_uploaded(lamsa_hp_handle *h);
int lamsa_hp_finish_uploaded(lamsa_hp_handle *h, lamsa_hp_result *res);

/* The streaming form, for the chunk loop of lamsa_aln_core (src/lamsa_aln.c:1140-1170): the reference overlaps nothing
 * (read chunk -> threads -> join -> print); here up to two chunks are in flight per handle.
 *   submit   validates the batch, copies it to the device on the copy stream -- while the kernel of the previously
 *            submitted batch is still running -- and queues its kernel behind that one.  It returns when the copy is
 *            done (the caller's arrays may be reused) without waiting for any kernel.  LAMSA_HP_EINVAL when two
 *            batches are already in flight.
 *   collect  waits for the OLDEST submitted batch, runs its second pass if a read needs one and fetches the results
 *            (callee-owned, valid until the next collect / run on this handle).  lamsa_hp_last_kernel_ms() then
 *            refers to that batch.  LAMSA_HP_EINVAL when nothing is in flight.
 * Loop of a streaming caller:  submit(B0); for i = 1..: prepare B_i; submit(B_i); collect(R_{i-1}); write R_{i-1}.
 * lamsa_hp_upload_batch / run_uploaded / align_batch must not be called while a submitted batch is in flight. */
int lamsa_hp_submit_batch(lamsa_hp_handle *h, const lamsa_hp_batch *batch);
int lamsa_hp_collect_batch(lamsa_hp_handle *h, lamsa_hp_result *res);

/* Optional: allocate now, for both batches that can be in flight, what batches within the given bounds will need on the device
 * (scratch, inter-launch state, input and output buffers) instead of inside the first submit / align call, where allocations of
 * tens of GB cost seconds.  A caller that knows its chunk size calls it once after lamsa_hp_create -- from a thread of its own if it
 * has input to read meanwhile (the handle must not be used concurrently).  Batches beyond the bounds still work; they allocate. */
int lamsa_hp_reserve(lamsa_hp_handle *h, int32_t n_reads, int64_t n_bases, int64_t n_hits, int64_t n_cig, int32_t max_read_len, int32_t max_hits_per_read);

/* Page-locked host memory for the arrays of a lamsa_hp_batch: copies from it run at the full PCIe rate and need no
 * staging pass through the runtime's bounce buffers.  Optional -- ordinary memory works everywhere. */
void *lamsa_hp_host_alloc(size_t bytes);
void  lamsa_hp_host_free(void *p);

/* Wall time in milliseconds of the kernel(s) of the most recent call on this handle (for the streaming form: of the
 * batch just collected), measured with HIP events on the stream the kernels ran on; which = 0: the main pass,
 * 1: the second pass over the reads that overflowed their scratch (0 when none did); 2..6: the five launches the main
 * pass consists of (chaining round 1, gap fill of its lines, chaining round 2, gap fill of its lines, result assembly); 7..10: how long each of the first four spent draining (first wave
 * that found the queue empty -> last wave done, i.e. time with idle wave slots); 11, 12: lines filled in round 1 / round 2; 13: of "gap fill, round 1"
 * the part before the fill launch proper (job listing + the two DP launches), 14: the listing, 15: the wave-per-job DP launch; 16: wave jobs listed,
 * 17: their algorithmic bytes (MB: sequences read, CIGARs written), 18: lane jobs listed, 19: MB of CIGARs computed ahead of the fill, 20: DP cells the wave jobs
 * updated (millions). */
float lamsa_hp_last_kernel_ms(const lamsa_hp_handle *h, int which);

/* Cap the per-wave scratch slab of the first pass at `bytes` (0 = size it from the batch, the default).  The slab
 * replaces the reference's per-thread malloc/realloc arenas (src/lamsa_aln.c:960-990, src/frag_check.h:142-146);
 * a read that does not fit is not truncated: it is flagged and re-run by the second pass with 8x capacities, as
 * always.  For memory-constrained devices and for testing that second pass. */
int lamsa_hp_set_scratch_limit(lamsa_hp_handle *h, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
