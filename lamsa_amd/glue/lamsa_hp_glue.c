/* lamsa_hp_glue.c -- the binding a maintainer of the reference adds to run stages (2),(3),(2'),(3') of every chunk on
 * the GPU through liblamsa_hp.so (include/lamsa_hp.h).
 *
 * New code, written against the reference's types.  lamsa_seq_t and thread_aux_t are private typedefs of
 * src/lamsa_aln.c (:782-819), so this file is #included into lamsa_aln.c right after them (under #ifdef LAMSA_HP) rather
 * than compiled on its own; tools/apply_glue.py makes that edit and the three others INTEGRATION.md lists on a scratch
 * copy of the reference, and tests/test_glue_cpu.py builds the result against the CPU emulation of the C-ABI and checks
 * that the reference binary then writes the same SAM as before.
 *
 * What it does per chunk (lamsa_hp_glue_chunk): the reference's own text->map_t parsing (gem_map_msg, map_cal_msg --
 * unchanged, moved here from the worker, src/lamsa_aln.c:848-853), the structure-of-arrays batch, one
 * lamsa_hp_align_batch call, and the result stream unpacked into a_res[0] (round 1) and a_res[1] (round 2) of every
 * read -- the records frag_check (src/frag_check.c:856) would have written.  The worker threads then run what is left
 * of lamsa_main_aln: get_reg, bwt_aln_remain (stage 4), get_cov_f, rearr_aln_res.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lamsa_hp.h"

extern void push_res(line_aln_res *la);                              /* src/frag_check.c:228 */
extern void aln_reloc_res(aln_res *a_res, int line_n, int XA_m);     /* src/lamsa_aln.c:413  */

static lamsa_hp_handle *HP_GLUE;                 /* one handle per process and GPU */
static int64_t *HP_GLUE_off; static int32_t *HP_GLUE_len;

/* lamsa_aln_core, after the index is loaded (src/lamsa_aln.c:1139) */
static int lamsa_hp_glue_open(const lamsa_aln_para *AP, const bntseq_t *bns, const uint8_t *pac, int device)
{
    lamsa_hp_para P; int i, rc; lamsa_hp_ref R;
    memset(&P, 0, sizeof P);                     /* AP is already resolved (presets + options): a field-by-field copy */
    P.seed_len = AP->seed_len; P.seed_step = AP->seed_step; P.seed_inv = AP->seed_inv;
    P.per_aln_m = AP->per_aln_m; P.first_loci_thd = AP->first_loci_thd;
    P.SV_len_thd = AP->SV_len_thd; P.ske_max = AP->ske_max; P.ovlp_rat = AP->ovlp_rat;
    P.bwt_seed_len = AP->bwt_seed_len; P.bwt_max_len = AP->bwt_max_len; P.bwt_min_len = AP->bwt_min_len;
    P.split_len = AP->split_len; P.split_pen = AP->split_pen; P.res_mul_max = AP->res_mul_max;
    P.hash_len = AP->hash_len; P.hash_key_len = AP->hash_key_len; P.hash_step = AP->hash_step; P.hash_size = AP->hash_size;
    P.match_dis = AP->match_dis; P.mismatch_thd = AP->mismatch_thd;
    P.ins_gapo = AP->ins_gapo; P.ins_gape = AP->ins_gape; P.del_gapo = AP->del_gapo; P.del_gape = AP->del_gape;
    P.ins_ext_o = AP->ins_ext_o; P.ins_ext_e = AP->ins_ext_e; P.del_ext_o = AP->del_ext_o; P.del_ext_e = AP->del_ext_e;
    P.match = AP->match; P.mis = AP->mis; P.band_w = AP->band_w; P.end_bonus = AP->end_bonus; P.zdrop = AP->zdrop;
    P.id_rate = AP->id_rate; P.read_type = AP->read_type; P.aln_mode = AP->aln_mode;

    HP_GLUE_off = (int64_t*)malloc(bns->n_seqs * sizeof(int64_t)); HP_GLUE_len = (int32_t*)malloc(bns->n_seqs * sizeof(int32_t));
    for (i = 0; i < bns->n_seqs; ++i) { HP_GLUE_off[i] = bns->anns[i].offset; HP_GLUE_len[i] = bns->anns[i].len; }
    R.pac = pac; R.l_pac = bns->l_pac; R.n_seqs = bns->n_seqs; R.seq_offset = HP_GLUE_off; R.seq_len = HP_GLUE_len;
    rc = lamsa_hp_create(&HP_GLUE, &P, &R, device);      /* copies pac and the tables to HBM */
    if (rc != LAMSA_HP_OK) fprintf(stderr, "[lamsa_hp] no usable MI355X (code %d)\n", rc);
    return rc;
}

static void lamsa_hp_glue_close(void)
{
    lamsa_hp_destroy(HP_GLUE); HP_GLUE = NULL;
    free(HP_GLUE_off); free(HP_GLUE_len);
}

/* one chunk: called by lamsa_aln_core before it hands the chunk to its worker threads (src/lamsa_aln.c:1144) */
static int lamsa_hp_glue_chunk(lamsa_seq_t *ls, kseq_t *ks, int n_seqs, const lamsa_aln_para *AP, bntseq_t *bns)
{
    lamsa_hp_batch B; lamsa_hp_result R;
    int64_t n_base = 0, n_slot = 0, n_hit = 0, n_cig = 0; int r, s, k, c, rc;

    /* pass 1: parse (the reference's own routines) and count */
    for (r = 0; r < n_seqs; ++r) {
        n_base += ks[r].seq.l; n_slot += ls[r].APP->seed_out;
        for (s = 0; s < ls[r].APP->seed_out; ++s) {
            map_msg *m = ls[r].m_msg + s;
            gem_map_msg(m, AP->per_aln_m); map_cal_msg(m, bns);
            n_hit += m->map_n;
            for (k = 0; k < m->map_n; ++k) n_cig += m->map[k].cigar->cigar_n;
        }
    }
    /* pass 2: structure of arrays */
    int64_t *read_off = (int64_t*)malloc((n_seqs + 1) * sizeof(int64_t)), *seed_off = (int64_t*)malloc((n_seqs + 1) * sizeof(int64_t));
    int64_t *hit_off = (int64_t*)malloc((n_slot + 1) * sizeof(int64_t)), *h_pos = (int64_t*)malloc((n_hit + 1) * sizeof(int64_t));
    int32_t *seed_all = (int32_t*)malloc((n_seqs + 1) * sizeof(int32_t)), *last_len = (int32_t*)malloc((n_seqs + 1) * sizeof(int32_t));
    int32_t *seed_id = (int32_t*)malloc((n_slot + 1) * sizeof(int32_t)), *h_chr = (int32_t*)malloc((n_hit + 1) * sizeof(int32_t));
    int32_t *h_cig_off = (int32_t*)malloc((n_hit + 1) * sizeof(int32_t)), *cig = (int32_t*)malloc((n_cig + 1) * sizeof(int32_t));
    int16_t *h_nm = (int16_t*)malloc((n_hit + 1) * sizeof(int16_t)), *h_len_dif = (int16_t*)malloc((n_hit + 1) * sizeof(int16_t));
    int8_t *h_strand = (int8_t*)malloc(n_hit + 1); uint8_t *h_cig_n = (uint8_t*)malloc(n_hit + 1), *read_seq = (uint8_t*)malloc(n_base + 1);
    if (n_cig >= ((int64_t)1 << 31)) { fprintf(stderr, "[lamsa_hp] chunk too large for 32-bit CIGAR offsets: lower CHUNK_READ_N\n"); exit(1); }
    n_base = n_slot = n_hit = n_cig = 0;
    for (r = 0; r < n_seqs; ++r) {
        read_off[r] = n_base;
        for (k = 0; k < (int)ks[r].seq.l; ++k) read_seq[n_base++] = nst_nt4_table[(int)ks[r].seq.s[k]];
        seed_all[r] = ls[r].APP->seed_all; last_len[r] = ls[r].APP->last_len; seed_off[r] = n_slot;
        for (s = 0; s < ls[r].APP->seed_out; ++s) {
            const map_msg *m = ls[r].m_msg + s;
            seed_id[n_slot] = m->seed_id; hit_off[n_slot++] = n_hit;
            for (k = 0; k < m->map_n; ++k, ++n_hit) {
                const map_t *t = m->map + k;
                h_pos[n_hit] = t->offset; h_chr[n_hit] = t->nchr; h_strand[n_hit] = t->nstrand;
                h_nm[n_hit] = (int16_t)t->NM; h_len_dif[n_hit] = (int16_t)t->len_dif;
                h_cig_off[n_hit] = (int32_t)n_cig; h_cig_n[n_hit] = (uint8_t)t->cigar->cigar_n;
                for (c = 0; c < t->cigar->cigar_n; ++c) cig[n_cig++] = (int32_t)t->cigar->cigar[c];
            }
        }
    }
    read_off[n_seqs] = n_base; seed_off[n_seqs] = n_slot; hit_off[n_slot] = n_hit;
    memset(&B, 0, sizeof B);
    B.n_reads = n_seqs; B.read_off = read_off; B.read_seq = read_seq; B.seed_all = seed_all; B.last_len = last_len;
    B.seed_off = seed_off; B.seed_id = seed_id; B.hit_off = hit_off; B.h_pos = h_pos; B.h_chr = h_chr; B.h_strand = h_strand;
    B.h_nm = h_nm; B.h_len_dif = h_len_dif; B.h_cig_off = h_cig_off; B.h_cig_n = h_cig_n; B.cig = cig; B.n_cig = n_cig;

    rc = lamsa_hp_align_batch(HP_GLUE, &B, &R);
    if (rc != LAMSA_HP_OK) { fprintf(stderr, "[lamsa_hp] %s\n", lamsa_hp_last_error(HP_GLUE)); exit(1); }

    /* result stream -> a_res[0] (round 1), a_res[1] (round 2): what frag_check leaves behind (src/frag_check.c:869-955) */
    for (r = 0; r < n_seqs; ++r) {
        const int32_t *w = R.stream + R.read_off[r], *p = w + 3; int st, l, j;
        if (w[0] & LAMSA_HP_ST_REFEXIT) {        /* the reference exit(1)s at this read too (ref window outside the contig) */
            fprintf(stderr, "[lamsa_hp] %s: reference window outside the contig\n", ks[r].name.s); exit(1);
        }
        if (w[0] != 0) { fprintf(stderr, "[lamsa_hp] %s: status %d\n", ks[r].name.s, w[0]); exit(1); }
        aln_reset_res(ls[r].a_res, 3, ks[r].seq.l);
        for (st = 0; st < 2; ++st) {
            aln_res *a = ls[r].a_res + st; const int line_n = w[1 + st];
            if (line_n > a->l_m) aln_reloc_res(a, line_n, AP->res_mul_max);
            a->l_n = line_n;
            for (l = 0; l < line_n; ++l) {
                line_aln_res *la = a->la + l; const int n_res = p[3];
                la->line_score = p[0]; la->tol_score = p[1]; la->tol_NM = p[2]; la->cur_res_n = 0; p += 4;
                for (j = 0; j < la->res_m; ++j) la->res[j].cigar_len = 0;
                for (j = 0; j < n_res; ++j) {
                    res_t *q;
                    if (j > 0) push_res(la);     /* grows la->res as the reference does and moves cur_res_n on */
                    q = la->res + j;
                    q->offset = (ref_pos_t)(((uint64_t)(uint32_t)p[1] << 32) | (uint32_t)p[0]); q->chr = p[2]; q->nstrand = (int8_t)p[3];
                    q->score = p[4]; q->NM = p[5]; q->cigar_len = 0;
                    _push_cigar(&q->cigar, &q->cigar_len, &q->c_m, (cigar32_t*)(p + 7), p[6]);
                    p += 7 + p[6];
                }
                if (n_res == 0) la->cur_res_n = -1;      /* every record dropped: src/frag_check.c:844-850 */
            }
        }
    }
    free(read_off); free(seed_off); free(hit_off); free(h_pos); free(seed_all); free(last_len); free(seed_id); free(h_chr);
    free(h_cig_off); free(cig); free(h_nm); free(h_len_dif); free(h_strand); free(h_cig_n); free(read_seq);
    return 0;
}
