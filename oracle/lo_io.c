/* lo_io.c -- file handling + whole-run driver of the oracle (see lo.h: test infrastructure).
 *
 * Restates the host side of `lamsa aln -N -I`:
 *   lo_index_load   <- bns_restore_core + .pac read   src/bntseq.c:114-166, src/lamsa_aln.c:1237-1239
 *   read_record     <- kseq_read with the modified name separator (KS_SEP_REF)  src/kseq.h:179-225
 *   lo_parse_hits   <- gem_map_read / gem_map_msg / md2cigar / map_cal_msg      src/gem_parse.c:74-286, lamsa_aln.c:767
 *   emit_sam        <- aln_res_output / print_sam_header                         src/lamsa_aln.c:1001-1100,1215
 *   lo_run_aln      <- lamsa_aln_core chunk loop                                 src/lamsa_aln.c:1116-1177
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <ctype.h>
#include <time.h>
#include <pthread.h>
#include <zlib.h>
#include "lo_io.h"

#define LINE_SIZE 65536                       /* src/gem_parse.h */

/* ---------------------------------------------------------------- index */
int lo_index_load(lo_index *ix, const char *prefix)
{
    char fn[2048], str[8192];
    memset(ix, 0, sizeof(*ix));
    snprintf(fn, sizeof fn, "%s.ann", prefix);
    FILE *fp = fopen(fn, "r");
    if (!fp) { fprintf(stderr, "[lo_io] cannot open %s\n", fn); return -1; }
    long long l_pac; int n_seqs; unsigned seed;
    if (fscanf(fp, "%lld%d%u", &l_pac, &n_seqs, &seed) != 3) { fclose(fp); return -1; }
    ix->name = (char**)calloc((size_t)n_seqs, sizeof(char*));
    ix->off = (int64_t*)calloc((size_t)n_seqs, sizeof(int64_t)); ix->len = (int32_t*)calloc((size_t)n_seqs, sizeof(int32_t));
    for (int i = 0; i < n_seqs; ++i) {
        unsigned gi; int c, n_ambs; long long off; char *q = str;
        if (fscanf(fp, "%u", &gi) != 1) { fclose(fp); return -1; }
        while (q - str < (long)sizeof(str) - 1 && (c = fgetc(fp)) != '\n' && c != EOF) *q++ = (char)c;
        *q = 0;
        ix->name[i] = strdup(str + 1);
        if (fscanf(fp, "%lld%d%d", &off, &ix->len[i], &n_ambs) != 3) { fclose(fp); return -1; }
        ix->off[i] = off;
    }
    fclose(fp);
    snprintf(fn, sizeof fn, "%s.pac", prefix);
    fp = fopen(fn, "rb");
    if (!fp) { fprintf(stderr, "[lo_io] cannot open %s\n", fn); return -1; }
    size_t nb = (size_t)(l_pac / 4 + 1);
    ix->pac = (uint8_t*)calloc(nb + 16, 1);
    if (fread(ix->pac, 1, nb, fp) == 0 && nb > 1) { fclose(fp); return -1; }
    fclose(fp);
    ix->ref.pac = ix->pac; ix->ref.l_pac = l_pac; ix->ref.n_seqs = n_seqs; ix->ref.seq_offset = ix->off; ix->ref.seq_len = ix->len;
    return 0;
}
void lo_index_free(lo_index *ix)
{
    for (int i = 0; i < ix->ref.n_seqs; ++i) free(ix->name[i]);
    free(ix->name); free(ix->off); free(ix->len); free(ix->pac);
}
static int chr_id(const lo_index *ix, const char *name)
{   /* bns_get_rid, src/bntseq.c:490 */
    for (int i = 0; i < ix->ref.n_seqs; ++i) if (strcmp(name, ix->name[i]) == 0) return i + 1;
    return -1;
}

/* ---------------------------------------------------------------- GEM hits */
static void cigp(lo_cig **c, int *n, int *m, int base, lo_cig w)
{   /* _push_cigar1 on the slice starting at `base` */
    if ((w >> 4) == 0) return;
    if (*n > base && ((*c)[*n - 1] & 0xf) == (w & 0xf)) { (*c)[*n - 1] += (w >> 4) << 4; return; }
    if (*n == *m) { *m = *m ? *m << 1 : 64; *c = (lo_cig*)realloc(*c, sizeof(lo_cig) * (size_t)*m); }
    (*c)[(*n)++] = w;
}

static void gigar_to_cigar(const char *md, lo_hit *h, lo_cig **cig, int *cig_n, int *cig_m)
{   /* md2cigar, src/gem_parse.c:74-112 */
    const int len = (int)strlen(md), base = *cig_n;
    int i = 0, bd = 0, bi = 0;
    h->NM = 0;
    while (i < len) {
        if (md[i] == '>') {
            int k = i + 1, n = 0;
            while (md[k] && md[k] != '+' && md[k] != '-') ++k;
            n = atoi(md + i + 1);
            if (md[k] == '+') { bd += n; cigp(cig, cig_n, cig_m, base, (n << 4) | LO_D); }
            else { bi += n; cigp(cig, cig_n, cig_m, base, (n << 4) | LO_I); }
            h->NM += n;
            int d = 1; for (int t = n; t >= 10; t /= 10) ++d;
            i += d + 2;
        } else {
            int m = 0, mm = 0, run = 0, in_run = 0;
            for (; md[i] && md[i] != '>'; ++i) {
                if (md[i] >= 'A' && md[i] <= 'T') { ++mm; if (in_run) { m += run; run = 0; in_run = 0; } }
                else if (md[i] == ' ') { if (in_run) { m += run; run = 0; in_run = 0; } }
                else { /* atoi() semantics on a digit run */ if (isdigit((unsigned char)md[i])) { run = in_run ? run * 10 + (md[i] - '0') : (md[i] - '0'); in_run = 1; } else if (in_run) { m += run; run = 0; in_run = 0; } }
            }
            if (in_run) m += run;
            cigp(cig, cig_n, cig_m, base, ((m + mm) << 4) | LO_M);
            h->NM += mm;
        }
    }
    h->len_dif = bd - bi; h->bmax = bd > bi ? bd : bi;
    h->cig_off = base; h->cig_n = *cig_n - base;
}

int lo_parse_hits(const char *s, const lo_index *ix, int max_n, lo_hit **hits, int *hit_n, lo_cig **cig, int *cig_n, int *cig_m)
{   /* gem_map_msg, src/gem_parse.c:230-286; appends to hits[*hit_n..]; hit array is grown by the caller's doubling */
    const int h0 = *hit_n, c0 = *cig_n;
    int n = 0;
    char chr[1024], os[128], md[2048], strand;
    const char *p = s;
    while (*p) {
        while (*p == ',') ++p;
        if (!*p) break;
        const char *e = p;
        while (*e && *e != ',') ++e;
        if (n >= max_n) { *hit_n = h0; *cig_n = c0; return 0; }           /* more than -p hits: seed emptied, :243-246 */
        char tok[4096];
        size_t tl = (size_t)(e - p); if (tl >= sizeof tok) tl = sizeof tok - 1;
        memcpy(tok, p, tl); tok[tl] = 0;
        md[0] = 0;
        sscanf(tok, "%1023[^:]:%c:%127[^:]:%2047[^:]", chr, &strand, os, md);
        *hits = (lo_hit*)realloc(*hits, sizeof(lo_hit) * (size_t)(*hit_n + 1));
        lo_hit *h = &(*hits)[*hit_n];
        memset(h, 0, sizeof(*h));
        gigar_to_cigar(md, h, cig, cig_n, cig_m);
        if (strand == '-') lo_cig_invert(*cig + h->cig_off, h->cig_n);
        h->offset = atoll(os); h->strand = strand == '+' ? 1 : -1; h->chr = chr_id(ix, chr);
        ++*hit_n; ++n;
        p = e;
    }
    return n;
}

/* ---------------------------------------------------------------- reads */
typedef struct { char *name, *seq, *qual; int l, has_qual; } rec_t;
typedef struct { gzFile f; char *buf; int cap; int have; } rdr_t;

static int rdr_line(rdr_t *r)
{   /* next line without the newline into r->buf; 0 on EOF */
    if (r->have) { r->have = 0; return 1; }
    int n = 0;
    for (;;) {
        if (n + 2 >= r->cap) { r->cap = r->cap ? r->cap * 2 : 1 << 16; r->buf = (char*)realloc(r->buf, (size_t)r->cap); }
        if (!gzgets(r->f, r->buf + n, r->cap - n)) { if (n == 0) return 0; break; }
        n += (int)strlen(r->buf + n);
        if (n > 0 && r->buf[n - 1] == '\n') break;
    }
    while (n > 0 && (r->buf[n - 1] == '\n' || r->buf[n - 1] == '\r')) r->buf[--n] = 0;
    return 1;
}
static int read_record(rdr_t *r, rec_t *x)
{   /* FASTA / FASTQ; the name ends at '\n', ':' or ',' (KS_SEP_REF, src/kseq.h:42,190) */
    x->l = 0; x->has_qual = 0;
    do { if (!rdr_line(r)) return 0; } while (r->buf[0] != '>' && r->buf[0] != '@');
    const int fastq = r->buf[0] == '@';
    size_t nl = strcspn(r->buf + 1, ":,");
    free(x->name); x->name = (char*)malloc(nl + 1); memcpy(x->name, r->buf + 1, nl); x->name[nl] = 0;
    size_t cap = 0; free(x->seq); x->seq = NULL;
    while (rdr_line(r)) {
        if (r->buf[0] == '>' || r->buf[0] == '+' || r->buf[0] == '@') { if (r->buf[0] != '+') r->have = 1; break; }
        size_t ll = strlen(r->buf);
        if ((size_t)x->l + ll + 1 > cap) { cap = ((size_t)x->l + ll + 1) * 2; x->seq = (char*)realloc(x->seq, cap); }
        for (size_t i = 0; i < ll; ++i) if (isgraph((unsigned char)r->buf[i])) x->seq[x->l++] = r->buf[i];
    }
    if (!x->seq) x->seq = (char*)calloc(1, 1);
    x->seq[x->l] = 0;
    if (fastq && !r->have) {
        free(x->qual); x->qual = (char*)malloc((size_t)x->l + 2); int ql = 0;
        while (ql < x->l && rdr_line(r)) { size_t ll = strlen(r->buf); for (size_t i = 0; i < ll && ql < x->l; ++i) x->qual[ql++] = r->buf[i]; }
        x->qual[ql] = 0; x->has_qual = 1;
    }
    return 1;
}

static const uint8_t *nt4(void)
{
    static uint8_t t[256]; static int init = 0;
    if (!init) { memset(t, 4, 256); t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3; t['-'] = 5; init = 1; }
    return t;
}
static char comp_char(char c)
{   /* com_nst_nt256_table, src/bntseq.c */
    switch (c) { case 'A': return 'T'; case 'a': return 't'; case 'C': return 'G'; case 'c': return 'g';
                 case 'G': return 'C'; case 'g': return 'c'; case 'T': return 'A'; case 't': return 'a'; default: return c; }
}

/* ---------------------------------------------------------------- SAM */
typedef struct { char *s; size_t l, m; } sbuf;
static void sb_put(sbuf *b, const char *p, size_t n) { if (b->l + n + 1 > b->m) { b->m = (b->l + n + 1) * 2; b->s = (char*)realloc(b->s, b->m); } memcpy(b->s + b->l, p, n); b->l += n; b->s[b->l] = 0; }
static void sb_printf(sbuf *b, const char *fmt, ...);
#include <stdarg.h>
static void sb_printf(sbuf *b, const char *fmt, ...)
{
    char tmp[512]; va_list ap; va_start(ap, fmt); int n = vsnprintf(tmp, sizeof tmp, fmt, ap); va_end(ap);
    if (n < (int)sizeof tmp) { sb_put(b, tmp, (size_t)n); return; }
    char *big = (char*)malloc((size_t)n + 1); va_start(ap, fmt); vsnprintf(big, (size_t)n + 1, fmt, ap); va_end(ap); sb_put(b, big, (size_t)n); free(big);
}

static void emit_sam(sbuf *o, const lo_para *P, lo_ares *res, const rec_t *rd, const lo_index *ix)
{   /* aln_res_output, src/lamsa_aln.c:1001-1100 (the `-C` + reverse-strand QUAL loop of :1043 never terminates
     * in the reference; here QUAL is reversed, the evident intent -- documented divergence) */
    static const char OPS[] = "MIDNSHP=XB", OPS_HC[] = "MIDNHHP=XB";
    int all = 0, prim = 0;
    const char *qual = rd->has_qual ? rd->qual : NULL;
    for (int n = 0; n < 3; ++n) {
        lo_ares *p = res + n;
        for (int i = 0; i < p->l_n; ++i) {
            lo_lres *la = &p->la[i];
            if (la->merg_x != 1) continue;
            for (int j = 0; j <= la->cur_res_n; ++j) {
                lo_res *r = &la->res[j];
                all++;
                int flag = r->nstrand ? 0 : 0x10;
                const int soft = (prim == 0 || P->supp_soft);
                if (!soft) flag |= 0x800;
                sb_printf(o, "%s\t%d\t%s\t%lld\t%d\t", rd->name, flag, ix->name[r->chr - 1], (long long)r->offset, la->mapQ);
                for (int l = 0; l < r->cig.n; ++l) sb_printf(o, "%d%c", r->cig.c[l] >> 4, (soft ? OPS : OPS_HC)[r->cig.c[l] & 0xf]);
                sb_put(o, "\t*\t0\t0", 6);
                if (soft) {
                    sb_put(o, "\t", 1);
                    if (r->nstrand == 1) sb_put(o, rd->seq, (size_t)rd->l);
                    else for (int si = res->read_len - 1; si >= 0; --si) { char c = comp_char(rd->seq[si]); sb_put(o, &c, 1); }
                    sb_put(o, "\t", 1);
                    if (qual && P->comm) { if (r->nstrand == 1) sb_put(o, qual, strlen(qual)); else for (int si = res->read_len - 1; si >= 0; --si) sb_put(o, &qual[si], 1); }
                    else sb_put(o, "*", 1);
                    prim = 1;
                } else {
                    sb_put(o, "\t", 1);
                    if (r->nstrand == 1) for (int si = r->reg_beg - 1; si < r->reg_end; ++si) sb_put(o, &rd->seq[si], 1);
                    else for (int si = r->reg_end - 1; si >= r->reg_beg - 1; --si) { char c = comp_char(rd->seq[si]); sb_put(o, &c, 1); }
                    sb_put(o, "\t", 1);
                    if (qual && P->comm) { if (r->nstrand == 1) for (int si = r->reg_beg - 1; si < r->reg_end; ++si) sb_put(o, &qual[si], 1); else for (int si = r->reg_end - 1; si >= r->reg_beg - 1; --si) sb_put(o, &qual[si], 1); }
                    else sb_put(o, "*", 1);
                }
                sb_printf(o, "\tNM:i:%d\tAS:i:%d", r->NM, r->score);
                if (j == 0 && la->XA_n > 0) {
                    sb_put(o, "\tXA:Z:", 6);
                    for (int l = 0; l < la->XA_n; ++l) {
                        lo_res *x = &res[la->XA_stage[l]].la[la->XA_line[l]].res[la->XA_res[l]];
                        sb_printf(o, "%s,%c%lld,", ix->name[x->chr - 1], "-+"[x->nstrand], (long long)x->offset);
                        for (int m = 0; m < x->cig.n; ++m) sb_printf(o, "%d%c", x->cig.c[m] >> 4, OPS[x->cig.c[m] & 0xf]);
                        sb_printf(o, ",%d;", x->NM);
                    }
                }
                sb_put(o, "\n", 1);
            }
        }
    }
    if (all == 0) {
        sb_printf(o, "%s\t%d\t*\t%lld\t%d\t*\t*\t0\t0\t", rd->name, 4, 0LL, 0);
        sb_put(o, rd->seq, (size_t)rd->l); sb_put(o, "\t", 1);
        if (qual) sb_put(o, qual, strlen(qual)); else sb_put(o, "*", 1);
        sb_put(o, "\n", 1);
    }
}

/* ---------------------------------------------------------------- chunk loop */
typedef struct {
    rec_t rd; lo_seeds S; uint8_t *bseq; sbuf sam; int failed;
} unit_t;
typedef struct {
    unit_t *u; int n; volatile int next; pthread_mutex_t mu;
    const lo_index *ix; const lo_para *P;
} work_t;

static void process_unit(unit_t *u, const lo_index *ix, const lo_para *P)
{
    lo_ares res3[3];
    for (int i = 0; i < 3; ++i) lo_ares_init(&res3[i], P->res_mul_max);
    lo_areg *a_reg = lo_areg_new(u->rd.l);
    u->failed = lo_align_read(&u->S, u->bseq, &ix->ref, P, res3, a_reg) < 0;
    u->sam.l = 0;
    if (u->failed) { for (int i = 0; i < 3; ++i) { lo_ares_free(&res3[i]); lo_ares_init(&res3[i], P->res_mul_max); lo_ares_reset(&res3[i], u->rd.l); } }
    res3[0].cov_f = lo_get_cov_f(res3, a_reg);                      /* lamsa_aln.c:876 */
    lo_rearr(res3, 3, P->ovlp_rat);                                 /* :878 */
    emit_sam(&u->sam, P, res3, &u->rd, ix);
    lo_areg_free(a_reg);
    for (int i = 0; i < 3; ++i) lo_ares_free(&res3[i]);
}
static void *worker(void *arg)
{
    work_t *w = (work_t*)arg;
    for (;;) {
        pthread_mutex_lock(&w->mu); int i = w->next++; pthread_mutex_unlock(&w->mu);
        if (i >= w->n) break;
        process_unit(&w->u[i], w->ix, w->P);
    }
    return NULL;
}

int lo_run_aln(const char *ref_prefix, const char *reads, lo_para *P, FILE *out, const char *pg_line, int n_threads, long max_reads,
               double *aln_seconds, long *n_reads_out, long *n_bases_out)
{
    lo_index ix;
    if (lo_index_load(&ix, ref_prefix) < 0) return -1;
    char fn[2048];
    snprintf(fn, sizeof fn, "%s.seed.gem.map", reads);
    FILE *mapf = fopen(fn, "r");
    if (!mapf) { fprintf(stderr, "[lo_io] cannot open %s\n", fn); return -1; }
    rdr_t rr; memset(&rr, 0, sizeof rr);
    rr.f = gzopen(reads, "r");
    if (!rr.f) { fprintf(stderr, "[lo_io] cannot open %s\n", reads); return -1; }
    if (out) {
        for (int i = 0; i < ix.ref.n_seqs; ++i) fprintf(out, "@SQ\tSN:%s\tLN:%d\n", ix.name[i], ix.len[i]);   /* print_sam_header, :1215 */
        if (pg_line) fprintf(out, "%s\n", pg_line);
    }
    const int CH = 512;
    unit_t *u = (unit_t*)calloc((size_t)CH, sizeof(unit_t));
    char *line = (char*)malloc(LINE_SIZE);
    const uint8_t *t4 = nt4();
    long n_reads = 0, n_bases = 0; double secs = 0;
    int eof = 0;
    while (!eof) {
        int n = 0;
        while (n < CH && (max_reads <= 0 || n_reads + n < max_reads)) {
            unit_t *x = &u[n];
            if (!read_record(&rr, &x->rd)) { eof = 1; break; }
            const int L = x->rd.l;
            lo_seeds *S = &x->S;
            free(S->seed_id); free(S->hit_off); free(S->hit); free(S->cig); memset(S, 0, sizeof(*S));
            S->read_len = L;
            S->seed_all = L < P->seed_len ? 0 : 1 + (L - P->seed_len) / P->seed_step;       /* lamsa_aln.c:252-253 */
            S->last_len = L - P->seed_len - (S->seed_all - 1) * P->seed_step;               /* :281 */
            S->seed_id = (int32_t*)malloc(sizeof(int32_t) * (size_t)(S->seed_all + 1));
            S->hit_off = (int32_t*)calloc((size_t)S->seed_all + 2, sizeof(int32_t));
            int hit_n = 0, cig_n = 0, cig_m = 0;
            for (int sd = 0; sd < S->seed_all; ++sd) {                                       /* lamsa_read_seq, :945-952 */
                if (!fgets(line, LINE_SIZE, mapf)) { fprintf(stderr, "[lo_io] GEM map result does not match the reads\n"); return -1; }
                size_t ll = strlen(line); if (ll && line[ll - 1] == '\n') line[ll - 1] = 0;
                int ct = 0; size_t k;
                for (k = 0; line[k]; ++k) if (line[k] == '\t') { if (ct == 3) break; ct++; }
                if (line[k + 1] == '-') continue;
                S->seed_id[S->seed_out] = sd + 1;
                lo_parse_hits(line + k + 1, &ix, P->per_aln_m, &S->hit, &hit_n, &S->cig, &cig_n, &cig_m);
                S->seed_out++;
                S->hit_off[S->seed_out] = hit_n;
            }
            free(x->bseq); x->bseq = (uint8_t*)malloc((size_t)L + 1);
            for (int i = 0; i < L; ++i) x->bseq[i] = t4[(unsigned char)x->rd.seq[i]];
            n_bases += L;
            ++n;
        }
        if (max_reads > 0 && n_reads + n >= max_reads) eof = 1;
        if (n == 0) break;
        struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
        work_t w; w.u = u; w.n = n; w.next = 0; w.ix = &ix; w.P = P; pthread_mutex_init(&w.mu, NULL);
        if (n_threads <= 1) worker(&w);
        else {
            pthread_t *th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
            for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, worker, &w);
            for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
            free(th);
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        secs += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (out) for (int i = 0; i < n; ++i) fwrite(u[i].sam.s, 1, u[i].sam.l, out);
        n_reads += n;
    }
    for (int i = 0; i < CH; ++i) { free(u[i].rd.name); free(u[i].rd.seq); free(u[i].rd.qual); free(u[i].bseq); free(u[i].sam.s);
                                   free(u[i].S.seed_id); free(u[i].S.hit_off); free(u[i].S.hit); free(u[i].S.cig); }
    free(u); free(line); free(rr.buf); gzclose(rr.f); fclose(mapf); lo_index_free(&ix);
    if (aln_seconds) *aln_seconds = secs;
    if (n_reads_out) *n_reads_out = n_reads;
    if (n_bases_out) *n_bases_out = n_bases;
    return 0;
}

/* ================================================================ SoA batches (parity-test plumbing) */
int lo_batch_load(const lo_index *ix, const char *reads, const lo_para *P, long max_reads, lo_batch *B)
{
    memset(B, 0, sizeof(*B));
    char fn[2048];
    snprintf(fn, sizeof fn, "%s.seed.gem.map", reads);
    FILE *mapf = fopen(fn, "r");
    if (!mapf) { fprintf(stderr, "[lo_io] cannot open %s\n", fn); return -1; }
    rdr_t rr; memset(&rr, 0, sizeof rr);
    rr.f = gzopen(reads, "r");
    if (!rr.f) { fclose(mapf); return -1; }
    rec_t rd; memset(&rd, 0, sizeof rd);
    char *line = (char*)malloc(LINE_SIZE);
    const uint8_t *t4 = nt4();
    size_t m_reads = 0, m_bases = 0, m_slots = 0;
    lo_hit *hits = NULL; int hit_n = 0; lo_cig *cig = NULL; int cig_n = 0, cig_m = 0;
    int64_t n_bases = 0, n_slots = 0;
    B->read_off = (int64_t*)calloc(1, sizeof(int64_t)); B->seed_off = (int64_t*)calloc(1, sizeof(int64_t)); B->hit_off = (int64_t*)calloc(1, sizeof(int64_t));
    while ((max_reads <= 0 || B->n_reads < max_reads) && read_record(&rr, &rd)) {
        const int L = rd.l, r = B->n_reads;
        if ((size_t)r + 2 > m_reads) {
            m_reads = m_reads ? m_reads * 2 : 256;
            B->read_off = (int64_t*)realloc(B->read_off, sizeof(int64_t) * (m_reads + 1)); B->seed_off = (int64_t*)realloc(B->seed_off, sizeof(int64_t) * (m_reads + 1));
            B->seed_all = (int32_t*)realloc(B->seed_all, sizeof(int32_t) * m_reads); B->last_len = (int32_t*)realloc(B->last_len, sizeof(int32_t) * m_reads);
        }
        if ((size_t)(n_bases + L) + 16 > m_bases) { m_bases = (size_t)(n_bases + L) * 2 + 1024; B->read_seq = (uint8_t*)realloc(B->read_seq, m_bases); }
        for (int i = 0; i < L; ++i) B->read_seq[n_bases + i] = t4[(unsigned char)rd.seq[i]];
        n_bases += L; B->read_off[r + 1] = n_bases;
        const int seed_all = L < P->seed_len ? 0 : 1 + (L - P->seed_len) / P->seed_step;
        B->seed_all[r] = seed_all; B->last_len[r] = L - P->seed_len - (seed_all - 1) * P->seed_step;
        for (int sd = 0; sd < seed_all; ++sd) {
            if (!fgets(line, LINE_SIZE, mapf)) { fprintf(stderr, "[lo_io] GEM map result does not match the reads\n"); return -1; }
            size_t ll = strlen(line); if (ll && line[ll - 1] == '\n') line[ll - 1] = 0;
            int ct = 0; size_t k;
            for (k = 0; line[k]; ++k) if (line[k] == '\t') { if (ct == 3) break; ct++; }
            if (line[k + 1] == '-') continue;
            if ((size_t)n_slots + 2 > m_slots) { m_slots = m_slots ? m_slots * 2 : 1024; B->seed_id = (int32_t*)realloc(B->seed_id, sizeof(int32_t) * m_slots); B->hit_off = (int64_t*)realloc(B->hit_off, sizeof(int64_t) * (m_slots + 1)); }
            B->seed_id[n_slots] = sd + 1;
            lo_parse_hits(line + k + 1, ix, P->per_aln_m, &hits, &hit_n, &cig, &cig_n, &cig_m);
            ++n_slots; B->hit_off[n_slots] = hit_n;
        }
        B->seed_off[r + 1] = n_slots;
        B->n_reads++;
    }
    B->n_slots = n_slots; B->n_hits = hit_n; B->n_cig = cig_n;
    B->h_pos = (int64_t*)malloc(sizeof(int64_t) * (size_t)(hit_n + 1)); B->h_chr = (int32_t*)malloc(sizeof(int32_t) * (size_t)(hit_n + 1));
    B->h_strand = (int8_t*)malloc((size_t)hit_n + 1); B->h_nm = (int16_t*)malloc(2 * (size_t)(hit_n + 1)); B->h_len_dif = (int16_t*)malloc(2 * (size_t)(hit_n + 1));
    B->h_cig_off = (int32_t*)malloc(sizeof(int32_t) * (size_t)(hit_n + 1)); B->h_cig_n = (uint8_t*)malloc((size_t)hit_n + 1);
    for (int k = 0; k < hit_n; ++k) {
        B->h_pos[k] = hits[k].offset; B->h_chr[k] = hits[k].chr; B->h_strand[k] = (int8_t)hits[k].strand; B->h_nm[k] = (int16_t)hits[k].NM;
        B->h_len_dif[k] = (int16_t)hits[k].len_dif; B->h_cig_off[k] = hits[k].cig_off; B->h_cig_n[k] = (uint8_t)hits[k].cig_n;
    }
    B->cig = cig ? cig : (lo_cig*)calloc(4, sizeof(lo_cig));
    if (!B->read_seq) B->read_seq = (uint8_t*)calloc(16, 1);
    if (!B->seed_id) B->seed_id = (int32_t*)calloc(4, sizeof(int32_t));
    if (!B->seed_all) { B->seed_all = (int32_t*)calloc(4, sizeof(int32_t)); B->last_len = (int32_t*)calloc(4, sizeof(int32_t)); }
    free(hits); free(line); free(rd.name); free(rd.seq); free(rd.qual); free(rr.buf); gzclose(rr.f); fclose(mapf);
    return 0;
}

void lo_batch_free(lo_batch *B)
{
    free(B->read_off); free(B->read_seq); free(B->seed_all); free(B->last_len); free(B->seed_off); free(B->seed_id); free(B->hit_off);
    free(B->h_pos); free(B->h_chr); free(B->h_strand); free(B->h_nm); free(B->h_len_dif); free(B->h_cig_off); free(B->h_cig_n); free(B->cig);
    memset(B, 0, sizeof(*B));
}

typedef struct { int32_t *w; int n, m; } wbuf;
static void wput(wbuf *b, int32_t v) { if (b->n == b->m) { b->m = b->m ? b->m * 2 : 256; b->w = (int32_t*)realloc(b->w, sizeof(int32_t) * (size_t)b->m); } b->w[b->n++] = v; }

typedef struct { const lo_batch *B; const lo_ref *R; const lo_para *P; wbuf *out; int32_t *status; volatile int next; pthread_mutex_t mu; } bwork_t;

static void batch_one(bwork_t *w, int r)
{
    const lo_batch *B = w->B;
    lo_seeds S; memset(&S, 0, sizeof S);
    const int64_t s0 = B->seed_off[r], s1 = B->seed_off[r + 1], hb = B->hit_off[s0];
    S.seed_all = B->seed_all[r]; S.last_len = B->last_len[r]; S.seed_out = (int)(s1 - s0); S.read_len = (int)(B->read_off[r + 1] - B->read_off[r]);
    S.seed_id = (int32_t*)malloc(sizeof(int32_t) * (size_t)(S.seed_out + 1)); S.hit_off = (int32_t*)malloc(sizeof(int32_t) * (size_t)(S.seed_out + 2));
    for (int i = 0; i < S.seed_out; ++i) S.seed_id[i] = B->seed_id[s0 + i];
    for (int i = 0; i <= S.seed_out; ++i) S.hit_off[i] = (int32_t)(B->hit_off[s0 + i] - hb);
    const int H = S.hit_off[S.seed_out];
    S.hit = (lo_hit*)calloc((size_t)H + 1, sizeof(lo_hit));
    for (int k = 0; k < H; ++k) {
        lo_hit *h = &S.hit[k];
        h->offset = B->h_pos[hb + k]; h->chr = B->h_chr[hb + k]; h->strand = B->h_strand[hb + k]; h->NM = B->h_nm[hb + k]; h->len_dif = B->h_len_dif[hb + k];
        h->cig_off = B->h_cig_off[hb + k]; h->cig_n = B->h_cig_n[hb + k];
    }
    S.cig = B->cig;
    lo_ares res3[3];
    for (int i = 0; i < 3; ++i) lo_ares_init(&res3[i], w->P->res_mul_max);
    lo_areg *a_reg = lo_areg_new(S.read_len);
    int rc = lo_align_read(&S, B->read_seq + B->read_off[r], w->R, w->P, res3, a_reg);
    wbuf *o = &w->out[r];
    w->status[r] = rc < 0 ? 2 : 0;
    wput(o, w->status[r]);
    if (rc < 0) { wput(o, 0); wput(o, 0); }
    else {
        wput(o, res3[0].l_n); wput(o, res3[1].l_n);
        for (int st = 0; st < 2; ++st)
            for (int i = 0; i < res3[st].l_n; ++i) {
                lo_lres *la = &res3[st].la[i];
                wput(o, la->line_score); wput(o, la->tol_score); wput(o, la->tol_NM); wput(o, la->cur_res_n + 1);
                for (int j = 0; j <= la->cur_res_n; ++j) {
                    lo_res *x = &la->res[j];
                    wput(o, (int32_t)(x->offset & 0xffffffffll)); wput(o, (int32_t)(x->offset >> 32)); wput(o, x->chr); wput(o, x->nstrand); wput(o, x->score); wput(o, x->NM); wput(o, x->cig.n);
                    for (int k = 0; k < x->cig.n; ++k) wput(o, x->cig.c[k]);
                }
            }
    }
    lo_areg_free(a_reg);
    for (int i = 0; i < 3; ++i) lo_ares_free(&res3[i]);
    free(S.seed_id); free(S.hit_off); free(S.hit);
}
static void *batch_worker(void *arg)
{
    bwork_t *w = (bwork_t*)arg;
    for (;;) {
        pthread_mutex_lock(&w->mu); int i = w->next++; pthread_mutex_unlock(&w->mu);
        if (i >= w->B->n_reads) break;
        batch_one(w, i);
    }
    return NULL;
}

int lo_batch_align_stream(const lo_batch *B, const lo_ref *R, const lo_para *P, int n_threads,
                          int32_t **stream, int64_t *n_words, int64_t *read_off, int32_t *read_len, int32_t *status)
{
    const int n = B->n_reads;
    wbuf *out = (wbuf*)calloc((size_t)n + 1, sizeof(wbuf));
    bwork_t w; w.B = B; w.R = R; w.P = P; w.out = out; w.status = status; w.next = 0; pthread_mutex_init(&w.mu, NULL);
    if (n_threads <= 1) batch_worker(&w);
    else {
        pthread_t *th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
        for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, batch_worker, &w);
        for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
        free(th);
    }
    int64_t tot = 0;
    for (int r = 0; r < n; ++r) { read_off[r] = tot; read_len[r] = out[r].n; tot += out[r].n; }
    *stream = (int32_t*)malloc(sizeof(int32_t) * (size_t)(tot + 4));
    for (int r = 0; r < n; ++r) { memcpy(*stream + read_off[r], out[r].w, sizeof(int32_t) * (size_t)out[r].n); free(out[r].w); }
    free(out);
    *n_words = tot;
    return 0;
}
