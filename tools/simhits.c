/* simhits.c -- synthetic benchmark input at scale, generated on the box that runs bench.py.
 *
 * GRCh37 and the GEM seeder are not available where the benchmark runs (no network; the GEM
 * binaries cannot travel), so bench.py needs:
 *   1. a GRCh37-sized stand-in reference WITH repeat families (without repeats the chaining
 *      cost -- 70% of the reference's time on a human-like genome -- would be hidden), as the
 *      2-bit .pac layout (base k at pac[k>>2] >> ((~k&3)<<1) & 3) plus the contig table;
 *   2. simulated long reads with per-base sub/ins/del errors and their true alignment;
 *   3. seed hits in the layout of lamsa_hp_batch: the true locus of every 50-bp seed whose
 *      alignment passes GEM's thresholds (edit distance, mismatches, indel length, matched
 *      bases), plus, for seeds inside a repeat copy, every other copy of the family that
 *      passes them, best 200 kept (`gem-mapper -d 200`).  CIGAR/NM/len_dif are exact for
 *      the simulated sequences (copies differ by substitutions only, so the indel structure
 *      of a seed is the same against every copy).
 * Everything is seeded and independent of the thread count.
 * Host-side tool code (C, gcc); not part of the product library.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <pthread.h>

typedef struct { uint64_t s; } rng_t;
static inline uint64_t rnd(rng_t *r) { uint64_t z = (r->s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static inline double rnd01(rng_t *r) { return (double)(rnd(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline uint64_t rndn(rng_t *r, uint64_t n) { return n ? rnd(r) % n : 0; }

typedef struct {
    int64_t l_pac; int n_seqs; int64_t *seq_off; int32_t *seq_len; uint8_t *pac;
    /* repeat copies, sorted by global start */
    int64_t n_copies; int64_t *c_start; int32_t *c_fam; int8_t *c_orient;
    int n_fam; int32_t *f_unit; int64_t *f_first;      /* copies of family f: f_list[f_first[f] .. f_first[f+1]) (copy indices) */
    int64_t *f_list;
} sim_ref;

#define GETB(pac, k) (((pac)[(k) >> 2] >> ((~(k) & 3) << 1)) & 3)
static inline void setb(uint8_t *pac, int64_t k, int b) { int sh = (int)((~k & 3) << 1); pac[k >> 2] = (uint8_t)((pac[k >> 2] & ~(3 << sh)) | (b << sh)); }

typedef struct { sim_ref *R; uint64_t seed; const uint8_t **unit; const double *div; int tid, nt; } cp_job;

static int cmp_i64(const void *a, const void *b) { int64_t x = *(const int64_t*)a, y = *(const int64_t*)b; return x < y ? -1 : x > y; }

static void *copy_worker(void *arg)
{   /* write the (mutated, possibly reverse-complemented) unit of every copy; copies never overlap */
    cp_job *j = (cp_job*)arg; sim_ref *R = j->R;
    for (int64_t c = j->tid; c < R->n_copies; c += j->nt) {
        /* a copy may share a pac byte with its neighbour: only touch whole bytes owned by this copy, the
         * ragged edges are written by the single-threaded pass below */
        rng_t g = { j->seed ^ (0x51ed27ull * (uint64_t)(c + 1)) };
        const int f = R->c_fam[c], ul = R->f_unit[f];
        const uint8_t *u = j->unit[f];
        const int64_t s = R->c_start[c];
        const double dv = j->div[f];
        for (int i = 0; i < ul; ++i) {
            int b = R->c_orient[c] > 0 ? u[i] : 3 - u[ul - 1 - i];
            if (rnd01(&g) < dv) b = (b + 1 + (int)rndn(&g, 3)) & 3;
            const int64_t k = s + i;
            if ((k >> 2) == (s >> 2) || (k >> 2) == ((s + ul - 1) >> 2)) continue;     /* edge bytes later */
            setb(R->pac, k, b);
        }
    }
    return NULL;
}

sim_ref *sim_ref_new(uint64_t seed, int n_seqs, const int64_t *lens, int n_shapes, const int32_t *unit, const int32_t *copies,
                     const int32_t *count, const double *div, int n_threads)
{
    sim_ref *R = (sim_ref*)calloc(1, sizeof(sim_ref));
    R->n_seqs = n_seqs; R->seq_off = (int64_t*)calloc((size_t)n_seqs + 1, 8); R->seq_len = (int32_t*)calloc((size_t)n_seqs + 1, 4);
    for (int i = 0; i < n_seqs; ++i) { R->seq_off[i] = R->l_pac; R->seq_len[i] = (int32_t)lens[i]; R->l_pac += lens[i]; }
    const size_t nb = (size_t)(R->l_pac / 4 + 1);
    R->pac = (uint8_t*)malloc(nb + 16);
    rng_t g = { seed };
    { uint64_t *w = (uint64_t*)R->pac; for (size_t i = 0; i < (nb + 15) / 8; ++i) w[i] = rnd(&g); }
    /* families */
    int n_fam = 0; int64_t n_copies = 0;
    for (int s = 0; s < n_shapes; ++s) { n_fam += count[s]; n_copies += (int64_t)count[s] * copies[s]; }
    R->n_fam = n_fam; R->f_unit = (int32_t*)calloc((size_t)n_fam + 1, 4); R->f_first = (int64_t*)calloc((size_t)n_fam + 2, 8);
    double *fdiv = (double*)calloc((size_t)n_fam + 1, sizeof(double));
    const uint8_t **units = (const uint8_t**)calloc((size_t)n_fam + 1, sizeof(uint8_t*));
    int32_t *fam_of = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_copies + 1));
    { int f = 0; int64_t c = 0;
      for (int s = 0; s < n_shapes; ++s) for (int k = 0; k < count[s]; ++k, ++f) {
          R->f_unit[f] = unit[s]; fdiv[f] = div[s];
          uint8_t *u = (uint8_t*)malloc((size_t)unit[s]); for (int i = 0; i < unit[s]; ++i) u[i] = (uint8_t)rndn(&g, 4); units[f] = u;
          for (int q = 0; q < copies[s]; ++q) fam_of[c++] = f;
      } }
    /* shuffle the family labels over the copies, draw sorted starts, push overlapping copies forward */
    for (int64_t i = n_copies - 1; i > 0; --i) { int64_t j = (int64_t)rndn(&g, (uint64_t)i + 1); int32_t t = fam_of[i]; fam_of[i] = fam_of[j]; fam_of[j] = t; }
    int64_t *st = (int64_t*)malloc(8 * (size_t)(n_copies + 1));
    for (int64_t i = 0; i < n_copies; ++i) st[i] = (int64_t)rndn(&g, (uint64_t)R->l_pac);
    qsort(st, (size_t)n_copies, 8, cmp_i64);
    R->c_start = (int64_t*)malloc(8 * (size_t)(n_copies + 1)); R->c_fam = (int32_t*)malloc(4 * (size_t)(n_copies + 1)); R->c_orient = (int8_t*)malloc((size_t)n_copies + 1);
    int64_t kept = 0, prev_end = 0; int ci = 0;
    for (int64_t i = 0; i < n_copies; ++i) {
        int64_t s = st[i] < prev_end + 8 ? prev_end + 8 : st[i];
        const int ul = R->f_unit[fam_of[i]];
        while (ci < n_seqs && s >= R->seq_off[ci] + R->seq_len[ci]) ++ci;
        if (ci >= n_seqs) break;
        if (s + ul > R->seq_off[ci] + R->seq_len[ci]) continue;          /* would cross a contig end: dropped */
        R->c_start[kept] = s; R->c_fam[kept] = fam_of[i]; R->c_orient[kept] = (rnd(&g) & 1) ? 1 : -1;
        prev_end = s + ul; ++kept;
    }
    R->n_copies = kept;
    free(st); free(fam_of);
    /* per-family copy lists */
    for (int64_t c = 0; c < kept; ++c) R->f_first[R->c_fam[c] + 1]++;
    for (int f = 0; f < n_fam; ++f) R->f_first[f + 1] += R->f_first[f];
    R->f_list = (int64_t*)malloc(8 * (size_t)(kept + 1));
    { int64_t *fill = (int64_t*)calloc((size_t)n_fam + 1, 8);
      for (int64_t c = 0; c < kept; ++c) { int f = R->c_fam[c]; R->f_list[R->f_first[f] + fill[f]++] = c; }
      free(fill); }
    /* write the copies (threads own disjoint byte ranges; the two edge bytes of each copy are done serially) */
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads); cp_job *jobs = (cp_job*)malloc(sizeof(cp_job) * (size_t)n_threads);
    for (int t = 0; t < n_threads; ++t) { jobs[t].R = R; jobs[t].seed = seed; jobs[t].unit = units; jobs[t].div = fdiv; jobs[t].tid = t; jobs[t].nt = n_threads; pthread_create(&th[t], NULL, copy_worker, &jobs[t]); }
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    for (int64_t c = 0; c < kept; ++c) {
        rng_t g2 = { seed ^ (0x51ed27ull * (uint64_t)(c + 1)) };
        const int f = R->c_fam[c], ul = R->f_unit[f]; const uint8_t *u = units[f]; const int64_t s = R->c_start[c];
        for (int i = 0; i < ul; ++i) {
            int b = R->c_orient[c] > 0 ? u[i] : 3 - u[ul - 1 - i];
            if (rnd01(&g2) < fdiv[f]) b = (b + 1 + (int)rndn(&g2, 3)) & 3;
            const int64_t k = s + i;
            if ((k >> 2) == (s >> 2) || (k >> 2) == ((s + ul - 1) >> 2)) setb(R->pac, k, b);
        }
    }
    for (int f = 0; f < n_fam; ++f) free((void*)units[f]);
    free(units); free(fdiv); free(th); free(jobs);
    return R;
}

void sim_ref_free(sim_ref *R)
{
    if (!R) return;
    free(R->seq_off); free(R->seq_len); free(R->pac); free(R->c_start); free(R->c_fam); free(R->c_orient); free(R->f_unit); free(R->f_first); free(R->f_list); free(R);
}
int64_t sim_ref_l_pac(const sim_ref *R) { return R->l_pac; }
const uint8_t *sim_ref_pac(const sim_ref *R) { return R->pac; }
const int64_t *sim_ref_seq_off(const sim_ref *R) { return R->seq_off; }
const int32_t *sim_ref_seq_len(const sim_ref *R) { return R->seq_len; }
int64_t sim_ref_n_copies(const sim_ref *R) { return R->n_copies; }

/* ---------------------------------------------------------------- reads + hits */
typedef struct {
    int n_reads; int length; double sub, ins, del, sv_frac;
    int seed_len, seed_step, max_edit, max_mis, min_match, max_indel, per_loci;
} sim_cfg;

typedef struct { int64_t pos; int32_t chr; int8_t strand; int16_t nm, len_dif; int32_t cig_off; uint8_t cig_n; } hit_t;

typedef struct {          /* growable per-thread output */
    uint8_t *seq; size_t seq_n, seq_m;
    int32_t *seed_id; int64_t *hit_cnt; size_t slot_n, slot_m;
    hit_t *hit; size_t hit_n, hit_m;
    int32_t *cig; size_t cig_n, cig_m;
    int64_t *r_len, *r_slots; int32_t *r_seed_all, *r_last; int32_t *t_chr; int64_t *t_pos; int8_t *t_strand;
} tout;

#define GROW(p, n, m, T) do { if ((n) + 1 > (m)) { (m) = (m) ? (m) * 2 : 1024; (p) = (T*)realloc((p), sizeof(T) * (m)); } } while (0)

typedef struct { const sim_ref *R; const sim_cfg *C; uint64_t seed; int r0, r1; tout o; } rd_job;

static int64_t find_copy(const sim_ref *R, int64_t g0, int64_t g1)
{   /* index of the copy that contains [g0, g1], or -1 */
    int64_t lo = 0, hi = R->n_copies - 1, best = -1;
    while (lo <= hi) { int64_t m = (lo + hi) >> 1; if (R->c_start[m] <= g0) { best = m; lo = m + 1; } else hi = m - 1; }
    if (best < 0) return -1;
    return (g1 < R->c_start[best] + R->f_unit[R->c_fam[best]]) ? best : -1;
}

typedef struct { int op; int len; } op_t;

static void *read_worker(void *arg)
{
    rd_job *J = (rd_job*)arg; const sim_ref *R = J->R; const sim_cfg *C = J->C; tout *o = &J->o;
    const int L0 = C->length;
    const int nr = J->r1 - J->r0;
    o->r_len = (int64_t*)calloc((size_t)nr + 1, 8); o->r_slots = (int64_t*)calloc((size_t)nr + 1, 8); o->r_seed_all = (int32_t*)calloc((size_t)nr + 1, 4); o->r_last = (int32_t*)calloc((size_t)nr + 1, 4);
    o->t_chr = (int32_t*)calloc((size_t)nr + 1, 4); o->t_pos = (int64_t*)calloc((size_t)nr + 1, 8); o->t_strand = (int8_t*)calloc((size_t)nr + 1, 1);
    uint8_t *fr = (uint8_t*)malloc((size_t)L0 * 2 + 64); int64_t *rmap = (int64_t*)malloc(8 * ((size_t)L0 * 2 + 64));
    hit_t *cand = (hit_t*)malloc(sizeof(hit_t) * 70000);
    op_t ops[256];
    for (int r = J->r0; r < J->r1; ++r) {
        rng_t g = { J->seed * 0x2545f4914f6cdd1dull + (uint64_t)r * 0x9e3779b97f4a7c15ull + 12345 };
        /* locus */
        int ci; int64_t pos;
        /* rejection sampling over the whole reference; sim_reads_new has checked that some contig is long enough, and after a million
         * misses (a reference that is nearly all short contigs) the read starts at the beginning of the first contig that is */
        const int64_t need = (int64_t)L0 * 2 + 16 + (C->sv_frac > 0 ? 10016 : 0);
        for (int tries = 0;; ++tries) {
            int64_t gp = (int64_t)rndn(&g, (uint64_t)R->l_pac); ci = 0; while (ci + 1 < R->n_seqs && gp >= R->seq_off[ci + 1]) ++ci;
            pos = gp - R->seq_off[ci]; if (pos + need < R->seq_len[ci]) break;
            if (tries >= 1000000) { ci = 0; while (ci + 1 < R->n_seqs && need >= R->seq_len[ci]) ++ci; pos = 0; break; }
        }
        const int strand = (rnd(&g) & 1) ? 1 : -1;
        /* one structural variant at the middle of the read: a deletion of U[1k,10k] reference
         * bases or a novel insertion of U[1k,5k] random bases.  No random number is drawn when sv_frac is 0, so the other
         * workloads' reads stay what they were. */
        int sv_kind = 0, sv_len = 0;
        if (C->sv_frac > 0 && rnd01(&g) < C->sv_frac) { sv_kind = 1 + (int)(rnd(&g) & 1); sv_len = sv_kind == 1 ? 1000 + (int)rndn(&g, 9001) : 1000 + (int)rndn(&g, 4001); }
        /* walk the reference, apply errors: forward-fragment read fr[] with ref coordinate per base (-1: inserted) */
        int L = 0; int64_t gk = R->seq_off[ci] + pos;
        while (L < L0) {
            if (sv_kind && L >= L0 / 2) {
                if (sv_kind == 1) gk += sv_len;
                else for (int i = 0; i < sv_len && L < L0; ++i) { fr[L] = (uint8_t)rndn(&g, 4); rmap[L] = -1; ++L; }
                sv_kind = 0;
                continue;
            }
            const double x = rnd01(&g);
            if (x < C->del) { ++gk; continue; }
            int b = GETB(R->pac, gk);
            if (x < C->del + C->sub) b = (b + 1 + (int)rndn(&g, 3)) & 3;
            fr[L] = (uint8_t)b; rmap[L] = gk; ++L; ++gk;
            while (L < L0 && rnd01(&g) < C->ins) { fr[L] = (uint8_t)rndn(&g, 4); rmap[L] = -1; ++L; }
        }
        const int li = r - J->r0;
        o->r_len[li] = L; o->t_chr[li] = ci + 1; o->t_pos[li] = pos + 1; o->t_strand[li] = (int8_t)strand;
        if (o->seq_n + (size_t)L + 16 > o->seq_m) { o->seq_m = (o->seq_n + (size_t)L) * 2 + 4096; o->seq = (uint8_t*)realloc(o->seq, o->seq_m); }
        if (strand > 0) memcpy(o->seq + o->seq_n, fr, (size_t)L); else for (int i = 0; i < L; ++i) o->seq[o->seq_n + i] = (uint8_t)(3 - fr[L - 1 - i]);
        o->seq_n += (size_t)L;
        const int seed_all = L < C->seed_len ? 0 : 1 + (L - C->seed_len) / C->seed_step;
        o->r_seed_all[li] = seed_all; o->r_last[li] = L - C->seed_len - (seed_all - 1) * C->seed_step;
        int64_t n_slots = 0;
        for (int k = 0; k < seed_all; ++k) {
            /* seed k covers read [k*step, k*step+seed_len) = forward-fragment [a, b) */
            int a = strand > 0 ? k * C->seed_step : L - k * C->seed_step - C->seed_len, b = a + C->seed_len;
            if (rmap[a] < 0 || rmap[b - 1] < 0) continue;                 /* starts / ends in an insertion: no hit */
            /* op list in reference-forward order */
            int n_ops = 0, nI = 0, nD = 0, nM = 0, bad = 0; int64_t prev = -1;
            for (int i = a; i < b && !bad; ++i) {
                if (rmap[i] < 0) { if (n_ops && ops[n_ops - 1].op == 1) ops[n_ops - 1].len++; else { ops[n_ops].op = 1; ops[n_ops++].len = 1; } ++nI; }
                else {
                    if (prev >= 0 && rmap[i] > prev + 1) { int d = (int)(rmap[i] - prev - 1); if (n_ops && ops[n_ops - 1].op == 1) bad = 1; ops[n_ops].op = 2; ops[n_ops++].len = d; nD += d; }
                    if (n_ops && ops[n_ops - 1].op == 0) ops[n_ops - 1].len++; else { ops[n_ops].op = 0; ops[n_ops++].len = 1; }
                    ++nM; prev = rmap[i];
                }
                if (n_ops > 250) bad = 1;
            }
            for (int q = 0; q < n_ops && !bad; ++q) if (ops[q].op && ops[q].len > C->max_indel) bad = 1;
            for (int q = 1; q < n_ops && !bad; ++q) if (ops[q].op && ops[q - 1].op) bad = 1;   /* adjacent I/D: ambiguous, skip */
            if (bad || nI + nD > C->max_edit) continue;
            const int64_t g0 = rmap[a], g1 = rmap[b - 1];
            const int64_t cp = find_copy(R, g0, g1);
            int n_cand = 0;
            const int64_t fb = cp >= 0 ? R->f_first[R->c_fam[cp]] : 0, fe = cp >= 0 ? R->f_first[R->c_fam[cp] + 1] : 1;
            for (int64_t q = fb; q < fe; ++q) {
                /* candidate locus: the true one (cp < 0) or copy c2 of the family */
                int same = 1; int64_t sc = 0, s2 = 0; int ul = 0;
                if (cp >= 0) { const int64_t c2 = R->f_list[q]; same = R->c_orient[c2] == R->c_orient[cp]; sc = R->c_start[cp]; s2 = R->c_start[c2]; ul = R->f_unit[R->c_fam[cp]]; }
                /* count mismatches over the M bases */
                int mm = 0;
                for (int i = a; i < b; ++i) {
                    if (rmap[i] < 0) continue;
                    int64_t gp2 = cp < 0 ? rmap[i] : (same ? s2 + (rmap[i] - sc) : s2 + (ul - 1 - (rmap[i] - sc)));
                    int rb = same ? fr[i] : 3 - fr[i];
                    if (GETB(R->pac, gp2) != rb) { if (++mm > C->max_mis) break; }
                }
                if (mm > C->max_mis || mm + nI + nD > C->max_edit || nM - mm < C->min_match) continue;
                int64_t left = cp < 0 ? g0 : (same ? s2 + (g0 - sc) : s2 + (ul - 1 - (g1 - sc)));
                int cj = 0; while (cj + 1 < R->n_seqs && left >= R->seq_off[cj + 1]) ++cj;
                hit_t *h = &cand[n_cand];
                h->pos = left - R->seq_off[cj] + 1; h->chr = cj + 1; h->strand = (int8_t)(same ? strand : -strand);
                h->nm = (int16_t)(mm + nI + nD); h->len_dif = (int16_t)(nD - nI); h->cig_n = (uint8_t)n_ops; h->cig_off = n_cand;    /* cig_off: candidate id for now */
                ++n_cand;
                if (n_cand >= 69999) break;
            }
            if (n_cand == 0) continue;
            /* best per_loci by NM (stable on locus order) */
            if (n_cand > C->per_loci) {
                int cnt[64] = {0};
                for (int i = 0; i < n_cand; ++i) cnt[cand[i].nm < 63 ? cand[i].nm : 63]++;
                int cut = 0, acc = 0; while (cut < 64 && acc + cnt[cut] <= C->per_loci) { acc += cnt[cut]; ++cut; }
                int w = 0, at_cut = 0;
                for (int i = 0; i < n_cand; ++i) { int v = cand[i].nm < 63 ? cand[i].nm : 63; if (v < cut || (v == cut && acc + at_cut < C->per_loci && ++at_cut)) cand[w++] = cand[i]; }
                n_cand = w;
            }
            if (o->slot_n + 1 > o->slot_m) { o->slot_m = o->slot_m ? o->slot_m * 2 : 1024; o->seed_id = (int32_t*)realloc(o->seed_id, 4 * o->slot_m); o->hit_cnt = (int64_t*)realloc(o->hit_cnt, 8 * o->slot_m); }
            o->seed_id[o->slot_n] = k + 1; o->hit_cnt[o->slot_n] = n_cand; ++o->slot_n; ++n_slots;
            for (int i = 0; i < n_cand; ++i) {
                GROW(o->hit, o->hit_n, o->hit_m, hit_t);
                hit_t h = cand[i];
                const int same = (h.strand == strand);
                if (o->cig_n + (size_t)n_ops + 1 > o->cig_m) { o->cig_m = (o->cig_n + (size_t)n_ops) * 2 + 4096; o->cig = (int32_t*)realloc(o->cig, 4 * o->cig_m); }
                h.cig_off = (int32_t)o->cig_n;
                for (int t = 0; t < n_ops; ++t) { const op_t *p = same ? &ops[t] : &ops[n_ops - 1 - t]; o->cig[o->cig_n++] = (p->len << 4) | p->op; }
                o->hit[o->hit_n++] = h;
            }
        }
        o->r_slots[li] = n_slots;
    }
    free(fr); free(rmap); free(cand);
    return NULL;
}

typedef struct {
    int32_t n_reads; int64_t n_slots, n_hits, n_cig;
    int64_t *read_off; uint8_t *read_seq; int32_t *seed_all, *last_len; int64_t *seed_off; int32_t *seed_id; int64_t *hit_off;
    int64_t *h_pos; int32_t *h_chr; int8_t *h_strand; int16_t *h_nm, *h_len_dif; int32_t *h_cig_off; uint8_t *h_cig_n; int32_t *cig;
    int32_t *t_chr; int64_t *t_pos; int8_t *t_strand;
} sim_batch;

sim_batch *sim_reads_new(const sim_ref *R, uint64_t seed, const sim_cfg *C, int n_threads)
{
    /* a read of L0 bases needs a contig with room for its locus (twice its length, plus the largest structural variant): NULL when the
     * reference has none -- the caller raises -- instead of a worker that draws loci for ever */
    { const int64_t need = (int64_t)C->length * 2 + 16 + (C->sv_frac > 0 ? 10016 : 0); int ok = 0;
      for (int c = 0; c < R->n_seqs; ++c) if (need < R->seq_len[c]) ok = 1;
      if (!ok) return NULL; }
    if (n_threads < 1) n_threads = 1;
    if (n_threads > C->n_reads) n_threads = C->n_reads > 0 ? C->n_reads : 1;
    rd_job *jobs = (rd_job*)calloc((size_t)n_threads, sizeof(rd_job)); pthread_t *th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int t = 0; t < n_threads; ++t) {
        jobs[t].R = R; jobs[t].C = C; jobs[t].seed = seed;
        jobs[t].r0 = (int)((int64_t)C->n_reads * t / n_threads); jobs[t].r1 = (int)((int64_t)C->n_reads * (t + 1) / n_threads);
        pthread_create(&th[t], NULL, read_worker, &jobs[t]);
    }
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    sim_batch *B = (sim_batch*)calloc(1, sizeof(sim_batch));
    const int n = C->n_reads;
    size_t tb = 0, ts = 0, thh = 0, tc = 0;
    for (int t = 0; t < n_threads; ++t) { tb += jobs[t].o.seq_n; ts += jobs[t].o.slot_n; thh += jobs[t].o.hit_n; tc += jobs[t].o.cig_n; }
    B->n_reads = n; B->n_slots = (int64_t)ts; B->n_hits = (int64_t)thh; B->n_cig = (int64_t)tc;
    B->read_off = (int64_t*)calloc((size_t)n + 1, 8); B->read_seq = (uint8_t*)malloc(tb + 16); B->seed_all = (int32_t*)calloc((size_t)n + 1, 4); B->last_len = (int32_t*)calloc((size_t)n + 1, 4);
    B->seed_off = (int64_t*)calloc((size_t)n + 1, 8); B->seed_id = (int32_t*)malloc(4 * (ts + 1)); B->hit_off = (int64_t*)calloc(ts + 1, 8);
    B->h_pos = (int64_t*)malloc(8 * (thh + 1)); B->h_chr = (int32_t*)malloc(4 * (thh + 1)); B->h_strand = (int8_t*)malloc(thh + 1); B->h_nm = (int16_t*)malloc(2 * (thh + 1));
    B->h_len_dif = (int16_t*)malloc(2 * (thh + 1)); B->h_cig_off = (int32_t*)malloc(4 * (thh + 1)); B->h_cig_n = (uint8_t*)malloc(thh + 1); B->cig = (int32_t*)malloc(4 * (tc + 4));
    B->t_chr = (int32_t*)calloc((size_t)n + 1, 4); B->t_pos = (int64_t*)calloc((size_t)n + 1, 8); B->t_strand = (int8_t*)calloc((size_t)n + 1, 1);
    size_t ob = 0, os = 0, oh = 0, oc = 0;
    for (int t = 0; t < n_threads; ++t) {
        tout *o = &jobs[t].o;
        memcpy(B->read_seq + ob, o->seq, o->seq_n);
        size_t lb = 0, ls = 0;
        for (int r = jobs[t].r0; r < jobs[t].r1; ++r) {
            const int li = r - jobs[t].r0;
            B->read_off[r] = (int64_t)(ob + lb); lb += (size_t)o->r_len[li];
            B->seed_off[r] = (int64_t)(os + ls); ls += (size_t)o->r_slots[li];
            B->seed_all[r] = o->r_seed_all[li]; B->last_len[r] = o->r_last[li]; B->t_chr[r] = o->t_chr[li]; B->t_pos[r] = o->t_pos[li]; B->t_strand[r] = o->t_strand[li];
        }
        size_t hh = oh;
        for (size_t s = 0; s < o->slot_n; ++s) { B->seed_id[os + s] = o->seed_id[s]; B->hit_off[os + s] = (int64_t)hh; hh += (size_t)o->hit_cnt[s]; }
        for (size_t k = 0; k < o->hit_n; ++k) {
            const hit_t *h = &o->hit[k];
            B->h_pos[oh + k] = h->pos; B->h_chr[oh + k] = h->chr; B->h_strand[oh + k] = h->strand; B->h_nm[oh + k] = h->nm; B->h_len_dif[oh + k] = h->len_dif;
            B->h_cig_off[oh + k] = (int32_t)(oc + (size_t)h->cig_off); B->h_cig_n[oh + k] = h->cig_n;
        }
        memcpy(B->cig + oc, o->cig, 4 * o->cig_n);
        ob += o->seq_n; os += o->slot_n; oh += o->hit_n; oc += o->cig_n;
        free(o->seq); free(o->seed_id); free(o->hit_cnt); free(o->hit); free(o->cig); free(o->r_len); free(o->r_slots); free(o->r_seed_all); free(o->r_last); free(o->t_chr); free(o->t_pos); free(o->t_strand);
    }
    B->read_off[n] = (int64_t)ob; B->seed_off[n] = (int64_t)os; B->hit_off[ts] = (int64_t)oh;
    free(jobs); free(th);
    return B;
}

void sim_batch_free(sim_batch *B)
{
    if (!B) return;
    free(B->read_off); free(B->read_seq); free(B->seed_all); free(B->last_len); free(B->seed_off); free(B->seed_id); free(B->hit_off);
    free(B->h_pos); free(B->h_chr); free(B->h_strand); free(B->h_nm); free(B->h_len_dif); free(B->h_cig_off); free(B->h_cig_n); free(B->cig);
    free(B->t_chr); free(B->t_pos); free(B->t_strand); free(B);
}
