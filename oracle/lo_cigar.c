/* lo_cigar.c -- CIGAR vector helpers (oracle; see lo.h header note).
 * Behaviour follows the inline helpers in src/frag_check.h:124-198 and the
 * length counters in src/frag_check.c:167-215. */
#include <stdlib.h>
#include <stdio.h>
#include "lo.h"

void lo_cigv_init(lo_cigv *v) { v->c = 0; v->n = v->m = 0; }
void lo_cigv_free(lo_cigv *v) { free(v->c); v->c = 0; v->n = v->m = 0; }
void lo_cigv_clear(lo_cigv *v) { v->n = 0; }

static void grow(lo_cigv *v, int need)
{
    if (need <= v->m) return;
    int m = v->m ? v->m : 16;
    while (m < need) m <<= 1;
    v->c = (lo_cig*)realloc(v->c, (size_t)m * sizeof(lo_cig));
    v->m = m;
}

/* same-op merge, zero length allowed (frag_check.h:136-151) */
void lo_cig_push0(lo_cigv *v, lo_cig w)
{
    if (v->n > 0 && (v->c[v->n-1] & 0xf) == (w & 0xf)) { v->c[v->n-1] += (w >> 4) << 4; return; }
    grow(v, v->n + 1);
    v->c[v->n++] = w;
}
/* zero length dropped (frag_check.h:153-156) */
void lo_cig_push1(lo_cigv *v, lo_cig w) { if ((w >> 4) == 0) return; lo_cig_push0(v, w); }

/* concatenation: first word merges on same op, and I+S / S+I fuse into S (frag_check.h:158-184) */
void lo_cig_pushv(lo_cigv *v, const lo_cig *c, int n)
{
    int j = 0;
    if (n == 0) return;
    if (v->n > 0) {
        lo_cig last = v->c[v->n-1];
        if ((last & 0xf) == (c[0] & 0xf)) { v->c[v->n-1] += (c[0] >> 4) << 4; j = 1; }
        else if (((last & 0xf) == LO_I && (c[0] & 0xf) == LO_S) || ((last & 0xf) == LO_S && (c[0] & 0xf) == LO_I)) {
            v->c[v->n-1] = (((last >> 4) + (c[0] >> 4)) << 4) | LO_S; j = 1;
        }
    }
    grow(v, v->n + n);
    for (; j < n; ++j) v->c[v->n++] = c[j];
}

void lo_cig_invert(lo_cig *c, int n)
{
    for (int i = 0; i < n / 2; ++i) { lo_cig t = c[i]; c[i] = c[n-1-i]; c[n-1-i] = t; }
}

static void cig_err(const char *who) { fprintf(stderr, "[lo] %s: unexpected CIGAR op\n", who); exit(1); }

int lo_cig_readlen(const lo_cig *c, int n)
{
    int l = 0;
    for (int i = 0; i < n; ++i) { int op = c[i] & 0xf; if (op == LO_M || op == LO_I || op == LO_S) l += c[i] >> 4; else if (op != LO_D && op != LO_H) cig_err("readlen"); }
    return l;
}
int lo_cig_reflen(const lo_cig *c, int n)
{
    int l = 0;
    for (int i = 0; i < n; ++i) { int op = c[i] & 0xf; if (op == LO_M || op == LO_D || op == LO_H) l += c[i] >> 4; else if (op != LO_I && op != LO_S) cig_err("reflen"); }
    return l;
}
int lo_cig_solid_readlen(const lo_cig *c, int n)
{
    int l = 0;
    for (int i = 0; i < n; ++i) { int op = c[i] & 0xf; if (op == LO_M || op == LO_I) l += c[i] >> 4; else if (op != LO_D && op != LO_H && op != LO_S) cig_err("solid_readlen"); }
    return l;
}
