/* lo_align.c -- per-read driver of the oracle (see lo.h): stages (2),(3),(2'),(3') of the
 * reference's worker lamsa_main_aln (src/lamsa_aln.c:825-891).  Stage (4) (BWT rescue,
 * src/bwt_aln.c) is not restated yet: res3[2] stays empty, which equals the reference run
 * with `-R 0` (bwt_max_len = 0 makes get_remain_reg return nothing, src/bwt_aln.c:401). */
#include <stdlib.h>
#include <string.h>
#include "lo_read.h"

int lo_align_read(lo_seeds *S, const uint8_t *read, const lo_ref *R, const lo_para *P, lo_ares *res3, lo_areg *a_reg)
{
    const int H = S->hit_off[S->seed_out];
    lo_node *nodes = (lo_node*)calloc((size_t)H + 1, sizeof(lo_node));
    uint8_t *rc_read = NULL;
    lo_fline *lines = NULL;
    int rc = 0;
    for (int i = 0; i < 3; ++i) lo_ares_reset(&res3[i], S->read_len);
    int line_n = lo_chain_first(S, P, nodes, &lines);                       /* :857 */
    if (line_n > 0) {
        rc = lo_frag_check(S, lines, line_n, &res3[0], R, read, &rc_read, P); /* :863 */
        if (rc == 0) lo_get_reg(&res3[0], a_reg);
    }
    lo_flines_free(lines, line_n); lines = NULL;
    if (rc == 0) {
        line_n = lo_chain_remain(a_reg, S, P, nodes, &lines);                /* :867 */
        if (line_n > 0) {
            rc = lo_frag_check(S, lines, line_n, &res3[1], R, read, &rc_read, P);
            if (rc == 0) lo_get_reg(&res3[1], a_reg);
        }
        lo_flines_free(lines, line_n);
    }
    for (int i = 0; i < H; ++i) free(nodes[i].son);
    free(nodes); free(rc_read);
    return rc;
}
