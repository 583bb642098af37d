/* lo_para.c -- parameter defaults and presets (oracle; see lo.h header note).
 * Follows init_aln_para (src/lamsa_aln.c:1281-1329), lamsa_set_aln_mode (:1342-1420),
 * lamsa_fill_mat (:1331-1340) and the constants of src/lamsa_aln.h:15-90,
 * src/split_mapping.h:49-56. */
#include <math.h>
#include "lo.h"

void lo_para_init(lo_para *P)
{
    P->seed_len = P->seed_step = -1; P->seed_inv = 0;
    P->per_aln_m = 200; P->first_loci_thd = 2;
    P->SV_len_thd = 10000; P->ske_max = 10; P->ovlp_rat = (float)0.7;
    P->split_len = 100; P->split_pen = 10; P->res_mul_max = 10;
    P->hash_len = P->hash_step = 0; P->hash_key_len = 2; P->hash_size = 16;
    P->bwt_seed_len = 19; P->bwt_max_len = 300; P->bwt_min_len = 0;
    P->match = P->mis = -1;
    P->ins_gapo = P->del_gapo = P->ins_gape = P->del_gape = -1;
    P->ins_ext_o = P->del_ext_o = P->ins_ext_e = P->del_ext_e = -1;
    P->id_rate = -1;
    P->read_type = 0; P->band_w = -1; P->end_bonus = -1; P->zdrop = 100;
    P->aln_mode = 0; P->supp_soft = 0; P->comm = 0;
    P->match_dis = 0; P->mismatch_thd = 0;
}

#define DFL(x, v) do { if ((x) < 0) (x) = (v); } while (0)
void lo_para_finish(lo_para *P)
{
    int i, j, k;
    if (P->read_type == 0) {          /* src/lamsa_aln.c:1344-1367 */
        DFL(P->seed_step, 100); DFL(P->seed_len, 50);
        P->bwt_min_len = P->bwt_seed_len; P->hash_len = 10; P->hash_step = 10;
        DFL(P->match, 1); DFL(P->mis, 3);
        DFL(P->ins_gapo, 5); DFL(P->ins_gape, 2); DFL(P->del_gapo, 5); DFL(P->del_gape, 2);
        DFL(P->ins_ext_o, 5); DFL(P->ins_ext_e, 2); DFL(P->del_ext_o, 5); DFL(P->del_ext_e, 2);
        DFL(P->id_rate, (float)0.04); DFL(P->band_w, 10); DFL(P->end_bonus, 5);
        P->match_dis = 5; P->mismatch_thd = 10;
    } else if (P->read_type == 1) {   /* :1368-1393 */
        DFL(P->seed_step, 25); DFL(P->seed_len, 50);
        P->hash_len = 8; P->hash_step = 4; P->bwt_min_len = 50;
        DFL(P->match, 1); DFL(P->mis, 1);
        DFL(P->ins_gapo, 1); DFL(P->ins_gape, 1); DFL(P->del_gapo, 1); DFL(P->del_gape, 1);
        DFL(P->ins_ext_o, 2); DFL(P->ins_ext_e, 1); DFL(P->del_ext_o, 2); DFL(P->del_ext_e, 1);
        DFL(P->id_rate, (float)0.3); DFL(P->band_w, 200); DFL(P->end_bonus, 0);
        P->match_dis = (int)ceilf(P->seed_step * P->id_rate); P->mismatch_thd = 10;
        P->aln_mode |= 2;
    } else {                          /* :1394-1418 */
        DFL(P->seed_step, 25); DFL(P->seed_len, 50);
        P->hash_len = 8; P->hash_step = 4; P->bwt_min_len = 100;
        DFL(P->match, 1); DFL(P->mis, 1);
        DFL(P->ins_gapo, 1); DFL(P->ins_gape, 1); DFL(P->del_gapo, 1); DFL(P->del_gape, 1);
        DFL(P->ins_ext_o, 1); DFL(P->ins_ext_e, 1); DFL(P->del_ext_o, 1); DFL(P->del_ext_e, 1);
        DFL(P->id_rate, (float)0.1); DFL(P->band_w, 100); DFL(P->end_bonus, 0);
        P->match_dis = (int)ceilf(P->seed_step * P->id_rate); P->mismatch_thd = 10;
        P->aln_mode |= 2;
    }
    if (P->seed_step < P->seed_len) P->aln_mode |= 1;   /* :1419 */
    P->seed_inv = P->seed_step - P->seed_len;           /* :1523 */
    for (i = k = 0; i < 4; ++i) {                        /* :1331-1340 */
        for (j = 0; j < 4; ++j) P->sc_mat[k++] = (int8_t)(i == j ? P->match : -P->mis);
        P->sc_mat[k++] = -1;
    }
    for (j = 0; j < 5; ++j) P->sc_mat[k++] = -1;
}
