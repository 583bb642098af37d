// hp_para.h -- parameter defaults / presets of the C-ABI (plain host C++; shared by the HIP library and the
// CPU lane-emulation build used by the tests).
#pragma once
#include <math.h>
#include <string.h>
#include "../../include/lamsa_hp.h"

extern "C" void lamsa_hp_para_init(lamsa_hp_para *P)
{   // init_aln_para, reference src/lamsa_aln.c:1281-1329
    memset(P, 0, sizeof(*P));
    P->seed_len = P->seed_step = -1;
    P->per_aln_m = 200; P->first_loci_thd = 2;
    P->SV_len_thd = 10000; P->ske_max = 10; P->ovlp_rat = (float)0.7;
    P->split_len = 100; P->split_pen = 10; P->res_mul_max = 10;
    P->hash_key_len = 2; P->hash_size = 16;
    P->bwt_seed_len = 19; P->bwt_max_len = 300;
    P->match = P->mis = -1;
    P->ins_gapo = P->del_gapo = P->ins_gape = P->del_gape = -1;
    P->ins_ext_o = P->del_ext_o = P->ins_ext_e = P->del_ext_e = -1;
    P->id_rate = -1; P->read_type = 0; P->band_w = -1; P->end_bonus = -1; P->zdrop = 100; P->aln_mode = 0;
}

static inline void dfl(int32_t &x, int v) { if (x < 0) x = v; }
extern "C" void lamsa_hp_para_finish(lamsa_hp_para *P)
{   // lamsa_set_aln_mode, reference src/lamsa_aln.c:1342-1420; seed_inv :1523
    const int t = P->read_type;
    const int ext_o = t == 1 ? 2 : (t == 2 ? 1 : 5), ext_e = t == 0 ? 2 : 1;
    dfl(P->seed_step, t == 0 ? 100 : 25); dfl(P->seed_len, 50);
    P->hash_len = t == 0 ? 10 : 8; P->hash_step = t == 0 ? 10 : 4;
    P->bwt_min_len = t == 0 ? P->bwt_seed_len : (t == 1 ? 50 : 100);
    dfl(P->match, 1); dfl(P->mis, t == 0 ? 3 : 1);
    dfl(P->ins_gapo, t == 0 ? 5 : 1); dfl(P->ins_gape, t == 0 ? 2 : 1);
    dfl(P->del_gapo, t == 0 ? 5 : 1); dfl(P->del_gape, t == 0 ? 2 : 1);
    dfl(P->ins_ext_o, ext_o); dfl(P->ins_ext_e, ext_e); dfl(P->del_ext_o, ext_o); dfl(P->del_ext_e, ext_e);
    if (P->id_rate < 0) P->id_rate = t == 0 ? (float)0.04 : (t == 1 ? (float)0.3 : (float)0.1);
    dfl(P->band_w, t == 0 ? 10 : (t == 1 ? 200 : 100)); dfl(P->end_bonus, t == 0 ? 5 : 0);
    P->match_dis = t == 0 ? 5 : (int)ceilf(P->seed_step * P->id_rate);
    P->mismatch_thd = 10;
    if (t != 0) P->aln_mode |= 2;
    if (P->seed_step < P->seed_len) P->aln_mode |= 1;
    P->seed_inv = P->seed_step - P->seed_len;
}

