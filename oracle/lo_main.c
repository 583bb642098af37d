/* lo_main.c -- command line of the oracle: `lamsa_oracle aln [options] <ref> <reads>` (see lo.h).
 * Option letters and semantics follow lamsa_aln (src/lamsa_aln.c:1459-1538), including the
 * missing `break` after -g (:1509-1510).  Seeding is never run: the GEM map file
 * <reads>.seed.gem.map must exist (the reference's -N -I mode).  Stage (4) is not
 * restated, so the output equals the reference's with `-R 0`. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <getopt.h>
#include "lo_io.h"

int main(int argc, char *argv[])
{
    if (argc < 2 || strcmp(argv[1], "aln") != 0) { fprintf(stderr, "usage: lamsa_oracle aln [options] <ref.fa> <reads.fa>\n"); return 1; }
    char pg[4096];
    snprintf(pg, sizeof pg, "@PG\tID:lamsa\tPN:lamsa\tVN:1.0.0\tCL:%s", argv[0]);
    for (int i = 1; i < argc; ++i) snprintf(pg + strlen(pg), sizeof pg - strlen(pg), " %s", argv[i]);
    lo_para P; lo_para_init(&P);
    int c, n_thread = 1; char *p; FILE *out = stdout; long max_reads = 0; int quiet_time = 0;
    static const struct option lopt[] = {
        {"thread",1,0,'t'},{"seed-len",1,0,'l'},{"seed-inv",1,0,'i'},{"max-loci",1,0,'p'},{"SV-len",1,0,'V'},{"ovlp-rat",1,0,'v'},
        {"max-skel",1,0,'s'},{"max-reg",1,0,'R'},{"bwt-kmer",1,0,'k'},{"fastest",0,0,'f'},{"ed-rate",1,0,'e'},{"diff-rate",1,0,'d'},
        {"mis-rate",1,0,'x'},{"read-type",1,0,'T'},{"match-sc",1,0,'m'},{"mis-pen",1,0,'M'},{"open-pen",1,0,'O'},{"ext-pen",1,0,'E'},
        {"band-width",1,0,'w'},{"end-bonus",1,0,'b'},{"max-out",1,0,'r'},{"gap-split",1,0,'g'},{"soft-clip",0,0,'S'},{"comment",0,0,'C'},
        {"output",1,0,'o'},{"max-reads",1,0,1000},{"time",0,0,1001},{0,0,0,0}};
    optind = 2;
    while ((c = getopt_long(argc, argv, "t:l:i:p:V:v:s:R:k:fm:M:O:E:w:b:e:d:x:T:r:g:SCo:NI", lopt, NULL)) >= 0) {
        switch (c) {
        case 't': n_thread = atoi(optarg); break;
        case 'l': P.seed_len = atoi(optarg); break;
        case 'i': P.seed_step = atoi(optarg); break;
        case 'p': P.per_aln_m = atoi(optarg); break;
        case 'V': P.SV_len_thd = atoi(optarg); break;
        case 'v': P.ovlp_rat = (float)atof(optarg); break;
        case 's': P.ske_max = atoi(optarg); break;
        case 'R': P.bwt_max_len = atoi(optarg); break;
        case 'k': P.bwt_seed_len = atoi(optarg); break;
        case 'f': break;
        case 'm': P.match = atoi(optarg); break;
        case 'M': P.mis = atoi(optarg); break;
        case 'O': P.ins_gapo = P.del_gapo = P.ins_ext_o = P.del_ext_o = (int)strtol(optarg, &p, 10);
                  if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) P.del_gapo = (int)strtol(p + 1, &p, 10);
                  if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) P.ins_ext_o = (int)strtol(p + 1, &p, 10);
                  if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) P.del_ext_o = (int)strtol(p + 1, &p, 10);
                  break;
        case 'E': P.ins_gape = P.del_gape = P.ins_ext_e = P.del_ext_e = (int)strtol(optarg, &p, 10);
                  if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) P.del_gape = (int)strtol(p + 1, &p, 10);
                  if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) P.ins_ext_e = (int)strtol(p + 1, &p, 10);
                  if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) P.del_ext_e = (int)strtol(p + 1, &p, 10);
                  break;
        case 'w': P.band_w = atoi(optarg); break;
        case 'b': P.end_bonus = atoi(optarg); break;
        case 'e': case 'x': break;                 /* GEM-only rates */
        case 'd': P.id_rate = (float)atof(optarg); break;
        case 'T': if (!strcmp(optarg, "pacbio")) P.read_type = 1; else if (!strcmp(optarg, "ont2d")) P.read_type = 2; else { fprintf(stderr, "unknown read type %s\n", optarg); return 1; } break;
        case 'r': P.res_mul_max = atoi(optarg); break;
        case 'g': P.split_len = atoi(optarg); /* fall through, as in the reference */
        case 'S': P.supp_soft = 1; break;
        case 'C': P.comm = 1; break;
        case 'o': out = fopen(optarg, "w"); if (!out) { fprintf(stderr, "cannot open %s\n", optarg); return 1; } break;
        case 'N': case 'I': break;
        case 1000: max_reads = atol(optarg); break;
        case 1001: quiet_time = 1; break;
        default: return 1;
        }
    }
    lo_para_finish(&P);
    if (argc - optind != 2) { fprintf(stderr, "usage: lamsa_oracle aln [options] <ref.fa> <reads.fa>\n"); return 1; }
    double secs = 0; long nr = 0, nb = 0;
    int rc = lo_run_aln(argv[optind], argv[optind + 1], &P, quiet_time ? NULL : out, pg, n_thread, max_reads, &secs, &nr, &nb);
    if (quiet_time) printf("{\"reads\": %ld, \"bases\": %ld, \"seconds\": %.6f, \"threads\": %d}\n", nr, nb, secs, n_thread);
    fprintf(stderr, "[lamsa_oracle] %ld reads, %ld bases, %.3f s in the per-read path (%d threads)\n", nr, nb, secs, n_thread);
    if (out != stdout) fclose(out);
    return rc < 0 ? 1 : 0;
}
