// main.cpp -- `lamsa aln` on the MI355X hot path: same command line, scoring options and SAM output as the
// reference's `lamsa aln` (src/main.c:29-44, src/lamsa_aln.c:1424-1538), including its quirks: `-g` also sets
// soft clipping (missing break, :1509-1510); `-C` gates QUAL instead of appending the comment (:1037).
// Seeding works as in the reference: unless -N (or --seed-result FILE) is given, the reads are cut into seeds and the
// GEM mapper of the reference's bundle is run on them (--gem-dir DIR, default <directory of this binary>/gem; the
// <ref>.gem index comes from `lamsa index`, here or the reference's).  Stage (4), the BWT rescue, needs <ref>.bwt and
// <ref>.sa and is skipped with a note when they are missing.  `lamsa index` (index.cpp) writes the reference's index files.
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <getopt.h>
#include <unistd.h>
#include <string>
#include <thread>
#include "lamsa_host.h"

static int usage()
{
    fprintf(stderr, "\nUsage:   lamsa aln [options] <ref.fa> <read.fa/fq>\n\n"
                    "         options of the reference's `lamsa aln` (-t -l -i -p -V -v -s -R -k -f -m -M -O -E -w -b -e -d -x -T -r -g -S -C -o -N -I);\n"
                    "         additionally --device INT (GPU ordinal), --devices INT,INT,... (several GPUs: the chunks of the read stream go round-robin over them,\n"
                    "         every GPU holds the whole reference, output stays in input order), --seed-result FILE (GEM map to use instead of seeding), --gem-dir DIR (where gem-mapper lives),\n"
                    "         --parse-only (read and parse the inputs, print the ingest time; no GPU work, no output),\n"
                    "         --seed-first (wait for the GEM mapper before aligning, as the reference does; default: its map is read while it is being written),\n"
                    "         --save-hits FILE (also write the parsed seed hits as a binary stream), --hits FILE (read that stream instead of the GEM map text:\n"
                    "         a re-run with other alignment options then skips seeding and text parsing; same reads file, same -T/-l/-i/-p),\n"
                    "         --batch INT (reads per GPU batch),\n"
                    "         --shard i/N (align the i-th of N contiguous parts of the read stream: one process -- one parser -- per GPU, e.g. with --device i;\n"
                    "         needs -N or --hits; the outputs of shards 0 .. N-1 written one after the other are the unsharded output, the header comes with shard 0)\n\n");
    return 1;
}

int main(int argc, char *argv[])
{
    if (argc < 2) return usage();
    if (strcmp(argv[1], "index") == 0) {                  // lamsa index [--no-gem] [--gem-dir DIR] <ref.fa>   (src/lamsa_index.c:113)
        std::string gem_dir, fasta; bool with_gem = true, from_pac = false;
        for (int i = 2; i < argc; ++i) {
            if (strcmp(argv[i], "--no-gem") == 0) with_gem = false;
            else if (strcmp(argv[i], "--from-pac") == 0) { from_pac = true; with_gem = false; }      // <ref>.pac / .ann exist: build <ref>.bwt / .sa only
            else if (strcmp(argv[i], "--gem-dir") == 0 && i + 1 < argc) gem_dir = argv[++i];
            else fasta = argv[i];
        }
        if (fasta.empty()) { fprintf(stderr, "\nUsage:   lamsa index [--no-gem] [--gem-dir DIR] [--from-pac] <ref.fa>\n\n"); return 1; }
        if (gem_dir.empty()) { std::string self = argv[0]; const size_t sl = self.rfind('/'); gem_dir = (sl == std::string::npos ? std::string(".") : self.substr(0, sl)) + "/gem"; }
        return lamsa::run_index(fasta, gem_dir, with_gem, from_pac);
    }
    if (strcmp(argv[1], "aln") != 0) { fprintf(stderr, "[main] unrecognized command '%s'\n", argv[1]); return 1; }
    std::string pg = std::string("@PG\tID:lamsa\tPN:lamsa\tVN:1.0.0\tCL:") + argv[0];
    for (int i = 1; i < argc; ++i) { pg += " "; pg += argv[i]; }
    lamsa_hp_para P; lamsa_hp_para_init(&P);
    lamsa::Options opt;
    opt.n_thread = 1;                                  // -t: host threads for parsing, stage (4) and SAM text; 1 unless given, like the reference (src/lamsa_aln.c:1293)
    FILE *out = stdout; char *p; int c;
    static const struct option lopt[] = {
        {"thread",1,0,'t'},{"seed-len",1,0,'l'},{"seed-inv",1,0,'i'},{"max-loci",1,0,'p'},{"SV-len",1,0,'V'},{"ovlp-rat",1,0,'v'},
        {"max-skel",1,0,'s'},{"max-reg",1,0,'R'},{"bwt-kmer",1,0,'k'},{"fastest",0,0,'f'},{"ed-rate",1,0,'e'},{"diff-rate",1,0,'d'},
        {"mis-rate",1,0,'x'},{"read-type",1,0,'T'},{"match-sc",1,0,'m'},{"mis-pen",1,0,'M'},{"open-pen",1,0,'O'},{"ext-pen",1,0,'E'},
        {"band-width",1,0,'w'},{"end-bonus",1,0,'b'},{"max-out",1,0,'r'},{"gap-split",1,0,'g'},{"soft-clip",0,0,'S'},{"comment",0,0,'C'},
        {"output",1,0,'o'},{"help",0,0,'h'},{"HELP",0,0,'H'},{"device",1,0,1000},{"gem-dir",1,0,1003},{"seed-result",1,0,1001},{"batch",1,0,1002},{"parse-only",0,0,1004},{"save-hits",1,0,1005},{"hits",1,0,1006},{"devices",1,0,1007},{"seed-first",0,0,1008},{"shard",1,0,1009},{0,0,0,0}};
    optind = 2;
    while ((c = getopt_long(argc, argv, "t:l:i:p:V:v:s:R:k:fm:M:O:E:w:b:e:d:x:T:r:g:SCo:hHNI", lopt, NULL)) >= 0) {
        switch (c) {
        case 't': opt.n_thread = atoi(optarg); break;
        case 'l': P.seed_len = atoi(optarg); break;
        case 'i': P.seed_step = atoi(optarg); break;
        case 'p': P.per_aln_m = atoi(optarg); break;
        case 'V': P.SV_len_thd = atoi(optarg); break;
        case 'v': P.ovlp_rat = (float)atof(optarg); if (P.ovlp_rat < 0 || P.ovlp_rat > 1) return usage(); break;
        case 's': P.ske_max = atoi(optarg); break;
        case 'R': P.bwt_max_len = atoi(optarg); break;
        case 'k': P.bwt_seed_len = atoi(optarg); break;
        case 'f': opt.fastest = 1; break;
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
        case 'e': opt.ed_rate = (float)atof(optarg); break;         // GEM seeding rates (host side only)
        case 'x': opt.mis_rate = (float)atof(optarg); break;
        case 'd': P.id_rate = (float)atof(optarg); break;
        case 'T': if (!strcmp(optarg, "pacbio")) P.read_type = 1; else if (!strcmp(optarg, "ont2d")) P.read_type = 2;
                  else { fprintf(stderr, "[lamsa_aln] Unkown parameter: %s\n", optarg); return usage(); } break;
        case 'r': P.res_mul_max = atoi(optarg); break;
        case 'g': P.split_len = atoi(optarg);                        // falls through, as in the reference
                  /* fall through */
        case 'S': opt.supp_soft = 1; break;
        case 'C': opt.comm = 1; break;
        case 'o': out = fopen(optarg, "w"); if (!out) { fprintf(stderr, "[lamsa_aln] Can not open output file: %s.\n", optarg); return 1; } break;
        case 'N': opt.no_seed_aln = 1; break;
        case 'I': break;                                            // seed info is recomputed from the read lengths either way
        case 1000: opt.device = atoi(optarg); break;
        case 1003: opt.gem_dir = optarg; break;
            case 1004: opt.parse_only = 1; break;
            case 1005: opt.save_hits = optarg; break;
            case 1006: opt.hits = optarg; opt.no_seed_aln = 1; break;
            case 1008: opt.seed_first = 1; break;
            case 1009: if (sscanf(optarg, "%d/%d", &opt.shard_i, &opt.shard_n) != 2 || opt.shard_n < 1 || opt.shard_i < 0 || opt.shard_i >= opt.shard_n) { fprintf(stderr, "[lamsa_aln] --shard takes i/N with 0 <= i < N\n"); return usage(); } break;
            case 1007: { opt.devices.clear(); for (const char *q = optarg; *q;) { opt.devices.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q == ',') ++q; } break; }
        case 1001: opt.seed_result = optarg; break;
        case 1002: opt.chunk_reads = atoi(optarg) > 0 ? atoi(optarg) : opt.chunk_reads; break;
        default: return usage();
        }
    }
    lamsa_hp_para_finish(&P);
    if (argc - optind != 2) return usage();
    opt.ref_prefix = argv[optind]; opt.reads = argv[optind + 1];
    if (opt.gem_dir.empty()) {                                       // get_bin_dir, src/lamsa_aln.c: <directory of the executable>/gem
        std::string self = argv[0];
        const size_t sl = self.rfind('/');
        opt.gem_dir = (sl == std::string::npos ? std::string(".") : self.substr(0, sl)) + "/gem";
    }
    if (!opt.no_seed_aln && opt.seed_result.empty()) {
        // the mapper is started here, before any GPU call of this process, and (unless --seed-first) left running: run_aln reads its map as it grows
        const int src = lamsa::run_seeding(opt, P, opt.seed_first || opt.parse_only ? nullptr : &opt.mapper_pid);
        if (src) return src;
    }
    lamsa::Stats st;
    fprintf(stderr, "[lamsa_aln] Mapping reads to genome ...\n");
    int rc = lamsa::run_aln(opt, P, out, pg, &st);
    fprintf(stderr, "[lamsa_aln] Mapping done! %ld reads, %ld bases, GPU kernels %.1f ms%s\n", st.n_reads, st.n_bases, st.kernel_ms, st.n_bad ? " (some reads reported unmapped, see above)" : "");
    fprintf(stderr, "[lamsa_aln] wall %.2f s: index load + device setup %.2f; device buffers reserved in %.2f (beside the parse of the first chunk); in the chunk loop (overlapping): read + parse %.2f, check + upload %.2f, wait for the GPU %.2f, rank + SAM %.2f\n",
            st.wall_s, st.load_s, st.reserve_s, st.parse_s, st.submit_s, st.wait_s, st.sam_s);
    if (out != stdout) fclose(out);
    return rc;
}
