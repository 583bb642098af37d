// lamsa_host.h -- host side of `lamsa aln` (see lamsa_host.cpp).  Plain C++17; talks to the GPU only through
// the C-ABI of include/lamsa_hp.h.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "../../include/lamsa_hp.h"

namespace lamsa {

struct Index {                       // .ann contig table + .pac (src/bntseq.h bntseq_t, the parts the path reads)
    std::vector<std::string> name; std::vector<int64_t> off; std::vector<int32_t> len;
    std::vector<uint8_t> pac; int64_t l_pac = 0;
    std::map<std::string, int> name_to_id;
};
bool load_index(const std::string &prefix, Index &ix, std::string &err);

struct Read { std::string name, seq, qual; bool has_qual = false; };

class FastxReader {
  public:
    struct Impl;
    FastxReader(); ~FastxReader();
    bool open(const std::string &path);
    bool next(Read &r);
    // Sharding (--shard i/N).  A mapped (uncompressed, regular) file is cut by bytes: restrict(lo, hi) makes next() deliver the records
    // that START in [lo, hi) of the file; other inputs are cut by record count (count the records with next(), re-open, skip).
    bool mapped() const; size_t size() const;
    void restrict(size_t lo, size_t hi);
    // A mapped FASTA file (first byte '>') can hand out its records as byte spans -- two line searches per single-line record -- and
    // have them parsed later, on any thread: next_span() delivers the next record's [beg, end) (false at the end, or when the file
    // is not a mapped FASTA file), parse_span() is next() on that span.
    bool spans() const;
    bool next_span(size_t &beg, size_t &end);
    bool parse_span(size_t beg, size_t end, Read &r) const;
  private:
    Impl *p;
};

// vector whose resize() leaves new elements uninitialised: the batch arrays are sized once per chunk and then filled
// by all host threads, and a chunk's buffers are recycled, so nothing is ever zero-filled or page-faulted twice
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { typedef NoInitAlloc<U> other; };
    template <class U> void construct(U *p) noexcept { ::new ((void *)p) U; }
    template <class U, class... A> void construct(U *p, A &&... a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};
template <class T> using RawVec = std::vector<T, NoInitAlloc<T>>;

struct Batch {                       // the arrays of lamsa_hp_batch, host side
    std::vector<Read> reads;
    RawVec<int64_t> read_off, seed_off, hit_off, h_pos;
    RawVec<uint8_t> read_seq, h_cig_n, cig8;      // cig8: the seed CIGARs as the GPU boundary's compact form takes them (op << 6 | len)
    bool cig_wide = false;                        // some element is longer than 63: the chunk goes in the word form (cig + h_cig_off)
    bool parse_compact = false, saw_wide = false; // a per-thread part of the parse: the seed CIGARs written as bytes (cig8) straight away; an element that does not fit was met
    RawVec<int32_t> seed_all, last_len, seed_id, h_chr, h_cig_off, cig;
    RawVec<int16_t> h_nm, h_len_dif;
    RawVec<int8_t> h_strand;
    void clear();
};

struct Rec { int64_t offset = 0; int chr = 0, nstrand = 0, score = 0, NM = 0, reg_beg = 0, reg_end = 0; std::vector<int32_t> cigar; };   // res_t
struct XaRef { int st, li, ri; };
struct Line { int line_score = 0, tol_score = 0, tol_NM = 0, merg_x = 0, merg_y = 0; uint8_t mapQ = 0; std::vector<Rec> rec; std::vector<XaRef> xa; };   // line_aln_res
struct ReadResult { int status = 0; std::vector<Line> stage[3]; };   // aln_res[3]: first round, remain round, BWT rescue (empty)

void parse_stream(const int32_t *s, int n_words, int read_len, ReadResult &R);
void rank_results(ReadResult &R, int read_len, const lamsa_hp_para &P);

// a path as one word of a POSIX shell command line: wrapped in single quotes, embedded ones spelled '\''
inline std::string shell_quote(const std::string &p) { std::string o = "'"; for (char c : p) { if (c == '\'') o += "'\\''"; else o += c; } return o + "'"; }

struct Options {
    std::string ref_prefix, reads, seed_result;
    int supp_soft = 0, comm = 0, device = 0, n_thread = 1;
    std::vector<int> devices;                             // --devices 0,1,...: chunks go round-robin over these GPUs (default: --device)
    // seeding front end (lamsa_aln_c, src/lamsa_aln.c:1224-1275): -N reuses <reads>.seed.gem.map, otherwise the read file is
    // cut into seeds and the GEM mapper of the reference's bundle is run on them
    int no_seed_aln = 0, fastest = 0;
    int seed_first = 0;                                   // --seed-first: wait for the mapper before aligning, as the reference does; default: read its map while it is being written
    long mapper_pid = 0;                                  // set by main(): the mapper started by run_seeding and still running
    std::string save_hits, hits;                          // --save-hits FILE: also write the parsed chunks as a binary hit stream; --hits FILE: read that instead of the GEM map text
    int shard_i = 0, shard_n = 1;                         // --shard i/N: this process aligns the i-th of N contiguous parts of the read stream (one process per GPU);
                                                          // the outputs of shards 0 .. N-1 concatenated are the unsharded output (only shard 0 writes the header)
    int parse_only = 0;                                   // --parse-only: read and parse the inputs, no GPU work, no output (ingest timing)
    float ed_rate = -1, mis_rate = -1, mat_rate = -1;     // -e, -x; defaults per read type (src/lamsa_aln.h:26-70)
    std::string gem_dir;                                  // directory holding gem-mapper (default: <directory of this binary>/gem)
    int chunk_reads = 16384; int64_t chunk_bases = 256ll << 20;     // reads per GPU batch (the reference's CHUNK_READ_N is 128 per thread pool)
};
struct Stats { long n_reads = 0, n_bases = 0, n_bad = 0; double kernel_ms = 0;
               double wall_s = 0, load_s = 0, parse_s = 0, submit_s = 0, wait_s = 0, sam_s = 0, reserve_s = 0; };   // where the chunk loop's time went

void sam_header(std::string &o, const Index &ix, const std::string &pg);
void write_sam(std::string &o, const ReadResult &R, const Read &rd, const Index &ix, const Options &opt);
int run_seeding(const Options &opt, const lamsa_hp_para &P, long *pid = nullptr);   // pid: leave the mapper running and return its process id
int run_index(const std::string &fasta, const std::string &gem_dir, bool with_gem, bool from_pac = false);      // index.cpp
int run_aln(const Options &opt, const lamsa_hp_para &P, FILE *out, const std::string &pg_line, Stats *stats);

}  // namespace lamsa
