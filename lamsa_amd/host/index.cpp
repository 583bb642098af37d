// index.cpp -- `lamsa index <ref.fa>`: the index files `lamsa aln` reads, in the reference's on-disk formats
// (reference src/lamsa_index.c:69-111 bwt_index, src/bntseq.c:244-345 bns_fasta2bntseq / :82 bns_dump,
// src/bwtindex.c:128 bwt_bwtupdate_core, src/bwt.c:62 bwt_cal_sa, :385-407 the two dumps):
//   <ref>.pac  2 bits per base, forward strand, trailing length-mod-4 byte      <ref>.ann / <ref>.amb  contigs / runs of N
//   <ref>.bwt  BWT of forward + reverse-complement text with occurrence checkpoints every 128 symbols interleaved
//   <ref>.sa   every 32nd suffix-array value
// and, unless --no-gem, the GEM index through the gem-indexer of the reference's bundle (gem/gem_index.sh).
// The files are byte-identical to the reference's.  The suffixes are sorted block by block over the 2-bit text (below): a human genome's
// 6.2 G suffixes need the packed text (1.55 GB), a K-mer histogram (1 GB) and 4 GB per block -- the reference's incremental BWT-SW
// builder (src/bwt_gen.c) is the other way to the same bytes.
#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <memory>
#include <time.h>
#include <string>
#include <thread>
#include <vector>
#include <zlib.h>
#include "lamsa_host.h"

namespace lamsa {

namespace {
struct Contig { std::string name; int64_t offset; int32_t len, n_ambs; };
struct Hole { int64_t offset; int32_t len; char amb; };

uint8_t nt4(int c)
{   // nst_nt4_table, src/bntseq.c:20
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; case '-': return 5; default: return 4; }
}

bool write_file(const std::string &path, const void *p, size_t n) { FILE *fp = fopen(path.c_str(), "wb"); if (!fp) return false; const bool ok = n == 0 || fwrite(p, 1, n, fp) == n; return fclose(fp) == 0 && ok; }

// ---------------------------------------------------------------------------------------------------------------------------------
// Suffix sorting of the forward + reverse-complement text for genomes of any size (a human genome: 6.2 G suffixes), block by block.
// The text is kept 2 bits per base, 32 bases per 64-bit word, first base in the top bits: 32 bases of two suffixes compare as integers.
// The suffixes are cut into blocks by their first K bases (a histogram over all K-mers says where a block of at most `block_max`
// suffixes ends); one pass over the text per block drops the block's suffixes into their K-mer buckets (a counting sort: the block is then
// sorted by K bases), every bucket is sorted by the 32 bases that follow and, where those tie, by comparing on -- all threads at work --
// and the block's part of the BWT and of the sampled suffix array is handed to `emit` in rank order.  Memory: the packed text (1.55 GB
// for a human genome), the histogram (4^K counts), and 16 bytes per suffix of a block; nothing is ever held per suffix of the text.
// (The reference's builder is incremental -- BWT-SW, src/bwt_gen.c -- to the same end; the files come out the same byte for byte.)
struct PackedText {
    std::vector<uint64_t> w; uint64_t N = 0;
    void init(uint64_t n) { N = n; w.assign((size_t)(n >> 5) + 4, 0); }
    void set(uint64_t i, unsigned c) { w[(size_t)(i >> 5)] |= (uint64_t)(c & 3) << (2 * (31 - (i & 31))); }
    unsigned at(uint64_t i) const { return (unsigned)(w[(size_t)(i >> 5)] >> (2 * (31 - (i & 31)))) & 3u; }
    // the 32 bases from i on (zeros beyond the text)
    uint64_t word(uint64_t i) const { const size_t k = (size_t)(i >> 5); const unsigned sh = 2 * (unsigned)(i & 31); return sh ? (w[k] << sh) | (w[k + 1] >> (64 - sh)) : w[k]; }
};

// is suffix a smaller than suffix b?  Compared from `skip` bases in (known equal); the end of the text is smaller than every base.
static inline bool suffix_less(const PackedText &T, uint64_t a, uint64_t b, uint64_t skip)
{
    if (a == b) return false;
    const uint64_t N = T.N;
    uint64_t la = N - a, lb = N - b;                       // lengths
    const uint64_t lim = la < lb ? la : lb;
    uint64_t k = skip < lim ? skip : lim;
    while (k + 32 <= lim) {
        const uint64_t x = T.word(a + k), y = T.word(b + k);
        if (x != y) return x < y;
        k += 32;
    }
    if (k < lim) {
        const unsigned r = (unsigned)(lim - k);              // fewer than 32 bases left of the shorter one
        const uint64_t x = T.word(a + k) >> (2 * (32 - r)), y = T.word(b + k) >> (2 * (32 - r));
        if (x != y) return x < y;
    }
    return la < lb;                                          // one is a prefix of the other: the shorter (it meets the sentinel) is smaller
}

struct SufEnt { uint64_t key, pos; };

template <class Emit>
static void sort_suffixes_blockwise(const PackedText &T, int threads, uint64_t block_max, Emit emit)
{
    const uint64_t N = T.N;
    int K = 14;
    while (K > 2 && ((uint64_t)1 << (2 * K)) > N / 4 + 16) --K;                 // small texts: fewer, fuller buckets
    const uint64_t NB = (uint64_t)1 << (2 * K), kmask = NB - 1;
    if (threads < 1) threads = 1;
    // K-mer of suffix i: its first K bases, zeros beyond the text (such a suffix is the smallest of the bucket it lands in: suffix_less knows the lengths)
    auto kmer_at = [&](uint64_t i) { return T.word(i) >> (2 * (32 - K)); };
    // ---- histogram, by slices of the text
    std::vector<uint32_t> cnt((size_t)NB + 1, 0);
    {
        std::vector<std::thread> th;
        const uint64_t per = (N + (uint64_t)threads - 1) / (uint64_t)threads;
        std::vector<uint8_t> over((size_t)threads, 0);
        for (int t = 0; t < threads; ++t) th.emplace_back([&, t]() {
            const uint64_t lo = per * (uint64_t)t, hi = std::min(N, lo + per);
            if (lo >= hi) return;
            uint64_t km = kmer_at(lo);
            for (uint64_t i = lo; i < hi; ++i) {
                __atomic_fetch_add(&cnt[(size_t)km], 1u, __ATOMIC_RELAXED);
                km = ((km << 2) & kmask) | (i + K < N ? T.at(i + K) : 0);
            }
        });
        for (auto &x : th) x.join();
    }
    // ---- blocks of whole buckets
    uint64_t rank0 = 1;                                                          // rank 0 is the empty suffix (the sentinel alone)
    emit.begin();
    emit.put(0, N);
    std::unique_ptr<SufEnt[]> ent_mem; uint64_t ent_cap = 0;                     // (not a vector: its elements need no initial value -- 4 GB of them)
    std::vector<uint64_t> start;
    const bool trace = getenv("LAMSA_TRACE") != nullptr;
    auto now = []() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; };
    double t_scan = 0, t_sort = 0, t_emit = 0; int n_blocks = 0;
    for (uint64_t b0 = 0; b0 < NB;) {
        uint64_t b1 = b0, tot = 0;
        while (b1 < NB && (tot == 0 || tot + cnt[(size_t)b1] <= block_max)) { tot += cnt[(size_t)b1]; ++b1; }
        if (tot == 0) { b0 = b1; continue; }
        // bucket starts inside the block
        const uint64_t nbk = b1 - b0;
        start.assign((size_t)nbk + 1, 0);
        for (uint64_t b = 0; b < nbk; ++b) start[(size_t)b + 1] = start[(size_t)b] + cnt[(size_t)(b0 + b)];
        if (tot > ent_cap) { ent_mem.reset(); ent_cap = std::max(tot, std::min(block_max, N)); ent_mem.reset(new SufEnt[(size_t)ent_cap]); }
        SufEnt *const ent = ent_mem.get();
        const double t0 = now();
        {   // one pass over the text: the block's suffixes into their buckets (slots claimed atomically; the order inside a bucket is settled by the sort)
            std::vector<uint64_t> fill(start.begin(), start.end() - 1);
            std::vector<std::thread> th;
            const uint64_t per = (N + (uint64_t)threads - 1) / (uint64_t)threads;
            for (int t = 0; t < threads; ++t) th.emplace_back([&, t]() {
                const uint64_t lo = per * (uint64_t)t, hi = std::min(N, lo + per);
                if (lo >= hi) return;
                uint64_t km = kmer_at(lo);
                for (uint64_t i = lo; i < hi; ++i) {
                    if (km >= b0 && km < b1) {
                        const uint64_t at = __atomic_fetch_add(&fill[(size_t)(km - b0)], (uint64_t)1, __ATOMIC_RELAXED);
                        ent[(size_t)at].key = T.word(i + K); ent[(size_t)at].pos = i;
                    }
                    km = ((km << 2) & kmask) | (i + K < N ? T.at(i + K) : 0);
                }
            });
            for (auto &x : th) x.join();
        }
        const double t1 = now();
        {   // every bucket sorted: by the 32 bases behind the K-mer, then by comparing on; buckets handed out to the threads by an atomic counter
            std::atomic<uint64_t> next(0);
            std::vector<std::thread> th;
            for (int t = 0; t < threads; ++t) th.emplace_back([&]() {
                for (;;) {
                    const uint64_t g = next.fetch_add(256);
                    if (g >= nbk) break;
                    for (uint64_t b = g; b < std::min(nbk, g + 256); ++b) {
                        SufEnt *e0 = ent + start[(size_t)b], *e1 = ent + start[(size_t)b + 1];
                        if (e1 - e0 < 2) continue;
                        std::sort(e0, e1, [&](const SufEnt &x, const SufEnt &y) {
                            if (x.key != y.key) {
                                // the keys only decide when both suffixes really have those 32 bases (zeros pad a suffix that ends inside them)
                                if (x.pos + K + 32 <= N && y.pos + K + 32 <= N) return x.key < y.key;
                                return suffix_less(T, x.pos, y.pos, 0);
                            }
                            return suffix_less(T, x.pos, y.pos, (x.pos + K + 32 <= N && y.pos + K + 32 <= N) ? (uint64_t)K + 32 : 0);
                        });
                    }
                }
            });
            for (auto &x : th) x.join();
        }
        const double t2 = now();
        for (uint64_t i = 0; i < tot; ++i) emit.put(rank0 + i, ent[(size_t)i].pos);
        rank0 += tot;
        b0 = b1;
        t_scan += t1 - t0; t_sort += t2 - t1; t_emit += now() - t2; ++n_blocks;
    }
    emit.end();
    if (trace) fprintf(stderr, "[index] %d blocks of at most %llu suffixes, K = %d, %d threads: text passes %.1f s, bucket sorts %.1f s, BWT / SA out %.1f s\n", n_blocks, (unsigned long long)block_max, K, threads, t_scan, t_sort, t_emit);
}

// what the sorter hands over in rank order: the BWT symbols with their occurrence checkpoints (bwt_bwtupdate_core, src/bwtindex.c:128: per 128
// symbols 4 x u64 counts, then 8 words of 16 symbols) and every 32nd suffix-array value (bwt_cal_sa, src/bwt.c:62), both streamed to their files
struct IndexEmit {
    const PackedText *T; FILE *fb, *fs; bool ok = true;
    uint64_t primary = 0, j = 0, c[4] = {0, 0, 0, 0}; uint32_t word = 0;
    std::vector<uint32_t> wbuf; std::vector<uint64_t> sbuf;
    void flush() { if (!wbuf.empty()) { ok = ok && fwrite(wbuf.data(), 4, wbuf.size(), fb) == wbuf.size(); wbuf.clear(); } if (!sbuf.empty()) { ok = ok && fwrite(sbuf.data(), 8, sbuf.size(), fs) == sbuf.size(); sbuf.clear(); } }
    void begin() { wbuf.reserve(1 << 20); sbuf.reserve(1 << 18); }
    void put(uint64_t rank, uint64_t pos) {
        if (rank && (rank & 31) == 0) sbuf.push_back(pos);
        if (pos == 0) { primary = rank; return; }                                // the row of the whole text: its symbol is the sentinel, not stored
        const unsigned sym = T->at(pos - 1);
        if ((j & 127) == 0) { uint32_t t[8]; memcpy(t, c, 32); wbuf.insert(wbuf.end(), t, t + 8); }
        word |= (uint32_t)sym << ((~j & 0xf) << 1);
        ++c[sym]; ++j;
        if ((j & 15) == 0) { wbuf.push_back(word); word = 0; }
        if (wbuf.size() >= (1u << 20) - 16 || sbuf.size() >= (1u << 18) - 2) flush();
    }
    void end() { if (j & 15) wbuf.push_back(word); uint32_t t[8]; memcpy(t, c, 32); wbuf.insert(wbuf.end(), t, t + 8); flush(); }
};
}  // namespace

int run_index(const std::string &fasta, const std::string &gem_dir, bool with_gem, bool from_pac)
{
    std::vector<Contig> contigs; std::vector<Hole> holes; std::vector<uint8_t> pac;
    int64_t n = 0;
    if (from_pac) {
        // <prefix>.pac / .ann exist already (written by this program or the reference's from the FASTA): only .bwt and .sa are built
        FILE *fa = fopen((fasta + ".ann").c_str(), "r");
        long long l_pac = 0; int n_seqs = 0; unsigned seed = 0;
        if (!fa || fscanf(fa, "%lld%d%u", &l_pac, &n_seqs, &seed) != 3 || l_pac <= 0) { fprintf(stderr, "[lamsa_index] Can't read %s.ann\n", fasta.c_str()); if (fa) fclose(fa); return 1; }
        fclose(fa);
        n = l_pac;
        pac.assign((size_t)(n >> 2) + 2, 0);
        FILE *fp2 = fopen((fasta + ".pac").c_str(), "rb");
        if (!fp2 || fread(pac.data(), 1, (size_t)((n + 3) >> 2), fp2) != (size_t)((n + 3) >> 2)) { fprintf(stderr, "[lamsa_index] Can't read %s.pac\n", fasta.c_str()); if (fp2) fclose(fp2); return 1; }
        fclose(fp2);
        fprintf(stderr, "[bwt_index] Building bwt-index for genome (from %s.pac, %lld bp)...\n", fasta.c_str(), l_pac);
    }
    gzFile fp = from_pac ? nullptr : gzopen(fasta.c_str(), "r");
    if (!from_pac && !fp) { fprintf(stderr, "[lamsa_index] Can't open %s\n", fasta.c_str()); return 1; }
    if (!from_pac) fprintf(stderr, "[bwt_index] Building bwt-index for genome...\n");
    // ---- bns_fasta2bntseq: contigs, holes, bases (an N becomes lrand48() & 3 of the generator seeded with 11), packed as they are read
    srand48(11);
    if (!from_pac) {
        std::string line; std::vector<char> buf(1 << 20); bool in_seq = false; int lasts = 0;
        auto flush_line = [&]() {
            if (line.empty()) return;
            if (line[0] == '>' || line[0] == '@') {
                Contig c; size_t e = 1; while (e < line.size() && line[e] != ':' && line[e] != ',') ++e;      // the reference's kseq ends a name at ':' or ',' (KS_SEP_REF, src/kseq.h:42), not at a blank
                c.name.assign(line, 1, e - 1); c.offset = n; c.len = 0; c.n_ambs = 0;
                contigs.push_back(c); in_seq = true; lasts = 0;
            } else if (in_seq) {
                Contig &c = contigs.back();
                for (char ch : line) {
                    if (!isgraph((unsigned char)ch)) continue;
                    int code = nt4(ch);
                    if (code >= 4) {
                        if (lasts == ch) ++holes.back().len;                    // the same ambiguity character again: the run grows
                        else { Hole h; h.offset = c.offset + c.len; h.len = 1; h.amb = ch; holes.push_back(h); ++c.n_ambs; }
                        code = (int)(lrand48() & 3);
                    }
                    lasts = ch;
                    if ((n & 3) == 0) pac.push_back(0);
                    pac.back() |= (uint8_t)(code << ((~n & 3) << 1));
                    ++n; ++c.len;
                }
            }
            line.clear();
        };
        while (gzgets(fp, buf.data(), (int)buf.size())) {
            const size_t k = strlen(buf.data());
            line.append(buf.data(), k);
            if (k && buf[k - 1] == '\n') { while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back(); flush_line(); }
        }
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        flush_line();
        gzclose(fp);
    }
    if ((contigs.empty() && !from_pac) || n == 0) { fprintf(stderr, "[lamsa_index] no sequence in %s\n", fasta.c_str()); return 1; }
    // ---- the text: forward strand, then its reverse complement (src/lamsa_index.c:72-76)
    const uint64_t N = 2 * (uint64_t)n;
    PackedText T; T.init(N);
    uint64_t L2[5] = {0, 0, 0, 0, 0};
    for (int64_t i = 0; i < n; ++i) {
        const unsigned c = (pac[(size_t)(i >> 2)] >> ((~i & 3) << 1)) & 3u;
        T.set((uint64_t)i, c); T.set(N - 1 - (uint64_t)i, 3 - c);
        ++L2[c + 1]; ++L2[(3 - c) + 1];
    }
    for (int c = 1; c <= 4; ++c) L2[c] += L2[c - 1];
    // ---- .pac (forward), .ann, .amb
    if (from_pac) std::vector<uint8_t>().swap(pac);
    else {
        if (n % 4 == 0) pac.push_back(0);
        pac.push_back((uint8_t)(n % 4));
        if (!write_file(fasta + ".pac", pac.data(), pac.size())) { fprintf(stderr, "[lamsa_index] Can't write %s.pac\n", fasta.c_str()); return 1; }
        std::vector<uint8_t>().swap(pac);
        std::string ann = std::to_string(n) + " " + std::to_string(contigs.size()) + " 11\n";
        for (const Contig &c : contigs) ann += "0 " + c.name + "\n" + std::to_string(c.offset) + " " + std::to_string(c.len) + " " + std::to_string(c.n_ambs) + "\n";
        std::string amb = std::to_string(n) + " " + std::to_string(contigs.size()) + " " + std::to_string(holes.size()) + "\n";
        for (const Hole &h : holes) { amb += std::to_string(h.offset) + " " + std::to_string(h.len) + " "; amb.push_back(h.amb); amb.push_back('\n'); }
        if (!write_file(fasta + ".ann", ann.data(), ann.size()) || !write_file(fasta + ".amb", amb.data(), amb.size())) { fprintf(stderr, "[lamsa_index] Can't write %s.ann/.amb\n", fasta.c_str()); return 1; }
    }
    // ---- .bwt and .sa, streamed while the suffixes are sorted block by block
    {
        int threads = (int)std::thread::hardware_concurrency();
        if (const char *e = getenv("LAMSA_INDEX_THREADS")) threads = atoi(e);
        if (threads < 1) threads = 1;
        if (threads > 64) threads = 64;
        uint64_t block_max = (uint64_t)1 << 28;                                  // 4 GB of sort entries per block
        if (const char *e = getenv("LAMSA_INDEX_BLOCK")) block_max = (uint64_t)atoll(e) > 0 ? (uint64_t)atoll(e) : block_max;
        FILE *fb = fopen((fasta + ".bwt").c_str(), "wb"), *fs = fopen((fasta + ".sa").c_str(), "wb");
        const uint64_t intv = 32, zero = 0;
        bool ok = fb && fs;
        // headers first (the primary index is known at the end and written over its place then)
        ok = ok && fwrite(&zero, 8, 1, fb) == 1 && fwrite(L2 + 1, 8, 4, fb) == 4;
        ok = ok && fwrite(&zero, 8, 1, fs) == 1 && fwrite(L2 + 1, 8, 4, fs) == 4 && fwrite(&intv, 8, 1, fs) == 1 && fwrite(&N, 8, 1, fs) == 1;
        if (!ok) { fprintf(stderr, "[lamsa_index] Can't write %s.bwt / .sa\n", fasta.c_str()); if (fb) fclose(fb); if (fs) fclose(fs); return 1; }
        IndexEmit E; E.T = &T; E.fb = fb; E.fs = fs;
        struct Ref { IndexEmit *e; void begin() { e->begin(); } void put(uint64_t r, uint64_t p) { e->put(r, p); } void end() { e->end(); } } R{&E};
        sort_suffixes_blockwise(T, threads, block_max, R);
        ok = E.ok && E.j == N;
        ok = ok && fseek(fb, 0, SEEK_SET) == 0 && fwrite(&E.primary, 8, 1, fb) == 1 && fseek(fs, 0, SEEK_SET) == 0 && fwrite(&E.primary, 8, 1, fs) == 1;
        ok = (fclose(fb) == 0) & (fclose(fs) == 0) & ok;
        if (!ok) { fprintf(stderr, "[lamsa_index] Can't write %s.bwt / .sa\n", fasta.c_str()); return 1; }
    }
    fprintf(stderr, "[bwt_index] Building done!\n");
    if (!with_gem) return 0;
    // ---- GEM index (gem_build, src/lamsa_index.c:30): the indexer of the reference's bundle, as gem/gem_index.sh calls it
    const std::string indexer = gem_dir + "/gem-indexer";
    FILE *probe = fopen(indexer.c_str(), "r");
    if (!probe) { fprintf(stderr, "[lamsa_index] gem-indexer not found at %s (it ships with the reference; give its directory with --gem-dir, or pass --no-gem)\n", indexer.c_str()); return 1; }
    fclose(probe);
    fprintf(stderr, "[lamsa_index] Executing gem-indexer ... ");
    const std::string cmd = "PATH=\"$PATH\":" + lamsa::shell_quote(gem_dir) + " " + lamsa::shell_quote(indexer) + " -i " + lamsa::shell_quote(fasta) + " -o " + lamsa::shell_quote(fasta) + " >/dev/null 2>/dev/null";
    if (system(cmd.c_str()) != 0) { fprintf(stderr, "\n[lamsa_index] Indexing undone, gem-indexer exit abnormally.\n"); return 1; }
    remove((fasta + ".log").c_str());
    fprintf(stderr, "done!\n");
    return 0;
}

}  // namespace lamsa
