// index.cpp -- `lamsa index <ref.fa>`: the index files `lamsa aln` reads, in the reference's on-disk formats
// (reference src/lamsa_index.c:69-111 bwt_index, src/bntseq.c:244-345 bns_fasta2bntseq / :82 bns_dump,
// src/bwtindex.c:128 bwt_bwtupdate_core, src/bwt.c:62 bwt_cal_sa, :385-407 the two dumps):
//   <ref>.pac  2 bits per base, forward strand, trailing length-mod-4 byte      <ref>.ann / <ref>.amb  contigs / runs of N
//   <ref>.bwt  BWT of forward + reverse-complement text with occurrence checkpoints every 128 symbols interleaved
//   <ref>.sa   every 32nd suffix-array value
// and, unless --no-gem, the GEM index through the gem-indexer of the reference's bundle (gem/gem_index.sh).
// The files are byte-identical to the reference's.  The suffix array is built in memory by prefix doubling with
// counting sorts (three 32-bit arrays of twice the genome length): fine for bacterial to ~ 500 Mbp genomes; for a human
// genome use the reference's indexer, whose incremental BWT construction needs far less memory.
#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

// suffix array of text[0..N) + sentinel (smaller than every symbol): sa has N + 1 entries, sa[0] = N
void suffix_array(const std::vector<uint8_t> &text, std::vector<uint32_t> &sa)
{
    const size_t N = text.size(), M = N + 1;
    std::vector<uint32_t> rank(M), tmp(M), cnt;
    sa.resize(M);
    for (size_t i = 0; i < N; ++i) rank[i] = (uint32_t)text[i] + 1;
    rank[N] = 0;
    {   // by the first symbol
        cnt.assign(6, 0);
        for (size_t i = 0; i < M; ++i) ++cnt[rank[i] + 1];
        for (size_t c = 1; c < cnt.size(); ++c) cnt[c] += cnt[c - 1];
        for (size_t i = 0; i < M; ++i) sa[cnt[rank[i]]++] = (uint32_t)i;
    }
    {   // dense ranks
        uint32_t r = 0; tmp[sa[0]] = 0;
        for (size_t i = 1; i < M; ++i) { if (rank[sa[i]] != rank[sa[i - 1]]) ++r; tmp[sa[i]] = r; }
        rank.swap(tmp);
    }
    std::vector<uint32_t> sa2(M);
    for (size_t k = 1;; k <<= 1) {
        if (rank[sa[M - 1]] == (uint32_t)(M - 1)) break;                   // all ranks distinct
        // order by the second key (rank of the suffix k further on; none = smallest): those without one first, the rest in sa order
        size_t p = 0;
        for (size_t i = M - k; i < M; ++i) sa2[p++] = (uint32_t)i;
        for (size_t i = 0; i < M; ++i) if (sa[i] >= k) sa2[p++] = sa[i] - (uint32_t)k;
        // stable counting sort by the first key
        cnt.assign(M + 1, 0);
        for (size_t i = 0; i < M; ++i) ++cnt[rank[i] + 1];
        for (size_t c = 1; c <= M; ++c) cnt[c] += cnt[c - 1];
        for (size_t i = 0; i < M; ++i) sa[cnt[rank[sa2[i]]]++] = sa2[i];
        uint32_t r = 0; tmp[sa[0]] = 0;
        for (size_t i = 1; i < M; ++i) {
            const uint32_t a = sa[i - 1], b = sa[i];
            const uint32_t a2 = a + k < M ? rank[a + k] + 1 : 0, b2 = b + k < M ? rank[b + k] + 1 : 0;
            if (rank[a] != rank[b] || a2 != b2) ++r;
            tmp[b] = r;
        }
        rank.swap(tmp);
    }
}
}  // namespace

int run_index(const std::string &fasta, const std::string &gem_dir, bool with_gem)
{
    gzFile fp = gzopen(fasta.c_str(), "r");
    if (!fp) { fprintf(stderr, "[lamsa_index] Can't open %s\n", fasta.c_str()); return 1; }
    fprintf(stderr, "[bwt_index] Building bwt-index for genome...\n");
    // ---- bns_fasta2bntseq: contigs, holes, bases (an N becomes lrand48() & 3 of the generator seeded with 11)
    std::vector<Contig> contigs; std::vector<Hole> holes; std::vector<uint8_t> fwd;
    srand48(11);
    {
        std::string line; char buf[1 << 16]; bool in_seq = false; int lasts = 0;
        auto flush_line = [&]() {
            if (line.empty()) return;
            if (line[0] == '>' || line[0] == '@') {
                Contig c; size_t e = 1; while (e < line.size() && line[e] != ':' && line[e] != ',') ++e;      // the reference's kseq ends a name at ':' or ',' (KS_SEP_REF, src/kseq.h:42), not at a blank
                c.name.assign(line, 1, e - 1); c.offset = (int64_t)fwd.size(); c.len = 0; c.n_ambs = 0;
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
                    fwd.push_back((uint8_t)code); ++c.len;
                }
            }
            line.clear();
        };
        while (gzgets(fp, buf, sizeof buf)) {
            const size_t n = strlen(buf);
            line.append(buf, n);
            if (n && buf[n - 1] == '\n') { while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back(); flush_line(); }
        }
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        flush_line();
        gzclose(fp);
    }
    const int64_t n = (int64_t)fwd.size();
    if (contigs.empty() || n == 0) { fprintf(stderr, "[lamsa_index] no sequence in %s\n", fasta.c_str()); return 1; }
    if (2 * (uint64_t)n + 1 >= 0xfffffff0ull) { fprintf(stderr, "[lamsa_index] genome too long for the in-memory suffix array of this builder (%lld bp); build the index with the reference's `lamsa index`\n", (long long)n); return 1; }
    // ---- .pac (forward), .ann, .amb
    {
        std::vector<uint8_t> pac((size_t)(n >> 2) + ((n & 3) ? 1 : 0), 0);
        for (int64_t i = 0; i < n; ++i) pac[(size_t)(i >> 2)] |= (uint8_t)(fwd[(size_t)i] << ((~i & 3) << 1));
        if (n % 4 == 0) pac.push_back(0);
        pac.push_back((uint8_t)(n % 4));
        if (!write_file(fasta + ".pac", pac.data(), pac.size())) { fprintf(stderr, "[lamsa_index] Can't write %s.pac\n", fasta.c_str()); return 1; }
        std::string ann = std::to_string(n) + " " + std::to_string(contigs.size()) + " 11\n";
        for (const Contig &c : contigs) ann += "0 " + c.name + "\n" + std::to_string(c.offset) + " " + std::to_string(c.len) + " " + std::to_string(c.n_ambs) + "\n";
        std::string amb = std::to_string(n) + " " + std::to_string(contigs.size()) + " " + std::to_string(holes.size()) + "\n";
        for (const Hole &h : holes) { amb += std::to_string(h.offset) + " " + std::to_string(h.len) + " "; amb.push_back(h.amb); amb.push_back('\n'); }
        if (!write_file(fasta + ".ann", ann.data(), ann.size()) || !write_file(fasta + ".amb", amb.data(), amb.size())) { fprintf(stderr, "[lamsa_index] Can't write %s.ann/.amb\n", fasta.c_str()); return 1; }
    }
    // ---- BWT of forward + reverse complement
    const uint64_t N = 2 * (uint64_t)n;
    std::vector<uint8_t> text((size_t)N);
    for (int64_t i = 0; i < n; ++i) { text[(size_t)i] = fwd[(size_t)i]; text[(size_t)(N - 1 - (uint64_t)i)] = (uint8_t)(3 - fwd[(size_t)i]); }
    std::vector<uint8_t>().swap(fwd);
    std::vector<uint32_t> sa;
    suffix_array(text, sa);
    uint64_t primary = 0, L2[5] = {0, 0, 0, 0, 0};
    std::vector<uint8_t> sym((size_t)N);                                    // the BWT string without the sentinel's row
    for (uint64_t i = 0, j = 0; i <= N; ++i) {
        if (sa[(size_t)i] == 0) { primary = i; continue; }
        const uint8_t c = text[(size_t)sa[(size_t)i] - 1];
        sym[(size_t)j++] = c; ++L2[c + 1];
    }
    for (int c = 1; c <= 4; ++c) L2[c] += L2[c - 1];
    {   // occurrence checkpoints interleaved (bwt_bwtupdate_core): per 128 symbols 4 x u64 counts, then 8 words of 16 symbols
        const uint64_t n_occ = (N + 127) / 128 + 1, words = ((N + 15) >> 4) + n_occ * 8;
        std::vector<uint32_t> buf((size_t)words, 0);
        uint64_t c[4] = {0, 0, 0, 0}, k = 0;
        for (uint64_t i = 0; i < N; ++i) {
            if (i % 128 == 0) { memcpy(buf.data() + k, c, 32); k += 8; }
            if (i % 16 == 0) ++k;
            buf[(size_t)(k - 1)] |= (uint32_t)sym[(size_t)i] << ((~i & 0xf) << 1);
            ++c[sym[(size_t)i]];
        }
        memcpy(buf.data() + k, c, 32);
        if (k + 8 != words) { fprintf(stderr, "[lamsa_index] internal error: inconsistent bwt size\n"); return 1; }
        FILE *out = fopen((fasta + ".bwt").c_str(), "wb");
        if (!out || fwrite(&primary, 8, 1, out) != 1 || fwrite(L2 + 1, 8, 4, out) != 4 || fwrite(buf.data(), 4, (size_t)words, out) != (size_t)words || fclose(out) != 0) { fprintf(stderr, "[lamsa_index] Can't write %s.bwt\n", fasta.c_str()); return 1; }
    }
    {   // every 32nd suffix-array value (bwt_cal_sa with intv 32; entry 0 is not stored)
        const uint64_t intv = 32, n_sa = (N + intv) / intv;
        std::vector<uint64_t> s((size_t)n_sa, 0);
        for (uint64_t j = 0; j < n_sa; ++j) s[(size_t)j] = sa[(size_t)(j * intv)];
        FILE *out = fopen((fasta + ".sa").c_str(), "wb");
        if (!out || fwrite(&primary, 8, 1, out) != 1 || fwrite(L2 + 1, 8, 4, out) != 4 || fwrite(&intv, 8, 1, out) != 1 || fwrite(&N, 8, 1, out) != 1 ||
            fwrite(s.data() + 1, 8, (size_t)(n_sa - 1), out) != (size_t)(n_sa - 1) || fclose(out) != 0) { fprintf(stderr, "[lamsa_index] Can't write %s.sa\n", fasta.c_str()); return 1; }
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
