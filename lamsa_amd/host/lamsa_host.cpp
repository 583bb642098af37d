// lamsa_host.cpp -- host side of `lamsa aln`: everything around the MI355X hot path.
//
// Mirrors the reference's driver for the path (same files, same option letters, same SAM):
//   load_index      <- bns_restore_core + .pac read            src/bntseq.c:114-166, src/lamsa_aln.c:1237-1239
//   FastxReader     <- kseq_read with KS_SEP_REF names          src/kseq.h:179-225 (name ends at '\n', ':' or ',')
//   parse_gem_hits  <- gem_map_read / gem_map_msg / md2cigar    src/gem_parse.c:74-286, map_cal_msg src/lamsa_aln.c:767
//   cov_fraction    <- get_reg + get_cov_f                      src/lamsa_aln.c:571-651
//   rank_results    <- rearr_aln_res                            src/lamsa_aln.c:654-724
//   write_sam       <- aln_res_output / print_sam_header        src/lamsa_aln.c:1001-1100,1215
//   run_aln         <- lamsa_aln_core chunk loop                src/lamsa_aln.c:1116-1177
// The per-read stages (2),(3),(2'),(3') are NOT here: they run on the GPU behind lamsa_hp_align_batch()
// (include/lamsa_hp.h).  Stage (4), the BWT rescue of short uncovered regions (src/bwt_aln.c), is in rescue.cpp.
#include "lamsa_host.h"
#include "rescue.h"
#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <condition_variable>
#include <functional>
#include <future>
#include <mutex>
#include <memory>
#include <thread>
#include <time.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <fcntl.h>
#include <signal.h>
#include <sstream>
#include <sys/wait.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace lamsa {

// ------------------------------------------------------------------ index
bool load_index(const std::string &prefix, Index &ix, std::string &err)
{
    FILE *fp = fopen((prefix + ".ann").c_str(), "r");
    if (!fp) { err = "cannot open " + prefix + ".ann"; return false; }
    long long l_pac; int n_seqs; unsigned seed;
    if (fscanf(fp, "%lld%d%u", &l_pac, &n_seqs, &seed) != 3) { fclose(fp); err = "bad .ann header"; return false; }
    ix.name.resize(n_seqs); ix.off.resize(n_seqs); ix.len.resize(n_seqs);
    for (int i = 0; i < n_seqs; ++i) {
        unsigned gi; int c, n_ambs; long long off; std::string line;
        if (fscanf(fp, "%u", &gi) != 1) { fclose(fp); err = "bad .ann record"; return false; }
        while ((c = fgetc(fp)) != '\n' && c != EOF) line.push_back((char)c);
        {   // bns_restore_core reads the name with %s: the first blank-delimited word; the rest of the line is the comment
            size_t a = 0; while (a < line.size() && isspace((unsigned char)line[a])) ++a;
            size_t b = a; while (b < line.size() && !isspace((unsigned char)line[b])) ++b;
            ix.name[i] = line.substr(a, b - a);
        }
        if (fscanf(fp, "%lld%d%d", &off, &ix.len[i], &n_ambs) != 3) { fclose(fp); err = "bad .ann record"; return false; }
        ix.off[i] = off;
    }
    fclose(fp);
    fp = fopen((prefix + ".pac").c_str(), "rb");
    if (!fp) { err = "cannot open " + prefix + ".pac"; return false; }
    ix.l_pac = l_pac;
    ix.pac.assign((size_t)(l_pac / 4 + 1) + 16, 0);
    size_t got = fread(ix.pac.data(), 1, (size_t)(l_pac / 4 + 1), fp);
    fclose(fp);
    if (got == 0 && l_pac > 4) { err = "empty .pac"; return false; }
    for (int i = 0; i < n_seqs; ++i) ix.name_to_id[ix.name[i]] = i + 1;
    return true;
}

// A mapped input file is touched page by page by a helper thread that runs ahead of the reader, so that the reader
// itself does not take the page faults (they cost more than finding the line ends).
struct Prefault {
    std::thread th; std::atomic<bool> stop{false};
    void start(const char *p, size_t n) {
        th = std::thread([this, p, n]() { unsigned sink = 0; for (size_t i = 0; i < n && !stop.load(std::memory_order_relaxed); i += 4096) sink += (unsigned char)((const volatile char *)p)[i]; (void)sink; });
    }
    void finish() { stop = true; if (th.joinable()) th.join(); }
};

// ------------------------------------------------------------------ FASTA / FASTQ (+gz)
struct FastxReader::Impl {
    gzFile f = nullptr; std::string line; bool have = false;
    const char *m = nullptr; size_t mn = 0, mpos = 0;       // an uncompressed regular file is mapped instead of read through zlib
    size_t limit = (size_t)-1; size_t line_at = 0;           // restrict(): records that start at or beyond `limit` are not delivered; line_at: where the current line began
    Prefault pf;
};
bool FastxReader::mapped() const { return p->m != nullptr; }
size_t FastxReader::size() const { return p->mn; }
// is there a record header at offset `at` (which is the beginning of a line)?  The file's first byte says which kind it is.  FASTA: a line
// that begins with '>'.  FASTQ (four lines per record here, as every shardable file has): a line that begins with '@' and whose line after
// next begins with '+' -- a quality line may begin with '@' or '>' (Phred 31 / 29) too, but the line two below such a one is a sequence line.
static bool record_starts_at(const char *m, size_t n, size_t at)
{
    if (at >= n || n == 0) return false;
    if (m[0] != '@') return m[at] == '>';
    if (m[at] != '@') return false;
    const char *l1 = (const char *)memchr(m + at, '\n', n - at);
    if (!l1) return false;
    const char *l2 = (const char *)memchr(l1 + 1, '\n', (size_t)(m + n - (l1 + 1)));
    return l2 && l2 + 1 < m + n && l2[1] == '+';
}
void FastxReader::restrict(size_t lo, size_t hi)
{
    if (!p->m) return;
    size_t at = lo;
    if (at > 0) {                                             // the first beginning of a line at or after lo ...
        const char *nl = (const char *)memchr(p->m + at - 1, '\n', p->mn - (at - 1));
        at = nl ? (size_t)(nl - p->m) + 1 : p->mn;
    }
    while (at < p->mn && !record_starts_at(p->m, p->mn, at)) {    // ... that begins a record
        const char *nl = (const char *)memchr(p->m + at, '\n', p->mn - at);
        at = nl ? (size_t)(nl - p->m) + 1 : p->mn;
    }
    p->mpos = at; p->have = false; p->limit = hi;
}
FastxReader::FastxReader() : p(new Impl) {}
FastxReader::~FastxReader() { p->pf.finish(); if (p->f) gzclose(p->f); if (p->m) munmap((void *)p->m, p->mn); delete p; }
bool FastxReader::open(const std::string &path)
{
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd >= 0) {
        struct stat st; unsigned char magic[2] = {0, 0};
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && pread(fd, magic, 2, 0) == 2 && !(magic[0] == 0x1f && magic[1] == 0x8b)) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) { madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL); p->m = (const char *)m; p->mn = (size_t)st.st_size; ::close(fd); p->pf.start(p->m, p->mn); return true; }
        }
        ::close(fd);
    }
    p->f = gzopen(path.c_str(), "r");
    return p->f != nullptr;
}
static bool next_line(FastxReader::Impl *p)
{
    if (p->have) { p->have = false; return true; }
    if (p->m) {
        if (p->mpos >= p->mn) return false;
        p->line_at = p->mpos;
        const char *a = p->m + p->mpos, *nl = (const char *)memchr(a, '\n', p->mn - p->mpos);
        const char *e = nl ? nl : p->m + p->mn;
        p->mpos = (size_t)(e - p->m) + (nl ? 1 : 0);
        while (e > a && (e[-1] == '\r' || e[-1] == '\n')) --e;
        p->line.assign(a, (size_t)(e - a));
        return true;
    }
    p->line.clear();
    char buf[1 << 16];
    bool any = false;
    while (gzgets(p->f, buf, sizeof buf)) {
        any = true;
        size_t n = strlen(buf);
        p->line.append(buf, n);
        if (n && buf[n - 1] == '\n') break;
    }
    if (!any) return false;
    while (!p->line.empty() && (p->line.back() == '\n' || p->line.back() == '\r')) p->line.pop_back();
    return true;
}
// A record of a mapped file, read straight out of the mapping: one copy of the sequence (into r.seq), none of the lines -- the sequential
// pass over the read file is what bounds the host side once the hits come from a binary stream (tools/cli_bench.py).
static bool next_mapped(FastxReader::Impl *p, Read &r)
{
    const char *m = p->m; const size_t n = p->mn;
    size_t at = p->mpos;
    auto line_end = [&](size_t from, size_t &next) { const char *nl = (const char *)memchr(m + from, '\n', n - from); size_t e = nl ? (size_t)(nl - m) : n; next = nl ? e + 1 : n; while (e > from && m[e - 1] == '\r') --e; return e; };
    // the next header line
    for (;;) {
        if (at >= n) { p->mpos = n; return false; }
        if (m[at] == '>' || m[at] == '@') break;
        size_t nx; line_end(at, nx); at = nx;
    }
    if (at >= p->limit) { p->mpos = at; return false; }                        // the record belongs to the next shard
    const bool fastq = m[at] == '@';
    size_t nx; size_t e = line_end(at, nx);
    { const char *h = m + at + 1; size_t k = 0, hl = e - (at + 1); while (k < hl && h[k] != ':' && h[k] != ',') ++k; r.name.assign(h, k); }
    at = nx;
    bool plus = false;
    while (at < n) {                                                            // sequence lines, up to the next header or the '+' of a FASTQ record
        const char c0 = m[at];
        if (c0 == '>' || c0 == '@') break;
        if (c0 == '+') { plus = true; line_end(at, nx); at = nx; break; }
        e = line_end(at, nx);
        const size_t k0 = r.seq.size();
        r.seq.append(m + at, e - at);
        unsigned dirty = 0;                                                     // kseq keeps the isgraph() characters of a sequence line (33..126 in the C locale)
        { const char *d = r.seq.data(); const size_t k1 = r.seq.size(); for (size_t k = k0; k < k1; ++k) dirty |= (unsigned)((unsigned char)(d[k] - 33) >= 94); }
        if (dirty) r.seq.erase(std::remove_if(r.seq.begin() + (long)k0, r.seq.end(), [](char c) { return (unsigned char)(c - 33) >= 94; }), r.seq.end());
        at = nx;
    }
    if (fastq && plus) {
        while (r.qual.size() < r.seq.size() && at < n) { e = line_end(at, nx); r.qual.append(m + at, e - at); at = nx; }
        if (r.qual.size() > r.seq.size()) r.qual.resize(r.seq.size());
        r.has_qual = true;
    }
    p->mpos = at;
    return true;
}

bool FastxReader::spans() const { return p->m != nullptr && p->mn > 0 && p->m[0] == '>'; }
bool FastxReader::next_span(size_t &beg, size_t &end)
{
    if (!spans() || p->have) return false;
    const char *m = p->m; const size_t n = p->mn;
    size_t at = p->mpos;
    while (at < n && m[at] != '>') { const char *nl = (const char *)memchr(m + at, '\n', n - at); at = nl ? (size_t)(nl - m) + 1 : n; }      // the next header line
    if (at >= n) { p->mpos = n; return false; }
    if (at >= p->limit) { p->mpos = at; return false; }                        // the record belongs to the next shard
    beg = at;
    do { const char *nl = (const char *)memchr(m + at, '\n', n - at); at = nl ? (size_t)(nl - m) + 1 : n; } while (at < n && m[at] != '>');
    end = at; p->mpos = at;
    return true;
}
bool FastxReader::parse_span(size_t beg, size_t end, Read &r) const
{
    Impl q; q.m = p->m; q.mn = end; q.mpos = beg;                              // a view of the mapping that ends with the record (it owns nothing)
    r.seq.clear(); r.qual.clear(); r.has_qual = false;
    const bool ok = next_mapped(&q, r);
    q.m = nullptr;
    return ok;
}

bool FastxReader::next(Read &r)
{
    r.seq.clear(); r.qual.clear(); r.has_qual = false;
    if (p->m && !p->have) return next_mapped(p, r);
    do { if (!next_line(p)) return false; } while (p->line.empty() || (p->line[0] != '>' && p->line[0] != '@'));
    if (p->m && p->line_at >= p->limit) { p->have = true; return false; }     // the record belongs to the next shard
    const bool fastq = p->line[0] == '@';
    size_t nl = strcspn(p->line.c_str() + 1, ":,");
    r.name.assign(p->line, 1, nl);
    while (next_line(p)) {
        const char c0 = p->line.empty() ? 0 : p->line[0];
        if (c0 == '>' || c0 == '+' || c0 == '@') { if (c0 != '+') p->have = true; break; }
        const size_t k0 = r.seq.size();                                   // kseq keeps the isgraph() characters of a sequence line
        r.seq.append(p->line);
        unsigned dirty = 0;                                               // isgraph() in the C locale: 33..126
        { const char *d = r.seq.data(); const size_t k1 = r.seq.size(); for (size_t k = k0; k < k1; ++k) dirty |= (unsigned)((unsigned char)(d[k] - 33) >= 94); }
        if (dirty) r.seq.erase(std::remove_if(r.seq.begin() + (long)k0, r.seq.end(), [](char c) { return (unsigned char)(c - 33) >= 94; }), r.seq.end());
    }
    if (fastq && !p->have) {
        while (r.qual.size() < r.seq.size() && next_line(p)) r.qual += p->line;
        if (r.qual.size() > r.seq.size()) r.qual.resize(r.seq.size());
        r.has_qual = true;
    }
    return true;
}

// ------------------------------------------------------------------ GEM map line -> hits
template <class V> static inline void cig_push1(V &c, size_t base, int32_t w)
{   // _push_cigar1, src/frag_check.h:153 (restricted to the CIGAR that starts at `base`)
    if ((w >> 4) == 0) return;
    if (c.size() > base && (c.back() & 0xf) == (w & 0xf)) { c.back() += (w >> 4) << 4; return; }
    c.push_back(w);
}

// the same on the compact form (op << 6 | len, len <= 63, ops M / I / D); false: the element does not fit
static inline bool cig8_push1(RawVec<uint8_t> &c, size_t base, int32_t w)
{
    const int len = w >> 4, op = w & 0xf;
    if (len == 0) return true;
    if (c.size() > base && (c.back() >> 6) == op) { const int nl = (c.back() & 63) + len; c.back() = (uint8_t)((op << 6) | (nl & 63)); return nl <= 63 && op <= 2; }
    c.push_back((uint8_t)((op << 6) | (len & 63)));
    return len <= 63 && op <= 2;
}

// one hit "chr:strand:pos:gigar" appended to the batch; md2cigar, src/gem_parse.c:74-112.  The text is a view into the
// mapped file: every scan is bounded by the token's end.
static void add_hit(Batch &B, const Index &ix, const char *tok, size_t len)
{
    const char *e = tok + len, *c1 = tok;
    while (c1 < e && *c1 != ':') ++c1;
    if (c1 >= e) return;
    const char strand = c1 + 1 < e ? c1[1] : '+';
    const char *c2 = c1 + 2;
    while (c2 < e && *c2 != ':') ++c2;
    if (c2 >= e) return;
    long long pos = 0;                                                     // atoll on the field
    const char *c3 = c2 + 1;
    { bool neg = false; if (c3 < e && (*c3 == '-' || *c3 == '+')) { neg = *c3 == '-'; ++c3; } for (; c3 < e && *c3 >= '0' && *c3 <= '9'; ++c3) pos = pos * 10 + (*c3 - '0'); if (neg) pos = -pos; }
    while (c3 < e && *c3 != ':') ++c3;
    const char *md = c3 < e ? c3 + 1 : e, *mdend = md;
    while (mdend < e && *mdend != ':') ++mdend;
    const bool compact = B.parse_compact;
    const size_t base = compact ? B.cig8.size() : B.cig.size();
    bool fits = true;
#define HP_HIT_PUSH(w_) do { if (compact) fits &= cig8_push1(B.cig8, base, (w_)); else cig_push1(B.cig, base, (w_)); } while (0)
    int nm = 0, bd = 0, bi = 0;
    for (const char *q = md; q < mdend;) {
        if (*q == '>') {
            int n = 0;                                                     // atoi(q + 1)
            const char *s = q + 1;
            bool neg = false; if (s < mdend && (*s == '-' || *s == '+')) { neg = *s == '-'; ++s; }
            const char *d0 = s;
            for (; s < mdend && *s >= '0' && *s <= '9'; ++s) n = n * 10 + (*s - '0');
            if (neg) { n = -n; s = d0 - 1; }                               // ">-3": atoi takes the sign, the scan below stops at it
            while (s < mdend && *s != '+' && *s != '-') ++s;
            if (s < mdend && *s == '+') { bd += n; HP_HIT_PUSH((n << 4) | 2); } else { bi += n; HP_HIT_PUSH((n << 4) | 1); }
            nm += n;
            int d = 1; for (int t = n; t >= 10; t /= 10) ++d;
            q += d + 2;
        } else {
            int m = 0, mm = 0, run = 0; bool in_run = false;
            for (; q < mdend && *q != '>'; ++q) {
                const char ch = *q;
                if (ch >= 'A' && ch <= 'T') { ++mm; if (in_run) { m += run; run = 0; in_run = false; } }
                else if (ch >= '0' && ch <= '9') { run = in_run ? run * 10 + (ch - '0') : (ch - '0'); in_run = true; }
                else if (in_run) { m += run; run = 0; in_run = false; }
            }
            if (in_run) m += run;
            HP_HIT_PUSH(((m + mm) << 4) | 0);
            nm += mm;
        }
    }
#undef HP_HIT_PUSH
    if (!fits) B.saw_wide = true;
    if (strand == '-') { if (compact) std::reverse(B.cig8.begin() + (long)base, B.cig8.end()); else std::reverse(B.cig.begin() + (long)base, B.cig.end()); }      // _invert_cigar, src/gem_parse.c:267
    // contig name -> id (map_cal_msg's strcmp scan, src/lamsa_aln.c:767); consecutive hits mostly share the contig
    // (the hits of a repeat seed go from contig to contig: a small cache per thread, keyed by a hash of the name, in front of the map)
    struct NameSlot { std::string name; int id = 0; const Index *ix = nullptr; };
    static thread_local NameSlot cache[256];
    const size_t nl = (size_t)(c1 - tok);
    unsigned hsh = 2166136261u;
    for (size_t k = 0; k < nl; ++k) hsh = (hsh ^ (unsigned char)tok[k]) * 16777619u;
    NameSlot &slot = cache[(hsh ^ (hsh >> 8) ^ (hsh >> 16)) & 255];
    if (slot.ix != &ix || slot.name.size() != nl || memcmp(slot.name.data(), tok, nl) != 0) {
        slot.name.assign(tok, nl); slot.ix = &ix;
        auto it = ix.name_to_id.find(slot.name);
        slot.id = it == ix.name_to_id.end() ? -1 : it->second;
    }
    const int last_id = slot.id;
    B.h_pos.push_back(pos); B.h_chr.push_back(last_id); B.h_strand.push_back(strand == '+' ? 1 : -1);
    B.h_nm.push_back((int16_t)nm); B.h_len_dif.push_back((int16_t)(bd - bi));
    if (compact) {
        if (B.cig8.size() - base > 255) B.cig8.resize(base + 255);         // the boundary counts a seed's CIGAR in 8 bits: the surplus goes, so that the counts add up to the arena
        B.h_cig_n.push_back((uint8_t)(B.cig8.size() - base));
    } else {
        B.h_cig_off.push_back((int32_t)base);
        if (B.cig.size() - base > 255) B.cig.resize(base + 255);
        B.h_cig_n.push_back((uint8_t)(B.cig.size() - base));
    }
}

// all hits of one seed; more than max_n hits: the seed keeps its slot but loses all hits (src/gem_parse.c:243-246)
static void parse_gem_hits(Batch &B, const Index &ix, const char *s, const char *end, int max_n)
{
    const size_t h0 = B.h_pos.size(), c0 = B.parse_compact ? B.cig8.size() : B.cig.size();
    int n = 0;
    for (const char *p = s; p < end;) {
        while (p < end && *p == ',') ++p;
        if (p >= end) break;
        const char *e = (const char *)memchr(p, ',', (size_t)(end - p)); if (!e) e = end;
        if (n >= max_n) {
            B.h_pos.resize(h0); B.h_chr.resize(h0); B.h_strand.resize(h0); B.h_nm.resize(h0); B.h_len_dif.resize(h0); B.h_cig_n.resize(h0);
            if (B.parse_compact) B.cig8.resize(c0); else { B.h_cig_off.resize(h0); B.cig.resize(c0); }
            return;
        }
        add_hit(B, ix, p, (size_t)(e - p));
        ++n; p = e;
    }
}

void Batch::clear()
{
    reads.clear(); read_off.assign(1, 0); read_seq.clear(); seed_all.clear(); last_len.clear(); seed_off.assign(1, 0); seed_id.clear(); hit_off.assign(1, 0);
    h_pos.clear(); h_chr.clear(); h_strand.clear(); h_nm.clear(); h_len_dif.clear(); h_cig_off.clear(); h_cig_n.clear(); cig.clear(); cig8.clear(); cig_wide = false; saw_wide = false;
}

static const uint8_t *nt4_table()
{   // nst_nt4_table, src/bntseq.c:20 ('-' maps to 5 there; the hot path accepts 0..4, so it is treated as N)
    static uint8_t t[256]; static bool init = false;
    if (!init) { memset(t, 4, 256); t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3; init = true; }
    return t;
}

// one read + its seed_all GEM map lines -> batch (lamsa_read_seq, src/lamsa_aln.c:927-956; split_seed :252-253,281).
// [lines, lines_end): the read's seed_all map lines as they are in the file, each ending in '\n'.
static void append_read_lines(Batch &B, const Index &ix, const lamsa_hp_para &P, const Read &rd, const char *lines, const char *lines_end, int seed_all)
{
    const int L = (int)rd.seq.size();
    const uint8_t *t4 = nt4_table();
    const size_t b0 = B.read_seq.size();
    B.read_seq.resize(b0 + (size_t)L);
    for (int i = 0; i < L; ++i) B.read_seq[b0 + (size_t)i] = t4[(unsigned char)rd.seq[(size_t)i]];
    B.read_off.push_back((int64_t)B.read_seq.size());
    B.seed_all.push_back(seed_all); B.last_len.push_back(L - P.seed_len - (seed_all - 1) * P.seed_step);
    const char *line = lines;
    for (int sd = 0; sd < seed_all && line < lines_end; ++sd) {
        const char *eol = (const char *)memchr(line, '\n', (size_t)(lines_end - line));
        if (!eol) eol = lines_end;
        const char *q = line; int ct = 0;
        for (; q < eol; ++q) if (*q == '\t') { if (ct == 3) break; ct++; }
        if (q < eol && q + 1 < eol && q[1] != '-') {                    // otherwise no map content: the seed gets no slot
            const char *he = eol; if (he > q + 1 && he[-1] == '\r') --he;
            B.seed_id.push_back(sd + 1);
            parse_gem_hits(B, ix, q + 1, he, P.per_aln_m);
            B.hit_off.push_back((int64_t)B.h_pos.size());
        }
        line = eol + 1;
    }
    B.seed_off.push_back((int64_t)B.seed_id.size());
}

static int seeds_of(const lamsa_hp_para &P, int L) { return L < P.seed_len ? 0 : 1 + (L - P.seed_len) / P.seed_step; }

// the seed-result file, mapped (or, where that fails, read) into memory: lines are handed out as spans, never copied
struct MapText {
    const char *p = nullptr; size_t n = 0, pos = 0; bool mapped = false; std::vector<char> owned; Prefault pf;
    // Follow mode (the reference runs the mapper to completion first, src/lamsa_aln.c:1179-1193): the map
    // is still being written by the mapper started by run_seeding; lines are read as they appear, the end of the file counts only once
    // the mapper has exited.  The text of a read is handed out as a string of its own (the chunk keeps it until it is parsed).
    int follow_fd = -1; long follow_pid = 0; int follow_status = 0; bool follow_done = false; std::string fbuf; size_t fpos = 0; std::string follow_path;
    bool following() const { return follow_pid != 0; }
    bool open_follow(const std::string &path, long pid) { follow_path = path; follow_pid = pid; p = ""; return true; }
    // more bytes of the growing file into fbuf; false: the mapper is gone and everything it wrote has been read
    bool follow_more() {
        for (;;) {
            if (follow_fd < 0) follow_fd = ::open(follow_path.c_str(), O_RDONLY);
            if (follow_fd >= 0) {
                if (fpos > (1u << 20)) { fbuf.erase(0, fpos); fpos = 0; }
                const size_t at = fbuf.size();
                fbuf.resize(at + (1u << 20));
                const ssize_t got = ::read(follow_fd, &fbuf[at], 1u << 20);
                fbuf.resize(at + (got > 0 ? (size_t)got : 0));
                if (got > 0) return true;
            }
            if (follow_done) return false;                                    // the mapper had exited before this (empty) read
            int st = 0;
            const pid_t w = waitpid((pid_t)follow_pid, &st, WNOHANG);
            if (w == (pid_t)follow_pid || w < 0) { follow_done = true; follow_status = w < 0 ? -1 : st; continue; }   // one more read picks up what it wrote last
            usleep(2000);
        }
    }
    // the next `lines` lines of the growing map as one string; false when the mapper's output ends first
    bool take_follow(int lines, std::string &out) {
        out.clear();
        size_t scan = fpos;
        for (int left = lines; left > 0; ) {
            const void *nl = scan < fbuf.size() ? memchr(fbuf.data() + scan, '\n', fbuf.size() - scan) : nullptr;
            if (nl) { scan = (size_t)((const char *)nl - fbuf.data()) + 1; --left; continue; }
            const size_t keep = scan - fpos;
            if (!follow_more()) return false;
            scan = fpos + keep;                                                // fbuf may have been compacted: fpos moved with it
        }
        out.assign(fbuf.data() + fpos, scan - fpos);
        fpos = scan;
        return true;
    }
    // after the last read: has the mapper ended well?  (waits for it)
    bool follow_finish() {
        if (!follow_pid) return true;
        if (!follow_done) { int st = 0; follow_status = waitpid((pid_t)follow_pid, &st, 0) < 0 ? -1 : st; follow_done = true; }
        if (follow_fd >= 0) { ::close(follow_fd); follow_fd = -1; }
        return follow_status != -1 && WIFEXITED(follow_status) && WEXITSTATUS(follow_status) == 0;
    }
    bool open(const std::string &path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            n = (size_t)st.st_size;
            if (n == 0) { ::close(fd); p = ""; return true; }
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) { madvise(m, n, MADV_SEQUENTIAL); p = (const char *)m; mapped = true; ::close(fd); pf.start(p, n); return true; }
        }
        char buf[1 << 16]; ssize_t got;                                    // a pipe or a file system without mmap
        while ((got = ::read(fd, buf, sizeof buf)) > 0) owned.insert(owned.end(), buf, buf + got);
        ::close(fd);
        p = owned.data(); n = owned.size();
        return true;
    }
    ~MapText() { pf.finish(); if (mapped) munmap((void *)p, n); if (follow_fd >= 0) ::close(follow_fd); }
    // the next `lines` lines as [a, b); false when the file ends first
    bool take(int lines, const char *&a, const char *&b) {
        a = p + pos;
        int left = lines;
#if defined(__x86_64__)
        static const bool avx2 = __builtin_cpu_supports("avx2");
        if (avx2) pos = skip_lines_avx2(p, pos, n, left);
#endif
        for (; left > 0; --left) {
            if (pos >= n) return false;
            const char *nl = (const char *)memchr(p + pos, '\n', n - pos);
            pos = nl ? (size_t)(nl - p) + 1 : n;
        }
        b = p + pos;
        return true;
    }
#if defined(__x86_64__)
    // advance over whole 32-byte blocks while they hold fewer newlines than are still wanted; `left` is updated.  The
    // lines are ~200 bytes, so counting newlines a block at a time beats one memchr call per line.
    __attribute__((target("avx2"))) static size_t skip_lines_avx2(const char *p, size_t pos, size_t n, int &left) {
        const __m256i nl = _mm256_set1_epi8('\n');
        while (left > 0 && pos + 32 <= n) {
            const unsigned m = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + pos)), nl));
            const int c = __builtin_popcount(m);
            if (c < left) { left -= c; pos += 32; continue; }
            unsigned mm = m;                                               // the left-th newline is inside this block
            for (int i = 1; i < left; ++i) mm &= mm - 1;
            pos += (size_t)__builtin_ctz(mm) + 1; left = 0;
        }
        return pos;
    }
#endif
};

template <class F> static void parallel_blocks(int n, int threads, F fn)
{
    if (threads < 2 || n < 2 * threads) { fn(0, 0, n); return; }
    std::vector<std::thread> th;
    const int per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) { const int a = t * per, b = std::min(n, a + per); if (a < b) th.emplace_back([=]() { fn(t, a, b); }); }
    for (auto &x : th) x.join();
}

// concatenate per-thread partial batches (offsets become absolute); every thread copies its own part
static void merge_batches(Batch &B, std::vector<Batch> &parts, int threads)
{
    const size_t np = parts.size();
    std::vector<size_t> r0(np + 1, 0), b0(np + 1, 0), s0(np + 1, 0), h0(np + 1, 0), c0(np + 1, 0);
    for (size_t i = 0; i < np; ++i) {
        const Batch &p = parts[i];
        r0[i + 1] = r0[i] + p.seed_all.size(); b0[i + 1] = b0[i] + p.read_seq.size(); s0[i + 1] = s0[i] + p.seed_id.size();
        h0[i + 1] = h0[i] + p.h_pos.size(); c0[i + 1] = c0[i] + (p.parse_compact ? p.cig8.size() : p.cig.size());
    }
    const bool compact_parts = np > 0 && parts[0].parse_compact;      // the parts hold the seed CIGARs as bytes already (none of them met an element that does not fit)
    B.read_off.resize(r0[np] + 1); B.seed_off.resize(r0[np] + 1); B.seed_all.resize(r0[np]); B.last_len.resize(r0[np]); B.read_seq.resize(b0[np]);
    B.seed_id.resize(s0[np]); B.hit_off.resize(s0[np] + 1);
    B.h_pos.resize(h0[np]); B.h_chr.resize(h0[np]); B.h_strand.resize(h0[np]); B.h_nm.resize(h0[np]); B.h_len_dif.resize(h0[np]); B.h_cig_n.resize(h0[np]);
    B.cig8.resize(c0[np]); B.cig.clear(); B.h_cig_off.clear();      // the word form of the CIGARs (4 B per element + 4 B per hit) is only made when an element does not fit a byte
    std::atomic<int> wide(0);
    B.read_off[0] = 0; B.seed_off[0] = 0; B.hit_off[0] = 0;
    auto copy_parts = [&](int i0, int i1) {
        for (int i = i0; i < i1; ++i) {
            Batch &p = parts[(size_t)i];
            for (size_t k = 1; k < p.read_off.size(); ++k) B.read_off[r0[i] + k] = p.read_off[k] + (int64_t)b0[i];
            for (size_t k = 1; k < p.seed_off.size(); ++k) B.seed_off[r0[i] + k] = p.seed_off[k] + (int64_t)s0[i];
            for (size_t k = 1; k < p.hit_off.size(); ++k) B.hit_off[s0[i] + k] = p.hit_off[k] + (int64_t)h0[i];
            std::copy(p.read_seq.begin(), p.read_seq.end(), B.read_seq.begin() + (long)b0[i]);
            std::copy(p.seed_all.begin(), p.seed_all.end(), B.seed_all.begin() + (long)r0[i]);
            std::copy(p.last_len.begin(), p.last_len.end(), B.last_len.begin() + (long)r0[i]);
            std::copy(p.seed_id.begin(), p.seed_id.end(), B.seed_id.begin() + (long)s0[i]);
            std::copy(p.h_pos.begin(), p.h_pos.end(), B.h_pos.begin() + (long)h0[i]);
            std::copy(p.h_chr.begin(), p.h_chr.end(), B.h_chr.begin() + (long)h0[i]);
            std::copy(p.h_strand.begin(), p.h_strand.end(), B.h_strand.begin() + (long)h0[i]);
            std::copy(p.h_nm.begin(), p.h_nm.end(), B.h_nm.begin() + (long)h0[i]);
            std::copy(p.h_len_dif.begin(), p.h_len_dif.end(), B.h_len_dif.begin() + (long)h0[i]);
            std::copy(p.h_cig_n.begin(), p.h_cig_n.end(), B.h_cig_n.begin() + (long)h0[i]);
            {   // the same CIGARs one byte per element (they lie back to back in hit order: no offsets needed on the way to the GPU)
                int w = 0;
                uint8_t *o8 = B.cig8.data() + c0[i];
                if (compact_parts) std::copy(p.cig8.begin(), p.cig8.end(), o8);
                else for (size_t k = 0; k < p.cig.size(); ++k) { const int32_t x = p.cig[k]; w |= (x >> 4) > 63 || (x & 0xf) > 2; o8[k] = (uint8_t)(((x & 3) << 6) | ((x >> 4) & 63)); }
                if (w) wide = 1;
            }
        }
    };
    auto copy_words = [&](int i0, int i1) {                   // second pass, rare: some element is longer than 63
        for (int i = i0; i < i1; ++i) {
            const Batch &p = parts[(size_t)i];
            for (size_t k = 0; k < p.h_cig_off.size(); ++k) B.h_cig_off[h0[i] + k] = (int32_t)(p.h_cig_off[k] + (int64_t)c0[i]);
            std::copy(p.cig.begin(), p.cig.end(), B.cig.begin() + (long)c0[i]);
        }
    };
    auto on_parts = [&](const std::function<void(int, int)> &fn) {
        if (threads < 2 || h0[np] < 4096) { fn(0, (int)np); return; }
        std::vector<std::thread> th;                         // one thread per part
        for (size_t i = 0; i < np; ++i) if (!parts[i].seed_all.empty()) th.emplace_back(fn, (int)i, (int)i + 1);
        for (auto &x : th) x.join();
    };
    on_parts(copy_parts);
    B.cig_wide = wide.load() != 0 || !compact_parts;          // (parts in words: the parse has met an element that does not fit a byte)
    if (B.cig_wide) { B.cig.resize(c0[np]); B.h_cig_off.resize(h0[np]); on_parts(copy_words); }
    for (Batch &p : parts) p.clear();
}

// ------------------------------------------------------------------ result stream -> records
void parse_stream(const int32_t *s, int n_words, int read_len, ReadResult &R)
{
    // (R may hold an earlier read's result: its lines, records and CIGAR vectors are reused, not freed and allocated again)
    R.stage[2].clear();
    R.status = n_words > 0 ? s[0] : LAMSA_HP_ST_OVERFLOW;
    if (n_words < 3 || R.status != 0) { R.stage[0].clear(); R.stage[1].clear(); return; }
    int i = 3;
    for (int st = 0; st < 2; ++st) {
        R.stage[st].resize((size_t)s[1 + st]);
        for (Line &ln : R.stage[st]) {
            ln.line_score = s[i]; ln.tol_score = s[i + 1]; ln.tol_NM = s[i + 2]; ln.merg_x = ln.merg_y = 0; ln.mapQ = 0; ln.xa.clear();
            const int n_res = s[i + 3]; i += 4;
            ln.rec.resize((size_t)n_res);
            for (Rec &r : ln.rec) {
                r.offset = (int64_t)(((uint64_t)(uint32_t)s[i + 1] << 32) | (uint32_t)s[i]); r.chr = s[i + 2]; r.nstrand = s[i + 3]; r.score = s[i + 4]; r.NM = s[i + 5];
                const int cn = s[i + 6]; i += 7;
                r.cigar.assign(s + i, s + i + cn); i += cn;
                // covered read interval, push_reg_res src/lamsa_aln.c:571-595
                if (r.cigar.empty()) { r.reg_beg = 1; r.reg_end = read_len; continue; }
                const int32_t c0 = r.cigar.front(), c1 = r.cigar.back();
                if (r.nstrand == 1) { r.reg_beg = (c0 & 0xf) == 4 ? (c0 >> 4) + 1 : 1; r.reg_end = (c1 & 0xf) == 4 ? read_len - (c1 >> 4) : read_len; }
                else { r.reg_beg = (c1 & 0xf) == 4 ? (c1 >> 4) + 1 : 1; r.reg_end = (c0 & 0xf) == 4 ? read_len - (c0 >> 4) : read_len; }
            }
        }
    }
}

// a stable sort that allocates nothing for the handful of elements a read has (std::stable_sort asks for a buffer every time)
template <class It, class Less> static inline void small_stable_sort(It b, It e, Less less)
{
    if (e - b > 24) { std::stable_sort(b, e, less); return; }
    for (It i = b + (b != e); i < e; ++i) { auto v = *i; It j = i; while (j > b && less(v, *(j - 1))) { *j = *(j - 1); --j; } *j = v; }
}

// get_cov_f, src/lamsa_aln.c:639-651
static float cov_fraction(const ReadResult &R, int read_len)
{
    std::vector<std::pair<int, int>> reg;
    for (int st = 0; st < 3; ++st) for (const Line &ln : R.stage[st]) { if (ln.tol_score < 0) continue; for (const Rec &r : ln.rec) reg.push_back({r.reg_beg, r.reg_end}); }
    if (reg.empty()) return (float)(0.0 / read_len);
    small_stable_sort(reg.begin(), reg.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
    int cov = 0; size_t cur = 0;
    for (size_t i = 1; i < reg.size(); ++i) {                              // aln_merg_reg with thd 0, :499
        if (reg[i].first - reg[cur].second - 1 < 0) { if (reg[i].second > reg[cur].second) reg[cur].second = reg[i].second; }
        else { ++cur; reg[cur] = reg[i]; }
    }
    for (size_t i = 0; i <= cur; ++i) cov += reg[i].second - reg[i].first + 1;
    return (float)((cov + 0.0) / read_len);
}

static float cover_rate(int s1, int e1, int s2, int e2)
{   // src/lamsa_dp_con.c:61-67
    const int s = s2 > s1 ? s2 : s1, e = e2 < e1 ? e2 : e1;
    const float rat1 = (float)((e - s + 1 + 0.0) / (e1 - s1 + 1 + 0.0)), rat2 = (float)((e - s + 1 + 0.0) / (e2 - s2 + 1 + 0.0));
    return rat1 > rat2 ? rat1 : rat2;
}

// rearr_aln_res, src/lamsa_aln.c:654-724
void rank_results(ReadResult &R, int read_len, const lamsa_hp_para &P)
{
    struct Q { int st, li, a, b; };
    std::vector<Q> qua;
    for (int st = 0; st < 3; ++st)
        for (size_t i = 0; i < R.stage[st].size(); ++i) {
            Line &l = R.stage[st][i];
            l.xa.clear(); l.mapQ = 0;
            if (l.tol_score < 0) { l.merg_x = 0; l.merg_y = -1; continue; }
            qua.push_back({st, (int)i, l.tol_score, l.line_score});
        }
    if (qua.empty()) return;
    small_stable_sort(qua.begin(), qua.end(), [](const Q &x, const Q &y) { return x.a != y.a ? x.a > y.a : x.b > y.b; });   // res_comp :632 on glibc's stable qsort
    auto L = [&](int i) -> Line & { return R.stage[qua[i].st][qua[i].li]; };
    std::vector<int> head;
    std::vector<std::pair<int, int>> reg;                                     // intervals of the accepted heads, in acceptance order
    for (const Rec &r : L(0).rec) reg.push_back({r.reg_beg, r.reg_end});
    head.push_back(0);
    L(0).merg_x = 1; L(0).merg_y = 0;
    for (size_t i = 0; i < qua.size(); ++i) L((int)i).mapQ = 255;
    const float cov_f = cov_fraction(R, read_len);
    const int mapq_max = (int)(254 * cov_f);
    for (int i = 1; i < (int)qua.size(); ++i) {
        int cov_qi = -1;
        for (const Rec &nr : L(i).rec) {                                      // get_cover_res :607-629
            size_t reg_i = 0;
            for (int hi : head) {
                for (size_t j = 0; j < L(hi).rec.size(); ++j, ++reg_i)
                    if (cover_rate(reg[reg_i].first, reg[reg_i].second, nr.reg_beg, nr.reg_end) >= P.ovlp_rat) { cov_qi = hi; break; }
                if (cov_qi >= 0) break;
            }
            if (cov_qi >= 0) break;
        }
        if (cov_qi < 0) {
            for (const Rec &r : L(i).rec) reg.push_back({r.reg_beg, r.reg_end});
            head.push_back(i);
            L(i).merg_x = 1; L(i).merg_y = 0;
        } else if (qua[i].a > qua[cov_qi].a / 2 && (int)L(cov_qi).xa.size() + (int)L(i).rec.size() - 1 < P.res_mul_max) {
            for (size_t j = 0; j < L(i).rec.size(); ++j) L(cov_qi).xa.push_back({qua[i].st, qua[i].li, (int)j});
            L(cov_qi).merg_y = 1;
            const uint8_t tmpQ = (uint8_t)(mapq_max * (qua[cov_qi].a - qua[i].a) / qua[cov_qi].a);
            if (tmpQ < L(cov_qi).mapQ) L(cov_qi).mapQ = tmpQ;
            L(i).merg_x = 2; L(i).merg_y = 0;
        } else { L(i).merg_x = 0; L(i).merg_y = -1; }
    }
    for (size_t i = 0; i < qua.size(); ++i) {
        Line &l = L((int)i);
        if (l.merg_x != 1) continue;
        if (l.mapQ == 255) l.mapQ = (uint8_t)(mapq_max / (int)head.size()); else l.mapQ = (uint8_t)(l.mapQ / (int)head.size());
    }
}

// ------------------------------------------------------------------ SAM
static char comp_char(char c)
{
    switch (c) { case 'A': return 'T'; case 'a': return 't'; case 'C': return 'G'; case 'c': return 'g';
                 case 'G': return 'C'; case 'g': return 'c'; case 'T': return 'A'; case 't': return 'a'; default: return c; }
}
static void appendf(std::string &o, const char *fmt, ...)
{
    char tmp[256]; va_list ap; va_start(ap, fmt); int n = vsnprintf(tmp, sizeof tmp, fmt, ap); va_end(ap);
    if (n < (int)sizeof tmp) { o.append(tmp, (size_t)n); return; }
    std::vector<char> big((size_t)n + 1); va_start(ap, fmt); vsnprintf(big.data(), big.size(), fmt, ap); va_end(ap); o.append(big.data(), (size_t)n);
}

void sam_header(std::string &o, const Index &ix, const std::string &pg)
{   // print_sam_header, src/lamsa_aln.c:1215
    for (size_t i = 0; i < ix.name.size(); ++i) appendf(o, "@SQ\tSN:%s\tLN:%d\n", ix.name[i].c_str(), ix.len[i]);
    o += pg; o += "\n";
}

// aln_res_output, src/lamsa_aln.c:1001-1100.  (`-C` with a reverse-strand FASTQ read never terminates in the
// reference, :1043; here QUAL is printed reversed -- the evident intent, documented divergence.)
// decimal digits of v appended to o (the CIGAR of a 10 kbp noisy read has ~1 500 operations: no printf per operation)
static inline void append_int(std::string &o, long long v)
{
    char buf[24]; int n = 0;
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { buf[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) buf[n++] = '-';
    while (n) o.push_back(buf[--n]);
}
// The CIGAR of a 10-kbp noisy read has ~1 500 operations, nearly all of one or two digits: they are formatted into a block of the string's
// own storage through a pointer (no per-character capacity check), lengths below 100 from a table of digit pairs.
static inline char *put_len(char *p, unsigned v)
{
    static const char D2[] = "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
    if (v < 10) { *p++ = (char)('0' + v); return p; }
    if (v < 100) { p[0] = D2[2 * v]; p[1] = D2[2 * v + 1]; return p + 2; }
    char buf[12]; int n = 0;
    while (v >= 100) { const unsigned q = v / 100, r = v - 100 * q; buf[n++] = D2[2 * r + 1]; buf[n++] = D2[2 * r]; v = q; }
    if (v >= 10) { buf[n++] = D2[2 * v + 1]; buf[n++] = D2[2 * v]; } else buf[n++] = (char)('0' + v);
    while (n) *p++ = buf[--n];
    return p;
}
static inline void append_cigar(std::string &o, const std::vector<int32_t> &cig, const char *ops)
{
    const size_t k0 = o.size();
    o.resize(k0 + 11 * cig.size());                          // at most ten digits and the operation
    char *p = &o[k0];
    for (int32_t w : cig) { p = put_len(p, (unsigned)(w >> 4)); *p++ = ops[w & 0xf]; }
    o.resize((size_t)(p - o.data()));
}

void write_sam(std::string &o, const ReadResult &R, const Read &rd, const Index &ix, const Options &opt)
{
    static const char OPS[] = "MIDNSHP=XB", OPS_HC[] = "MIDNHHP=XB";
    const int read_len = (int)rd.seq.size();
    const bool with_qual = rd.has_qual && opt.comm;
    int all = 0; bool prim = false;
    if (o.capacity() - o.size() < 4 * (size_t)read_len + 4096) o.reserve(o.size() + std::max<size_t>(o.size() / 2, 8 * (size_t)read_len + 65536));      // grown in large steps, not per append
    for (int st = 0; st < 3; ++st)
        for (const Line &la : R.stage[st]) {
            if (la.merg_x != 1) continue;
            for (size_t j = 0; j < la.rec.size(); ++j) {
                const Rec &r = la.rec[j];
                ++all;
                int flag = r.nstrand ? 0 : 0x10;
                const bool soft = !prim || opt.supp_soft;
                if (!soft) flag |= 0x800;
                o += rd.name; o.push_back('\t'); append_int(o, flag); o.push_back('\t'); o += ix.name[(size_t)r.chr - 1]; o.push_back('\t');
                append_int(o, (long long)r.offset); o.push_back('\t'); append_int(o, (int)la.mapQ); o.push_back('\t');
                append_cigar(o, r.cigar, soft ? OPS : OPS_HC);
                o += "\t*\t0\t0\t";
                const int b = soft ? 0 : r.reg_beg - 1, e = soft ? read_len : r.reg_end;
                if (r.nstrand == 1) o.append(rd.seq, (size_t)b, (size_t)(e - b));
                else { const size_t k0 = o.size(); o.resize(k0 + (size_t)(e - b)); char *d = &o[k0]; for (int si = e - 1; si >= b; --si) *d++ = comp_char(rd.seq[(size_t)si]); }
                o.push_back('\t');
                if (with_qual) { if (r.nstrand == 1) o.append(rd.qual, (size_t)b, (size_t)(e - b)); else for (int si = e - 1; si >= b; --si) o.push_back(rd.qual[(size_t)si]); }
                else o.push_back('*');
                if (soft) prim = true;
                o += "\tNM:i:"; append_int(o, r.NM); o += "\tAS:i:"; append_int(o, r.score);
                if (j == 0 && !la.xa.empty()) {
                    o += "\tXA:Z:";
                    for (const XaRef &x : la.xa) {
                        const Rec &xr = R.stage[x.st][(size_t)x.li].rec[(size_t)x.ri];
                        o += ix.name[(size_t)xr.chr - 1]; o.push_back(','); o.push_back("-+"[xr.nstrand]); append_int(o, (long long)xr.offset); o.push_back(',');
                        append_cigar(o, xr.cigar, OPS);
                        o.push_back(','); append_int(o, xr.NM); o.push_back(';');
                    }
                }
                o.push_back('\n');
            }
        }
    if (all == 0) {
        appendf(o, "%s\t%d\t*\t%lld\t%d\t*\t*\t0\t0\t", rd.name.c_str(), 4, 0LL, 0);
        o += rd.seq; o.push_back('\t');
        if (rd.has_qual) o += rd.qual; else o.push_back('*');
        o.push_back('\n');
    }
}

// ------------------------------------------------------------------ seeding front end
// split_seed (src/lamsa_aln.c:223-303) + lamsa_gem (:1179-1193, gem/gem_map.sh): the read file is cut into overlapping
// seeds (<reads>.seed, names "<read>_<i>:<offset>"; <reads>.seed.info as the reference writes it) and the GEM mapper is
// run on them with the reference's arguments; its output <reads>.seed.gem.map is what run_aln reads.  The mapper and
// the <ref>.gem index belong to the reference's bundle (`lamsa index`); nothing of them is part of this repository.
// Cuts the seeds as split_seed does (src/lamsa_aln.c:225-303), then starts the bundle's gem-mapper with the arguments of gem/gem_map.sh.
// pid == nullptr: waits for it, as the reference does (lamsa_gem, :1179-1193).  Otherwise the mapper is left running and its process id
// returned: run_aln reads the map while it is being written, so seeding overlaps parsing and the GPU (it must be started before any GPU
// call of this process: a process that has initialised the GPU must not fork + exec on this pool).
int run_seeding(const Options &opt, const lamsa_hp_para &P, long *pid)
{
    FastxReader fx;
    if (!fx.open(opt.reads)) { fprintf(stderr, "[lamsa_aln] Can't open read file %s\n", opt.reads.c_str()); return 1; }
    const std::string seed_f = opt.reads + ".seed", info_f = opt.reads + ".seed.info";
    FILE *sf = fopen(seed_f.c_str(), "w"), *inf = fopen(info_f.c_str(), "w");
    if (!sf || !inf) { fprintf(stderr, "[lamsa_aln] Can't open seed file %s\n", sf ? info_f.c_str() : seed_f.c_str()); if (sf) fclose(sf); if (inf) fclose(inf); return 1; }
    fprintf(stderr, "[lamsa_aln] Generating seed ... ");
    Read rd; std::string o;
    while (fx.next(rd)) {
        const int L = (int)rd.seq.size(), seed_all = seeds_of(P, L);
        o.clear();
        for (int i = 0; i < seed_all; ++i) {
            o += ">"; o += rd.name; o += "_"; o += std::to_string(i); o += ":"; o += std::to_string(i * P.seed_step); o += "\n";
            o.append(rd.seq, (size_t)i * P.seed_step, (size_t)P.seed_len); o += "\n";
        }
        fwrite(o.data(), 1, o.size(), sf);
        fprintf(inf, "%s %d %d %d\n", rd.name.c_str(), seed_all, L - P.seed_len - (seed_all - 1) * P.seed_step, L);
    }
    fclose(sf); fclose(inf);
    fprintf(stderr, "done!\n");
    const int t = P.read_type;
    const float ed = opt.ed_rate >= 0 ? opt.ed_rate : (t == 1 ? 0.3f : (t == 2 ? 0.25f : 0.04f));
    const float mis = opt.mis_rate >= 0 ? opt.mis_rate : (t == 2 ? 0.06f : 0.04f);
    const float mat = opt.mat_rate >= 0 ? opt.mat_rate : (t == 1 ? 0.7f : (t == 2 ? 0.6f : 0.80f));
    const std::string mapper = opt.gem_dir + "/gem-mapper", idx = opt.ref_prefix + ".gem", outp = seed_f + ".gem";
    FILE *probe = fopen(mapper.c_str(), "r");
    if (!probe) { fprintf(stderr, "[lamsa_aln] GEM mapper not found at %s (it ships with the reference; give its directory with --gem-dir, or run with -N on an existing %s.gem.map)\n", mapper.c_str(), seed_f.c_str()); return 1; }
    fclose(probe);
    char num[256];
    snprintf(num, sizeof num, " -m %f -e %f --min-matched-bases %f --max-big-indel-length 3 -d %d -D 0 -T %d %s", mis, ed, mat, P.per_aln_m, opt.n_thread > 0 ? opt.n_thread : 1,
             opt.fastest ? "--fast-mapping=0" : "--fast-mapping");
    ::remove((outp + ".map").c_str());                   // a stale map of an earlier run must not be taken for this one's
    std::vector<std::string> av = {mapper, "-I", idx, "-i", seed_f, "-o", outp};
    { std::istringstream is(num); std::string w; while (is >> w) av.push_back(w); }
    fprintf(stderr, "[lamsa_aln] Executing gem-mapper ... \n");
    fflush(stderr);
    const pid_t child = fork();
    if (child < 0) { fprintf(stderr, "[lamsa_aln] Seeding undone, cannot start gem-mapper.\n"); return 1; }
    if (child == 0) {
        const int lf = ::open((outp + ".log").c_str(), O_WRONLY | O_CREAT | O_APPEND, 0644);
        if (lf >= 0) { dup2(lf, 2); ::close(lf); }
        std::vector<char *> argv;
        for (std::string &a : av) argv.push_back(&a[0]);
        argv.push_back(nullptr);
        execv(mapper.c_str(), argv.data());
        _exit(127);
    }
    if (pid) { *pid = (long)child; return 0; }
    int st = 0;
    if (waitpid(child, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) { fprintf(stderr, "[lamsa_aln] Seeding undone, gem-mapper exit abnormally.\n"); return 1; }
    fprintf(stderr, "[lamsa_aln] gem-mapper done!\n");
    return 0;
}

// ------------------------------------------------------------------ binary hit stream
// The arrays of lamsa_hp_batch, chunk after chunk, as they go to the GPU: a re-run on the same reads (other scoring /
// filtering options) maps the file and uploads straight from the page cache -- no seeding, no text, no parse.  Native
// byte order; the header pins the options that shape the arrays (-T, -l, -i, -p).
namespace {
const char HITS_MAGIC[8] = {'L', 'A', 'M', 'S', 'A', 'H', 'P', '2'};
struct HitsHeader { char magic[8]; int32_t read_type, seed_len, seed_step, per_aln_m; int64_t reserved[2]; };
struct HitsChunkHeader { int64_t n_reads, n_bases, n_slots, n_hits, n_cig, bytes, cig_elem; };   // bytes: of the arrays that follow (each padded to 64);
                                                                                              // cig_elem 1: cig8, no offsets; 4: cig words + 32-bit offsets
inline size_t pad64(size_t x) { return (x + 63) & ~(size_t)63; }

struct HitsWriter {
    FILE *fp = nullptr;
    bool open(const std::string &path, const lamsa_hp_para &P) {
        fp = fopen(path.c_str(), "wb");
        if (!fp) return false;
        HitsHeader h; memset(&h, 0, sizeof h); memcpy(h.magic, HITS_MAGIC, 8);
        h.read_type = P.read_type; h.seed_len = P.seed_len; h.seed_step = P.seed_step; h.per_aln_m = P.per_aln_m;
        return fwrite(&h, sizeof h, 1, fp) == 1;
    }
    template <class V> bool put(const V &v, size_t n) {
        static const char zeros[64] = {0};
        const size_t bytes = n * sizeof(v[0]);
        if (bytes && fwrite(v.data(), 1, bytes, fp) != bytes) return false;
        const size_t padn = pad64(bytes) - bytes;
        return padn == 0 || fwrite(zeros, 1, padn, fp) == padn;
    }
    bool write(const lamsa::Batch &B) {
        const size_t n = B.reads.size(), nb = B.read_seq.size(), ns = B.seed_id.size(), nh = B.h_pos.size(), nc = B.cig8.size();
        HitsChunkHeader c; c.n_reads = (int64_t)n; c.n_bases = (int64_t)nb; c.n_slots = (int64_t)ns; c.n_hits = (int64_t)nh; c.n_cig = (int64_t)nc; c.cig_elem = B.cig_wide ? 4 : 1;
        c.bytes = (int64_t)(pad64(8 * (n + 1)) * 2 + pad64(nb) + pad64(4 * n) * 2 + pad64(4 * ns) + pad64(8 * (ns + 1)) + pad64(8 * nh) + pad64(4 * nh) + pad64(nh) * 2 + pad64(2 * nh) * 2 +
                            (B.cig_wide ? pad64(4 * nh) + pad64(4 * nc) : pad64(nc)));
        if (fwrite(&c, sizeof c, 1, fp) != 1) return false;
        return put(B.read_off, n + 1) && put(B.read_seq, nb) && put(B.seed_all, n) && put(B.last_len, n) && put(B.seed_off, n + 1) && put(B.seed_id, ns) && put(B.hit_off, ns + 1) &&
               put(B.h_pos, nh) && put(B.h_chr, nh) && put(B.h_strand, nh) && put(B.h_nm, nh) && put(B.h_len_dif, nh) && put(B.h_cig_n, nh) &&
               (B.cig_wide ? put(B.h_cig_off, nh) && put(B.cig, nc) : put(B.cig8, nc));
    }
    void close() { if (fp) fclose(fp); fp = nullptr; }
};
}  // namespace

// ------------------------------------------------------------------ chunk loop
static double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int run_aln(const Options &opt, const lamsa_hp_para &P, FILE *out, const std::string &pg_line, Stats *stats)
{
    const double t_begin = now_s();
    double parse_s = 0, submit_s = 0, wait_s = 0, sam_s = 0;
    long n_mapped_chunks = 0;                                // chunks whose SAM text the threads wrote side by side
    // Every way out of this function before the map has been read to its end -- an index that does not load, a map that does not match
    // the reads, a failed submit -- must not leave the mapper started by run_seeding running (and writing, on -t cores): it is told to
    // stop and waited for, and its exit status reported.
    struct MapperGuard { long pid = 0; MapText *m = nullptr; ~MapperGuard() {
        if (!pid || (m && m->follow_done)) return;
        kill((pid_t)pid, SIGTERM);
        int st = 0; const pid_t w = waitpid((pid_t)pid, &st, 0);
        if (m) m->follow_done = true;
        if (w > 0 && WIFEXITED(st)) fprintf(stderr, "[lamsa_aln] gem-mapper stopped (exit status %d)\n", WEXITSTATUS(st));
        else fprintf(stderr, "[lamsa_aln] gem-mapper stopped (signal %d)\n", w > 0 && WIFSIGNALED(st) ? WTERMSIG(st) : 0);
    } };
    MapText mapt, hitsf;                                // GEM map text, or (--hits) the binary hit stream written by --save-hits (declared before the guard that ends the mapper: destroyed after it)
    MapperGuard orphan; orphan.pid = opt.mapper_pid; orphan.m = &mapt;
    Index ix; std::string err;
    if (!load_index(opt.ref_prefix, ix, err)) { fprintf(stderr, "[lamsa_aln] %s\n", err.c_str()); return 1; }
    const std::string map_path = opt.seed_result.empty() ? opt.reads + ".seed.gem.map" : opt.seed_result;
    const bool from_hits = !opt.hits.empty();
    if (from_hits) {
        HitsHeader hh;
        if (!hitsf.open(opt.hits) || hitsf.n < sizeof hh) { fprintf(stderr, "[lamsa_aln] Can't open hit stream %s\n", opt.hits.c_str()); return 1; }
        memcpy(&hh, hitsf.p, sizeof hh);
        if (memcmp(hh.magic, HITS_MAGIC, 8) != 0 || hh.read_type != P.read_type || hh.seed_len != P.seed_len || hh.seed_step != P.seed_step || hh.per_aln_m != P.per_aln_m) {
            fprintf(stderr, "[lamsa_aln] %s is not a hit stream written with these seeding options (-T, -l, -i, -p)\n", opt.hits.c_str()); return 1;
        }
        hitsf.pos = sizeof hh;
    } else if (opt.mapper_pid) { mapt.open_follow(map_path, opt.mapper_pid);
    } else if (!mapt.open(map_path)) { fprintf(stderr, "[lamsa_aln] Can't open seed-result file %s (seeding is not run by this build: provide the GEM map, as with the reference's -N)\n", map_path.c_str()); return 1; }
    FastxReader fx;
    if (!fx.open(opt.reads)) { fprintf(stderr, "[lamsa_aln] Can't open read file %s\n", opt.reads.c_str()); return 1; }
    // ---- --shard i/N (the read stream shards naturally; one process -- one parser -- per GPU).  This process
    // takes the i-th of N contiguous parts: a mapped read file is cut by bytes (no pass over the parts before it: the first read's
    // name finds its lines in the GEM map, whose lines begin "<read name>_<seed>:"), a compressed one by record count, a hit stream
    // by chunks.  Shards 0 .. N-1 written one after the other are the unsharded output; only shard 0 writes the header.
    long reads_left = -1, chunks_left = -1;                 // -1: no limit
    if (opt.shard_n > 1) {
        const int si = opt.shard_i, sn = opt.shard_n;
        if (opt.mapper_pid) { fprintf(stderr, "[lamsa_aln] --shard needs an existing seed result (-N or --hits): the shards would each run the seeding\n"); return 1; }
        if (from_hits) {
            std::vector<int64_t> n_in; std::vector<size_t> at;
            for (size_t q = hitsf.pos; q + sizeof(HitsChunkHeader) <= hitsf.n; ) {
                HitsChunkHeader ch; memcpy(&ch, hitsf.p + q, sizeof ch);
                if (ch.n_reads <= 0 || ch.bytes < 0 || q + sizeof ch + (size_t)ch.bytes > hitsf.n) { fprintf(stderr, "[lamsa_read_seq] damaged hit stream\n"); return 1; }
                n_in.push_back(ch.n_reads); at.push_back(q); q += sizeof ch + (size_t)ch.bytes;
            }
            const size_t C = n_in.size(), c0 = C * (size_t)si / (size_t)sn, c1 = C * (size_t)(si + 1) / (size_t)sn;
            int64_t skip = 0;
            for (size_t c = 0; c < c0; ++c) skip += n_in[c];
            Read d;
            for (int64_t k = 0; k < skip; ++k) if (!fx.next(d)) { fprintf(stderr, "[lamsa_read_seq] the hit stream does not match the reads\n"); return 1; }
            hitsf.pos = c0 < C ? at[c0] : hitsf.n;
            chunks_left = (long)(c1 - c0);
        } else if (fx.mapped()) {
            const size_t S = fx.size(), lo = S / (size_t)sn * (size_t)si, hi = si + 1 == sn ? S : S / (size_t)sn * (size_t)(si + 1);
            fx.restrict(lo, hi);
            if (si > 0) {
                FastxReader probe; Read first;
                if (probe.open(opt.reads)) probe.restrict(lo, hi);
                // the first read of the part that has seeds at all (a read shorter than a seed has no line in the map)
                bool have_first = probe.next(first);
                while (have_first && seeds_of(P, (int)first.seq.size()) == 0) have_first = probe.next(first);
                if (!have_first) mapt.pos = mapt.n;              // no record with seeds starts in this part: no map line is taken
                else {
                    std::string nm = first.name.substr(0, first.name.find_first_of(" \t"));
                    // the lines of a read's seeds are consecutive and begin "<read name>_<seed>:" (split_seed, src/lamsa_aln.c:287): the first
                    // of them at or after a guess of where this part begins, then back over the lines of the same read before it.  A line of
                    // ANOTHER read whose name merely begins with "<name>_" ("r1_2_0:0" for the read "r1") is told apart by what follows the
                    // key: digits, then a colon (or the end of the name).
                    const std::string key = "\n" + nm + "_";
                    auto is_seed_line = [&](const char *q) {       // q: just behind "<name>_"
                        const char *e = mapt.p + mapt.n, *d = q;
                        while (d < e && *d >= '0' && *d <= '9') ++d;
                        return d > q && d < e && (*d == ':' || *d == '\t' || *d == ' ');       // "<name>_<seed>:<offset>" as split_seed writes it; a mapper may cut the name at the colon
                    };
                    auto find_from = [&](size_t from) -> const char * {
                        while (from < mapt.n) {
                            const char *h = (const char *)memmem(mapt.p + from, mapt.n - from, key.data(), key.size());
                            if (!h) return nullptr;
                            if (is_seed_line(h + key.size())) return h;
                            from = (size_t)(h - mapt.p) + 1;
                        }
                        return nullptr;
                    };
                    const size_t est = (size_t)((double)mapt.n * ((double)lo / (double)(S ? S : 1)));
                    const size_t from = est > (mapt.n >> 3) ? est - (mapt.n >> 3) : 0;
                    const char *hit = find_from(from);
                    if (!hit && from) hit = find_from(0);
                    if (!hit && mapt.n >= key.size() - 1 && memcmp(mapt.p, key.data() + 1, key.size() - 1) == 0 && is_seed_line(mapt.p + key.size() - 1)) hit = mapt.p - 1;      // the very first line
                    if (!hit) { fprintf(stderr, "[lamsa_aln] --shard %d/%d: no line of %s begins with %s_<seed>: (the seeds of the shard's first read)\n", si, sn, map_path.c_str(), nm.c_str()); return 1; }
                    size_t at = (size_t)(hit + 1 - mapt.p);
                    while (at > 1) {                            // the line before `at`: does it belong to the same read?
                        const char *pe = mapt.p + at - 1;       // its newline
                        const char *ps = (const char *)memrchr(mapt.p, '\n', (size_t)(pe - mapt.p));
                        const size_t b = ps ? (size_t)(ps - mapt.p) + 1 : 0;
                        if (at - 1 - b >= key.size() - 1 && memcmp(mapt.p + b, key.data() + 1, key.size() - 1) == 0 && is_seed_line(mapt.p + b + key.size() - 1)) at = b; else break;
                    }
                    mapt.pos = at;
                }
            }
        } else {                                            // compressed reads: count the records, then skip to this shard's first
            long total = 0; { FastxReader cnt; Read d; if (cnt.open(opt.reads)) while (cnt.next(d)) ++total; }
            const long r0 = total * si / sn, r1 = total * (si + 1) / sn;
            Read d; const char *la, *lb;
            for (long k = 0; k < r0; ++k) {
                if (!fx.next(d) || !mapt.take(seeds_of(P, (int)d.seq.size()), la, lb)) { fprintf(stderr, "[lamsa_read_seq] seeds' GEM map result does not match the reads\n"); return 1; }
            }
            reads_left = r1 - r0;
        }
    }
    lamsa_hp_ref ref; ref.pac = ix.pac.data(); ref.l_pac = ix.l_pac; ref.n_seqs = (int32_t)ix.name.size(); ref.seq_offset = ix.off.data(); ref.seq_len = ix.len.data();
    // one handle per GPU: the read stream is dealt out chunk by chunk, the reference is resident on
    // every device, nothing is exchanged between devices
    std::vector<int> devs = opt.devices.empty() ? std::vector<int>(1, opt.device) : opt.devices;
    std::vector<lamsa_hp_handle *> hs;
    int rc = LAMSA_HP_OK;
    for (size_t g = 0; g < devs.size() && rc == LAMSA_HP_OK && !opt.parse_only; ++g) { lamsa_hp_handle *hh = nullptr; rc = lamsa_hp_create(&hh, &P, &ref, devs[g]); if (rc == LAMSA_HP_OK) hs.push_back(hh); }
    if (rc != LAMSA_HP_OK) for (lamsa_hp_handle *hh : hs) lamsa_hp_destroy(hh);
    const int G = (int)std::max<size_t>(1, hs.size());
    if (rc != LAMSA_HP_OK) { fprintf(stderr, "[lamsa_aln] no usable MI355X / HIP device (lamsa_hp_create: %d); this build has no CPU path\n", rc); return 2; }
    // stage (4): needs the reference's FM index files and a second handle for its DP batches (a handle is single-threaded)
    FmIndex fm; lamsa_hp_handle *h_dp = nullptr; bool rescue = false; long n_rescue_jobs = 0;
    if (P.bwt_max_len > 0 && !opt.parse_only) {
        std::string e2;
        if (!fm.load(opt.ref_prefix, e2)) fprintf(stderr, "[lamsa_aln] note: stage 4 (BWT rescue of uncovered regions <= -R %d bp) is skipped: %s; output equals the reference's with -R 0\n", P.bwt_max_len, e2.c_str());
        else if (lamsa_hp_create(&h_dp, &P, nullptr, devs[0]) != LAMSA_HP_OK) { fprintf(stderr, "[lamsa_aln] cannot create the stage-4 handle\n"); for (lamsa_hp_handle *hh : hs) lamsa_hp_destroy(hh); return 2; }
        else rescue = true;
    }
    const double load_s = now_s() - t_begin;
    std::string sam;
    sam_header(sam, ix, pg_line);
    if (opt.shard_i == 0) fwrite(sam.data(), 1, sam.size(), out);
    long n_reads = 0, n_bases = 0, n_bad = 0; double kernel_ms = 0;
    const int threads = opt.n_thread > 0 ? opt.n_thread : 1;
    bool eof = false; int ret = 0;
    // One chunk = reads + their hit records, ready for the GPU.  Four stages overlap: the sequential pass over the
    // input files that cuts the next chunk (reads + the line span of every read in the mapped GEM file), the parse of
    // the chunk before it on all host threads, the GPU on the one before that, and the SAM text of the oldest.
    // The chunk buffers are recycled.
    struct Chunk { Batch B; int ret = 0; std::vector<std::pair<const char *, const char *>> span; std::vector<std::pair<size_t, size_t>> rspan; std::vector<std::string> text; bool mapped = false; lamsa_hp_batch hb; int dev = 0;
                   int sub_rc = 0, col_rc = 0; lamsa_hp_result res; std::promise<void> collected; };
    // One worker thread per device runs that device's submit / collect calls in the order they are queued (a handle is not
    // thread-safe, and the copies of a submit block their caller): the uploads of different devices then run side by side.
    struct DevQ {
        std::thread th; std::mutex m; std::condition_variable cv; std::deque<std::function<void()>> ops; bool stop = false;
        void finish() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv.notify_one(); if (th.joinable()) th.join(); }   // runs what is queued, then ends
        ~DevQ() { finish(); }                                   // every way out of run_aln (--parse-only, errors) ends the thread
    };
    std::mutex stat_m;
    HitsWriter saver;
    if (!opt.save_hits.empty() && !from_hits && !saver.open(opt.save_hits, P)) { fprintf(stderr, "[lamsa_aln] Can't write hit stream %s\n", opt.save_hits.c_str()); return 1; }
    std::vector<Chunk> pool((size_t)2 * G + 6); int n_scanned = 0;  // scanning, parsing, submitted (2 per device), being written, slack
    std::vector<Batch> parts((size_t)threads);              // per-thread partial batches of the parse, recycled too
    const bool trace = getenv("LAMSA_TRACE") != nullptr;
    auto scan = [&]() -> Chunk * {                          // sequential: FASTA/FASTQ records and their seed_all map lines
        Chunk *c = &pool[(size_t)(n_scanned++ % (2 * G + 6))];
        Batch &B = c->B;
        B.clear(); c->ret = 0; c->span.clear(); c->rspan.clear(); c->text.clear(); c->mapped = false;
        if (eof) return c;
        const double t0 = now_s();
        Read rd;
        int64_t chunk_bases = 0;
        if (from_hits) {                                    // the chunk as it was stored: its arrays are views into the mapped stream
            if (chunks_left == 0) { eof = true; return c; }
            if (chunks_left > 0) --chunks_left;
            if (hitsf.pos + sizeof(HitsChunkHeader) > hitsf.n) { eof = true; if (opt.shard_n <= 1 && fx.next(rd)) { fprintf(stderr, "[lamsa_read_seq] the hit stream ends before the reads\n"); c->ret = 1; } return c; }
            HitsChunkHeader ch; memcpy(&ch, hitsf.p + hitsf.pos, sizeof ch);
            const char *a = hitsf.p + hitsf.pos + sizeof ch;
            if (ch.n_reads <= 0 || ch.bytes < 0 || hitsf.pos + sizeof ch + (size_t)ch.bytes > hitsf.n) { fprintf(stderr, "[lamsa_read_seq] damaged hit stream\n"); c->ret = 1; eof = true; return c; }
            const size_t n = (size_t)ch.n_reads, nb = (size_t)ch.n_bases, ns = (size_t)ch.n_slots, nh = (size_t)ch.n_hits, nc = (size_t)ch.n_cig;
            lamsa_hp_batch &hb = c->hb;
            auto take = [&](size_t bytes) { const char *q = a; a += pad64(bytes); return q; };
            hb.n_reads = (int32_t)n; hb.n_cig = (int64_t)nc;
            hb.read_off = (const int64_t *)take(8 * (n + 1)); hb.read_seq = (const uint8_t *)take(nb); hb.seed_all = (const int32_t *)take(4 * n); hb.last_len = (const int32_t *)take(4 * n);
            hb.seed_off = (const int64_t *)take(8 * (n + 1)); hb.seed_id = (const int32_t *)take(4 * ns); hb.hit_off = (const int64_t *)take(8 * (ns + 1));
            hb.h_pos = (const int64_t *)take(8 * nh); hb.h_chr = (const int32_t *)take(4 * nh); hb.h_strand = (const int8_t *)take(nh); hb.h_nm = (const int16_t *)take(2 * nh);
            hb.h_len_dif = (const int16_t *)take(2 * nh); hb.h_cig_n = (const uint8_t *)take(nh);
            if (ch.cig_elem == 4) { hb.h_cig_off = (const int32_t *)take(4 * nh); hb.cig = (const int32_t *)take(4 * nc); hb.cig8 = nullptr; }
            else { hb.h_cig_off = nullptr; hb.cig = nullptr; hb.cig8 = (const uint8_t *)take(nc); }
            if ((size_t)(a - (hitsf.p + hitsf.pos + sizeof ch)) != (size_t)ch.bytes) { fprintf(stderr, "[lamsa_read_seq] damaged hit stream\n"); c->ret = 1; eof = true; return c; }
            {   // The chunk's pages of the mapping, touched by all threads now: a page of a file mapping costs a fault the first time it is read, and the
                // upload (one thread's copy into the runtime's page-locked staging buffers) otherwise pays 230 000 of them per chunk, one after the other.
                const char *p0 = hitsf.p + hitsf.pos; const size_t np = (sizeof ch + (size_t)ch.bytes + 4095) / 4096;
                static std::atomic<unsigned> sink(0);
                parallel_blocks((int)np, threads, [&](int, int b0, int b1) { unsigned v = 0; for (int k = b0; k < b1; ++k) v += (unsigned char)p0[(size_t)k * 4096]; sink += v; });
            }
            hitsf.pos += sizeof ch + (size_t)ch.bytes;
            if (fx.spans()) {                               // only where the records lie: they are parsed by all threads (prepare)
                c->rspan.resize(n);
                for (size_t r = 0; r < n; ++r)
                    if (!fx.next_span(c->rspan[r].first, c->rspan[r].second)) { fprintf(stderr, "[lamsa_read_seq] the hit stream does not match the reads\n"); c->ret = 1; eof = true; return c; }
                B.reads.resize(n);
            } else for (size_t r = 0; r < n; ++r) {
                if (!fx.next(rd) || (int64_t)rd.seq.size() != hb.read_off[r + 1] - hb.read_off[r]) { fprintf(stderr, "[lamsa_read_seq] the hit stream does not match the reads\n"); c->ret = 1; eof = true; return c; }
                B.reads.emplace_back(); std::swap(B.reads.back(), rd);
            }
            c->mapped = true;
            if (trace) fprintf(stderr, "[scan] %d reads of the hit stream in %.3f s\n", (int)n, now_s() - t0);
            return c;
        }
        while ((int)B.reads.size() < opt.chunk_reads && chunk_bases < opt.chunk_bases) {
            if (reads_left == 0) { eof = true; break; }
            if (reads_left > 0) --reads_left;
            if (!fx.next(rd)) {
                eof = true;
                if (mapt.following()) {                      // the reads are through: the mapper must have ended, and ended well
                    if (!mapt.follow_finish()) { fprintf(stderr, "[lamsa_aln] Seeding undone, gem-mapper exit abnormally.\n"); c->ret = 1; B.clear(); return c; }
                    fprintf(stderr, "[lamsa_aln] gem-mapper done!\n");
                }
                break;
            }
            const char *la, *lb;
            bool got;
            if (mapt.following()) {
                if (c->text.capacity() < (size_t)opt.chunk_reads + 1) c->text.reserve((size_t)opt.chunk_reads + 1);      // no reallocation: the spans point into the strings
                c->text.emplace_back();
                got = mapt.take_follow(seeds_of(P, (int)rd.seq.size()), c->text.back());
                la = c->text.back().data(); lb = la + c->text.back().size();
            } else got = mapt.take(seeds_of(P, (int)rd.seq.size()), la, lb);
            if (!got) { fprintf(stderr, "[lamsa_read_seq] seeds' GEM map result does not match the reads\n"); c->ret = 1; eof = true; return c; }
            c->span.emplace_back(la, lb);
            chunk_bases += (int64_t)rd.seq.size();
            B.reads.emplace_back(); std::swap(B.reads.back(), rd);
        }
        if (trace) fprintf(stderr, "[scan] %d reads in %.3f s\n", (int)B.reads.size(), now_s() - t0);
        return c;
    };
    std::future<Chunk *> scanned = std::async(std::launch::async, scan);
    // Device buffers (scratch slabs, inter-launch state, input / output: tens of GB) are allocated while the first chunk is being
    // parsed, sized from that chunk, instead of inside the first submit.
    std::thread reserver; double reserve_s = 0; bool reserve_started = false;
    auto prepare = [&]() -> Chunk * {
        const double t0 = now_s();
        Chunk *c = scanned.get();
        scanned = std::async(std::launch::async, scan);      // the following chunk is cut while this one is parsed
        Batch &B = c->B;
        const int n = (int)B.reads.size();
        if (n == 0 || c->ret) return c;
        if (!reserve_started && !opt.parse_only && !hs.empty()) {
            reserve_started = true;
            int64_t nb = 0; int max_len = 0;
            if (c->mapped) { for (int r = 0; r < n; ++r) { const int len = (int)(c->hb.read_off[r + 1] - c->hb.read_off[r]); nb += len; max_len = std::max(max_len, len); } }     // (the records of a chunk of the hit stream may not be parsed yet)
            else for (const Read &rd : B.reads) { nb += (int64_t)rd.seq.size(); max_len = std::max(max_len, (int)rd.seq.size()); }
            const int32_t r_n = (int32_t)std::max(n, opt.chunk_reads > n ? std::min(opt.chunk_reads, 2 * n) : n);
            const int64_t r_nb = nb + nb / 4 + 4096;
            reserver = std::thread([&hs, &reserve_s, r_n, r_nb, max_len]() {
                const double t = now_s();
                for (lamsa_hp_handle *hh : hs) {
                    const int e = lamsa_hp_reserve(hh, r_n, r_nb, r_nb / 3, 3 * r_nb, max_len + max_len / 4 + 256, 8192);
                    if (e != LAMSA_HP_OK) fprintf(stderr, "[lamsa_aln] note: device buffers could not be reserved ahead (%d %s); the first batch allocates what it needs\n", e, lamsa_hp_last_error(hh));
                }
                reserve_s = now_s() - t;
            });
        }
        if (c->mapped) {
            if (!c->rspan.empty()) {                        // the records of a chunk of the hit stream, parsed side by side
                std::vector<int> bad((size_t)threads, 0);
                parallel_blocks(n, threads, [&](int t, int r0, int r1) {
                    for (int r = r0; r < r1; ++r)
                        if (!fx.parse_span(c->rspan[(size_t)r].first, c->rspan[(size_t)r].second, B.reads[(size_t)r]) || (int64_t)B.reads[(size_t)r].seq.size() != c->hb.read_off[r + 1] - c->hb.read_off[r]) bad[(size_t)t] = 1;
                });
                for (int b : bad) if (b) { fprintf(stderr, "[lamsa_read_seq] the hit stream does not match the reads\n"); c->ret = 1; break; }
            }
            parse_s += now_s() - t0; return c;
        }
        const double t1 = now_s();
        // text -> hit records (gem_map_msg / map_cal_msg run inside the worker threads in the reference too)
        // The seed CIGARs are written in the boundary's compact form (a byte per element) as they are parsed; a chunk in which an element
        // does not fit a byte (longer than 63 -- no 50-base seed has one) is parsed once more into words.
        static const bool words_only = getenv("LAMSA_WIDE_CIGARS") != nullptr;      // tests: every chunk through the word form
        for (int pass = words_only ? 1 : 0; pass < 2; ++pass) {
            for (Batch &p : parts) { p.clear(); p.parse_compact = pass == 0; }
            parallel_blocks(n, threads, [&](int t, int r0, int r1) {
                for (int r = r0; r < r1; ++r) append_read_lines(parts[(size_t)t], ix, P, B.reads[(size_t)r], c->span[(size_t)r].first, c->span[(size_t)r].second, seeds_of(P, (int)B.reads[(size_t)r].seq.size()));
            });
            bool wide = false;
            for (const Batch &p : parts) wide |= p.saw_wide;
            if (!wide) break;
        }
        const double t2 = now_s();
        merge_batches(B, parts, threads);
        if (trace) fprintf(stderr, "[prepare] waited %.3f s for the scan, parse %.3f, merge %.3f\n", t1 - t0, t2 - t1, now_s() - t2);
        for (int32_t ch : B.h_chr) if (ch < 1) { fprintf(stderr, "[lamsa_aln] seed hit on a contig that is not in the index\n"); c->ret = 1; break; }
        if (saver.fp && !c->ret && !saver.write(B)) { fprintf(stderr, "[lamsa_aln] writing %s failed\n", opt.save_hits.c_str()); c->ret = 1; }
        parse_s += now_s() - t0;                            // one prepare() runs at a time
        return c;
    };
    std::vector<std::unique_ptr<DevQ>> devq;
    for (int g = 0; g < G; ++g) {
        devq.emplace_back(new DevQ);
        DevQ *q = devq.back().get();
        q->th = std::thread([q]() {
            for (;;) {
                std::function<void()> op;
                { std::unique_lock<std::mutex> lk(q->m); q->cv.wait(lk, [q] { return q->stop || !q->ops.empty(); }); if (q->ops.empty()) return; op = std::move(q->ops.front()); q->ops.pop_front(); }
                op();
            }
        });
    }
    auto enqueue = [&](int dev, std::function<void()> op) { DevQ *q = devq[(size_t)dev].get(); { std::lock_guard<std::mutex> lk(q->m); q->ops.push_back(std::move(op)); } q->cv.notify_one(); };
    auto stop_workers = [&]() { for (auto &q : devq) q->finish(); };
    long n_submitted = 0;
    auto submit = [&](Chunk &ck) -> int {
        Batch &B = ck.B;
        lamsa_hp_batch hb;
        if (ck.mapped) hb = ck.hb;
        else {
        hb.n_reads = (int32_t)B.reads.size(); hb.read_off = B.read_off.data(); hb.read_seq = B.read_seq.data(); hb.seed_all = B.seed_all.data(); hb.last_len = B.last_len.data();
        hb.seed_off = B.seed_off.data(); hb.seed_id = B.seed_id.data(); hb.hit_off = B.hit_off.data(); hb.h_pos = B.h_pos.data(); hb.h_chr = B.h_chr.data(); hb.h_strand = B.h_strand.data();
        hb.h_nm = B.h_nm.data(); hb.h_len_dif = B.h_len_dif.data(); hb.h_cig_n = B.h_cig_n.data(); hb.n_cig = (int64_t)B.cig8.size();
        if (B.cig_wide) { hb.h_cig_off = B.h_cig_off.data(); hb.cig = B.cig.data(); hb.cig8 = nullptr; }
        else { hb.h_cig_off = nullptr; hb.cig = nullptr; hb.cig8 = B.cig8.data(); }                 // compact form: a quarter of the CIGAR bytes, no offsets
        static const int32_t zero32 = 0; static const uint8_t zero8 = 0; static const int64_t zero64 = 0; static const int16_t zero16 = 0; static const int8_t zeroi8 = 0;
        if (!hb.seed_id) hb.seed_id = &zero32;
        if (!hb.h_pos) { hb.h_pos = &zero64; hb.h_chr = &zero32; hb.h_strand = &zeroi8; hb.h_nm = &zero16; hb.h_len_dif = &zero16; hb.h_cig_n = &zero8; }
        if (B.cig_wide && !hb.cig) hb.cig = &zero32;
        if (!B.cig_wide && !hb.cig8) hb.cig8 = &zero8;
        if (!hb.read_seq) hb.read_seq = &zero8;
        }
        const double t0 = now_s();
        const int e = lamsa_hp_submit_batch(hs[(size_t)ck.dev], &hb);
        { std::lock_guard<std::mutex> lk(stat_m); submit_s += now_s() - t0; }
        if (e != LAMSA_HP_OK) { fprintf(stderr, "[lamsa_aln] lamsa_hp_submit_batch failed: %d %s\n", e, lamsa_hp_last_error(hs[(size_t)ck.dev])); return 2; }
        return 0;
    };
    // records -> MAPQ / XA -> SAM text, on all host threads; written in input order
    // Everything after the GPU for one chunk: result streams -> records, stage (4) (rescue.h: plan on the host threads,
    // one DP batch on the GPU through the second handle, finish on the host threads), MAPQ / XA, SAM text in input order.
    // It runs as a task of its own beside the next collect; `res` stays valid until the collect after that.
    // The SAM text of a chunk, a part per thread, and the records it is made from: kept from chunk to chunk (one write_chunk runs at a time), so
    // their pages are touched and their vectors allocated once.  Two sets of text buffers: while the parts of chunk i are being written into the
    // file (pwrite, every thread its own part at its own offset) the text of chunk i + 1 is made in the other set.
    std::vector<std::string> sams_set[2] = {std::vector<std::string>((size_t)threads), std::vector<std::string>((size_t)threads)};
    std::future<int> file_write[2];
    int sams_turn = 0;
    off_t out_at = -1;                                        // where the next chunk goes in the output file (-1: not a regular file, or not asked yet)
    std::vector<ReadResult> RR;
    auto write_chunk = [&](Chunk *ck, lamsa_hp_result res) -> int {
        const double t1 = now_s();
        Batch &B = ck->B;
        const int n = (int)B.reads.size();
        const uint8_t *codes = ck->mapped ? ck->hb.read_seq : B.read_seq.data();
        const int64_t *roff = ck->mapped ? ck->hb.read_off : B.read_off.data();
        if (RR.size() < (size_t)n) RR.resize((size_t)n);
        std::vector<RescuePlan> plans(rescue ? (size_t)n : 0);
        std::vector<RescueJobs> tj((size_t)threads);
        std::vector<int> t_first((size_t)threads, 0), t_last((size_t)threads, 0);
        std::vector<long> bad_of((size_t)threads, 0);
        parallel_blocks(n, threads, [&](int t, int r0, int r1) {
            t_first[(size_t)t] = r0; t_last[(size_t)t] = r1;
            for (int r = r0; r < r1; ++r) {
                ReadResult &R = RR[(size_t)r];
                const int L = (int)B.reads[(size_t)r].seq.size();
                parse_stream(res.stream + res.read_off[r], res.read_len[r], L, R);
                if (rescue && R.status == 0) rescue_plan(R, codes + roff[r], L, ix, fm, P, plans[(size_t)r], tj[(size_t)t]);
            }
        });
        const double t2 = now_s();
        // the DP jobs of all threads as one batch
        std::vector<int64_t> base((size_t)threads + 1, 0);
        for (int t = 0; t < threads; ++t) base[(size_t)t + 1] = base[(size_t)t] + (int64_t)tj[(size_t)t].qlen.size();
        lamsa_hp_dp_out dpo; memset(&dpo, 0, sizeof dpo);
        if (base[(size_t)threads] > 0) {
            RescueJobs all;
            for (int t = 0; t < threads; ++t) {
                const RescueJobs &j = tj[(size_t)t];
                const int64_t s0 = (int64_t)all.seq.size();
                all.seq.insert(all.seq.end(), j.seq.begin(), j.seq.end());
                for (size_t k = 0; k < j.qlen.size(); ++k) { all.q_off.push_back(j.q_off[k] + s0); all.t_off.push_back(j.t_off[k] + s0); }
                all.qlen.insert(all.qlen.end(), j.qlen.begin(), j.qlen.end()); all.tlen.insert(all.tlen.end(), j.tlen.begin(), j.tlen.end());
                all.kind.insert(all.kind.end(), j.kind.begin(), j.kind.end()); all.w.insert(all.w.end(), j.w.begin(), j.w.end()); all.h0.insert(all.h0.end(), j.h0.begin(), j.h0.end());
            }
            all.seq.resize(all.seq.size() + 16, 0);
            lamsa_hp_dp_jobs dj;
            dj.n_jobs = (int32_t)all.qlen.size(); dj.seq = all.seq.data(); dj.seq_bytes = (int64_t)all.seq.size(); dj.q_off = all.q_off.data(); dj.qlen = all.qlen.data();
            dj.t_off = all.t_off.data(); dj.tlen = all.tlen.data(); dj.kind = all.kind.data(); dj.w = all.w.data(); dj.h0 = all.h0.data();
            const int e = lamsa_hp_dp_batch(h_dp, &dj, &dpo);
            if (e != LAMSA_HP_OK) { fprintf(stderr, "[lamsa_aln] lamsa_hp_dp_batch (stage 4) failed: %d %s\n", e, lamsa_hp_last_error(h_dp)); return 2; }
            for (int32_t k = 0; k < dj.n_jobs; ++k) if (dpo.status[k] != 0) { fprintf(stderr, "[lamsa_aln] a stage-4 DP job overflowed its work buffer\n"); return 2; }
            n_rescue_jobs += dj.n_jobs;
        }
        const double t3 = now_s();
        const int turn = sams_turn; sams_turn ^= 1;
        if (file_write[turn].valid()) { const int e = file_write[turn].get(); if (e) return e; }          // the write that still reads this set
        std::vector<std::string> &sams = sams_set[turn];
        for (std::string &x : sams) x.clear();
        parallel_blocks(n, threads, [&](int t, int, int) {
            std::string &o = sams[(size_t)t];
            DpResults dp; dp.score = dpo.score; dp.qle = dpo.qle; dp.tle = dpo.tle; dp.cig_off = dpo.cig_off; dp.cigar = dpo.cigar; dp.base = base[(size_t)t];
            for (int r = t_first[(size_t)t]; r < t_last[(size_t)t]; ++r) {
                ReadResult &R = RR[(size_t)r];
                const int L = (int)B.reads[(size_t)r].seq.size();
                if (rescue && R.status == 0 && !plans[(size_t)r].lines.empty()) rescue_finish(R, codes + roff[r], L, ix, P, plans[(size_t)r], dp);
                if (R.status != 0) {
                    ++bad_of[(size_t)t];
                    fprintf(stderr, "[lamsa_aln] read %s: %s; reported unmapped\n", B.reads[(size_t)r].name.c_str(), (R.status & LAMSA_HP_ST_UNSUPPORTED) ? "more than 32767 seeds, or a seed hit with |len_dif| > 127: beyond what the device keeps" : (R.status & LAMSA_HP_ST_REFEXIT) ? "input on which the reference aligner exits" : "device work buffer overflow");
                    for (int st = 0; st < 3; ++st) R.stage[st].clear();
                }
                rank_results(R, L, P);
                write_sam(o, R, B.reads[(size_t)r], ix, opt);
            }
        });
        // The chunk's text (15 KB per 10-kbp read) goes out in input order.  Into a regular file every thread writes its own part at its own
        // offset (pwrite: one thread's write() of 250 MB per chunk was the slowest stage of the loop once the hits came from a binary stream;
        // a mapping of the file's new end, round 3's way, pays a page fault per 4 KB and was slower still: profiles/r04_cli_bench.txt); a pipe
        // or a terminal takes the parts one after the other.
        const double t4 = now_s();
        bool mapped_out = false;
        {
            size_t total = 0; for (const std::string &x : sams) total += x.size();
            struct stat st; const int fd = fileno(out);
            static const size_t map_min = getenv("LAMSA_MAP_OUT_MIN") ? (size_t)atol(getenv("LAMSA_MAP_OUT_MIN")) : ((size_t)8 << 20);      // (tests lower it)
            if (threads > 1 && total >= map_min && fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && !(fcntl(fd, F_GETFL) & O_APPEND)) {
                if (out_at < 0 && fflush(out) == 0) out_at = lseek(fd, 0, SEEK_CUR);
                if (out_at >= 0) {
                    const off_t base = out_at;
                    out_at += (off_t)total;
                    // (the stream's own position follows: whatever is written through `out` later -- a small chunk, nothing else today -- lands behind this one)
                    if (fflush(out) != 0 || lseek(fd, out_at, SEEK_SET) != out_at) { fprintf(stderr, "[lamsa_aln] cannot position the output file\n"); return 2; }
                    file_write[turn] = std::async(std::launch::async, [&sams, fd, base, threads]() -> int {
                        std::vector<size_t> at((size_t)threads + 1, 0);
                        for (int t = 0; t < threads; ++t) at[(size_t)t + 1] = at[(size_t)t] + sams[(size_t)t].size();
                        std::atomic<int> werr(0);
                        std::vector<std::thread> th;
                        for (int t = 0; t < threads; ++t) if (!sams[(size_t)t].empty()) th.emplace_back([&, t]() {
                            const char *p = sams[(size_t)t].data(); size_t left = sams[(size_t)t].size(); off_t o = base + (off_t)at[(size_t)t];
                            while (left) { const ssize_t k = pwrite(fd, p, left, o); if (k <= 0) { if (k < 0 && errno == EINTR) continue; werr = errno ? errno : EIO; return; } p += k; left -= (size_t)k; o += k; }
                        });
                        for (auto &x : th) x.join();
                        if (werr.load()) { fprintf(stderr, "[lamsa_aln] cannot write the output file: %s\n", strerror(werr.load())); return 2; }
                        return 0;
                    });
                    mapped_out = true;
                    ++n_mapped_chunks;
                }
            }
        }
        if (!mapped_out) for (int k = 0; k < 2; ++k) if (file_write[k].valid()) { const int e = file_write[k].get(); if (e) return e; }     // (text through `out` goes behind what is still being written)
        if (!mapped_out) out_at = -1;                           // (asked again after this chunk has gone through the stream)
        for (int t = 0; t < threads; ++t) { if (!mapped_out) fwrite(sams[(size_t)t].data(), 1, sams[(size_t)t].size(), out); n_bad += bad_of[(size_t)t]; }
        for (const Read &q : B.reads) n_bases += (long)q.seq.size();
        n_reads += (long)B.reads.size();
        sam_s += now_s() - t1;
        if (trace) fprintf(stderr, "[write] %d reads: records%s %.3f s, stage-4 DP batch (%ld jobs) %.3f s, finish + rank + SAM text %.3f s, write started in %.3f s (it runs beside the next chunk's text)\n", n, rescue ? " + stage-4 plan" : "", t2 - t1, (long)base[(size_t)threads], t3 - t2, t4 - t3, now_s() - t4);
        return 0;
    };
    std::future<int> writer;                                 // the write_chunk task of the chunk collected last
    std::vector<std::future<int>> dev_writer((size_t)G);     // per device: the write task that still reads the results of its last collect
    auto collect_and_write = [&](Chunk *ck) -> int {
        // the results of a collect stay valid until the next collect on the same handle: the chunk collected from this device
        // before must have been written
        if (dev_writer[(size_t)ck->dev].valid()) { const int e = dev_writer[(size_t)ck->dev].get(); if (e) return e; }
        ck->collected = std::promise<void>();
        std::future<void> done = ck->collected.get_future();
        enqueue(ck->dev, [&, ck]() {
            const double t0 = now_s();
            lamsa_hp_handle *hc = hs[(size_t)ck->dev];
            ck->col_rc = ck->sub_rc ? ck->sub_rc : lamsa_hp_collect_batch(hc, &ck->res);
            { std::lock_guard<std::mutex> lk(stat_m); wait_s += now_s() - t0; if (ck->col_rc == LAMSA_HP_OK) kernel_ms += lamsa_hp_last_kernel_ms(hc, 0) + lamsa_hp_last_kernel_ms(hc, 1); }
            if (ck->col_rc != LAMSA_HP_OK && !ck->sub_rc) fprintf(stderr, "[lamsa_aln] lamsa_hp_collect_batch failed: %d %s\n", ck->col_rc, lamsa_hp_last_error(hc));
            ck->collected.set_value();
        });
        done.wait();
        if (ck->col_rc != LAMSA_HP_OK) return 2;
        if (writer.valid()) { const int e = writer.get(); if (e) return e; }      // SAM text is written in input order, one chunk at a time
        std::shared_future<int> sf = std::async(std::launch::async, write_chunk, ck, ck->res).share();
        writer = std::async(std::launch::deferred, [sf]() { return sf.get(); });
        dev_writer[(size_t)ck->dev] = std::async(std::launch::deferred, [sf]() { return sf.get(); });
        return 0;
    };
    // Two chunks are in flight on the device (lamsa_hp_submit_batch): while the GPU aligns chunk i-1, chunk i is
    // uploaded behind it, chunk i+1 is read and parsed, and -- once collected -- chunk i-1's SAM text is written.
    if (opt.parse_only) {
        for (;;) {
            Chunk *c = prepare();
            if (c->ret || c->B.reads.empty()) { ret = c->ret; break; }
            n_reads += (long)c->B.reads.size(); for (const Read &q : c->B.reads) n_bases += (long)q.seq.size();
        }
        if (scanned.valid()) scanned.wait();
        saver.close();
        if (stats) { stats->n_reads = n_reads; stats->n_bases = n_bases; stats->wall_s = now_s() - t_begin; stats->load_s = load_s; stats->parse_s = parse_s; }
        return ret;
    }
    std::future<Chunk *> next = std::async(std::launch::async, prepare);
    std::deque<Chunk *> flying;                              // submitted, not yet collected: at most two per device
    for (;;) {
        Chunk *cur = next.get();
        if (cur->ret) ret = cur->ret;
        const bool have = ret == 0 && !cur->B.reads.empty();
        if (have) {
            next = std::async(std::launch::async, prepare);  // overlaps with everything below
            if (reserver.joinable()) reserver.join();
            cur->dev = (int)(n_submitted++ % G); cur->sub_rc = 0;
            enqueue(cur->dev, [&, cur]() { cur->sub_rc = submit(*cur); });       // on the device's own thread: the next chunk can go to the next device meanwhile
            flying.push_back(cur);
        }
        while (!flying.empty() && ((int)flying.size() >= 2 * G || !have || ret)) {
            const int e = collect_and_write(flying.front()); if (e && !ret) ret = e;
            flying.pop_front();
        }
        if (!have || ret) break;
    }
    if (writer.valid()) { const int e = writer.get(); if (e && !ret) ret = e; }
    for (auto &f : dev_writer) if (f.valid()) f.get();
    for (auto &f : file_write) if (f.valid()) { const int e = f.get(); if (e && !ret) ret = e; }
    stop_workers();
    if (next.valid()) next.wait();                       // the reader threads must be done before the files are closed
    if (scanned.valid()) scanned.wait();
    saver.close();
    if (reserver.joinable()) reserver.join();
    for (lamsa_hp_handle *hh : hs) lamsa_hp_destroy(hh);
    if (h_dp) lamsa_hp_destroy(h_dp);
    if (trace) fprintf(stderr, "[write] %ld chunks written by all threads side by side (pwrite)\n", n_mapped_chunks);
    if (stats) { stats->n_reads = n_reads; stats->n_bases = n_bases; stats->n_bad = n_bad; stats->kernel_ms = kernel_ms;
                 stats->wall_s = now_s() - t_begin; stats->load_s = load_s; stats->parse_s = parse_s; stats->submit_s = submit_s; stats->wait_s = wait_s; stats->sam_s = sam_s; stats->reserve_s = reserve_s; }
    return ret;
}

}  // namespace lamsa
