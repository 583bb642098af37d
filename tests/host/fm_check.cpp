// fm_check.cpp -- TEST INFRASTRUCTURE: the FM-index queries of lamsa_amd/host/rescue.cpp (occ, exact match, suffix-array
// lookup on the reference's .bwt/.sa) against a brute-force k-mer table of the same reference (.pac, both strands).
// usage: fm_check <index prefix> <k> <n queries>; prints "ok <hits checked>" or the first disagreement; exit code 0/1.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>
#include <algorithm>
#include "rescue.h"

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const std::string prefix = argv[1]; const int k = atoi(argv[2]), nq = atoi(argv[3]);
    lamsa::Index ix; lamsa::FmIndex fm; std::string err;
    if (!lamsa::load_index(prefix, ix, err) || !fm.load(prefix, err)) { fprintf(stderr, "%s\n", err.c_str()); return 2; }
    const int64_t n = ix.l_pac;
    std::vector<uint8_t> text((size_t)(2 * n));                      // forward strand, then its reverse complement
    for (int64_t i = 0; i < n; ++i) { const uint8_t c = ix.pac[(size_t)(i >> 2)] >> ((~i & 3) << 1) & 3; text[(size_t)i] = c; text[(size_t)(2 * n - 1 - i)] = 3 - c; }
    if (fm.seq_len != (uint64_t)(2 * n)) { printf("seq_len %llu != %lld\n", (unsigned long long)fm.seq_len, (long long)(2 * n)); return 1; }
    std::unordered_map<uint64_t, std::vector<int64_t>> table;
    uint64_t w = 0; const uint64_t mask = (1ull << (2 * k)) - 1;
    for (int64_t i = 0; i < 2 * n; ++i) { w = ((w << 2) | text[(size_t)i]) & mask; if (i >= k - 1) table[w].push_back(i - k + 1); }
    std::mt19937_64 rng(7);
    long checked = 0;
    for (int q = 0; q < nq; ++q) {
        std::vector<uint8_t> s((size_t)k);
        if (q % 4 == 3) for (int j = 0; j < k; ++j) s[(size_t)j] = (uint8_t)(rng() & 3);                 // mostly absent
        else { const int64_t p = (int64_t)(rng() % (uint64_t)(2 * n - k)); for (int j = 0; j < k; ++j) s[(size_t)j] = text[(size_t)(p + j)]; }
        uint64_t key = 0; for (int j = 0; j < k; ++j) key = (key << 2) | s[(size_t)j];
        uint64_t lo = 0, hi = fm.seq_len;
        const uint64_t cnt = fm.match(k, s.data(), &lo, &hi);
        auto it = table.find(key);
        const size_t want = it == table.end() ? 0 : it->second.size();
        if (cnt != want) { printf("query %d: %llu hits, expected %zu\n", q, (unsigned long long)cnt, want); return 1; }
        if (!cnt) continue;
        std::vector<int64_t> got;
        for (uint64_t m = lo; m <= hi; ++m) got.push_back((int64_t)fm.sa_at(m));
        std::sort(got.begin(), got.end());
        std::vector<int64_t> exp = it->second; std::sort(exp.begin(), exp.end());
        if (got != exp) { printf("query %d: positions differ (first %lld vs %lld)\n", q, (long long)got[0], (long long)exp[0]); return 1; }
        checked += (long)cnt;
    }
    // a query with an N never matches
    { std::vector<uint8_t> s((size_t)k, 0); s[(size_t)(k / 2)] = 4; uint64_t lo = 0, hi = fm.seq_len; if (fm.match(k, s.data(), &lo, &hi) != 0) { printf("N matched\n"); return 1; } }
    printf("ok %ld\n", checked);
    return 0;
}
