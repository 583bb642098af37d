// emu_stats.cpp -- TEST/DIAGNOSTIC ONLY: histogram of the chaining passes (range size, active hits, due targets) when the
// device sources are compiled with -DHP_EMU_STATS on the CPU lane emulation:
//   g++ -O1 -std=c++17 -fPIC -shared -DHP_EMU_STATS -I tests/emu -I lamsa_amd/csrc -o /tmp/libemu_stats.so tests/emu/emu_api.cpp tests/emu/emu_stats.cpp
#include <stdio.h>
#include <map>
namespace hp { void hp_emu_stat_call(int, int, int, bool); void hp_emu_stat_due(int, int, int); void hp_emu_stat_run(int, int); }
static std::map<int, long> runs[4];
void hp::hp_emu_stat_run(int n, int is_run) { int b = 0; while ((64 << b) < n) ++b; ++runs[is_run][b]; }
static std::map<int, long> calls, due_by_span, targets_by_range;
static long forced = 0;
static int bucket(int x) { int b = 0; while ((1 << b) < x) ++b; return b; }
void hp::hp_emu_stat_call(int range, int lo, int hi, bool force) { if (force) { ++forced; return; } ++calls[bucket(range)]; }
void hp::hp_emu_stat_due(int range, int due, int span) { targets_by_range[bucket(range)] += due; due_by_span[bucket(span)] += due; }
extern "C" void hp_emu_stat_dump() {
    { const char *nm[4] = {"pass with a head, run", "pass from START, run", "pass with a head, listed", "pass from START, listed"};
      for (int t = 0; t < 4; ++t) for (auto &k : runs[t]) fprintf(stderr, "%s <= %5d hits: %ld passes\n", nm[t], 64 << k.first, k.second); }
    fprintf(stderr, "forced %ld\n", forced);
    for (auto &k : calls) fprintf(stderr, "range<=%6d calls %8ld due targets %9ld\n", 1 << k.first, k.second, targets_by_range[k.first]);
    for (auto &k : due_by_span) fprintf(stderr, "candidate span (hits in [start_slot,k1)) <=%6d : due targets %9ld\n", 1 << k.first, k.second);
}
