// hp_hostprep.h -- host-side helpers of the batch entry points (plain C++, no HIP): the validation pass over a
// batch runs on all host threads, one contiguous block of reads each.
#pragma once
#include <stdint.h>
#include <algorithm>
#include <thread>
#include <vector>

// fn(first, last) over [0, n) split into contiguous blocks, one per host thread (reads are independent)
template <class F> static inline void hp_parallel_blocks(int n, F fn)
{
    unsigned hw = std::thread::hardware_concurrency();
    int T = (int)(hw ? hw : 1); if (T > 32) T = 32;
    if (n < 256 || T < 2) { fn(0, n); return; }
    std::vector<std::thread> th;
    const int per = (n + T - 1) / T;
    for (int t = 0; t < T; ++t) { const int a = t * per, b = std::min(n, a + per); if (a < b) th.emplace_back([=]() { fn(a, b); }); }
    for (auto &x : th) x.join();
}
