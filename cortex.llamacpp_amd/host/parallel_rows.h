// parallel_rows.h - rows [0, n) of an image over a few host threads (the image side of a LLaVA request: resampling, inverse DCT, colour conversion).  Every
// row is independent, so the result does not depend on the split; small jobs (work = pixels or so below 64 K) stay on the calling thread.
#pragma once

#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

namespace mi355 {

template <class F>
inline void parallel_rows(int n, int64_t work, F f) {
    const int nt = work < (1 << 16) ? 1 : (int)std::min<int64_t>(8, std::min<int64_t>((int64_t)std::thread::hardware_concurrency(), n / 16));
    if (nt <= 1) { f(0, n); return; }
    std::vector<std::thread> th;
    int started = 1;                                            // ranges [n t / nt, n (t + 1) / nt): range 0 is the caller's
    try {
        for (; started < nt; started++) th.emplace_back(f, (int)((int64_t)n * started / nt), (int)((int64_t)n * (started + 1) / nt));
    } catch (...) {}                                            // (no more threads to be had: the caller does the rest)
    f(0, (int)((int64_t)n / nt));
    if (started < nt) f((int)((int64_t)n * started / nt), n);
    for (auto &t : th) t.join();
}

}  // namespace mi355
