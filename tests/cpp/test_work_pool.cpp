// CPU test of prlib_amd/csrc/prl/work_pool.h (the copy-thread pool of the host-list entries): several callers inside
// parallel_for at once, nested sizes from 0 to thousands, every index run exactly once, callers only wait for their own batch.
// Built by tests/cpp/Makefile (g++), also under ThreadSanitizer by tools/sanitize_cpu.sh.
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "../../prlib_amd/csrc/prl/work_pool.h"

int main()
{
    // pools are never destroyed (the library's lives as long as the process; its detached workers wait on its condition variable)
    prl_hip::WorkPool& pool = *new prl_hip::WorkPool(6);
    if (pool.threads() != 6) { std::printf("threads %d\n", pool.threads()); return 1; }
    std::atomic<long long> bad{0};
    auto caller = [&](int seed) {
        for (int round = 0; round < 200; ++round) {
            const int n = (seed * 7919 + round * 104729) % 3000;   // includes 0 and 1
            std::vector<std::atomic<int>> hit((size_t)n);
            for (auto& h : hit) h.store(0);
            long long sum = 0;
            std::atomic<long long> got{0};
            pool.parallel_for(n, [&](int i) { hit[(size_t)i].fetch_add(1); got.fetch_add(i); });
            for (int i = 0; i < n; ++i) { sum += i; if (hit[(size_t)i].load() != 1) bad.fetch_add(1); }
            if (got.load() != sum) bad.fetch_add(1);
        }
    };
    std::vector<std::thread> callers;
    for (int t = 0; t < 5; ++t) callers.emplace_back(caller, t + 1);
    for (auto& t : callers) t.join();
    // a single-thread pool runs everything on the caller
    prl_hip::WorkPool& solo = *new prl_hip::WorkPool(1);
    int count = 0;
    solo.parallel_for(100, [&](int) { ++count; });
    if (count != 100) bad.fetch_add(1);
    std::printf("work_pool: %s\n", bad.load() == 0 ? "OK" : "FAILED");
    return bad.load() == 0 ? 0 : 1;
}
