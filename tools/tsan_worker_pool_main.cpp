// ThreadSanitizer harness of host/src/workerPool.h: many parallelFor calls of varying size from several caller threads at
// once (each with its own pool), results checked; run by tools/tsan_worker_pool.sh.
#include "../trajectory_planner_amd/host/src/workerPool.h"

#include <cstdio>
#include <numeric>

int main() {
    std::atomic<long> bad{0};
    auto caller = [&](int seed) {
        unsigned s = 12345u + seed;
        for (int rep = 0; rep < 300; ++rep) {
            s = s * 1664525u + 1013904223u;
            const size_t n = (s >> 8) % 700;
            std::vector<int> out(n, 0);
            vigo_host::parallelFor(n, [&](size_t i) { out[i] += (int)i + 1; });
            long sum = 0;
            for (size_t i = 0; i < n; ++i) sum += out[i];
            if (sum != (long)n * (long)(n + 1) / 2) ++bad;
        }
    };
    std::vector<std::thread> callers;
    for (int t = 0; t < 4; ++t) callers.emplace_back(caller, t);
    caller(99);   // and the main thread
    for (auto& t : callers) t.join();
    std::printf("%s\n", bad.load() ? "FAILED" : "worker pool: 1500 parallelFor calls from 5 caller threads, all sums right");
    return bad.load() != 0;
}
