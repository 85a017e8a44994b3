// ThreadSanitizer harness of host/src/workerPool.h: many parallelFor calls of varying size from several caller threads at
// once (each with its own pool), results checked; run by tools/tsan_worker_pool.sh.
#include "../trajectory_planner_amd/host/src/workerPool.h"

#include <cstdio>
#include <numeric>
#include <stdexcept>

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
    // a parallelFor nested on the calling thread (same pool: must run serially, not overwrite the job in flight) and on
    // the workers (their own pools)
    {
        const size_t n = 400, m = 64;
        std::vector<std::atomic<int>> hits(n * m);
        for (auto& h : hits) h.store(0);
        vigo_host::parallelFor(n, [&](size_t i) { vigo_host::parallelFor(m, [&](size_t j) { hits[i * m + j].fetch_add(1); }); });
        for (auto& h : hits) if (h.load() != 1) ++bad;
    }
    // fn throws, on whichever thread draws index 137: every other thread has left the job before run() unwinds, the
    // exception arrives on the caller, and the pool is usable afterwards
    for (int rep = 0; rep < 50; ++rep) {
        std::atomic<int> ran{0};
        bool caught = false;
        try {
            vigo_host::parallelFor(600, [&](size_t i) {
                if (i == 137) throw std::runtime_error("boom");
                ran.fetch_add(1);
            });
        } catch (const std::runtime_error&) {
            caught = true;
        }
        if (!caught || ran.load() >= 600) ++bad;
        std::vector<int> out(300, 0);
        vigo_host::parallelFor(out.size(), [&](size_t i) { out[i] = 1; });
        if (std::accumulate(out.begin(), out.end(), 0) != 300) ++bad;
    }
    // the companion threads of makePlanBatch: jobs started and awaited in turn from two caller threads, each with its own
    // companions, the jobs themselves running parallelFor on the companion's own pool; a throwing job ends cleanly
    {
        auto user = [&](int seed) {
            vigo_host::Companion comp[3];
            for (int rep = 0; rep < 40; ++rep) {
                std::vector<long> sums(3, 0);
                for (int k = 0; k < 3; ++k)
                    comp[k].start([&sums, k, rep, seed]() {
                        std::vector<int> out(200 + 10 * k, 0);
                        vigo_host::parallelFor(out.size(), [&](size_t i) { out[i] = (int)i + rep + seed; });
                        sums[k] = std::accumulate(out.begin(), out.end(), 0L);
                        if (rep == 7 && k == 1) throw std::runtime_error("job failed");
                    });
                for (int k = 0; k < 3; ++k) comp[k].wait();
                for (int k = 0; k < 3; ++k) {
                    const long n = 200 + 10 * k;
                    if (sums[k] != n * (n - 1) / 2 + n * (rep + seed)) ++bad;
                }
            }
        };
        std::thread other(user, 5);
        user(9);
        other.join();
    }
    std::printf("%s\n", bad.load() ? "FAILED" : "worker pool: 1500 parallelFor calls from 5 caller threads, all sums right; nested and throwing jobs handled; companion threads: 240 jobs from 2 callers");
    return bad.load() != 0;
}
