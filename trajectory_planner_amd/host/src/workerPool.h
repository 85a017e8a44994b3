// workerPool.h — fn(i) for i in [0, n) on up to 16 host threads (planner-local work of the facades' batch entry points).
// The workers are kept: a pool per CALLING thread (two host threads planning two batches do not share one), created on
// first use, parked on a condition variable between calls — spawning 15 threads per call cost ~0.3 ms of every
// makePlanBatch phase.  tools/tsan_worker_pool.sh runs it under ThreadSanitizer.
#ifndef VIGO_HOST_WORKER_POOL_H
#define VIGO_HOST_WORKER_POOL_H
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace vigo_host {

class WorkerPool {
public:
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cvStart_.notify_all();
        for (auto& t : workers_) t.join();
    }
    // fn(i) for i in [0, n).  A call made from inside a job of this same pool (the calling thread's fn nesting another
    // parallelFor) runs serially: the pool has one job slot.  An exception thrown by fn — on the caller or on a worker —
    // stops the hand-out of further indices, is held until every worker has left the job (they read `job`, which lives
    // on this frame), and is then rethrown on the caller; the first one wins.
    template <typename F>
    void run(size_t n, unsigned nthr, F& fn) {
        if (nthr <= 1 || n <= 1 || inRun_) {
            for (size_t i = 0; i < n; ++i) fn(i);
            return;
        }
        struct Guard {
            bool& f;
            explicit Guard(bool& x) : f(x) { f = true; }
            ~Guard() { f = false; }
        } guard(inRun_);
        while (workers_.size() + 1 < nthr) workers_.emplace_back([this]() { this->loop(); });
        std::function<void(size_t)> job = [&fn](size_t i) { fn(i); };
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &job;
            n_ = n;
            next_.store(0);
            wanted_ = nthr - 1;          // workers that may join this job (the caller is the nthr-th)
            pending_ = 0;
            error_ = nullptr;
            ++generation_;
        }
        cvStart_.notify_all();
        std::exception_ptr mine;
        try {
            for (size_t i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) fn(i);
        } catch (...) {
            mine = std::current_exception();
            next_.store(n);              // nobody takes another index
        }
        std::exception_ptr theirs;
        {
            std::unique_lock<std::mutex> lk(m_);
            wanted_ = 0;                 // late wakers find nothing to join
            cvDone_.wait(lk, [this]() { return pending_ == 0; });
            job_ = nullptr;
            theirs = error_;
            error_ = nullptr;
        }
        if (mine) std::rethrow_exception(mine);
        if (theirs) std::rethrow_exception(theirs);
    }

private:
    void loop() {
        unsigned seen = 0;
        for (;;) {
            const std::function<void(size_t)>* job = nullptr;
            size_t n = 0;
            {
                std::unique_lock<std::mutex> lk(m_);
                cvStart_.wait(lk, [&]() { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                if (wanted_ == 0) continue;
                --wanted_;
                ++pending_;
                job = job_;
                n = n_;
            }
            std::exception_ptr err;
            try {
                for (size_t i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) (*job)(i);
            } catch (...) {
                err = std::current_exception();
                next_.store(n);
            }
            {
                std::lock_guard<std::mutex> lk(m_);
                if (err && !error_) error_ = err;
                --pending_;
            }
            cvDone_.notify_one();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cvStart_, cvDone_;
    const std::function<void(size_t)>* job_ = nullptr;
    size_t n_ = 0;
    std::atomic<size_t> next_{0};
    unsigned wanted_ = 0, pending_ = 0, generation_ = 0;
    bool stop_ = false;
    bool inRun_ = false;                 // only the owning (calling) thread touches it: the pool is thread_local
    std::exception_ptr error_;           // first exception thrown by a worker's share of the current job
};

// A further host thread kept by a caller of bsplineTraj::makePlanBatch for the batches it splits into pipelined parts: it
// lives as long as the calling thread, so its HIP stream, staging buffers and worker pool (all thread_local) are created
// once.  start() hands it one job, wait() returns when that job has run (an exception inside a job ends the job).
// tools/tsan_worker_pool.sh runs it under ThreadSanitizer.
class Companion {
public:
    Companion() : th_([this]() { this->loop(); }) {}
    ~Companion() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void start(std::function<void()> job) {
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = std::move(job);
            busy_ = true;
        }
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this]() { return !busy_; });
    }

private:
    void loop() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this]() { return stop_ || (busy_ && job_); });
                if (stop_) return;
                job = std::move(job_);
                job_ = nullptr;
            }
            try { job(); } catch (...) { }      // (makePlanBatch reports failure through its result vector)
            {
                std::lock_guard<std::mutex> lk(m_);
                busy_ = false;
            }
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool busy_ = false, stop_ = false;
    std::thread th_;
};

template <typename F>
void parallelFor(size_t n, F fn) {
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const unsigned nthr = (unsigned)std::min<size_t>(hw, std::max<size_t>(1, n / 8));
    static thread_local WorkerPool pool;
    pool.run(n, nthr, fn);
}

}  // namespace vigo_host
#endif
