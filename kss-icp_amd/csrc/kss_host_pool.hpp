// kss_host_pool.hpp -- a few host worker threads for the per-pair part of a batched ICP iteration.
//
// A 1024-pair batch (config C3) solves 1024 independent 3x3 SVDs + convergence tests per iteration on the host
// (~3 us each): done serially that is several times the GPU time of the iteration.  The pool splits an index
// range into contiguous chunks, one per thread (the caller takes the first); every pair is handled by exactly
// one thread with the same code as the serial loop, so results do not depend on the thread count.  Workers sleep
// on a condition variable between calls and are joined when the context is destroyed.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace kss {

class HostPool {
public:
    HostPool() = default;
    HostPool(const HostPool&) = delete;
    HostPool& operator=(const HostPool&) = delete;
    ~HostPool() { shutdown(); }

    // fn(begin, end) over [0, n) in up to threads() contiguous chunks; returns when all chunks are done
    void parallel_for(int n, const std::function<void(int, int)>& fn) {
        const int nt = std::min(threads(), std::max(1, n / kMinChunk));
        if (nt <= 1) { fn(0, n); return; }
        start_workers(nt - 1);
        const int chunk = (n + nt - 1) / nt;
        const int nchunks = (n + chunk - 1) / chunk;   // <= nt; the rounding of `chunk` can leave trailing threads empty
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn; n_ = n; chunk_ = chunk; nchunks_ = nchunks; pending_ = nchunks - 1; ++gen_;
        }
        cv_.notify_all();
        fn(0, std::min(n, chunk));   // the caller's share
        // the workers are at most a few microseconds behind: spin for them first (a condition-variable wake-up costs the
        // calling thread tens of microseconds, once per ICP pass of a batch), sleep only if they take long
        for (int spin = 0; spin < 200000 && pending_.load(std::memory_order_acquire) != 0; ++spin) __builtin_ia32_pause();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_.load(std::memory_order_acquire) == 0; });
        fn_ = nullptr;
    }

    // fn(t, nt) on nt threads at once (t = 0 is the caller): for loops in which every thread serves an interleaved share of
    // a set of pairs for as long as the whole set is in flight (the pair-resident engine)
    void run_threads(int nt, const std::function<void(int, int)>& fn) {
        if (nt <= 1) { fn(0, 1); return; }
        const std::function<void(int, int)> chunk = [&](int b, int) { fn(b, nt); };
        start_workers(nt - 1);
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &chunk; n_ = nt; chunk_ = 1; nchunks_ = nt; pending_ = nt - 1; ++gen_;
        }
        cv_.notify_all();
        fn(0, nt);
        for (int spin = 0; spin < 200000 && pending_.load(std::memory_order_acquire) != 0; ++spin) __builtin_ia32_pause();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_.load(std::memory_order_acquire) == 0; });
        fn_ = nullptr;
    }

    static int threads() {
        static const int t = [] {
            int v = (int)std::thread::hardware_concurrency();
            if (v <= 0) v = 1;
            v = std::min(v, 8);
            if (const char* e = std::getenv("KSS_HOST_THREADS")) { const int u = std::atoi(e); if (u >= 1 && u <= 64) v = u; }
            return v;
        }();
        return t;
    }

private:
    static constexpr int kMinChunk = 32;   // pairs per thread below which waking a worker costs more than it saves

    void start_workers(int want) {
        while ((int)workers_.size() < want) {
            const int id = (int)workers_.size() + 1;   // chunk index of this worker (0 is the caller)
            workers_.emplace_back([this, id] { worker(id); });
        }
    }

    void worker(int id) {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(int, int)>* fn;
            int n, chunk, nchunks;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                fn = fn_; n = n_; chunk = chunk_; nchunks = nchunks_;
            }
            if (id < nchunks) {   // workers beyond this call's chunk count have nothing to do and were not counted
                (*fn)(id * chunk, std::min(n, (id + 1) * chunk));
                if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                    std::lock_guard<std::mutex> lk(m_);
                    done_.notify_one();
                }
            }
        }
    }

    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread& t : workers_) t.join();
        workers_.clear();
    }

    std::mutex m_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> workers_;
    const std::function<void(int, int)>* fn_ = nullptr;
    int n_ = 0, chunk_ = 0, nchunks_ = 0;
    std::atomic<int> pending_{0};
    unsigned long long gen_ = 0;
    bool stop_ = false;
};

}  // namespace kss
