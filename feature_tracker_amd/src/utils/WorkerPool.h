// Small persistent worker pool for per-stream host bookkeeping.  Independent camera streams share nothing on the
// host (each has its own tracker state and feature database), so their prepare/finish steps run concurrently, like
// the reference's per-camera host threads (src/track/TrackKLT.cpp:91-98) but without creating threads per frame.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace feature_tracker {

class WorkerPool {
public:
    explicit WorkerPool(int n_threads, int spin_microseconds = 2000) : spin_us(spin_microseconds) {
        workers.reserve(n_threads);   // run() reads workers.size(): fix it before the first thread starts
        n_workers = n_threads;
        for (int i = 0; i < n_threads; i++) workers.emplace_back([this, i] { run(i + 1); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
            generation.fetch_add(1, std::memory_order_release);
        }
        cv.notify_all();
        for (auto& t : workers) t.join();
    }
    WorkerPool(const WorkerPool&) = delete;
    WorkerPool& operator=(const WorkerPool&) = delete;

    int threads() const { return n_workers; }

    // calls fn(i) for i in [0, n) and returns when all are done.  The split is static: index i always runs on
    // participant i % (workers + 1) (the calling thread is participant 0), so a stream's state and the heap blocks
    // it allocates stay with one thread (no allocator arena ping-pong, warm caches).
    void parallel_for(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        if (workers.empty() || n == 1) {
            for (int i = 0; i < n; i++) fn(i);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m);
            job = &fn;
            job_n = n;
            done.store(0, std::memory_order_relaxed);
            generation.fetch_add(1, std::memory_order_release);
        }
        if (sleepers.load(std::memory_order_acquire) > 0) cv.notify_all();
        const int parts = n_workers + 1;
        for (int i = 0; i < n; i += parts) fn(i);
        while (done.load(std::memory_order_acquire) < n_workers) {}
        std::lock_guard<std::mutex> lk(m);
        job = nullptr;
    }

private:
    void run(int self) {
        unsigned seen = 0;
        for (;;) {
            // stay hot across the stages of one feed and across back-to-back feeds: a condition-variable wake-up of a
            // sleeping worker costs tens of microseconds, as much as the work it would be given
            const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us);
            while (generation.load(std::memory_order_acquire) == seen && std::chrono::steady_clock::now() < spin_until) {}
            const std::function<void(int)>* fn = nullptr;
            int n = 0;
            {
                std::unique_lock<std::mutex> lk(m);
                if (generation.load(std::memory_order_acquire) == seen) {
                    sleepers.fetch_add(1, std::memory_order_release);
                    cv.wait(lk, [&] { return generation.load(std::memory_order_acquire) != seen; });
                    sleepers.fetch_sub(1, std::memory_order_release);
                }
                seen = generation.load(std::memory_order_acquire);
                if (stop) return;
                fn = job;
                n = job_n;
            }
            if (fn) {
                const int parts = n_workers + 1;
                for (int i = self; i < n; i += parts) (*fn)(i);
                done.fetch_add(1, std::memory_order_release);
            }
        }
    }

    int spin_us;
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv;
    std::atomic<unsigned> generation{0};
    std::atomic<int> done{0}, sleepers{0};
    int n_workers = 0;
    const std::function<void(int)>* job = nullptr;
    int job_n = 0;
    bool stop = false;
};

}  // namespace feature_tracker
