// work_pool.h - a persistent pool of host threads for the page copies of the host-list entries (host_batch.hip).
// Plain C++11 (no HIP): tests/cpp/test_work_pool.cpp exercises it on the CPU, tools/sanitize_cpu.sh under ThreadSanitizer.
//
// parallel_for(n, f) runs f(0) .. f(n-1) on the pool's threads and the caller; several callers may be inside at once (the
// upload and the download side of a device worker, the workers of several devices): every call waits for its own batch
// only.  Threads are created once and never joined, and a pool is never destroyed (allocate it with `new` and keep it: its
// detached workers wait on its condition variable for the life of the process; no joins during process teardown); after a
// fork() the child runs everything on the calling thread.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

namespace prl_hip {

class WorkPool {
public:
    explicit WorkPool(int threads)   // `threads` includes the calling thread: threads - 1 workers are started
    {
        owner_pid_ = getpid();
        for (int i = 0; i + 1 < threads; ++i) workers_.emplace_back([this] { work_on(nullptr); });
        for (auto& t : workers_) t.detach();
    }
    WorkPool(const WorkPool&) = delete;
    WorkPool& operator=(const WorkPool&) = delete;

    template <typename F>
    void parallel_for(int n, F&& f)
    {
        if (n <= 0) return;
        if (n == 1 || workers_.empty() || getpid() != owner_pid_) {
            for (int i = 0; i < n; ++i) f(i);
            return;
        }
        Batch b;
        b.fn = [&](int i) { f(i); };
        b.n = n;
        {
            std::lock_guard<std::mutex> lk(mu_);
            batches_.push_back(&b);
        }
        cv_.notify_all();
        work_on(&b);   // the caller helps with its own batch
        std::unique_lock<std::mutex> lk(mu_);
        b.done_cv.wait(lk, [&] { return b.finished == b.n; });
        batches_.erase(std::remove(batches_.begin(), batches_.end(), &b), batches_.end());   // (it lives on this stack frame)
    }
    int threads() const { return (int)workers_.size() + 1; }

private:
    struct Batch {
        std::function<void(int)> fn;
        int n = 0, next = 0, finished = 0;   // guarded by the pool's mutex
        std::condition_variable done_cv;
    };
    void work_on(Batch* only)
    {
        for (;;) {
            Batch* b = nullptr;
            int i = -1;
            {
                std::unique_lock<std::mutex> lk(mu_);
                if (only) {
                    if (only->next < only->n) { b = only; i = b->next++; }
                    else return;
                } else {
                    cv_.wait(lk, [&] {
                        while (!batches_.empty() && batches_.front()->next >= batches_.front()->n) batches_.pop_front();
                        return !batches_.empty();
                    });
                    b = batches_.front();
                    i = b->next++;
                }
            }
            b->fn(i);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (++b->finished == b->n) b->done_cv.notify_all();   // (under the lock: the batch lives on its caller's stack)
            }
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Batch*> batches_;
    std::vector<std::thread> workers_;
    pid_t owner_pid_ = 0;
};

}  // namespace prl_hip
