// Host-side helper threads for rpt_set_objects: the per-object screen bounds (rpt_screen_bounds.hpp) are pure functions of one
// object's record, 1-13 us each, and an animated scene with dozens of objects recomputes all of them every frame on the thread
// that submits the frame — as much time as the device needs for the frame (profiles/r02_host_cost.txt).  A handful of process-wide
// workers share such a batch with the submitting thread.  Design constraints:
//   * the caller never depends on the workers: it takes items itself until none are left, so a process whose workers are gone
//     (after fork(), or with RPT_HOST_THREADS=0) computes everything on its own;
//   * no allocation, no system call on the fast path: a job is published through one atomic pointer, items are claimed with
//     fetch_add, workers that have just worked keep polling for ~200 us before they sleep on a condition variable (frames
//     arrive every 30-200 us), and the caller wakes them only if somebody sleeps;
//   * a job lives on the caller's stack: the caller leaves only after every item is done AND no worker holds the pointer
//     (workers announce themselves before they read it; both sides use sequentially consistent operations).
// The workers are never joined (the pool is leaked on purpose: a library has no safe moment to join threads at process exit).
// fork(): the child has none of the parent's threads but inherits the pool's counters and, possibly, a mutex some worker held
// at that instant.  A pthread_atfork child handler empties the pool (n_threads_ = 0, counters 0, no job): every parallel_for of
// the child then takes the solo path and never touches m_ or callers_ again.  The pool is created in rpt_create (not lazily in
// the first frame): a fork() that overlapped its construction would leave the child waiting on the initialisation guard of
// instance() for a thread it does not have (tests/native/workers_stress.cpp found exactly that).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <pthread.h>
#include <thread>

namespace rpth {

class Workers {
public:
    struct Job {
        void (*fn)(void *arg, int index);
        void *arg;
        int count;
        std::atomic<int> next{0};
        std::atomic<int> pending{0};
    };

    static Workers &instance() {
        static Workers *pool = new Workers();      // leaked: see above
        return *pool;
    }
    int threads() const { return n_threads_.load(); }

    // fn(arg, i) for every i in [0, count), on the calling thread and on whichever workers show up; returns when all are done
    void parallel_for(int count, void (*fn)(void *, int), void *arg) {
        if (count <= 0) return;
        Job job;
        job.fn = fn;
        job.arg = arg;
        job.count = count;
        job.pending.store(count);
        std::unique_lock<std::mutex> one_caller(callers_, std::defer_lock);
        const bool shared = n_threads_.load() > 0 && one_caller.try_lock();     // a second submitting thread does its batch alone
        if (shared) {
            job_.store(&job);
            epoch_.fetch_add(1);
            if (sleepers_.load() > 0) {
                std::lock_guard<std::mutex> lk(m_);
                cv_.notify_all();
            }
        }
        run(job);
        if (shared) {
            job_.store(nullptr);
            while (job.pending.load() > 0 || holders_.load() > 0) pause();
        }
    }

private:
    Workers() {
        int n = 4;
        if (const char *e = std::getenv("RPT_HOST_THREADS")) n = std::atoi(e);
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && n > hw / 2 - 1) n = hw / 2 - 1;          // leave the submitting threads and the runtime their cores
        if (n < 0) n = 0;
        if (n > 16) n = 16;
        for (int k = 0; k < n; k++) {
            try {
                std::thread([this] { worker(); }).detach();
                n_threads_.fetch_add(1);
            } catch (...) {
                break;
            }
        }
        pthread_atfork(nullptr, nullptr, [] { instance().after_fork_in_child(); });
    }
    // the child of a fork(): no workers exist here; whatever the counters and mutexes said in the parent is void
    void after_fork_in_child() {
        n_threads_.store(0);
        job_.store(nullptr);
        holders_.store(0);
        sleepers_.store(0);
    }

    static void pause() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }

    static void run(Job &job) {
        for (;;) {
            const int i = job.next.fetch_add(1);
            if (i >= job.count) break;
            job.fn(job.arg, i);
            job.pending.fetch_sub(1);
        }
    }

    void worker() {
        unsigned seen = epoch_.load();
        for (;;) {
            // wait for a new epoch: poll for a while, then sleep
            const auto t0 = std::chrono::steady_clock::now();
            int polls = 0;
            while (epoch_.load() == seen) {
                pause();
                if ((++polls & 255) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) {
                    std::unique_lock<std::mutex> lk(m_);
                    sleepers_.fetch_add(1);
                    cv_.wait(lk, [&] { return epoch_.load() != seen; });
                    sleepers_.fetch_sub(1);
                }
            }
            seen = epoch_.load();
            holders_.fetch_add(1);                   // announced BEFORE the pointer is read (see parallel_for's exit condition)
            if (Job *job = job_.load()) run(*job);
            holders_.fetch_sub(1);
        }
    }

    std::atomic<int> n_threads_{0};
    std::atomic<Job *> job_{nullptr};
    std::atomic<unsigned> epoch_{0};
    std::atomic<int> holders_{0}, sleepers_{0};
    std::mutex m_, callers_;
    std::condition_variable cv_;
};

}  // namespace rpth
