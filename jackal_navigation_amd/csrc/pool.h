// pool.h — task pool for the host stage (product code, no HIP): a fixed set of threads, each owning a
// HostWorker; several callers (the slot workers of one handle) may submit groups of tasks concurrently.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <pthread.h>
#include "host_stage.h"

namespace jnav {

// a polite spin: x86 `pause`, a yield elsewhere (this header is plain host code, also built by the thread-sanitizer test)
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}

class Pool {
 public:
  // spin_us > 0 (latency-mode handles): a worker that has just run a task polls the queue for that long before it goes back to sleep,
  // and run() polls for its group's completion before it blocks — a futex wake-up costs 20-30 us, which is a tenth of a lone pair's
  // call; a batch pipeline has no use for it (its stages last milliseconds) and keeps the container's CPU quota for the triangulations.
  Pool(int threads, const HostParams& hp, int spin_us = 0) : spin_us_(spin_us) {
    for (int i = 0; i < threads; i++) workers_.emplace_back(new HostWorker(hp));
    for (int i = 0; i < threads; i++) threads_.emplace_back([this, i] { loop(i); });
  }
  ~Pool() {
    { std::lock_guard<std::mutex> l(m_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  // Runs fn(worker, i) for i in [0,n) on the pool and blocks until all are done.  Several callers
  // (slot workers) may use the pool at the same time.
  void run(int n, const std::function<void(HostWorker&, int)>& fn) {
    if (n <= 0) return;
    Group g; g.fn = &fn; g.left = n; g.done.store(false, std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> l(m_);
      for (int i = 0; i < n; i++) q_.push_back({&g, i});
      queued_.fetch_add(n, std::memory_order_release);
    }
    if (n >= (int)threads_.size()) cv_.notify_all();
    else for (int i = 0; i < n; i++) cv_.notify_one();      // wake only as many workers as there are tasks (spinning ones need no wake-up)
    if (spin_us_ > 0) {
      // The caller polls for its group only while that can pay: at most 4 x spin_us (a lone pair's host stage is two ~60 us tasks), and not
      // at all once every task has been picked up and the deadline of one task's length has passed — a long task is then cheaper to sleep
      // through (the caller's core is part of the container's CPU quota, which the host stage itself needs).
      const auto t0 = std::chrono::steady_clock::now();
      const auto limit = std::chrono::microseconds(4 * spin_us_), picked_up_limit = std::chrono::microseconds(spin_us_);
      while (!g.done.load(std::memory_order_acquire)) {
        const auto waited = std::chrono::steady_clock::now() - t0;
        if (waited >= limit || (waited >= picked_up_limit && queued_.load(std::memory_order_acquire) == 0)) break;
        cpu_relax();
      }
    }
    std::unique_lock<std::mutex> l(g.m);
    g.cv.wait(l, [&g] { return g.left == 0; });
  }
  int size() const { return (int)threads_.size(); }

 private:
  struct Group { const std::function<void(HostWorker&, int)>* fn; int left; std::mutex m; std::condition_variable cv; std::atomic<bool> done; };
  struct Item { Group* g; int i; };
  void loop(int id) {
    pthread_setname_np(pthread_self(), "jn-pool");          // visible in /proc/<pid>/task/*/comm: CPU accounting per role
    bool hot = false;                                        // has run a task since it last slept
    for (;;) {
      Item it;
      if (hot && spin_us_ > 0) {                             // poll for the next task before going back to sleep
        const auto t0 = std::chrono::steady_clock::now();
        while (queued_.load(std::memory_order_acquire) == 0 && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(spin_us_)) cpu_relax();
      }
      {
        std::unique_lock<std::mutex> l(m_);
        if (q_.empty()) hot = false;
        cv_.wait(l, [this] { return stop_ || !q_.empty(); });
        if (stop_ && q_.empty()) return;
        it = q_.front(); q_.pop_front();
        queued_.fetch_sub(1, std::memory_order_relaxed);
      }
      (*it.g->fn)(*workers_[id], it.i);
      hot = true;
      std::lock_guard<std::mutex> l(it.g->m);
      if (--it.g->left == 0) { it.g->done.store(true, std::memory_order_release); it.g->cv.notify_all(); }
    }
  }
  std::vector<std::unique_ptr<HostWorker>> workers_;
  std::vector<std::thread> threads_;
  std::deque<Item> q_;
  std::mutex m_;
  std::condition_variable cv_;
  bool stop_ = false;
  std::atomic<int> queued_{0};                               // tasks in q_ (what a spinning worker polls)
  const int spin_us_;
};


}  // namespace jnav
