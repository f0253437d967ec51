// pool.h — task pool for the host stage (product code, no HIP): a fixed set of threads, each owning a
// HostWorker; several callers (the slot workers of one handle) may submit groups of tasks concurrently.
#pragma once
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

class Pool {
 public:
  Pool(int threads, const HostParams& hp) {
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
    Group g; g.fn = &fn; g.left = n;
    {
      std::lock_guard<std::mutex> l(m_);
      for (int i = 0; i < n; i++) q_.push_back({&g, i});
    }
    if (n >= (int)threads_.size()) cv_.notify_all();
    else for (int i = 0; i < n; i++) cv_.notify_one();      // wake only as many workers as there are tasks
    std::unique_lock<std::mutex> l(g.m);
    g.cv.wait(l, [&g] { return g.left == 0; });
  }
  int size() const { return (int)threads_.size(); }

 private:
  struct Group { const std::function<void(HostWorker&, int)>* fn; int left; std::mutex m; std::condition_variable cv; };
  struct Item { Group* g; int i; };
  void loop(int id) {
    pthread_setname_np(pthread_self(), "jn-pool");          // visible in /proc/<pid>/task/*/comm: CPU accounting per role
    for (;;) {
      Item it;
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [this] { return stop_ || !q_.empty(); });
        if (stop_ && q_.empty()) return;
        it = q_.front(); q_.pop_front();
      }
      (*it.g->fn)(*workers_[id], it.i);
      std::lock_guard<std::mutex> l(it.g->m);
      if (--it.g->left == 0) it.g->cv.notify_all();
    }
  }
  std::vector<std::unique_ptr<HostWorker>> workers_;
  std::vector<std::thread> threads_;
  std::deque<Item> q_;
  std::mutex m_;
  std::condition_variable cv_;
  bool stop_ = false;
};


}  // namespace jnav
