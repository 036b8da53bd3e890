// host_pool.h — one persistent host thread per device of a group (csrc/group.hip).  The reference is one C++ node with one stepping thread
// (src/mj_main.cpp:203); sharded over the GPUs of a node, that one thread would issue every device's launches one after the other, and
// for light scenes (C5: 87 us of device time per step) eight devices' worth of launch calls is as long as the step itself.  With a
// thread per device the host time of a call is that of ONE device.  Plain C++ (no HIP): tests/tsan/pool_threads.cpp runs it under
// ThreadSanitizer.
//
// run(fn): fn(k) on worker k for every k, returns when all are done (the caller's thread only posts and waits: 200 us spinning, then asleep).  A worker spins for a
// short while after a job (calls of a stepping loop come back to back) before it blocks on its condition variable.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

class HostPool {
 public:
  // init(k) runs once on worker k before its first job (e.g. hipSetDevice); last_error: the calling convention's per-thread error text
  HostPool(int n, std::function<void(int)> init, const char* (*last_error)()) : last_error_(last_error) {
    for (int k = 0; k < n; k++) w_.emplace_back(new W());
    for (int k = 0; k < n; k++) w_[k]->th = std::thread([this, k, init] { if (init) init(k); loop(k); });
  }
  ~HostPool() {
    for (auto& w : w_) { { std::lock_guard<std::mutex> l(w->m); w->quit = true; w->state.store(1, std::memory_order_release); } w->cv.notify_one(); }
    for (auto& w : w_) if (w->th.joinable()) w->th.join();
  }
  int size() const { return (int)w_.size(); }
  // first non-zero result in worker order; *err receives that worker's error text
  int run(const std::function<int(int)>& fn, std::string* err) {
    job_ = &fn;
    pending_.store((int)w_.size(), std::memory_order_release);
    for (auto& w : w_) {
      { std::lock_guard<std::mutex> l(w->m); w->state.store(1, std::memory_order_release); }
      w->cv.notify_one();
    }
    // short jobs (a step's launches: microseconds) are met spinning; a blocking job (mjh_group_synchronize: milliseconds of GPU time)
    // puts the caller to sleep on the completion count instead of burning a core
    const auto t0 = std::chrono::steady_clock::now();
    bool done = false;
    for (int i = 0; !done; i++) {
      if (pending_.load(std::memory_order_acquire) == 0) { done = true; break; }
      if ((i & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
    }
    if (!done) { std::unique_lock<std::mutex> l(dm_); dcv_.wait(l, [&] { return pending_.load(std::memory_order_acquire) == 0; }); }
    int rc = 0;
    for (auto& w : w_) {
      while (w->state.load(std::memory_order_acquire) != 2) std::this_thread::yield();     // (already there: pending_ reached 0 after every state went to 2)
      if (w->rc && !rc) { rc = w->rc; if (err) *err = w->err; }
      w->state.store(0, std::memory_order_relaxed);
    }
    job_ = nullptr;
    return rc;
  }

 private:
  struct W {
    std::thread th; std::mutex m; std::condition_variable cv;
    std::atomic<int> state{0};      // 0 idle, 1 job posted (or quit), 2 done
    int rc = 0; std::string err; bool quit = false;
  };
  void loop(int k) {
    W& w = *w_[k];
    for (;;) {
      // spin ~50 us for the next job, then sleep
      bool got = false;
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; !got; i++) {
        if (w.state.load(std::memory_order_acquire) == 1) { got = true; break; }
        if ((i & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(50)) break;
      }
      if (!got) { std::unique_lock<std::mutex> l(w.m); w.cv.wait(l, [&] { return w.state.load(std::memory_order_acquire) == 1; }); }
      { std::lock_guard<std::mutex> l(w.m); if (w.quit) return; }
      w.rc = (*job_)(k);
      if (w.rc && last_error_) w.err = last_error_(); else w.err.clear();
      w.state.store(2, std::memory_order_release);
      if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> l(dm_); dcv_.notify_one(); }
    }
  }
  std::vector<std::unique_ptr<W>> w_;
  const std::function<int(int)>* job_ = nullptr;      // written before the workers' state goes to 1 (release), read after they see it (acquire)
  const char* (*last_error_)() = nullptr;
  std::atomic<int> pending_{0}; std::mutex dm_; std::condition_variable dcv_;     // workers still on the current job; the caller sleeps on it after a short spin
};
