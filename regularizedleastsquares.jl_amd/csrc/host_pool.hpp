// Host-side concurrency of the library, free of any device API so that it also compiles with plain g++ under
// -fsanitize=thread / -fsanitize=address (tests/host_concurrency.cpp, run by tests/test_host_concurrency.py):
//   * the per-rank worker pool of the row-sharded solver loops and its sense-reversing spin barrier (comm.hip);
//   * the process-wide free list of small pinned host blocks (api.hip);
//   * the bookkeeping of the resident-launch chain (solvers.hip): which stream issued the last resident kernel of a device.
// The device layer enters only through callables (what a phase enqueues, how a pinned block is obtained, what to do when
// the chain changes streams).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

static inline void rls_host_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield");
#endif
}

// one step of a row-sharded loop: what rank `rank` does in repetition `rep`, and whether every rank must have CALLED it
// before any rank starts the next phase
struct rls_comm_phase {
  std::function<int32_t(int rank, int rep)> run;
  bool barrier_after = true;
};

constexpr int RLS_POOL_MAX_RANKS = 16;

// One host worker thread per rank (the fan-out of src/MultiThreading.jl:60-78, `Threads.@threads`, inside the library):
// a row-sharded solver call hands every worker the WHOLE loop of its rank -- phases separated by a spinning host barrier
// -- so that the host side of an iteration costs what ONE rank's launches cost, not the sum over ranks.
struct comm_pool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv;
  uint64_t gen = 0;
  bool quit = false;
  const std::vector<rls_comm_phase>* phases = nullptr;
  int reps = 0;
  int n = 0;
  std::atomic<int> arrived{0};
  std::atomic<unsigned> sense{0};
  std::atomic<int> finished{0};
  std::atomic<int32_t> status{0};
  double busy_s[RLS_POOL_MAX_RANKS] = {0};  // per rank: wall clock spent INSIDE phase bodies (enqueueing), barriers and idling excluded
};

// sense-reversing barrier of the pool's n workers: the last arriver resets the count and flips the shared sense
static inline void pool_barrier(comm_pool* P, int n, unsigned* my_sense) {
  *my_sense ^= 1u;
  if (P->arrived.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
    P->arrived.store(0, std::memory_order_relaxed);
    P->sense.store(*my_sense, std::memory_order_release);
  } else {
    for (unsigned spins = 0; P->sense.load(std::memory_order_acquire) != *my_sense; ++spins) {
      rls_host_relax();
      if ((spins & 1023u) == 1023u) std::this_thread::yield();  // oversubscribed hosts: let the others run
    }
  }
}

// the worker of rank r: `on_start(r)` once (binds the thread to its device), then one whole run per generation
static inline void pool_worker(comm_pool* P, int r, const std::function<void(int)>& on_start) {
  uint64_t seen = 0;
  if (on_start) on_start(r);
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(P->m);
      P->cv.wait(lk, [&] { return P->quit || P->gen != seen; });
      if (P->quit) return;
      seen = P->gen;
    }
    unsigned my_sense = P->sense.load(std::memory_order_acquire);
    const std::vector<rls_comm_phase>& ph = *P->phases;
    for (int k = 0; k < P->reps; ++k) {
      for (size_t i = 0; i < ph.size(); ++i) {
        if (P->status.load(std::memory_order_relaxed) == 0) {  // after a failure the ranks only keep each other company
          const auto t0 = std::chrono::steady_clock::now();
          const int32_t st = ph[i].run(r, k);
          P->busy_s[r] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          if (st != 0) {
            int32_t zero = 0;
            P->status.compare_exchange_strong(zero, st);
          }
        }
        if (ph[i].barrier_after) pool_barrier(P, P->n, &my_sense);
      }
    }
    if (P->finished.fetch_add(1, std::memory_order_acq_rel) == P->n - 1) {
      std::lock_guard<std::mutex> lk(P->m);
      P->cv.notify_all();
    }
  }
}

static inline void pool_stop(comm_pool*& P) {
  if (!P) return;
  {
    std::lock_guard<std::mutex> lk(P->m);
    P->quit = true;
  }
  P->cv.notify_all();
  for (std::thread& t : P->th) t.join();
  delete P;
  P = nullptr;
}

// reps x (the phases in order) for every one of n ranks, each on its own worker thread (created on first use).  Returns the
// first non-zero status a phase body returned.
static inline int32_t pool_run(comm_pool*& P, int n, const std::vector<rls_comm_phase>& phases, int reps,
                               const std::function<void(int)>& on_start) {
  if (!P) {
    P = new comm_pool();
    P->n = n;
    for (int r = 0; r < n; ++r) P->th.emplace_back(pool_worker, P, r, on_start);
  }
  {
    std::lock_guard<std::mutex> lk(P->m);
    P->phases = &phases;
    P->reps = reps;
    P->finished.store(0);
    P->status.store(0);
    P->arrived.store(0);
    ++P->gen;
  }
  P->cv.notify_all();
  {
    std::unique_lock<std::mutex> lk(P->m);
    P->cv.wait(lk, [&] { return P->finished.load(std::memory_order_acquire) == n; });
  }
  return P->status.load();
}

// ---- free list of small pinned host blocks -----------------------------------------------------------------------------------
// hipHostMalloc costs ~100 us a call, so the status mirrors of the plans are recycled.  Only SMALL blocks are kept (classes of
// 256 bytes up to RLS_PIN_KEEP_MAX, at most RLS_PIN_KEEP_PER_CLASS of a class): larger ones -- an ADMM log, the scalar mirrors
// of a wide batch -- go back to the driver, so a long-running process with varying iteration counts or batch sizes does not grow
// its pinned memory without bound.
constexpr size_t RLS_PIN_HDR = 64;  // keeps the payload 64-byte aligned; holds the class size
constexpr size_t RLS_PIN_KEEP_MAX = 4096;
constexpr size_t RLS_PIN_KEEP_PER_CLASS = 64;
struct pinned_cache {
  std::mutex m;
  std::map<size_t, std::vector<void*>> free_;
  // `raw_alloc(bytes)` returns a block or nullptr; `raw_free(p)` releases one
  template <typename A>
  void* get(size_t bytes, A&& raw_alloc) {
    const size_t cls = (bytes + 255) / 256 * 256;
    {
      std::lock_guard<std::mutex> lk(m);
      auto it = free_.find(cls);
      if (it != free_.end() && !it->second.empty()) {
        void* p = it->second.back();
        it->second.pop_back();
        return p;
      }
    }
    char* raw = static_cast<char*>(raw_alloc(cls + RLS_PIN_HDR));
    if (!raw) return nullptr;
    *reinterpret_cast<size_t*>(raw) = cls;
    return raw + RLS_PIN_HDR;
  }
  template <typename F>
  void put(void* p, F&& raw_free) {
    if (!p) return;
    char* raw = static_cast<char*>(p) - RLS_PIN_HDR;
    const size_t cls = *reinterpret_cast<size_t*>(raw);
    if (cls <= RLS_PIN_KEEP_MAX) {
      std::lock_guard<std::mutex> lk(m);
      std::vector<void*>& fl = free_[cls];
      if (fl.size() < RLS_PIN_KEEP_PER_CLASS) {
        fl.push_back(p);
        return;
      }
    }
    raw_free(raw);
  }
};

// ---- resident-launch chain -----------------------------------------------------------------------------------------------------
// Two resident kernels running side by side (two streams of one process) could each hold CUs the other is waiting for, so the
// resident launches of a device form ONE chain across streams.  A launch on the stream that issued the previous one is ordered
// by the stream itself; only when the stream CHANGES does `on_switch(previous_stream)` run (record an event there, wait for it
// here) before `launch()`.  Everything happens under the caller's lock (the library's capture / chain mutex).
struct resident_chain_state {
  void* last[64] = {nullptr};
};
template <typename S, typename L>
static inline int32_t resident_chain_step(std::mutex& mu, resident_chain_state& st, int device, void* stream, S&& on_switch, L&& launch) {
  std::lock_guard<std::mutex> lock(mu);
  const int d = device < 0 ? 0 : (device < 64 ? device : 63);
  if (st.last[d] && st.last[d] != stream) {
    const int32_t e = on_switch(st.last[d]);
    if (e != 0) return e;
  }
  const int32_t rc = launch();
  st.last[d] = stream;
  return rc;
}
static inline void resident_chain_forget(std::mutex& mu, resident_chain_state& st, int device, void* stream) {
  std::lock_guard<std::mutex> lock(mu);
  const int d = device < 0 ? 0 : (device < 64 ? device : 63);
  if (st.last[d] == stream) st.last[d] = nullptr;
}
