// Host-side concurrency of the library under ThreadSanitizer / AddressSanitizer, on the CPU (no device layer: the phases'
// bodies, the pinned allocator and the chain's stream switches are stand-ins).  Built and run by
// tests/test_host_concurrency.py:   g++ -std=c++17 -O1 -g -fsanitize=thread   tests/host_concurrency.cpp -lpthread
//                                   g++ -std=c++17 -O1 -g -fsanitize=address  ...
// What it drives (regularizedleastsquares.jl_amd/csrc/host_pool.hpp, the code comm.hip / api.hip / solvers.hip use):
//   1. the per-rank worker pool and its sense-reversing barrier: 8 workers x 10^4 rounds, a per-rank counter that every other
//      rank reads right behind the barrier (a race if the barrier leaks), runs with a failure injected in every phase position
//      (the "ranks keep each other company" path), and a pool stop / start between runs;
//   2. the pinned free list from 8 threads (small blocks recycled, large ones returned; every block freed exactly once);
//   3. the resident-chain bookkeeping from 8 "streams".
#include "../regularizedleastsquares.jl_amd/csrc/host_pool.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>

static int fails = 0;
#define CHECK(cond)                                                    \
  do {                                                                 \
    if (!(cond)) {                                                     \
      std::printf("CHECK failed: %s (line %d)\n", #cond, __LINE__);    \
      ++fails;                                                         \
    }                                                                  \
  } while (0)

static void test_pool(int n, int rounds) {
  comm_pool* P = nullptr;
  std::vector<long> counter(n, 0);  // rank r's slot: written by r in phase A, read by everyone in phase B
  std::vector<long> seen_bad(n, 0);
  std::atomic<int> started{0};
  auto on_start = [&](int) { started.fetch_add(1); };
  for (int fail_phase = -1; fail_phase < 3; ++fail_phase) {  // -1: no failure; 0..2: the body of that phase fails on rank 3 at rep 7
    std::fill(counter.begin(), counter.end(), 0);
    std::vector<rls_comm_phase> ph(3);
    ph[0].run = [&, fail_phase](int r, int k) -> int32_t {  // "publish"
      counter[r] = k + 1;
      return (fail_phase == 0 && r == 3 % n && k == 7) ? 42 : 0;
    };
    ph[0].barrier_after = true;
    ph[1].run = [&, fail_phase](int r, int k) -> int32_t {  // "collect": everyone's publish of this round is visible
      for (int t = 0; t < n; ++t)
        if (counter[t] != k + 1) ++seen_bad[r];
      return (fail_phase == 1 && r == 3 % n && k == 7) ? 43 : 0;
    };
    ph[1].barrier_after = true;  // nobody starts round k + 1's publish while someone still reads round k
    ph[2].run = [&, fail_phase](int r, int k) -> int32_t { return (fail_phase == 2 && r == 3 % n && k == 7) ? 44 : 0; };
    ph[2].barrier_after = false;
    const int32_t st = pool_run(P, n, ph, rounds, on_start);
    CHECK(st == (fail_phase < 0 ? 0 : 42 + fail_phase));
    if (fail_phase < 0)
      for (int r = 0; r < n; ++r) CHECK(counter[r] == rounds);
    for (int r = 0; r < n; ++r) CHECK(seen_bad[r] == 0);
    double busy = 0;
    for (int r = 0; r < n; ++r) busy += P->busy_s[r];
    CHECK(busy >= 0.0);
    if (fail_phase == 0) pool_stop(P);  // stop / start between runs: the next run creates fresh workers
  }
  pool_stop(P);
  CHECK(P == nullptr);
  CHECK(started.load() == 2 * n);
}

static void test_pinned(int nthreads, int iters) {
  pinned_cache cache;
  std::atomic<long> live{0}, allocs{0}, frees{0};
  auto raw_alloc = [&](size_t nbytes) -> void* {
    live.fetch_add(1);
    allocs.fetch_add(1);
    return std::malloc(nbytes);
  };
  auto raw_free = [&](void* p) {
    live.fetch_sub(1);
    frees.fetch_add(1);
    std::free(p);
  };
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; ++t)
    th.emplace_back([&, t] {
      unsigned s = 12345u + 977u * (unsigned)t;
      std::vector<std::pair<void*, size_t>> mine;
      for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        const size_t bytes = (s >> 8) % 3 == 0 ? 8192 + (s >> 12) % 60000 : 16 + (s >> 12) % 3000;  // a third are "ADMM logs"
        void* p = cache.get(bytes, raw_alloc);
        std::memset(p, t, bytes);  // ASan: the block really is that large
        mine.emplace_back(p, bytes);
        if (mine.size() > 8) {
          cache.put(mine.front().first, raw_free);
          mine.erase(mine.begin());
        }
      }
      for (auto& m : mine) cache.put(m.first, raw_free);
    });
  for (auto& t : th) t.join();
  // what is still cached: only small classes, bounded per class; give it back
  long cached = 0;
  for (auto& kv : cache.free_) {
    CHECK(kv.first <= RLS_PIN_KEEP_MAX);
    CHECK(kv.second.size() <= RLS_PIN_KEEP_PER_CLASS);
    for (void* p : kv.second) {
      ++cached;
      raw_free(static_cast<char*>(p) - RLS_PIN_HDR);
    }
  }
  CHECK(live.load() == 0);
  CHECK(frees.load() == allocs.load());
  std::printf("pinned cache: %ld raw allocations for %d requests, %ld blocks cached at the end\n", allocs.load(), nthreads * iters, cached);
}

static void test_chain(int nthreads, int iters) {
  std::mutex mu;
  resident_chain_state st;
  long launches = 0, switches = 0;  // guarded by mu (inside the step)
  void* last_seen = nullptr;
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; ++t)
    th.emplace_back([&, t] {
      void* stream = reinterpret_cast<void*>((uintptr_t)(0x1000 + 16 * t));
      for (int i = 0; i < iters; ++i) {
        const int32_t rc = resident_chain_step(
            mu, st, /*device*/ t % 2, stream,
            [&](void* prev) -> int32_t {
              if (prev == stream) ++fails;  // a switch is only announced when the stream really changes
              ++switches;
              return 0;
            },
            [&]() -> int32_t {
              ++launches;
              last_seen = stream;
              return 0;
            });
        if (rc != 0) ++fails;
        if (i % 97 == 0) resident_chain_forget(mu, st, t % 2, stream);
      }
    });
  for (auto& t : th) t.join();
  CHECK(launches == (long)nthreads * iters);
  std::printf("resident chain: %ld launches, %ld stream switches\n", launches, switches);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 10000;
  test_pool(8, rounds);
  test_pool(2, rounds / 4);
  test_pinned(8, 4000);
  test_chain(8, 20000);
  std::printf(fails ? "FAILED (%d)\n" : "host concurrency OK\n", fails);
  return fails ? 1 : 0;
}
