// What one "launch a kernel, see its result on the host" round trip costs on this box -- the floor under every per-iterate
// rls_*_step_status call: chains of 1..4 dependent, (nearly) empty kernels on one stream, the last one
// storing a sequence word into pinned host-mapped memory the host spins on; the same with hipStreamSynchronize.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void link_kernel(unsigned* d) { if (threadIdx.x == 0) d[0] += 1; }
__global__ void last_kernel(unsigned* d, unsigned* seq_h, unsigned seq) {
  if (threadIdx.x == 0) {
    d[0] += 1;
    __hip_atomic_store(seq_h, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
int main() {
  unsigned *d, *seq_h;
  CK(hipMalloc(&d, 64));
  CK(hipMemset(d, 0, 64));
  CK(hipHostMalloc(&seq_h, 64, hipHostMallocMapped | hipHostMallocCoherent));
  *seq_h = 0;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned seq = 0;
  for (int mode = 0; mode < 2; ++mode)
    for (int chain = 1; chain <= 4; ++chain) {
      double best = 1e9, sum = 0;
      const int reps = 2000;
      for (int r = 0; r < reps + 200; ++r) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k + 1 < chain; ++k) hipLaunchKernelGGL(link_kernel, dim3(1), dim3(64), 0, st, d);
        ++seq;
        hipLaunchKernelGGL(last_kernel, dim3(1), dim3(64), 0, st, d, seq_h, seq);
        if (mode == 0) {
          while (*(volatile unsigned*)seq_h != seq) __builtin_ia32_pause();
        } else {
          CK(hipStreamSynchronize(st));
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (r >= 200) { best = std::min(best, us); sum += us; }
      }
      printf("%s, %d dependent kernel(s): %.1f us mean, %.1f us best\n", mode == 0 ? "mailbox spin" : "hipStreamSynchronize", chain, sum / reps, best);
    }
  return 0;
}
