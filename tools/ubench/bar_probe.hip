// probe: can the CPU store directly into device memory (large BAR), and how long does a GPU thread take to see it
// compared with polling pinned host memory?  usage: ./bar_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <csignal>
#include <csetjmp>
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
__global__ void echo_kernel(volatile unsigned* cmd, volatile unsigned* ack, int n) {
  unsigned want = 1;
  for (int i = 0; i < n; ++i, ++want) {
    for (;;) {
      unsigned v = __hip_atomic_load((unsigned*)cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (v == want) break;
      __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_store((unsigned*)ack, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
static double round_trips(unsigned* cmd_dev_view, volatile unsigned* cmd_host_view, unsigned* ack_h, int n) {
  *cmd_host_view = 0;
  *ack_h = 0;
  hipLaunchKernelGGL(echo_kernel, dim3(1), dim3(1), 0, 0, cmd_dev_view, ack_h, n);
  auto t0 = std::chrono::steady_clock::now();
  for (unsigned k = 1; k <= (unsigned)n; ++k) {
    *cmd_host_view = k;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    while (*(volatile unsigned*)ack_h != k) {}
  }
  auto t1 = std::chrono::steady_clock::now();
  hipDeviceSynchronize();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}
int main() {
  unsigned* ack_h = nullptr;
  hipHostMalloc(&ack_h, 4096, hipHostMallocDefault);
  unsigned* cmd_h = nullptr;
  hipHostMalloc(&cmd_h, 4096, hipHostMallocDefault);
  printf("pinned host command word : %.2f us per round trip (GPU polls over the link)\n", round_trips(cmd_h, cmd_h, ack_h, 2000));
  unsigned* cmd_d = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&cmd_d, 4096, hipDeviceMallocFinegrained);
  printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
  if (e == hipSuccess) {
    hipMemset(cmd_d, 0, 4096);
    hipDeviceSynchronize();
    signal(SIGSEGV, on_segv);
    signal(SIGBUS, on_segv);
    if (sigsetjmp(jb, 1) == 0) {
      volatile unsigned* hv = cmd_d;
      unsigned v = *hv;  // CPU load from device memory
      printf("CPU read of device memory ok (%u)\n", v);
      *hv = 0;
      printf("CPU write to device memory ok\n");
      printf("device-memory command word: %.2f us per round trip (CPU stores through the BAR, GPU polls its own memory)\n",
             round_trips(cmd_d, hv, ack_h, 2000));
    } else {
      printf("CPU access to device memory faults: no host-visible device memory on this system\n");
    }
  }
  return 0;
}
