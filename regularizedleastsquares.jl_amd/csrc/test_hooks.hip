// TEST-ONLY hooks, built into their own shared object (librls_test_hooks.so): nothing here is part of the product library or
// of its C ABI (include/rls_mi355x.h).  tests/test_gpu_parity.py loads it for the co-tenancy tests.
#include <hip/hip_runtime.h>

#include <cstdint>

// workgroups that sit on whole CUs (1024 threads, all of the CU's LDS) for a given wall-clock time and do nothing else -- what
// another tenant of the device looks like to a kernel that needs every CU at once (s_memrealtime ticks at 100 MHz)
__global__ __launch_bounds__(1024) void hold_cus_kernel(unsigned long long ticks, unsigned* sink) {
  extern __shared__ char hold_lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (ticks == ~0ull) sink[threadIdx.x] = (unsigned)hold_lds[threadIdx.x];  // never: keeps the LDS allocation alive
}

// enqueue on `stream` (a hipStream_t of `device`, e.g. rls_ctx_stream(ctx)); bounded: <= 2 s.  Returns a hipError_t.
extern "C" int32_t rls_test_hold_cus(void* stream, int32_t device, int32_t n_workgroups, int32_t microseconds) {
  if (n_workgroups <= 0 || microseconds < 0 || microseconds > 2000000) return (int32_t)hipErrorInvalidValue;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return (int32_t)e;
  const size_t lds = 160 * 1024;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(hold_cus_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int32_t)e;
  hipLaunchKernelGGL(hold_cus_kernel, dim3((unsigned)n_workgroups), dim3(1024), lds, (hipStream_t)stream,
                     (unsigned long long)microseconds * 100ull, (unsigned*)nullptr);
  return (int32_t)hipGetLastError();
}
