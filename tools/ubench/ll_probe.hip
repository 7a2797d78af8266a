// probe: does a workgroup that POLLS 16-byte pieces {payload, tag, payload, tag} with sc1 loads see what another workgroup publishes
// with sc1 stores, and after how long?  (the hand-off without a separate flag word)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ static inline u4 poll16(const float* base, uint32_t off) {  // two 8-byte atomic loads: the optimiser must leave them in the loop
  const unsigned long long* p = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(base) + off);
  const unsigned long long a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return u4{(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(512) void probe(float* buf, unsigned* out, int rounds) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 0xffffffff, 0x00020000);
  const int tid = threadIdx.x;
  const uint32_t off = (uint32_t)tid * 16u;
  unsigned fails = 0;
  unsigned long long t_total = 0;
  for (int r = 1; r <= rounds; ++r) {
    const unsigned tag = 0x40000000u + (unsigned)r;   // a normal float's bit pattern
    if (blockIdx.x == 0) {                            // publisher (after a pause, so that the reader polls stale data first)
      __builtin_amdgcn_s_sleep(100);
      u4 v = {(unsigned)r, tag, (unsigned)r + 7u, tag};
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
      // wait for the reader's acknowledgement (same protocol, the other direction)
      u4 a;
      unsigned n = 0;
      do {
        a = poll16(buf, 8192u + off);
      } while ((a[1] != tag || a[3] != tag) && ++n < 4000000u);
      fails += n >= 4000000u;
    } else {
      const unsigned long long t0 = wall_clock64();
      u4 a;
      unsigned n = 0;
      do {
        a = poll16(buf, off);
      } while ((a[1] != tag || a[3] != tag) && ++n < 4000000u);
      t_total += wall_clock64() - t0;
      fails += n >= 4000000u || a[0] != (unsigned)r || a[2] != (unsigned)r + 7u;
      u4 v = {a[0], tag, a[2], tag};
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, 8192u + off, 0, 16);
    }
  }
  atomicAdd(&out[blockIdx.x], fails);
  if (blockIdx.x == 1 && tid == 0) out[2] = (unsigned)(t_total / (unsigned long long)rounds);
}
int main() {
  float* buf; unsigned* out;
  CK(hipMalloc(&buf, 16384)); CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 0, 16384)); CK(hipMemset(out, 0, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rounds = 2000;
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(probe, dim3(2), dim3(512), 0, 0, buf, out, rounds);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned h[4]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
  printf("ping-pong of 8 KiB tagged pieces between two workgroups: %.2f us per round trip, failures %u / %u, reader wait %u ticks (10 ns)\n", ms * 1e3 / rounds, h[0], h[1], h[2]);
  return 0;
}
