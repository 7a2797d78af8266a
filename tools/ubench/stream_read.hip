// Microbenchmark: how fast can the chip re-read a buffer of a given size (MALL-resident vs HBM)?
// usage: stream_read <MiB> <threads> <loads_per_thread_in_flight> <reps>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int U>
__global__ void rd(const f4* __restrict__ a, size_t n16, float* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  f4 acc = {0, 0, 0, 0};
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = a[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  for (; i < n16; i += stride) acc += a[i];
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
// contiguous-per-workgroup variant: each WG reads one contiguous slab
template <int U>
__global__ void rd_slab(const f4* __restrict__ a, size_t n16, float* out) {
  const size_t per = n16 / gridDim.x;
  const f4* p = a + per * blockIdx.x;
  f4 acc = {0, 0, 0, 0};
  for (size_t i = threadIdx.x; i + (U - 1) * blockDim.x < per; i += U * blockDim.x) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[i + u * blockDim.x];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

int main(int argc, char** argv) {
  std::vector<size_t> sizes = {16, 32, 64, 128, 256, 512, 2048};
  float* out; CK(hipMalloc(&out, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (size_t mib : sizes) {
    size_t bytes = mib << 20, n16 = bytes / 16;
    f4* a; CK(hipMalloc(&a, bytes)); CK(hipMemset(a, 1, bytes));
    struct Cfg { int grid, threads, u, slab; };
    std::vector<Cfg> cfgs = {{256, 1024, 16, 1}, {512, 1024, 8, 1}, {1024, 256, 8, 0}, {2048, 256, 8, 0}, {2048, 256, 16, 0}, {4096, 256, 8, 0}, {512, 512, 16, 1}, {2048,256,8,1}, {1024, 1024, 4, 1}};
    for (auto c : cfgs) {
      auto launch = [&]() {
        if (c.slab) { if (c.u == 16) rd_slab<16><<<c.grid, c.threads>>>(a, n16, out); else if (c.u == 8) rd_slab<8><<<c.grid, c.threads>>>(a, n16, out); else rd_slab<4><<<c.grid, c.threads>>>(a, n16, out); }
        else { if (c.u == 16) rd<16><<<c.grid, c.threads>>>(a, n16, out); else rd<8><<<c.grid, c.threads>>>(a, n16, out); }
      };
      for (int i = 0; i < 5; ++i) launch();
      CK(hipDeviceSynchronize());
      int reps = mib >= 512 ? 20 : 200;
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double us = ms * 1e3 / reps;
      printf("%5zu MiB grid %5d thr %4d U %2d slab %d : %8.2f us/launch  %7.1f GB/s\n", mib, c.grid, c.threads, c.u, c.slab, us, bytes / us / 1e3);
    }
    CK(hipFree(a));
  }
  return 0;
}
