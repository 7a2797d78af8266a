// Microbenchmark: a grid-wide reduce-scatter of LARGE per-workgroup partials through memory inside one launch -- the
// exchange a register-resident batched solver iteration would need (K right-hand sides: every workgroup produces an
// N x 16-float partial of V = A^H T, 128 KiB at N = 2048, and workgroup j needs column block j of all of them).
//   per round:  every workgroup stores PB bytes (its partial) | grid barrier | workgroup j reads PB / nwg bytes of each of
//               the nwg partials, sums them in a fixed order and stores the block | grid barrier
// Variants: sc1 (write-through) stores, or plain stores followed by a release fence (L2 write-back) before the arrival.
// Checks its sums (integer-valued floats) and reports us per round; usage: bulk_exchange [rounds=300] [KiB per workgroup=128]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                            \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                           \
    }                                                                    \
  } while (0)

constexpr int NT = 512;
constexpr unsigned SPIN = 4000000u;

struct sync_block {
  unsigned cnt[8 * 32];
  unsigned fail, bad;
};

__device__ static inline __amdgpu_buffer_rsrc_t rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xffffffff, 0x00020000);
}

__device__ static inline bool arrive_wait(unsigned* cnt, unsigned target, int* lds_flag) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    int ok = 0;
    if (tid == 0) __hip_atomic_fetch_add(cnt + (blockIdx.x & 7) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned spins = 0; spins < SPIN; ++spins) {
      unsigned c = __hip_atomic_load(cnt + (tid & 7) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      c += (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0xB1, 0xF, 0xF, true);
      c += (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0x4E, 0xF, 0xF, true);
      c += (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0x141, 0xF, 0xF, true);
      if (__builtin_amdgcn_readfirstlane((int)c) >= (int)target) {
        ok = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (tid == 0) *lds_flag = ok;
  }
  __syncthreads();
  return *lds_flag != 0;
}

// MODE 0: sc1 stores + sc1 loads; MODE 1: plain stores, release fence (agent) before the arrival, sc1 loads;
// MODE 2: as 0, but only the barriers and the stores (no read phase) ; MODE 3: as 0 without the stores (reads only)
template <int MODE>
__global__ __launch_bounds__(NT) void bulk_kernel(sync_block* S, float* parts, float* out, int pb, int rounds) {
  __shared__ int flag;
  __shared__ f4 red[NT];
  const int tid = threadIdx.x;
  const unsigned nwg = gridDim.x;
  const __amdgpu_buffer_rsrc_t p_rs = rsrc(parts), o_rs = rsrc(out);
  const int stores = pb / 16 / NT;       // b128 stores per thread (16 at 128 KiB)
  const int blk = pb / (int)nwg;         // bytes of every partial this workgroup owns (512 at 128 KiB, 256 workgroups)
  const int pieces = blk / 16;           // 32
  const int groups = NT / pieces;        // 16 thread groups, each sums nwg / groups partials
  const int per = (int)nwg / groups;     // 16
  unsigned epoch = 0, bad = 0;
  float want0 = 0.f;
  for (unsigned b = 0; b < nwg; ++b) want0 += (float)(b % 7);
  for (int r = 1; r <= rounds; ++r) {
    const float val = (float)((blockIdx.x % 7) + (r % 5));
    if (MODE != 3) {
      for (int q = 0; q < stores; ++q) {
        const uint32_t off = (uint32_t)blockIdx.x * (uint32_t)pb + (uint32_t)(q * NT + tid) * 16u;
        if (MODE == 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, f4{val, val, val, val}), p_rs, off, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, f4{val, val, val, val}), p_rs, off, 0, 16);
      }
      if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (!arrive_wait(S->cnt, nwg * ++epoch, &flag)) break;
    if (MODE != 2) {
      const int piece = tid % pieces, grp = tid / pieces;
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      if (grp < groups) {
#pragma unroll 4
        for (int k = 0; k < per; ++k) {
          const uint32_t row = (uint32_t)(grp * per + k);
          const f4 t = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(p_rs, row * (uint32_t)pb + (uint32_t)blockIdx.x * (uint32_t)blk + (uint32_t)piece * 16u, 0, 16));
          acc += t;
        }
      }
      red[tid] = acc;
      __syncthreads();
      if (tid < pieces) {
        f4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < groups; ++g) sum += red[g * pieces + tid];
        const float want = want0 + (float)nwg * (float)(r % 5);
        if (MODE == 0 || MODE == 1)
          if (sum.x != want || sum.y != want || sum.z != want || sum.w != want) ++bad;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, sum), o_rs, (uint32_t)blockIdx.x * (uint32_t)blk + (uint32_t)tid * 16u, 0, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (!arrive_wait(S->cnt, nwg * ++epoch, &flag)) break;
  }
  if (epoch != 2u * (unsigned)rounds && tid == 0) S->fail = 1;
  if (bad) atomicAdd(&S->bad, bad);
}

template <int MODE>
static void run(const char* name, sync_block* S, float* parts, float* out, int pb, int rounds, int nwg, hipEvent_t e0, hipEvent_t e1) {
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(S, 0, sizeof(sync_block)));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(bulk_kernel<MODE>, dim3(nwg), dim3(NT), 140 * 1024, 0, S, parts, out, pb, rounds);
    CK(hipGetLastError());
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    sync_block h;
    CK(hipMemcpy(&h, S, sizeof(h), hipMemcpyDeviceToHost));
    const double us = ms * 1e3 / rounds;
    printf("%-28s %8.3f us per round  (%.2f TB/s of partial bytes each way)  fail %u  wrong sums %u\n", name, us,
           (double)nwg * pb / us * 1e-6, h.fail, h.bad);
  }
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 300;
  const int kib = argc > 2 ? atoi(argv[2]) : 128;
  const int pb = kib * 1024;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int nwg = prop.multiProcessorCount;
  printf("device %s, %d CUs; %d rounds, %d KiB per workgroup (%.1f MiB per round each way)\n", prop.name, nwg, rounds, kib,
         (double)nwg * pb / 1048576.0);
  sync_block* S;
  float *parts, *out;
  CK(hipMalloc(&S, sizeof(sync_block)));
  CK(hipMalloc(&parts, (size_t)nwg * pb));
  CK(hipMalloc(&out, pb));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bulk_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bulk_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bulk_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bulk_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  run<0>("sc1 stores, sc1 loads", S, parts, out, pb, rounds, nwg, e0, e1);
  run<1>("plain stores + release fence", S, parts, out, pb, rounds, nwg, e0, e1);
  run<2>("stores + barriers only", S, parts, out, pb, rounds, nwg, e0, e1);
  run<3>("loads + barriers only", S, parts, out, pb, rounds, nwg, e0, e1);
  return 0;
}
