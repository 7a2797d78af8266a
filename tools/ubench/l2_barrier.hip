// micro-benchmark (measurement only, not the product): the GROUP barrier of the two-level exchange -- 32 workgroups that share one
// XCD's L2 -- with its arrival counter at AGENT scope (the shipped form: the atomic and the polls travel to the memory side) against
// WORKGROUP scope (the atomic is performed in the XCD's L2, the polls are L1-bypassing loads that hit there).  Each round every
// member stores the round number into its slot (sc0: stops in the L2) ahead of its arrival and checks all 32 slots behind the
// barrier: `bad` counts stale slots, `fail` timeouts.  Workgroups whose XCC_ID is not blockIdx % 8 are counted (`misplaced`).
//   hipcc -O3 --offload-arch=gfx950 l2_barrier.hip -o l2_barrier && ./l2_barrier [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NT = 512;
constexpr unsigned SPIN = 200000;
struct blk { unsigned cnt[8 * 32]; unsigned slots[8 * 32 * 32]; unsigned fail, bad, misplaced; };

__device__ static inline unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xfu; }

template <int MODE>  // 0: agent-scope counter; 1: workgroup-scope atomic + sc1 polls; 2: workgroup-scope atomic + sc0 polls; 3: workgroup scope both (compiler's choice)
__global__ __launch_bounds__(NT) void gbar(blk* S, int rounds) {
  __shared__ int flag;
  const unsigned members = gridDim.x / 8, g = blockIdx.x & 7, me = blockIdx.x >> 3;
  unsigned* word = S->cnt + g * 32;
  unsigned* slots = S->slots + g * 32 * 32;
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(word, 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(slots, 0, 0xffffffff, 0x00020000);
  if (threadIdx.x == 0 && xcc_id() != g) atomicAdd(&S->misplaced, 1u);
  for (int r = 1; r <= rounds; ++r) {
    if (threadIdx.x == 0) {
      __builtin_amdgcn_raw_buffer_store_b32((unsigned)r, sr, me * 128u, 0, 1);  // sc0: stops in this XCD's L2
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      int ok = 0;
      const unsigned target = members * (unsigned)r;
      if (threadIdx.x == 0) {
        if (MODE == 0) __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      for (unsigned spins = 0; spins < SPIN; ++spins) {
        unsigned c;
        if (MODE == 0) c = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1) c = __builtin_amdgcn_raw_buffer_load_b32(wr, 0, 0, 16);
        else if (MODE == 2) c = __builtin_amdgcn_raw_buffer_load_b32(wr, 0, 0, 1);
        else c = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__builtin_amdgcn_readfirstlane((int)c) >= (int)target) { ok = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (threadIdx.x == 0) flag = ok;
    }
    __syncthreads();
    if (!flag) { if (threadIdx.x == 0) S->fail = 1; return; }
    if (threadIdx.x < members) {
      const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(sr, threadIdx.x * 128u, 0, 16);  // sc1: bypasses this CU's L1
      if (v < (unsigned)r) atomicAdd(&S->bad, 1u);
    }
    __syncthreads();
  }
}

template <int MODE>
static void run(blk* S, int rounds, const char* what) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(S, 0, sizeof(blk)));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(gbar<MODE>, dim3(256), dim3(NT), 0, 0, S, rounds);
    CK(hipGetLastError());
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    blk h; CK(hipMemcpy(&h, S, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-58s %7.3f us per round   fail %u  stale slots %u  misplaced %u\n", what, ms * 1e3 / rounds, h.fail, h.bad, h.misplaced);
  }
}
int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  blk* S; CK(hipMalloc(&S, sizeof(blk)));
  run<0>(S, rounds, "agent-scope atomic, agent-scope polls (shipped)");
  run<1>(S, rounds, "workgroup-scope atomic (in the L2), sc1 polls");
  run<2>(S, rounds, "workgroup-scope atomic (in the L2), sc0 polls");
  run<3>(S, rounds, "workgroup-scope atomic and polls (compiler's encoding)");
  return 0;
}
