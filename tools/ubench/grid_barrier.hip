// Microbenchmark: what do the in-kernel grid exchanges of the resident solver kernels cost with NOTHING else in the
// kernel?  (normal.hip, cgnr_resident_kernel: per CGNR iteration two grid-wide exchanges of a 16 KiB vector.)
//
//   bar      : R rounds of the product's arrive + wait (8-shard monotonic counter, sc1 polls) and nothing else
//   flat     : R rounds of the product's two-hop all-reduce of a per-workgroup N-vector, arithmetic stripped:
//              publish a partial row (sc1) | barrier | workgroup j sums 64-byte chunk j over all rows, publishes it and
//              three doubles | barrier | every workgroup reads the vector and the 256 x 3 doubles
//   hier     : the XCD-hierarchical alternative: workgroups enrol in a per-XCD roster (HW_REG_XCC_ID, one atomic each),
//              publish | per-XCD barrier (32 arrivals) | roster member l sums slice l (1/32 of the vector) over its XCD's
//              32 rows and publishes that XCD-partial slice | grid barrier | every workgroup reads the 8 XCD-partial
//              vectors and sums them in XCD order (one full-grid barrier per round instead of two, 8x the final read)
//   hier2    : the same, but the second level is a second reduce-scatter + all-gather (two grid barriers, small reads)
//
// Every variant checks its sums (integer-valued floats: exact) and reports us per round from hipEvents around ONE launch.
// usage: grid_barrier [rounds=2000] [N_elems_of_8_bytes=2048]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);       \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

constexpr int NT = 512;
constexpr unsigned SPIN = 4000000u;

__device__ static inline __amdgpu_buffer_rsrc_t sc1_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xffffffff, 0x00020000);
}
__device__ static inline f4 sc1_load16(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16));
}
__device__ static inline void sc1_store16(__amdgpu_buffer_rsrc_t r, uint32_t off, f4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, off, 0, 16);
}
__device__ static inline float dpp_f(float v, int ctrl) {
  switch (ctrl) {
    case 0xB1: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    case 0x4E: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    case 0x141: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    default: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  }
}

// the product's barrier: `shards` counter words a 128-byte line apart; arrivals by one lane, polls by lanes 0..shards-1
template <int SHARDS>
__device__ static inline bool arrive_wait(unsigned* cnt, unsigned shard, unsigned target, int* lds_flag) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    int ok = 0;
    if (tid == 0) __hip_atomic_fetch_add(cnt + shard * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned spins = 0; spins < SPIN; ++spins) {
      unsigned c = tid < SHARDS ? __hip_atomic_load(cnt + (tid % SHARDS) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      if constexpr (SHARDS >= 2) c += (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0xB1, 0xF, 0xF, true);
      if constexpr (SHARDS >= 4) c += (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0x4E, 0xF, 0xF, true);
      if constexpr (SHARDS >= 8) c += (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0x141, 0xF, 0xF, true);
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

struct sync_block {
  unsigned cnt[8 * 32];       // grid counter shards
  unsigned xcnt[8 * 32];      // one counter per XCD
  unsigned roster[8 * 32];    // enrolment counter per XCD
  unsigned fail;
  unsigned bad;               // wrong sums seen
  unsigned xcd_hist[8];
};

extern __shared__ char smem[];

__global__ __launch_bounds__(NT) void bar_kernel(sync_block* S, int rounds) {
  int* flag = reinterpret_cast<int*>(smem);
  const unsigned nwg = gridDim.x;
  for (int r = 1; r <= rounds; ++r) {
    if (!arrive_wait<8>(S->cnt, blockIdx.x & 7, nwg * (unsigned)r, flag)) {
      if (threadIdx.x == 0) S->fail = 1;
      return;
    }
  }
}

// the bare GROUP barrier of the two-level exchange: 8 groups (blockIdx % 8), one counter word per group, members wait for each other
__global__ __launch_bounds__(NT) void gbar_kernel(sync_block* S, int rounds) {
  int* flag = reinterpret_cast<int*>(smem);
  const unsigned members = gridDim.x / 8, g = blockIdx.x & 7;
  for (int r = 1; r <= rounds; ++r) {
    if (!arrive_wait<1>(S->xcnt + g * 32, 0, members * (unsigned)r, flag)) {
      if (threadIdx.x == 0) S->fail = 1;
      return;
    }
  }
}

// rows: [nwg][nb] bytes; v: [nb]; dots: [nwg][4] doubles.  nb = bytes of the vector (16 KiB at N = 2048 complex)
__global__ __launch_bounds__(NT) void flat_kernel(sync_block* S, float* rows, float* v, double* dots, int nb, int rounds) {
  int* flag = reinterpret_cast<int*>(smem);
  f4* rp = reinterpret_cast<f4*>(smem + 64);  // [8][4]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned nwg = gridDim.x;
  const __amdgpu_buffer_rsrc_t rows_rs = sc1_rsrc(rows), v_rs = sc1_rsrc(v), d_rs = sc1_rsrc(dots);
  unsigned epoch = 0;
  const int per_thread = nb / 16 / NT;  // 16-byte pieces per thread (2 at 16 KiB)
  unsigned bad = 0;
  float want0 = 0.f;  // sum over workgroups of (b % 7): the round-independent part of the expected sum
  for (unsigned b = 0; b < nwg; ++b) want0 += (float)(b % 7);
  for (int r = 1; r <= rounds; ++r) {
    // 1. publish my partial row: element value = (blockIdx % 7) + r % 5  (integer-valued: sums are exact)
    const float val = (float)((blockIdx.x % 7) + (r % 5));
    for (int q = 0; q < per_thread; ++q)
      sc1_store16(rows_rs, (uint32_t)blockIdx.x * (uint32_t)nb + (uint32_t)(q * NT + tid) * 16u, f4{val, val, val, val});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!arrive_wait<8>(S->cnt, blockIdx.x & 7, nwg * ++epoch, flag)) break;
    // 2. sum 64-byte chunk blockIdx over all rows (fixed order), publish it + three doubles
    for (int ch = blockIdx.x; ch < nb / 64; ch += nwg) {
      const int piece = lane >> 4, r16 = lane & 15;
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      for (unsigned row0 = 0; row0 < nwg; row0 += 256) {
        const unsigned ra = row0 + w * 16 + r16, rb = ra + 128;
        const uint32_t col = (uint32_t)ch * 64u + (uint32_t)piece * 16u;
        const f4 ta = sc1_load16(rows_rs, (ra < nwg ? ra : 0) * (uint32_t)nb + col);
        const f4 tb = sc1_load16(rows_rs, (rb < nwg ? rb : 0) * (uint32_t)nb + col);
        if (ra < nwg) acc += ta;
        if (rb < nwg) acc += tb;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float t = acc[q];
        t += dpp_f(t, 0xB1);
        t += dpp_f(t, 0x4E);
        t += dpp_f(t, 0x141);
        t += dpp_f(t, 0x140);
        acc[q] = t;
      }
      if (r16 == 0) rp[w * 4 + piece] = acc;
      __syncthreads();
      if (tid < 4) {
        f4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int ww = 0; ww < 8; ++ww) sum += rp[ww * 4 + tid];
        sc1_store16(v_rs, (uint32_t)ch * 64u + (uint32_t)tid * 16u, sum);
      }
      __syncthreads();
    }
    if (tid < 3)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(dots + 4 * blockIdx.x + tid),
                         __builtin_bit_cast(unsigned long long, (double)(r + tid)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!arrive_wait<8>(S->cnt, blockIdx.x & 7, nwg * ++epoch, flag)) break;
    // 3. everyone reads the vector and the partial dots
    const float want = want0 + (float)nwg * (float)(r % 5);
    for (int q = 0; q < per_thread; ++q) {
      const f4 c = sc1_load16(v_rs, (uint32_t)(q * NT + tid) * 16u);
      if (c.x != want || c.y != want || c.z != want || c.w != want) ++bad;
    }
    if ((unsigned)tid < nwg) {
      const f4 lo = sc1_load16(d_rs, (uint32_t)tid * 32u);
      const double d0 = __builtin_bit_cast(double, __builtin_shufflevector(lo, lo, 0, 1));
      if (d0 != (double)r) ++bad;
    }
  }
  if (epoch != 2u * (unsigned)rounds && tid == 0) S->fail = 1;
  if (bad) atomicAdd(&S->bad, bad);
}

__device__ static inline unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

// XCD-hierarchical.  rows: [8 xcd][32 members][nb]; xpart: [8][nb] XCD-partial vectors; v as above (LEVEL2 == 2).
// Visibility is sc1 stores / sc1 loads everywhere (the guide's validated hand-off), so a wrong idea about which
// workgroups share an L2 could only cost speed; the roster makes the summation order a function of (xcd, member)
// slots, not of which physical workgroup filled them.
template <int LEVEL2>
__global__ __launch_bounds__(NT) void hier_kernel(sync_block* S, float* rows, float* xpart, float* v, int nb, int rounds) {
  int* flag = reinterpret_cast<int*>(smem);
  int* ids = flag + 1;
  f4* rp = reinterpret_cast<f4*>(smem + 64);  // [8 waves][up to 8 pieces]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned nwg = gridDim.x;
  if (tid == 0) {
    const unsigned x = xcc_id();
    const unsigned m = __hip_atomic_fetch_add(S->roster + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ids[0] = (int)x;
    ids[1] = (int)m;
    atomicAdd(&S->xcd_hist[x & 7], 1u);
  }
  __syncthreads();
  const unsigned xcd = (unsigned)ids[0], mem = (unsigned)ids[1];
  const unsigned per_xcd = nwg / 8;
  if (xcd >= 8 || mem >= per_xcd) {  // an uneven placement: this variant does not apply (the product would fall back)
    if (tid == 0) S->fail = 2;
    return;
  }
  const unsigned slot = xcd * per_xcd + mem;  // my row slab index in the product
  const __amdgpu_buffer_rsrc_t rows_rs = sc1_rsrc(rows), x_rs = sc1_rsrc(xpart), v_rs = sc1_rsrc(v);
  const int per_thread = nb / 16 / NT;
  const int slice_b = nb / (int)per_xcd;  // bytes of my slice of the vector (512 at 16 KiB / 32)
  const int pieces = slice_b / 16;        // 16-byte pieces per slice (32)
  unsigned gepoch = 0, xepoch = 0, bad = 0;
  float want0 = 0.f;
  for (unsigned b = 0; b < nwg; ++b) want0 += (float)(b % 7);
  for (int r = 1; r <= rounds; ++r) {
    const float val = (float)((slot % 7) + (r % 5));
    for (int q = 0; q < per_thread; ++q)
      sc1_store16(rows_rs, slot * (uint32_t)nb + (uint32_t)(q * NT + tid) * 16u, f4{val, val, val, val});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // per-XCD barrier: one counter word per XCD
    if (!arrive_wait<1>(S->xcnt + xcd * 32, 0, per_xcd * ++xepoch, flag)) break;
    // slice `mem` over my XCD's rows: pieces x per_xcd 16-byte loads = 1024 at the default shape (2 per thread)
    {
      // thread -> (piece, row): piece = tid % pieces, row group = tid / pieces; NT / pieces row groups of rows each
      const int piece = tid % pieces, rg = tid / pieces, nrg = NT / pieces;  // 32 pieces x 16 row groups
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      for (unsigned row = rg; row < per_xcd; row += nrg)
        acc += sc1_load16(rows_rs, (xcd * per_xcd + row) * (uint32_t)nb + mem * (uint32_t)slice_b + (uint32_t)piece * 16u);
      // fixed-order sum over the row groups through LDS
      f4* ex = reinterpret_cast<f4*>(smem + 1024);  // [nrg][pieces]
      ex[rg * pieces + piece] = acc;
      __syncthreads();
      if (tid < pieces) {
        f4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < nrg; ++g) sum += ex[g * pieces + tid];
        if (LEVEL2 == 1) sc1_store16(x_rs, xcd * (uint32_t)nb + mem * (uint32_t)slice_b + (uint32_t)tid * 16u, sum);
        else sc1_store16(x_rs, xcd * (uint32_t)nb + mem * (uint32_t)slice_b + (uint32_t)tid * 16u, sum);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (!arrive_wait<8>(S->cnt, xcd, nwg * ++gepoch, flag)) break;
    const float want = want0 + (float)nwg * (float)(r % 5);
    if constexpr (LEVEL2 == 1) {
      // every workgroup: the 8 XCD-partial vectors, summed in XCD order
      for (int q = 0; q < per_thread; ++q) {
        f4 t[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) t[x] = sc1_load16(x_rs, (uint32_t)x * (uint32_t)nb + (uint32_t)(q * NT + tid) * 16u);
        f4 c = t[0];
#pragma unroll
        for (int x = 1; x < 8; ++x) c += t[x];
        if (c.x != want || c.y != want || c.z != want || c.w != want) ++bad;
      }
    } else {
      // second reduce-scatter: workgroup `slot` owns 1/nwg of the vector (64 bytes), sums 8 XCD partials, publishes
      const int my_b = nb / (int)nwg;  // 64
      if (tid < my_b / 16) {
        f4 c = {0.f, 0.f, 0.f, 0.f};
        for (int x = 0; x < 8; ++x) c += sc1_load16(x_rs, (uint32_t)x * (uint32_t)nb + slot * (uint32_t)my_b + (uint32_t)tid * 16u);
        sc1_store16(v_rs, slot * (uint32_t)my_b + (uint32_t)tid * 16u, c);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (!arrive_wait<8>(S->cnt, xcd, nwg * ++gepoch, flag)) break;
      for (int q = 0; q < per_thread; ++q) {
        const f4 c = sc1_load16(v_rs, (uint32_t)(q * NT + tid) * 16u);
        if (c.x != want || c.y != want || c.z != want || c.w != want) ++bad;
      }
    }
  }
  if (xepoch != (unsigned)rounds && tid == 0) S->fail = 1;
  if (bad) atomicAdd(&S->bad, bad);
}


// Grouped two-level all-reduce, placement-free: group = blockIdx % GN (XCD-aligned for GN = 8 under the observed round-robin
// dispatch; CONTIG: group = blockIdx / (nwg / GN)), member = the other index.  Everything handed over is sc1-stored and
// sc1-loaded, so which workgroups share an L2 can only change speed, and the summation order is a function of blockIdx only.
//   publish partial row | group barrier (nwg / GN arrivals on the group's counter) | member l sums slice l of the vector over
//   its group's rows and publishes that group-partial slice | grid barrier | everyone reads the GN group-partial vectors
template <int GN, bool CONTIG>
__global__ __launch_bounds__(NT) void group_kernel(sync_block* S, float* rows, float* xpart, int nb, int rounds) {
  int* flag = reinterpret_cast<int*>(smem);
  f4* ex = reinterpret_cast<f4*>(smem + 1024);
  const int tid = threadIdx.x;
  const unsigned nwg = gridDim.x, per = nwg / GN;
  const unsigned grp = CONTIG ? blockIdx.x / per : blockIdx.x % GN, mem = CONTIG ? blockIdx.x % per : blockIdx.x / GN;
  if (grp >= GN || mem >= per) return;
  const unsigned slot = grp * per + mem;
  const __amdgpu_buffer_rsrc_t rows_rs = sc1_rsrc(rows), x_rs = sc1_rsrc(xpart);
  const int per_thread = nb / 16 / NT;
  const int slice_b = nb / (int)per, pieces = slice_b / 16, nrg = NT / pieces;
  unsigned gepoch = 0, xepoch = 0, bad = 0;
  float want0 = 0.f;
  for (unsigned b = 0; b < nwg; ++b) want0 += (float)(b % 7);
  for (int r = 1; r <= rounds; ++r) {
    const float val = (float)((slot % 7) + (r % 5));
    for (int q = 0; q < per_thread; ++q)
      sc1_store16(rows_rs, slot * (uint32_t)nb + (uint32_t)(q * NT + tid) * 16u, f4{val, val, val, val});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!arrive_wait<1>(S->xcnt + grp * 32, 0, per * ++xepoch, flag)) break;
    {
      const int piece = tid % pieces, rg = tid / pieces;
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      for (unsigned row = rg; row < per; row += nrg)
        acc += sc1_load16(rows_rs, (grp * per + row) * (uint32_t)nb + mem * (uint32_t)slice_b + (uint32_t)piece * 16u);
      ex[rg * pieces + piece] = acc;
      __syncthreads();
      if (tid < pieces) {
        f4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < nrg; ++g) sum += ex[g * pieces + tid];
        sc1_store16(x_rs, grp * (uint32_t)nb + mem * (uint32_t)slice_b + (uint32_t)tid * 16u, sum);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (!arrive_wait<8>(S->cnt, blockIdx.x & 7, nwg * ++gepoch, flag)) break;
    const float want = want0 + (float)nwg * (float)(r % 5);
    for (int q = 0; q < per_thread; ++q) {
      f4 t[GN];
#pragma unroll
      for (int x = 0; x < GN; ++x) t[x] = sc1_load16(x_rs, (uint32_t)x * (uint32_t)nb + (uint32_t)(q * NT + tid) * 16u);
      f4 c = t[0];
#pragma unroll
      for (int x = 1; x < GN; ++x) c += t[x];
      if (c.x != want || c.y != want || c.z != want || c.w != want) ++bad;
    }
  }
  if (xepoch != (unsigned)rounds && tid == 0) S->fail = 1;
  if (bad) atomicAdd(&S->bad, bad);
}

// The same exchange with the partial ROWS at L2 scope: under the round-robin dispatch the members of group blockIdx % 8 sit on ONE
// XCD and share its L2, so the rows they hand to EACH OTHER need not travel to the memory side: stored with sc0 they stop in the
// XCD's L2 (whose acknowledgement is what the hand-off waits for), and the members' sc1 loads -- which bypass the CU's L1 as
// before -- find them there.  The group counter and everything that leaves the group (the group-partial slices, the grid barrier)
// stay at agent scope.  Only valid when the placement is what it is assumed to be: the kernel checks HW_REG_XCC_ID against
// blockIdx % 8 and counts the workgroups that are somewhere else (the product decides grid-wide and falls back to sc1 stores).
// Tried on the way and dropped: the group barrier itself inside the L2 (arrivals at workgroup scope + polls by returning atomics:
// 62 us per round -- returning atomics on one contended line; polls by sc0 loads never see the arrivals: they hit in the CU's L1,
// and `buffer_inv sc0` does not invalidate it), sc0 loads of the rows (same L1 problem; `buffer_inv sc1` from one wave costs 1.8 us).
__device__ static inline void l2_store16(__amdgpu_buffer_rsrc_t r, uint32_t off, f4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, off, 0, 1);  // aux 1 = sc0
}
__global__ __launch_bounds__(NT) void group_l2_kernel(sync_block* S, float* rows, float* xpart, int nb, int rounds) {
  constexpr int GN = 8;
  int* flag = reinterpret_cast<int*>(smem);
  f4* ex = reinterpret_cast<f4*>(smem + 1024);
  const int tid = threadIdx.x;
  const unsigned nwg = gridDim.x, per = nwg / GN;
  const unsigned grp = blockIdx.x % GN, mem = blockIdx.x / GN;
  if (mem >= per) return;
  if (tid == 0 && xcc_id() != grp) atomicAdd(&S->xcd_hist[0], 1u);  // misplaced workgroups (the product would decide this grid-wide and fall back to sc1)
  const unsigned slot = grp * per + mem;
  const __amdgpu_buffer_rsrc_t rows_rs = sc1_rsrc(rows), x_rs = sc1_rsrc(xpart);
  const int per_thread = nb / 16 / NT;
  const int slice_b = nb / (int)per, pieces = slice_b / 16, nrg = NT / pieces;
  unsigned gepoch = 0, xepoch = 0, bad = 0;
  float want0 = 0.f;
  for (unsigned b = 0; b < nwg; ++b) want0 += (float)(b % 7);
  for (int r = 1; r <= rounds; ++r) {
    const float val = (float)((slot % 7) + (r % 5));
    for (int q = 0; q < per_thread; ++q)
      l2_store16(rows_rs, slot * (uint32_t)nb + (uint32_t)(q * NT + tid) * 16u, f4{val, val, val, val});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!arrive_wait<1>(S->xcnt + grp * 32, 0, per * ++xepoch, flag)) break;

    if (pieces == 32 && nrg == 16 && per == 32u) {
      // what the product does at this arrangement since round 5 (resident_allreduce, normal.hip): the 16 row groups of a piece in the
      // 16 lanes of one DPP row -- four cross-lane additions per component instead of the LDS hand-over and the 16-term sum below
      const unsigned lane = (unsigned)tid & 63u, rg = lane & 15u, piece = 4u * ((unsigned)tid >> 6) + (lane >> 4);
      const uint32_t col = mem * (uint32_t)slice_b + piece * 16u;
      f4 acc = sc1_load16(rows_rs, (grp * per + rg) * (uint32_t)nb + col) + sc1_load16(rows_rs, (grp * per + rg + 16u) * (uint32_t)nb + col);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v = acc[c];
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
        acc[c] = v;
      }
      if (rg == 0u) sc1_store16(x_rs, grp * (uint32_t)nb + col, acc);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    } else {
      const int piece = tid % pieces, rg = tid / pieces;
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      for (unsigned row = rg; row < per; row += nrg)
        acc += sc1_load16(rows_rs, (grp * per + row) * (uint32_t)nb + mem * (uint32_t)slice_b + (uint32_t)piece * 16u);
      ex[rg * pieces + piece] = acc;
      __syncthreads();
      if (tid < pieces) {
        f4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < nrg; ++g) sum += ex[g * pieces + tid];
        sc1_store16(x_rs, grp * (uint32_t)nb + mem * (uint32_t)slice_b + (uint32_t)tid * 16u, sum);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (!arrive_wait<8>(S->cnt, blockIdx.x & 7, nwg * ++gepoch, flag)) break;
    const float want = want0 + (float)nwg * (float)(r % 5);
    for (int q = 0; q < per_thread; ++q) {
      f4 t[GN];
#pragma unroll
      for (int x = 0; x < GN; ++x) t[x] = sc1_load16(x_rs, (uint32_t)x * (uint32_t)nb + (uint32_t)(q * NT + tid) * 16u);
      f4 c = t[0];
#pragma unroll
      for (int x = 1; x < GN; ++x) c += t[x];
      if (c.x != want || c.y != want || c.z != want || c.w != want) ++bad;
    }
  }
  if (xepoch != (unsigned)rounds && tid == 0) S->fail = 1;
  if (bad) atomicAdd(&S->bad, bad);
}

// The grouped two-level all-reduce WITHOUT barriers: every 8 bytes handed over are {one float of payload, a 32-bit tag = the round
// number as a float}, written in 16-byte pieces (two items) and polled by the reader with 8-byte atomic loads (a buffer-load
// intrinsic in a polling loop is hoisted out of it by the optimiser; 8-byte granules are atomic, so a torn 16-byte store shows at
// worst one stale tag and is read again).  A reader polls the data itself: no "stores drained -> arrive -> poll -> load" chain, at
// twice the bytes.
//   publish my row (tagged) | member l re-reads slice l of its group's rows until every tag is this round's, sums in row order,
//   publishes the group-partial slice (tagged, parity r & 1) | everyone re-reads the 8 group partials until tagged, sums in group order
// Reuse is safe for the reasons the barriered form is: nobody can be two rounds ahead of anybody.
__device__ static inline f4 ll_poll16(const float* base, uint32_t off, unsigned tag, unsigned* fail) {
  const unsigned long long* p = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(base) + off);
  unsigned long long a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (unsigned n = 0; (unsigned)(a >> 32) != tag || (unsigned)(b >> 32) != tag; ++n) {
    if (n > SPIN) {
      *fail = 1;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
    a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  f4 t;
  t.x = __builtin_bit_cast(float, (unsigned)a);
  t.y = __builtin_bit_cast(float, (unsigned)(a >> 32));
  t.z = __builtin_bit_cast(float, (unsigned)b);
  t.w = __builtin_bit_cast(float, (unsigned)(b >> 32));
  return t;
}
__global__ __launch_bounds__(NT) void ll_kernel(sync_block* S, float* rows, float* xpart, int nb, int rounds) {
  float* ex = reinterpret_cast<float*>(smem + 1024);
  const int tid = threadIdx.x;
  constexpr unsigned GN = 8;
  const unsigned nwg = gridDim.x, per = nwg / GN;
  const unsigned grp = blockIdx.x % GN, mem = blockIdx.x / GN;
  const unsigned slot = grp * per + mem;
  const uint32_t llb = 2u * (uint32_t)nb;                 // bytes of a tagged row
  const __amdgpu_buffer_rsrc_t rows_rs = sc1_rsrc(rows), x_rs = sc1_rsrc(xpart);
  const int pieces_row = (int)(llb / 16u);                // 16-byte pieces (2 floats each) per row: 2048 at 16 KiB
  const int per_thread = pieces_row / NT;                 // 4
  const int pieces_slice = pieces_row / (int)per;         // 64
  const int nrg = NT / pieces_slice;                      // 8 row groups
  unsigned bad = 0, fail = 0;
  float want0 = 0.f;
  for (unsigned b = 0; b < nwg; ++b) want0 += (float)(b % 7);
  for (int r = 1; r <= rounds && !fail; ++r) {
    const float tf = (float)r;  // (a NORMAL float as the tag)
    const unsigned tag = __builtin_bit_cast(unsigned, tf);
    const float val = (float)((slot % 7) + (r % 5));
    for (int q = 0; q < per_thread; ++q)
      sc1_store16(rows_rs, slot * llb + (uint32_t)(q * NT + tid) * 16u, f4{val, tf, val, tf});
    {
      const int piece = tid % pieces_slice, rg = tid / pieces_slice;
      float a0 = 0.f, a1 = 0.f;
      {  // all four rows requested at once (per / nrg = 4 at the default shape), stale ones polled again one by one
        unsigned long long lo[4], hi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned row = rg + (unsigned)k * nrg;
          const unsigned long long* p = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(rows) + (grp * per + (row < per ? row : rg)) * llb + (mem * (uint32_t)pieces_slice + (uint32_t)piece) * 16u);
          lo[k] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          hi[k] = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned row = rg + (unsigned)k * nrg;
          if (row >= per) continue;
          f4 t;
          if ((unsigned)(lo[k] >> 32) != tag || (unsigned)(hi[k] >> 32) != tag) {
            t = ll_poll16(rows, (grp * per + row) * llb + (mem * (uint32_t)pieces_slice + (uint32_t)piece) * 16u, tag, &fail);
          } else {
            t.x = __builtin_bit_cast(float, (unsigned)lo[k]);
            t.z = __builtin_bit_cast(float, (unsigned)hi[k]);
          }
          a0 += t.x;
          a1 += t.z;
        }
      }
      __syncthreads();  // (the previous round's readers of `ex` are done)
      ex[2 * (rg * pieces_slice + piece)] = a0;
      ex[2 * (rg * pieces_slice + piece) + 1] = a1;
      __syncthreads();
      if (tid < pieces_slice) {
        float s0 = 0.f, s1 = 0.f;
        for (int g = 0; g < nrg; ++g) {
          s0 += ex[2 * (g * pieces_slice + tid)];
          s1 += ex[2 * (g * pieces_slice + tid) + 1];
        }
        sc1_store16(x_rs, ((uint32_t)(r & 1) * GN + grp) * llb + (mem * (uint32_t)pieces_slice + (uint32_t)tid) * 16u, f4{s0, tf, s1, tf});
      }
    }
    const float want = want0 + (float)nwg * (float)(r % 5);
    for (int q = 0; q < per_thread; ++q) {
      float c0 = 0.f, c1 = 0.f;
      unsigned long long lo[GN], hi[GN];
#pragma unroll
      for (unsigned x = 0; x < GN; ++x) {
        const unsigned long long* p = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(xpart) + ((uint32_t)(r & 1) * GN + x) * llb + (uint32_t)(q * NT + tid) * 16u);
        lo[x] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hi[x] = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (unsigned x = 0; x < GN; ++x) {
        f4 t;
        if ((unsigned)(lo[x] >> 32) != tag || (unsigned)(hi[x] >> 32) != tag) {
          t = ll_poll16(xpart, ((uint32_t)(r & 1) * GN + x) * llb + (uint32_t)(q * NT + tid) * 16u, tag, &fail);
        } else {
          t.x = __builtin_bit_cast(float, (unsigned)lo[x]);
          t.z = __builtin_bit_cast(float, (unsigned)hi[x]);
        }
        c0 += t.x;
        c1 += t.z;
      }
      if (c0 != want || c1 != want) ++bad;
    }
  }
  if (fail && tid == 0) S->fail = 1;
  if (bad) atomicAdd(&S->bad, bad);
}

static void run_ll(sync_block* S, float* rows, float* xpart, int nb, int rounds, int nwg, size_t lds, hipEvent_t e0, hipEvent_t e1) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ll_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(S, 0, sizeof(sync_block)));
    CK(hipMemset(rows, 0, (size_t)nwg * nb * 2));
    CK(hipMemset(xpart, 0, (size_t)32 * nb * 2));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(ll_kernel, dim3(nwg), dim3(NT), lds, 0, S, rows, xpart, nb, rounds);
    CK(hipGetLastError());
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    sync_block h;
    CK(hipMemcpy(&h, S, sizeof(h), hipMemcpyDeviceToHost));
    printf("ll     %8.3f us per round   fail %u  wrong sums %u   (tagged items polled by the reader, no barriers: 8 strided groups)\n", ms * 1e3 / rounds, h.fail, h.bad);
  }
}

template <int GN, bool CONTIG>
static void run_group(sync_block* S, float* rows, float* xpart, int nb, int rounds, int nwg, size_t lds, hipEvent_t e0, hipEvent_t e1) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(group_kernel<GN, CONTIG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(S, 0, sizeof(sync_block)));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((group_kernel<GN, CONTIG>), dim3(nwg), dim3(NT), lds, 0, S, rows, xpart, nb, rounds);
    CK(hipGetLastError());
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    sync_block h;
    CK(hipMemcpy(&h, S, sizeof(h), hipMemcpyDeviceToHost));
    printf("group GN=%2d %s %8.3f us per round   fail %u  wrong sums %u\n", GN, CONTIG ? "contiguous" : "strided   ", ms * 1e3 / rounds, h.fail, h.bad);
  }
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  const int n8 = argc > 2 ? atoi(argv[2]) : 2048;
  const int nb = n8 * 8;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int nwg = prop.multiProcessorCount;
  printf("device %s, %d CUs; %d rounds, vector %d bytes\n", prop.name, nwg, rounds, nb);
  sync_block* S;
  float *rows, *xpart, *v;
  double* dots;
  CK(hipMalloc(&S, sizeof(sync_block)));
  CK(hipMalloc(&rows, (size_t)nwg * nb * 2));   // (twice: the tagged variant hands over {float, tag} pairs)
  CK(hipMalloc(&xpart, (size_t)32 * nb * 2));
  CK(hipMalloc(&v, nb));
  CK(hipMalloc(&dots, (size_t)nwg * 4 * sizeof(double)));
  const size_t lds = 148 * 1024;  // one workgroup per CU, as the product's kernels
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bar_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(flat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(hier_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(hier_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const char* names[] = {"bar", "flat", "hier", "hier2"};
  for (int rep = 0; rep < 2; ++rep) {
    for (int which = 0; which < 4; ++which) {
      CK(hipMemset(S, 0, sizeof(sync_block)));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      switch (which) {
        case 0: hipLaunchKernelGGL(bar_kernel, dim3(nwg), dim3(NT), lds, 0, S, rounds); break;
        case 1: hipLaunchKernelGGL(flat_kernel, dim3(nwg), dim3(NT), lds, 0, S, rows, v, dots, nb, rounds); break;
        case 2: hipLaunchKernelGGL(hier_kernel<1>, dim3(nwg), dim3(NT), lds, 0, S, rows, xpart, v, nb, rounds); break;
        default: hipLaunchKernelGGL(hier_kernel<2>, dim3(nwg), dim3(NT), lds, 0, S, rows, xpart, v, nb, rounds); break;
      }
      CK(hipGetLastError());
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      sync_block h;
      CK(hipMemcpy(&h, S, sizeof(h), hipMemcpyDeviceToHost));
      printf("%-6s %8.3f us per round   fail %u  wrong sums %u", names[which], ms * 1e3 / rounds, h.fail, h.bad);
      if (which >= 2) {
        printf("   workgroups per XCD:");
        for (int x = 0; x < 8; ++x) printf(" %u", h.xcd_hist[x]);
      }
      printf("\n");
    }
  }
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gbar_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(S, 0, sizeof(sync_block)));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(gbar_kernel, dim3(nwg), dim3(NT), lds, 0, S, rounds);
    CK(hipGetLastError());
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    sync_block h;
    CK(hipMemcpy(&h, S, sizeof(h), hipMemcpyDeviceToHost));
    printf("gbar   %8.3f us per round   fail %u  wrong sums %u   (bare group barrier: 8 strided groups of %d)\n", ms * 1e3 / rounds, h.fail, h.bad, nwg / 8);
  }
  run_ll(S, rows, xpart, nb, rounds, nwg, lds, e0, e1);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(group_l2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(S, 0, sizeof(sync_block)));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(group_l2_kernel, dim3(nwg), dim3(NT), lds, 0, S, rows, xpart, nb, rounds);
    CK(hipGetLastError());
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    sync_block h;
    CK(hipMemcpy(&h, S, sizeof(h), hipMemcpyDeviceToHost));
    printf("groupl2  %8.3f us per round   fail %u  wrong sums %u   (GN = 8 strided, partial rows stored at L2 scope; misplaced workgroups %u)\n", ms * 1e3 / rounds, h.fail, h.bad, h.xcd_hist[0]);
  }
  run_group<2, false>(S, rows, xpart, nb, rounds, nwg, lds, e0, e1);
  run_group<4, false>(S, rows, xpart, nb, rounds, nwg, lds, e0, e1);
  run_group<8, false>(S, rows, xpart, nb, rounds, nwg, lds, e0, e1);
  run_group<16, false>(S, rows, xpart, nb, rounds, nwg, lds, e0, e1);
  run_group<32, false>(S, rows, xpart, nb, rounds, nwg, lds, e0, e1);
  run_group<4, true>(S, rows, xpart, nb, rounds, nwg, lds, e0, e1);
  run_group<8, true>(S, rows, xpart, nb, rounds, nwg, lds, e0, e1);
  run_group<16, true>(S, rows, xpart, nb, rounds, nwg, lds, e0, e1);
  return 0;
}
