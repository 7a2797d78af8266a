// Fused normal operator  v = A^H (A p)  in ONE pass over A.
//
// The reference evaluates mul!(v, AHA, p) (src/CGNR.jl:151, src/FISTA.jl:152, cg! from
// src/ADMM.jl:244) as two dependent GEMVs, t = A p then v = A^H t, each streaming all of A from
// memory.  On MI355X a CU's register file (512 KiB) is larger than its share of A at the headline
// shape (64 MiB / 256 CUs = 256 KiB), so a workgroup that owns a ROW slab A_w (G*V rows x all N
// columns) can
//     1. load the slab once, 16 bytes per lane per load, K loads per lane kept in VGPRs,
//     2. form its rows of t = A_w p completely (no other workgroup contributes to those rows),
//     3. form its contribution A_w^H t_w to every column of v from the SAME registers,
// and only the N-vector partials (one per workgroup) go back to memory.  A second small kernel sums
// the partials in a fixed order (deterministic, no atomics).  HBM traffic per apply: M*N*s for A
// once + 2 * nwg*N*s for the partials, instead of 2*M*N*s.
//
// Lane layout inside a wave: g = lane % G picks the 16-byte row chunk, s = lane / G one of the
// 64/G column slots; wave w of 16 and load k cover column (k*16 + w)*(64/G) + s.
#include "rls_common.hpp"

namespace {

constexpr int NWAVES = 16;
constexpr int NTHREADS = NWAVES * 64;

// blocks b and b+8 are observed to share an XCD (speed only, never correctness): with 64-byte
// row chunks (G = 4) two neighbouring row blocks split every 128-byte line, so give them to
// blocks that share an L2.
__device__ static inline int64_t row_block_of(int64_t b, int pair) {
  if (!pair) return b;
  return (b / 16) * 16 + (b % 8) * 2 + ((b / 8) % 2);
}

// sum over the G = 2/4/8 consecutive lanes of a group with DPP (full-rate VALU) instead of
// ds_bpermute shuffles: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror
template <int G>
__device__ static inline float group_sum(float v) {
  if constexpr (G >= 2) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  if constexpr (G >= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  if constexpr (G >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}

template <typename E, int G, int K>
__global__ __launch_bounds__(NTHREADS) void normal_slab_kernel(const E* __restrict__ A, int64_t lda,
                                                               const E* __restrict__ p, E* __restrict__ slab,
                                                               int64_t Mc, int64_t N, int pair,
                                                               const int* __restrict__ skip) {
  if (skip && *skip) return;
  constexpr int NV = elem<E>::vec;
  constexpr int S = 64 / G;
  constexpr int CPR = NWAVES * S;
  constexpr int NMAX = K * CPR;
  __shared__ E xs[NMAX];  // p, later this workgroup's partial v
  __shared__ E part[NWAVES][G][NV];
  __shared__ E tw[G * NV];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane % G, s = lane / G;
  const int slot = w * S + s;
  const int64_t rb = row_block_of(blockIdx.x, pair);  // pair only when gridDim.x % 16 == 0 (bijective)
  const int64_t chunk_id = rb * G + g;
  const bool row_ok = chunk_id < Mc;
  const E* Ab = A + (row_ok ? chunk_id : (Mc - 1)) * NV;

  // the slab: every load is issued before anything waits (addresses clamped, never branched)
  chunk<E, NV> a[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int jl = k * CPR + slot;
    const int jc = jl < N ? jl : (int)(N - 1);
    a[k] = load_chunk<E, NV>(Ab + (int64_t)jc * lda);
  }
  for (int i = tid; i < NMAX; i += NTHREADS) xs[i] = i < N ? p[i] : elem<E>::zero();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int jl = k * CPR + slot;
    if (jl >= N || !row_ok) a[k] = zero_chunk<E, NV>();
  }
  __syncthreads();

  // t_w = A_w p
  E acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = elem<E>::zero();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const E xv = xs[k * CPR + slot];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = elem<E>::fma(a[k].e[i], xv, acc[i]);
  }
#pragma unroll
  for (int off = G; off < 64; off <<= 1) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float re = elem<E>::re(acc[i]) + __shfl_xor(elem<E>::re(acc[i]), off, 64);
      float im = elem<E>::cplx ? elem<E>::im(acc[i]) + __shfl_xor(elem<E>::im(acc[i]), off, 64) : 0.f;
      acc[i] = elem<E>::make(re, im);
    }
  }
  if (s == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) part[w][g][i] = acc[i];
  }
  __syncthreads();
  if (tid < G * NV) {
    const int gg = tid / NV, i = tid % NV;
    E sum = elem<E>::zero();
#pragma unroll
    for (int ww = 0; ww < NWAVES; ++ww) sum = elem<E>::add(sum, part[ww][gg][i]);
    tw[tid] = sum;
  }
  __syncthreads();

  // partial v = A_w^H t_w from the same registers
  E tr[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) tr[i] = tw[g * NV + i];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    E q = elem<E>::zero();
#pragma unroll
    for (int i = 0; i < NV; ++i) q = elem<E>::fmac(a[k].e[i], tr[i], q);
    q = elem<E>::make(group_sum<G>(elem<E>::re(q)), elem<E>::cplx ? group_sum<G>(elem<E>::im(q)) : 0.f);
    if (g == 0) xs[k * CPR + slot] = q;
  }
  __syncthreads();
  E* out = slab + (int64_t)blockIdx.x * N;
  for (int i = tid; i < N; i += NTHREADS) out[i] = xs[i];
}

// v[j] = sum_w slab[w][j] in a fixed order: 16 columns per workgroup, 16 row groups per column
template <typename E>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const E* __restrict__ slab, int nwg, int64_t N,
                                                          E* __restrict__ v, const int* __restrict__ skip) {
  if (skip && *skip) return;
  __shared__ E sm[16][16];
  const int cx = threadIdx.x % 16, wy = threadIdx.x / 16;
  const int64_t j = (int64_t)blockIdx.x * 16 + cx;
  const int64_t jc = j < N ? j : (N - 1);
  E s0 = elem<E>::zero(), s1 = elem<E>::zero();
  int wgi = wy;
  for (; wgi + 16 < nwg; wgi += 32) {  // two independent chains keep more loads in flight
    s0 = elem<E>::add(s0, slab[(int64_t)wgi * N + jc]);
    s1 = elem<E>::add(s1, slab[(int64_t)(wgi + 16) * N + jc]);
  }
  if (wgi < nwg) s0 = elem<E>::add(s0, slab[(int64_t)wgi * N + jc]);
  sm[wy][cx] = elem<E>::add(s0, s1);
  __syncthreads();
  if (wy == 0 && j < N) {
    E t = elem<E>::zero();
#pragma unroll
    for (int i = 0; i < 16; ++i) t = elem<E>::add(t, sm[i][cx]);
    v[j] = t;
  }
}

struct fused_cfg {
  int G, K;
};

static int g_force_g = 0;  // measurement override (rls_tune_set "slab_g"): 0 = heuristic

template <typename E>
static bool pick_cfg(int64_t N, fused_cfg* c) {
  // NMAX = K * 16 * (64 / G) columns fit in the register slab
  for (int G : {8, 4}) {
    if (g_force_g && G != g_force_g) continue;
    const int cpr = 16 * (64 / G);
    for (int K : {4, 8, 16}) {
      if (N <= (int64_t)K * cpr) {
        *c = {G, K};
        return true;
      }
    }
  }
  return false;
}

template <typename E>
static bool fused_ok(int64_t M, int64_t N, const void* A, int64_t lda) {
  constexpr int V = elem<E>::vec;
  fused_cfg c;
  return A && M > 0 && N > 0 && M % V == 0 && lda % V == 0 && (reinterpret_cast<uintptr_t>(A) % 16 == 0) &&
         pick_cfg<E>(N, &c);
}

template <typename E>
static int64_t fused_nwg(int64_t M, int64_t N) {
  fused_cfg c;
  pick_cfg<E>(N, &c);
  const int64_t Mc = M / elem<E>::vec;
  return (Mc + c.G - 1) / c.G;
}

template <typename E, int G, int K>
static void launch_slab(rls_ctx* ctx, const E* A, int64_t lda, const E* p, E* slab, int64_t M, int64_t N, int nwg,
                        const int* skip) {
  hipLaunchKernelGGL((normal_slab_kernel<E, G, K>), dim3(nwg), dim3(NTHREADS), 0, ctx->stream, A, lda, p, slab,
                     M / elem<E>::vec, N, (G == 4 && nwg % 16 == 0) ? 1 : 0, skip);
}

template <typename E>
static int32_t normal_typed(rls_ctx* ctx, int64_t M, int64_t N, const E* A, int64_t lda, const E* p, E* v, E* slab,
                            const int* skip) {
  fused_cfg c;
  if (!pick_cfg<E>(N, &c)) return rls_fail(ctx, RLS_E_UNSUPPORTED, "normal_fused: N too large for a register slab");
  const int nwg = (int)fused_nwg<E>(M, N);
#define RLS_SLAB_CASE(GG, KK) \
  if (c.G == GG && c.K == KK) launch_slab<E, GG, KK>(ctx, A, lda, p, slab, M, N, nwg, skip);
  RLS_SLAB_CASE(8, 4) RLS_SLAB_CASE(8, 8) RLS_SLAB_CASE(8, 16) RLS_SLAB_CASE(4, 4) RLS_SLAB_CASE(4, 8) RLS_SLAB_CASE(4, 16)
#undef RLS_SLAB_CASE
  hipLaunchKernelGGL(slab_reduce_kernel<E>, dim3((unsigned)((N + 15) / 16)), dim3(256), 0, ctx->stream, slab, nwg, N,
                     v, skip);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

}  // namespace

void rls_normal_force_group(int g) { g_force_g = g; }

size_t rls_normal_fused_workspace(int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda) {
  if (dtype == RLS_F32) return fused_ok<float>(M, N, A, lda) ? (size_t)fused_nwg<float>(M, N) * N * 4 : 0;
  if (dtype == RLS_C32) return fused_ok<float2>(M, N, A, lda) ? (size_t)fused_nwg<float2>(M, N) * N * 8 : 0;
  return 0;
}

int32_t rls_launch_normal_fused(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda,
                                const void* p, void* v, void* slab, const int* skip) {
  RLS_CHECK_CTX(ctx);
  if (!A || !p || !v || !slab) return rls_fail(ctx, RLS_E_INVALID, "normal_fused: null pointer");
  RLS_HIP(ctx, hipSetDevice(ctx->device));
  if (dtype == RLS_F32)
    return normal_typed<float>(ctx, M, N, (const float*)A, lda, (const float*)p, (float*)v, (float*)slab, skip);
  return normal_typed<float2>(ctx, M, N, (const float2*)A, lda, (const float2*)p, (float2*)v, (float2*)slab, skip);
}
