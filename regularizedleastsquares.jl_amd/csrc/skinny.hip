// K right-hand sides sharing one A on the matrix cores (BASELINE config 4, shared-A flavour; the
// per-column semantics of solve!(solver, B; scheduler = MultiThreadingState), src/MultiThreading.jl:30-79).
//
// One CGNR iteration (src/CGNR.jl:143-178) of all K columns is three launches:
//   skinny_t_kernel   T = A P        M x K   contraction over the columns of A
//   skinny_v_kernel   V = A^H T      N x K   contraction over the rows of A (row-split partial sums)
//   skinny_u_kernel   per column: v = sum of the partials, alpha, x, r, beta, p  (:153-176)
// Both products run on v_mfma_f32_16x16x4_f32 (exact f32 FMA chains): the 16-wide free index of the
// B operand carries 16 right-hand sides, so A is streamed once per product for 16 solves; complex
// arithmetic is four real MFMAs on the (re, im) parts of the operands.  K > 16 runs as groups of 16
// (a grid dimension).  The skinny operands are kept as row-major panels of 16 right-hand sides,
//   Pp[g][n][j] = P[n][16 g + j]   (N x 16),     Tp[g][m][j] = T[m][16 g + j]   (M x 16),
// which is exactly the MFMA B-operand order: a wave reads 4 consecutive rows (64 elements) with one
// contiguous load.  The partial-row slab of the one-pass kernel (normal.hip) would grow with K
// (nwg x N x K), which is why this path streams A twice instead: 2 x 64 MiB out of the Infinity Cache
// per 16 solve-iterations; at 16 right-hand sides both products are bound by the f32 MFMA rate.
//
// Measured on MI355X (4096 x 2048 ComplexF32, 16 RHS): one wave per SIMD (4-wave workgroups) overlaps
// the loads with the MFMAs, two waves per SIMD do not (21.6 vs 13.6 us for T = A P); small load
// batches beat deep ones (U = 4: 12.6 us, U = 16: 17.0 us); the MFMAs alone take 11.2 us.
#include "rls_common.hpp"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ static inline f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// T = A P
// workgroup = 16 rows of A x all columns; wave w takes the 4-column blocks nb = w, w + WV, ...
// lane l: A operand = A[16 mb + (l & 15)][4 nb + (l >> 4)], B operand = Pp[g][4 nb + (l >> 4)][l & 15]
// ---------------------------------------------------------------------------------------------
// PE = element type of the operand panel: E, or float for the half layout (lane l reads float
// Pp[4 nb + (l >> 4)][l & 15] = re (slots 0..7) or im (slots 8..15) of right-hand side l & 7)
template <typename E, typename PE, int WV, int U>
__device__ static inline void t_load(E (&a)[U], PE (&p)[U], const E* __restrict__ Ap, int64_t lda,
                                     const PE* __restrict__ Pg, int w, int64_t i0) {
  // no predication here: a select on a wave-uniform condition becomes a branch around the load, and the
  // compiler then waits vmcnt(0) instead of counting (the callers only pass in-range steps)
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t nb = w + WV * (i0 + u);
    a[u] = Ap[nb * 4 * lda];
    p[u] = Pg[nb * 64];
  }
}

template <typename E, typename PE, int U>
__device__ static inline void t_mma(f32x4 (&acc)[4], const E (&a)[U], const PE (&p)[U]) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if constexpr (!std::is_same<E, PE>::value) {
      // half layout: ar x (pr | pi) and ai x (pr | pi) -- the same four FMA chains as below, in two instructions
      acc[0] = mfma4(elem<E>::re(a[u]), p[u], acc[0]);
      acc[1] = mfma4(elem<E>::im(a[u]), p[u], acc[1]);
    } else if constexpr (elem<E>::cplx) {
      const float ar = elem<E>::re(a[u]), ai = elem<E>::im(a[u]);
      const float pr = elem<E>::re(p[u]), pi = elem<E>::im(p[u]);
      acc[0] = mfma4(ar, pr, acc[0]);
      acc[1] = mfma4(ai, pi, acc[1]);
      acc[2] = mfma4(ar, pi, acc[2]);
      acc[3] = mfma4(ai, pr, acc[3]);
    } else {
      acc[u & 1] = mfma4(elem<E>::re(a[u]), elem<E>::re(p[u]), acc[u & 1]);  // two chains: 40-cycle dependent latency
    }
  }
}

// VOUT (Gram mode, V = AHA P with the explicit N x N matrix as `A`: the reference constructors' default operator,
// src/CGNR.jl:49,151): the contraction is split over blockIdx.z and the result goes out as partial rows in the layout
// of skinny_v_kernel (Vpart[split][right-hand side][row]), so that the per-column update kernels read it unchanged.
template <typename E, int WV, int U, bool H, int D = 0, bool VOUT = false>
__global__ __launch_bounds__(WV * 64) void skinny_t_kernel(const E* __restrict__ A, int64_t lda,
                                                            const E* __restrict__ Pp, E* __restrict__ Tp, int64_t M,
                                                            int64_t N, int nrhs_pad = 0, int64_t ldvp = 0) {
  constexpr bool CX = elem<E>::cplx;
  static_assert(!H || CX, "the half layout is a complex layout");
  using PE = typename std::conditional<H, float, E>::type;
  __shared__ float red[WV][2][4][64];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t mb = blockIdx.x;
  const int g = blockIdx.y;
  const int sp = VOUT ? (int)blockIdx.z : 0, SP = VOUT ? (int)gridDim.z : 1;
  const int64_t nb_lo = sp * (N / 4) / SP, NB = (sp + 1) * (N / 4) / SP - nb_lo;  // this split's 4-column blocks
  const PE* Pg = reinterpret_cast<const PE*>(Pp) + (int64_t)g * N * 16 + nb_lo * 64 + lane;
  const E* Ap = A + mb * 16 + (lane & 15) + ((int64_t)(lane >> 4) + nb_lo * 4) * lda;
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t ns = NB > w ? (NB - w + WV - 1) / WV : 0;
  bool rolled = false;
  if constexpr (D > 0) {
    // rolling window: D steps always in flight -- step i's MFMAs are followed at once by the loads of step i + D
    // into the registers they have just freed, so the waits are vmcnt(2 (D - 1)) throughout and there is no bubble
    // between batches (the two-set pipeline below drains one set before it refills it).  ns a multiple of D.
    if (ns >= 2 * D && ns % D == 0) {
      rolled = true;
      E a[D];
      PE p[D];
#pragma unroll
      for (int u = 0; u < D; ++u) {
        const int64_t nb = w + WV * (int64_t)u;
        a[u] = Ap[nb * 4 * lda];
        p[u] = Pg[nb * 64];
        // same issue order as in the loop: the wait counts at the loop head are the minimum over both ways in
        __builtin_amdgcn_sched_barrier(0);
      }
      for (int64_t i = D; i < ns; i += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
          E a1[1] = {a[u]};
          PE p1[1] = {p[u]};
          t_mma<E, PE, 1>(acc, a1, p1);
          const int64_t nb = w + WV * (i + u);
          a[u] = Ap[nb * 4 * lda];
          p[u] = Pg[nb * 64];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int u = 0; u < D; ++u) {
        E a1[1] = {a[u]};
        PE p1[1] = {p[u]};
        t_mma<E, PE, 1>(acc, a1, p1);
      }
    }
  }
  // software pipeline over batches of U steps, two register sets; the loop body has no branch around
  // a load, so the waits are counted (vmcnt(N)) and the next batch stays in flight under the MFMAs
  const int64_t nfull = rolled ? 0 : ns / U;
  if (nfull > 0) {
    E a0[U], a1[U];
    PE p0[U], p1[U];
    t_load<E, PE, WV, U>(a0, p0, Ap, lda, Pg, w, 0);
    int64_t b = 0;
    for (; b + 2 < nfull; b += 2) {
      // sched_barrier: keep the whole next batch of loads AHEAD of this batch's MFMAs (the scheduler
      // otherwise sinks the loads towards their uses and the prefetch distance collapses)
      t_load<E, PE, WV, U>(a1, p1, Ap, lda, Pg, w, (b + 1) * U);
      __builtin_amdgcn_sched_barrier(0);
      t_mma<E, PE, U>(acc, a0, p0);
      __builtin_amdgcn_sched_barrier(0);
      t_load<E, PE, WV, U>(a0, p0, Ap, lda, Pg, w, (b + 2) * U);
      __builtin_amdgcn_sched_barrier(0);
      t_mma<E, PE, U>(acc, a1, p1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (b + 1 < nfull) {
      t_load<E, PE, WV, U>(a1, p1, Ap, lda, Pg, w, (b + 1) * U);
      t_mma<E, PE, U>(acc, a0, p0);
      t_mma<E, PE, U>(acc, a1, p1);
    } else {
      t_mma<E, PE, U>(acc, a0, p0);
    }
  }
  for (int64_t i = rolled ? ns : nfull * U; i < ns; ++i) {  // remainder steps, one at a time
    E a2[1];
    PE p2[1];
    t_load<E, PE, WV, 1>(a2, p2, Ap, lda, Pg, w, i);
    t_mma<E, PE, 1>(acc, a2, p2);
  }
  f32x4 tre, tim;
  if constexpr (H) {
    tre = acc[0];  // combined across the (re | im) slots below
    tim = acc[1];
  } else if constexpr (CX) {
    tre = acc[0] - acc[1];
    tim = acc[2] + acc[3];
  } else {
    tre = acc[0] + acc[1];
    tim = tre;
  }
  // accumulator register t of lane (q = l >> 4, j = l & 15) is T[16 mb + 4 q + t][j]
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    red[w][0][t][lane] = tre[t];
    if constexpr (CX) red[w][1][t][lane] = tim[t];
  }
  __syncthreads();
  if constexpr (VOUT) {  // idx = right-hand side * 16 + row: a right-hand side's 16 rows are one contiguous piece
    constexpr int NJ = H ? 8 : 16;
    for (int idx = threadIdx.x; idx < NJ * 16; idx += WV * 64) {
      const int j = idx >> 4, row = idx & 15;
      const int l = (row >> 2) * 16 + j, t = row & 3;
      float re = 0.f, im = 0.f;
#pragma unroll
      for (int ww = 0; ww < WV; ++ww) {
        if constexpr (H) {  // re = acc0[j] - acc1[j + 8], im = acc0[j + 8] + acc1[j]
          re += red[ww][0][t][l] - red[ww][1][t][l + 8];
          im += red[ww][0][t][l + 8] + red[ww][1][t][l];
        } else {
          re += red[ww][0][t][l];
          if constexpr (CX) im += red[ww][1][t][l];
        }
      }
      Tp[((int64_t)sp * nrhs_pad + 16 * g + j) * ldvp + mb * 16 + row] = elem<E>::make(re, im);
    }
    return;
  }
  if constexpr (H) {
    // slot j < 8: re = ar pr - ai pi = acc0[j] - acc1[j + 8]; slot j >= 8: im = ar pi + ai pr = acc0[j] + acc1[j - 8]
    // (per wave first, then over the waves: the same additions in the same order as the full layout)
    float* out = reinterpret_cast<float*>(Tp) + mb * 16 * 16;
    for (int idx = threadIdx.x; idx < 256; idx += WV * 64) {  // idx = row * 16 + slot
      const int row = idx >> 4, j = idx & 15;
      const int l = (row >> 2) * 16 + j, t = row & 3;
      float val = 0.f;
#pragma unroll
      for (int ww = 0; ww < WV; ++ww)
        val += j < 8 ? red[ww][0][t][l] - red[ww][1][t][l + 8] : red[ww][0][t][l] + red[ww][1][t][l - 8];
      out[idx] = val;
    }
    return;
  }
  E* out = Tp + ((int64_t)g * M + mb * 16) * 16;
  for (int idx = threadIdx.x; idx < 256; idx += WV * 64) {  // idx = row * 16 + j
    const int row = idx >> 4, j = idx & 15;
    const int l = (row >> 2) * 16 + j, t = row & 3;
    float re = 0.f, im = 0.f;
#pragma unroll
    for (int ww = 0; ww < WV; ++ww) {
      re += red[ww][0][t][l];
      if constexpr (CX) im += red[ww][1][t][l];
    }
    out[idx] = elem<E>::make(re, im);
  }
}

// ---------------------------------------------------------------------------------------------
// V = A^H T (partial over a row split)
// workgroup = 16 columns of A x the row blocks [lo, hi) of split s; wave w takes mb = lo + w, lo + w + WV, ...
// lane l (q = l >> 4): A operand of step register t = conj(A[16 mb + 4 q + t][n0 + (l & 15)]) (one 32-byte
// piece of the column per lane), B operand = Tp[g][16 mb + 4 q + t][l & 15]
// ---------------------------------------------------------------------------------------------
template <typename E, typename PE>
struct v_regs {
  float4 x0, x1;  // complex: rows (0,1), (2,3) as (re, im) pairs; real: x0 = rows 0..3
  PE t[4];        // PE = E, or float for the half layout (slots 0..7 = re, 8..15 = im of 8 right-hand sides)
};

template <typename E, typename PE, int WV, int U>
__device__ static inline void v_load(v_regs<E, PE> (&q)[U], const E* __restrict__ Ap, const PE* __restrict__ Tg, int w,
                                     int64_t lo, int64_t i0) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t mb = lo + w + WV * (i0 + u);
    const float4* ap = reinterpret_cast<const float4*>(Ap + mb * 16);
    q[u].x0 = ap[0];
    if constexpr (elem<E>::cplx) q[u].x1 = ap[1];
    const PE* tp = Tg + mb * 256;
#pragma unroll
    for (int t = 0; t < 4; ++t) q[u].t[t] = tp[t * 16];
  }
}

template <typename E, typename PE, int U>
__device__ static inline void v_mma(f32x4 (&acc)[4], const v_regs<E, PE> (&q)[U]) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if constexpr (!std::is_same<E, PE>::value) {
      const float ar[4] = {q[u].x0.x, q[u].x0.z, q[u].x1.x, q[u].x1.z};
      const float ai[4] = {q[u].x0.y, q[u].x0.w, q[u].x1.y, q[u].x1.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[0] = mfma4(ar[t], q[u].t[t], acc[0]);  // ar x (tr | ti)
        acc[1] = mfma4(ai[t], q[u].t[t], acc[1]);  // ai x (tr | ti)
      }
    } else if constexpr (elem<E>::cplx) {
      const float ar[4] = {q[u].x0.x, q[u].x0.z, q[u].x1.x, q[u].x1.z};
      const float ai[4] = {q[u].x0.y, q[u].x0.w, q[u].x1.y, q[u].x1.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float tr = elem<E>::re(q[u].t[t]), ti = elem<E>::im(q[u].t[t]);
        acc[0] = mfma4(ar[t], tr, acc[0]);
        acc[1] = mfma4(ai[t], ti, acc[1]);
        acc[2] = mfma4(ar[t], ti, acc[2]);
        acc[3] = mfma4(ai[t], tr, acc[3]);
      }
    } else {
      const float ar[4] = {q[u].x0.x, q[u].x0.y, q[u].x0.z, q[u].x0.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t & 1] = mfma4(ar[t], elem<E>::re(q[u].t[t]), acc[t & 1]);
    }
  }
}

template <typename E, int WV, int U, bool H, int D = 0>
__global__ __launch_bounds__(WV * 64) void skinny_v_kernel(const E* __restrict__ A, int64_t lda,
                                                            const E* __restrict__ Tp, E* __restrict__ Vpart,
                                                            int64_t M, int64_t N, int nrhs_pad, int64_t ldvp) {
  constexpr bool CX = elem<E>::cplx;
  static_assert(!H || CX, "the half layout is a complex layout");
  using PE = typename std::conditional<H, float, E>::type;
  __shared__ float red[WV][2][4][64];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n0 = (int64_t)blockIdx.x * 16, MB = M / 16;
  const int s = blockIdx.y, S = gridDim.y, g = blockIdx.z;
  const int64_t lo = s * MB / S, hi = (s + 1) * MB / S;
  const E* Ap = A + (n0 + (lane & 15)) * lda + 4 * (lane >> 4);
  const PE* Tg = reinterpret_cast<const PE*>(Tp) + (int64_t)g * M * 16 + (lane >> 4) * 64 + (lane & 15);
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t ns = hi - lo > w ? (hi - lo - w + WV - 1) / WV : 0;
  bool rolled = false;
  if constexpr (D > 0) {  // rolling window of D steps, as in skinny_t_kernel
    if (ns >= 2 * D && ns % D == 0) {
      rolled = true;
      v_regs<E, PE> q[D];
#pragma unroll
      for (int u = 0; u < D; ++u) {
        v_regs<E, PE> q1[1];
        v_load<E, PE, WV, 1>(q1, Ap, Tg, w, lo, u);
        q[u] = q1[0];
        __builtin_amdgcn_sched_barrier(0);
      }
      for (int64_t i = D; i < ns; i += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
          v_regs<E, PE> q1[1] = {q[u]};
          v_mma<E, PE, 1>(acc, q1);
          v_load<E, PE, WV, 1>(q1, Ap, Tg, w, lo, i + u);
          q[u] = q1[0];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int u = 0; u < D; ++u) {
        v_regs<E, PE> q1[1] = {q[u]};
        v_mma<E, PE, 1>(acc, q1);
      }
    }
  }
  const int64_t nfull = rolled ? 0 : ns / U;
  if (nfull > 0) {
    v_regs<E, PE> q0[U], q1[U];
    v_load<E, PE, WV, U>(q0, Ap, Tg, w, lo, 0);
    int64_t b = 0;
    for (; b + 2 < nfull; b += 2) {
      v_load<E, PE, WV, U>(q1, Ap, Tg, w, lo, (b + 1) * U);
      __builtin_amdgcn_sched_barrier(0);
      v_mma<E, PE, U>(acc, q0);
      __builtin_amdgcn_sched_barrier(0);
      v_load<E, PE, WV, U>(q0, Ap, Tg, w, lo, (b + 2) * U);
      __builtin_amdgcn_sched_barrier(0);
      v_mma<E, PE, U>(acc, q1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (b + 1 < nfull) {
      v_load<E, PE, WV, U>(q1, Ap, Tg, w, lo, (b + 1) * U);
      v_mma<E, PE, U>(acc, q0);
      v_mma<E, PE, U>(acc, q1);
    } else {
      v_mma<E, PE, U>(acc, q0);
    }
  }
  for (int64_t i = rolled ? ns : nfull * U; i < ns; ++i) {
    v_regs<E, PE> q2[1];
    v_load<E, PE, WV, 1>(q2, Ap, Tg, w, lo, i);
    v_mma<E, PE, 1>(acc, q2);
  }
  f32x4 vre, vim;
  if constexpr (H) {
    vre = acc[0];  // combined across the (re | im) slots below
    vim = acc[1];
  } else if constexpr (CX) {
    vre = acc[0] + acc[1];
    vim = acc[2] - acc[3];
  } else {
    vre = acc[0] + acc[1];
    vim = vre;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    red[w][0][u][lane] = vre[u];
    if constexpr (CX) red[w][1][u][lane] = vim[u];
  }
  __syncthreads();
  if constexpr (H) {
    // re = ar tr + ai ti = acc0[j] + acc1[j + 8], im = ar ti - ai tr = acc0[j + 8] - acc1[j]   (j < 8)
    for (int idx = threadIdx.x; idx < 128; idx += WV * 64) {  // idx = j * 16 + column
      const int j = idx >> 4, c = idx & 15;
      const int l = (c >> 2) * 16 + j, u = c & 3;
      float re = 0.f, im = 0.f;
#pragma unroll
      for (int ww = 0; ww < WV; ++ww) {
        re += red[ww][0][u][l] + red[ww][1][u][l + 8];
        im += red[ww][0][u][l + 8] - red[ww][1][u][l];
      }
      Vpart[((int64_t)s * nrhs_pad + j) * ldvp + n0 + c] = elem<E>::make(re, im);
    }
    return;
  }
  // accumulator register u of lane (q, j) is V[n0 + 4 q + u][j]; stored per right-hand side (16 columns
  // = one 128-byte line each) so that the update kernel reads its column contiguously
  for (int idx = threadIdx.x; idx < 256; idx += WV * 64) {  // idx = j * 16 + column
    const int j = idx >> 4, c = idx & 15;
    const int l = (c >> 2) * 16 + j, u = c & 3;
    float re = 0.f, im = 0.f;
#pragma unroll
    for (int ww = 0; ww < WV; ++ww) {
      re += red[ww][0][u][l];
      if constexpr (CX) im += red[ww][1][u][l];
    }
    Vpart[((int64_t)s * nrhs_pad + 16 * g + j) * ldvp + n0 + c] = elem<E>::make(re, im);
  }
}

// ---------------------------------------------------------------------------------------------
// B (M x nrhs, column-major) -> Tp panels, so that the init product A^H B runs on skinny_v_kernel
// ---------------------------------------------------------------------------------------------
template <typename E, bool H>
__global__ __launch_bounds__(256) void skinny_pack_rows_kernel(const E* __restrict__ B, int64_t ldb, int nrhs,
                                                               E* __restrict__ Tp, int64_t M, int ngroups) {
  __shared__ E tile[16][65];
  // one workgroup transposes 64 rows x 16 right-hand sides
  const int64_t m0 = (int64_t)blockIdx.x * 64;
  const int g = blockIdx.y;
  for (int idx = threadIdx.x; idx < 1024; idx += 256) {
    const int j = idx >> 6, r = idx & 63;
    const int rhs = 16 * g + j;
    tile[j][r] = (rhs < nrhs && m0 + r < M) ? B[(int64_t)rhs * ldb + m0 + r] : elem<E>::zero();
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 1024; idx += 256) {
    const int r = idx >> 4, j = idx & 15;
    if (m0 + r >= M) continue;
    if constexpr (H) {  // slots 0..7 = re, 8..15 = im of right-hand side j & 7
      const E t = tile[j & 7][r];
      reinterpret_cast<float*>(Tp)[(m0 + r) * 16 + j] = j < 8 ? elem<E>::re(t) : elem<E>::im(t);
    } else {
      Tp[((int64_t)g * M + m0 + r) * 16 + j] = tile[j][r];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// per-column CG update (one workgroup per right-hand side), src/CGNR.jl:108-126 (INIT) / :153-176.
// EPT > 0: the column lives in registers (N <= 1024 EPT), every load is issued before the first
// reduction; EPT == 0: any N, the vectors are re-read between the reductions.
// ---------------------------------------------------------------------------------------------
// NT threads per workgroup: 512 where the column fits in 8 elements per thread (N = 2048: 7.0 us per launch against
// 7.9 with 1024 and 7.6 with 256), 1024 beyond

template <typename E>
__device__ static inline double abs2d(E a) {
  return (double)elem<E>::re(a) * (double)elem<E>::re(a) + (double)elem<E>::im(a) * (double)elem<E>::im(a);
}

template <typename E>
__device__ static inline E sum_parts(const E* __restrict__ Vpart, int S, int nrhs_pad, int b, int64_t N, int64_t i) {
  E v = Vpart[(int64_t)b * N + i];
  for (int s = 1; s < S; ++s) v = elem<E>::add(v, Vpart[((int64_t)s * nrhs_pad + b) * N + i]);
  return v;
}

__device__ static inline void cg_scalars_init(cgnr_scalars* sc, double rr, float lambda, float rel_tol, int max_iter) {
  sc->rr = rr;
  sc->z0 = sqrt(rr);
  sc->zeta = 0.0;
  sc->alpha_re = sc->alpha_im = sc->beta_re = sc->beta_im = 0.0;
  sc->lambda = lambda;
  sc->rel_tol = rel_tol;
  sc->iteration = 0;
  sc->max_iter = max_iter;
  sc->pending = 0;
  sc->cur = 0;
  sc->fresh = 0;
  const float ratio = (float)(sqrt(rr) / sqrt(rr));  // NaN when r == 0, as in the reference
  sc->done = (ratio <= rel_tol) || (0 >= max_iter);
}

__device__ static inline void cg_scalars_step(cgnr_scalars* sc, double zeta, double rr, dcomplex alpha, double beta) {
  sc->zeta = zeta;
  sc->rr = rr;
  sc->alpha_re = alpha.re;
  sc->alpha_im = alpha.im;
  sc->beta_re = beta;
  sc->beta_im = 0.0;
  const int it = sc->iteration + 1;
  sc->iteration = it;
  const float ratio = (float)(sqrt(rr) / sc->z0);
  sc->done = (ratio <= sc->rel_tol) || (it >= sc->max_iter);  // src/CGNR.jl:181-185
}

template <typename E, bool INIT, int EPT, int NT>
__global__ __launch_bounds__(NT) void skinny_u_kernel(E* __restrict__ X, E* __restrict__ R, E* __restrict__ P,
                                                               E* __restrict__ V, int64_t ldv,
                                                               const E* __restrict__ Vpart, int S, int nrhs_pad,
                                                               int64_t N, E* __restrict__ Pp, int half,
                                                               cgnr_scalars* scv, float lambda_in, float rel_tol,
                                                               int max_iter) {
  __shared__ double sm[48];
  const int b = blockIdx.x;
  cgnr_scalars* sc = scv + b;
  E* x = X + (int64_t)b * ldv;
  E* r = R + (int64_t)b * ldv;
  E* p = P + (int64_t)b * ldv;
  E* v = V + (int64_t)b * ldv;
  const panel_col<E> pp_out = panel_column<E>(Pp, N, b, half);
  if constexpr (INIT) {
    double rr = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += NT) {
      const E ri = sum_parts<E>(Vpart, S, nrhs_pad, b, N, i);
      x[i] = elem<E>::zero();
      v[i] = elem<E>::zero();
      r[i] = ri;
      p[i] = ri;
      pp_out.put(i, ri);
      rr += abs2d<E>(ri);
    }
    rr = block_sum(rr, sm);
    if (threadIdx.x == 0) cg_scalars_init(sc, rr, lambda_in, rel_tol, max_iter);
  } else if constexpr (EPT > 0) {
    // Every load of this kernel is requested before anything is waited for: the state vectors, the partial rows of
    // V in batches of four splits (independent loads; the additions keep the order s = 0, 1, 2, ...) and the
    // scalars.  Written the obvious way -- scalars, then vectors, then one split after the other -- this was
    // five dependent memory round trips (8.8 us per launch at K = 16).
    const cgnr_scalars S0 = *sc;  // one scalar fetch, requested first, used after the vector loads are out
    E pv[EPT], xv[EPT], rv[EPT], vv[EPT];
    int64_t ic[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = threadIdx.x + (int64_t)e * NT;
      ic[e] = i < N ? i : N - 1;
      pv[e] = p[ic[e]];
      xv[e] = x[ic[e]];
      rv[e] = r[ic[e]];
      vv[e] = elem<E>::zero();
    }
    for (int s0 = 0; s0 < S; s0 += 4) {
      E part[4][EPT];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int sq = s0 + q < S ? s0 + q : S - 1;  // clamped address, masked below
#pragma unroll
        for (int e = 0; e < EPT; ++e) part[q][e] = Vpart[((int64_t)sq * nrhs_pad + b) * N + ic[e]];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (s0 + q < S) {
#pragma unroll
          for (int e = 0; e < EPT; ++e) vv[e] = s0 + q == 0 ? part[q][e] : elem<E>::add(vv[e], part[q][e]);
        }
      }
    }
    if (S0.done) return;  // this column has retired (src/MultiThreading.jl:60-78); its panel entry stays
    const float lambda = S0.lambda;
    const double zeta = S0.rr;
    double nre = 0.0, nim = 0.0, pp = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = threadIdx.x + (int64_t)e * NT;
      if (i >= N) {
        pv[e] = elem<E>::zero();
        vv[e] = elem<E>::zero();
        rv[e] = elem<E>::zero();
      }
      nre += (double)elem<E>::re(pv[e]) * (double)elem<E>::re(vv[e]) + (double)elem<E>::im(pv[e]) * (double)elem<E>::im(vv[e]);
      if constexpr (elem<E>::cplx)
        nim += (double)elem<E>::re(pv[e]) * (double)elem<E>::im(vv[e]) - (double)elem<E>::im(pv[e]) * (double)elem<E>::re(vv[e]);
      pp += abs2d<E>(pv[e]);
    }
    // (wave count as a constant: blockDim.x would be a scalar load from the dispatch packet in host-visible memory)
    if constexpr (NT == 512) block_sum3_nolead<8>(nre, nim, pp, sm);
    else block_sum3_n<NT / 64>(nre, nim, pp, sm);
    const dcomplex den = {nre + (lambda > 0.f ? (double)lambda * pp : 0.0), nim};
    const dcomplex alpha = dc_div({zeta, 0.0}, den);
    const E a = elem<E>::make((float)alpha.re, (float)alpha.im);
    const E na = elem<E>::make(-(float)alpha.re, -(float)alpha.im);
    double rr = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      xv[e] = elem<E>::fma(pv[e], a, xv[e]);
      E ri = elem<E>::fma(vv[e], na, rv[e]);
      if (lambda > 0.f) ri = elem<E>::fma(elem<E>::scale(-lambda, pv[e]), a, ri);
      rv[e] = ri;
      rr += abs2d<E>(ri);
    }
    if constexpr (NT == 512) rr = block_sum_nolead<8>(rr, sm);
    else rr = block_sum_n<NT / 64>(rr, sm);
    const double beta = rr / zeta;
    const float bf = (float)beta;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = threadIdx.x + (int64_t)e * NT;
      if (i < N) {
        const E pn = elem<E>::add(elem<E>::scale(bf, pv[e]), rv[e]);
        x[i] = xv[e];
        r[i] = rv[e];
        v[i] = vv[e];
        p[i] = pn;
        pp_out.put(i, pn);
      }
    }
    if (threadIdx.x == 0) {  // cg_scalars_step on the copy fetched at entry (no second fetch at the tail)
      sc->zeta = zeta;
      sc->rr = rr;
      sc->alpha_re = alpha.re;
      sc->alpha_im = alpha.im;
      sc->beta_re = beta;
      sc->beta_im = 0.0;
      const int it = S0.iteration + 1;
      sc->iteration = it;
      const float ratio = (float)(sqrt(rr) / S0.z0);
      sc->done = (ratio <= S0.rel_tol) || (it >= S0.max_iter);  // src/CGNR.jl:181-185
    }
  } else {
    if (sc->done) return;
    const float lambda = sc->lambda;
    const double zeta = sc->rr;
    double nre = 0.0, nim = 0.0, pp = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += NT) {
      const E vi = sum_parts<E>(Vpart, S, nrhs_pad, b, N, i);
      v[i] = vi;
      const E pi = p[i];
      nre += (double)elem<E>::re(pi) * (double)elem<E>::re(vi) + (double)elem<E>::im(pi) * (double)elem<E>::im(vi);
      if constexpr (elem<E>::cplx)
        nim += (double)elem<E>::re(pi) * (double)elem<E>::im(vi) - (double)elem<E>::im(pi) * (double)elem<E>::re(vi);
      pp += abs2d<E>(pi);
    }
    block_sum3(nre, nim, pp, sm);
    const dcomplex den = {nre + (lambda > 0.f ? (double)lambda * pp : 0.0), nim};
    const dcomplex alpha = dc_div({zeta, 0.0}, den);
    const E a = elem<E>::make((float)alpha.re, (float)alpha.im);
    const E na = elem<E>::make(-(float)alpha.re, -(float)alpha.im);
    double rr = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += NT) {
      const E pi = p[i];
      x[i] = elem<E>::fma(pi, a, x[i]);
      E ri = elem<E>::fma(v[i], na, r[i]);
      if (lambda > 0.f) ri = elem<E>::fma(elem<E>::scale(-lambda, pi), a, ri);
      r[i] = ri;
      rr += abs2d<E>(ri);
    }
    rr = block_sum(rr, sm);
    const double beta = rr / zeta;
    const float bf = (float)beta;
    for (int64_t i = threadIdx.x; i < N; i += NT) {
      const E pn = elem<E>::add(elem<E>::scale(bf, p[i]), r[i]);
      p[i] = pn;
      pp_out.put(i, pn);
    }
    if (threadIdx.x == 0) cg_scalars_step(sc, zeta, rr, alpha, beta);
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// The measurement switches of this file are the context's (rls_tuning: skinny_t_waves / _t_u / _v_waves / _v_u / _v_splits, defaults
// from tools/skinny_probe.py on MI355X; skinny_t_roll / _v_roll / _g_roll; skinny_half; gram_lds).  Rolling-window depth of the
// complex T / V kernels (0 = two-set batch pipeline): MI355X, 4096 x 2048 CF32, (8, 2) against (0, 0): K = 8 32.7 -> 31.6 us,
// K = 16 36.6 -> 33.9 us, K = 64 120.3 -> 98.7 us per batched iteration; (16, 2), (8, 4) and deeper windows measure within noise
// or worse: the kernels are not short of loads in flight.  gram_lds: dynamic LDS requested by the Gram tile kernel purely as an
// occupancy limiter -- one workgroup (one wave per SIMD) per CU keeps the MFMA pipe fed by a single instruction stream (0.92 ms
// vs 1.05 ms with three co-resident workgroups at 4096 x 2048 CF32).

bool rls_skinny_ok(int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda) {
  if (!A || M < 16 || N < 16 || M % 16 || N % 16) return false;
  const int V = dtype == RLS_C32 ? 2 : 4;
  return ((uintptr_t)A % 16 == 0) && (lda % V == 0);
}

// Row splits of the A^H T product: a function of the shape only, NOT of the number of right-hand sides, so
// that a column's arithmetic (and therefore its bits) does not depend on how many columns ride along.
static int skinny_splits(const rls_ctx* ctx, int64_t M, int64_t N, int /*ngroups*/) {
  if (ctx->tune.skinny_v_splits > 0) return ctx->tune.skinny_v_splits;
  const int64_t MB = M / 16, wgs = N / 16;
  int64_t S = (512 + wgs - 1) / wgs;               // ~2 workgroups of 4 waves per CU for one group
  const int64_t smax = MB / 16 > 0 ? MB / 16 : 1;  // keep >= 16 row blocks (4 per wave) per split
  if (S > smax) S = smax;
  if (S > 16) S = 16;
  if (S < 1) S = 1;
  return (int)S;
}

int rls_skinny_half(const rls_ctx* ctx, int32_t dtype, int nrhs) { return ctx->tune.skinny_half && dtype == RLS_C32 && nrhs <= 8; }

void rls_skinny_sizes(const rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, int nrhs, size_t* p_bytes, size_t* t_bytes,
                      size_t* v_bytes, int* splits) {
  const int half = rls_skinny_half(ctx, dtype, nrhs);
  const int G = rls_skinny_groups(nrhs, half);
  const int S = skinny_splits(ctx, M, N, G);
  // the half layout holds 16 floats (8 re | 8 im) per row where the full one holds 16 elements; sized for the
  // full layout either way so that the switch can be flipped on a live plan by the measurement tools
  *p_bytes = (size_t)G * N * 16 * rls_elem_size(dtype);
  *t_bytes = (size_t)G * M * 16 * rls_elem_size(dtype);
  *v_bytes = (size_t)S * G * 16 * N * rls_elem_size(dtype);
  *splits = S;
}

static int32_t sk_status(rls_ctx* ctx) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

template <typename E>
static void launch_t(rls_ctx* ctx, const rls_skinny& K) {
  const dim3 grid((unsigned)(K.M / 16), (unsigned)K.ngroups);
#define SK_T(W, UU, HH)                                                                                            \
  if (ctx->tune.skinny_t_waves == W && ctx->tune.skinny_t_u == UU) {                                                                             \
    hipLaunchKernelGGL((skinny_t_kernel<E, W, UU, HH>), grid, dim3(W * 64), 0, ctx->stream, (const E*)K.A, K.lda, \
                       (const E*)K.Ppack, (E*)K.Tpack, K.M, K.N);                                                  \
    return;                                                                                                        \
  }
#define SK_TR(HH, DD)                                                                                              \
  if (ctx->tune.skinny_t_roll == DD && (K.half != 0) == HH) {                                                                     \
    if (ctx->tune.skinny_t_waves == 8)                                                                                            \
      hipLaunchKernelGGL((skinny_t_kernel<E, 8, 4, HH, DD>), grid, dim3(512), 0, ctx->stream, (const E*)K.A, K.lda, \
                         (const E*)K.Ppack, (E*)K.Tpack, K.M, K.N);                                                \
    else                                                                                                           \
      hipLaunchKernelGGL((skinny_t_kernel<E, 4, 4, HH, DD>), grid, dim3(256), 0, ctx->stream, (const E*)K.A, K.lda, \
                         (const E*)K.Ppack, (E*)K.Tpack, K.M, K.N);                                                \
    return;                                                                                                        \
  }
  if constexpr (elem<E>::cplx) {
    SK_TR(true, 8) SK_TR(true, 16) SK_TR(true, 32) SK_TR(false, 8) SK_TR(false, 16)
  }
#undef SK_TR
  if constexpr (elem<E>::cplx) {
    if (K.half) {
      SK_T(8, 4, true) SK_T(8, 8, true) SK_T(4, 8, true) SK_T(4, 2, true) SK_T(4, 16, true) SK_T(2, 4, true)
      SK_T(8, 2, true) SK_T(8, 16, true)
      hipLaunchKernelGGL((skinny_t_kernel<E, 4, 4, true>), grid, dim3(256), 0, ctx->stream, (const E*)K.A, K.lda,
                         (const E*)K.Ppack, (E*)K.Tpack, K.M, K.N);
      return;
    }
  }
  SK_T(8, 4, false) SK_T(8, 8, false) SK_T(4, 8, false) SK_T(4, 2, false) SK_T(4, 16, false) SK_T(2, 4, false)
#undef SK_T
  hipLaunchKernelGGL((skinny_t_kernel<E, 4, 4, false>), grid, dim3(256), 0, ctx->stream, (const E*)K.A, K.lda,
                     (const E*)K.Ppack, (E*)K.Tpack, K.M, K.N);
}

template <typename E>
static void launch_v(rls_ctx* ctx, const rls_skinny& K) {
  const dim3 grid((unsigned)(K.N / 16), (unsigned)K.splits, (unsigned)K.ngroups);
  const int pad = rls_skinny_pad(K.nrhs, K.half);
#define SK_V(W, UU, HH)                                                                                            \
  if (ctx->tune.skinny_v_waves == W && ctx->tune.skinny_v_u == UU) {                                                                             \
    hipLaunchKernelGGL((skinny_v_kernel<E, W, UU, HH>), grid, dim3(W * 64), 0, ctx->stream, (const E*)K.A, K.lda, \
                       (const E*)K.Tpack, (E*)K.Vpart, K.M, K.N, pad, K.ldvp);                                     \
    return;                                                                                                        \
  }
#define SK_VR(HH, DD)                                                                                              \
  if (ctx->tune.skinny_v_roll == DD && (K.half != 0) == HH) {                                                                     \
    if (ctx->tune.skinny_v_waves == 8)                                                                                            \
      hipLaunchKernelGGL((skinny_v_kernel<E, 8, 1, HH, DD>), grid, dim3(512), 0, ctx->stream, (const E*)K.A, K.lda, \
                         (const E*)K.Tpack, (E*)K.Vpart, K.M, K.N, pad, K.ldvp);                                   \
    else                                                                                                           \
      hipLaunchKernelGGL((skinny_v_kernel<E, 4, 1, HH, DD>), grid, dim3(256), 0, ctx->stream, (const E*)K.A, K.lda, \
                         (const E*)K.Tpack, (E*)K.Vpart, K.M, K.N, pad, K.ldvp);                                   \
    return;                                                                                                        \
  }
  if constexpr (elem<E>::cplx) {
    SK_VR(true, 2) SK_VR(true, 4) SK_VR(true, 8) SK_VR(false, 2) SK_VR(false, 4) SK_VR(false, 8)
  }
#undef SK_VR
  if constexpr (elem<E>::cplx) {
    if (K.half) {
      SK_V(8, 1, true) SK_V(8, 2, true) SK_V(4, 2, true) SK_V(4, 4, true) SK_V(2, 1, true) SK_V(2, 2, true)
      hipLaunchKernelGGL((skinny_v_kernel<E, 4, 1, true>), grid, dim3(256), 0, ctx->stream, (const E*)K.A, K.lda,
                         (const E*)K.Tpack, (E*)K.Vpart, K.M, K.N, pad, K.ldvp);
      return;
    }
  }
  SK_V(8, 1, false) SK_V(8, 2, false) SK_V(4, 2, false) SK_V(4, 4, false) SK_V(2, 1, false) SK_V(2, 2, false)
#undef SK_V
  hipLaunchKernelGGL((skinny_v_kernel<E, 4, 1, false>), grid, dim3(256), 0, ctx->stream, (const E*)K.A, K.lda,
                     (const E*)K.Tpack, (E*)K.Vpart, K.M, K.N, pad, K.ldvp);
}

// Gram mode: V = AHA P as ONE product over the explicit N x N matrix (K.G), the contraction split K.splits ways
template <typename E>
static void launch_g(rls_ctx* ctx, const rls_skinny& K) {
  const dim3 grid((unsigned)(K.N / 16), (unsigned)K.ngroups, (unsigned)K.splits);
  const int pad = rls_skinny_pad(K.nrhs, K.half);
#define SK_G(HH, DD)                                                                                                   \
  hipLaunchKernelGGL((skinny_t_kernel<E, 4, 4, HH, DD, true>), grid, dim3(256), 0, ctx->stream, (const E*)K.G, K.ldg, \
                     (const E*)K.Ppack, (E*)K.Vpart, K.N, K.N, pad, K.ldvp)
  if constexpr (elem<E>::cplx) {
    if (K.half) {
      if (ctx->tune.skinny_g_roll == 16) SK_G(true, 16);
      else if (ctx->tune.skinny_g_roll == 0) SK_G(true, 0);
      else SK_G(true, 8);
    } else {
      if (ctx->tune.skinny_g_roll == 16) SK_G(false, 16);
      else if (ctx->tune.skinny_g_roll == 0) SK_G(false, 0);
      else SK_G(false, 8);
    }
  } else {
    SK_G(false, 0);
  }
#undef SK_G
}

template <typename E, bool INIT, int EPT, int NT>
static void launch_u_ept(rls_ctx* ctx, const rls_skinny& K, float lambda, float rel_tol, int max_iter) {
  hipLaunchKernelGGL((skinny_u_kernel<E, INIT, EPT, NT>), dim3((unsigned)K.nrhs), dim3(NT), 0, ctx->stream,
                     (E*)K.X, (E*)K.R, (E*)K.P, (E*)K.V, K.ldv, (const E*)K.Vpart, K.splits,
                     rls_skinny_pad(K.nrhs, K.half), K.N, (E*)K.Ppack, K.half, K.sc, lambda, rel_tol, max_iter);
}

template <typename E, bool INIT>
static void launch_u(rls_ctx* ctx, const rls_skinny& K, float lambda, float rel_tol, int max_iter) {
  if (INIT || K.N > 8 * 1024)
    launch_u_ept<E, INIT, 0, 1024>(ctx, K, lambda, rel_tol, max_iter);
  else if (K.N <= 2 * 512)
    launch_u_ept<E, INIT, 2, 512>(ctx, K, lambda, rel_tol, max_iter);
  else if (K.N <= 4 * 512)
    launch_u_ept<E, INIT, 4, 512>(ctx, K, lambda, rel_tol, max_iter);
  else if (K.N <= 8 * 512)
    launch_u_ept<E, INIT, 8, 512>(ctx, K, lambda, rel_tol, max_iter);
  else if constexpr (!elem<E>::cplx)
    launch_u_ept<E, INIT, 8, 1024>(ctx, K, lambda, rel_tol, max_iter);
  else  // complex columns of more than 4096 elements: 8 per thread at 1024 threads (128 VGPRs) spilled 148 bytes per lane
    launch_u_ept<E, INIT, 0, 1024>(ctx, K, lambda, rel_tol, max_iter);
}

template <typename E>
static void launch_pack_rows(rls_ctx* ctx, const rls_skinny& K, const E* B, int64_t ldb) {
  const dim3 grid((unsigned)((K.M + 63) / 64), (unsigned)K.ngroups);
  if constexpr (elem<E>::cplx) {
    if (K.half) {
      hipLaunchKernelGGL((skinny_pack_rows_kernel<E, true>), grid, dim3(256), 0, ctx->stream, B, ldb, K.nrhs,
                         (E*)K.Tpack, K.M, K.ngroups);
      return;
    }
  }
  hipLaunchKernelGGL((skinny_pack_rows_kernel<E, false>), grid, dim3(256), 0, ctx->stream, B, ldb, K.nrhs, (E*)K.Tpack,
                     K.M, K.ngroups);
}

template <typename E>
static int32_t skinny_init_typed(rls_ctx* ctx, const rls_skinny& K, const void* B, int64_t ldb, float lambda,
                                 float rel_tol, int max_iter) {
  launch_pack_rows<E>(ctx, K, (const E*)B, ldb);
  launch_v<E>(ctx, K);
  launch_u<E, true>(ctx, K, lambda, rel_tol, max_iter);
  return sk_status(ctx);
}

int32_t rls_skinny_init(rls_ctx* ctx, int32_t dtype, const rls_skinny& K, const void* B, int64_t ldb, float lambda,
                        float rel_tol, int max_iter) {
  return dtype == RLS_F32 ? skinny_init_typed<float>(ctx, K, B, ldb, lambda, rel_tol, max_iter)
                          : skinny_init_typed<float2>(ctx, K, B, ldb, lambda, rel_tol, max_iter);
}

// partial rows of A^H B into K.Vpart (B: M x nrhs column-major): the init product of the batched plans
int32_t rls_skinny_atb(rls_ctx* ctx, int32_t dtype, const rls_skinny& K, const void* B, int64_t ldb) {
  if (dtype == RLS_F32) {
    launch_pack_rows<float>(ctx, K, (const float*)B, ldb);
    launch_v<float>(ctx, K);
  } else {
    launch_pack_rows<float2>(ctx, K, (const float2*)B, ldb);
    launch_v<float2>(ctx, K);
  }
  return sk_status(ctx);
}

// which: bit 0 = T kernel, bit 1 = V kernel, bit 2 = update kernel
int32_t rls_skinny_launch(rls_ctx* ctx, int32_t dtype, const rls_skinny& K, int which) {
  if (K.G && (which & 3)) {  // explicit AHA: the two products over A are one product over the Gram matrix
    if (dtype == RLS_F32) launch_g<float>(ctx, K);
    else launch_g<float2>(ctx, K);
    which &= ~3;
  }
  if (dtype == RLS_F32) {
    if (which & 1) launch_t<float>(ctx, K);
    if (which & 2) launch_v<float>(ctx, K);
    if (which & 4) launch_u<float, false>(ctx, K, 0.f, 0.f, 0);
  } else {
    if (which & 1) launch_t<float2>(ctx, K);
    if (which & 2) launch_v<float2>(ctx, K);
    if (which & 4) launch_u<float2, false>(ctx, K, 0.f, 0.f, 0);
  }
  return sk_status(ctx);
}

// ---------------------------------------------------------------------------------------------
// G = A^H A as a Hermitian rank-M update on the matrix cores (setup GEMM of src/CGNR.jl:49, src/FISTA.jl:58,
// src/ADMM.jl:82).  One workgroup = one 64 x 64 tile of the UPPER triangle (the mirror image is written as
// the conjugate), 4 waves = 2 x 2 sub-tiles of 32 x 32, each two 16-column blocks on either side.  Both
// MFMA operands are read straight from A in the same register layout (lane l: column 16 cb + (l & 15),
// rows 16 mb + 4 (l >> 4) .. + 3 = one 32-byte piece), so nothing is staged or packed; the 64-column strips
// come out of L2 / the Infinity Cache.  Entry (i, j) and entry (j, i) run the same products in the same
// order, so the result is Hermitian bit for bit.
// ---------------------------------------------------------------------------------------------
template <typename E>
struct g_regs {
  float4 x0, x1;  // complex: rows (0,1), (2,3) as (re, im) pairs; real: x0 = rows 0..3
};

template <typename E>
__device__ static inline void g_load(g_regs<E> (&r)[4], const E* const (&col)[4], int64_t mb) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float4* ap = reinterpret_cast<const float4*>(col[c] + mb * 16);
    r[c].x0 = ap[0];
    if constexpr (elem<E>::cplx) r[c].x1 = ap[1];
  }
}

template <typename E>
__device__ static inline void g_mma(f32x4 (&acc)[2][2][4], const g_regs<E> (&r)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if constexpr (elem<E>::cplx) {
          const float4 ia = t < 2 ? r[a].x0 : r[a].x1, jb = t < 2 ? r[2 + b].x0 : r[2 + b].x1;
          const float ar = (t & 1) ? ia.z : ia.x, ai = (t & 1) ? ia.w : ia.y;
          const float br = (t & 1) ? jb.z : jb.x, bi = (t & 1) ? jb.w : jb.y;
          acc[a][b][0] = mfma4(ar, br, acc[a][b][0]);
          acc[a][b][1] = mfma4(ai, bi, acc[a][b][1]);
          acc[a][b][2] = mfma4(ar, bi, acc[a][b][2]);
          acc[a][b][3] = mfma4(ai, br, acc[a][b][3]);
        } else {
          const float ia[4] = {r[a].x0.x, r[a].x0.y, r[a].x0.z, r[a].x0.w};
          const float jb[4] = {r[2 + b].x0.x, r[2 + b].x0.y, r[2 + b].x0.z, r[2 + b].x0.w};
          acc[a][b][t & 1] = mfma4(ia[t], jb[t], acc[a][b][t & 1]);
        }
      }
    }
  }
}

template <typename E>
__global__ __launch_bounds__(256) void gram_mfma_kernel(const E* __restrict__ A, int64_t lda, E* __restrict__ G,
                                                        int64_t ldg, int64_t M, int64_t N) {
  constexpr bool CX = elem<E>::cplx;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // tile index -> (bi <= bj) of the upper triangle, T tiles per side
  const int T = (int)(N / 64);
  int bj = (int)((sqrtf(8.f * (float)blockIdx.x + 1.f) - 1.f) * 0.5f);
  while ((bj + 1) * (bj + 2) / 2 <= (int)blockIdx.x) ++bj;
  while (bj * (bj + 1) / 2 > (int)blockIdx.x) --bj;
  const int bi = (int)blockIdx.x - bj * (bj + 1) / 2;
  (void)T;
  const int wi = w >> 1, wj = w & 1;
  const int ib0 = bi * 4 + 2 * wi, jb0 = bj * 4 + 2 * wj;  // first 16-column block on either side
  const E* col[4];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    col[c] = A + ((int64_t)(ib0 + c) * 16 + (lane & 15)) * lda + 4 * (lane >> 4);
    col[2 + c] = A + ((int64_t)(jb0 + c) * 16 + (lane & 15)) * lda + 4 * (lane >> 4);
  }
  f32x4 acc[2][2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[a][b][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t MB = M / 16;
  g_regs<E> r0[4], r1[4];
  g_load<E>(r0, col, 0);
  int64_t mb = 0;
  for (; mb + 2 < MB; mb += 2) {
    g_load<E>(r1, col, mb + 1);
    __builtin_amdgcn_sched_barrier(0);
    g_mma<E>(acc, r0);
    __builtin_amdgcn_sched_barrier(0);
    g_load<E>(r0, col, mb + 2);
    __builtin_amdgcn_sched_barrier(0);
    g_mma<E>(acc, r1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (mb + 1 < MB) {
    g_load<E>(r1, col, mb + 1);
    g_mma<E>(acc, r0);
    g_mma<E>(acc, r1);
  } else {
    g_mma<E>(acc, r0);
  }
  // accumulator register u of lane (q = l >> 4, j' = l & 15) is entry (row 4 q + u, column j') of the block
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      f32x4 gre, gim;
      if constexpr (CX) {
        gre = acc[a][b][0] + acc[a][b][1];
        gim = acc[a][b][2] - acc[a][b][3];
      } else {
        gre = acc[a][b][0] + acc[a][b][1];
        gim = gre;
      }
      const int64_t j = (int64_t)(jb0 + b) * 16 + (lane & 15);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t i = (int64_t)(ib0 + a) * 16 + 4 * (lane >> 4) + u;
        G[i + j * ldg] = elem<E>::make(gre[u], gim[u]);
        G[j + i * ldg] = elem<E>::make(gre[u], -gim[u]);  // the mirror image (for a diagonal tile: the same values)
      }
    }
  }
}

// G = A^H A (setup GEMM of src/CGNR.jl:49) on the matrix cores: the A^H T product with T = A, every
// 16 columns of A forming one panel; `panels` is an M x N scratch in the operand layout.
template <typename E>
static int32_t skinny_gram_typed(rls_ctx* ctx, int64_t M, int64_t N, const E* A, int64_t lda, E* G, int64_t ldg,
                                 E* panels) {
  rls_skinny K{};
  K.A = A;
  K.lda = lda;
  K.M = M;
  K.N = N;
  K.nrhs = (int)N;
  K.ngroups = (int)(N / 16);
  K.splits = 1;
  K.Tpack = (float*)panels;
  K.Vpart = G;
  K.ldvp = ldg;
  const dim3 grid((unsigned)((M + 63) / 64), (unsigned)K.ngroups);
  hipLaunchKernelGGL((skinny_pack_rows_kernel<E, false>), grid, dim3(256), 0, ctx->stream, A, lda, (int)N, panels, M,
                     K.ngroups);
  launch_v<E>(ctx, K);
  return sk_status(ctx);
}

bool rls_gram_tiles_ok(int64_t M, int64_t N) { return N % 64 == 0 && M % 16 == 0 && N / 64 <= 2000; }

int32_t rls_gram_tiles(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda, void* G,
                       int64_t ldg) {
  const int64_t T = N / 64;
  const dim3 grid((unsigned)(T * (T + 1) / 2));
  const size_t lds = (size_t)ctx->tune.gram_lds;  // occupancy limiter (rls_tuning::gram_lds)
  if (lds > 64 * 1024) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_mfma_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_mfma_kernel<float2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(gram_mfma_kernel<float>, grid, dim3(256), lds, ctx->stream, (const float*)A, lda, (float*)G, ldg, M, N);
  else
    hipLaunchKernelGGL(gram_mfma_kernel<float2>, grid, dim3(256), lds, ctx->stream, (const float2*)A, lda, (float2*)G, ldg,
                       M, N);
  return sk_status(ctx);
}

int32_t rls_skinny_gram(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda, void* G,
                        int64_t ldg, void* panels) {
  return dtype == RLS_F32 ? skinny_gram_typed<float>(ctx, M, N, (const float*)A, lda, (float*)G, ldg, (float*)panels)
                          : skinny_gram_typed<float2>(ctx, M, N, (const float2*)A, lda, (float2*)G, ldg, (float2*)panels);
}
