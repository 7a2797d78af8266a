// Dense GEMV kernels for gfx950: y = alpha * op(A) * x + beta * y on a column-major A.
//
// These replace LinearAlgebra.mul!(y, A, x) / mul!(x, adjoint(A), y) on the reference's hot path
// (src/CGNR.jl:132,151  src/FISTA.jl:114,152  src/ADMM.jl:198, inside cg! from src/ADMM.jl:244).
// Both are HBM-bound (1 flop/B complex, 0.5 flop/B real): no LDS staging of A (read once), 16-byte
// loads straight to VGPRs with many in flight, wave-shuffle + LDS reductions, deterministic sums.
//
//  gemv_t (op = T / C):  y[j] = sum_i op(A[i,j]) x[i].  Columns are contiguous: a workgroup owns
//      COLS columns, its threads stride down the rows with 16-byte loads, x is loaded once per
//      workgroup into registers and reused for all COLS columns.
//  gemv_n (op = N):      y[i] = sum_j A[i,j] x[j].  Coalescing runs along rows, so a group of G
//      lanes reads G*16 contiguous bytes of one column while the 64/G groups of a wave (and the
//      WAVES waves of the workgroup) work on different columns; partial sums are combined with
//      wave shuffles and one LDS pass, so no cross-workgroup reduction (and no atomics) is needed.
#include "rls_common.hpp"

// ---------------------------------------------------------------------------------------------
// gemv_t
// ---------------------------------------------------------------------------------------------
template <typename E, bool CONJ, int NV, int COLS, int WAVES, int U>
__global__ __launch_bounds__(WAVES * 64) void gemv_t_kernel(const E* __restrict__ A, int64_t lda,
                                                            const E* __restrict__ x, E* __restrict__ y,
                                                            int64_t Mc, int64_t N, E alpha, E beta,
                                                            const int* __restrict__ skip, int reverse) {
  if (skip && *skip) return;
  constexpr int THREADS = WAVES * 64;
  const int tid = threadIdx.x;
  // `reverse`: the first workgroups take the LAST columns (which column a workgroup owns changes no bit of any result)
  const int64_t j0 = (int64_t)(reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * COLS;

  E acc[COLS];
#pragma unroll
  for (int c = 0; c < COLS; ++c) acc[c] = elem<E>::zero();

  // Loads must never sit under a branch: hipcc then waits vmcnt(0) after each one and a wave has a
  // single 16-byte load in flight.  Full tiles run unpredicated; the ragged tail clamps its
  // addresses (always-valid loads) and zeroes the dead lanes with selects.
  const int64_t span = (int64_t)THREADS * U;
  const int64_t full = (Mc / span) * span;
  for (int64_t base = 0; base < full; base += span) {
    chunk<E, NV> xr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) xr[u] = load_chunk<E, NV>(x + (base + (int64_t)u * THREADS + tid) * NV);
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
      const int64_t j = (j0 + c < N) ? (j0 + c) : (N - 1);
      const E* col = A + j * lda + (base + tid) * NV;
      chunk<E, NV> a[U];
#pragma unroll
      for (int u = 0; u < U; ++u) a[u] = load_chunk<E, NV>(col + (int64_t)u * THREADS * NV);
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          if constexpr (CONJ)
            acc[c] = elem<E>::fmac_pk(a[u].e[i], xr[u].e[i], acc[c]);
          else
            acc[c] = elem<E>::fma_pk(a[u].e[i], xr[u].e[i], acc[c]);
        }
      }
    }
  }
  if (full < Mc) {
    chunk<E, NV> xr[U];
    int64_t idc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t idx = full + (int64_t)u * THREADS + tid;
      const bool ok = idx < Mc;
      idc[u] = ok ? idx : (Mc - 1);
      xr[u] = load_chunk<E, NV>(x + idc[u] * NV);
      if (!ok) xr[u] = zero_chunk<E, NV>();
    }
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
      const int64_t j = (j0 + c < N) ? (j0 + c) : (N - 1);
      const E* col = A + j * lda;
      chunk<E, NV> a[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        a[u] = load_chunk<E, NV>(col + idc[u] * NV);
        if (full + (int64_t)u * THREADS + tid >= Mc) a[u] = zero_chunk<E, NV>();
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          if constexpr (CONJ)
            acc[c] = elem<E>::fmac_pk(a[u].e[i], xr[u].e[i], acc[c]);
          else
            acc[c] = elem<E>::fma_pk(a[u].e[i], xr[u].e[i], acc[c]);
        }
      }
    }
  }

  // block reduction: shuffle tree inside each wave, then fixed-order sum over waves through LDS
  __shared__ float red[WAVES][COLS][2];
  const int lane = tid & 63, w = tid >> 6;
#pragma unroll
  for (int c = 0; c < COLS; ++c) {
    float re = wave_sum(elem<E>::re(acc[c]));
    float im = elem<E>::cplx ? wave_sum(elem<E>::im(acc[c])) : 0.f;
    if (lane == 0) {
      red[w][c][0] = re;
      red[w][c][1] = im;
    }
  }
  __syncthreads();
  if (tid < COLS && j0 + tid < N) {
    float re = 0.f, im = 0.f;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) {
      re += red[i][tid][0];
      im += red[i][tid][1];
    }
    E s = elem<E>::make(re, im);
    E out = elem<E>::mul(alpha, s);
    if (elem<E>::re(beta) != 0.f || elem<E>::im(beta) != 0.f) out = elem<E>::fma(beta, y[j0 + tid], out);
    y[j0 + tid] = out;
  }
}

// ---------------------------------------------------------------------------------------------
// gemv_n
// ---------------------------------------------------------------------------------------------
template <typename E, int NV, int G, int WAVES, int U>
__global__ __launch_bounds__(WAVES * 64) void gemv_n_kernel(const E* __restrict__ A, int64_t lda,
                                                            const E* __restrict__ x, E* __restrict__ y,
                                                            int64_t Mc, int64_t M, int64_t N, E alpha, E beta,
                                                            const int* __restrict__ skip) {
  if (skip && *skip) return;
  constexpr int THREADS = WAVES * 64;
  constexpr int S = 64 / G;           // column slots per wave
  constexpr int CPR = WAVES * S;      // columns per round of the whole workgroup
  constexpr int TILE = 16384 / sizeof(E);  // x staged through LDS 16 KiB at a time
  __shared__ E xs[TILE];
  __shared__ E part[WAVES][G][NV];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane % G, s = lane / G;
  const int64_t chunk_id = (int64_t)blockIdx.x * G + g;
  const int64_t chunk_c = chunk_id < Mc ? chunk_id : (Mc - 1);
  const E* Ab = A + chunk_c * NV;

  E acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = elem<E>::zero();

  for (int64_t t0 = 0; t0 < N; t0 += TILE) {
    const int nt = (int)((N - t0) < TILE ? (N - t0) : TILE);
    const E* At = Ab + t0 * lda;
    const int slot = w * S + s;
    const int full_rounds = nt / CPR;
    const int main_rounds = (full_rounds / U) * U;
    // The x tile goes to registers first and the first U loads of A go out right behind it, BEFORE the tile is
    // written to LDS: the staging then hides under the first loads' latency instead of preceding it (the barriers
    // order LDS only -- __syncthreads() would wait for the A loads as well).  Clamped columns, no branch around loads.
    constexpr int XPT = (TILE + THREADS - 1) / THREADS;
    E xr[XPT];
#pragma unroll
    for (int j = 0; j < XPT; ++j) {
      const int i = tid + j * THREADS;
      xr[j] = x[t0 + (i < nt ? i : nt - 1)];
    }
    chunk<E, NV> a0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int jl = u * CPR + slot;
      a0[u] = load_chunk<E, NV>(At + (int64_t)(jl < nt ? jl : nt - 1) * lda);
    }
    lds_barrier();  // the previous tile's readers are done
#pragma unroll
    for (int j = 0; j < XPT; ++j) {
      const int i = tid + j * THREADS;
      if (i < nt) xs[i] = xr[j];
    }
    lds_barrier();
    if (main_rounds > 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const E xv = xs[u * CPR + slot];
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = elem<E>::fma_pk(a0[u].e[i], xv, acc[i]);
      }
    }
    // full rounds (every slot of every wave has a valid column) run unpredicated, U loads in
    // flight per lane; the remainder clamps the column and zeroes x instead of branching
    for (int k = U; k < main_rounds; k += U) {
      chunk<E, NV> a[U];
      E xv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jl = (k + u) * CPR + slot;
        a[u] = load_chunk<E, NV>(At + (int64_t)jl * lda);
        xv[u] = xs[jl];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = elem<E>::fma_pk(a[u].e[i], xv[u], acc[i]);
      }
    }
    const int rounds = (nt + CPR - 1) / CPR;
    for (int k = main_rounds; k < rounds; ++k) {
      const int jl = k * CPR + slot;
      const bool ok = jl < nt;
      const int jc = ok ? jl : (nt - 1);
      chunk<E, NV> a = load_chunk<E, NV>(At + (int64_t)jc * lda);
      E xv = xs[jc];
      if (!ok) {
        xv = elem<E>::zero();
        a = zero_chunk<E, NV>();
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) acc[i] = elem<E>::fma_pk(a.e[i], xv, acc[i]);
    }
  }

  // combine the S column slots of the wave (lanes that share g), then the waves through LDS
#pragma unroll
  for (int off = G; off < 64; off <<= 1) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float re = add_xor(elem<E>::re(acc[i]), off);
      float im = elem<E>::cplx ? add_xor(elem<E>::im(acc[i]), off) : 0.f;
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
    const int64_t row = ((int64_t)blockIdx.x * G + gg) * NV + i;
    if (row < M) {
      E sum = elem<E>::zero();
#pragma unroll
      for (int ww = 0; ww < WAVES; ++ww) sum = elem<E>::add(sum, part[ww][gg][i]);
      E out = elem<E>::mul(alpha, sum);
      if (elem<E>::re(beta) != 0.f || elem<E>::im(beta) != 0.f) out = elem<E>::fma(beta, y[row], out);
      y[row] = out;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------------
template <typename E, bool CONJ, int NV, int COLS>
static void launch_t(rls_ctx* ctx, const E* A, int64_t lda, const E* x, E* y, int64_t M, int64_t N, E alpha, E beta,
                     const int* skip) {
  constexpr int WAVES = 4, U = 8;
  const int64_t Mc = M / NV;
  const unsigned grid = (unsigned)((N + COLS - 1) / COLS);
  const bool out_of_cache = (double)M * (double)N * (double)sizeof(E) > 256.0 * 1024 * 1024;
  const int reverse = ctx->tune.gemvt_reverse < 0 ? (out_of_cache ? 1 : 0) : (ctx->tune.gemvt_reverse ? 1 : 0);
  hipLaunchKernelGGL((gemv_t_kernel<E, CONJ, NV, COLS, WAVES, U>), dim3(grid), dim3(WAVES * 64), 0, ctx->stream, A,
                     lda, x, y, Mc, N, alpha, beta, skip, reverse);
}

template <typename E, bool CONJ, int NV>
static void dispatch_t(rls_ctx* ctx, const E* A, int64_t lda, const E* x, E* y, int64_t M, int64_t N, E alpha,
                       E beta, const int* skip) {
  int cols = ctx->tune.gemvt_cols;
  if (cols == 0) {
    // keep >= ~2 workgroups per CU while amortising the x loads over several columns
    cols = N >= 4096 ? 8 : (N >= 2048 ? 4 : (N >= 1024 ? 2 : 1));
    // a matrix beyond the Infinity Cache streams from HBM: there two columns per workgroup are faster (8192 x 8192
    // CF32, the config-5 shard: 86.9 us at 2 columns, 93.7 at 4, 100.8 at 8; tools/tune_gemv.py)
    if ((double)M * (double)N * (double)sizeof(E) > 256.0 * 1024 * 1024 && cols > 2) cols = 2;
  }
  switch (cols) {
    case 8: launch_t<E, CONJ, NV, 8>(ctx, A, lda, x, y, M, N, alpha, beta, skip); break;
    case 4: launch_t<E, CONJ, NV, 4>(ctx, A, lda, x, y, M, N, alpha, beta, skip); break;
    case 2: launch_t<E, CONJ, NV, 2>(ctx, A, lda, x, y, M, N, alpha, beta, skip); break;
    default: launch_t<E, CONJ, NV, 1>(ctx, A, lda, x, y, M, N, alpha, beta, skip); break;
  }
}

template <typename E, int NV, int G, int WAVES>
static void launch_n(rls_ctx* ctx, const E* A, int64_t lda, const E* x, E* y, int64_t M, int64_t N, E alpha, E beta,
                     const int* skip) {
  constexpr int U = 8;
  const int64_t Mc = (M + NV - 1) / NV;
  const unsigned grid = (unsigned)((Mc + G - 1) / G);
  hipLaunchKernelGGL((gemv_n_kernel<E, NV, G, WAVES, U>), dim3(grid), dim3(WAVES * 64), 0, ctx->stream, A, lda, x, y,
                     Mc, M, N, alpha, beta, skip);
}

template <typename E, int NV, int G>
static void dispatch_n_w(rls_ctx* ctx, int waves, const E* A, int64_t lda, const E* x, E* y, int64_t M, int64_t N,
                         E alpha, E beta, const int* skip) {
  switch (waves) {
    case 16: launch_n<E, NV, G, 16>(ctx, A, lda, x, y, M, N, alpha, beta, skip); break;
    case 8: launch_n<E, NV, G, 8>(ctx, A, lda, x, y, M, N, alpha, beta, skip); break;
    default: launch_n<E, NV, G, 4>(ctx, A, lda, x, y, M, N, alpha, beta, skip); break;
  }
}

template <typename E, int NV>
static void dispatch_n(rls_ctx* ctx, const E* A, int64_t lda, const E* x, E* y, int64_t M, int64_t N, E alpha,
                       E beta, const int* skip) {
  const int64_t Mc = (M + NV - 1) / NV;
  int G = ctx->tune.gemvn_g, waves = ctx->tune.gemvn_waves;
  if (G == 0) {
    // widest contiguous run per column that still yields >= 256 workgroups (one per CU)
    G = 8;
    for (int cand : {64, 32, 16}) {
      if ((Mc + cand - 1) / cand >= 256) {
        G = cand;
        break;
      }
    }
  }
  const int64_t rb = (Mc + G - 1) / G;
  if (waves == 0) waves = rb <= 256 ? 16 : (rb <= 512 ? 8 : 4);
  // out-of-cache matrices (see dispatch_t): 4 waves per workgroup stream best (8192 x 8192 CF32, G = 16: 87.8 us at 4
  // waves, 92.7 at 8, 96.1 at 16)
  if (ctx->tune.gemvn_waves == 0 && (double)M * (double)N * (double)sizeof(E) > 256.0 * 1024 * 1024) waves = 4;
  if constexpr (NV == 1) {
    dispatch_n_w<E, NV, 64>(ctx, waves, A, lda, x, y, M, N, alpha, beta, skip);
  } else {
    switch (G) {
      case 64: dispatch_n_w<E, NV, 64>(ctx, waves, A, lda, x, y, M, N, alpha, beta, skip); break;
      case 32: dispatch_n_w<E, NV, 32>(ctx, waves, A, lda, x, y, M, N, alpha, beta, skip); break;
      case 16: dispatch_n_w<E, NV, 16>(ctx, waves, A, lda, x, y, M, N, alpha, beta, skip); break;
      default: dispatch_n_w<E, NV, 8>(ctx, waves, A, lda, x, y, M, N, alpha, beta, skip); break;
    }
  }
}

template <typename E>
static int32_t gemv_typed(rls_ctx* ctx, int32_t op, int64_t M, int64_t N, E alpha, E beta, const E* A, int64_t lda,
                          const E* x, E* y, const int* skip) {
  constexpr int V = elem<E>::vec;
  const bool vec_ok = (reinterpret_cast<uintptr_t>(A) % 16 == 0) && (lda % V == 0);
  if (op == RLS_OP_N) {
    // rows need not be a multiple of V: the last chunk is clamped on load and masked on store only
    // when M % V == 0; otherwise fall back to element-granular loads.
    if (vec_ok && M % V == 0)
      dispatch_n<E, V>(ctx, A, lda, x, y, M, N, alpha, beta, skip);
    else
      dispatch_n<E, 1>(ctx, A, lda, x, y, M, N, alpha, beta, skip);
  } else {
    const bool conj = (op == RLS_OP_C) && elem<E>::cplx;
    const bool v = vec_ok && (M % V == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
    if (conj) {
      if (v)
        dispatch_t<E, true, V>(ctx, A, lda, x, y, M, N, alpha, beta, skip);
      else
        dispatch_t<E, true, 1>(ctx, A, lda, x, y, M, N, alpha, beta, skip);
    } else {
      if (v)
        dispatch_t<E, false, V>(ctx, A, lda, x, y, M, N, alpha, beta, skip);
      else
        dispatch_t<E, false, 1>(ctx, A, lda, x, y, M, N, alpha, beta, skip);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

int32_t rls_launch_gemv(rls_ctx* ctx, int32_t dtype, int32_t op, int64_t M, int64_t N, float ar, float ai,
                        const void* A, int64_t lda, const void* x, float br, float bi, void* y, const int* skip) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || op < RLS_OP_N || op > RLS_OP_C) return rls_fail(ctx, RLS_E_INVALID, "gemv: bad dtype/op");
  if (M < 0 || N < 0 || lda < (M > 1 ? M : 1)) return rls_fail(ctx, RLS_E_INVALID, "gemv: bad shape/lda");
  if (!A || !x || !y) return rls_fail(ctx, RLS_E_INVALID, "gemv: null pointer");
  if (M == 0 || N == 0) {
    // empty contraction: y = beta * y over the output length (BLAS semantics)
    const int64_t ny = (op == RLS_OP_N) ? M : N;
    if (ny == 0) return 0;
    extern int32_t rls_launch_scale_or_zero(rls_ctx*, int32_t, int64_t, float, float, void*);
    return rls_launch_scale_or_zero(ctx, dtype, ny, br, bi, y);
  }
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F32)
    return gemv_typed<float>(ctx, op, M, N, ar, br, (const float*)A, lda, (const float*)x, (float*)y, skip);
  return gemv_typed<float2>(ctx, op, M, N, make_float2(ar, ai), make_float2(br, bi), (const float2*)A, lda,
                            (const float2*)x, (float2*)y, skip);
}
