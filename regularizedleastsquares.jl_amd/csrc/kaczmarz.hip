// Kaczmarz row-action sweeps: src/Kaczmarz.jl:283-308 (iterate / iterate_row_index),
// dot_with_matrix_row (src/Utils.jl:55-88) and kaczmarz_update! (src/Kaczmarz.jl:435-517; the GPU
// extension's version, ext/RegularizedLeastSquaresGPUArraysExt/Kaczmarz.jl:1-31, is two broadcast
// launches plus two scalar read-backs PER ROW).
//
// The method is sequential over rows: row k+1 needs the x that row k produced.  One workgroup therefore
// owns one right-hand side for the whole sweep: x lives in its registers, the rows of A arrive through
// a software pipeline D rows deep (A is kept as transpose(A), row k = one contiguous 16-byte-aligned
// column, the "structure for row access" of src/Kaczmarz.jl:391), and the only synchronisation per row
// is ONE workgroup barrier for the dot product.  Independent right-hand sides (the columns of a matrix
// solve, src/MultiThreading.jl:30-79) run as independent workgroups of the same launch -- that is where
// the other 255 CUs come from; a single sweep is latency-bound by construction.
#include "rls_common.hpp"

// wave-wide sum with DPP only (no ds_bpermute): row reductions, then row_bcast:15 / row_bcast:31; the total
// ends up in lane 63
__device__ static inline float wave_sum_to_last(float v) {
  v += dpp_f(v, 0xB1);
  v += dpp_f(v, 0x4E);
  v += dpp_f(v, 0x141);
  v += dpp_f(v, 0x140);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
  return v;
}

template <typename E, bool VEC>
struct kz_vec {
  static constexpr int NV = VEC ? elem<E>::vec : 1;
};

// rows[i], den[i]: the rows of this sweep in processing order (rowindex[usedIndices[i]]) and their
// 1 / (rownorm² + lambda).  total = nused * n_sweeps row steps.
template <typename E, bool VEC, int CPT, int NT, int D, bool FULL>
__global__ __launch_bounds__(NT) void kaczmarz_sweep_kernel(const E* __restrict__ At, int64_t ldat, E* __restrict__ X,
                                                            int64_t ldx, const E* __restrict__ U, int64_t ldu,
                                                            E* VL, int64_t ldvl, const int32_t* __restrict__ rows,
                                                            const float* __restrict__ den, int nused, int n_sweeps,
                                                            float eps_w, int64_t N) {
  constexpr int NV = kz_vec<E, VEC>::NV;
  constexpr int NW = NT / 64;
  __shared__ float red[2][NW][2];
  __shared__ float scal[2][5];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.x;
  E* x = X + (int64_t)b * ldx;
  const E* u = U + (int64_t)b * ldu;
  E* vl = VL + (int64_t)b * ldvl;

  int base[CPT];
  bool valid[CPT];
  chunk<E, NV> xv[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int64_t e0 = ((int64_t)tid + (int64_t)c * NT) * NV;
    valid[c] = e0 < N;
    base[c] = valid[c] ? (int)e0 : 0;  // clamped address, never a predicated load
    xv[c] = load_chunk<E, NV>(x + base[c]);
    if (!valid[c]) xv[c] = zero_chunk<E, NV>();
  }

  const int64_t total = (int64_t)nused * n_sweeps;
  // pipeline: slot s holds row step j (j % D == s); rnext[s] / knext[s] = row id and position of step j + D
  chunk<E, NV> a[D][CPT];
  E uu[D], vv[D];
  float dd[D];
  int rcur[D], rnext[D], knext[D];
  int kq = 0;  // position of the next row id to request (step index mod nused)
#pragma unroll
  for (int s = 0; s < D; ++s) {
    rcur[s] = rows[kq];
    dd[s] = den[kq];
    kq = kq + 1 == nused ? 0 : kq + 1;
  }
#pragma unroll
  for (int s = 0; s < D; ++s) {
    knext[s] = kq;
    rnext[s] = rows[kq];
    kq = kq + 1 == nused ? 0 : kq + 1;
  }
#pragma unroll
  for (int s = 0; s < D; ++s) {
    const E* ar = At + (int64_t)rcur[s] * ldat;
#pragma unroll
    for (int c = 0; c < CPT; ++c) a[s][c] = load_chunk<E, NV>(ar + base[c]);
    uu[s] = u[rcur[s]];
    vv[s] = vl[rcur[s]];
  }

  // full trips of D row steps carry no branch around a load: the waits stay counted (vmcnt(N)) and the
  // rows of the next D steps remain in flight; the guarded tail handles total % D
  const int64_t total_full = total / D * D;
  for (int64_t j0 = 0; j0 < total_full; j0 += D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const int64_t j = j0 + s;
      // tau = sum_n A[row, n] x[n]   (dotu: no conjugation)                       src/Kaczmarz.jl:304
      float pr = 0.f, pi = 0.f, qr = 0.f, qi = 0.f;  // two chains per part: shorter dependent FMA chains
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const E av = a[s][c].e[i], xe = xv[c].e[i];
          if ((c * NV + i) & 1) {
            qr = fmaf(elem<E>::re(av), elem<E>::re(xe), qr);
            if constexpr (elem<E>::cplx) {
              qr = fmaf(-elem<E>::im(av), elem<E>::im(xe), qr);
              qi = fmaf(elem<E>::re(av), elem<E>::im(xe), qi);
              qi = fmaf(elem<E>::im(av), elem<E>::re(xe), qi);
            }
          } else {
            pr = fmaf(elem<E>::re(av), elem<E>::re(xe), pr);
            if constexpr (elem<E>::cplx) {
              pr = fmaf(-elem<E>::im(av), elem<E>::im(xe), pr);
              pi = fmaf(elem<E>::re(av), elem<E>::im(xe), pi);
              pi = fmaf(elem<E>::im(av), elem<E>::re(xe), pi);
            }
          }
        }
      }
      pr += qr;
      pi += qi;
      pr = wave_sum_to_last(pr);
      if constexpr (elem<E>::cplx) pi = wave_sum_to_last(pi);
      const int par = (int)(j & 1);
      if (lane == 63) {
        red[par][w][0] = pr;
        red[par][w][1] = pi;
      }
      if (tid == 0) {  // only thread 0's copies of u, vl are used: vl is written by thread 0 alone
        scal[par][0] = elem<E>::re(uu[s]);
        scal[par][1] = elem<E>::im(uu[s]);
        scal[par][2] = elem<E>::re(vv[s]);
        scal[par][3] = elem<E>::im(vv[s]);
        scal[par][4] = dd[s];
      }
      __syncthreads();  // the one barrier of a row step (slots alternate, so no second one is needed)
      float tr = 0.f, ti = 0.f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) {
        tr += red[par][ww][0];
        ti += red[par][ww][1];
      }
      // alpha = denom * (u[row] - tau - eps_w * vl[row])                           src/Kaczmarz.jl:305
      const float dn = scal[par][4];
      const float are = dn * ((scal[par][0] - tr) - eps_w * scal[par][2]);
      const float aim = dn * ((scal[par][1] - ti) - eps_w * scal[par][3]);
      const E alpha = elem<E>::make(are, aim);
      // x += alpha * conj(A[row, :])                                                src/Kaczmarz.jl:306
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
#pragma unroll
        for (int i = 0; i < NV; ++i) xv[c].e[i] = elem<E>::fmac(a[s][c].e[i], alpha, xv[c].e[i]);
        if constexpr (!FULL) {
          if (!valid[c]) xv[c] = zero_chunk<E, NV>();
        }
      }
      // vl[row] += alpha * eps_w                                                    src/Kaczmarz.jl:307
      if (tid == 0) vl[rcur[s]] = elem<E>::make(scal[par][2] + are * eps_w, scal[par][3] + aim * eps_w);
      // refill the slot with row step j + D, request the row id of step j + 2 D
      rcur[s] = rnext[s];
      dd[s] = den[knext[s]];
      const E* ar = At + (int64_t)rcur[s] * ldat;
#pragma unroll
      for (int c = 0; c < CPT; ++c) a[s][c] = load_chunk<E, NV>(ar + base[c]);
      uu[s] = u[rcur[s]];
      vv[s] = vl[rcur[s]];
      knext[s] = kq;
      rnext[s] = rows[kq];
      kq = kq + 1 == nused ? 0 : kq + 1;
    }
  }
  {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const int64_t j = total_full + s;
      if (j < total) {  // uniform
        // tau = sum_n A[row, n] x[n]   (dotu: no conjugation)                       src/Kaczmarz.jl:304
        float pr = 0.f, pi = 0.f, qr = 0.f, qi = 0.f;  // two chains per part: shorter dependent FMA chains
  #pragma unroll
        for (int c = 0; c < CPT; ++c) {
  #pragma unroll
          for (int i = 0; i < NV; ++i) {
            const E av = a[s][c].e[i], xe = xv[c].e[i];
            if ((c * NV + i) & 1) {
              qr = fmaf(elem<E>::re(av), elem<E>::re(xe), qr);
              if constexpr (elem<E>::cplx) {
                qr = fmaf(-elem<E>::im(av), elem<E>::im(xe), qr);
                qi = fmaf(elem<E>::re(av), elem<E>::im(xe), qi);
                qi = fmaf(elem<E>::im(av), elem<E>::re(xe), qi);
              }
            } else {
              pr = fmaf(elem<E>::re(av), elem<E>::re(xe), pr);
              if constexpr (elem<E>::cplx) {
                pr = fmaf(-elem<E>::im(av), elem<E>::im(xe), pr);
                pi = fmaf(elem<E>::re(av), elem<E>::im(xe), pi);
                pi = fmaf(elem<E>::im(av), elem<E>::re(xe), pi);
              }
            }
          }
        }
        pr += qr;
        pi += qi;
        pr = wave_sum_to_last(pr);
        if constexpr (elem<E>::cplx) pi = wave_sum_to_last(pi);
        const int par = (int)(j & 1);
        if (lane == 63) {
          red[par][w][0] = pr;
          red[par][w][1] = pi;
        }
        if (tid == 0) {  // only thread 0's copies of u, vl are used: vl is written by thread 0 alone
          scal[par][0] = elem<E>::re(uu[s]);
          scal[par][1] = elem<E>::im(uu[s]);
          scal[par][2] = elem<E>::re(vv[s]);
          scal[par][3] = elem<E>::im(vv[s]);
          scal[par][4] = dd[s];
        }
        __syncthreads();  // the one barrier of a row step (slots alternate, so no second one is needed)
        float tr = 0.f, ti = 0.f;
  #pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
          tr += red[par][ww][0];
          ti += red[par][ww][1];
        }
        // alpha = denom * (u[row] - tau - eps_w * vl[row])                           src/Kaczmarz.jl:305
        const float dn = scal[par][4];
        const float are = dn * ((scal[par][0] - tr) - eps_w * scal[par][2]);
        const float aim = dn * ((scal[par][1] - ti) - eps_w * scal[par][3]);
        const E alpha = elem<E>::make(are, aim);
        // x += alpha * conj(A[row, :])                                                src/Kaczmarz.jl:306
  #pragma unroll
        for (int c = 0; c < CPT; ++c) {
  #pragma unroll
          for (int i = 0; i < NV; ++i) xv[c].e[i] = elem<E>::fmac(a[s][c].e[i], alpha, xv[c].e[i]);
          if constexpr (!FULL) {
            if (!valid[c]) xv[c] = zero_chunk<E, NV>();
          }
        }
        // vl[row] += alpha * eps_w                                                    src/Kaczmarz.jl:307
        if (tid == 0) vl[rcur[s]] = elem<E>::make(scal[par][2] + are * eps_w, scal[par][3] + aim * eps_w);
        // refill the slot with row step j + D, request the row id of step j + 2 D
        rcur[s] = rnext[s];
        dd[s] = den[knext[s]];
        const E* ar = At + (int64_t)rcur[s] * ldat;
  #pragma unroll
        for (int c = 0; c < CPT; ++c) a[s][c] = load_chunk<E, NV>(ar + base[c]);
        uu[s] = u[rcur[s]];
        vv[s] = vl[rcur[s]];
        knext[s] = kq;
        rnext[s] = rows[kq];
        kq = kq + 1 == nused ? 0 : kq + 1;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    if (valid[c]) {
      if constexpr (NV * sizeof(E) == 16) {
        *reinterpret_cast<f4*>(x + base[c]) = __builtin_bit_cast(f4, xv[c]);
      } else {
        x[base[c]] = xv[c].e[0];
      }
    }
  }
}

// At = transpose(A) (no conjugation): 32 x 32 tiles through LDS
template <typename E>
__global__ __launch_bounds__(256) void transpose_kernel(const E* __restrict__ A, int64_t lda, E* __restrict__ At,
                                                        int64_t ldat, int64_t M, int64_t N) {
  __shared__ E tile[32][33];
  const int64_t m0 = (int64_t)blockIdx.x * 32, n0 = (int64_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int64_t m = m0 + tx, n = n0 + r;
    if (m < M && n < N) tile[r][tx] = A[n * lda + m];
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t n = n0 + tx, m = m0 + r;
    if (m < M && n < N) At[m * ldat + n] = tile[tx][r];
  }
}


static int32_t kz_status(rls_ctx* ctx) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

template <typename E, bool VEC>
static int32_t kz_launch(rls_ctx* ctx, int64_t N, const E* At, int64_t ldat, int nrhs, E* X, int64_t ldx, const E* U,
                         int64_t ldu, E* VL, int64_t ldvl, const int32_t* rows, const float* den, int nused, int n_sweeps,
                         float eps_w) {
  constexpr int NV = kz_vec<E, VEC>::NV;
  const int64_t chunks = (N + NV - 1) / NV;
#define KZ(CPT, NT, DD)                                                                                              \
  do {                                                                                                               \
    if (chunks == (int64_t)(CPT) * (NT))                                                                             \
      hipLaunchKernelGGL((kaczmarz_sweep_kernel<E, VEC, CPT, NT, DD, true>), dim3((unsigned)nrhs), dim3(NT), 0,      \
                         ctx->stream, At, ldat, X, ldx, U, ldu, VL, ldvl, rows, den, nused, n_sweeps, eps_w, N);      \
    else                                                                                                             \
      hipLaunchKernelGGL((kaczmarz_sweep_kernel<E, VEC, CPT, NT, DD, false>), dim3((unsigned)nrhs), dim3(NT), 0,     \
                         ctx->stream, At, ldat, X, ldx, U, ldu, VL, ldvl, rows, den, nused, n_sweeps, eps_w, N);      \
  } while (0)
  // measured at 4096 x 2048 ComplexF32 (tools/bench_kaczmarz.py): 512 threads x 2 chunks 0.453 us per row
  // step, 256 x 4 0.499, 1024 x 1 0.616, 256 x 4 with an 8-deep pipeline 0.493
  if (ctx->tune.kaczmarz_nt == 256 && chunks <= 1024) {
    KZ(4, 256, 4);
  } else if (ctx->tune.kaczmarz_nt == 1024 && chunks <= 1024) {
    KZ(1, 1024, 4);
  } else if (chunks <= 256) {
    KZ(1, 256, 4);
  } else if (chunks <= 1024) {
    KZ(2, 512, 4);
  } else if (chunks <= 4096) {
    KZ(4, 1024, 4);
  } else if (chunks <= 8192) {
    KZ(8, 1024, 1);
  } else {
    return rls_fail(ctx, RLS_E_UNSUPPORTED, "kaczmarz_sweep: N too large for the register-resident sweep");
  }
#undef KZ
  return kz_status(ctx);
}

extern "C" {

int32_t rls_transpose(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda, void* At,
                      int64_t ldat) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || M <= 0 || N <= 0 || !A || !At || lda < M || ldat < N)
    return rls_fail(ctx, RLS_E_INVALID, "transpose: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  const dim3 grid((unsigned)((M + 31) / 32), (unsigned)((N + 31) / 32));
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, ctx->stream, (const float*)A, lda, (float*)At, ldat,
                       M, N);
  else
    hipLaunchKernelGGL(transpose_kernel<float2>, grid, dim3(256), 0, ctx->stream, (const float2*)A, lda, (float2*)At,
                       ldat, M, N);
  return kz_status(ctx);
}

int32_t rls_kaczmarz_sweep(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* At, int64_t ldat,
                           int32_t nrhs, void* X, int64_t ldx, const void* U, int64_t ldu, void* VL, int64_t ldvl,
                           const int32_t* rows_d, const float* denom_d, int32_t nused, float eps_w, int32_t n_sweeps) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || M <= 0 || N <= 0 || !At || !X || !U || !VL || nrhs < 1 || ldat < N || ldx < N ||
      ldu < M || ldvl < M || nused < 0 || n_sweeps < 0 || (nused > 0 && (!rows_d || !denom_d)))
    return rls_fail(ctx, RLS_E_INVALID, "kaczmarz_sweep: bad argument");
  if (nused == 0 || n_sweeps == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  const int V = dtype == RLS_C32 ? 2 : 4;
  const bool vec = (N % V == 0) && (ldat % V == 0) && (ldx % V == 0) && ((uintptr_t)At % 16 == 0) &&
                   ((uintptr_t)X % 16 == 0);
  // several sweeps in one launch re-use the row order: the pipeline may prefetch vl[row] of the next
  // sweep before this sweep's update of the same row unless the sweep is longer than the pipeline
  const int per_launch = nused > 16 ? n_sweeps : 1;
  for (int done = 0; done < n_sweeps; done += per_launch) {
    int32_t st;
    if (dtype == RLS_F32)
      st = vec ? kz_launch<float, true>(ctx, N, (const float*)At, ldat, nrhs, (float*)X, ldx, (const float*)U, ldu,
                                        (float*)VL, ldvl, rows_d, denom_d, nused, per_launch, eps_w)
               : kz_launch<float, false>(ctx, N, (const float*)At, ldat, nrhs, (float*)X, ldx, (const float*)U, ldu,
                                         (float*)VL, ldvl, rows_d, denom_d, nused, per_launch, eps_w);
    else
      st = vec ? kz_launch<float2, true>(ctx, N, (const float2*)At, ldat, nrhs, (float2*)X, ldx, (const float2*)U, ldu,
                                         (float2*)VL, ldvl, rows_d, denom_d, nused, per_launch, eps_w)
               : kz_launch<float2, false>(ctx, N, (const float2*)At, ldat, nrhs, (float2*)X, ldx, (const float2*)U,
                                          ldu, (float2*)VL, ldvl, rows_d, denom_d, nused, per_launch, eps_w);
    if (st != 0) return st;
  }
  return 0;
}

}  // extern "C"
