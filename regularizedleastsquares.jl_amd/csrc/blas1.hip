// BLAS-1 kernels: the reference's norm / dot / rmul! / fused broadcasts on length-N state vectors
// (src/CGNR.jl:125,153-174,182  src/FISTA.jl:118,147-156,172  src/ADMM.jl:236-309).
// Bytes are negligible next to the GEMVs; what matters is launch count and deterministic sums, so
// reductions accumulate in double, use a fixed shuffle tree and a fixed-order combine of partials.
#include "rls_common.hpp"

namespace {

constexpr int EW_THREADS = 256;

static inline unsigned ew_grid(int64_t n) {
  int64_t g = (n + EW_THREADS - 1) / EW_THREADS;
  if (g > 2048) g = 2048;  // grid-stride beyond 8 workgroups per CU
  if (g < 1) g = 1;
  return (unsigned)g;
}

template <typename E>
__global__ void fill_kernel(E* x, int64_t n, E v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = v;
}

template <typename E>
__global__ void scal_kernel(E* x, int64_t n, E a) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = elem<E>::mul(a, x[i]);
}

// z = a x + b y ; HAS_Y=false drops the b*y term (never reads y)
template <typename E, bool HAS_Y>
__global__ void lincomb_kernel(E* z, const E* x, const E* y, int64_t n, E a, E b) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    E out = elem<E>::mul(a, x[i]);
    if constexpr (HAS_Y) out = elem<E>::fma(b, y[i], out);
    z[i] = out;
  }
}

// y += a x  (b == 1 fast path keeps y exact where a*x == 0)
template <typename E>
__global__ void axpy_kernel(E* y, const E* x, int64_t n, E a) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = elem<E>::fma(a, x[i], y[i]);
}

enum { RED_NRM2 = 0, RED_DOTC = 1, RED_ASUM = 2 };

// stage 1: per-workgroup partial (double re, double im) ; a single-workgroup launch finalises directly
template <typename E, int OP>
__global__ __launch_bounds__(1024) void reduce_kernel(const E* __restrict__ x, const E* __restrict__ y, int64_t n,
                                                      double* __restrict__ partial, float* __restrict__ out) {
  __shared__ double sm[16];
  double re = 0.0, im = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if constexpr (OP == RED_NRM2) {
      E v = x[i];
      re += (double)elem<E>::re(v) * (double)elem<E>::re(v) + (double)elem<E>::im(v) * (double)elem<E>::im(v);
    } else if constexpr (OP == RED_ASUM) {
      re += (double)elem<E>::absv(x[i]);
    } else {
      E a = x[i], b = y[i];  // conj(a) * b
      re += (double)elem<E>::re(a) * (double)elem<E>::re(b) + (double)elem<E>::im(a) * (double)elem<E>::im(b);
      if constexpr (elem<E>::cplx)
        im += (double)elem<E>::re(a) * (double)elem<E>::im(b) - (double)elem<E>::im(a) * (double)elem<E>::re(b);
    }
  }
  re = block_sum(re, sm);
  if constexpr (OP == RED_DOTC && elem<E>::cplx) im = block_sum(im, sm);
  if (threadIdx.x == 0) {
    if (gridDim.x == 1) {
      out[0] = (float)(OP == RED_NRM2 ? sqrt(re) : re);
      out[1] = (float)im;
    } else {
      partial[2 * blockIdx.x] = re;
      partial[2 * blockIdx.x + 1] = im;
    }
  }
}

template <int OP>
__global__ __launch_bounds__(256) void reduce_final_kernel(const double* __restrict__ partial, int nwg,
                                                           float* __restrict__ out) {
  __shared__ double sm[16];
  double re = 0.0, im = 0.0;
  for (int i = threadIdx.x; i < nwg; i += blockDim.x) {
    re += partial[2 * i];
    im += partial[2 * i + 1];
  }
  re = block_sum(re, sm);
  im = block_sum(im, sm);
  if (threadIdx.x == 0) {
    out[0] = (float)(OP == RED_NRM2 ? sqrt(re) : re);
    out[1] = (float)im;
  }
}

template <typename E, int OP>
int32_t reduce_launch(rls_ctx* ctx, int64_t n, const E* x, const E* y, float* out_d) {
  int64_t per = 1024 * 8;
  int nwg = (int)((n + per - 1) / per);
  if (nwg < 1) nwg = 1;
  if (nwg > RLS_RED_SLOTS / 2) nwg = RLS_RED_SLOTS / 2;
  hipLaunchKernelGGL((reduce_kernel<E, OP>), dim3(nwg), dim3(1024), 0, ctx->stream, x, y, n, ctx->red_d, out_d);
  if (nwg > 1)
    hipLaunchKernelGGL((reduce_final_kernel<OP>), dim3(1), dim3(256), 0, ctx->stream, ctx->red_d, nwg, out_d);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

template <int OP>
int32_t reduce_dispatch(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, const void* y, float* out_d) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || n < 0 || (n > 0 && (!x || (OP == RED_DOTC && !y))) || !out_d)
    return rls_fail(ctx, RLS_E_INVALID, "reduction: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F32) return reduce_launch<float, OP>(ctx, n, (const float*)x, (const float*)y, out_d);
  return reduce_launch<float2, OP>(ctx, n, (const float2*)x, (const float2*)y, out_d);
}

int32_t fetch_result(rls_ctx* ctx, float* result_h, int nfloats) {
  RLS_HIP(ctx, hipMemcpyAsync(ctx->res_h, ctx->res_d, sizeof(float) * 2, hipMemcpyDeviceToHost, ctx->stream));
  RLS_HIP(ctx, rls_stream_wait(ctx->stream));
  for (int i = 0; i < nfloats; ++i) result_h[i] = ctx->res_h[i];
  return 0;
}

#define EW_PRELUDE(name)                                                                     \
  RLS_CHECK_CTX(ctx);                                                                        \
  if (!rls_dtype_ok(dtype) || n < 0) return rls_fail(ctx, RLS_E_INVALID, name ": bad argument"); \
  if (n == 0) return 0;                                                                      \
  RLS_HIP(ctx, rls_enter(ctx));

static int32_t ew_status(rls_ctx* ctx) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

}  // namespace

// y = beta * y, with beta == 0 writing exact zeros (used by gemv for empty contractions)
int32_t rls_launch_scale_or_zero(rls_ctx* ctx, int32_t dtype, int64_t n, float br, float bi, void* y) {
  if (br == 0.f && bi == 0.f) return rls_fill(ctx, dtype, n, y, 0.f, 0.f);
  return rls_scal(ctx, dtype, n, br, bi, y);
}

extern "C" {

int32_t rls_fill(rls_ctx* ctx, int32_t dtype, int64_t n, void* x, float re, float im) {
  EW_PRELUDE("fill");
  if (!x) return rls_fail(ctx, RLS_E_INVALID, "fill: null pointer");
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(fill_kernel<float>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, ctx->stream, (float*)x, n, re);
  else
    hipLaunchKernelGGL(fill_kernel<float2>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, ctx->stream, (float2*)x, n,
                       make_float2(re, im));
  return ew_status(ctx);
}

int32_t rls_scal(rls_ctx* ctx, int32_t dtype, int64_t n, float a_re, float a_im, void* x) {
  EW_PRELUDE("scal");
  if (!x) return rls_fail(ctx, RLS_E_INVALID, "scal: null pointer");
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(scal_kernel<float>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, ctx->stream, (float*)x, n, a_re);
  else
    hipLaunchKernelGGL(scal_kernel<float2>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, ctx->stream, (float2*)x, n,
                       make_float2(a_re, a_im));
  return ew_status(ctx);
}

int32_t rls_axpy(rls_ctx* ctx, int32_t dtype, int64_t n, float a_re, float a_im, const void* x, void* y) {
  EW_PRELUDE("axpy");
  if (!x || !y) return rls_fail(ctx, RLS_E_INVALID, "axpy: null pointer");
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(axpy_kernel<float>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, ctx->stream, (float*)y,
                       (const float*)x, n, a_re);
  else
    hipLaunchKernelGGL(axpy_kernel<float2>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, ctx->stream, (float2*)y,
                       (const float2*)x, n, make_float2(a_re, a_im));
  return ew_status(ctx);
}

int32_t rls_lincomb(rls_ctx* ctx, int32_t dtype, int64_t n, float a_re, float a_im, const void* x, float b_re,
                    float b_im, const void* y, void* z) {
  EW_PRELUDE("lincomb");
  if (!x || !z) return rls_fail(ctx, RLS_E_INVALID, "lincomb: null pointer");
  const bool has_y = (b_re != 0.f || b_im != 0.f);
  if (has_y && !y) return rls_fail(ctx, RLS_E_INVALID, "lincomb: null y");
  const dim3 g(ew_grid(n)), b(EW_THREADS);
  if (dtype == RLS_F32) {
    if (has_y)
      hipLaunchKernelGGL((lincomb_kernel<float, true>), g, b, 0, ctx->stream, (float*)z, (const float*)x,
                         (const float*)y, n, a_re, b_re);
    else
      hipLaunchKernelGGL((lincomb_kernel<float, false>), g, b, 0, ctx->stream, (float*)z, (const float*)x,
                         (const float*)y, n, a_re, b_re);
  } else {
    const float2 a = make_float2(a_re, a_im), bb = make_float2(b_re, b_im);
    if (has_y)
      hipLaunchKernelGGL((lincomb_kernel<float2, true>), g, b, 0, ctx->stream, (float2*)z, (const float2*)x,
                         (const float2*)y, n, a, bb);
    else
      hipLaunchKernelGGL((lincomb_kernel<float2, false>), g, b, 0, ctx->stream, (float2*)z, (const float2*)x,
                         (const float2*)y, n, a, bb);
  }
  return ew_status(ctx);
}

int32_t rls_axpby(rls_ctx* ctx, int32_t dtype, int64_t n, float a_re, float a_im, const void* x, float b_re,
                  float b_im, void* y) {
  return rls_lincomb(ctx, dtype, n, a_re, a_im, x, b_re, b_im, y, y);
}

int32_t rls_nrm2_dev(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, float* result_d) {
  return reduce_dispatch<RED_NRM2>(ctx, dtype, n, x, nullptr, result_d);
}
int32_t rls_dotc_dev(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, const void* y, float* result_d) {
  return reduce_dispatch<RED_DOTC>(ctx, dtype, n, x, y, result_d);
}
int32_t rls_nrm2(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, float* result_h) {
  RLS_CHECK_CTX(ctx);
  if (!result_h) return rls_fail(ctx, RLS_E_INVALID, "nrm2: null result");
  RLS_TRY(reduce_dispatch<RED_NRM2>(ctx, dtype, n, x, nullptr, ctx->res_d));
  return fetch_result(ctx, result_h, 1);
}
int32_t rls_asum(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, float* result_h) {
  RLS_CHECK_CTX(ctx);
  if (!result_h) return rls_fail(ctx, RLS_E_INVALID, "asum: null result");
  RLS_TRY(reduce_dispatch<RED_ASUM>(ctx, dtype, n, x, nullptr, ctx->res_d));
  return fetch_result(ctx, result_h, 1);
}
int32_t rls_dotc(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, const void* y, float* result_h) {
  RLS_CHECK_CTX(ctx);
  if (!result_h) return rls_fail(ctx, RLS_E_INVALID, "dotc: null result");
  RLS_TRY(reduce_dispatch<RED_DOTC>(ctx, dtype, n, x, y, ctx->res_d));
  return fetch_result(ctx, result_h, 2);
}

}  // extern "C"
