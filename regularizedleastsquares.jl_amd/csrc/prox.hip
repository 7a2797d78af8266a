// Proximal maps of the reference (src/proximalMaps/*.jl), one in-place pass each.
#include "rls_common.hpp"

namespace {

constexpr int PX_THREADS = 256;
static inline unsigned px_grid(int64_t n) {
  int64_t g = (n + PX_THREADS - 1) / PX_THREADS;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (unsigned)g;
}

#define GRID_STRIDE(i, n) \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// src/proximalMaps/ProxL1.jl:18-22:  x <- max(|x|-lam, 0) * (x + eps) / (|x| + eps), eps on the real part
template <typename E>
__device__ static inline E soft_threshold(E v, float lam) {
  const float eps = 1.1920929e-07f;  // eps(Float32)
  const float a = elem<E>::absv(v);
  const float sh = fmaxf(a - lam, 0.f);
  const float den = a + eps;
  if constexpr (elem<E>::cplx) {
    return make_float2((sh * (v.x + eps)) / den, (sh * v.y) / den);
  } else {
    return (sh * (v + eps)) / den;
  }
}

template <typename E>
__global__ void prox_l1_kernel(E* x, int64_t n, float lam) {
  GRID_STRIDE(i, n) x[i] = soft_threshold<E>(x[i], lam);
}

// src/proximalMaps/ProxL2.jl:18-21: the factor is Float64 (literal promotion), product rounded on store
template <typename E>
__global__ void prox_l2_kernel(E* x, int64_t n, double factor) {
  GRID_STRIDE(i, n) {
    E v = x[i];
    x[i] = elem<E>::make((float)((double)elem<E>::re(v) * factor), (float)((double)elem<E>::im(v) * factor));
  }
}

// src/Utils.jl:114-144 via src/proximalMaps/ProxPositive.jl:16-20 / ProxReal.jl:16-19
template <typename E, bool POS>
__global__ void project_kernel(E* x, int64_t n) {
  GRID_STRIDE(i, n) {
    E v = x[i];
    float re = elem<E>::re(v);
    if (POS && re < 0.f) re = 0.f;
    x[i] = elem<E>::make(re, 0.f);
  }
}

// src/proximalMaps/ProxL21.jl:30-35.  Group i = {x[k] : k mod slen == i} (the reference's
// x[i:sliceLength:end] runs to the END of x, so a ragged tail joins its group).  One thread per
// group: consecutive threads touch consecutive addresses for every slice, so both passes coalesce.
// (g-lam)/g on an all-zero group: -Inf -> 0 for lam>0; 0/0 = NaN for lam==0, propagated as Julia's max does.
template <typename E, bool APPLY>
__global__ void l21_kernel(E* x, int64_t n, int64_t slen, float lam, double* norm_partial) {
  __shared__ double sm[16];
  double local = 0.0;
  GRID_STRIDE(i, slen) {
    float s2 = 0.f;
    for (int64_t k = i; k < n; k += slen) s2 += elem<E>::abs2(x[k]);
    const float g = sqrtf(s2);
    if constexpr (APPLY) {
      const float q = (g - lam) / g;
      const float fac = (q != q) ? q : fmaxf(q, 0.f);
      for (int64_t k = i; k < n; k += slen) x[k] = elem<E>::scale(fac, x[k]);
    } else {
      local += (double)g;
    }
  }
  if constexpr (!APPLY) {
    local = block_sum(local, sm);
    if (threadIdx.x == 0) norm_partial[2 * blockIdx.x] = local, norm_partial[2 * blockIdx.x + 1] = 0.0;
  }
}

__global__ void l21_norm_final(const double* partial, int nwg, float lam, float* out) {
  __shared__ double sm[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < nwg; i += blockDim.x) s += partial[2 * i];
  s = block_sum(s, sm);
  if (threadIdx.x == 0) out[0] = (float)((double)lam * s), out[1] = 0.f;
}

static int32_t px_status(rls_ctx* ctx) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

#define PX_PRELUDE(name)                                                                           \
  RLS_CHECK_CTX(ctx);                                                                              \
  if (!rls_dtype_ok(dtype) || n < 0 || (n > 0 && !x)) return rls_fail(ctx, RLS_E_INVALID, name ": bad argument"); \
  if (n == 0) return 0;                                                                            \
  RLS_HIP(ctx, rls_enter(ctx));

}  // namespace

extern "C" {

int32_t rls_prox_l1(rls_ctx* ctx, int32_t dtype, int64_t n, void* x, float lambda) {
  PX_PRELUDE("prox_l1");
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(prox_l1_kernel<float>, dim3(px_grid(n)), dim3(PX_THREADS), 0, ctx->stream, (float*)x, n, lambda);
  else
    hipLaunchKernelGGL(prox_l1_kernel<float2>, dim3(px_grid(n)), dim3(PX_THREADS), 0, ctx->stream, (float2*)x, n, lambda);
  return px_status(ctx);
}

int32_t rls_prox_l2(rls_ctx* ctx, int32_t dtype, int64_t n, void* x, float lambda) {
  PX_PRELUDE("prox_l2");
  const double factor = 1.0 / (1.0 + 2.0 * (double)lambda);
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(prox_l2_kernel<float>, dim3(px_grid(n)), dim3(PX_THREADS), 0, ctx->stream, (float*)x, n, factor);
  else
    hipLaunchKernelGGL(prox_l2_kernel<float2>, dim3(px_grid(n)), dim3(PX_THREADS), 0, ctx->stream, (float2*)x, n, factor);
  return px_status(ctx);
}

int32_t rls_prox_positive(rls_ctx* ctx, int32_t dtype, int64_t n, void* x) {
  PX_PRELUDE("prox_positive");
  if (dtype == RLS_F32)
    hipLaunchKernelGGL((project_kernel<float, true>), dim3(px_grid(n)), dim3(PX_THREADS), 0, ctx->stream, (float*)x, n);
  else
    hipLaunchKernelGGL((project_kernel<float2, true>), dim3(px_grid(n)), dim3(PX_THREADS), 0, ctx->stream, (float2*)x, n);
  return px_status(ctx);
}

int32_t rls_prox_real(rls_ctx* ctx, int32_t dtype, int64_t n, void* x) {
  PX_PRELUDE("prox_real");
  if (dtype == RLS_F32) return 0;  // enfReal!(::AbstractArray{<:Real}) = nothing  (src/Utils.jl:125)
  hipLaunchKernelGGL((project_kernel<float2, false>), dim3(px_grid(n)), dim3(PX_THREADS), 0, ctx->stream, (float2*)x, n);
  return px_status(ctx);
}

int32_t rls_prox_l21(rls_ctx* ctx, int32_t dtype, int64_t n, int64_t slices, void* x, float lambda) {
  PX_PRELUDE("prox_l21");
  if (slices <= 0 || n / slices == 0) return rls_fail(ctx, RLS_E_INVALID, "prox_l21: slices must be in 1..n");
  const int64_t slen = n / slices;
  if (dtype == RLS_F32)
    hipLaunchKernelGGL((l21_kernel<float, true>), dim3(px_grid(slen)), dim3(PX_THREADS), 0, ctx->stream, (float*)x, n,
                       slen, lambda, (double*)nullptr);
  else
    hipLaunchKernelGGL((l21_kernel<float2, true>), dim3(px_grid(slen)), dim3(PX_THREADS), 0, ctx->stream, (float2*)x,
                       n, slen, lambda, (double*)nullptr);
  return px_status(ctx);
}

int32_t rls_norm_l21(rls_ctx* ctx, int32_t dtype, int64_t n, int64_t slices, const void* x, float lambda,
                     float* result_h) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || n <= 0 || !x || !result_h || slices <= 0 || n / slices == 0)
    return rls_fail(ctx, RLS_E_INVALID, "norm_l21: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  const int64_t slen = n / slices;
  unsigned g = px_grid(slen);
  if (g > RLS_RED_SLOTS / 2) g = RLS_RED_SLOTS / 2;
  if (dtype == RLS_F32)
    hipLaunchKernelGGL((l21_kernel<float, false>), dim3(g), dim3(PX_THREADS), 0, ctx->stream, (float*)x, n, slen,
                       lambda, ctx->red_d);
  else
    hipLaunchKernelGGL((l21_kernel<float2, false>), dim3(g), dim3(PX_THREADS), 0, ctx->stream, (float2*)x, n, slen,
                       lambda, ctx->red_d);
  hipLaunchKernelGGL(l21_norm_final, dim3(1), dim3(256), 0, ctx->stream, ctx->red_d, (int)g, lambda, ctx->res_d);
  RLS_TRY(px_status(ctx));
  RLS_HIP(ctx, hipMemcpyAsync(ctx->res_h, ctx->res_d, sizeof(float) * 2, hipMemcpyDeviceToHost, ctx->stream));
  RLS_HIP(ctx, rls_stream_wait(ctx->stream));
  result_h[0] = ctx->res_h[0];
  return 0;
}

}  // extern "C"
