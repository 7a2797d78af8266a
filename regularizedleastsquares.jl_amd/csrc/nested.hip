// Device pieces of the nested regularisation terms and of the plug-and-play input transforms:
//   MaskedRegularization        src/Regularization/MaskedRegularization.jl:27-37   gather / scatter by index list
//   AutoScaledRegularization    src/Regularization/ScaledRegularization.jl:55-77   maximum(abs.(x))
//   PlugAndPlayRegularization   src/Regularization/PlugAndPlayRegularization.jl:24-48   real / imag split + merge
//   MinMax / Z / ClampedScaling transforms   src/Transforms.jl:4-68   min, max, mean, std, affine maps, clamp
// All O(n), latency-bound; one statistics pass serves every transform (min, max, sum, sum of squares, max |x|).
#include "rls_common.hpp"

namespace {

constexpr int NS_THREADS = 256;
constexpr int NSTAT = 5;

static inline unsigned ns_grid(int64_t n) {
  int64_t g = (n + NS_THREADS - 1) / NS_THREADS;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (unsigned)g;
}

#define NS_STRIDE(i, n) \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

template <typename E>
__global__ void gather_kernel(const int32_t* __restrict__ idx, int64_t m, const E* __restrict__ x, E* __restrict__ out) {
  NS_STRIDE(k, m) out[k] = x[idx[k]];
}
template <typename E>
__global__ void scatter_kernel(const int32_t* __restrict__ idx, int64_t m, const E* __restrict__ in, E* __restrict__ x) {
  NS_STRIDE(k, m) x[idx[k]] = in[k];
}

__device__ static inline double wave_min(double v) {
  for (int o = 1; o < 64; o <<= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ static inline double wave_max(double v) {
  for (int o = 1; o < 64; o <<= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// block reductions of (min, max, sum, sumsq, absmax); result valid in thread 0
struct stats5 {
  double mn, mx, s, ss, am;
};
__device__ static inline stats5 block_stats(stats5 v, double* sm /* 16 * 5 */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v.mn = wave_min(v.mn);
  v.mx = wave_max(v.mx);
  v.s = wave_sum(v.s);
  v.ss = wave_sum(v.ss);
  v.am = wave_max(v.am);
  __syncthreads();
  if (lane == 0) {
    sm[w] = v.mn;
    sm[16 + w] = v.mx;
    sm[32 + w] = v.s;
    sm[48 + w] = v.ss;
    sm[64 + w] = v.am;
  }
  __syncthreads();
  stats5 r = {sm[0], sm[16], 0.0, 0.0, sm[64]};
  for (int i = 0; i < nw; ++i) {  // fixed order: deterministic sums
    r.mn = fmin(r.mn, sm[i]);
    r.mx = fmax(r.mx, sm[16 + i]);
    r.s += sm[32 + i];
    r.ss += sm[48 + i];
    r.am = fmax(r.am, sm[64 + i]);
  }
  return r;
}

// min / max / sum / sum of squares run over the REAL parts (the transforms act on real arrays); max |x| is the
// modulus for complex input (AutoScaledRegularization's maximum(abs.(x)))
template <typename E>
__global__ __launch_bounds__(1024) void stats_kernel(const E* __restrict__ x, int64_t n, double* __restrict__ partial) {
  __shared__ double sm[80];
  stats5 v = {INFINITY, -INFINITY, 0.0, 0.0, 0.0};
  NS_STRIDE(i, n) {
    const E e = x[i];
    const double r = (double)elem<E>::re(e);
    v.mn = fmin(v.mn, r);
    v.mx = fmax(v.mx, r);
    v.s += r;
    v.ss += r * r;
    v.am = fmax(v.am, (double)elem<E>::absv(e));
  }
  v = block_stats(v, sm);
  if (threadIdx.x == 0) {
    double* p = partial + (int64_t)NSTAT * blockIdx.x;
    p[0] = v.mn;
    p[1] = v.mx;
    p[2] = v.s;
    p[3] = v.ss;
    p[4] = v.am;
  }
}
__global__ __launch_bounds__(256) void stats_final_kernel(const double* __restrict__ partial, int nwg, double* out) {
  __shared__ double sm[80];
  stats5 v = {INFINITY, -INFINITY, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < nwg; i += blockDim.x) {
    const double* p = partial + (int64_t)NSTAT * i;
    v.mn = fmin(v.mn, p[0]);
    v.mx = fmax(v.mx, p[1]);
    v.s += p[2];
    v.ss += p[3];
    v.am = fmax(v.am, p[4]);
  }
  v = block_stats(v, sm);
  if (threadIdx.x == 0) {
    out[0] = v.mn;
    out[1] = v.mx;
    out[2] = v.s;
    out[3] = v.ss;
    out[4] = v.am;
  }
}

// mode 0: x = (x - shift) / scale        transform(::MinMaxTransform / ::ZTransform)          Transforms.jl:11,41
// mode 1: x = x * scale + shift          inverse_transform                                    Transforms.jl:15,45
__global__ void shift_scale_kernel(float* x, int64_t n, float shift, float scale, int mode) {
  NS_STRIDE(i, n) x[i] = mode == 0 ? f32_sub(x[i], shift) / scale : f32_add(f32_mul(x[i], scale), shift);  // never fused
}
__global__ void clamp_kernel(float* x, int64_t n, float lo, float hi) {
  NS_STRIDE(i, n) x[i] = fminf(fmaxf(x[i], lo), hi);
}
// out[mask] = orig[mask], mask = (orig < lo) | (orig >= hi)     inverse_transform(::ClampedScalingTransform) :62-66
__global__ void restore_outside_kernel(float* out, const float* __restrict__ orig, int64_t n, float lo, float hi) {
  NS_STRIDE(i, n) {
    const float o = orig[i];
    if (o < lo || o >= hi) out[i] = o;
  }
}
__global__ void split_kernel(const float2* __restrict__ z, int64_t n, float* __restrict__ re, float* __restrict__ im) {
  NS_STRIDE(i, n) {
    const float2 v = z[i];
    re[i] = v.x;
    im[i] = v.y;
  }
}
__global__ void merge_kernel(const float* __restrict__ re, const float* __restrict__ im, int64_t n, float2* __restrict__ z) {
  NS_STRIDE(i, n) z[i] = make_float2(re[i], im[i]);
}

static int32_t ns_status(rls_ctx* ctx) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

}  // namespace

extern "C" {

int32_t rls_gather(rls_ctx* ctx, int32_t dtype, int64_t m, const int32_t* idx, const void* x, void* out) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || m < 0 || (m > 0 && (!idx || !x || !out))) return rls_fail(ctx, RLS_E_INVALID, "gather: bad argument");
  if (m == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(gather_kernel<float>, dim3(ns_grid(m)), dim3(NS_THREADS), 0, ctx->stream, idx, m, (const float*)x, (float*)out);
  else
    hipLaunchKernelGGL(gather_kernel<float2>, dim3(ns_grid(m)), dim3(NS_THREADS), 0, ctx->stream, idx, m, (const float2*)x, (float2*)out);
  return ns_status(ctx);
}

int32_t rls_scatter(rls_ctx* ctx, int32_t dtype, int64_t m, const int32_t* idx, const void* in, void* x) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || m < 0 || (m > 0 && (!idx || !x || !in))) return rls_fail(ctx, RLS_E_INVALID, "scatter: bad argument");
  if (m == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(scatter_kernel<float>, dim3(ns_grid(m)), dim3(NS_THREADS), 0, ctx->stream, idx, m, (const float*)in, (float*)x);
  else
    hipLaunchKernelGGL(scatter_kernel<float2>, dim3(ns_grid(m)), dim3(NS_THREADS), 0, ctx->stream, idx, m, (const float2*)in, (float2*)x);
  return ns_status(ctx);
}

int32_t rls_stats(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, double* out_h) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || n <= 0 || !x || !out_h) return rls_fail(ctx, RLS_E_INVALID, "stats: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  int nwg = (int)((n + 8191) / 8192);
  const int cap = RLS_RED_SLOTS / NSTAT - 1;
  if (nwg > cap) nwg = cap;
  double* fin = ctx->red_d + (size_t)NSTAT * cap;  // last record of the scratch block
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(stats_kernel<float>, dim3(nwg), dim3(1024), 0, ctx->stream, (const float*)x, n, ctx->red_d);
  else
    hipLaunchKernelGGL(stats_kernel<float2>, dim3(nwg), dim3(1024), 0, ctx->stream, (const float2*)x, n, ctx->red_d);
  hipLaunchKernelGGL(stats_final_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->red_d, nwg, fin);
  RLS_TRY(ns_status(ctx));
  RLS_HIP(ctx, hipMemcpyAsync(out_h, fin, sizeof(double) * NSTAT, hipMemcpyDeviceToHost, ctx->stream));
  RLS_HIP(ctx, rls_stream_wait(ctx->stream));
  return 0;
}

int32_t rls_shift_scale(rls_ctx* ctx, int64_t n, float* x, float shift, float scale, int32_t inverse) {
  RLS_CHECK_CTX(ctx);
  if (n < 0 || (n > 0 && !x)) return rls_fail(ctx, RLS_E_INVALID, "shift_scale: bad argument");
  if (n == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  hipLaunchKernelGGL(shift_scale_kernel, dim3(ns_grid(n)), dim3(NS_THREADS), 0, ctx->stream, x, n, shift, scale, inverse ? 1 : 0);
  return ns_status(ctx);
}

int32_t rls_clamp(rls_ctx* ctx, int64_t n, float* x, float lo, float hi) {
  RLS_CHECK_CTX(ctx);
  if (n < 0 || (n > 0 && !x)) return rls_fail(ctx, RLS_E_INVALID, "clamp: bad argument");
  if (n == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  hipLaunchKernelGGL(clamp_kernel, dim3(ns_grid(n)), dim3(NS_THREADS), 0, ctx->stream, x, n, lo, hi);
  return ns_status(ctx);
}

int32_t rls_restore_outside(rls_ctx* ctx, int64_t n, float* out, const float* orig, float lo, float hi) {
  RLS_CHECK_CTX(ctx);
  if (n < 0 || (n > 0 && (!out || !orig))) return rls_fail(ctx, RLS_E_INVALID, "restore_outside: bad argument");
  if (n == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  hipLaunchKernelGGL(restore_outside_kernel, dim3(ns_grid(n)), dim3(NS_THREADS), 0, ctx->stream, out, orig, n, lo, hi);
  return ns_status(ctx);
}

int32_t rls_complex_split(rls_ctx* ctx, int64_t n, const void* z, float* re, float* im) {
  RLS_CHECK_CTX(ctx);
  if (n < 0 || (n > 0 && (!z || !re || !im))) return rls_fail(ctx, RLS_E_INVALID, "complex_split: bad argument");
  if (n == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  hipLaunchKernelGGL(split_kernel, dim3(ns_grid(n)), dim3(NS_THREADS), 0, ctx->stream, (const float2*)z, n, re, im);
  return ns_status(ctx);
}

int32_t rls_complex_merge(rls_ctx* ctx, int64_t n, const float* re, const float* im, void* z) {
  RLS_CHECK_CTX(ctx);
  if (n < 0 || (n > 0 && (!z || !re || !im))) return rls_fail(ctx, RLS_E_INVALID, "complex_merge: bad argument");
  if (n == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  hipLaunchKernelGGL(merge_kernel, dim3(ns_grid(n)), dim3(NS_THREADS), 0, ctx->stream, re, im, n, (float2*)z);
  return ns_status(ctx);
}

}  // extern "C"
