// Setup-time kernels outside the per-iteration loop: squared row norms for
// SystemMatrixBasedNormalization, and the row-weighted copy diag(w) A of an operator.
#include "rls_common.hpp"

// partial[cs][m] = sum over the columns n = cs, cs + CS, ... of |A[m][n]|^2 ; one thread per row
template <typename E>
__global__ __launch_bounds__(256) void rownorm2_partial_kernel(const E* __restrict__ A, int64_t lda, int64_t M, int64_t N,
                                                               float* __restrict__ partial) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int cs = blockIdx.y, CS = gridDim.y;
  if (m >= M) return;
  float acc = 0.f;
  for (int64_t n = cs; n < N; n += CS) acc += elem<E>::abs2(A[n * lda + m]);
  partial[(int64_t)cs * M + m] = acc;
}

__global__ __launch_bounds__(256) void rownorm2_sum_kernel(const float* __restrict__ partial, int CS, int64_t M,
                                                           float* __restrict__ out) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  double s = 0.0;
  for (int cs = 0; cs < CS; ++cs) s += (double)partial[(int64_t)cs * M + m];  // fixed order
  out[m] = (float)s;
}

template <typename E>
__global__ __launch_bounds__(256) void scale_rows_kernel(const E* __restrict__ w, const E* __restrict__ A, int64_t lda,
                                                         E* __restrict__ B, int64_t ldb, int64_t M, int64_t N) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const E wm = w[m];
  for (int64_t n = blockIdx.y; n < N; n += gridDim.y) B[n * ldb + m] = elem<E>::mul(wm, A[n * lda + m]);
}

static int32_t setup_status(rls_ctx* ctx) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

extern "C" {

int32_t rls_rownorm2(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda, float* out_d) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || M <= 0 || N <= 0 || !A || !out_d || lda < M)
    return rls_fail(ctx, RLS_E_INVALID, "rownorm2: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  const int CS = (int)(N < 64 ? N : 64);
  float* partial = nullptr;
  RLS_HIP(ctx, hipMalloc((void**)&partial, sizeof(float) * (size_t)CS * (size_t)M));
  const dim3 grid((unsigned)((M + 255) / 256), (unsigned)CS);
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(rownorm2_partial_kernel<float>, grid, dim3(256), 0, ctx->stream, (const float*)A, lda, M, N, partial);
  else
    hipLaunchKernelGGL(rownorm2_partial_kernel<float2>, grid, dim3(256), 0, ctx->stream, (const float2*)A, lda, M, N,
                       partial);
  hipLaunchKernelGGL(rownorm2_sum_kernel, dim3(grid.x), dim3(256), 0, ctx->stream, partial, CS, M, out_d);
  int32_t st = setup_status(ctx);
  hipError_t e = rls_stream_wait(ctx->stream);  // setup path: the scratch is freed before returning
  hipFree(partial);
  if (st == 0 && e != hipSuccess) st = rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return st;
}

int32_t rls_scale_rows(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* w, const void* A, int64_t lda,
                       void* B, int64_t ldb) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || M <= 0 || N <= 0 || !w || !A || !B || lda < M || ldb < M)
    return rls_fail(ctx, RLS_E_INVALID, "scale_rows: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  const dim3 grid((unsigned)((M + 255) / 256), (unsigned)(N < 256 ? N : 256));
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(scale_rows_kernel<float>, grid, dim3(256), 0, ctx->stream, (const float*)w, (const float*)A, lda,
                       (float*)B, ldb, M, N);
  else
    hipLaunchKernelGGL(scale_rows_kernel<float2>, grid, dim3(256), 0, ctx->stream, (const float2*)w, (const float2*)A,
                       lda, (float2*)B, ldb, M, N);
  return setup_status(ctx);
}

}  // extern "C"
