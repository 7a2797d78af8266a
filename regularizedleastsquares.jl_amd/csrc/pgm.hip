// Fused elementwise halves of the OptISTA and POGM iterations: everything of iterate that follows
// res = AHA x (src/OptISTA.jl:176-204, src/POGM.jl:176-233) in ONE launch instead of 8-12 BLAS-1 style launches.
// The momentum coefficients depend on the iteration index only and are computed by the host in Float32 exactly as
// the reference does; the kernels return the data-dependent scalars (||res||, and for POGM's gradient restart the
// real parts of <w,x>, <w,z>, <w,res>) through the context's result block.
#include "rls_common.hpp"

constexpr int PGM_THREADS = 1024;

// Deferred mode (rls_*_update_async): the iteration count, ||res|| and the reference's stopping test
// `rel_res_norm < relTol` (src/OptISTA.jl:206-209, src/POGM.jl:234-237) live in a 4-word device record, every
// launch of the sequence is a no-op once `done` is set, and the host reads the record once per solve.
// (struct pgm_state: rls_common.hpp)
__device__ static inline void pgm_state_step(pgm_state* st, float res_norm, float norm_x0, float rel_tol) {
  if (!st) return;
  st->iteration += 1;
  st->res_norm = res_norm;
  st->done = ((double)res_norm / (double)norm_x0) < (double)rel_tol;  // the host forms this quotient in double
}

template <typename E>
__device__ static inline double redot(E a, E b) {  // real(conj(a) * b)
  return (double)elem<E>::re(a) * (double)elem<E>::re(b) + (double)elem<E>::im(a) * (double)elem<E>::im(b);
}

// zold = z; z = y; res -= x0; y -= step * res; prox(y, thr); z = z / (-gamma) + x + y / gamma;
// x = -beta x + (1 + alpha + beta) z - alpha zold
template <typename E>
__global__ __launch_bounds__(PGM_THREADS) void optista_update_kernel(E* __restrict__ res, const E* __restrict__ x0,
                                                                     E* __restrict__ x, E* __restrict__ y,
                                                                     E* __restrict__ z, E* __restrict__ zold, int64_t n,
                                                                     float step, int reg_kind, float thr, float c_z,
                                                                     float c_y, float c_x, float c_zn, float c_zo,
                                                                     float* __restrict__ out, pgm_state* state,
                                                                     float norm_x0, float rel_tol) {
  __shared__ double sm[16];
  if (state && state->done) return;
  double rn = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += PGM_THREADS) {
    const E zo = z[i], ztmp = y[i], xi = x[i];
    const E r = elem<E>::sub(res[i], x0[i]);
    res[i] = r;
    rn += redot<E>(r, r);
    E yn = elem<E>::add(ztmp, elem<E>::scale(-step, r));
    yn = fista_prox_elem<E>(yn, reg_kind, thr);
    E zn = elem<E>::add(elem<E>::scale(c_z, ztmp), xi);
    zn = elem<E>::add(zn, elem<E>::scale(c_y, yn));
    E xn = elem<E>::add(elem<E>::scale(c_x, xi), elem<E>::scale(c_zn, zn));
    xn = elem<E>::add(xn, elem<E>::scale(c_zo, zo));
    zold[i] = zo;
    y[i] = yn;
    z[i] = zn;
    x[i] = xn;
  }
  rn = block_sum(rn, sm);
  if (threadIdx.x == 0) {
    out[0] = (float)sqrt(rn);
    pgm_state_step(state, out[0], norm_x0, rel_tol);
  }
}

// xbuf holds x_k, ybuf holds y_{k-1} on entry; on exit xbuf holds the gradient point (the new y after the
// reference's swap, src/POGM.jl:203) and ybuf the new x, so the caller swaps its two references.
// Returns (valid in every thread) ||res||^2 and, with RESTART, real <w,x>, <w,z>, <w,res>.
template <typename E, bool RESTART>
__device__ static inline void pogm_update_body(E* __restrict__ res, const E* __restrict__ x0, E* __restrict__ xbuf,
                                               E* __restrict__ ybuf, E* __restrict__ xold, E* __restrict__ z,
                                               E* __restrict__ w, int64_t n, float rho, float c_y, float c_x1,
                                               float c_xo, float c_z, int reg_kind, float thr, int proj_kind, float rg,
                                               double* sm /* 48 */, double& rn, double& dwx, double& dwz, double& dwr) {
  rn = dwx = dwz = dwr = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += PGM_THREADS) {
    const E xo = xbuf[i], yp = ybuf[i];
    const E r = elem<E>::sub(res[i], x0[i]);              // res .-= x0                         :178
    res[i] = r;
    rn += redot<E>(r, r);
    const E x1 = elem<E>::add(xo, elem<E>::scale(-rho, r));  // x .-= rho .* res               :179
    E xn = elem<E>::add(elem<E>::scale(c_y, yp), elem<E>::scale(c_x1, x1));  // after the swap  :204-205
    xn = elem<E>::add(xn, elem<E>::scale(c_xo, xo));
    xn = elem<E>::add(xn, elem<E>::scale(c_z, z[i]));
    const E zn = xn;                                      // z .= x                              :210
    xn = fista_proj_elem<E>(fista_prox_elem<E>(xn, reg_kind, thr), proj_kind);
    xold[i] = xo;
    z[i] = zn;
    xbuf[i] = x1;
    ybuf[i] = xn;
    if constexpr (RESTART) {                              // gradient restart                    :218-232
      E wi = elem<E>::add(w[i], x1);
      wi = elem<E>::add(wi, elem<E>::scale(rg, xn));
      wi = elem<E>::add(wi, elem<E>::scale(-rg, zn));
      dwx += redot<E>(wi, xn);
      dwz += redot<E>(wi, zn);
      dwr += redot<E>(wi, r);
      E wn = elem<E>::add(elem<E>::scale(rg, zn), elem<E>::scale(-rg, xn));
      w[i] = elem<E>::sub(wn, x1);
    }
  }
  rn = block_sum(rn, sm);
  if constexpr (RESTART) block_sum3(dwx, dwz, dwr, sm);
}

template <typename E, bool RESTART>
__global__ __launch_bounds__(PGM_THREADS) void pogm_update_kernel(E* __restrict__ res, const E* __restrict__ x0,
                                                                  E* __restrict__ xbuf, E* __restrict__ ybuf,
                                                                  E* __restrict__ xold, E* __restrict__ z,
                                                                  E* __restrict__ w, int64_t n, float rho, float c_y,
                                                                  float c_x1, float c_xo, float c_z, int reg_kind,
                                                                  float thr, int proj_kind, float rg,
                                                                  float* __restrict__ out, pgm_state* state,
                                                                  float norm_x0, float rel_tol) {
  __shared__ double sm[48];
  if (state && state->done) return;
  double rn, dwx, dwz, dwr;
  pogm_update_body<E, RESTART>(res, x0, xbuf, ybuf, xold, z, w, n, rho, c_y, c_x1, c_xo, c_z, reg_kind, thr, proj_kind, rg,
                               sm, rn, dwx, dwz, dwr);
  if (threadIdx.x == 0) {
    out[0] = (float)sqrt(rn);
    out[1] = (float)dwx;
    out[2] = (float)dwz;
    out[3] = (float)dwr;
    pgm_state_step(state, out[0], norm_x0, rel_tol);
  }
}

// POGM with gradient restart, deferred: theta, sigma and gamma live in the device record and the coefficients of an
// iteration (src/POGM.jl:183-201) are formed HERE from them, in Float32 with the host's operation order (explicit
// round-to-nearest intrinsics: no contraction), so that the data-dependent restart decision (:218-232) never has to
// travel to the host.
// (struct pogm_auto_state: rls_common.hpp)
template <typename E>
__global__ __launch_bounds__(PGM_THREADS) void pogm_auto_kernel(E* __restrict__ res, const E* __restrict__ x0,
                                                                E* __restrict__ xbuf, E* __restrict__ ybuf,
                                                                E* __restrict__ xold, E* __restrict__ z,
                                                                E* __restrict__ w, int64_t n, float rho, float lam,
                                                                float sigma_fac, int max_iter, int reg_kind,
                                                                int proj_kind, pogm_auto_state* st, float norm_x0,
                                                                float rel_tol) {
  __shared__ double sm[48];
  if (st->done) return;
  float th, alpha, c_x1, gamma, c_z, c_xo, thr, rg;
  const float tho = st->theta, sigma = st->sigma, gamma_old = st->gamma;
  {  // one rounding per operation, as NumPy's Float32 scalars on the host (f32_mul / f32_add: never fused)
    const bool last = st->iteration == max_iter - 1;                     // :183-187
    const float t2 = f32_mul(f32_mul(last ? 8.f : 4.f, tho), tho);
    th = f32_add(1.f, sqrtf(f32_add(1.f, t2))) / 2.f;  // sqrtf: correctly rounded (v_sqrt_f32 alone is not)
    alpha = f32_sub(tho, 1.f) / th;                                      // :189
    const float beta = f32_mul(sigma, tho) / th;                         // :190
    c_x1 = f32_add(f32_add(1.f, alpha), beta);
    gamma = f32_mul(rho, c_x1);                                          // :195  rho (1 + alpha + beta)
    c_z = f32_mul(rho, alpha) / gamma_old;
    c_xo = -f32_add(beta, c_z);
    thr = f32_mul(gamma, lam);
    rg = rho / gamma;
  }
  double rn, dwx, dwz, dwr;
  pogm_update_body<E, true>(res, x0, xbuf, ybuf, xold, z, w, n, rho, -alpha, c_x1, c_xo, c_z, reg_kind, thr, proj_kind, rg,
                            sm, rn, dwx, dwz, dwr);
  if (threadIdx.x == 0) {
    const float crit = f32_sub(f32_sub((float)dwx, (float)dwz) / gamma, (float)dwr);   // :224
    const bool restart = crit < 0.f;
    st->theta_old = tho;
    st->theta = restart ? 1.f : th;
    st->sigma = restart ? 1.f : f32_mul(sigma, sigma_fac);
    st->gamma = gamma;
    const float rnorm = (float)sqrt(rn);
    st->res_norm = rnorm;
    st->iteration += 1;
    st->done = ((double)rnorm / (double)norm_x0) < (double)rel_tol;
  }
}

static int32_t pgm_fetch(rls_ctx* ctx, float* out_h, int nfloats) {
  RLS_HIP(ctx, hipMemcpyAsync(ctx->res_h, ctx->res_d, sizeof(float) * (size_t)nfloats, hipMemcpyDeviceToHost, ctx->stream));
  RLS_HIP(ctx, rls_stream_wait(ctx->stream));
  for (int i = 0; i < nfloats; ++i) out_h[i] = ctx->res_h[i];
  return 0;
}

extern "C" {

static int32_t optista_launch(rls_ctx* ctx, int32_t dtype, int64_t n, void* res, const void* x0, void* x, void* y, void* z,
                              void* zold, float step, int32_t reg_kind, float thr, float c_z, float c_y, float c_x,
                              float c_zn, float c_zo, pgm_state* state, float norm_x0, float rel_tol) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || n <= 0 || !res || !x0 || !x || !y || !z || !zold || reg_kind < RLS_REG_NONE ||
      reg_kind > RLS_REG_L2)
    return rls_fail(ctx, RLS_E_INVALID, "optista_update: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(optista_update_kernel<float>, dim3(1), dim3(PGM_THREADS), 0, ctx->stream, (float*)res,
                       (const float*)x0, (float*)x, (float*)y, (float*)z, (float*)zold, n, step, reg_kind, thr, c_z, c_y,
                       c_x, c_zn, c_zo, ctx->res_d, state, norm_x0, rel_tol);
  else
    hipLaunchKernelGGL(optista_update_kernel<float2>, dim3(1), dim3(PGM_THREADS), 0, ctx->stream, (float2*)res,
                       (const float2*)x0, (float2*)x, (float2*)y, (float2*)z, (float2*)zold, n, step, reg_kind, thr, c_z,
                       c_y, c_x, c_zn, c_zo, ctx->res_d, state, norm_x0, rel_tol);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

int32_t rls_optista_update(rls_ctx* ctx, int32_t dtype, int64_t n, void* res, const void* x0, void* x, void* y, void* z,
                           void* zold, float step, int32_t reg_kind, float thr, float c_z, float c_y, float c_x,
                           float c_zn, float c_zo, float* res_norm_h) {
  RLS_CHECK_CTX(ctx);
  if (!res_norm_h) return rls_fail(ctx, RLS_E_INVALID, "optista_update: null result pointer");
  RLS_TRY(optista_launch(ctx, dtype, n, res, x0, x, y, z, zold, step, reg_kind, thr, c_z, c_y, c_x, c_zn, c_zo, nullptr,
                         1.f, 0.f));
  return pgm_fetch(ctx, res_norm_h, 1);
}

int32_t rls_optista_update_async(rls_ctx* ctx, int32_t dtype, int64_t n, void* res, const void* x0, void* x, void* y,
                                 void* z, void* zold, float step, int32_t reg_kind, float thr, float c_z, float c_y,
                                 float c_x, float c_zn, float c_zo, float norm_x0, float rel_tol, void* state_d) {
  RLS_CHECK_CTX(ctx);
  if (!state_d) return rls_fail(ctx, RLS_E_INVALID, "optista_update_async: null state");
  return optista_launch(ctx, dtype, n, res, x0, x, y, z, zold, step, reg_kind, thr, c_z, c_y, c_x, c_zn, c_zo,
                        (pgm_state*)state_d, norm_x0, rel_tol);
}

static int32_t pogm_launch(rls_ctx* ctx, int32_t dtype, int64_t n, void* res, const void* x0, void* xbuf, void* ybuf,
                           void* xold, void* z, void* w, float rho, float c_y, float c_x1, float c_xo, float c_z,
                           int32_t reg_kind, float thr, int32_t proj_kind, int32_t restart, float rho_over_gamma,
                           pgm_state* state, float norm_x0, float rel_tol) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || n <= 0 || !res || !x0 || !xbuf || !ybuf || !xold || !z || (restart && !w) ||
      reg_kind < RLS_REG_NONE || reg_kind > RLS_REG_L2 || proj_kind < RLS_PROJ_NONE || proj_kind > RLS_PROJ_POSITIVE)
    return rls_fail(ctx, RLS_E_INVALID, "pogm_update: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
#define RLS_POGM(EE, RR)                                                                                             \
  hipLaunchKernelGGL((pogm_update_kernel<EE, RR>), dim3(1), dim3(PGM_THREADS), 0, ctx->stream, (EE*)res, (const EE*)x0, \
                     (EE*)xbuf, (EE*)ybuf, (EE*)xold, (EE*)z, (EE*)w, n, rho, c_y, c_x1, c_xo, c_z, reg_kind, thr,     \
                     proj_kind, rho_over_gamma, ctx->res_d, state, norm_x0, rel_tol)
  if (dtype == RLS_F32) {
    if (restart) RLS_POGM(float, true);
    else RLS_POGM(float, false);
  } else {
    if (restart) RLS_POGM(float2, true);
    else RLS_POGM(float2, false);
  }
#undef RLS_POGM
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

int32_t rls_pogm_update(rls_ctx* ctx, int32_t dtype, int64_t n, void* res, const void* x0, void* xbuf, void* ybuf,
                        void* xold, void* z, void* w, float rho, float c_y, float c_x1, float c_xo, float c_z,
                        int32_t reg_kind, float thr, int32_t proj_kind, int32_t restart, float rho_over_gamma,
                        float* out_h) {
  RLS_CHECK_CTX(ctx);
  if (!out_h) return rls_fail(ctx, RLS_E_INVALID, "pogm_update: null result pointer");
  RLS_TRY(pogm_launch(ctx, dtype, n, res, x0, xbuf, ybuf, xold, z, w, rho, c_y, c_x1, c_xo, c_z, reg_kind, thr, proj_kind,
                      restart, rho_over_gamma, nullptr, 1.f, 0.f));
  return pgm_fetch(ctx, out_h, 4);
}

// restart = :none only (the gradient restart feeds data-dependent theta / sigma back into the next coefficients)
int32_t rls_pogm_update_async(rls_ctx* ctx, int32_t dtype, int64_t n, void* res, const void* x0, void* xbuf, void* ybuf,
                              void* xold, void* z, float rho, float c_y, float c_x1, float c_xo, float c_z,
                              int32_t reg_kind, float thr, int32_t proj_kind, float norm_x0, float rel_tol,
                              void* state_d) {
  RLS_CHECK_CTX(ctx);
  if (!state_d) return rls_fail(ctx, RLS_E_INVALID, "pogm_update_async: null state");
  return pogm_launch(ctx, dtype, n, res, x0, xbuf, ybuf, xold, z, nullptr, rho, c_y, c_x1, c_xo, c_z, reg_kind, thr,
                     proj_kind, 0, 0.f, (pgm_state*)state_d, norm_x0, rel_tol);
}

// POGM, restart = :gradient, deferred.  state_d: 8 device words {int32 iteration, int32 done, float ||res||, pad,
// float theta, theta_old, sigma, gamma}; the caller writes theta, sigma, gamma (and zeroes the rest) before the
// first iteration and reads all of it back after the last.
int32_t rls_pogm_update_auto(rls_ctx* ctx, int32_t dtype, int64_t n, void* res, const void* x0, void* xbuf, void* ybuf,
                             void* xold, void* z, void* w, float rho, float lambda, float sigma_fac, int32_t iterations,
                             int32_t reg_kind, int32_t proj_kind, float norm_x0, float rel_tol, void* state_d) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || n <= 0 || !res || !x0 || !xbuf || !ybuf || !xold || !z || !w || !state_d ||
      reg_kind < RLS_REG_NONE || reg_kind > RLS_REG_L2 || proj_kind < RLS_PROJ_NONE || proj_kind > RLS_PROJ_POSITIVE)
    return rls_fail(ctx, RLS_E_INVALID, "pogm_update_auto: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(pogm_auto_kernel<float>, dim3(1), dim3(PGM_THREADS), 0, ctx->stream, (float*)res, (const float*)x0,
                       (float*)xbuf, (float*)ybuf, (float*)xold, (float*)z, (float*)w, n, rho, lambda, sigma_fac,
                       iterations, reg_kind, proj_kind, (pogm_auto_state*)state_d, norm_x0, rel_tol);
  else
    hipLaunchKernelGGL(pogm_auto_kernel<float2>, dim3(1), dim3(PGM_THREADS), 0, ctx->stream, (float2*)res,
                       (const float2*)x0, (float2*)xbuf, (float2*)ybuf, (float2*)xold, (float2*)z, (float2*)w, n, rho,
                       lambda, sigma_fac, iterations, reg_kind, proj_kind, (pogm_auto_state*)state_d, norm_x0, rel_tol);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

}  // extern "C"
