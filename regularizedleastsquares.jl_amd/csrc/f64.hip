// Float64 / ComplexF64 element types: the L1 protocol of SURVEY 8b -- the methods the UNCHANGED solvers call on the vector and
// operator types (mul!, dot, norm, rmul!, the fused broadcasts, prox!) -- with double scalars and double results (rls_*_d).
// The reference's own suites run every solver in Float32 AND Float64 (test/testSolvers.jl:242) and its prox tests in ComplexF64
// (test/testProxMaps.jl:47,78,106); SURVEY 8a / north_star scope the tuned path (fused plans, resident kernels, matrix cores) to
// Float32 / ComplexF32, so these are plain coalesced kernels with Float64 fixed-order reductions: a double-precision caller gets the
// reference's arithmetic and its own loops (src/CGNR.jl:143-178 etc. on the primitives), not the fused fast paths.
// Entry points: rls_fill_d, rls_scal_d, rls_axpy_d, rls_lincomb_d, rls_nrm2_d, rls_dotc_d, rls_asum_d, rls_gemv_d,
// rls_prox_l1_d / _l2_d / _l21_d / _positive_d / _real_d, rls_prox_tv_fgp_d; rls_transpose_d, rls_rownorm2_d, rls_scale_rows_d,
// rls_kaczmarz_sweep_d (the row-action solver's setup and sweep).
#include "rls_common.hpp"

namespace {

template <typename D>
struct del;
template <>
struct del<double> {
  static constexpr bool cplx = false;
  __device__ static inline double zero() { return 0.0; }
  __device__ static inline double make(double re, double) { return re; }
  __device__ static inline double re(double a) { return a; }
  __device__ static inline double im(double) { return 0.0; }
  __device__ static inline double mul(double a, double b) { return a * b; }
  __device__ static inline double mulc(double a, double b) { return a * b; }
  __device__ static inline double add(double a, double b) { return a + b; }
  __device__ static inline double sub(double a, double b) { return a - b; }
  __device__ static inline double scale(double s, double a) { return s * a; }
  __device__ static inline double abs2(double a) { return a * a; }
  __device__ static inline double absv(double a) { return fabs(a); }
};
template <>
struct del<double2> {
  static constexpr bool cplx = true;
  __device__ static inline double2 zero() { return make_double2(0.0, 0.0); }
  __device__ static inline double2 make(double re, double im) { return make_double2(re, im); }
  __device__ static inline double re(double2 a) { return a.x; }
  __device__ static inline double im(double2 a) { return a.y; }
  __device__ static inline double2 mul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
  __device__ static inline double2 mulc(double2 a, double2 b) {  // conj(a) * b
    return make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
  }
  __device__ static inline double2 add(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
  __device__ static inline double2 sub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
  __device__ static inline double2 scale(double s, double2 a) { return make_double2(s * a.x, s * a.y); }
  __device__ static inline double abs2(double2 a) { return a.x * a.x + a.y * a.y; }
  __device__ static inline double absv(double2 a) { return hypot(a.x, a.y); }
};

constexpr int DT = 256;
static inline unsigned dgrid(int64_t n) {
  int64_t g = (n + DT - 1) / DT;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (unsigned)g;
}
#define DSTRIDE(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

template <typename D>
__global__ void d_fill_kernel(D* x, int64_t n, D v) {
  DSTRIDE(i, n) x[i] = v;
}
template <typename D>
__global__ void d_scal_kernel(D* x, int64_t n, D a) {
  DSTRIDE(i, n) x[i] = del<D>::mul(a, x[i]);
}
template <typename D>
__global__ void d_axpy_kernel(D* y, const D* x, int64_t n, D a) {
  DSTRIDE(i, n) y[i] = del<D>::add(del<D>::mul(a, x[i]), y[i]);
}
template <typename D, bool HAS_Y>
__global__ void d_lincomb_kernel(D* z, const D* x, const D* y, int64_t n, D a, D b) {
  DSTRIDE(i, n) {
    D out = del<D>::mul(a, x[i]);
    if constexpr (HAS_Y) out = del<D>::add(del<D>::mul(b, y[i]), out);
    z[i] = out;
  }
}

enum { DRED_NRM2 = 0, DRED_DOTC = 1, DRED_ASUM = 2 };
// stage 1: per-workgroup partial (re, im); a single-workgroup launch finalises directly.  Fixed order: grid-stride inside a thread,
// wave_sum tree, waves in order, workgroups in order.
template <typename D, int OP>
__global__ __launch_bounds__(1024) void d_reduce_kernel(const D* __restrict__ x, const D* __restrict__ y, int64_t n,
                                                        double* __restrict__ partial, double* __restrict__ out) {
  __shared__ double sm[16];
  double re = 0.0, im = 0.0;
  DSTRIDE(i, n) {
    if constexpr (OP == DRED_NRM2) {
      re += del<D>::abs2(x[i]);
    } else if constexpr (OP == DRED_ASUM) {
      re += del<D>::absv(x[i]);
    } else {
      const D p = del<D>::mulc(x[i], y[i]);
      re += del<D>::re(p);
      im += del<D>::im(p);
    }
  }
  re = block_sum(re, sm);
  if constexpr (OP == DRED_DOTC && del<D>::cplx) im = block_sum(im, sm);
  if (threadIdx.x == 0) {
    if (gridDim.x == 1) {
      out[0] = OP == DRED_NRM2 ? sqrt(re) : re;
      out[1] = im;
    } else {
      partial[2 * blockIdx.x] = re;
      partial[2 * blockIdx.x + 1] = im;
    }
  }
}
template <int OP>
__global__ __launch_bounds__(256) void d_reduce_final_kernel(const double* __restrict__ partial, int nwg, double* __restrict__ out) {
  __shared__ double sm[16];
  double re = 0.0, im = 0.0;
  for (int i = threadIdx.x; i < nwg; i += blockDim.x) {
    re += partial[2 * i];
    im += partial[2 * i + 1];
  }
  re = block_sum(re, sm);
  im = block_sum(im, sm);
  if (threadIdx.x == 0) {
    out[0] = OP == DRED_NRM2 ? sqrt(re) : re;
    out[1] = im;
  }
}

// y = alpha A x + beta y, A column-major: 64 rows per workgroup, the 4 waves take a quarter of the columns each (coalesced along the
// rows, x[j] a wave-uniform load), combined through LDS in wave order
template <typename D>
__global__ __launch_bounds__(256) void d_gemv_n_kernel(const D* __restrict__ A, int64_t lda, const D* __restrict__ x, D* y, int64_t M,
                                                       int64_t N, D alpha, D beta, int beta_zero) {
  __shared__ D part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 64 + lane;
  const int64_t j0 = w * N / 4, j1 = (w + 1) * N / 4;
  D acc = del<D>::zero();
  if (row < M)
    for (int64_t j = j0; j < j1; ++j) acc = del<D>::add(acc, del<D>::mul(A[row + j * lda], x[j]));
  part[w][lane] = acc;
  __syncthreads();
  if (w == 0 && row < M) {
    D s = part[0][lane];
    for (int ww = 1; ww < 4; ++ww) s = del<D>::add(s, part[ww][lane]);
    D out = del<D>::mul(alpha, s);
    if (!beta_zero) out = del<D>::add(out, del<D>::mul(beta, y[row]));
    y[row] = out;
  }
}
// x = alpha op(A)^T-ish y + beta x for op in {T, C}: one wave per column (contiguous), wave_sum in Float64
template <typename D, bool CONJ>
__global__ __launch_bounds__(256) void d_gemv_t_kernel(const D* __restrict__ A, int64_t lda, const D* __restrict__ yv, D* xo, int64_t M,
                                                       int64_t N, D alpha, D beta, int beta_zero) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 4 + w;
  if (col >= N) return;
  const D* a = A + col * lda;
  double re = 0.0, im = 0.0;
  for (int64_t i = lane; i < M; i += 64) {
    const D p = CONJ ? del<D>::mulc(a[i], yv[i]) : del<D>::mul(a[i], yv[i]);
    re += del<D>::re(p);
    im += del<D>::im(p);
  }
  re = wave_sum(re);
  if constexpr (del<D>::cplx) im = wave_sum(im);
  if (lane == 0) {
    D out = del<D>::mul(alpha, del<D>::make(re, im));
    if (!beta_zero) out = del<D>::add(out, del<D>::mul(beta, xo[col]));
    xo[col] = out;
  }
}

// ---- prox maps (src/proximalMaps/*.jl), eps = eps(Float64) ----
template <typename D>
__global__ void d_prox_l1_kernel(D* x, int64_t n, double lam) {
  const double eps = 2.220446049250313e-16;
  DSTRIDE(i, n) {  // max(|x| - lam, 0) * (x + eps) / (|x| + eps), eps on the real part   ProxL1.jl:18-22
    const D v = x[i];
    const double a = del<D>::absv(v), sh = fmax(a - lam, 0.0), den = a + eps;
    x[i] = del<D>::make((sh * (del<D>::re(v) + eps)) / den, (sh * del<D>::im(v)) / den);
  }
}
template <typename D>
__global__ void d_prox_l2_kernel(D* x, int64_t n, double factor) {
  DSTRIDE(i, n) x[i] = del<D>::scale(factor, x[i]);  // x / (1 + 2 lam)   ProxL2.jl:18-21
}
template <typename D, bool POS>
__global__ void d_project_kernel(D* x, int64_t n) {
  DSTRIDE(i, n) {
    double re = del<D>::re(x[i]);
    if (POS && re < 0.0) re = 0.0;
    x[i] = del<D>::make(re, 0.0);
  }
}
template <typename D>
__global__ void d_l21_kernel(D* x, int64_t n, int64_t slen, double lam) {  // ProxL21.jl:30-35, one thread per group
  DSTRIDE(i, slen) {
    double s2 = 0.0;
    for (int64_t k = i; k < n; k += slen) s2 += del<D>::abs2(x[k]);
    const double g = sqrt(s2), q = (g - lam) / g;
    const double fac = (q != q) ? q : fmax(q, 0.0);
    for (int64_t k = i; k < n; k += slen) x[k] = del<D>::scale(fac, x[k]);
  }
}

// ---- TV prox, fast gradient projection (src/proximalMaps/ProxTV.jl:89-125; GradientOp: g = x[i] - x[i + e_d], no boundary row) ----
struct dtv_geom {
  int ndims, ntv;
  int64_t shape[4], stride[4];
  int dims[4];
  int64_t goff[5];  // offsets of the gradient blocks
  int64_t n;
};
// index of the gradient component of block k at pixel multi-index idx (valid when idx[d] < shape[d] - 1)
__device__ static inline int64_t dtv_gidx(const dtv_geom& G, int k, const int64_t (&idx)[4]) {
  const int d = G.dims[k];
  int64_t g = 0, st = 1;
  for (int q = 0; q < G.ndims; ++q) {
    g += idx[q] * st;
    st *= (q == d) ? G.shape[q] - 1 : G.shape[q];
  }
  return G.goff[k] + g;
}
// out[i] = x[i] - lam * (grad^T g)[i]
template <typename D>
__global__ void d_tv_xupdate_kernel(D* out, const D* x, const D* g, dtv_geom G, double lam) {
  DSTRIDE(i, G.n) {
    int64_t idx[4] = {0, 0, 0, 0}, rem = i;
    for (int q = 0; q < G.ndims; ++q) {
      idx[q] = rem % G.shape[q];
      rem /= G.shape[q];
    }
    D s = del<D>::zero();
    for (int k = 0; k < G.ntv; ++k) {
      const int d = G.dims[k];
      if (idx[d] < G.shape[d] - 1) s = del<D>::add(s, g[dtv_gidx(G, k, idx)]);
      if (idx[d] > 0) {
        idx[d] -= 1;
        s = del<D>::sub(s, g[dtv_gidx(G, k, idx)]);
        idx[d] += 1;
      }
    }
    out[i] = del<D>::sub(x[i], del<D>::scale(lam, s));
  }
}
// pq = clip(step * grad(xTmp) + rs) written over rs; rs_new = t3 pq - t2 pqOld written into `rs_out`   (:109-123)
template <typename D>
__global__ void d_tv_dual_kernel(D* rs_pq, const D* xtmp, const D* pq_old, D* rs_out, dtv_geom G, double step, double t2, double t3) {
  const int64_t ng = G.goff[G.ntv];
  DSTRIDE(gi, ng) {
    int k = 0;
    while (k + 1 < G.ntv && gi >= G.goff[k + 1]) ++k;
    const int d = G.dims[k];
    int64_t rem = gi - G.goff[k], i = 0;
    for (int q = 0; q < G.ndims; ++q) {
      const int64_t ext = (q == d) ? G.shape[q] - 1 : G.shape[q];
      i += (rem % ext) * G.stride[q];
      rem /= ext;
    }
    const D gr = del<D>::sub(xtmp[i], xtmp[i + G.stride[d]]);
    D v = del<D>::add(del<D>::scale(step, gr), rs_pq[gi]);
    const double a = del<D>::absv(v);
    v = del<D>::scale(1.0 / fmax(1.0, a), v);                // tv_restrictMagnitude!   :135-139
    rs_pq[gi] = v;
    rs_out[gi] = del<D>::sub(del<D>::scale(t3, v), del<D>::scale(t2, pq_old[gi]));  // tv_linearcomb!   :141-145
  }
}

// ---- Kaczmarz in double precision (src/Kaczmarz.jl:283-308): one workgroup per right-hand side walks the rows in order; every thread
// owns the x entries tid, tid + 1024, ... (read and written by it alone), the rows come from transpose(A) (contiguous), and the only
// synchronisation of a row step is one workgroup barrier for tau = dot_with_matrix_row (src/Utils.jl:55-88, no conjugation)
template <typename D>
__global__ __launch_bounds__(1024) void d_kaczmarz_kernel(const D* __restrict__ At, int64_t ldat, D* X, int64_t ldx, const D* __restrict__ U,
                                                          int64_t ldu, D* VL, int64_t ldvl, const int32_t* __restrict__ rows,
                                                          const double* __restrict__ den, int nused, int n_sweeps, double eps_w, int64_t N) {
  __shared__ double red[2][16][2];
  __shared__ double scal[2][4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  D* x = X + (int64_t)blockIdx.x * ldx;
  const D* u = U + (int64_t)blockIdx.x * ldu;
  D* vl = VL + (int64_t)blockIdx.x * ldvl;
  int64_t j = 0;
  for (int sw = 0; sw < n_sweeps; ++sw) {
    for (int k = 0; k < nused; ++k, ++j) {
      const int row = rows[k];
      const D* ar = At + (int64_t)row * ldat;
      double pr = 0.0, pi = 0.0;
      for (int64_t i = tid; i < N; i += 1024) {
        const D p = del<D>::mul(ar[i], x[i]);
        pr += del<D>::re(p);
        pi += del<D>::im(p);
      }
      for (int off = 32; off > 0; off >>= 1) {
        pr += __shfl_xor(pr, off);
        pi += __shfl_xor(pi, off);
      }
      const int par = (int)(j & 1);
      if (lane == 0) {
        red[par][w][0] = pr;
        red[par][w][1] = pi;
      }
      if (tid == 0) {  // vl[row] is written by thread 0 alone: the others take u, vl from LDS, never from memory
        scal[par][0] = del<D>::re(u[row]);
        scal[par][1] = del<D>::im(u[row]);
        scal[par][2] = del<D>::re(vl[row]);
        scal[par][3] = del<D>::im(vl[row]);
      }
      __syncthreads();  // the one barrier of a row step (the slots alternate)
      double tr = 0.0, ti = 0.0;
#pragma unroll
      for (int ww = 0; ww < 16; ++ww) {
        tr += red[par][ww][0];
        ti += red[par][ww][1];
      }
      const double dn = den[k];
      const double are = dn * ((scal[par][0] - tr) - eps_w * scal[par][2]);   // alpha = denom (u[row] - tau - eps_w vl[row])   :305
      const double aim = dn * ((scal[par][1] - ti) - eps_w * scal[par][3]);
      const D alpha = del<D>::make(are, aim);
      for (int64_t i = tid; i < N; i += 1024) x[i] = del<D>::add(x[i], del<D>::mulc(ar[i], alpha));  // x += alpha conj(A[row, :])   :306
      if (tid == 0) vl[row] = del<D>::make(scal[par][2] + are * eps_w, scal[par][3] + aim * eps_w);      // vl[row] += alpha eps_w      :307
    }
  }
}

template <typename D>
__global__ __launch_bounds__(256) void d_transpose_kernel(const D* __restrict__ A, int64_t lda, D* __restrict__ At, int64_t ldat, int64_t M,
                                                          int64_t N) {
  __shared__ D tile[32][33];
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

// rownorm²(A, m) = sum_n |A[m, n]|² (src/Utils.jl:20-23): a thread per row, the columns in order
template <typename D>
__global__ void d_rownorm2_kernel(const D* __restrict__ A, int64_t lda, int64_t M, int64_t N, double* __restrict__ out) {
  DSTRIDE(m, M) {
    double s = 0.0;
    for (int64_t n = 0; n < N; ++n) s += del<D>::abs2(A[n * lda + m]);
    out[m] = s;
  }
}

// B = diag(w) A (ProdOp(WeightingOp(w), A) materialised)
template <typename D>
__global__ void d_scale_rows_kernel(const D* __restrict__ w, const D* __restrict__ A, int64_t lda, D* __restrict__ B, int64_t ldb, int64_t M,
                                    int64_t N) {
  DSTRIDE(i, M * N) {
    const int64_t m = i % M, n = i / M;
    B[n * ldb + m] = del<D>::mul(w[m], A[n * lda + m]);
  }
}

static int32_t d_status(rls_ctx* ctx) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}
static inline bool d_dtype_ok(int32_t dtype) { return dtype == RLS_F64 || dtype == RLS_C64; }
static inline size_t d_elem(int32_t dtype) { return dtype == RLS_C64 ? 16 : 8; }

#define D_PRELUDE(name)                                                                            \
  RLS_CHECK_CTX(ctx);                                                                              \
  if (!d_dtype_ok(dtype) || n < 0) return rls_fail(ctx, RLS_E_INVALID, name ": bad argument (Float64 / ComplexF64 entry point)"); \
  if (n == 0) return 0;                                                                            \
  RLS_HIP(ctx, rls_enter(ctx));

template <typename D, int OP>
static int32_t d_reduce_launch(rls_ctx* ctx, int64_t n, const D* x, const D* y, double* result_h) {
  int nwg = (int)((n + 8191) / 8192);
  if (nwg < 1) nwg = 1;
  if (nwg > RLS_RED_SLOTS / 2 - 2) nwg = RLS_RED_SLOTS / 2 - 2;
  double* out_d = ctx->red_d + (RLS_RED_SLOTS - 2);  // the last two slots of the reduction scratch: the result
  hipLaunchKernelGGL((d_reduce_kernel<D, OP>), dim3(nwg), dim3(1024), 0, ctx->stream, x, y, n, ctx->red_d, out_d);
  if (nwg > 1) hipLaunchKernelGGL((d_reduce_final_kernel<OP>), dim3(1), dim3(256), 0, ctx->stream, ctx->red_d, nwg, out_d);
  RLS_TRY(d_status(ctx));
  static_assert(RLS_RES_FLOATS >= 4, "the pinned result block holds two doubles");
  RLS_HIP(ctx, hipMemcpyAsync(ctx->res_h, out_d, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RLS_HIP(ctx, rls_stream_wait(ctx->stream));
  memcpy(result_h, ctx->res_h, 2 * sizeof(double));
  return 0;
}
template <int OP>
static int32_t d_reduce(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, const void* y, double* result_h, const char* what) {
  RLS_CHECK_CTX(ctx);
  if (!d_dtype_ok(dtype) || n < 0 || !result_h || (n > 0 && (!x || (OP == DRED_DOTC && !y)))) return rls_fail(ctx, RLS_E_INVALID, what);
  result_h[0] = result_h[1] = 0.0;
  if (n == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F64) return d_reduce_launch<double, OP>(ctx, n, (const double*)x, (const double*)y, result_h);
  return d_reduce_launch<double2, OP>(ctx, n, (const double2*)x, (const double2*)y, result_h);
}

static bool dtv_make(int32_t ndims, const int64_t* shape, int32_t ntv, const int32_t* dims, dtv_geom* G) {
  if (ndims < 1 || ndims > 4 || ntv < 0 || ntv > 4 || !shape || (ntv > 0 && !dims)) return false;
  G->ndims = ndims;
  G->ntv = ntv;
  G->n = 1;
  for (int q = 0; q < 4; ++q) {
    G->shape[q] = q < ndims ? shape[q] : 1;
    if (G->shape[q] < 1) return false;
    G->stride[q] = G->n;
    G->n *= G->shape[q];
  }
  G->goff[0] = 0;
  for (int k = 0; k < 4; ++k) {
    G->dims[k] = k < ntv ? dims[k] : 0;
    if (k < ntv && (dims[k] < 0 || dims[k] >= ndims)) return false;
    int64_t len = 0;
    if (k < ntv) {
      len = 1;
      for (int q = 0; q < ndims; ++q) len *= (q == dims[k]) ? G->shape[q] - 1 : G->shape[q];
    }
    G->goff[k + 1] = G->goff[k] + len;
  }
  return true;
}

template <typename D>
static int32_t d_fgp(rls_ctx* ctx, const dtv_geom& G, D* x, double lam, int iters) {
  const int64_t ng = G.goff[G.ntv];
  if (ng == 0 || iters == 0) return 0;  // nothing to differentiate along: prox = identity (grad^T 0 = 0)
  D* ws = nullptr;
  RLS_HIP(ctx, rls_dev_alloc(ctx, (void**)&ws, (size_t)(3 * ng + G.n) * sizeof(D)));
  RLS_HIP(ctx, hipMemsetAsync(ws, 0, (size_t)3 * ng * sizeof(D), ctx->stream));
  D *pq = ws, *rs = ws + ng, *pq_old = ws + 2 * ng, *xtmp = ws + 3 * ng;
  double t = 1.0;
  const double step = 1.0 / (8.0 * lam);
  for (int it = 0; it < iters; ++it) {
    D* pq_tmp = pq_old;  // buffer rotation of :104-108: pqOld <- pq, pq <- rs (updated in place), rs <- the oldest buffer
    pq_old = pq;
    pq = rs;
    hipLaunchKernelGGL(d_tv_xupdate_kernel<D>, dim3(dgrid(G.n)), dim3(DT), 0, ctx->stream, xtmp, (const D*)x, (const D*)rs, G, lam);
    const double t_old = t;
    t = (1.0 + sqrt(1.0 + 4.0 * t_old * t_old)) / 2.0;
    const double t2 = (t_old - 1.0) / t, t3 = 1.0 + t2;
    hipLaunchKernelGGL(d_tv_dual_kernel<D>, dim3(dgrid(ng)), dim3(DT), 0, ctx->stream, pq, (const D*)xtmp, (const D*)pq_old, pq_tmp, G, step, t2, t3);
    rs = pq_tmp;
  }
  hipLaunchKernelGGL(d_tv_xupdate_kernel<D>, dim3(dgrid(G.n)), dim3(DT), 0, ctx->stream, x, (const D*)x, (const D*)pq, G, lam);
  const int32_t st = d_status(ctx);
  RLS_HIP(ctx, rls_dev_free(ctx, ws));
  return st;
}

}  // namespace

extern "C" {

int32_t rls_fill_d(rls_ctx* ctx, int32_t dtype, int64_t n, void* x, double re, double im) {
  D_PRELUDE("fill_d");
  if (!x) return rls_fail(ctx, RLS_E_INVALID, "fill_d: null pointer");
  if (dtype == RLS_F64) hipLaunchKernelGGL(d_fill_kernel<double>, dim3(dgrid(n)), dim3(DT), 0, ctx->stream, (double*)x, n, re);
  else hipLaunchKernelGGL(d_fill_kernel<double2>, dim3(dgrid(n)), dim3(DT), 0, ctx->stream, (double2*)x, n, make_double2(re, im));
  return d_status(ctx);
}
int32_t rls_scal_d(rls_ctx* ctx, int32_t dtype, int64_t n, double a_re, double a_im, void* x) {
  D_PRELUDE("scal_d");
  if (!x) return rls_fail(ctx, RLS_E_INVALID, "scal_d: null pointer");
  if (dtype == RLS_F64) hipLaunchKernelGGL(d_scal_kernel<double>, dim3(dgrid(n)), dim3(DT), 0, ctx->stream, (double*)x, n, a_re);
  else hipLaunchKernelGGL(d_scal_kernel<double2>, dim3(dgrid(n)), dim3(DT), 0, ctx->stream, (double2*)x, n, make_double2(a_re, a_im));
  return d_status(ctx);
}
int32_t rls_axpy_d(rls_ctx* ctx, int32_t dtype, int64_t n, double a_re, double a_im, const void* x, void* y) {
  D_PRELUDE("axpy_d");
  if (!x || !y) return rls_fail(ctx, RLS_E_INVALID, "axpy_d: null pointer");
  if (dtype == RLS_F64)
    hipLaunchKernelGGL(d_axpy_kernel<double>, dim3(dgrid(n)), dim3(DT), 0, ctx->stream, (double*)y, (const double*)x, n, a_re);
  else
    hipLaunchKernelGGL(d_axpy_kernel<double2>, dim3(dgrid(n)), dim3(DT), 0, ctx->stream, (double2*)y, (const double2*)x, n,
                       make_double2(a_re, a_im));
  return d_status(ctx);
}
int32_t rls_lincomb_d(rls_ctx* ctx, int32_t dtype, int64_t n, double a_re, double a_im, const void* x, double b_re, double b_im,
                      const void* y, void* z) {
  D_PRELUDE("lincomb_d");
  if (!x || !z) return rls_fail(ctx, RLS_E_INVALID, "lincomb_d: null pointer");
  const bool has_y = b_re != 0.0 || b_im != 0.0;
  if (has_y && !y) return rls_fail(ctx, RLS_E_INVALID, "lincomb_d: null y");
  const dim3 g(dgrid(n)), b(DT);
  if (dtype == RLS_F64) {
    if (has_y) hipLaunchKernelGGL((d_lincomb_kernel<double, true>), g, b, 0, ctx->stream, (double*)z, (const double*)x, (const double*)y, n, a_re, b_re);
    else hipLaunchKernelGGL((d_lincomb_kernel<double, false>), g, b, 0, ctx->stream, (double*)z, (const double*)x, (const double*)y, n, a_re, b_re);
  } else {
    const double2 a = make_double2(a_re, a_im), bb = make_double2(b_re, b_im);
    if (has_y) hipLaunchKernelGGL((d_lincomb_kernel<double2, true>), g, b, 0, ctx->stream, (double2*)z, (const double2*)x, (const double2*)y, n, a, bb);
    else hipLaunchKernelGGL((d_lincomb_kernel<double2, false>), g, b, 0, ctx->stream, (double2*)z, (const double2*)x, (const double2*)y, n, a, bb);
  }
  return d_status(ctx);
}
int32_t rls_nrm2_d(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, double* result_h) {
  return d_reduce<DRED_NRM2>(ctx, dtype, n, x, nullptr, result_h, "nrm2_d: bad argument");
}
int32_t rls_asum_d(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, double* result_h) {
  return d_reduce<DRED_ASUM>(ctx, dtype, n, x, nullptr, result_h, "asum_d: bad argument");
}
int32_t rls_dotc_d(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, const void* y, double* result_h) {
  return d_reduce<DRED_DOTC>(ctx, dtype, n, x, y, result_h, "dotc_d: bad argument");
}

int32_t rls_gemv_d(rls_ctx* ctx, int32_t dtype, int32_t op, int64_t M, int64_t N, double alpha_re, double alpha_im, const void* A,
                   int64_t lda, const void* x, double beta_re, double beta_im, void* y) {
  RLS_CHECK_CTX(ctx);
  if (!d_dtype_ok(dtype) || M < 0 || N < 0 || lda < (M > 1 ? M : 1) || (op != RLS_OP_N && op != RLS_OP_T && op != RLS_OP_C) || !y ||
      (M > 0 && N > 0 && (!A || !x)))
    return rls_fail(ctx, RLS_E_INVALID, "gemv_d: bad argument");
  const int64_t nout = op == RLS_OP_N ? M : N, nin = op == RLS_OP_N ? N : M;
  if (nout == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  const int bz = beta_re == 0.0 && beta_im == 0.0;
  if (nin == 0) return bz ? rls_fill_d(ctx, dtype, nout, y, 0.0, 0.0) : rls_scal_d(ctx, dtype, nout, beta_re, beta_im, y);
  if (dtype == RLS_F64) {
    if (op == RLS_OP_N)
      hipLaunchKernelGGL(d_gemv_n_kernel<double>, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, ctx->stream, (const double*)A, lda,
                         (const double*)x, (double*)y, M, N, alpha_re, beta_re, bz);
    else
      hipLaunchKernelGGL((d_gemv_t_kernel<double, false>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, ctx->stream, (const double*)A, lda,
                         (const double*)x, (double*)y, M, N, alpha_re, beta_re, bz);
  } else {
    const double2 al = make_double2(alpha_re, alpha_im), be = make_double2(beta_re, beta_im);
    if (op == RLS_OP_N)
      hipLaunchKernelGGL(d_gemv_n_kernel<double2>, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, ctx->stream, (const double2*)A, lda,
                         (const double2*)x, (double2*)y, M, N, al, be, bz);
    else if (op == RLS_OP_T)
      hipLaunchKernelGGL((d_gemv_t_kernel<double2, false>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, ctx->stream, (const double2*)A,
                         lda, (const double2*)x, (double2*)y, M, N, al, be, bz);
    else
      hipLaunchKernelGGL((d_gemv_t_kernel<double2, true>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, ctx->stream, (const double2*)A,
                         lda, (const double2*)x, (double2*)y, M, N, al, be, bz);
  }
  return d_status(ctx);
}

int32_t rls_prox_l1_d(rls_ctx* ctx, int32_t dtype, int64_t n, void* x, double lambda) {
  D_PRELUDE("prox_l1_d");
  if (!x) return rls_fail(ctx, RLS_E_INVALID, "prox_l1_d: null pointer");
  if (dtype == RLS_F64) hipLaunchKernelGGL(d_prox_l1_kernel<double>, dim3(dgrid(n)), dim3(DT), 0, ctx->stream, (double*)x, n, lambda);
  else hipLaunchKernelGGL(d_prox_l1_kernel<double2>, dim3(dgrid(n)), dim3(DT), 0, ctx->stream, (double2*)x, n, lambda);
  return d_status(ctx);
}
int32_t rls_prox_l2_d(rls_ctx* ctx, int32_t dtype, int64_t n, void* x, double lambda) {
  D_PRELUDE("prox_l2_d");
  if (!x) return rls_fail(ctx, RLS_E_INVALID, "prox_l2_d: null pointer");
  const double factor = 1.0 / (1.0 + 2.0 * lambda);
  if (dtype == RLS_F64) hipLaunchKernelGGL(d_prox_l2_kernel<double>, dim3(dgrid(n)), dim3(DT), 0, ctx->stream, (double*)x, n, factor);
  else hipLaunchKernelGGL(d_prox_l2_kernel<double2>, dim3(dgrid(n)), dim3(DT), 0, ctx->stream, (double2*)x, n, factor);
  return d_status(ctx);
}
int32_t rls_prox_l21_d(rls_ctx* ctx, int32_t dtype, int64_t n, int64_t slices, void* x, double lambda) {
  D_PRELUDE("prox_l21_d");
  if (!x || slices <= 0 || n / slices == 0) return rls_fail(ctx, RLS_E_INVALID, "prox_l21_d: bad argument");
  const int64_t slen = n / slices;
  if (dtype == RLS_F64) hipLaunchKernelGGL(d_l21_kernel<double>, dim3(dgrid(slen)), dim3(DT), 0, ctx->stream, (double*)x, n, slen, lambda);
  else hipLaunchKernelGGL(d_l21_kernel<double2>, dim3(dgrid(slen)), dim3(DT), 0, ctx->stream, (double2*)x, n, slen, lambda);
  return d_status(ctx);
}
int32_t rls_prox_positive_d(rls_ctx* ctx, int32_t dtype, int64_t n, void* x) {
  D_PRELUDE("prox_positive_d");
  if (!x) return rls_fail(ctx, RLS_E_INVALID, "prox_positive_d: null pointer");
  if (dtype == RLS_F64) hipLaunchKernelGGL((d_project_kernel<double, true>), dim3(dgrid(n)), dim3(DT), 0, ctx->stream, (double*)x, n);
  else hipLaunchKernelGGL((d_project_kernel<double2, true>), dim3(dgrid(n)), dim3(DT), 0, ctx->stream, (double2*)x, n);
  return d_status(ctx);
}
int32_t rls_prox_real_d(rls_ctx* ctx, int32_t dtype, int64_t n, void* x) {
  D_PRELUDE("prox_real_d");
  if (!x) return rls_fail(ctx, RLS_E_INVALID, "prox_real_d: null pointer");
  if (dtype == RLS_F64) return 0;  // enfReal! on a real array is the identity (src/Utils.jl:114-120)
  hipLaunchKernelGGL((d_project_kernel<double2, false>), dim3(dgrid(n)), dim3(DT), 0, ctx->stream, (double2*)x, n);
  return d_status(ctx);
}
int32_t rls_prox_tv_fgp_d(rls_ctx* ctx, int32_t dtype, int32_t ndims, const int64_t* shape, int32_t ntv, const int32_t* dims, void* x,
                          double lambda, int32_t iterations) {
  RLS_CHECK_CTX(ctx);
  dtv_geom G;
  if (!d_dtype_ok(dtype) || !x || iterations < 0 || !dtv_make(ndims, shape, ntv, dims, &G))
    return rls_fail(ctx, RLS_E_INVALID, "prox_tv_fgp_d: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F64) return d_fgp<double>(ctx, G, (double*)x, lambda, iterations);
  return d_fgp<double2>(ctx, G, (double2*)x, lambda, iterations);
}

int32_t rls_transpose_d(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda, void* At, int64_t ldat) {
  RLS_CHECK_CTX(ctx);
  if (!d_dtype_ok(dtype) || M <= 0 || N <= 0 || !A || !At || lda < M || ldat < N) return rls_fail(ctx, RLS_E_INVALID, "transpose_d: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  const dim3 grid((unsigned)((M + 31) / 32), (unsigned)((N + 31) / 32));
  if (dtype == RLS_F64) hipLaunchKernelGGL(d_transpose_kernel<double>, grid, dim3(256), 0, ctx->stream, (const double*)A, lda, (double*)At, ldat, M, N);
  else hipLaunchKernelGGL(d_transpose_kernel<double2>, grid, dim3(256), 0, ctx->stream, (const double2*)A, lda, (double2*)At, ldat, M, N);
  return d_status(ctx);
}
int32_t rls_rownorm2_d(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda, double* out_d) {
  RLS_CHECK_CTX(ctx);
  if (!d_dtype_ok(dtype) || M <= 0 || N <= 0 || !A || !out_d || lda < M) return rls_fail(ctx, RLS_E_INVALID, "rownorm2_d: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F64) hipLaunchKernelGGL(d_rownorm2_kernel<double>, dim3(dgrid(M)), dim3(DT), 0, ctx->stream, (const double*)A, lda, M, N, out_d);
  else hipLaunchKernelGGL(d_rownorm2_kernel<double2>, dim3(dgrid(M)), dim3(DT), 0, ctx->stream, (const double2*)A, lda, M, N, out_d);
  return d_status(ctx);
}
int32_t rls_scale_rows_d(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* w, const void* A, int64_t lda, void* B, int64_t ldb) {
  RLS_CHECK_CTX(ctx);
  if (!d_dtype_ok(dtype) || M <= 0 || N <= 0 || !w || !A || !B || lda < M || ldb < M) return rls_fail(ctx, RLS_E_INVALID, "scale_rows_d: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F64)
    hipLaunchKernelGGL(d_scale_rows_kernel<double>, dim3(dgrid(M * N)), dim3(DT), 0, ctx->stream, (const double*)w, (const double*)A, lda, (double*)B, ldb, M, N);
  else
    hipLaunchKernelGGL(d_scale_rows_kernel<double2>, dim3(dgrid(M * N)), dim3(DT), 0, ctx->stream, (const double2*)w, (const double2*)A, lda, (double2*)B, ldb, M, N);
  return d_status(ctx);
}
int32_t rls_kaczmarz_sweep_d(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* At, int64_t ldat, int32_t nrhs, void* X,
                             int64_t ldx, const void* U, int64_t ldu, void* VL, int64_t ldvl, const int32_t* rows_d, const double* denom_d,
                             int32_t nused, double eps_w, int32_t n_sweeps) {
  RLS_CHECK_CTX(ctx);
  if (!d_dtype_ok(dtype) || M <= 0 || N <= 0 || !At || !X || !U || !VL || nrhs < 1 || ldat < N || ldx < N || ldu < M || ldvl < M || nused < 0 ||
      n_sweeps < 0 || (nused > 0 && (!rows_d || !denom_d)))
    return rls_fail(ctx, RLS_E_INVALID, "kaczmarz_sweep_d: bad argument");
  if (nused == 0 || n_sweeps == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F64)
    hipLaunchKernelGGL(d_kaczmarz_kernel<double>, dim3((unsigned)nrhs), dim3(1024), 0, ctx->stream, (const double*)At, ldat, (double*)X, ldx,
                       (const double*)U, ldu, (double*)VL, ldvl, rows_d, denom_d, nused, n_sweeps, eps_w, N);
  else
    hipLaunchKernelGGL(d_kaczmarz_kernel<double2>, dim3((unsigned)nrhs), dim3(1024), 0, ctx->stream, (const double2*)At, ldat, (double2*)X, ldx,
                       (const double2*)U, ldu, (double2*)VL, ldvl, rows_d, denom_d, nused, n_sweeps, eps_w, N);
  return d_status(ctx);
}

}  // extern "C"
