// Total-variation proximal map: GradientOp stencils + the FGP loop of
// src/proximalMaps/ProxTV.jl:89-125 (helpers :127-145, GPU helpers ext/..GPUArraysExt/ProxTV.jl:1-17).
//
// GradientOp (LinearOperatorCollection v2, not in the reference tree; restated): for every dim d in
// `dims`, g_d[i] = x[i] - x[i + e_d] on the column-major array of extents `shape`, extent
// shape[d]-1 along d; blocks concatenated in dims order.  The transpose is written as a gather
// (+g at i, -g at i - e_d) so it needs no atomics and is bit-reproducible.
//
// FGP is launch-bound, not byte-bound (hard part 9): for images whose duals fit in one
// CU's 160 KiB LDS the whole loop is ONE single-workgroup kernel with workgroup barriers;
// larger images use two launches per FGP iteration.  Two dual buffers suffice: pq is formed in
// place over rs, and the new rs overwrites the old pq element by element.
#include "rls_common.hpp"

#include <map>
#include <mutex>

namespace {

constexpr int TV_MAXD = 4;

struct tv_geom {
  int ndims, ntv;
  int64_t shape[TV_MAXD];
  int64_t stride[TV_MAXD];           // x strides (column-major)
  int dims[TV_MAXD];                 // differenced dims, in order
  int64_t goff[TV_MAXD + 1];         // start of block k inside g; goff[ntv] = total length
  int64_t bstride[TV_MAXD][TV_MAXD]; // strides of block k (extent shape[d]-1 along its dim)
  int64_t n;
};

static bool make_geom(int32_t ndims, const int64_t* shape, int32_t ntv, const int32_t* dims, tv_geom* g) {
  if (ndims < 1 || ndims > TV_MAXD || ntv < 0 || ntv > TV_MAXD || !shape || (ntv > 0 && !dims)) return false;
  g->ndims = ndims;
  g->ntv = ntv;
  int64_t n = 1;
  for (int k = 0; k < TV_MAXD; ++k) {
    g->shape[k] = k < ndims ? shape[k] : 1;
    if (g->shape[k] < 1) return false;
    g->stride[k] = n;
    n *= g->shape[k];
  }
  g->n = n;
  int64_t off = 0;
  for (int k = 0; k < ntv; ++k) {
    const int d = dims[k];
    if (d < 0 || d >= ndims) return false;
    g->dims[k] = d;
    g->goff[k] = off;
    int64_t bs = 1;
    for (int m = 0; m < TV_MAXD; ++m) {
      g->bstride[k][m] = bs;
      bs *= (m == d) ? (g->shape[m] - 1) : g->shape[m];
    }
    off += bs;
  }
  for (int k = ntv; k <= TV_MAXD; ++k) g->goff[k] = off;
  return true;
}

// (grad x)[gi] for a global dual index gi
template <typename E>
__device__ static inline E grad_at(const E* x, const tv_geom& G, int64_t gi) {
  int k = 0;
  while (k + 1 < G.ntv && gi >= G.goff[k + 1]) ++k;
  const int d = G.dims[k];
  int64_t li = gi - G.goff[k], xi = 0;
#pragma unroll
  for (int m = 0; m < TV_MAXD; ++m) {
    const int64_t ext = (m == d) ? (G.shape[m] - 1) : G.shape[m];
    const int64_t c = li % ext;
    li /= ext;
    xi += c * G.stride[m];
  }
  return elem<E>::sub(x[xi], x[xi + G.stride[d]]);
}

// (grad^T g)[xi]
template <typename E>
__device__ static inline E gradt_at(const E* g, const tv_geom& G, int64_t xi) {
  int64_t c[TV_MAXD];
  int64_t r = xi;
#pragma unroll
  for (int m = 0; m < TV_MAXD; ++m) {
    c[m] = r % G.shape[m];
    r /= G.shape[m];
  }
  E s = elem<E>::zero();
  for (int k = 0; k < G.ntv; ++k) {
    const int d = G.dims[k];
    int64_t bi = 0;
#pragma unroll
    for (int m = 0; m < TV_MAXD; ++m) bi += c[m] * G.bstride[k][m];
    const E* gb = g + G.goff[k];
    if (c[d] < G.shape[d] - 1) s = elem<E>::add(s, gb[bi]);
    if (c[d] > 0) s = elem<E>::sub(s, gb[bi - G.bstride[k][d]]);
  }
  return s;
}

// 32-bit twin of tv_geom for the LDS-resident kernel (n, ng < 2^31 there): the stencil index math
// is what that kernel spends its time on, and 64-bit div/mod is ~4x the instructions of 32-bit
struct tv_geom32 {
  int ndims, ntv;
  unsigned shape[TV_MAXD], stride[TV_MAXD];
  int dims[TV_MAXD];
  unsigned goff[TV_MAXD + 1];
  unsigned bstride[TV_MAXD][TV_MAXD];
  unsigned n;
};
static tv_geom32 narrow_geom(const tv_geom& G) {
  tv_geom32 g;
  g.ndims = G.ndims;
  g.ntv = G.ntv;
  g.n = (unsigned)G.n;
  for (int k = 0; k < TV_MAXD; ++k) {
    g.shape[k] = (unsigned)G.shape[k];
    g.stride[k] = (unsigned)G.stride[k];
    g.dims[k] = G.dims[k];
    for (int m = 0; m < TV_MAXD; ++m) g.bstride[k][m] = (unsigned)G.bstride[k][m];
  }
  for (int k = 0; k <= TV_MAXD; ++k) g.goff[k] = (unsigned)G.goff[k];
  return g;
}

template <typename E>
__device__ static inline E grad_at32(const E* x, const tv_geom32& G, unsigned gi) {
  int k = 0;
  while (k + 1 < G.ntv && gi >= G.goff[k + 1]) ++k;
  const int d = G.dims[k];
  unsigned li = gi - G.goff[k], xi = 0;
  for (int m = 0; m < G.ndims; ++m) {
    const unsigned ext = (m == d) ? (G.shape[m] - 1) : G.shape[m];
    const unsigned q = li / ext;
    xi += (li - q * ext) * G.stride[m];
    li = q;
  }
  return elem<E>::sub(x[xi], x[xi + G.stride[d]]);
}

template <typename E>
__device__ static inline E gradt_at32(const E* g, const tv_geom32& G, unsigned xi) {
  unsigned c[TV_MAXD];
  unsigned r = xi;
  for (int m = 0; m < G.ndims; ++m) {
    const unsigned q = r / G.shape[m];
    c[m] = r - q * G.shape[m];
    r = q;
  }
  E s = elem<E>::zero();
  for (int k = 0; k < G.ntv; ++k) {
    const int d = G.dims[k];
    unsigned bi = 0;
    for (int m = 0; m < G.ndims; ++m) bi += c[m] * G.bstride[k][m];
    const E* gb = g + G.goff[k];
    if (c[d] < G.shape[d] - 1) s = elem<E>::add(s, gb[bi]);
    if (c[d] > 0) s = elem<E>::sub(s, gb[bi - G.bstride[k][d]]);
  }
  return s;
}

// tv_restrictMagnitude!: q /= max(1, |q|)   (ProxTV.jl:135-139)
template <typename E>
__device__ static inline E tv_clip(E q) {
  if constexpr (!elem<E>::cplx) {
    // real: q / max(1, |q|) IS clamp(q, -1, 1), bit for bit (|q| <= 1: q / 1; otherwise q / |q| = +-1 exactly; a NaN stays a
    // NaN) -- three instructions instead of an IEEE division (~12), twice per pixel and FGP iteration on kernels that are
    // bound by the VALU rate of the one CU they run on
    const float v = elem<E>::re(q);
    const float c = fminf(fmaxf(v, -1.f), 1.f);
    return elem<E>::make(v == v ? c : v, 0.f);
  } else {
    const float m = fmaxf(1.f, elem<E>::absv(q));
    return elem<E>::make(elem<E>::re(q) / m, elem<E>::im(q) / m);
  }
}

#define GRID_STRIDE(i, n) \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

template <typename E>
__global__ void grad_kernel(const E* x, E* g, tv_geom G, float alpha, float beta) {
  GRID_STRIDE(gi, G.goff[G.ntv]) {
    E v = elem<E>::scale(alpha, grad_at<E>(x, G, gi));
    if (beta != 0.f) v = elem<E>::add(v, elem<E>::scale(beta, g[gi]));
    g[gi] = v;
  }
}

// out = alpha * grad^T(g) + beta * xin   (out may alias xin)
template <typename E>
__global__ void gradt_kernel(const E* g, const E* xin, E* out, tv_geom G, float alpha, float beta) {
  GRID_STRIDE(xi, G.n) {
    E v = elem<E>::scale(alpha, gradt_at<E>(g, G, xi));
    if (beta != 0.f) v = elem<E>::add(v, elem<E>::scale(beta, xin[xi]));
    out[xi] = v;
  }
}

template <typename E>
__global__ void restrict_kernel(E* pq, int64_t n) {
  GRID_STRIDE(i, n) pq[i] = tv_clip<E>(pq[i]);
}

template <typename E>
__global__ void tv_lincomb_kernel(E* rs, float t3, const E* pq, float t2, const E* pqOld, int64_t n) {
  GRID_STRIDE(i, n) rs[i] = elem<E>::sub(elem<E>::scale(t3, pq[i]), elem<E>::scale(t2, pqOld[i]));
}

// FGP dual update (multi-launch path): pq = clip(rs + step * grad(xTmp)) in place over rs, then
// the new rs = t3*pq - t2*pqOld overwrites pqOld.
template <typename E>
__global__ void fgp_dual_kernel(E* brs, E* bpq, const E* xtmp, tv_geom G, float step, float t2, float t3) {
  GRID_STRIDE(gi, G.goff[G.ntv]) {
    E q = elem<E>::add(elem<E>::scale(step, grad_at<E>(xtmp, G, gi)), brs[gi]);
    q = tv_clip<E>(q);
    const E old = bpq[gi];
    brs[gi] = q;
    bpq[gi] = elem<E>::sub(elem<E>::scale(t3, q), elem<E>::scale(t2, old));
  }
}

// images processed by one launch of the single-workgroup FGP kernels: workgroup b works on the image that starts
// b * ldv elements further on and honours the skip flag skip[b * skip_stride]
struct tv_batch {
  int count = 1;
  int64_t ldv = 0;
  int skip_stride = 0;
};

// whole FGP loop in one workgroup; the image, xTmp and both dual buffers live in LDS (a global
// re-read of x in every FGP iteration cost ~6 us of dependent latency per iteration)
// out = prox_TV(xin [+ add]); `skip` (nullable) is a device flag that turns the launch into a no-op (ADMM plan)
template <typename E>
__global__ __launch_bounds__(1024) void fgp_fused_kernel(const E* xin, const E* add, E* x, tv_geom32 G, float lam,
                                                         int iters, const int* skip, tv_batch Bt) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // one workgroup per image (batched plans: image b = column b of N x K matrices, its own skip flag)
  xin += (int64_t)blockIdx.x * Bt.ldv;
  if (add) add += (int64_t)blockIdx.x * Bt.ldv;
  x += (int64_t)blockIdx.x * Bt.ldv;
  if (skip && skip[(int64_t)blockIdx.x * Bt.skip_stride]) return;
  const unsigned ng = G.goff[G.ntv], n = G.n;
  E* b0 = reinterpret_cast<E*>(smem_raw);
  E* b1 = b0 + ng;
  E* xt = b1 + ng;
  E* xl = xt + n;
  const unsigned tid = threadIdx.x, nth = blockDim.x;
  for (unsigned i = tid; i < ng; i += nth) {
    b0[i] = elem<E>::zero();
    b1[i] = elem<E>::zero();
  }
  for (unsigned i = tid; i < n; i += nth) xl[i] = add ? elem<E>::add(xin[i], add[i]) : xin[i];
  __syncthreads();
  E* brs = b0;
  E* bpq = b1;
  float t = 1.f;
  const float step = 1.f / (8.f * lam);
  for (int it = 0; it < iters; ++it) {
#pragma unroll 4
    for (unsigned i = tid; i < n; i += nth)
      xt[i] = elem<E>::add(xl[i], elem<E>::scale(-lam, gradt_at32<E>(brs, G, i)));
    __syncthreads();
    const float tOld = t;
    t = (1.f + sqrtf(1.f + 4.f * tOld * tOld)) / 2.f;
    const float t2 = (tOld - 1.f) / t, t3 = 1.f + t2;
#pragma unroll 4
    for (unsigned gi = tid; gi < ng; gi += nth) {
      E q = elem<E>::add(elem<E>::scale(step, grad_at32<E>(xt, G, gi)), brs[gi]);
      q = tv_clip<E>(q);
      const E old = bpq[gi];
      brs[gi] = q;
      bpq[gi] = elem<E>::sub(elem<E>::scale(t3, q), elem<E>::scale(t2, old));
    }
    __syncthreads();
    E* tmp = brs;
    brs = bpq;
    bpq = tmp;  // bpq now holds the newest pq
  }
#pragma unroll 4
  for (unsigned i = tid; i < n; i += nth) x[i] = elem<E>::add(xl[i], elem<E>::scale(-lam, gradt_at32<E>(bpq, G, i)));
}

// 2-D images (or 1-D signals), dims differenced in their natural order: every thread keeps its PPT pixels -- image,
// both dual buffers, both components -- in registers; LDS only carries what a NEIGHBOUR reads (xTmp, and the dual
// that grad^T is applied to), laid out on the pixel grid so that the neighbours are k+1 / k+nx and k-1 / k-nx.
// Two workgroup barriers per FGP iteration, no index arithmetic inside the loop (the masks are formed once).
// Same operation order as grad_at32 / gradt_at32 above.
template <typename E, int PPT>
__global__ __launch_bounds__(1024) void fgp2d_kernel(const E* __restrict__ xin, const E* __restrict__ add,
                                                     E* __restrict__ out, unsigned nx, unsigned ny, int use0, int use1,
                                                     float lam, int iters, const int* __restrict__ skip, tv_batch Bt) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  xin += (int64_t)blockIdx.x * Bt.ldv;  // one workgroup per image (see fgp_fused_kernel)
  if (add) add += (int64_t)blockIdx.x * Bt.ldv;
  out += (int64_t)blockIdx.x * Bt.ldv;
  if (skip && skip[(int64_t)blockIdx.x * Bt.skip_stride]) return;
  const unsigned n = nx * ny, tid = threadIdx.x, nth = blockDim.x;
  E* xt = reinterpret_cast<E*>(smem_raw);
  E* P = xt + n;
  E* Q = P + n;
  // LEAN (8 pixels per thread at 1024 threads = 128 VGPRs; the full form spilled 64 B per lane): xTmp is not kept across the
  // barrier but read back from LDS, where the neighbours read it anyway, and the four clamped neighbour indices are formed
  // from the masks where they are used (pixel index -/+ mask bit x stride) instead of being held in 4 x 8 registers
  constexpr bool LEAN = PPT >= 8;
  E xl[PPT], rp[PPT], rq[PPT], pp[PPT], pq[PPT], xv[LEAN ? 1 : PPT];
  unsigned mIn = 0, mP = 0, mPm = 0, mQ = 0, mQm = 0;
#pragma unroll
  for (int m = 0; m < PPT; ++m) {
    const unsigned k = tid + m * nth;
    xl[m] = rp[m] = rq[m] = pp[m] = pq[m] = elem<E>::zero();
    if (k < n) {
      const unsigned j = k / nx, i = k - j * nx;
      mIn |= 1u << m;
      if (use0 && i + 1 < nx) mP |= 1u << m;
      if (use0 && i > 0) mPm |= 1u << m;
      if (use1 && j + 1 < ny) mQ |= 1u << m;
      if (use1 && j > 0) mQm |= 1u << m;
      xl[m] = add ? elem<E>::add(xin[k], add[k]) : xin[k];
      P[k] = elem<E>::zero();
      Q[k] = elem<E>::zero();
    }
  }
  __syncthreads();
  float t = 1.f;
  const float step = 1.f / (8.f * lam);
  // Neighbour addresses, clamped once (a masked-out neighbour is read from the pixel itself and its value discarded): the
  // loop body then has no branch around an LDS access and the 2 PPT neighbour reads of a phase go out back to back.
  // Measured: 64 x 64 Float32, 10 iterations 20.6 -> 20.1 us -- little, because the kernel is bound by the VALU rate of
  // the ONE CU it runs on (time per FGP iteration scales with the pixels per thread: 0.36 us at one, 1.5 us at four),
  // not by LDS latency.  The arithmetic is unchanged: a masked-out term is +0 under a subtraction and -0 under an
  // addition, which leave every value -- and the sign of a zero -- as it was.
  constexpr int NI = LEAN ? 1 : PPT;
  unsigned kc[PPT], kPm_[NI], kQm_[NI], kPp_[NI], kQp_[NI];
#pragma unroll
  for (int m = 0; m < PPT; ++m) {
    const unsigned k = tid + m * nth;
    kc[m] = (mIn >> m & 1) ? k : 0u;
    if constexpr (!LEAN) {
      kPm_[m] = (mPm >> m & 1) ? k - 1 : kc[m];
      kQm_[m] = (mQm >> m & 1) ? k - nx : kc[m];
      kPp_[m] = (mP >> m & 1) ? k + 1 : kc[m];
      kQp_[m] = (mQ >> m & 1) ? k + nx : kc[m];
    }
  }
  // (a set mask bit implies the pixel is inside the image, i.e. kc[m] == k)
  auto kPm = [&](int m) { if constexpr (LEAN) return kc[m] - (mPm >> m & 1u); else return kPm_[m]; };
  auto kQm = [&](int m) { if constexpr (LEAN) return kc[m] - (mQm >> m & 1u) * nx; else return kQm_[m]; };
  auto kPp = [&](int m) { if constexpr (LEAN) return kc[m] + (mP >> m & 1u); else return kPp_[m]; };
  auto kQp = [&](int m) { if constexpr (LEAN) return kc[m] + (mQ >> m & 1u) * nx; else return kQp_[m]; };
  const E pz = elem<E>::zero(), nz = elem<E>::make(-0.f, -0.f);
  for (int it = 0; it < iters; ++it) {
    if constexpr (LEAN) {  // the neighbour indices are loop invariants: without this they are hoisted back into 32 registers
      asm volatile("" : "+v"(mPm), "+v"(mQm), "+v"(mP), "+v"(mQ));
    }
    E nP[PPT], nQ[PPT];
#pragma unroll
    for (int m = 0; m < PPT; ++m) {
      nP[m] = P[kPm(m)];
      nQ[m] = Q[kQm(m)];
    }
#pragma unroll
    for (int m = 0; m < PPT; ++m) {
      E s = elem<E>::zero();
      s = elem<E>::add(s, (mP >> m & 1) ? rp[m] : nz);
      s = elem<E>::sub(s, (mPm >> m & 1) ? nP[m] : pz);
      s = elem<E>::add(s, (mQ >> m & 1) ? rq[m] : nz);
      s = elem<E>::sub(s, (mQm >> m & 1) ? nQ[m] : pz);
      const E xm = elem<E>::add(xl[m], elem<E>::scale(-lam, s));
      if constexpr (!LEAN) xv[m] = xm;
      if (mIn >> m & 1) xt[kc[m]] = xm;
    }
    __syncthreads();
    const float tOld = t;
    t = (1.f + sqrtf(1.f + 4.f * tOld * tOld)) / 2.f;
    const float t2 = (tOld - 1.f) / t, t3 = 1.f + t2;
    E xP[PPT], xQ[PPT];
#pragma unroll
    for (int m = 0; m < PPT; ++m) {
      xP[m] = xt[kPp(m)];
      xQ[m] = xt[kQp(m)];
    }
#pragma unroll
    for (int m = 0; m < PPT; ++m) {
      E xc;  // xTmp of this pixel (a pixel outside the image reads xt[0]: its updates below are masked too)
      if constexpr (LEAN) xc = xt[kc[m]];
      else xc = xv[m];
      {
        E q = elem<E>::add(elem<E>::scale(step, elem<E>::sub(xc, xP[m])), rp[m]);
        q = tv_clip<E>(q);
        const E rn = elem<E>::sub(elem<E>::scale(t3, q), elem<E>::scale(t2, pp[m]));
        const bool on = mP >> m & 1;
        rp[m] = on ? rn : rp[m];
        pp[m] = on ? q : pp[m];
      }
      {
        E q = elem<E>::add(elem<E>::scale(step, elem<E>::sub(xc, xQ[m])), rq[m]);
        q = tv_clip<E>(q);
        const E rn = elem<E>::sub(elem<E>::scale(t3, q), elem<E>::scale(t2, pq[m]));
        const bool on = mQ >> m & 1;
        rq[m] = on ? rn : rq[m];
        pq[m] = on ? q : pq[m];
      }
      if (mIn >> m & 1) {  // masked-out duals stay zero, so the unconditional store writes what is there already
        P[kc[m]] = rp[m];
        Q[kc[m]] = rq[m];
      }
    }
    __syncthreads();
  }
  // x = x - lam * grad^T(newest pq)
#pragma unroll
  for (int m = 0; m < PPT; ++m) {
    const unsigned k = tid + m * nth;
    if (mP >> m & 1) P[k] = pp[m];
    if (mQ >> m & 1) Q[k] = pq[m];
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < PPT; ++m) {
    const unsigned k = tid + m * nth;
    if (mIn >> m & 1) {
      E s = elem<E>::zero();
      if (mP >> m & 1) s = elem<E>::add(s, pp[m]);
      if (mPm >> m & 1) s = elem<E>::sub(s, P[k - 1]);
      if (mQ >> m & 1) s = elem<E>::add(s, pq[m]);
      if (mQm >> m & 1) s = elem<E>::sub(s, Q[k - nx]);
      out[k] = elem<E>::add(xl[m], elem<E>::scale(-lam, s));
    }
  }
}

// small per-context cache of captured FGP launch sequences (multi-launch path)
struct fgp_graph_key {
  const void *x, *ws;
  float lam;
  int iters, dtype;
  tv_geom G;
};
struct fgp_graph_cache {
  struct entry {
    fgp_graph_key key;
    hipGraphExec_t exec = nullptr;
    unsigned long long stamp = 0;
  };
  entry entries[8];
  unsigned long long clock = 0;
};
static fgp_graph_cache& fgp_cache_for(rls_ctx* ctx) {
  // contexts are few and long-lived; the caches are leaked with the process (graphs hold no device memory)
  static std::map<rls_ctx*, fgp_graph_cache*> caches;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  auto it = caches.find(ctx);
  if (it == caches.end()) it = caches.emplace(ctx, new fgp_graph_cache()).first;
  return *it->second;
}

constexpr size_t FGP_LDS_BUDGET = 160 * 1024 - 512;
// (rls_tuning::tv_fused_max_n -- larger images: one CU is slower than 2 chip-wide launches per FGP iteration -- and tv_fused_2d, the
//  register-resident 2-D kernel for n <= 8192 pixels, are the context's)
constexpr int FGP2D_MAX_PPT = 8;

static int32_t tv_status(rls_ctx* ctx) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}
static inline unsigned tv_grid(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// geometry of the register-resident kernel: (nx, ny, use0, use1), or false
static bool fgp2d_geom(const rls_ctx* ctx, const tv_geom& G, size_t es, unsigned* nx, unsigned* ny, int* use0, int* use1) {
  // complex images: 4 pixels per thread is the register limit at 1024 threads
  if (!ctx->tune.tv_fused_2d || G.ndims > 2 || G.ntv < 1 || G.n > 1024 * (es > 4 ? FGP2D_MAX_PPT / 2 : FGP2D_MAX_PPT)) return false;
  if ((size_t)3 * G.n * es > FGP_LDS_BUDGET) return false;
  *use0 = *use1 = 0;
  for (int k = 0; k < G.ntv; ++k) {
    if (k > 0 && G.dims[k] <= G.dims[k - 1]) return false;  // natural order only (summation order of grad^T)
    (G.dims[k] == 0 ? *use0 : *use1) = 1;
  }
  *nx = (unsigned)G.shape[0];
  *ny = (unsigned)G.shape[1];
  return true;
}

template <typename E, int PPT>
static void fgp2d_launch(rls_ctx* ctx, unsigned nx, unsigned ny, int use0, int use1, const E* xin, const E* add, E* out,
                         float lam, int iters, const int* skip, const tv_batch& Bt) {
  const unsigned n = nx * ny;
  const size_t lds = (size_t)3 * n * sizeof(E);
  unsigned nth = (n + PPT - 1) / PPT;
  nth = (nth + 63) / 64 * 64;
  if (nth > 1024) nth = 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&fgp2d_kernel<E, PPT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)lds);
  hipLaunchKernelGGL((fgp2d_kernel<E, PPT>), dim3((unsigned)Bt.count), dim3(nth), lds, ctx->stream, xin, add, out, nx, ny,
                     use0, use1, lam, iters, skip, Bt);
}

// single-workgroup FGP (either kernel) when the geometry allows: out = prox_TV(xin [+ add]).  Returns false
// (nothing launched) otherwise.
template <typename E>
static bool fgp_single_launch(rls_ctx* ctx, const tv_geom& G, const E* xin, const E* add, E* out, float lam, int iters,
                              const int* skip, const tv_batch& Bt = tv_batch()) {
  unsigned nx, ny;
  int use0, use1;
  if (fgp2d_geom(ctx, G, sizeof(E), &nx, &ny, &use0, &use1)) {
    const unsigned n = nx * ny;
    if (n <= 1024)
      fgp2d_launch<E, 1>(ctx, nx, ny, use0, use1, xin, add, out, lam, iters, skip, Bt);
    else if (n <= 2048)
      fgp2d_launch<E, 2>(ctx, nx, ny, use0, use1, xin, add, out, lam, iters, skip, Bt);
    else if (n <= 4096)
      fgp2d_launch<E, 4>(ctx, nx, ny, use0, use1, xin, add, out, lam, iters, skip, Bt);
    else if constexpr (sizeof(E) == 4)  // (fgp2d_geom: complex images stop at 4 pixels per thread)
      fgp2d_launch<E, 8>(ctx, nx, ny, use0, use1, xin, add, out, lam, iters, skip, Bt);
    return true;
  }
  const int64_t ng = G.goff[G.ntv], n = G.n;
  const size_t lds = (size_t)(2 * ng + 2 * n) * sizeof(E);
  if (lds <= FGP_LDS_BUDGET && n <= ctx->tune.tv_fused_max_n) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&fgp_fused_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds);
    hipLaunchKernelGGL(fgp_fused_kernel<E>, dim3((unsigned)Bt.count), dim3(1024), lds, ctx->stream, xin, add, out,
                       narrow_geom(G), lam, iters, skip, Bt);
    return true;
  }
  return false;
}

template <typename E>
int32_t fgp_typed(rls_ctx* ctx, const tv_geom& G, E* x, float lam, int iters, E* ws) {
  const int64_t ng = G.goff[G.ntv], n = G.n;
  if (fgp_single_launch<E>(ctx, G, x, nullptr, x, lam, iters, nullptr)) return tv_status(ctx);
  // Multi-launch path: 2 chip-wide launches per FGP iteration.  Eager, the 2*iters + 2 launches are
  // host-bound (~3.8 us each measured); the whole sequence is therefore captured once per
  // (x, workspace, lambda, iterations, geometry) and replayed as a hipGraph (ADMM calls prox! with the
  // same few argument sets every outer iteration).  Capture failure falls back to eager launches.
  auto enqueue = [&]() -> int32_t {
    E* brs = ws;
    E* bpq = ws + ng;
    E* xt = ws + 2 * ng;
    RLS_HIP(ctx, hipMemsetAsync(ws, 0, (size_t)2 * ng * sizeof(E), ctx->stream));
    float t = 1.f;
    const float step = 1.f / (8.f * lam);
    for (int it = 0; it < iters; ++it) {
      hipLaunchKernelGGL(gradt_kernel<E>, dim3(tv_grid(n)), dim3(256), 0, ctx->stream, brs, x, xt, G, -lam, 1.f);
      const float tOld = t;
      t = (1.f + sqrtf(1.f + 4.f * tOld * tOld)) / 2.f;
      const float t2 = (tOld - 1.f) / t, t3 = 1.f + t2;
      hipLaunchKernelGGL(fgp_dual_kernel<E>, dim3(tv_grid(ng)), dim3(256), 0, ctx->stream, brs, bpq, xt, G, step, t2, t3);
      E* tmp = brs;
      brs = bpq;
      bpq = tmp;
    }
    hipLaunchKernelGGL(gradt_kernel<E>, dim3(tv_grid(n)), dim3(256), 0, ctx->stream, bpq, x, x, G, -lam, 1.f);
    return tv_status(ctx);
  };
  if (!ctx->tune.use_graph) return enqueue();
  fgp_graph_key key;
  memset(&key, 0, sizeof(key));
  key.x = x;
  key.ws = ws;
  key.lam = lam;
  key.iters = iters;
  key.dtype = (int)sizeof(E);
  key.G = G;
  fgp_graph_cache& cache = fgp_cache_for(ctx);
  for (auto& e : cache.entries) {
    if (e.exec && memcmp(&e.key, &key, sizeof(key)) == 0) {
      e.stamp = ++cache.clock;
      RLS_HIP(ctx, hipGraphLaunch(e.exec, ctx->stream));
      return 0;
    }
  }
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  std::unique_lock<std::mutex> capture_lock(rls_capture_mutex());
  hipError_t err = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed);
  int32_t st = 0;
  if (err == hipSuccess) {
    st = enqueue();
    err = hipStreamEndCapture(ctx->stream, &graph);
  }
  capture_lock.unlock();
  if (err != hipSuccess || st != 0 || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
    if (graph) hipGraphDestroy(graph);
    (void)hipGetLastError();
    return enqueue();  // eager, still the HIP kernels
  }
  hipGraphDestroy(graph);
  fgp_graph_cache::entry* victim = &cache.entries[0];
  for (auto& e : cache.entries)
    if (!e.exec || e.stamp < victim->stamp) victim = (!victim->exec ? victim : &e);
  for (auto& e : cache.entries)
    if (!e.exec) victim = &e;
  if (victim->exec) hipGraphExecDestroy(victim->exec);
  victim->key = key;
  victim->exec = exec;
  victim->stamp = ++cache.clock;
  RLS_HIP(ctx, hipGraphLaunch(exec, ctx->stream));
  return 0;
}

}  // namespace

static bool tv_single_ok(const rls_ctx* ctx, const tv_geom& G, size_t es) {
  unsigned nx, ny;
  int u0, u1;
  if (fgp2d_geom(ctx, G, es, &nx, &ny, &u0, &u1)) return true;
  return (size_t)(2 * G.goff[G.ntv] + 2 * G.n) * es <= FGP_LDS_BUDGET && G.n <= ctx->tune.tv_fused_max_n;
}

bool rls_tv_single_ok(const rls_ctx* ctx, int32_t dtype, int32_t ndims, const int64_t* shape, int32_t ntv, const int32_t* dims) {
  tv_geom G;
  return ctx && rls_dtype_ok(dtype) && make_geom(ndims, shape, ntv, dims, &G) && tv_single_ok(ctx, G, rls_elem_size(dtype));
}

int32_t rls_tv_single_launch(rls_ctx* ctx, int32_t dtype, int32_t ndims, const int64_t* shape, int32_t ntv,
                             const int32_t* dims, const void* xin, const void* add, void* out, float lam, int iters,
                             const int* skip, int count, int64_t ldv, int skip_stride) {
  tv_geom G;
  if (!rls_dtype_ok(dtype) || !make_geom(ndims, shape, ntv, dims, &G) || count < 1)
    return rls_fail(ctx, RLS_E_INVALID, "tv_single_launch: bad geometry");
  tv_batch Bt;
  Bt.count = count;
  Bt.ldv = ldv;
  Bt.skip_stride = skip_stride;
  const bool ok = dtype == RLS_F32
                      ? fgp_single_launch<float>(ctx, G, (const float*)xin, (const float*)add, (float*)out, lam, iters, skip, Bt)
                      : fgp_single_launch<float2>(ctx, G, (const float2*)xin, (const float2*)add, (float2*)out, lam,
                                                  iters, skip, Bt);
  if (!ok) return rls_fail(ctx, RLS_E_UNSUPPORTED, "tv_single_launch: image does not fit one workgroup");
  return tv_status(ctx);
}

extern "C" {

int64_t rls_tv_grad_len(int32_t ndims, const int64_t* shape, int32_t ntv, const int32_t* dims) {
  tv_geom G;
  if (!make_geom(ndims, shape, ntv, dims, &G)) return -1;
  return G.goff[G.ntv];
}

size_t rls_prox_tv_workspace_bytes(int32_t dtype, int32_t ndims, const int64_t* shape, int32_t ntv,
                                   const int32_t* dims) {
  tv_geom G;
  if (!rls_dtype_ok(dtype) || !make_geom(ndims, shape, ntv, dims, &G)) return 0;
  return (size_t)(2 * G.goff[G.ntv] + G.n) * rls_elem_size(dtype);
}

int32_t rls_tv_grad(rls_ctx* ctx, int32_t dtype, int32_t ndims, const int64_t* shape, int32_t ntv, const int32_t* dims,
                    const void* x, void* g, float alpha, float beta) {
  RLS_CHECK_CTX(ctx);
  tv_geom G;
  if (!rls_dtype_ok(dtype) || !x || !g || !make_geom(ndims, shape, ntv, dims, &G))
    return rls_fail(ctx, RLS_E_INVALID, "tv_grad: bad argument");
  if (G.goff[G.ntv] == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(grad_kernel<float>, dim3(tv_grid(G.goff[G.ntv])), dim3(256), 0, ctx->stream, (const float*)x,
                       (float*)g, G, alpha, beta);
  else
    hipLaunchKernelGGL(grad_kernel<float2>, dim3(tv_grid(G.goff[G.ntv])), dim3(256), 0, ctx->stream, (const float2*)x,
                       (float2*)g, G, alpha, beta);
  return tv_status(ctx);
}

int32_t rls_tv_grad_t(rls_ctx* ctx, int32_t dtype, int32_t ndims, const int64_t* shape, int32_t ntv,
                      const int32_t* dims, const void* g, void* x, float alpha, float beta) {
  RLS_CHECK_CTX(ctx);
  tv_geom G;
  if (!rls_dtype_ok(dtype) || !x || !g || !make_geom(ndims, shape, ntv, dims, &G))
    return rls_fail(ctx, RLS_E_INVALID, "tv_grad_t: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(gradt_kernel<float>, dim3(tv_grid(G.n)), dim3(256), 0, ctx->stream, (const float*)g,
                       (const float*)x, (float*)x, G, alpha, beta);
  else
    hipLaunchKernelGGL(gradt_kernel<float2>, dim3(tv_grid(G.n)), dim3(256), 0, ctx->stream, (const float2*)g,
                       (const float2*)x, (float2*)x, G, alpha, beta);
  return tv_status(ctx);
}

int32_t rls_tv_restrict(rls_ctx* ctx, int32_t dtype, int64_t n, void* pq) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || n < 0 || (n > 0 && !pq)) return rls_fail(ctx, RLS_E_INVALID, "tv_restrict: bad argument");
  if (n == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(restrict_kernel<float>, dim3(tv_grid(n)), dim3(256), 0, ctx->stream, (float*)pq, n);
  else
    hipLaunchKernelGGL(restrict_kernel<float2>, dim3(tv_grid(n)), dim3(256), 0, ctx->stream, (float2*)pq, n);
  return tv_status(ctx);
}

int32_t rls_tv_lincomb(rls_ctx* ctx, int32_t dtype, int64_t n, void* rs, float t3, const void* pq, float t2,
                       const void* pqOld) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || n < 0 || (n > 0 && (!rs || !pq || !pqOld)))
    return rls_fail(ctx, RLS_E_INVALID, "tv_lincomb: bad argument");
  if (n == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(tv_lincomb_kernel<float>, dim3(tv_grid(n)), dim3(256), 0, ctx->stream, (float*)rs, t3,
                       (const float*)pq, t2, (const float*)pqOld, n);
  else
    hipLaunchKernelGGL(tv_lincomb_kernel<float2>, dim3(tv_grid(n)), dim3(256), 0, ctx->stream, (float2*)rs, t3,
                       (const float2*)pq, t2, (const float2*)pqOld, n);
  return tv_status(ctx);
}

int32_t rls_prox_tv_fgp(rls_ctx* ctx, int32_t dtype, int32_t ndims, const int64_t* shape, int32_t ntv,
                        const int32_t* dims, void* x, float lambda, int32_t iterations, void* workspace,
                        size_t workspace_bytes) {
  RLS_CHECK_CTX(ctx);
  tv_geom G;
  if (!rls_dtype_ok(dtype) || !x || iterations < 0 || !make_geom(ndims, shape, ntv, dims, &G))
    return rls_fail(ctx, RLS_E_INVALID, "prox_tv_fgp: bad argument");
  const size_t need = (size_t)(2 * G.goff[G.ntv] + G.n) * rls_elem_size(dtype);
  const bool fused = tv_single_ok(ctx, G, rls_elem_size(dtype));
  if (!fused && (!workspace || workspace_bytes < need))
    return rls_fail(ctx, RLS_E_WORKSPACE, "prox_tv_fgp: workspace too small");
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F32) return fgp_typed<float>(ctx, G, (float*)x, lambda, iterations, (float*)workspace);
  return fgp_typed<float2>(ctx, G, (float2*)x, lambda, iterations, (float2*)workspace);
}

}  // extern "C"
