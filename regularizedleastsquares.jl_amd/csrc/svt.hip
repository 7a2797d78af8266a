// Singular-value soft-thresholding of many small matrices: the proximal maps of
// NuclearRegularization (src/proximalMaps/ProxNuclear.jl:26-31) and of LLRRegularization with distinct blocks
// (src/proximalMaps/ProxLLR.jl:43-88; the reference moves GPU arrays to the CPU for this one).
//
// One wave owns one matrix X (rows = the block's voxels, columns = the K images) in LDS and runs a one-sided
// (Hestenes) Jacobi SVD on the SHORTER side: the nv = min(rows, K) vectors are rotated pairwise until they are
// orthogonal, X = W V^H with orthogonal columns w_u of norm sigma_u, and the thresholded matrix is
// sum_u max(sigma_u - lambda, 0) / sigma_u * w_u v_u^H.  Working on X^T when K > rows is exact:
// SVT(X^T) = SVT(X)^T.  No Gram matrix is formed, so small singular values keep their relative accuracy.
#include "rls_common.hpp"

struct svt_geom {
  int ndims;           // spatial dimensions (1..3)
  int64_t shape[3];    // image shape
  int64_t block[3];    // block size
  int64_t shift[3];    // circshift of the block grid (0 <= shift < shape)
  int64_t nblk[3];     // blocks per dimension (ceil)
  int64_t sstride;     // prod(shape): distance between images
  int K;               // number of images (last dimension)
  int mb;              // prod(block)
};

constexpr int SVT_MAX_SWEEPS = 40;

// element (row r of the block, image k) -> linear index into x, or -1 outside the image
__device__ static inline int64_t svt_index(const svt_geom& G, const int64_t (&b)[3], int r, int k) {
  int64_t lin = 0, stride = 1;
  int rr = r;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (d < G.ndims) {
      const int64_t i = rr % G.block[d];
      rr /= (int)G.block[d];
      const int64_t s = b[d] * G.block[d] + i;  // coordinate in the shifted image xs = circshift(x, shift)
      if (s >= G.shape[d]) return -1;
      int64_t pos = s - G.shift[d];             // xs[s] = x[s - shift]
      if (pos < 0) pos += G.shape[d];
      lin += pos * stride;
      stride *= G.shape[d];
    }
  }
  return lin + (int64_t)k * G.sstride;
}

// NW waves per matrix: the pairs of one round of the round-robin tournament are disjoint, so they are rotated at the same
// time -- one pair per group of SVT_SG = 16 lanes (a DPP row: its sums need no cross-row traffic), four pairs per wave,
// 4 NW per matrix -- and the waves meet at a barrier after every round (NW = 1: many small blocks, one wave each; NW up to
// 16: a single larger matrix, e.g. the nuclear-norm prox).  Round 3: a pair used to occupy a whole wave (64 lanes for
// vectors of 16-96 elements, every lane repeating the rotation's scalar arithmetic); a 96 x 96 nuclear-norm prox ran
// three pairs per wave one after the other, 4.9 ms against 3-4 ms for LAPACK on the host.
constexpr int SVT_SG = 16;
__device__ static inline float svt_row_sum(float v) {  // sum over the 16 lanes of a DPP row, in every lane of the row
  v += dpp_f(v, 0xB1);
  v += dpp_f(v, 0x4E);
  v += dpp_f(v, 0x141);
  v += dpp_f(v, 0x140);
  return v;
}
template <typename E, int NW>
__global__ __launch_bounds__(64 * NW) void svt_blocks_kernel(E* __restrict__ x, svt_geom G, float lam) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  constexpr int NT = 64 * NW;
  const bool tr = G.K > G.mb;             // orthogonalise the rows of X (= columns of X^T)
  const int nv = tr ? G.mb : G.K;         // number of vectors
  const int len = tr ? G.K : G.mb;        // their length
  E* W = reinterpret_cast<E*>(smem_raw);  // W[v * len + t]
  E* V = W + (size_t)nv * len;            // V[u * nv + v]  (column u of V)
  float* fac = reinterpret_cast<float*>(V + (size_t)nv * nv);
  float* sigv = fac + nv;                 // singular values
  float* red = sigv + nv;                 // [NW] per-wave scratch
  int64_t b[3] = {0, 0, 0};
  {
    int64_t id = blockIdx.x;
    for (int d = 0; d < G.ndims; ++d) {
      b[d] = id % G.nblk[d];
      id /= G.nblk[d];
    }
  }
  // gather (rows past the image edge are zero, ProxLLR.jl:66-67)
  float fro2 = 0.f;
  for (int e = tid; e < nv * len; e += NT) {
    const int v = e / len, t = e % len;
    const int r = tr ? v : t, k = tr ? t : v;
    const int64_t idx = svt_index(G, b, r, k);
    const E val = idx >= 0 ? x[idx] : elem<E>::zero();
    W[e] = val;
    fro2 += elem<E>::abs2(val);
  }
  for (int e = tid; e < nv * nv; e += NT) V[e] = (e / nv == e % nv) ? elem<E>::make(1.f, 0.f) : elem<E>::zero();
  fro2 = wave_sum(fro2);
  if (lane == 0) red[w] = fro2;
  __syncthreads();
  fro2 = 0.f;
  for (int i = 0; i < NW; ++i) fro2 += red[i];
  __syncthreads();
  const float tol = 2e-7f;
  const float null_tol = 1e-6f * sqrtf(fro2);  // singular values below this are noise of the Float32 data anyway
  const int m = nv + (nv & 1);                 // players of the tournament (a dummy one when nv is odd)
  for (int sweep = 0; sweep < SVT_MAX_SWEEPS; ++sweep) {
    float off = 0.f;
    for (int round = 0; round < m - 1; ++round) {
      const int sl = lane % SVT_SG;                                     // lane inside its group of 16
      for (int i = w * (64 / SVT_SG) + lane / SVT_SG; i < m / 2; i += NW * (64 / SVT_SG)) {  // uniform per group of 16 lanes
        int p = i == 0 ? m - 1 : (round + i) % (m - 1);
        int q = i == 0 ? round : (round - i + (m - 1)) % (m - 1);
        if (p > q) {
          const int tmp = p;
          p = q;
          q = tmp;
        }
        if (q >= nv) continue;  // the dummy player
        float a = 0.f, bb = 0.f, gr = 0.f, gi = 0.f;
        for (int t = sl; t < len; t += SVT_SG) {
          const E wp = W[p * len + t], wq = W[q * len + t];
          a += elem<E>::abs2(wp);
          bb += elem<E>::abs2(wq);
          const E g = elem<E>::mulc(wp, wq);  // conj(wp) * wq
          gr += elem<E>::re(g);
          gi += elem<E>::im(g);
        }
        a = svt_row_sum(a);
        bb = svt_row_sum(bb);
        gr = svt_row_sum(gr);
        if constexpr (elem<E>::cplx) gi = svt_row_sum(gi);
        // |g| and the phase without squaring g (g*g underflows for nearly-null vectors and the "unit" phase
        // g/sqrt(g*g) then is not of modulus one, which breaks X = W V^H); vectors that have been rotated down to
        // rounding level (rank-deficient blocks: zero-padded edge blocks, K > voxels) are left alone
        const float gabs = elem<E>::cplx ? hypotf(gr, gi) : fabsf(gr);
        const float na = sqrtf(a), nb = sqrtf(bb);
        if (gabs > tol * na * nb && na > null_tol && nb > null_tol) {  // uniform per group: its 16 lanes hold the same sums
          const float er = gr / gabs, ei = gi / gabs;       // e^{i phi}
          const float zeta = (bb - a) / (2.f * gabs);
          const float tt = (zeta >= 0.f ? 1.f : -1.f) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta));
          const float c = 1.f / sqrtf(1.f + tt * tt), s = c * tt;
          const E ph = elem<E>::make(er, ei);
          for (int t = sl; t < len; t += SVT_SG) {
            const E wp = W[p * len + t];
            const E qt = elem<E>::mulc(ph, W[q * len + t]);  // e^{-i phi} w_q
            W[p * len + t] = elem<E>::sub(elem<E>::scale(c, wp), elem<E>::scale(s, qt));
            W[q * len + t] = elem<E>::add(elem<E>::scale(s, wp), elem<E>::scale(c, qt));
          }
          for (int v = sl; v < nv; v += SVT_SG) {
            const E vp = V[p * nv + v];
            const E qt = elem<E>::mulc(ph, V[q * nv + v]);
            V[p * nv + v] = elem<E>::sub(elem<E>::scale(c, vp), elem<E>::scale(s, qt));
            V[q * nv + v] = elem<E>::add(elem<E>::scale(s, vp), elem<E>::scale(c, qt));
          }
          off = fmaxf(off, gabs / (na * nb));
        }
      }
      __syncthreads();  // the pairs of a round are disjoint; the next round re-pairs the vectors
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) off = fmaxf(off, __shfl_xor(off, sh, 64));  // over the wave's groups
    if (lane == 0) red[w] = off;
    __syncthreads();
    float offm = 0.f;
    for (int i = 0; i < NW; ++i) offm = fmaxf(offm, red[i]);
    __syncthreads();
    if (offm < tol) break;
  }
  // shrink factors  max(sigma - lambda, 0) / sigma   (prox!(L1Regularization, S, lambda) on the singular values)
  for (int u = w; u < nv; u += NW) {
    float a = 0.f;
    for (int t = lane; t < len; t += 64) a += elem<E>::abs2(W[u * len + t]);
    a = wave_sum(a);
    if (lane == 0) {
      const float sig = sqrtf(a);
      sigv[u] = sig;
      fac[u] = sig > 0.f ? fmaxf(sig - lam, 0.f) / sig : 0.f;
    }
  }
  __syncthreads();
  // Y = sum_u fac_u w_u V_u^H.  W and V carry the rounding of every rotation applied to them (about 1e-5 relative
  // at 96 x 96 in Float32), so the result is assembled from whichever part is SMALLER: the kept part directly, or
  // the input minus the removed part, Y = X - sum_u (1 - fac_u) w_u V_u^H with X re-read from memory -- the
  // accumulated rounding then scales with the smaller of the two norms (uniform choice per matrix; the direct form
  // also gives exact zeros when everything is thresholded away).
  float kept2 = 0.f, gone2 = 0.f;
  for (int u = 0; u < nv; ++u) {
    const float k1 = fac[u] * sigv[u], g1 = (1.f - fac[u]) * sigv[u];
    kept2 += k1 * k1;
    gone2 += g1 * g1;
  }
  const bool subtract = gone2 < kept2;
  for (int e = tid; e < nv * len; e += NT) {
    const int v = e / len, t = e % len;
    const int r = tr ? v : t, k = tr ? t : v;
    const int64_t idx = svt_index(G, b, r, k);
    E y = elem<E>::zero();
    for (int u = 0; u < nv; ++u) {
      const E vc = V[u * nv + v];
      const E wu = elem<E>::scale(subtract ? 1.f - fac[u] : fac[u], W[u * len + t]);
      // wu * conj(vc)
      y = elem<E>::add(y, elem<E>::make(elem<E>::re(wu) * elem<E>::re(vc) + elem<E>::im(wu) * elem<E>::im(vc),
                                         elem<E>::im(wu) * elem<E>::re(vc) - elem<E>::re(wu) * elem<E>::im(vc)));
    }
    if (idx >= 0) x[idx] = subtract ? elem<E>::sub(x[idx], y) : y;
  }
}

static int32_t svt_launch(rls_ctx* ctx, int32_t dtype, const svt_geom& G, void* x, float lam) {
  const int nv = G.K > G.mb ? G.mb : G.K, len = G.K > G.mb ? G.K : G.mb;
  const size_t es = rls_elem_size(dtype);
  const size_t lds = ((size_t)nv * len + (size_t)nv * nv) * es + (size_t)(2 * nv + 16) * sizeof(float);
  if (lds > 150 * 1024)
    return rls_fail(ctx, RLS_E_UNSUPPORTED, "singular-value thresholding: the matrix does not fit one CU's LDS");
  int64_t nb = 1;
  for (int d = 0; d < G.ndims; ++d) nb *= G.nblk[d];
  if (nb <= 0 || nb > 0x7fffffff) return rls_fail(ctx, RLS_E_INVALID, "singular-value thresholding: bad block count");
  // waves per matrix: one when there are blocks enough to keep every CU's wave slots busy, otherwise as many as give a CU
  // ~16 waves over the blocks it holds -- but no more than the pairs of a round need at four pairs per wave
  const int pairs = (nv + 1) / 2;
  int cap = 1;
  while (cap * 4 < pairs && cap < 16) cap *= 2;
  const int64_t blocks_per_cu = nb > 256 ? (nb + 255) / 256 : 1;
  const int want = (int)(16 / blocks_per_cu > 1 ? 16 / blocks_per_cu : 1);
  int nw = 1;
  while (nw * 2 <= want && nw * 2 <= cap) nw *= 2;
#define RLS_SVT_LAUNCH(EE, NWV)                                                                                        \
  do {                                                                                                                 \
    RLS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&svt_blocks_kernel<EE, NWV>),                       \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                          \
    hipLaunchKernelGGL((svt_blocks_kernel<EE, NWV>), dim3((unsigned)nb), dim3(64 * NWV), lds, ctx->stream, (EE*)x, G, lam); \
  } while (0)
  if (dtype == RLS_F32) {
    if (nw == 1) RLS_SVT_LAUNCH(float, 1);
    else if (nw == 2) RLS_SVT_LAUNCH(float, 2);
    else if (nw == 4) RLS_SVT_LAUNCH(float, 4);
    else if (nw == 8) RLS_SVT_LAUNCH(float, 8);
    else RLS_SVT_LAUNCH(float, 16);
  } else {
    if (nw == 1) RLS_SVT_LAUNCH(float2, 1);
    else if (nw == 2) RLS_SVT_LAUNCH(float2, 2);
    else if (nw == 4) RLS_SVT_LAUNCH(float2, 4);
    else if (nw == 8) RLS_SVT_LAUNCH(float2, 8);
    else RLS_SVT_LAUNCH(float2, 16);
  }
#undef RLS_SVT_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

extern "C" {

int32_t rls_prox_nuclear(rls_ctx* ctx, int32_t dtype, int64_t m, int64_t n, void* x, float lambda) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || m <= 0 || n <= 0 || !x || m > 0x7fffffff || n > 0x7fffffff)
    return rls_fail(ctx, RLS_E_INVALID, "prox_nuclear: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  svt_geom G{};
  G.ndims = 1;
  G.shape[0] = G.block[0] = m;
  G.nblk[0] = 1;
  G.sstride = m;
  G.K = (int)n;
  G.mb = (int)m;
  return svt_launch(ctx, dtype, G, x, lambda);
}

int32_t rls_prox_llr(rls_ctx* ctx, int32_t dtype, int32_t ndims, const int64_t* shape, const int64_t* block,
                     const int64_t* shift, int64_t n, void* x, float lambda) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || ndims < 1 || ndims > 3 || !shape || !block || !x || n <= 0)
    return rls_fail(ctx, RLS_E_INVALID, "prox_llr: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  svt_geom G{};
  G.ndims = ndims;
  int64_t ns = 1, mb = 1;
  for (int d = 0; d < ndims; ++d) {
    if (shape[d] <= 0 || block[d] <= 0) return rls_fail(ctx, RLS_E_INVALID, "prox_llr: bad shape / blockSize");
    G.shape[d] = shape[d];
    G.block[d] = block[d];
    int64_t sh = shift ? shift[d] % shape[d] : 0;
    if (sh < 0) sh += shape[d];
    G.shift[d] = sh;
    G.nblk[d] = (shape[d] + block[d] - 1) / block[d];
    ns *= shape[d];
    mb *= block[d];
  }
  if (n % ns != 0 || mb > 0x7fffffff || n / ns > 0x7fffffff)
    return rls_fail(ctx, RLS_E_INVALID, "prox_llr: length(x) is not a multiple of prod(shape)");
  G.sstride = ns;
  G.K = (int)(n / ns);
  G.mb = (int)mb;
  return svt_launch(ctx, dtype, G, x, lambda);
}

}  // extern "C"
