// Resident CGNR on a 16 x 16 grid of workgroups: three light hand-offs per iteration instead of one heavy all-reduce.
//
// The resident kernel of normal.hip gives a workgroup 16 ROWS of A and all N columns: every workgroup produces a partial row of
// v = A^H (A p) (16 KiB at the headline shape) and 256 of those are summed grid-wide -- 7.1 of the 11.2 us an iteration costs
// (rows out, group barrier, slice sums, grid barrier, 128 KiB read back by every workgroup).  Here workgroup
// (i, j) keeps the TILE of rows block i x columns block j (256 x 128 complex at 4096 x 2048: the same 256 KiB of registers), and
// what travels per iteration (src/CGNR.jl:143-178) is small:
//   A  t_i  = sum_j A_ij p_j        256 values per workgroup, summed over the 16 workgroups of ROW group i  (one XCD)
//   B  v_j  = sum_i A_ij^H t_i      128 values per workgroup, summed over the 16 workgroups of COLUMN group j, and with them
//                                   the 256 partial dots <p_j, A_ij^H t_i> (linear in the partials): alpha after ONE grid barrier
//   C  ||r||^2 = sum_j ||r_j||^2    one number per workgroup, summed over the row group (which holds every j): beta
// The vectors are distributed: workgroup (i, j) owns x_j, r_j, p_j (replicated over i, one element per thread of its first two
// waves).  Every sum runs in a fixed order (j = 0..15, i = 0..15, workgroup 0..255), so the bits are a function of the shape
// alone: one launch of 32 iterations equals four of 8.  Hand-offs are sc1 stores / sc1 loads behind bounded arrival counters
// (resident_sync.hpp); a launch in which any wait runs out changes nothing -- the state is written back behind a final grid
// barrier that only a fully alive grid passes -- and the host re-runs it on the streaming pipeline, as for the other resident
// kernels.
#include "rls_common.hpp"
#include "resident_sync.hpp"

#ifdef RLS_STAMPS
// diagnostic build (tools/build_stamps.sh): wall-clock stamps (100 MHz) of workgroup 0's last iteration
__device__ unsigned long long g_r2_stamps[32];
#define R2_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_r2_stamps[k] = wall_clock64(); } while (0)
extern "C" int rls_debug_r2_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_r2_stamps), sizeof(unsigned long long) * 32);
}
#else
#define R2_STAMP(k)
#endif

namespace {

constexpr int R2_NT = 512, R2_WV = 8;
typedef float2 C2;

// exchange block (one allocation per plan), for RB rows / CB columns per workgroup
template <int R, int C>
struct r2_cfg {
  static constexpr int RB = 32 * R, CB = 16 * C;
  static constexpr size_t T_OFF = 0;                                               // T  [16 i][16 j][RB]        complex
  static constexpr size_t V_OFF = T_OFF + (size_t)256 * RB * sizeof(C2);           // V  [2][16 j][16 i][CB]     complex
  static constexpr size_t D_OFF = V_OFF + (size_t)2 * 256 * CB * sizeof(C2);       // D  [2][256][4]             double
  static constexpr size_t RR_OFF = D_OFF + (size_t)2 * 256 * 4 * sizeof(double);   // RR [16 i][16 j]            double
  static constexpr size_t BYTES = RR_OFF + (size_t)256 * sizeof(double);
};

template <int R, int C>
struct r2_lds {
  C2 ps[16 * C];          // p_j
  C2 half[2][32 * R];     // hand-off A: the two halves (sources 0..7, 8..15) of t_i
  C2 vpart[R2_WV][16 * C];  // product 2 per wave; then the four quarters of v_j (hand-off B)
  double dred[4][4];      // hand-off B: the dots, per wave
  double wpart[4];        // a second wave's share of a two-wave sum
  int flag;
};

__device__ static inline C2 row16_sum_c(C2 v) {  // all-reduce over the 16 lanes of a DPP row, fixed order
  float re = v.x, im = v.y;
  re += dpp_f(re, 0xB1); im += dpp_f(im, 0xB1);
  re += dpp_f(re, 0x4E); im += dpp_f(im, 0x4E);
  re += dpp_f(re, 0x141); im += dpp_f(im, 0x141);
  re += dpp_f(re, 0x140); im += dpp_f(im, 0x140);
  return make_float2(re, im);
}
__device__ static inline double sc1_load_f64(const double* p) {
  return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ static inline void sc1_store_f64(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int R, int C>
__global__ __launch_bounds__(R2_NT) void cgnr_resident2d_kernel(const C2* __restrict__ A, int64_t lda, C2* x, C2* r, C2* p, C2* v,
                                                                cgnr_scalars* sc, resident_sync* sync, char* xb, int n_steps,
                                                                unsigned spin_limit) {
  using K = r2_cfg<R, C>;
  constexpr int RB = K::RB, CB = K::CB;
  static_assert(CB <= 128 && RB <= 256, "one element of the column block per thread of two waves; t_i in two halves of 256");
  __shared__ r2_lds<R, C> L;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int cb = tid & 15, rb = tid >> 4;
  // workgroup -> (i, j): the 16 workgroups of a row group share blockIdx % 8 (one XCD under round-robin dispatch; placement
  // changes speed only)
  const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
  const int gi = xcd * 2 + (slot >> 4), gj = slot & 15;
  C2* T = reinterpret_cast<C2*>(xb + K::T_OFF);
  C2* V = reinterpret_cast<C2*>(xb + K::V_OFF);
  double* D = reinterpret_cast<double*>(xb + K::D_OFF);
  double* RRb = reinterpret_cast<double*>(xb + K::RR_OFF);
  cgnr_scalars S = *sc;
  // ---- this workgroup's tile into registers; its slice of the vectors ------------------------------------------------------------
  C2 a[R][C];
  {
    const C2* At = A + ((int64_t)gj * CB + cb * C) * lda + (int64_t)gi * RB + rb * R;
#pragma unroll
    for (int jc = 0; jc < C; ++jc) {
#pragma unroll
      for (int q = 0; q < R / 2; ++q) {
        const f4 two = *reinterpret_cast<const f4*>(At + (int64_t)jc * lda + 2 * q);
        a[2 * q][jc] = make_float2(two[0], two[1]);
        a[2 * q + 1][jc] = make_float2(two[2], two[3]);
      }
    }
  }
  const bool own = tid < CB;  // the threads that hold x_j, r_j, p_j, v_j (one element each)
  const int64_t col = (int64_t)gj * CB + (own ? tid : 0);
  C2 xe = x[col], re_ = r[col], pe = p[col], ve = v[col];
  if (own) L.ps[tid] = pe;
  if (S.done || n_steps <= 0) return;  // uniform
  __syncthreads();
  unsigned eg = 0, ep = 0;  // arrivals so far on the row group's word / on the grid counter
  unsigned* gword = sync->gcnt + gi * 16;
  bool alive = true;
  int par = 0;
  for (int it = 0; it < n_steps; ++it) {
    R2_STAMP(0);
    // ---- product 1: this tile's share of t_i, out to the row group ---------------------------------------------------------------
    {
      C2 pj[C];
#pragma unroll
      for (int jc = 0; jc < C; ++jc) pj[jc] = L.ps[cb * C + jc];
      C2 mine = make_float2(0.f, 0.f);
#pragma unroll
      for (int i = 0; i < R; ++i) {
        C2 s = make_float2(0.f, 0.f);
#pragma unroll
        for (int jc = 0; jc < C; ++jc) s = elem<C2>::fma_pk(a[i][jc], pj[jc], s);
        s = row16_sum_c(s);
        mine = cb == i ? s : mine;  // lane cb < R hands row cb of the row block over
      }
      if (cb < R) sc1_store_elem<C2>(T + ((int64_t)(gi * 16 + gj) * RB + rb * R + cb), mine);
    }
    R2_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    R2_STAMP(2);
    if (!group_arrive_wait(gword, 16u * ++eg, spin_limit, &L.flag)) {
      alive = false;
      break;
    }
    R2_STAMP(3);
    {  // t_i = sum over the 16 column blocks, sources in order; thread (element e, half h) sums 8 of them
      const int e = tid & 255, h = tid >> 8;
      if (e < RB) {
        C2 part[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) part[q] = sc1_load_elem<C2>(T + ((int64_t)(gi * 16 + 8 * h + q) * RB + e));
        C2 s = part[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) s = elem<C2>::add(s, part[q]);
        L.half[h][e] = s;
      }
    }
    __syncthreads();
    R2_STAMP(4);
    // ---- product 2: this tile's share of v_j ------------------------------------------------------------------------------------------
    {
      C2 tv[R];
#pragma unroll
      for (int i = 0; i < R; ++i) tv[i] = elem<C2>::add(L.half[0][rb * R + i], L.half[1][rb * R + i]);
#pragma unroll
      for (int jc = 0; jc < C; ++jc) {
        C2 s = make_float2(0.f, 0.f);
#pragma unroll
        for (int i = 0; i < R; ++i) s = elem<C2>::fmac_pk(a[i][jc], tv[i], s);  // conj(a) t
        s = make_float2(s.x + __shfl_xor(s.x, 16, 64), s.y + __shfl_xor(s.y, 16, 64));
        s = make_float2(s.x + __shfl_xor(s.x, 32, 64), s.y + __shfl_xor(s.y, 32, 64));
        if (lane < 16) L.vpart[w][cb * C + jc] = s;
      }
    }
    __syncthreads();
    R2_STAMP(5);
    double dre = 0.0, dim_ = 0.0, dpp = 0.0;
    if (own) {
      C2 s = L.vpart[0][tid];
#pragma unroll
      for (int ww = 1; ww < R2_WV; ++ww) s = elem<C2>::add(s, L.vpart[ww][tid]);
      sc1_store_elem<C2>(V + (((int64_t)par * 16 + gj) * 16 + gi) * CB + tid, s);
      // this tile's term of <p, v> (linear in the partial) and, from the first row group only, of ||p||^2
      dre = (double)pe.x * (double)s.x + (double)pe.y * (double)s.y;
      dim_ = (double)pe.x * (double)s.y - (double)pe.y * (double)s.x;
      if (gi == 0) dpp = (double)pe.x * (double)pe.x + (double)pe.y * (double)pe.y;
    }
    if (w < 2) {  // (CB <= 64: wave 1 contributes zeros)
      dre = wave_sum(dre);
      dim_ = wave_sum(dim_);
      dpp = wave_sum(dpp);
      if (w == 1 && lane == 0) {
        L.wpart[0] = dre;
        L.wpart[1] = dim_;
        L.wpart[2] = dpp;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      double* d = D + ((int64_t)par * 256 + b) * 4;
      sc1_store_f64(d, dre + L.wpart[0]);
      sc1_store_f64(d + 1, dim_ + L.wpart[1]);
      sc1_store_f64(d + 2, dpp + L.wpart[2]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    R2_STAMP(6);
    if (!grid_arrive_wait(sync->cnt, ++ep, 256u, spin_limit, &L.flag)) {
      alive = false;
      break;
    }
    R2_STAMP(7);
    // ---- v_j (16 row blocks, in order; thread (element e, quarter q) sums 4) and the dots (256 tiles, in order) -----------------------
    {
      const int e = tid & 127, q = tid >> 7;
      if (e < CB) {
        C2 part[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) part[k] = sc1_load_elem<C2>(V + (((int64_t)par * 16 + gj) * 16 + 4 * q + k) * CB + e);
        C2 s = elem<C2>::add(elem<C2>::add(elem<C2>::add(part[0], part[1]), part[2]), part[3]);
        L.vpart[q][e] = s;
      }
      double d0 = 0.0, d1 = 0.0, d2 = 0.0;
      if (tid < 256) {
        const double* d = D + ((int64_t)par * 256 + tid) * 4;
        d0 = sc1_load_f64(d);
        d1 = sc1_load_f64(d + 1);
        d2 = sc1_load_f64(d + 2);
      }
      if (w < 4) {
        d0 = wave_sum(d0);
        d1 = wave_sum(d1);
        d2 = wave_sum(d2);
        if (lane == 0) {
          L.dred[w][0] = d0;
          L.dred[w][1] = d1;
          L.dred[w][2] = d2;
        }
      }
    }
    __syncthreads();
    R2_STAMP(8);
    const double nre = ((L.dred[0][0] + L.dred[1][0]) + L.dred[2][0]) + L.dred[3][0];
    const double nim = ((L.dred[0][1] + L.dred[1][1]) + L.dred[2][1]) + L.dred[3][1];
    const double pp = ((L.dred[0][2] + L.dred[1][2]) + L.dred[2][2]) + L.dred[3][2];
    const float lambda = S.lambda;
    const double zeta = S.rr;
    const dcomplex alpha = dc_div({zeta, 0.0}, {nre + (lambda > 0.f ? (double)lambda * pp : 0.0), nim});  // src/CGNR.jl:153-158
    const C2 al = make_float2((float)alpha.re, (float)alpha.im), na = make_float2(-(float)alpha.re, -(float)alpha.im);
    double rrj = 0.0;
    if (own) {
      ve = elem<C2>::add(elem<C2>::add(elem<C2>::add(L.vpart[0][tid], L.vpart[1][tid]), L.vpart[2][tid]), L.vpart[3][tid]);
      xe = elem<C2>::fma(pe, al, xe);
      C2 rn = elem<C2>::fma(ve, na, re_);
      if (lambda > 0.f) rn = elem<C2>::fma(elem<C2>::scale(-lambda, pe), al, rn);
      re_ = rn;
      rrj = (double)rn.x * (double)rn.x + (double)rn.y * (double)rn.y;
    }
    if (w < 2) {
      rrj = wave_sum(rrj);
      if (w == 1 && lane == 0) L.wpart[3] = rrj;
    }
    __syncthreads();
    if (tid == 0) {
      sc1_store_f64(RRb + gi * 16 + gj, rrj + L.wpart[3]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    R2_STAMP(9);
    if (!group_arrive_wait(gword, 16u * ++eg, spin_limit, &L.flag)) {
      alive = false;
      break;
    }
    R2_STAMP(10);
    // ---- ||r||^2 over the 16 column blocks (in order), beta, p ------------------------------------------------------------------------
    double rr = lane < 16 ? sc1_load_f64(RRb + gi * 16 + lane) : 0.0;
    rr = wave_sum(rr);
    const double beta = rr / zeta;
    const float bf = (float)beta;
    if (own) {
      pe = elem<C2>::add(elem<C2>::scale(bf, pe), re_);
      L.ps[tid] = pe;
    }
    S.zeta = zeta;
    S.rr = rr;
    S.alpha_re = alpha.re;
    S.alpha_im = alpha.im;
    S.beta_re = beta;
    S.beta_im = 0.0;
    S.iteration += 1;
    const float ratio = (float)(sqrt(rr) / S.z0);
    S.done = (ratio <= S.rel_tol) || (S.iteration >= S.max_iter);  // src/CGNR.jl:181-185
    par ^= 1;
    __syncthreads();
    R2_STAMP(11);
    if (S.done) break;  // uniform: every workgroup derived the same scalars
  }
  // ---- commit: only a grid in which every workgroup got here writes anything back ---------------------------------------------------
  if (alive) alive = grid_arrive_wait(sync->cnt, ++ep, 256u, spin_limit, &L.flag);
  if (!alive) {
    resident_give_up(sync, nullptr);
    return;
  }
  if (gi == 0 && own) {
    x[col] = xe;
    r[col] = re_;
    p[col] = pe;
    v[col] = ve;
  }
  if (b == 0 && tid == 0) {
    S.pending = 0;
    S.cur = 0;
    S.fresh = 0;
    *sc = S;
    sync->completed = 1u;
  }
}

template <int R, int C>
static bool r2_fits(int device) {
  int cus = 0, blocks = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 256) return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, cgnr_resident2d_kernel<R, C>, R2_NT, 0) != hipSuccess) return false;
  return blocks >= 1;
}

}  // namespace

// shapes: ComplexF32, M = 16 x 32 R rows, N = 16 x 16 C columns with R, C in {4, 8}: 4096 / 2048 x 2048 / 1024
static bool r2_shape(int64_t M, int64_t N, int* R, int* C) {
  if (M % 512 || N % 256) return false;
  *R = (int)(M / 512);
  *C = (int)(N / 256);
  return (*R == 4 || *R == 8) && (*C == 4 || *C == 8);
}

size_t rls_resident2d_bytes(int64_t M, int64_t N) {
  int R, C;
  if (!r2_shape(M, N, &R, &C)) return 0;
  if (R == 8 && C == 8) return r2_cfg<8, 8>::BYTES;
  if (R == 8 && C == 4) return r2_cfg<8, 4>::BYTES;
  if (R == 4 && C == 8) return r2_cfg<4, 8>::BYTES;
  return r2_cfg<4, 4>::BYTES;
}

bool rls_resident2d_ok(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda) {
  int R, C;
  if (!ctx || dtype != RLS_C32 || !A || !r2_shape(M, N, &R, &C)) return false;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (lda & 1) || lda < M) return false;  // 16-byte pieces of a column
  if (R == 8 && C == 8) return r2_fits<8, 8>(ctx->device);
  if (R == 8 && C == 4) return r2_fits<8, 4>(ctx->device);
  if (R == 4 && C == 8) return r2_fits<4, 8>(ctx->device);
  return r2_fits<4, 4>(ctx->device);
}

int32_t rls_resident2d_launch(rls_ctx* ctx, const rls_cgnr_pipe& P, void* sync, void* xb, int n_steps, unsigned spin_limit) {
  int R, C;
  if (!r2_shape(P.M, P.N, &R, &C)) return rls_fail(ctx, RLS_E_UNSUPPORTED, "resident 2-D CGNR: shape not instantiated");
#define R2_LAUNCH(RR, CC)                                                                                                          \
  hipLaunchKernelGGL((cgnr_resident2d_kernel<RR, CC>), dim3(256), dim3(R2_NT), 0, ctx->stream, (const C2*)P.A, P.lda, (C2*)P.x,   \
                     (C2*)P.r0, (C2*)P.p0, (C2*)P.v, P.sc, (resident_sync*)sync, (char*)xb, n_steps, spin_limit)
  if (R == 8 && C == 8) R2_LAUNCH(8, 8);
  else if (R == 8 && C == 4) R2_LAUNCH(8, 4);
  else if (R == 4 && C == 8) R2_LAUNCH(4, 8);
  else R2_LAUNCH(4, 4);
#undef R2_LAUNCH
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}
