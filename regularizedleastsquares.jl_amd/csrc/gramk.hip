// Up to 8 ComplexF32 right-hand sides sharing one EXPLICIT Gram matrix, the whole rls_cgnr_step call in ONE launch.
//
// The reference's matrix solve deep-copies the solver STATE per column; every column shares the one solver.AHA
// (src/MultiThreading.jl:30-48), which for a dense matrix is AHA = A' * A by the constructors' default (src/CGNR.jl:49),
// applied as mul!(v, AHA, p) (src/CGNR.jl:151).  AHA at N = 2048 is 32 MiB -- a quarter of the chip's register files -- so,
// as in cgnr_gram_resident_kernel (normal.hip), every workgroup keeps its rows of AHA in VGPRs for the whole call:
//   * workgroup b owns rows 8 b .. 8 b + 7 of AHA (64 bytes per column) as the A operand of v_mfma_f32_16x16x4_f32:
//     the 16 operand rows are the (re, im) halves of its 8 complex rows, the 16 operand columns the (re | im) halves of
//     the 8 right-hand sides, so ONE MFMA per four columns of AHA yields all four real products of the complex block;
//   * the operand panel P (N x 8 complex, 128 KiB) lives in LDS, replicated in every workgroup, and so do r and the
//     per-column scalars: the CG update (src/CGNR.jl:153-176) runs redundantly in every workgroup on identical inputs in
//     an identical order, so no scalar travels and the replicas stay bit-identical;
//   * the only exchange per iteration is the all-gather of V = AHA P (N x 8 values: each workgroup publishes its 8 rows,
//     512 bytes) and of the per-workgroup partial dots <p, v>, ||p||^2 -- ONE grid barrier, against the 32 MiB of partial
//     rows a matrix-free batched variant would have to exchange;
//   * x is distributed (a workgroup advances its own 8 rows) and gathered once, at the end of the launch.
// Hand-offs follow the protocol of resident_sync.hpp (sc1 stores drained by the storing wave, workgroup barrier, one
// arrival per workgroup, bounded polls); a launch that cannot get its grid onto the chip changes nothing (only workgroup
// 0 writes the caller's state, after its last barrier) and the host re-runs it on the streaming kernels (skinny.hip).
#include "resident_sync.hpp"
#include "rls_common.hpp"

typedef float gk_f32x4 __attribute__((ext_vector_type(4)));

// Diagnostic build only (-DRLS_STAMPS, tools/build_stamps.sh): 100 MHz wall-clock stamps of the phases of the LAST iteration
// of a launch in seven sampled workgroups.  Never compiled into the shipped library.
#ifdef RLS_STAMPS
__device__ unsigned long long g_gk_stamps[16 * 8];
#define GK_STAMP(slot)                                                                           \
  do {                                                                                           \
    if (threadIdx.x == 0 && (blockIdx.x % 37) == 5 && blockIdx.x / 37 < 7)                       \
      g_gk_stamps[(blockIdx.x / 37) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime();           \
    if (threadIdx.x == 0 && blockIdx.x == 0) g_gk_stamps[7 * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); /* row 7: workgroup 0 */ \
  } while (0)
extern "C" int32_t rls_debug_gk_stamps(unsigned long long* out_h) {
  return (int32_t)hipMemcpyFromSymbol(out_h, HIP_SYMBOL(g_gk_stamps), sizeof(unsigned long long) * 16 * 8);
}
#else
#define GK_STAMP(slot) \
  do {                 \
  } while (0)
#endif

#ifdef GK_SYNC
#define lds_barrier __syncthreads
#endif

namespace {

constexpr int GK_WV = 8, GK_NT = GK_WV * 64;
constexpr int GK_SPW = 64;                  // MFMA steps (4 columns of AHA each) per wave at N = GK_NMAX
constexpr int GK_NMAX = GK_WV * GK_SPW * 4; // 2048
constexpr int GK_ROWS = 8, GK_KB = 8;       // complex rows of AHA per workgroup; right-hand sides per launch
#ifndef GK_VB_EARLY
#define GK_VB_EARLY 1  // all of a thread's V loads requested right behind the grid barrier (0: the second half when alpha is known)
#endif

// ---- operand panel in LDS ------------------------------------------------------------------------------------------
// Rows in groups of 32; a group is four REGIONS of 32 pieces of 16 bytes -- region 2 part + h holds, for each of the 32 rows,
// the re (part 0) or im (part 1) values of right-hand sides 4 h .. 4 h + 3 -- and every region is followed by 32 bytes of
// padding, so that
//   * the update's accesses (a wave = 64 consecutive rows of ONE region kind) are 512 contiguous bytes per group:
//     conflict-free 16-byte accesses;
//   * the MFMA B operand of a step (4 consecutive rows x 4 regions x 4 words) is 64 distinct words whose two half-waves
//     (2 rows x 4 regions) land in 8 different 16-byte bank groups (the padding shifts region r by 2 r pieces).
constexpr uint32_t GK_REG = 32u * 16u + 32u;  // region stride (bytes)
constexpr uint32_t GK_GRP = 4u * GK_REG;      // group stride: 2176 bytes per 32 rows
__device__ static inline uint32_t gk_piece_off(uint32_t n, uint32_t h, uint32_t part) {
  return (n >> 5) * GK_GRP + (2u * part + h) * GK_REG + (n & 31u) * 16u;
}
// byte offset of operand column j (j < 8: re of right-hand side j, j >= 8: im of right-hand side j - 8) of row n
__device__ static inline uint32_t gk_elem_off(uint32_t n, uint32_t j) {
  const uint32_t k = j & 7u;
  return gk_piece_off(n, k >> 2, j >> 3) + (k & 3u) * 4u;
}

// slots per thread for N rows: N rounded up to 256 x {1, 2, 4, 8}
__host__ __device__ static inline int gk_ne(int64_t N) { return N <= 256 ? 1 : N <= 512 ? 2 : N <= 1024 ? 4 : 8; }
__host__ __device__ static inline size_t gk_panel_bytes(int64_t N) { return (size_t)gk_ne(N) * 8 * GK_GRP; }

struct gk_col {  // one right-hand side's solver scalars, replicated in LDS
  double rr, z0, zeta, alpha_re, alpha_im, beta;
  float lambda, rel_tol;
  int iteration, max_iter, done, active;  // active: takes part in the current iteration (not done at its start)
  float a_re, a_im, b_f, pad;
};

struct gk_lds_tail {
  float red[GK_WV][256];
  double dsum[16 * 32];
  double rrw[GK_WV][4];
  gk_col cs[GK_KB];
  float stage[GK_ROWS * 16];
  float2 xown[GK_ROWS * GK_KB], vown[GK_ROWS * GK_KB];  // this workgroup's rows of x and of the last applied V (wave 0)
  int flag, alld;
};

__device__ static inline void gk_sc1_store16(__amdgpu_buffer_rsrc_t rs, uint32_t off, f4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rs, off, 0, 16);
}
__device__ static inline void gk_sc1_store_f64(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// Work split of the replicated update: WAVE w owns right-hand sides 4 h .. 4 h + 3 with h = w & 1 -- so alpha, beta and the
// retirement flags of "its" columns are wave-uniform (scalar registers, no per-lane selects) and ||r||^2 is a plain wave sum --
// for rows n = 256 e + 64 (w >> 1) + lane, e = 0 .. 7 ("slots").  r lives in registers as (re, im) pairs, p in the panel.
//
// FULL: N == 2048 (no clamps).  Otherwise N is a multiple of 16, N <= 2048 (slots / MFMA steps past N skipped uniformly).
// NE: slots per thread = panel rows / 256 (1, 2, 4 or 8: N rounded up to 256, 512, 1024, 2048 rows; rows past N are zero and
// stay zero); FULL: N == 256 NE, no clamps anywhere.  Inside the iteration loop nothing depends on N.
template <int NE, bool FULL>
__global__ __launch_bounds__(GK_NT) void cgnr_gramk_resident_kernel(rls_gramk D, resident_sync* sync, int n_steps,
                                                                    unsigned spin_limit) {
  extern __shared__ __align__(16) char gk_lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  GK_STAMP(8);
  const int N = FULL ? 256 * NE : (int)D.N;
  const int nwg = gridDim.x, b = blockIdx.x, row0 = b * GK_ROWS;
  // MFMA steps (4 columns of AHA) per wave, a multiple of 8 = one 32-row group of the panel: wave w takes steps
  // [w spw, (w + 1) spw); the panel has npad = 32 spw >= N rows (zero beyond N), i.e. ne = npad / 256 slots per thread
  constexpr int spw = 8 * NE, npad = 256 * NE;
  char* pl = gk_lds;
  const uint32_t tail0 = (uint32_t)(npad / 32) * GK_GRP;
  gk_lds_tail& T0 = *reinterpret_cast<gk_lds_tail*>(gk_lds + tail0);
  const int nrhs = D.nrhs;
  const int h = w & 1;                        // this wave's column half
  const int nl = (w >> 1) * 64 + lane;        // its row in slot 0

  // ---- state in: scalars, AHA rows (registers), r (registers), p (LDS panel), own rows of x ------------------------------
  if (tid < GK_KB) {  // field by field: a struct copy of the device scalars can end up in a private alloca (rls_common.hpp)
    gk_col& c = T0.cs[tid];
    const bool real_col = tid < nrhs;
    const cgnr_scalars* s = D.sc + (real_col ? tid : 0);
    c.rr = real_col ? s->rr : 0.0;
    c.z0 = real_col ? s->z0 : 0.0;
    c.zeta = real_col ? s->zeta : 0.0;
    c.alpha_re = real_col ? s->alpha_re : 0.0;
    c.alpha_im = real_col ? s->alpha_im : 0.0;
    c.beta = real_col ? s->beta_re : 0.0;
    c.lambda = real_col ? s->lambda : 0.f;
    c.rel_tol = real_col ? s->rel_tol : 0.f;
    c.iteration = real_col ? s->iteration : 0;
    c.max_iter = real_col ? s->max_iter : 0;
    c.done = real_col ? s->done : 1;  // padding columns never take part
    c.active = real_col ? !s->done : 0;  // takes part in the next iteration
    c.a_re = c.a_im = c.b_f = c.pad = 0.f;
  }
  // LDS offsets of this thread's two pieces of slot e: lane part + 17408 e (+ 2 regions for the im piece).  DS immediates
  // reach 64 KiB - 1, so slots 4 .. 7 go through a second (opaque) base; left to the compiler every slot got its own register.
  const uint32_t pc_lo = (uint32_t)(2 * (w >> 1) + (lane >> 5)) * GK_GRP + (uint32_t)h * GK_REG + (uint32_t)(lane & 31) * 16u;
  uint32_t pc_hi = pc_lo + (NE > 4 ? 4u * 8u * GK_GRP : 0u);
  asm volatile("" : "+v"(pc_hi));
#define GK_PC(e, part) (pl + ((e) < 4 ? pc_lo + (uint32_t)(e) * 8u * GK_GRP : pc_hi + (uint32_t)((e) - 4) * 8u * GK_GRP) + (part) * 2u * GK_REG)
  float2 r2[NE][4];
  {
    // column-major state: column 4 h + j (uniform), row nl + 256 e: one buffer resource per array, ONE lane offset, the rest
    // uniform.  Columns past nrhs re-read a valid one and are zeroed.
    const __amdgpu_buffer_rsrc_t p_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(D.P), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(D.R), 0, 0xffffffff, 0x00020000);
    const uint32_t ldvb = (uint32_t)D.ldv * 8u;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      {
        const int n = nl + 256 * e;
        const bool rowok = FULL || n < N;
        const uint32_t voff = (uint32_t)(FULL ? nl : (rowok ? n : 0)) * 8u;
        f4 pr4, pi4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = 4 * h + j;
          const bool ok = rowok && k < nrhs;
          const uint32_t so = (uint32_t)(k < nrhs ? k : 0) * ldvb + (FULL ? (uint32_t)e * 2048u : 0u);
          const float2 pv = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(p_rs, voff, so, 0));
          const float2 rv = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r_rs, voff, so, 0));
          pr4[j] = ok ? pv.x : 0.f;
          pi4[j] = ok ? pv.y : 0.f;
          r2[e][j] = ok ? rv : make_float2(0.f, 0.f);
        }
        *reinterpret_cast<f4*>(GK_PC(e, 0)) = pr4;
        *reinterpret_cast<f4*>(GK_PC(e, 1)) = pi4;
      }
    }
  }
  if (tid < GK_ROWS * GK_KB) {
    const int xr = tid >> 3, xk = tid & 7;
    T0.xown[tid] = xk < nrhs ? reinterpret_cast<const float2*>(D.X)[(int64_t)xk * D.ldv + row0 + xr] : make_float2(0.f, 0.f);
    T0.vown[tid] = make_float2(0.f, 0.f);
  }
  float ga[spw];
  {
    const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(D.G), 0, 0xffffffff, 0x00020000);
    const uint32_t colb = (uint32_t)D.ldg * 8u;  // host: ldg * 8 * N < 2^32
    const uint32_t voff = (uint32_t)row0 * 8u + (uint32_t)(lane & 15) * 4u + (uint32_t)(lane >> 4) * colb;
#pragma unroll
    for (int s = 0; s < spw; ++s) {
      const int c0 = 4 * (w * spw + s);  // uniform; N % 4 == 0, so a step is inside the matrix or outside as a whole
      const bool ok = FULL || c0 < N;
      const float g = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_rs, voff, (uint32_t)(ok ? c0 : 0) * colb, 0));
      ga[s] = ok ? g : 0.f;
    }
  }
  __syncthreads();
  GK_STAMP(9);
  const __amdgpu_buffer_rsrc_t vx_rs = sc1_rsrc(D.Vx);
  // exchanged V: [parity][column half h][piece][row][16 bytes]; piece 0 = the (re, im) pairs of columns 4 h, 4 h + 1, piece 1 =
  // of columns 4 h + 2, 4 h + 3 (operands of the update's FMAs as they come); a wave's load of one piece is 1 KiB contiguous
  const uint32_t vx_piece = (uint32_t)npad * 16u, vx_par = 4u * vx_piece;
  const uint32_t vx_lane = (uint32_t)(2 * h) * vx_piece + (uint32_t)nl * 16u;
  // this lane's B-operand address inside a step: rows 4 s' + (lane >> 4), operand column lane & 15
  const uint32_t jq = (uint32_t)(lane >> 4), jj = (uint32_t)(lane & 15);
  const uint32_t lane_const = (2u * (jj >> 3) + ((jj >> 2) & 1u)) * GK_REG + jq * 16u + (jj & 3u) * 4u;
  unsigned epoch = 0;
  bool alive = true;
  int it = 0;
  for (; it < n_steps; ++it) {
    const int q = it & 1;
    // Per-iteration opaque copies of the thread id and of the LDS tail's offset: every address derived from them is
    // recomputed where it is used (one or two VALU instructions) instead of living in a register across the whole loop --
    // with 128 registers of AHA and r per lane, hoisted invariants were what spilled.
    int tl = tid;
    uint32_t tail_off = tail0;
#ifndef GK_NO_OPAQUE
    asm volatile("" : "+v"(tl));
    asm volatile("" : "+s"(tail_off));
#endif
    gk_lds_tail& T = *reinterpret_cast<gk_lds_tail*>(gk_lds + tail_off);
    const int ll = tl & 63;
    GK_STAMP(0);
    // ---- V_w = AHA_w P on the matrix cores ---------------------------------------------------------------------------------
    gk_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    {
      const char* bp = pl + (uint32_t)(w * (spw >> 3)) * GK_GRP + lane_const;
#pragma unroll
      for (int s = 0; s < spw; ++s) {
        {
          const float bv = *reinterpret_cast<const float*>(bp + (s >> 3) * (int)GK_GRP + (s & 7) * 64);
          if (s & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], bv, acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], bv, acc0, 0, 0, 0);
        }
        if ((s & 15) == 15) __builtin_amdgcn_sched_barrier(0);  // at most 16 operand reads ahead of their MFMAs (register budget)
      }
    }
    acc0 += acc1;
    // accumulator register t of lane (jq, jj) is D[4 jq + t][jj]: operand row 2 r + part (part 0 = re, 1 = im of AHA row r)
#pragma unroll
    for (int t = 0; t < 4; ++t) T.red[w][(4 * (int)jq + t) * 16 + (int)jj] = acc0[t];
    lds_barrier();
    GK_STAMP(1);
    // ---- the hand-off is wave 0's alone (its 64 lanes = the 8 x 8 elements of V_w): sum of the wave partials, partial dots,
    // publish, drain, arrive -- one wave's LDS traffic is ordered, so no workgroup barrier until the arrival's own
    float2 vnew = make_float2(0.f, 0.f);
    if (w == 0) {
      const int xr = tl >> 3, xk = tl & 7;
      float vre = 0.f, vim = 0.f;
#pragma unroll
      for (int ww = 0; ww < GK_WV; ++ww) {  // (g_re + i g_im)(p_re + i p_im), summed over the waves in a fixed order
        vre += T.red[ww][(2 * xr) * 16 + xk] - T.red[ww][(2 * xr + 1) * 16 + xk + 8];
        vim += T.red[ww][(2 * xr) * 16 + xk + 8] + T.red[ww][(2 * xr + 1) * 16 + xk];
      }
      vnew = make_float2(vre, vim);
      const float pr = *reinterpret_cast<const float*>(pl + gk_elem_off((uint32_t)(row0 + xr), (uint32_t)xk));
      const float pi = *reinterpret_cast<const float*>(pl + gk_elem_off((uint32_t)(row0 + xr), (uint32_t)xk + 8u));
      T.stage[xr * 16 + xk] = vre;
      T.stage[xr * 16 + 8 + xk] = vim;
      // <p, v> (first argument conjugated) and ||p||^2 over this workgroup's rows, per right-hand side
      T.dsum[(xr * 8 + xk) * 4 + 0] = (double)pr * (double)vre + (double)pi * (double)vim;
      T.dsum[(xr * 8 + xk) * 4 + 1] = (double)pr * (double)vim - (double)pi * (double)vre;
      T.dsum[(xr * 8 + xk) * 4 + 2] = (double)pr * (double)pr + (double)pi * (double)pi;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (tl < 32) {  // j = 4 k + c: partial dot c of right-hand side k over the 8 rows, fixed order
        const int k = tl >> 2, c = tl & 3;
        double s = 0.0;
        if (c < 3) {
#pragma unroll
          for (int r = 0; r < GK_ROWS; ++r) s += T.dsum[(r * 8 + k) * 4 + c];
        }
        gk_sc1_store_f64(D.dots + ((size_t)q * 256 + b) * 32 + tl, s);
      } else {  // the 8 rows of V: lane i -> (column half, row, piece): (re, im) PAIRS of two columns, operands of packed FMAs
        const int i = tl - 32, hh = i >> 4, r = (i >> 1) & 7, pp = i & 1;
        const float* sp = T.stage + r * 16 + hh * 4 + pp * 2;
        const f4 val = {sp[0], sp[8], sp[1], sp[9]};
        gk_sc1_store16(vx_rs, (uint32_t)q * vx_par + (uint32_t)(2 * hh + pp) * vx_piece + (uint32_t)(row0 + r) * 16u, val);
      }
      if (tl == 0) {  // every column had retired before this iteration: it is not applied (uniform: replicated scalars)
        int all = 1;
        for (int k = 0; k < GK_KB; ++k) all &= T.cs[k].done;
        T.alld = all;
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the storing wave drains its own stores
    }
    GK_STAMP(2);
    resident_sync* sy = sync;
    asm volatile("" : "+s"(sy));  // (the arrival's address arithmetic stays inside the iteration, like the rest)
    if (!grid_arrive_wait_t(sy->cnt, ++epoch, (unsigned)nwg, spin_limit, &T.flag, tl)) {
      alive = false;
      break;
    }
    GK_STAMP(3);
    if (T.alld) break;
    if (w == 0) T.vown[tl] = vnew;
    // ---- everything below is replicated: identical inputs, identical order in every workgroup ------------------------------
    // Requested at once: the 16 partial-dot slots this thread sums and its 8 slots of V (two 16-byte pieces each).
    f4 v01[NE], v23[NE];
    {
      // partial dots [256 slots][32] per parity (slots >= nwg stay zero): thread (bg = tid >> 5, j = tid & 31) sums slots
      // 16 bg .. 16 bg + 15 in that order
      const uint32_t voff = (uint32_t)(tl >> 5) * 16u * 256u + (uint32_t)(tl & 31) * 8u;
      const uint32_t qoff = (uint32_t)q * 256u * 256u;
      // The partial dots are wanted first and the 16 KiB of V behind them should stay in flight meanwhile -- but with all 32
      // loads written as builtins hipcc waits vmcnt(0) in front of the first addition (seen in the ISA).  So the 16 small loads
      // are issued by hand and waited for by hand: vmcnt(2 NE) = "everything but the V loads behind them" (loads return in order).
      double part[16];
      const unsigned long long da = (unsigned long long)D.dots;  // raw buffer descriptor: base, stride 0, no bound, dword format
      const u4 d_desc = {(unsigned)da, (unsigned)(da >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen sc1" : "=v"(part[i]) : "v"(voff), "s"(d_desc), "s"(qoff + (uint32_t)i * 256u) : "memory");
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        {  // rows >= N hold zeros in r, p and (never written) Vx: they stay zero
          v01[e] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(vx_rs, vx_lane, (uint32_t)q * vx_par + e * 4096u, 16));
          v23[e] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(vx_rs, vx_lane, (uint32_t)q * vx_par + vx_piece + e * 4096u, 16));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(%16)"
                   : "+v"(part[0]), "+v"(part[1]), "+v"(part[2]), "+v"(part[3]), "+v"(part[4]), "+v"(part[5]), "+v"(part[6]), "+v"(part[7]),
                     "+v"(part[8]), "+v"(part[9]), "+v"(part[10]), "+v"(part[11]), "+v"(part[12]), "+v"(part[13]), "+v"(part[14]),
                     "+v"(part[15])
                   : "n"(2 * NE));
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) s += part[i];
      T.dsum[tl] = s;  // [bg][j]
    }
    lds_barrier();
    if (tl < 32) {
      double s = 0.0;
#pragma unroll
      for (int g = 0; g < 16; ++g) s += T.dsum[g * 32 + tl];
      const double nim = __shfl(s, (ll & ~3) + 1, 64), pp = __shfl(s, (ll & ~3) + 2, 64);
      if ((tl & 3) == 0) {
        gk_col& c = T.cs[tl >> 2];
        if (c.active) {
          const double zeta = c.rr;
          const dcomplex alpha = dc_div({zeta, 0.0}, {s + (c.lambda > 0.f ? (double)c.lambda * pp : 0.0), nim});
          c.zeta = zeta;
          c.alpha_re = alpha.re;
          c.alpha_im = alpha.im;
          c.a_re = (float)alpha.re;
          c.a_im = (float)alpha.im;
        }
      }
    }
    lds_barrier();
    GK_STAMP(4);
    // ---- r -= alpha v (- lambda alpha p), ||r||^2 per right-hand side; own rows of x += alpha p -----------------------------
    // (branch-free: a column that does not take part gets alpha = 0, which leaves its r bit for bit -- fma(v, -0, r) = r --
    // and its sum is never looked at.)  alpha of this wave's four columns is wave-uniform: scalar registers.
    float2 na[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const gk_col& c = T.cs[4 * h + j];
      const bool on = uni(c.active) != 0;
      na[j] = make_float2(on ? -uni(c.a_re) : 0.f, on ? -uni(c.a_im) : 0.f);
    }
    const float lam0 = uni(T.cs[0].lambda);  // one lambda for all columns of a plan (rls_cgnr_init_batched)
    const bool anylam = lam0 > 0.f;
    if (tl < GK_ROWS * GK_KB) {
      const int xr2 = tl >> 3, xk2 = tl & 7;
      const gk_col& c = T.cs[xk2];
      if (c.active) {
        const float pr = *reinterpret_cast<const float*>(pl + gk_elem_off((uint32_t)(row0 + xr2), (uint32_t)xk2));
        const float pi = *reinterpret_cast<const float*>(pl + gk_elem_off((uint32_t)(row0 + xr2), (uint32_t)xk2 + 8u));
        T.xown[tl] = elem<float2>::fma(make_float2(pr, pi), make_float2(c.a_re, c.a_im), T.xown[tl]);
      }
    }
    double rrp[4] = {0.0, 0.0, 0.0, 0.0};
    auto update_r = [&](int e, bool with_p) {
      f4 pr4 = {0.f, 0.f, 0.f, 0.f}, pi4 = {0.f, 0.f, 0.f, 0.f};
      if (with_p) {  // the L2 term needs p
        pr4 = *reinterpret_cast<const f4*>(GK_PC(e, 0));
        pi4 = *reinterpret_cast<const f4*>(GK_PC(e, 1));
      }
      const float2 vj[4] = {make_float2(v01[e][0], v01[e][1]), make_float2(v01[e][2], v01[e][3]), make_float2(v23[e][0], v23[e][1]),
                            make_float2(v23[e][2], v23[e][3])};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float2 ri = elem<float2>::fma(vj[j], na[j], r2[e][j]);
        if (with_p) ri = elem<float2>::fma(elem<float2>::scale(lam0, make_float2(pr4[j], pi4[j])), na[j], ri);  // (-lambda p) alpha
        r2[e][j] = ri;
        rrp[j] = fma((double)ri.x, (double)ri.x, rrp[j]);
        rrp[j] = fma((double)ri.y, (double)ri.y, rrp[j]);
      }
    };
    if (anylam) {  // uniform
#pragma unroll
      for (int e = 0; e < NE; ++e)
        update_r(e, true);
    } else {
#pragma unroll
      for (int e = 0; e < NE; ++e)
        update_r(e, false);
    }
    GK_STAMP(5);
    // ||r||^2 of this wave's four columns over its rows: a wave sum; two waves' worth of rows x 4 (w >> 1) are added below
#pragma unroll
    for (int j = 0; j < 4; ++j) rrp[j] = wave_sum(rrp[j]);
    if (ll == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) T.rrw[w][j] = rrp[j];
    }
    lds_barrier();
    if (tl < GK_KB) {
      gk_col& c = T.cs[tl];
      if (c.active) {
        double rr = 0.0;
#pragma unroll
        for (int i = 0; i < GK_WV / 2; ++i) rr += T.rrw[2 * i + (tl >> 2)][tl & 3];  // the waves that own this column, in order
        const double beta = rr / c.zeta;
        c.rr = rr;
        c.beta = beta;
        c.b_f = (float)beta;
        c.iteration += 1;
        const float ratio = (float)(sqrt(rr) / c.z0);
        c.done = (ratio <= c.rel_tol) || (c.iteration >= c.max_iter);  // src/CGNR.jl:181-185
      }
    }
    lds_barrier();
    GK_STAMP(6);
    // ---- p = beta p + r in the operand panel (columns that took part); then the retirement flags of the next iteration ------
    float bf[4];
    bool on[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const gk_col& c = T.cs[4 * h + j];
      on[j] = uni(c.active) != 0;
      bf[j] = uni(c.b_f);
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      {
        f4 pr4 = *reinterpret_cast<const f4*>(GK_PC(e, 0));
        f4 pi4 = *reinterpret_cast<const f4*>(GK_PC(e, 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // uniform selects
          const float nr = bf[j] * pr4[j] + r2[e][j].x, ni = bf[j] * pi4[j] + r2[e][j].y;
          pr4[j] = on[j] ? nr : pr4[j];
          pi4[j] = on[j] ? ni : pi4[j];
        }
        *reinterpret_cast<f4*>(GK_PC(e, 0)) = pr4;
        *reinterpret_cast<f4*>(GK_PC(e, 1)) = pi4;
      }
    }
    lds_barrier();  // (every read of `active` lies before this barrier, the write below behind it)
    if (tl < GK_KB) T.cs[tl].active = !T.cs[tl].done;
    GK_STAMP(7);
  }
  gk_lds_tail& T = T0;
  GK_STAMP(10);
  if (alive) {
    // ---- gather x: every workgroup publishes its rows, one more barrier, workgroup 0 writes the caller's state -------------
    float2* Xx = reinterpret_cast<float2*>(D.Xx);
    if (tid < GK_ROWS * GK_KB) {
      const int xr = tid >> 3, xk = tid & 7;
      sc1_store_elem<float2>(Xx + (size_t)xk * N + row0 + xr, T.xown[tid]);  // [column][row]: workgroup 0's gather reads contiguously
      // V = AHA P of the last applied iteration: a workgroup's own rows are final
      if (xk < nrhs && it > 0) reinterpret_cast<float2*>(D.V)[(int64_t)xk * D.ldv + row0 + xr] = T.vown[tid];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    alive = grid_arrive_wait(sync->cnt, ++epoch, (unsigned)nwg, spin_limit, &T.flag);
  }
  if (!alive) {
    resident_give_up(sync, nullptr);
    return;  // x, r, p and the scalars are untouched: the call was a no-op (V's rows may hold products of the lost launch)
  }
  GK_STAMP(11);
  if (b != 0) return;
  {
    float2* Xo = reinterpret_cast<float2*>(D.X);
    float2* Ro = reinterpret_cast<float2*>(D.R);
    float2* Po = reinterpret_cast<float2*>(D.P);
    const float2* Xx = reinterpret_cast<const float2*>(D.Xx);
    // x: 8 gathered elements in flight per thread (one load, one store at a time was 32 dependent round trips: 48 us)
    const int total = N * nrhs;
    for (int i0 = 0; i0 < total; i0 += GK_NT * 8) {
      float2 xg[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = i0 + u * GK_NT + tid;
        const int ic = idx < total ? idx : 0;
        const int k = ic / N, n = ic - k * N;
        xg[u] = sc1_load_elem<float2>(Xx + (size_t)k * N + n);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = i0 + u * GK_NT + tid;
        if (idx < total) {
          const int k = idx / N, n = idx - k * N;
          Xo[(int64_t)k * D.ldv + n] = xg[u];
        }
      }
    }
    float* pp = D.Ppack;  // the streaming kernels' operand panel ([n][8 re | 8 im]) kept in step
    int nl2 = nl;
    asm volatile("" : "+v"(nl2));  // (addresses derived here, behind the loop -- not in front of it and carried across)
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int n = nl2 + 256 * e;
      if (FULL || n < N) {
        const f4 pr4 = *reinterpret_cast<const f4*>(GK_PC(e, 0));
        const f4 pi4 = *reinterpret_cast<const f4*>(GK_PC(e, 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = 4 * h + j;
          if (k < nrhs) {
            Ro[(int64_t)k * D.ldv + n] = r2[e][j];
            Po[(int64_t)k * D.ldv + n] = make_float2(pr4[j], pi4[j]);
          }
        }
        if (pp) {
          *reinterpret_cast<f4*>(pp + (size_t)n * 16 + 4 * h) = pr4;
          *reinterpret_cast<f4*>(pp + (size_t)n * 16 + 8 + 4 * h) = pi4;
        }
      }
    }
    if (tid < nrhs) {
      const gk_col& c = T.cs[tid];
      cgnr_scalars* s = D.sc + tid;
      s->rr = c.rr;
      s->zeta = c.zeta;
      s->alpha_re = c.alpha_re;
      s->alpha_im = c.alpha_im;
      s->beta_re = c.beta;
      s->beta_im = 0.0;
      s->iteration = c.iteration;
      s->done = c.done;
      s->pending = 0;
      s->cur = 0;
      s->fresh = 0;
    }
    if (tid == 0) sync->completed = 1u;
    GK_STAMP(12);
  }
}

// ---- batched FISTA on the explicit Gram matrix (src/FISTA.jl:141-189 per column, src/MultiThreading.jl:30-79) ---------------
// FISTA without gradient restart has NO global scalar on its critical path: theta follows a data-independent recursion and
// ||res|| only decides retirement.  So nothing is replicated here: workgroup b forms res = AHA y - x0, the prox and the NEXT
// extrapolated point for ITS 8 rows (wave 0, one element per lane), publishes those rows of y (512 bytes) and its 8 partial
// ||res||^2, and behind the ONE grid barrier of the iteration every workgroup gathers the new panel straight into LDS (two
// 16-byte pieces per slot, already in the panel's layout) while the 2048 partial norms are summed for the retirement flags.
// A column that retires in an iteration still gets its next extrapolated point into the plan-owned y (the streaming update
// leaves y alone then); nothing reads y of a retired column.
struct fk_col {
  double norm_x0, res_norm, rel;
  float rho, theta, theta_old, rel_tol, lambda;
  int iteration, max_iter, done, active, reg_kind, proj_kind;
};
struct fk_lds_tail {
  float red[GK_WV][256];
  double dsum[64 * 8];
  fk_col cs[GK_KB];
  float stage[GK_ROWS * 16];
  int flag, alld;
};

template <int NE, bool FULL>
__global__ __launch_bounds__(GK_NT) void fista_gramk_resident_kernel(rls_fgramk D, resident_sync* sync, int n_steps,
                                                                     unsigned spin_limit) {
  extern __shared__ __align__(16) char gk_lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  GK_STAMP(8);
  const int N = FULL ? 256 * NE : (int)D.N;
  const int nwg = gridDim.x, b = blockIdx.x, row0 = b * GK_ROWS;
  constexpr int spw = 8 * NE, npad = 256 * NE;
  char* pl = gk_lds;
  const uint32_t tail0 = (uint32_t)(npad / 32) * GK_GRP;
  fk_lds_tail& T = *reinterpret_cast<fk_lds_tail*>(gk_lds + tail0);
  const int nrhs = D.nrhs;
  const int h = w & 1;
  const int nl = (w >> 1) * 64 + lane;

  // ---- state in: scalars, AHA rows (registers), y (LDS panel), own rows of x, xold, x0, res (wave 0: lane = (row, column)) ----
  if (tid < GK_KB) {
    fk_col& c = T.cs[tid];
    const bool real_col = tid < nrhs;
    const fista_scalars* s = D.sc + (real_col ? tid : 0);
    c.norm_x0 = real_col ? s->norm_x0 : 1.0;
    c.res_norm = real_col ? s->res_norm : 0.0;
    c.rel = real_col ? s->rel_res_norm : 0.0;
    c.rho = real_col ? s->rho : 0.f;
    c.theta = real_col ? s->theta : 1.f;
    c.theta_old = real_col ? s->theta_old : 1.f;
    c.rel_tol = real_col ? s->rel_tol : 0.f;
    c.lambda = real_col ? s->lambda : 0.f;
    c.iteration = real_col ? s->iteration : 0;
    c.max_iter = real_col ? s->max_iter : 0;
    c.done = real_col ? s->done : 1;  // padding columns never take part
    c.active = real_col ? !s->done : 0;
    c.reg_kind = real_col ? s->reg_kind : RLS_REG_NONE;
    c.proj_kind = real_col ? s->proj_kind : RLS_PROJ_NONE;
    const unsigned long long alld = __ballot(c.done != 0);
    if (tid == 0) T.alld = (alld & 0xffull) == 0xffull;
  }
  const uint32_t pc_lo = (uint32_t)(2 * (w >> 1) + (lane >> 5)) * GK_GRP + (uint32_t)h * GK_REG + (uint32_t)(lane & 31) * 16u;
  const uint32_t pc_hi = pc_lo + (NE > 4 ? 4u * 8u * GK_GRP : 0u);
  {
    const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(D.y), 0, 0xffffffff, 0x00020000);
    const uint32_t ldvb = (uint32_t)D.ldv * 8u;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int n = nl + 256 * e;
      const bool rowok = FULL || n < N;
      const uint32_t voff = (uint32_t)(FULL ? nl : (rowok ? n : 0)) * 8u;
      f4 yr4, yi4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 4 * h + j;
        const bool ok = rowok && k < nrhs;
        const uint32_t so = (uint32_t)(k < nrhs ? k : 0) * ldvb + (FULL ? (uint32_t)e * 2048u : 0u);
        const float2 yv = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(y_rs, voff, so, 0));
        yr4[j] = ok ? yv.x : 0.f;
        yi4[j] = ok ? yv.y : 0.f;
      }
      *reinterpret_cast<f4*>(GK_PC(e, 0)) = yr4;
      *reinterpret_cast<f4*>(GK_PC(e, 1)) = yi4;
    }
  }
  // wave 0, lane (xr = lane >> 3, xk = lane & 7): element (row0 + xr, column xk).  state.x of a column is buf[iteration & 1]
  // seen from the update that produced it: the update of iteration count `it` wrote (it & 1) ? b0 : b1.
  float2 xcur = make_float2(0.f, 0.f), xprev = xcur, x0own = xcur, resown = xcur;
  if (w == 0) {
    const int xr = lane >> 3, xk = lane & 7;
    if (xk < nrhs) {
      const int itk = D.sc[xk].iteration;
      const int64_t at = (int64_t)xk * D.ldv + row0 + xr;
      const float2* cur = reinterpret_cast<const float2*>(((itk - 1) & 1) ? D.b0 : D.b1);
      const float2* prv = reinterpret_cast<const float2*>(((itk - 1) & 1) ? D.b1 : D.b0);
      xcur = cur[at];
      xprev = prv[at];
      x0own = reinterpret_cast<const float2*>(D.x0)[at];
      resown = reinterpret_cast<const float2*>(D.res)[at];
    }
  }
  float ga[spw];
  {
    const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(D.G), 0, 0xffffffff, 0x00020000);
    const uint32_t colb = (uint32_t)D.ldg * 8u;
    const uint32_t voff = (uint32_t)row0 * 8u + (uint32_t)(lane & 15) * 4u + (uint32_t)(lane >> 4) * colb;
#pragma unroll
    for (int s = 0; s < spw; ++s) {
      const int c0 = 4 * (w * spw + s);
      const bool ok = FULL || c0 < N;
      const float g = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_rs, voff, (uint32_t)(ok ? c0 : 0) * colb, 0));
      ga[s] = ok ? g : 0.f;
    }
  }
  __syncthreads();
  GK_STAMP(9);
  const __amdgpu_buffer_rsrc_t yx_rs = sc1_rsrc(D.Yx);
  // exchanged rows of y: [parity][part (re | im)][column half][row][16 bytes] -- a thread's gather is the panel piece it stores
  const uint32_t yx_piece = (uint32_t)npad * 16u, yx_par = 4u * yx_piece;
  const uint32_t yx_lane = (uint32_t)h * yx_piece + (uint32_t)nl * 16u;
  const uint32_t jq = (uint32_t)(lane >> 4), jj = (uint32_t)(lane & 15);
  const uint32_t lane_const = (2u * (jj >> 3) + ((jj >> 2) & 1u)) * GK_REG + jq * 16u + (jj & 3u) * 4u;
  unsigned epoch = 0;
  bool alive = true;
  for (int it = 0; it < n_steps; ++it) {
    const int q = it & 1;
    GK_STAMP(0);
    // ---- (AHA y) rows of this workgroup on the matrix cores ------------------------------------------------------------------
    gk_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    {
      const char* bp = pl + (uint32_t)(w * (spw >> 3)) * GK_GRP + lane_const;
#pragma unroll
      for (int s = 0; s < spw; ++s) {
        const float bv = *reinterpret_cast<const float*>(bp + (s >> 3) * (int)GK_GRP + (s & 7) * 64);
        if (s & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], bv, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], bv, acc0, 0, 0, 0);
        if ((s & 15) == 15) __builtin_amdgcn_sched_barrier(0);
      }
    }
    acc0 += acc1;
#pragma unroll
    for (int t = 0; t < 4; ++t) T.red[w][(4 * (int)jq + t) * 16 + (int)jj] = acc0[t];
    lds_barrier();
    GK_STAMP(1);
    // every column had retired before this iteration (replicated scalars: every workgroup leaves in the same iteration; wave 0
    // set the flag ahead of ITS products -- the barrier above orders it -- and the products just formed are dropped)
    if (T.alld) break;
    // ---- wave 0: the whole update of this workgroup's 8 x 8 elements, then the hand-off ----------------------------------------
    if (w == 0) {
      const int xr = lane >> 3, xk = lane & 7;
      float vre = 0.f, vim = 0.f;
#pragma unroll
      for (int ww = 0; ww < GK_WV; ++ww) {
        vre += T.red[ww][(2 * xr) * 16 + xk] - T.red[ww][(2 * xr + 1) * 16 + xk + 8];
        vim += T.red[ww][(2 * xr) * 16 + xk + 8] + T.red[ww][(2 * xr + 1) * 16 + xk];
      }
      const fk_col& c = T.cs[xk];
      const bool on = c.active != 0;
      const float yr = *reinterpret_cast<const float*>(pl + gk_elem_off((uint32_t)(row0 + xr), (uint32_t)xk));
      const float yi = *reinterpret_cast<const float*>(pl + gk_elem_off((uint32_t)(row0 + xr), (uint32_t)xk + 8u));
      const float2 ri = make_float2(vre - x0own.x, vim - x0own.y);                                   // res .-= x0       :153
      float2 xi = elem<float2>::sub(make_float2(yr, yi), elem<float2>::scale(c.rho, ri));            // x .-= rho .* res :154
      xi = fista_proj_elem<float2>(fista_prox_elem<float2>(xi, c.reg_kind, c.rho * c.lambda), c.proj_kind);  //         :164
      const float theta_old = c.theta;                                                               // :179
      const float theta = (1.f + sqrtf(1.f + 4.f * theta_old * theta_old)) / 2.f;                    // :180
      const float c1 = (1.f - theta_old) / theta, c2 = (theta_old - 1.f) / theta + 1.f;
      const float2 yn = elem<float2>::add(elem<float2>::scale(c1, xcur), elem<float2>::scale(c2, xi));  // :147-148 of the next one
      double rn = 0.0;
      float2 yout = make_float2(yr, yi);
      if (on) {
        rn = (double)ri.x * (double)ri.x + (double)ri.y * (double)ri.y;
        xprev = xcur;
        xcur = xi;
        resown = ri;
        yout = yn;
      }
      T.stage[xr * 16 + xk] = yout.x;
      T.stage[xr * 16 + 8 + xk] = yout.y;
      T.dsum[xr * 8 + xk] = rn;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (lane < GK_KB) {  // ||res||^2 of column `lane` over the 8 rows, fixed order
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < GK_ROWS; ++r) s += T.dsum[r * 8 + lane];
        gk_sc1_store_f64(D.dots + ((size_t)q * 256 + b) * 8 + lane, s);
      } else if (lane >= 32) {  // the 8 rows of y as panel pieces: (part, column half, row)
        const int i = lane - 32, hh = i >> 4, r = (i >> 1) & 7, part = i & 1;
        const float* sp = T.stage + r * 16 + part * 8 + hh * 4;
        const f4 val = {sp[0], sp[1], sp[2], sp[3]};
        gk_sc1_store16(yx_rs, (uint32_t)q * yx_par + (uint32_t)(2 * part + hh) * yx_piece + (uint32_t)(row0 + r) * 16u, val);
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the storing wave drains its own stores
    }
    GK_STAMP(2);
    if (!grid_arrive_wait(sync->cnt, ++epoch, (unsigned)nwg, spin_limit, &T.flag)) {
      alive = false;
      break;
    }
    GK_STAMP(3);
    // ---- behind the barrier: partial norms (wanted first) and the new panel, all requested at once -----------------------------
    {
      // partial norms [256 slots][8] per parity (slots >= nwg stay zero): thread (bg = tid >> 3, j = tid & 7) sums slots 4 bg .. 4 bg + 3
      const double* dp = D.dots + ((size_t)q * 256 + (size_t)(tid >> 3) * 4) * 8 + (tid & 7);
      double part[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        part[i] = __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(dp + i * 8), __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT));
      f4 yre[NE], yim[NE];
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        yre[e] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(yx_rs, yx_lane, (uint32_t)q * yx_par + e * 4096u, 16));
        yim[e] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(yx_rs, yx_lane, (uint32_t)q * yx_par + 2u * yx_piece + e * 4096u, 16));
      }
      T.dsum[tid] = (part[0] + part[1]) + (part[2] + part[3]);  // [bg][j]
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        *reinterpret_cast<f4*>(GK_PC(e, 0)) = yre[e];
        *reinterpret_cast<f4*>(GK_PC(e, 1)) = yim[e];
      }
    }
    lds_barrier();
    GK_STAMP(4);
    if (w == 0) {  // lane (g8 = lane >> 3, j = lane & 7): groups 8 g8 .. 8 g8 + 7, then a fixed butterfly over g8
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += T.dsum[((lane >> 3) * 8 + i) * 8 + (lane & 7)];
      s += dpp_d(s, 0x128);   // lane ^ 8: row_ror:8 inside the row of 16; lane ^ 16, lane ^ 32: permlane swaps (the same pairs: the same bits)
      s = pair_sum16(s);
      s = pair_sum32(s);
      int done = 1;
      if (lane < GK_KB) {
        fk_col& c = T.cs[lane];
        if (c.active) {
          const float theta_old = c.theta;
          c.theta = (1.f + sqrtf(1.f + 4.f * theta_old * theta_old)) / 2.f;
          c.theta_old = theta_old;
          c.res_norm = sqrt(s);
          const float rel = (float)(c.res_norm / c.norm_x0);                  // :156
          c.rel = (double)rel;
          c.iteration += 1;
          c.done = (rel < c.rel_tol) || (c.iteration >= c.max_iter);          // :187-189
          c.active = !c.done;
        }
        done = c.done;
      }
      const unsigned long long nd = __ballot(done == 0);
      if (lane == 0) T.alld = nd == 0ull;
    }
    // no barrier here: the scalars are wave 0's own business until its update phase, `alld` is read behind the next
    // iteration's product barrier -- the other seven waves are already multiplying
    GK_STAMP(5);
  }
  GK_STAMP(10);
  if (alive) {
    // ---- gather: every workgroup publishes its rows of x, xold and res; one more barrier; workgroup 0 writes the caller's state
    if (w == 0) {
      float2* Xx = reinterpret_cast<float2*>(D.Xx);
      const int xr = lane >> 3, xk = lane & 7;
      const size_t at = (size_t)xk * N + row0 + xr, plane = (size_t)N * GK_KB;  // [array][column][row]: contiguous for the gather
      sc1_store_elem<float2>(Xx + at, xcur);
      sc1_store_elem<float2>(Xx + plane + at, xprev);
      sc1_store_elem<float2>(Xx + 2 * plane + at, resown);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    alive = grid_arrive_wait(sync->cnt, ++epoch, (unsigned)nwg, spin_limit, &T.flag);
  }
  if (!alive) {
    resident_give_up(sync, nullptr);
    return;  // the plan's vectors and scalars are untouched: the call was a no-op
  }
  GK_STAMP(11);
  if (b != 0) return;
  {
    // [array][column][row] planes: 16 bytes (two rows of a column) per load and per store, contiguous across the lanes; all
    // 48 loads of a thread in flight.  (Row-major planes read with a 64-byte lane stride cost 13.5 us here: every wave load
    // touched 64 cache lines.)
    const __amdgpu_buffer_rsrc_t xx_rs = sc1_rsrc(D.Xx);
    const int half_n = N >> 1, items = GK_KB * half_n;
    const uint32_t plane_b = (uint32_t)N * GK_KB * 8u;
    f4 xg[3][16];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int idx = u * GK_NT + tid;
        xg[a][u] = sc1_load16(xx_rs, (uint32_t)a * plane_b + (uint32_t)(idx < items ? idx : 0) * 16u);
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int idx = u * GK_NT + tid;
        const int k = FULL ? idx >> (7 + (NE == 8 ? 3 : NE == 4 ? 2 : NE == 2 ? 1 : 0)) : idx / half_n, i = idx - k * half_n;
        if (idx < items && k < nrhs) {
          const int odd = (T.cs[k].iteration - 1) & 1;  // where the column's last update wrote state.x
          float2* dst = a == 2 ? reinterpret_cast<float2*>(D.res)
                               : reinterpret_cast<float2*>((a == 0) == (odd != 0) ? D.b0 : D.b1);
          *reinterpret_cast<f4*>(dst + (int64_t)k * D.ldv + 2 * i) = xg[a][u];  // (host: ldv even, 16-byte aligned vectors)
        }
      }
    }
    GK_STAMP(13);
    float2* Yo = reinterpret_cast<float2*>(D.y);
    float* pp = D.Ypack;  // the streaming kernels' operand panel ([n][8 re | 8 im]) kept in step
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int n = nl + 256 * e;
      if (FULL || n < N) {
        const f4 yr4 = *reinterpret_cast<const f4*>(GK_PC(e, 0));
        const f4 yi4 = *reinterpret_cast<const f4*>(GK_PC(e, 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = 4 * h + j;
          if (k < nrhs) Yo[(int64_t)k * D.ldv + n] = make_float2(yr4[j], yi4[j]);
        }
        if (pp) {
          *reinterpret_cast<f4*>(pp + (size_t)n * 16 + 4 * h) = yr4;
          *reinterpret_cast<f4*>(pp + (size_t)n * 16 + 8 + 4 * h) = yi4;
        }
      }
    }
    GK_STAMP(14);
    if (tid < nrhs) {
      const fk_col& c = T.cs[tid];
      fista_scalars* s = D.sc + tid;
      s->res_norm = c.res_norm;
      s->rel_res_norm = c.rel;
      s->theta = c.theta;
      s->theta_old = c.theta_old;
      s->iteration = c.iteration;
      s->done = c.done;
    }
    if (tid == 0) sync->completed = 1u;
    GK_STAMP(12);
  }
}

static size_t gk_lds_bytes(int64_t N) { return gk_panel_bytes(N) + sizeof(gk_lds_tail); }
static size_t fk_lds_bytes(int64_t N) { return gk_panel_bytes(N) + sizeof(fk_lds_tail); }

// KIND 0: CGNR, 1: FISTA
template <int KIND, int NE, bool FULL>
static const void* gk_kernel() {
  if constexpr (KIND == 0) return reinterpret_cast<const void*>(&cgnr_gramk_resident_kernel<NE, FULL>);
  else return reinterpret_cast<const void*>(&fista_gramk_resident_kernel<NE, FULL>);
}
template <int KIND>
static size_t gk_kind_lds(int64_t N) { return KIND == 0 ? gk_lds_bytes(N) : fk_lds_bytes(N); }

template <int KIND, int NE>
static void gk_allow_lds() {
  (void)hipFuncSetAttribute(gk_kernel<KIND, NE, true>(), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gk_kind_lds<KIND>(256 * NE));
  (void)hipFuncSetAttribute(gk_kernel<KIND, NE, false>(), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gk_kind_lds<KIND>(256 * NE));
}
template <int KIND, int NE>
static hipError_t gk_occupancy(int* blocks, bool full) {
  if constexpr (KIND == 0)
    return full ? hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, cgnr_gramk_resident_kernel<NE, true>, GK_NT, gk_lds_bytes(256 * NE))
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, cgnr_gramk_resident_kernel<NE, false>, GK_NT, gk_lds_bytes(256 * NE));
  else
    return full ? hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, fista_gramk_resident_kernel<NE, true>, GK_NT, fk_lds_bytes(256 * NE))
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, fista_gramk_resident_kernel<NE, false>, GK_NT, fk_lds_bytes(256 * NE));
}
template <int NE>
static void gk_launch(rls_ctx* ctx, const rls_gramk& D, void* sync, int n_steps, unsigned spin_limit) {
  const int nwg = (int)(D.N / GK_ROWS);
  const size_t lds = gk_lds_bytes(D.N);
  if (D.N == 256 * NE)
    hipLaunchKernelGGL((cgnr_gramk_resident_kernel<NE, true>), dim3(nwg), dim3(GK_NT), lds, ctx->stream, D, (resident_sync*)sync, n_steps,
                       spin_limit);
  else
    hipLaunchKernelGGL((cgnr_gramk_resident_kernel<NE, false>), dim3(nwg), dim3(GK_NT), lds, ctx->stream, D, (resident_sync*)sync, n_steps,
                       spin_limit);
}
template <int NE>
static void fk_launch(rls_ctx* ctx, const rls_fgramk& D, void* sync, int n_steps, unsigned spin_limit) {
  const int nwg = (int)(D.N / GK_ROWS);
  const size_t lds = fk_lds_bytes(D.N);
  if (D.N == 256 * NE)
    hipLaunchKernelGGL((fista_gramk_resident_kernel<NE, true>), dim3(nwg), dim3(GK_NT), lds, ctx->stream, D, (resident_sync*)sync, n_steps,
                       spin_limit);
  else
    hipLaunchKernelGGL((fista_gramk_resident_kernel<NE, false>), dim3(nwg), dim3(GK_NT), lds, ctx->stream, D, (resident_sync*)sync, n_steps,
                       spin_limit);
}

// shapes both kernels take, and whether one workgroup of kernel KIND fits a CU
template <int KIND>
static bool gk_resident_ok(rls_ctx* ctx, int32_t dtype, int64_t N, int nrhs, const void* G, int64_t ldg) {
  if (dtype != RLS_C32 || !G || nrhs < 1 || nrhs > GK_KB || N < 16 || N % 16 || N > GK_NMAX) return false;
  if (((uintptr_t)G & 15) || (ldg % 2) || ldg * 8 * N >= (int64_t)0xffffffffll) return false;
  const int nwg = (int)(N / GK_ROWS);
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess) return false;
  if (!(nwg <= cus && nwg <= 256)) return false;
  static rls_device_once attr_once;
  if (auto once_ = attr_once.first(ctx->device)) {
    gk_allow_lds<KIND, 1>();
    gk_allow_lds<KIND, 2>();
    gk_allow_lds<KIND, 4>();
    gk_allow_lds<KIND, 8>();
  }
  int blocks = 0;
  const int ne = gk_ne(N);
  const bool full = N == 256 * ne;
  const hipError_t e = ne == 1 ? gk_occupancy<KIND, 1>(&blocks, full) : ne == 2 ? gk_occupancy<KIND, 2>(&blocks, full)
                       : ne == 4 ? gk_occupancy<KIND, 4>(&blocks, full) : gk_occupancy<KIND, 8>(&blocks, full);
  (void)hipGetLastError();
  return e == hipSuccess && blocks >= 1;
}

}  // namespace

// exchange scratch behind the plan: V panel [2 parities][2 column halves][rows][32 bytes], gathered x [N][8] complex, partial
// dots [2][256 slots][32] f64
void rls_gramk_sizes(int64_t N, size_t* vx_bytes, size_t* xx_bytes, size_t* dots_bytes) {
  *vx_bytes = (size_t)2 * 2 * gk_ne(N) * 256 * 32;  // zero-filled by the plan: rows >= N are read (as zeros), never written
  *xx_bytes = (size_t)N * GK_KB * sizeof(float2);
  *dots_bytes = (size_t)2 * 256 * 32 * sizeof(double);  // zero-filled by the plan: slots of absent workgroups add 0.0
}

bool rls_gramk_resident_ok(rls_ctx* ctx, int32_t dtype, int64_t N, int nrhs, const void* G, int64_t ldg) {
  return gk_resident_ok<0>(ctx, dtype, N, nrhs, G, ldg);
}

int32_t rls_gramk_resident_launch(rls_ctx* ctx, const rls_gramk& D, void* sync, int n_steps, unsigned spin_limit) {
  const int ne = gk_ne(D.N);
  if (ne == 1) gk_launch<1>(ctx, D, sync, n_steps, spin_limit);
  else if (ne == 2) gk_launch<2>(ctx, D, sync, n_steps, spin_limit);
  else if (ne == 4) gk_launch<4>(ctx, D, sync, n_steps, spin_limit);
  else gk_launch<8>(ctx, D, sync, n_steps, spin_limit);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

// FISTA: exchanged rows of y [2 parities][re | im][2 column halves][rows][16 bytes], gathered x / xold / res [3][N][8] complex,
// partial norms [2][256 slots][8] f64
void rls_fgramk_sizes(int64_t N, size_t* yx_bytes, size_t* xx_bytes, size_t* dots_bytes) {
  *yx_bytes = (size_t)2 * 4 * gk_ne(N) * 256 * 16;  // zero-filled by the plan: rows >= N are read (as zeros), never written
  *xx_bytes = (size_t)3 * N * GK_KB * sizeof(float2);
  *dots_bytes = (size_t)2 * 256 * 8 * sizeof(double);  // zero-filled by the plan: slots of absent workgroups add 0.0
}

bool rls_fgramk_resident_ok(rls_ctx* ctx, int32_t dtype, int64_t N, int nrhs, const void* G, int64_t ldg) {
  return gk_resident_ok<1>(ctx, dtype, N, nrhs, G, ldg);  // (the plan also checks its vectors: 16-byte aligned, ldv even)
}

int32_t rls_fgramk_resident_launch(rls_ctx* ctx, const rls_fgramk& D, void* sync, int n_steps, unsigned spin_limit) {
  const int ne = gk_ne(D.N);
  if (ne == 1) fk_launch<1>(ctx, D, sync, n_steps, spin_limit);
  else if (ne == 2) fk_launch<2>(ctx, D, sync, n_steps, spin_limit);
  else if (ne == 4) fk_launch<4>(ctx, D, sync, n_steps, spin_limit);
  else fk_launch<8>(ctx, D, sync, n_steps, spin_limit);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}
