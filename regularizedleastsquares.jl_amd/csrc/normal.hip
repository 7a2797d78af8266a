// Fused normal operator  v = A^H (A p)  in ONE pass over A.
//
// The reference evaluates mul!(v, AHA, p) (src/CGNR.jl:151, src/FISTA.jl:152, cg! from
// src/ADMM.jl:244) as two dependent GEMVs, t = A p then v = A^H t, each streaming all of A from
// memory.  On MI355X a CU's register file (512 KiB) is larger than its share of A at the headline
// shape (64 MiB / 256 CUs = 256 KiB), so a workgroup that owns a ROW slab A_w (G*V rows x all N
// columns) can
//     1. load the slab once, 16 bytes per lane per load, K loads per lane kept in VGPRs,
//     2. form its rows of t = A_w p completely (no other workgroup contributes to those rows),
//     3. form its contribution A_w^H t_w to every column of v from the SAME registers,
// and only the N-vector partials (one per workgroup) go back to memory.  A second small kernel sums
// the partials in a fixed order (deterministic, no atomics).  HBM traffic per apply: M*N*s for A
// once + 2 * nwg*N*s for the partials, instead of 2*M*N*s.
//
// Lane layout inside a wave: g = lane % G picks the 16-byte row chunk, s = lane / G one of the
// 64/G column slots; wave w of WV and load k cover column (k*WV + w)*(64/G) + s.
#include "rls_common.hpp"
#include "resident_sync.hpp"

// Diagnostic build only (-DRLS_STAMPS): wall-clock stamps (100 MHz s_memrealtime) of one workgroup's
// phases, written to a buffer nothing else reads.  Never compiled into the shipped library.
#ifdef RLS_STAMPS
__device__ unsigned long long g_stamps[16 * 8];
#define STAMP(slot)                                                                         \
  do {                                                                                      \
    if (threadIdx.x == 0 && (blockIdx.x % 37) == 5 && blockIdx.x / 37 < 8)                  \
      g_stamps[(blockIdx.x / 37) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime();         \
  } while (0)
// row 7: the reduce kernel's workgroup 0 (slots 0..3) and, copied at the next K_A start, the one before (4..7)
#define STAMP_R(slot)                                                                    \
  do {                                                                                      \
    if (threadIdx.x == 0 && blockIdx.x == 0) g_stamps[7 * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define STAMP_R_KEEP()                                                                    \
  do {                                                                                      \
    if (threadIdx.x == 0 && blockIdx.x == 5)                                                \
      for (int q = 0; q < 4; ++q) g_stamps[7 * 16 + 4 + q] = g_stamps[7 * 16 + q];          \
  } while (0)
#else
#define STAMP(slot) \
  do {              \
  } while (0)
#define STAMP_R(slot) \
  do {                \
  } while (0)
#define STAMP_R_KEEP() \
  do {                 \
  } while (0)
#endif

namespace {

constexpr int FIN_THREADS = 1024;  // the single-workgroup finish kernel

// blocks b and b+8 are observed to share an XCD (speed only, never correctness): with 64-byte
// row chunks (G = 4) two neighbouring row blocks split every 128-byte line, so give them to
// blocks that share an L2.
// pair: 0 = no, 1 = every block (the count is a multiple of 16), n > 1 = the first n blocks (n a multiple of 16; a ragged count's
// last few blocks stay where they are -- without this a 516-block launch had every line fetched into two L2s: 62 us against 33).
__device__ static inline int64_t row_block_of(int64_t b, int pair) {
  if (!pair || (pair > 1 && b >= pair)) return b;
  return (b / 16) * 16 + (b % 8) * 2 + ((b / 8) % 2);
}
// the `pair` argument of the slab launches over nwg row blocks of G-lane row chunks
static inline int slab_pairing(int G, int nwg) { return G != 4 ? 0 : (nwg % 16 == 0 ? 1 : (nwg / 16) * 16); }

// sum over the G = 2/4/8 consecutive lanes of a group with DPP (full-rate VALU) instead of
// ds_bpermute shuffles: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror
template <int G>
__device__ static inline float group_sum(float v) {
  if constexpr (G >= 2) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  if constexpr (G >= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  if constexpr (G >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}

// ---- the slab core, shared by the plain normal operator and the CGNR pipeline ---------------
// G lanes per row chunk, K loads per lane, WV waves per workgroup.  WV = 8 gives every lane 256
// VGPRs (2 waves per SIMD), so a K = 32 slab (128 VGPRs) leaves room for the rest of the kernel;
// at WV = 16 / K = 16 the same slab sits against a 128-VGPR ceiling and the kernel spills.
template <typename E, int G, int K, int WV>
struct slab_cfg {
  static constexpr int NV = elem<E>::vec;
  static constexpr int S = 64 / G;
  static constexpr int NT = WV * 64;
  static constexpr int CPR = WV * S;
  static constexpr int NMAX = K * CPR;
  static constexpr int EPT = NMAX >= NT ? NMAX / NT : 1;  // vector elements per thread
};

// resident kernels: may the slab be re-arranged into the column-owner layout (owner_cfg below)?
template <typename E, int G, int K, int WV>
constexpr bool owner_cfg_ok() {
  using C = slab_cfg<E, G, K, WV>;
  return (K % 16 == 0) && (C::EPT % C::NV == 0) && (C::EPT / C::NV == K / 16);
}

// LDS image of one workgroup (dynamic LDS: > 64 KiB).  xg is the exchange area of the second product:
// every lane stores its two-row (four-row for real) contribution to a column, one padded plane per
// row chunk g (pad = 64 B so the G planes start on different banks: 2-way at worst on the store,
// free; the column sums then read consecutive 8-byte words, conflict-free).
// ComplexF32 with 64-byte row chunks (G = 4, N in (2048, 4096]): four planes of 4096 values + the input vector are 256 bytes MORE
// than a CU's 160 KiB -- the shape class ran on the two-GEMV path until round 5.  Its lanes g and g ^ 1 are neighbours, so one
// full-rate DPP add per component combines them before the store: two planes (slab_planes), half the LDS traffic of that phase.
// The 128-byte layouts (G = 8) do the same -- four planes: the phase is LDS-bound, measured -0.4 us per CGNR iteration on the
// pipeline at 4096 x 2048 ComplexF32, -0.8 us per plain apply, -0.6 us per iteration of the Float32 resident kernel at N = 2048;
// Float32 with G = 4 keeps its four planes (two were 0.5 us SLOWER per apply at 8192 x 4096: one float per lane is too little to pair).
template <typename E, int G>
__host__ __device__ constexpr int slab_planes() {
  return G == 8 ? 4 : (elem<E>::cplx && G == 4) ? 2 : G;
}
template <typename E, int G, int K, int WV>
struct slab_lds {
  static constexpr int PAD = 64 / (int)sizeof(E);
  static constexpr int XP = slab_planes<E, G>();
  E xg[XP][slab_cfg<E, G, K, WV>::NMAX + PAD];
  E xs[slab_cfg<E, G, K, WV>::NMAX];  // the GEMV input vector
  E part[WV][G][elem<E>::vec];
  E tw[G * elem<E>::vec];
  double red[48];
};

// the second product's contribution of this lane to column `col` goes to the exchange planes (see slab_lds / slab_planes)
template <typename E, int G, int K, int WV>
__device__ static __forceinline__ void slab_xg_store(slab_lds<E, G, K, WV>& L, int g, int col, E q) {
  if constexpr (slab_planes<E, G>() < G) {
    const float re = elem<E>::re(q) + dpp_f(elem<E>::re(q), 0xB1);  // quad_perm [1,0,3,2]: lane ^ 1
    float im = 0.f;
    if constexpr (elem<E>::cplx) im = elem<E>::im(q) + dpp_f(elem<E>::im(q), 0xB1);
    if ((g & 1) == 0) L.xg[g >> 1][col] = elem<E>::make(re, im);
  } else {
    L.xg[g][col] = q;
  }
}
template <typename E, int G, int K, int WV>
__device__ static __forceinline__ E slab_xg_sum(const slab_lds<E, G, K, WV>& L, int c) {
  E sum = L.xg[0][c];
#pragma unroll
  for (int gg = 1; gg < slab_planes<E, G>(); ++gg) sum = elem<E>::add(sum, L.xg[gg][c]);
  return sum;
}

// Issue every load of the slab before anything waits.  Addresses are a wave-uniform 64-bit base per
// load (SGPRs) plus ONE 32-bit lane offset shared by all K loads, so the K addresses cost a single
// VGPR instead of 2K (the slab itself already takes 4K of the 128 available).  Ragged shapes clamp
// the lane offset per load (always-valid addresses, never a branch); the dead lanes are zeroed in
// slab_finish.
template <typename E, int G, int K, int WV, bool FULL>
__device__ static inline void slab_load(chunk<E, elem<E>::vec> (&a)[K], const E* __restrict__ A, int64_t lda,
                                        int64_t Mc, int64_t N, int pair) {
  using C = slab_cfg<E, G, K, WV>;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = lane % G, slot = w * C::S + lane / G;
  const int64_t chunk_id = row_block_of(blockIdx.x, pair) * G + g;  // pair only if gridDim.x % 16 == 0
  const uint32_t row_off = (uint32_t)((chunk_id < Mc ? chunk_id : (Mc - 1)) * 16);
  const uint32_t col_b = (uint32_t)(lda * (int64_t)sizeof(E));  // host guarantees 128 * col_b < 2^32
  const char* base = reinterpret_cast<const char*>(A);
  if constexpr (FULL) {  // N == NMAX and every row chunk valid: no clamps at all
    const uint32_t off = row_off + (uint32_t)slot * col_b;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const char* bk = base + (int64_t)(k * C::CPR) * (lda * (int64_t)sizeof(E));
      a[k] = load_chunk<E, C::NV>(reinterpret_cast<const E*>(bk + off));
    }
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int last = (int)N - 1 - k * C::CPR;  // last valid slot of this round (may be negative)
      const int kc = last >= 0 ? k : 0;          // rounds entirely past N re-read round 0 (zeroed later)
      // rounds past N re-read round 0, whose own last valid slot is min(CPR, N) - 1 (an unclamped slot read
      // up to CPR - N columns past the end of A when N < CPR: harmless until the page behind A is unmapped)
      const int last0 = (int)N - 1 < C::CPR - 1 ? (int)N - 1 : C::CPR - 1;
      const int sc = last >= 0 ? (slot < last ? slot : last) : (slot < last0 ? slot : last0);
      const char* bk = base + (int64_t)(kc * C::CPR) * (lda * (int64_t)sizeof(E));
      a[k] = load_chunk<E, C::NV>(reinterpret_cast<const E*>(bk + (row_off + (uint32_t)sc * col_b)));
    }
  }
}

// with L.xs holding the input vector (zero beyond N): t_w = A_w xs, partial v = A_w^H t_w -> slab row
template <typename E, int G, int K, int WV, bool FULL, bool SC1 = false, bool TT = false>
__device__ static inline double slab_finish(chunk<E, elem<E>::vec> (&a)[K], slab_lds<E, G, K, WV>& L, E* __restrict__ slab,
                                            int64_t Mc, int64_t N, int pair) {
  using C = slab_cfg<E, G, K, WV>;
  constexpr int NV = C::NV;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane % G, s = lane / G, slot = w * C::S + s;
  const bool row_ok = row_block_of(blockIdx.x, pair) * G + g < Mc;
  if constexpr (!FULL) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k * C::CPR + slot >= N || !row_ok) a[k] = zero_chunk<E, NV>();
    }
  }
  __syncthreads();  // xs complete
  STAMP(4);

  E acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = elem<E>::zero();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    // bound the scheduler's look-ahead: without this it hoists all K LDS reads (2K more VGPRs on
    // top of the 4K the slab holds) and spills
    if (k % 8 == 0) __builtin_amdgcn_sched_barrier(0);
    const E xv = L.xs[k * C::CPR + slot];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = elem<E>::fma_pk(a[k].e[i], xv, acc[i]);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int off = G; off < 64; off <<= 1) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      // (off 8: row_ror:8 inside the row of 16; off 16 / 32: permlane swaps -- add_xor, rls_common.hpp)
      float re = add_xor(elem<E>::re(acc[i]), off), im = 0.f;
      if constexpr (elem<E>::cplx) im = add_xor(elem<E>::im(acc[i]), off);
      acc[i] = elem<E>::make(re, im);
    }
  }
  if (s == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) L.part[w][g][i] = acc[i];
  }
  __syncthreads();
  // every thread adds the WV per-wave partials of ITS rows itself (broadcast reads, the order ww = 0, 1, ... of the version that
  // had G * NV threads do it and hand the result over through L.tw behind a second barrier: the same bits, one barrier less)
  STAMP(5);
  E tr[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    E sum = elem<E>::zero();
#pragma unroll
    for (int ww = 0; ww < WV; ++ww) sum = elem<E>::add(sum, L.part[ww][g][i]);
    tr[i] = sum;
  }
  double tt = 0.0;  // TT: ||t_w||^2 in Float64, the G row chunks summed in DPP steps (fixed order); the same value in every lane
  if constexpr (TT) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
      tt += (double)elem<E>::re(tr[i]) * (double)elem<E>::re(tr[i]) + (double)elem<E>::im(tr[i]) * (double)elem<E>::im(tr[i]);
    if constexpr (G >= 2) tt += dpp_d(tt, 0xB1);
    if constexpr (G >= 4) tt += dpp_d(tt, 0x4E);
    if constexpr (G >= 8) tt += dpp_d(tt, 0x141);
  }
  // The sum over the G lanes that share a column goes through LDS, not DPP: per column a lane does
  // one store here and the G-term sum below costs G reads per OUTPUT column, against 2*log2(G)
  // cross-lane adds per lane per load with shuffles (that phase was VALU-bound at 3 us).
  E* out = slab + (int64_t)blockIdx.x * N;
  // The resident kernels (SC1) drain these write-through stores before they may signal: the columns go out in parts
  // (4, or 2), so that the earlier parts' stores are already on their way while the later ones are still being formed.
  constexpr int HALVES = !(SC1 && C::NMAX == C::EPT * C::NT) ? 1 : (K % 4 == 0 && C::EPT % 4 == 0) ? 4 : (K % 2 == 0 && C::EPT % 2 == 0) ? 2 : 1;
#pragma unroll
  for (int h = 0; h < HALVES; ++h) {
#pragma unroll
    for (int k = h * (K / HALVES); k < (h + 1) * (K / HALVES); ++k) {
      if (k % 8 == 0) __builtin_amdgcn_sched_barrier(0);
      E q = elem<E>::zero();
#pragma unroll
      for (int i = 0; i < NV; ++i) q = elem<E>::fmac_pk(a[k].e[i], tr[i], q);
      slab_xg_store<E, G, K, WV>(L, g, k * C::CPR + slot, q);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (HALVES > 1) lds_barrier();  // LDS only: the earlier parts' stores stay in flight
    else __syncthreads();
    if (h == 0) STAMP(6);
#pragma unroll
    for (int e = h * (C::EPT / HALVES); e < (h + 1) * (C::EPT / HALVES); ++e) {
      const int c = tid + e * C::NT;
      if (c < C::NMAX) {
        const E sum = slab_xg_sum<E, G, K, WV>(L, c);
        if constexpr (SC1) {
          if (c < N) sc1_store_elem<E>(out + c, sum);
        } else {
          if (c < N) out[c] = sum;
        }
      }
    }
  }
  return tt;
}

// ---- several slabs per workgroup --------------------------------------------------------------------------------
// A shape with more row blocks than the chip has CUs (8192 x 4096 Float32, BASELINE configs[2]: 512 blocks of 16 rows) used to
// run as that many workgroups, one per CU at a time (a 128-register slab + 86 KiB of LDS): the second round started from
// nothing when the first had drained -- another 3.5 us until loads are out, another 3.9 us of products with the memory system
// idle.  Here ONE workgroup walks its row blocks itself: the second product of a block frees the slab registers chunk by
// chunk, and every chunk is re-used at once for the same chunk of the NEXT block, so that block streams in under the
// products of this one.  Block s of workgroup b is the block that workgroup s * gridDim.x + b of the one-slab launch owned
// (so row_block_of still pairs the workgroups of an XCD that split 128-byte lines), the column sums of the blocks are added
// in registers in that order and leave as ONE partial row per workgroup (the reduce kernel reads half as many).  Barriers
// are LDS-only: a __syncthreads() would wait for the loads in flight.
template <typename E, int G, int K, int WV, bool FULL>
struct slab_walk {
  using C = slab_cfg<E, G, K, WV>;
  const char* base;
  int64_t colstep;  // bytes between two columns CPR apart
  uint32_t col_b;
  // lane offsets of the current row block.  Ragged shapes: rounds before the last valid one (k < kl) take `off`, round kl clamps the
  // slot to the last column (`off + off_l`), rounds past N re-read round 0 (`off + off_0`, zeroed where they are used) -- three
  // registers and a scalar choice per load instead of a clamp per load (K of those, hoisted out of the block loop, spilled)
  uint32_t off, off_l, off_0;
  int slot, g, kl;
  int64_t Mc, N;
  int pair;
  bool row_ok, dead_l;  // dead_l: this lane's column of round kl is past N
  __device__ __forceinline__ void init(const E* A, int64_t lda, int64_t Mc_, int64_t N_, int pair_) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    g = lane % G;
    slot = w * C::S + lane / G;
    base = reinterpret_cast<const char*>(A);
    col_b = (uint32_t)(lda * (int64_t)sizeof(E));
    colstep = (int64_t)C::CPR * (lda * (int64_t)sizeof(E));
    Mc = Mc_, N = N_, pair = pair_;
    kl = (int)((N - 1) / C::CPR);
    dead_l = kl * C::CPR + slot >= N;
  }
  __device__ __forceinline__ void aim(int64_t vb) {  // vb: the workgroup index of the one-slab launch that owned this block
    const int64_t chunk_id = row_block_of(vb, pair) * G + g;
    row_ok = chunk_id < Mc;
    const uint32_t row_off = (uint32_t)((row_ok ? chunk_id : (Mc - 1)) * 16);
    off = row_off + (uint32_t)slot * col_b;
    if constexpr (!FULL) {
      const int last = (int)N - 1 - kl * C::CPR;  // last valid slot of round kl
      const int last0 = kl > 0 ? C::CPR - 1 : last;
      off_l = (uint32_t)(slot < last ? slot : last) * col_b - (uint32_t)slot * col_b;  // differences to `off` (mod 2^32)
      off_0 = (uint32_t)(slot < last0 ? slot : last0) * col_b - (uint32_t)slot * col_b;
    }
  }
  __device__ __forceinline__ chunk<E, C::NV> load(int k) const {
    if constexpr (FULL) {
      return load_chunk<E, C::NV>(reinterpret_cast<const E*>(base + (int64_t)k * colstep + off));
    } else {  // always-valid addresses, never a branch
      int klo = kl;  // (opaque as well: K lane masks of comparisons against kl, hoisted, spill the scalar registers)
      asm volatile("" : "+s"(klo));
      const int kc = k <= klo ? k : 0;
      // (a select between the three fields becomes an indexed load from the struct, which then lives in scratch: blend differences)
      // and opaque copies: the differences do not change from block to block, so the K blended values would be hoisted out of the
      // block loop as K live registers
      uint32_t dl = off_l, d0 = off_0;
      asm volatile("" : "+v"(dl), "+v"(d0));
      const uint32_t o = off + (k == klo ? dl : 0u) + (k > klo ? d0 : 0u);
      return load_chunk<E, C::NV>(reinterpret_cast<const E*>(base + (int64_t)kc * colstep + o));
    }
  }
  __device__ __forceinline__ bool dead(int k, bool dead_rows) const {
    int klo = kl;
    asm volatile("" : "+s"(klo));
    return dead_rows || k > klo || (k == klo && dead_l);
  }
};

// one block: t_w = A_w xs, the block's contribution A_w^H t_w added to colsum[]; with RELOAD every chunk is re-loaded for
// the block `W` aims at as soon as the second product has used it
template <typename E, int G, int K, int WV, bool FULL, bool RELOAD>
__device__ static __forceinline__ void slab_pass(chunk<E, elem<E>::vec> (&a)[K], slab_lds<E, G, K, WV>& L,
                                        E (&colsum)[slab_cfg<E, G, K, WV>::EPT], bool dead_rows,
                                        const slab_walk<E, G, K, WV, FULL>& W, int stamp_base = 8) {
  (void)stamp_base;  // -DRLS_STAMPS only (tools/stamps_walk.py): slots stamp_base .. +3 = first product done, t_w known, second product done, column sums done
  using C = slab_cfg<E, G, K, WV>;
  constexpr int NV = C::NV;
  // an opaque copy of the thread index per block: the LDS addresses of the planes beyond the 64 KiB immediate range are re-derived
  // where they are used instead of being hoisted out of the block loop as 16 more live registers (the ComplexF32 forms sit at 256)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, w = tid >> 6;
  const int g = lane % G, s = lane / G, slot = w * C::S + s;
  // The first KE rounds of the NEXT block are requested HERE, into registers of their own, not when the second product frees theirs:
  // between the last chunk of this block and the first re-load nothing was in flight (this block's t_w is being reduced: about 1 us per
  // block, plus the ramp of the re-loads).  8 rounds for real element types (12 spilled), 6 for ComplexF32 with 128-byte row pieces; the
  // ComplexF32 forms with 64-byte pieces have no registers left (250 of 256).
  constexpr int KE = !(RELOAD && K >= 16) ? 0 : !elem<E>::cplx ? 8 : G == 8 ? 6 : 0;
  chunk<E, NV> early[KE > 0 ? KE : 1];
  if constexpr (KE > 0) {
#pragma unroll
    for (int k = 0; k < KE; ++k) early[k] = W.load(k);
    __builtin_amdgcn_sched_barrier(0);
  }
  lds_barrier();  // xs complete / the previous block's reads of xg, part, tw done
  E acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = elem<E>::zero();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (k % 8 == 0) __builtin_amdgcn_sched_barrier(0);
    if constexpr (!FULL) {
      if (W.dead(k, dead_rows)) a[k] = zero_chunk<E, NV>();
    }
    const E xv = L.xs[k * C::CPR + slot];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = elem<E>::fma_pk(a[k].e[i], xv, acc[i]);
  }
  __builtin_amdgcn_sched_barrier(0);
  STAMP(stamp_base);
#pragma unroll
  for (int off = G; off < 64; off <<= 1) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float re = add_xor(elem<E>::re(acc[i]), off), im = 0.f;
      if constexpr (elem<E>::cplx) im = add_xor(elem<E>::im(acc[i]), off);
      acc[i] = elem<E>::make(re, im);
    }
  }
  if (s == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) L.part[w][g][i] = acc[i];
  }
  lds_barrier();
  STAMP(stamp_base + 1);
  E tr[NV];  // (every thread sums the per-wave partials of its rows itself: see slab_finish)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    E sum = elem<E>::zero();
#pragma unroll
    for (int ww = 0; ww < WV; ++ww) sum = elem<E>::add(sum, L.part[ww][g][i]);
    tr[i] = sum;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    // (with the re-loads in the loop a look-ahead of 8 hoists them over the products that still read the old chunks: both alive)
    if (k % (RELOAD ? 2 : 8) == 0) __builtin_amdgcn_sched_barrier(0);
    E q = elem<E>::zero();
#pragma unroll
    for (int i = 0; i < NV; ++i) q = elem<E>::fmac_pk(a[k].e[i], tr[i], q);
    slab_xg_store<E, G, K, WV>(L, g, k * C::CPR + slot, q);
    if constexpr (RELOAD) {
      if (k < KE) a[k] = early[k];
      else a[k] = W.load(k);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  STAMP(stamp_base + 2);
  lds_barrier();
#pragma unroll
  for (int e = 0; e < C::EPT; ++e) {
    const int c = tid + e * C::NT;
    if (c < C::NMAX) {
      colsum[e] = elem<E>::add(colsum[e], slab_xg_sum<E, G, K, WV>(L, c));
    }
  }
  STAMP(stamp_base + 3);
}

// with L.xs holding the input vector and a[] the first block of this workgroup (loaded through W): all its blocks, one partial row
template <typename E, int G, int K, int WV, bool FULL>
__device__ static __forceinline__ void slab_finish_multi(chunk<E, elem<E>::vec> (&a)[K], slab_lds<E, G, K, WV>& L, E* __restrict__ slab,
                                                slab_walk<E, G, K, WV, FULL>& W, int nblocks) {
  using C = slab_cfg<E, G, K, WV>;
  E colsum[C::EPT];
#pragma unroll
  for (int e = 0; e < C::EPT; ++e) colsum[e] = elem<E>::zero();
  int64_t vb = blockIdx.x;
  bool dead_rows = !W.row_ok;
  for (; vb + gridDim.x < nblocks; vb += gridDim.x) {
    W.aim(vb + gridDim.x);
    slab_pass<E, G, K, WV, FULL, true>(a, L, colsum, dead_rows, W, 8);
    dead_rows = !W.row_ok;
  }
  slab_pass<E, G, K, WV, FULL, false>(a, L, colsum, dead_rows, W, 12);
  E* out = slab + (int64_t)blockIdx.x * W.N;
#pragma unroll
  for (int e = 0; e < C::EPT; ++e) {
    const int c = threadIdx.x + e * C::NT;
    if (c < W.N) out[c] = colsum[e];
  }
}

template <typename E, int G, int K, int WV, bool FULL>
__global__ __launch_bounds__(WV * 64) void normal_slab_multi_kernel(const E* __restrict__ A, int64_t lda,
                                                                     const E* __restrict__ p, E* __restrict__ slab,
                                                                     int64_t Mc, int64_t N, int pair, int nblocks,
                                                                     const int* __restrict__ skip) {
  if (skip && *skip) return;
  using C = slab_cfg<E, G, K, WV>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  slab_lds<E, G, K, WV>& L = *reinterpret_cast<slab_lds<E, G, K, WV>*>(smem_raw);
  E pv[C::EPT];
#pragma unroll
  for (int e = 0; e < C::EPT; ++e) {
    const int64_t i = threadIdx.x + (int64_t)e * C::NT;
    pv[e] = p[i < N ? i : (N - 1)];
    if (i >= N) pv[e] = elem<E>::zero();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  slab_walk<E, G, K, WV, FULL> W;
  W.init(A, lda, Mc, N, pair);
  W.aim(blockIdx.x);
  chunk<E, C::NV> a[K];
#pragma unroll
  for (int k = 0; k < K; ++k) a[k] = W.load(k);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int e = 0; e < C::EPT; ++e) {
    const int i = threadIdx.x + e * C::NT;
    if (i < C::NMAX) L.xs[i] = pv[e];
  }
  slab_finish_multi<E, G, K, WV, FULL>(a, L, slab, W, nblocks);
}

template <typename E, int G, int K, int WV, bool FULL>
__global__ __launch_bounds__(WV * 64) void normal_slab_kernel(const E* __restrict__ A, int64_t lda,
                                                               const E* __restrict__ p, E* __restrict__ slab,
                                                               int64_t Mc, int64_t N, int pair,
                                                               const int* __restrict__ skip) {
  if (skip && *skip) return;
  using C = slab_cfg<E, G, K, WV>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  slab_lds<E, G, K, WV>& L = *reinterpret_cast<slab_lds<E, G, K, WV>*>(smem_raw);
  // p first (it must not queue behind the slab: loads return in issue order), then the slab
  E pv[C::EPT];
#pragma unroll
  for (int e = 0; e < C::EPT; ++e) {
    const int64_t i = threadIdx.x + (int64_t)e * C::NT;
    pv[e] = p[i < N ? i : (N - 1)];
    if (i >= N) pv[e] = elem<E>::zero();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  chunk<E, C::NV> a[K];
  slab_load<E, G, K, WV, FULL>(a, A, lda, Mc, N, pair);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int e = 0; e < C::EPT; ++e) {
    const int i = threadIdx.x + e * C::NT;
    if (i < C::NMAX) L.xs[i] = pv[e];
  }
  slab_finish<E, G, K, WV, FULL>(a, L, slab, Mc, N, pair);
}

// ---- CGNR pipeline: iteration = K_A (finish the previous update + one pass over A) + K_R -----
// The BLAS-1 part of src/CGNR.jl:153-176 for the elements one thread owns.  Every workgroup runs it
// redundantly (same inputs, same summation order => identical alpha, beta, done); `writer` says
// whether this workgroup also stores x, r, p.  Returns p_new in pn[].
// LEADFREE: the calling pattern of block_sum3_nolead -- two workgroup barriers per update instead of four.  Valid where a workgroup
// barrier lies between any read of one of the two scratch areas and the next write to it: the resident kernels (the exchange's
// barriers between two updates) and, since round 5, the pipeline kernels (ONE update per launch: nothing has read either area before)
template <typename E, int EPT, int NT, bool NOMASK = false, bool LEADFREE = false>
__device__ static inline bool cg_update_elems(const cgnr_scalars& S, double nre, double nim, double pp,
                                              const E (&pv)[EPT], const E (&rv)[EPT], const E (&vv)[EPT], int64_t N,
                                              double* red, E (&pn)[EPT], E (&rn)[EPT], E& a_out,
                                              cgnr_scalars& Sn) {
  const int tid = threadIdx.x;
  if constexpr (LEADFREE) block_sum3_nolead<NT / 64>(nre, nim, pp, red);
  else block_sum3_n<NT / 64>(nre, nim, pp, red);
  const float lambda = S.lambda;
  const double zeta = S.rr;
  const dcomplex alpha = dc_div({zeta, 0.0}, {nre + (lambda > 0.f ? (double)lambda * pp : 0.0), nim});
  const E a = elem<E>::make((float)alpha.re, (float)alpha.im);
  const E na = elem<E>::make(-(float)alpha.re, -(float)alpha.im);
  double rr = 0.0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = tid + (int64_t)e * NT;
    E ri = elem<E>::fma(vv[e], na, rv[e]);
    if (lambda > 0.f) ri = elem<E>::fma(elem<E>::scale(-lambda, pv[e]), a, ri);
    if (!NOMASK && i >= N) ri = elem<E>::zero();  // NOMASK: every owned index is < N (any ownership layout)
    rn[e] = ri;
    rr += (double)elem<E>::re(ri) * (double)elem<E>::re(ri) + (double)elem<E>::im(ri) * (double)elem<E>::im(ri);
  }
  if constexpr (LEADFREE) rr = block_sum_nolead<NT / 64>(rr, red);
  else rr = block_sum_n<NT / 64>(rr, red);
  const double beta = rr / zeta;
  const float bf = (float)beta;
#pragma unroll
  for (int e = 0; e < EPT; ++e) pn[e] = elem<E>::add(elem<E>::scale(bf, pv[e]), rn[e]);
  a_out = a;
  Sn = S;
  Sn.zeta = zeta;
  Sn.rr = rr;
  Sn.alpha_re = alpha.re;
  Sn.alpha_im = alpha.im;
  Sn.beta_re = beta;
  Sn.beta_im = 0.0;
  Sn.iteration = S.iteration + 1;
  const float ratio = (float)(sqrt(rr) / S.z0);
  Sn.done = (ratio <= S.rel_tol) || (Sn.iteration >= S.max_iter);
  return Sn.done != 0;
}

// The same update with alpha in its CGLS form: <p, (A^H A + lambda) p> = ||A p||^2 + lambda ||p||^2, where ||A p||^2 (`tt`) is the
// sum of the workgroups' ||t_w||^2 -- known as soon as the first product is, and summed by the exchange that sums the partial
// rows -- and ||p||^2 (`pp`, lambda > 0 only) does not depend on the exchange either.  Both arrive COMPLETE (the same bits in every
// thread): of the two dependent block reductions of src/CGNR.jl:153-176 only ||r||^2 is left behind the exchange.  alpha is real
// (the reference's dot(p, v) carries a rounding-level imaginary part, SURVEY section 7 hard part 5: differences of O(eps)); no
// cancellation anywhere -- a sum of squares.  ||r||^2 stays the norm of the STORED r (src/CGNR.jl:171).
template <typename E, int EPT, int NT, bool NOMASK = false>
__device__ static inline bool cg_update_elems_tt(const cgnr_scalars& S, double tt, double pp, const E (&pv)[EPT], const E (&rv)[EPT],
                                                 const E (&vv)[EPT], int64_t N, double* red, E (&pn)[EPT], E (&rn)[EPT],
                                                 float& a_out, cgnr_scalars& Sn) {
  const int tid = threadIdx.x;
  const float lambda = S.lambda;
  const double zeta = S.rr;
  const double alpha = zeta / (tt + (lambda > 0.f ? (double)lambda * pp : 0.0));
  const float a = (float)alpha;
  double rr = 0.0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = tid + (int64_t)e * NT;
    E ri = elem<E>::make(fmaf(elem<E>::re(vv[e]), -a, elem<E>::re(rv[e])), fmaf(elem<E>::im(vv[e]), -a, elem<E>::im(rv[e])));
    if (lambda > 0.f) {  // (p .* -lambda) * alpha, src/CGNR.jl:168
      const E t = elem<E>::scale(-lambda, pv[e]);
      ri = elem<E>::make(fmaf(elem<E>::re(t), a, elem<E>::re(ri)), fmaf(elem<E>::im(t), a, elem<E>::im(ri)));
    }
    if (!NOMASK && i >= N) ri = elem<E>::zero();
    rn[e] = ri;
    rr += (double)elem<E>::re(ri) * (double)elem<E>::re(ri) + (double)elem<E>::im(ri) * (double)elem<E>::im(ri);
  }
  rr = block_sum_nolead<NT / 64>(rr, red);
  const double beta = rr / zeta;
  const float bf = (float)beta;
#pragma unroll
  for (int e = 0; e < EPT; ++e) pn[e] = elem<E>::add(elem<E>::scale(bf, pv[e]), rn[e]);
  a_out = a;
  Sn = S;
  Sn.zeta = zeta;
  Sn.rr = rr;
  Sn.alpha_re = alpha;
  Sn.alpha_im = 0.0;
  Sn.beta_re = beta;
  Sn.beta_im = 0.0;
  Sn.iteration = S.iteration + 1;
  const float ratio = (float)(sqrt(rr) / S.z0);
  Sn.done = (ratio <= S.rel_tol) || (Sn.iteration >= S.max_iter);
  return Sn.done != 0;
}

// Which elements of the length-N vectors a thread owns in K_A.  Strided (i = tid + e NT: any N, masked tail) or, for
// the hinted full-size instantiation, in 16-byte pieces (i = q NT V + tid V + j, e = q V + j, V = 2 complex / 4
// real): the vectors then come in with one 16-byte load per piece -- 9 load instructions per lane instead of 15.
template <typename E, int EPT, int NT, bool WIDE>
__device__ static inline int64_t own_index(int tid, int e) {
  if constexpr (WIDE) {
    constexpr int V = elem<E>::vec;
    return (int64_t)(e / V) * (NT * V) + (int64_t)tid * V + (e % V);
  } else {
    return tid + (int64_t)e * NT;
  }
}
// the same for a vector of any length n (a multiple of the piece size): pieces at or beyond n come back as zeros (clamped
// address, no branch around the load)
template <typename E, int EPT, int NT>
__device__ static inline void load_owned_wide_masked(E (&dst)[EPT], const E* __restrict__ src, int tid, int64_t n) {
  constexpr int V = elem<E>::vec;
#pragma unroll
  for (int q = 0; q < EPT / V; ++q) {
    const int64_t o = (int64_t)q * (NT * V) + (int64_t)tid * V;
    const chunk<E, V> c = load_chunk<E, V>(src + (o < n ? o : 0));
#pragma unroll
    for (int j = 0; j < V; ++j) dst[q * V + j] = o < n ? c.e[j] : elem<E>::zero();
  }
}
template <typename E, int EPT, int NT>
__device__ static inline void load_owned_wide(E (&dst)[EPT], const E* __restrict__ src, int tid) {
  constexpr int V = elem<E>::vec;
#pragma unroll
  for (int q = 0; q < EPT / V; ++q) {
    const chunk<E, V> c = load_chunk<E, V>(src + (int64_t)q * (NT * V) + (int64_t)tid * V);
#pragma unroll
    for (int j = 0; j < V; ++j) dst[q * V + j] = c.e[j];
  }
}

// the small per-RHS vector loads of K_A (both candidate buffers; the right one is selected once the
// scalars are known) and this thread's share of the partial dots
template <typename E, int EPT>
struct pipe_small {
  E pa[EPT], pb[EPT], ra[EPT], rb[EPT], vv[EPT], xv[EPT];
  double d0, d1, d2;
};

// `hint` >= 0: the host knows which (r, p) pair is current, only that one is loaded (into both slots); x is
// loaded by the one workgroup that stores it.  27 -> 15 loads per lane: every one of them is issued by all 8
// waves ahead of the slab, so they cost issue slots as well as latency (0.7 us per iteration at the headline).
// (A compile-time switch, not a branch on `hint`: loads under a wave-uniform branch make the compiler wait
// vmcnt(0) at the join.)
template <typename E, int EPT, int NT, bool HINTED, bool WIDE = false>
__device__ static inline void pipe_load_small(pipe_small<E, EPT>& s, const E* x, const E* r0, const E* p0, const E* r1,
                                              const E* p1, const E* v, const double* dots, int ndots, int64_t N,
                                              int64_t vo, int b, int hint) {
  const int tid = threadIdx.x;
  const bool writer = blockIdx.x == 0;
  const E* rh = hint == 1 ? r1 : r0;
  const E* ph = hint == 1 ? p1 : p0;
  if constexpr (WIDE) {  // hinted, N == NT * EPT: 16-byte pieces
    load_owned_wide<E, EPT, NT>(s.pa, ph + vo, tid);
    load_owned_wide<E, EPT, NT>(s.ra, rh + vo, tid);
    load_owned_wide<E, EPT, NT>(s.vv, v + vo, tid);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      s.pb[e] = s.pa[e];
      s.rb[e] = s.ra[e];
      s.xv[e] = writer ? x[vo + own_index<E, EPT, NT, true>(tid, e)] : elem<E>::zero();
    }
  } else
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = tid + (int64_t)e * NT;
    const int64_t ic = vo + (i < N ? i : (N - 1));
    if constexpr (HINTED) {
      s.xv[e] = writer ? x[ic] : elem<E>::zero();  // only workgroup 0 stores x
      s.pa[e] = s.pb[e] = ph[ic];
      s.ra[e] = s.rb[e] = rh[ic];
    } else {
      s.xv[e] = x[ic];  // only workgroup 0 stores x, but a late load would queue behind the slab
      s.pa[e] = p0[ic];
      s.pb[e] = p1[ic];
      s.ra[e] = r0[ic];
      s.rb[e] = r1[ic];
    }
    s.vv[e] = v[ic];
  }
  const int dtid = tid < ndots ? tid : 0;  // clamped address; dead lanes zeroed by the caller
  const double* db = dots + (int64_t)b * 4 * ndots;
  s.d0 = db[4 * dtid];
  s.d1 = db[4 * dtid + 1];
  s.d2 = db[4 * dtid + 2];
}

// CG update of one right-hand side (every workgroup redundantly; workgroup 0 stores) followed by the
// two products from the register slab.  Wave-uniform control flow: every thread reads the same scalars.
// the 16-byte ownership layout applies to the hinted, full-size, single-right-hand-side instantiation whose element
// count per thread is a whole number of 16-byte pieces
template <typename E, int G, int K, int WV, bool FULL, bool HINTED>
__device__ __host__ constexpr bool pipe_wide() {
  return FULL && HINTED && (slab_cfg<E, G, K, WV>::EPT % elem<E>::vec == 0);
}

template <typename E, int G, int K, int WV, bool FULL, bool HINTED, bool MULTI = false>
__device__ static inline void pipe_process_rhs(chunk<E, elem<E>::vec> (&a)[K], slab_lds<E, G, K, WV>& L,
                                               pipe_small<E, slab_cfg<E, G, K, WV>::EPT>& sm, E* x, E* r0, E* p0,
                                               E* r1, E* p1, E* slab_b, const cgnr_scalars* sc_b, cgnr_scalars* scn_b,
                                               int ndots, int64_t Mc, int64_t N, int64_t vo, int pair, int hint,
                                               slab_walk<E, G, K, WV, FULL>* W = nullptr, int nblocks = 0, double* ttw_b = nullptr) {
  using C = slab_cfg<E, G, K, WV>;
  constexpr int EPT = C::EPT;
  constexpr bool WIDE = pipe_wide<E, G, K, WV, FULL, HINTED>();
  const int tid = threadIdx.x;
  const bool writer = blockIdx.x == 0;
  if (tid >= ndots) sm.d0 = sm.d1 = sm.d2 = 0.0;
  const cgnr_scalars S = *sc_b;
  if (S.done) {
    if (writer && tid == 0) {
      cgnr_scalars Sn = S;
      Sn.fresh = 0;
      *scn_b = Sn;
    }
    return;
  }
  E pv[EPT], rv[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = own_index<E, EPT, C::NT, WIDE>(tid, e);
    pv[e] = S.cur ? sm.pb[e] : sm.pa[e];
    rv[e] = S.cur ? sm.rb[e] : sm.ra[e];
    if (i >= N) pv[e] = elem<E>::zero();
  }
  if constexpr (HINTED) {
    if (S.cur != hint) {  // wrong hint (never with the host's bookkeeping): fetch the right pair, late
      const E* rc = (S.cur ? r1 : r0) + vo;
      const E* pc = (S.cur ? p1 : p0) + vo;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int64_t i = own_index<E, EPT, C::NT, WIDE>(tid, e);
        const int64_t ic = i < N ? i : (N - 1);
        pv[e] = pc[ic];
        rv[e] = rc[ic];
        if (i >= N) pv[e] = elem<E>::zero();
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) HERE, so that the common path has nothing to wait for at the join
    }
  }
  STAMP(2);
  cgnr_scalars Sn;
  if (S.pending) {
    E pn[EPT], rn[EPT], al;
    const bool done = cg_update_elems<E, EPT, C::NT, WIDE, true>(S, sm.d0, sm.d1, sm.d2, pv, rv, sm.vv, N, L.red, pn, rn, al, Sn);
    if (writer) {
      E* rw = (S.cur ? r0 : r1) + vo;
      E* pw = (S.cur ? p0 : p1) + vo;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int64_t i = own_index<E, EPT, C::NT, WIDE>(tid, e);
        if (i < N) {
          x[vo + i] = elem<E>::fma(pv[e], al, sm.xv[e]);
          rw[i] = rn[e];
          pw[i] = pn[e];
        }
      }
    }
    Sn.cur = 1 - S.cur;
    Sn.pending = done ? 0 : 1;
    Sn.fresh = done ? 0 : 1;
    if (writer && tid == 0) *scn_b = Sn;
    if (done) return;  // uniform: every workgroup derived the same scalars
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = (int)own_index<E, EPT, C::NT, WIDE>(tid, e);
      if (i < C::NMAX) L.xs[i] = pn[e];
    }
  } else {
    Sn = S;
    Sn.pending = 1;
    Sn.fresh = 1;
    if (writer && tid == 0) *scn_b = Sn;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = (int)own_index<E, EPT, C::NT, WIDE>(tid, e);
      if (i < C::NMAX) L.xs[i] = pv[e];
    }
  }
  STAMP(3);
  if constexpr (MULTI) {
    slab_finish_multi<E, G, K, WV, FULL>(a, L, slab_b, *W, nblocks);
  } else {
    const double ttw = slab_finish<E, G, K, WV, FULL, false, true>(a, L, slab_b, Mc, N, pair);
    if (ttw_b && tid == 0) ttw_b[blockIdx.x] = ttw;   // this row block's share of ||A p||^2
  }
  STAMP(7);
}

// One or several right-hand sides share the slab: the A loads happen once, then each RHS runs its CG
// update + the two products from the same registers (BASELINE config 4, shared-A flavour).  Per-RHS
// arrays are `vstride` elements apart (x, r0, p0, r1, p1, v), `slab_stride` (slab) and 4*ndots (dots).
struct pipe_rhs_ptrs {
  int64_t vstride, slab_stride;
  int nrhs;
  int hint;  // rls_cgnr_pipe::cur_hint
  // alpha in its CGLS form (cg_update_elems_tt's identity: <p, A^H A p> = ||A p||^2 = sum_w ||t_w||^2): K_A leaves ||t_w||^2 of
  // its row block in ttw[rhs * tt_rows + workgroup], K_R folds those into the first slot of the partial dots (the second, the
  // imaginary part, is zero) instead of forming <p, v> -- the update in the next K_A is unchanged.  Null: K_R forms <p, v> as before
  // (the MULTI instantiations, whose complex forms have no registers left for one more accumulator across their blocks).
  double* ttw = nullptr;
  int tt_rows = 0;
};

// MULTI: gridDim.x < nblocks, every workgroup walks several row blocks (slab_finish_multi) and leaves one partial row
template <typename E, int G, int K, int WV, bool FULL, bool BATCHED, bool HINTED, bool MULTI = false>
__global__ __launch_bounds__(WV * 64) void cgnr_pipe_a_kernel(const E* __restrict__ A, int64_t lda, E* __restrict__ x,
                                                               E* r0, E* p0, E* r1, E* p1, const E* __restrict__ v,
                                                               E* __restrict__ slab, const double* __restrict__ dots,
                                                               int ndots, const cgnr_scalars* __restrict__ sc,
                                                               cgnr_scalars* __restrict__ scn, int64_t Mc, int64_t N,
                                                               int pair, int order_mode, pipe_rhs_ptrs R, int nblocks) {
  using C = slab_cfg<E, G, K, WV>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  slab_lds<E, G, K, WV>& L = *reinterpret_cast<slab_lds<E, G, K, WV>*>(smem_raw);
  STAMP(0);
  STAMP_R_KEEP();
  // First right-hand side: its small loads go out ahead of the slab.  A CU's vector-memory path
  // returns loads in issue order (measured with stamps), so every wave issues them, a workgroup
  // barrier makes sure no wave has slab loads queued in front of another wave's small loads, and only
  // then the 256 KiB slab goes out; the CG update runs under its flight.
  pipe_small<E, C::EPT> sm;
  pipe_load_small<E, C::EPT, C::NT, HINTED, pipe_wide<E, G, K, WV, FULL, HINTED>()>(sm, x, r0, p0, r1, p1, v, dots, ndots, N, 0,
                                                                                     0, R.hint);
  if (order_mode == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    __builtin_amdgcn_s_barrier();
  }
  __builtin_amdgcn_sched_barrier(0);
  chunk<E, C::NV> a[K];
  if constexpr (MULTI) {
    static_assert(!BATCHED, "one right-hand side");
    slab_walk<E, G, K, WV, FULL> W;
    W.init(A, lda, Mc, N, pair);
    W.aim(blockIdx.x);
#pragma unroll
    for (int k = 0; k < K; ++k) a[k] = W.load(k);
    __builtin_amdgcn_sched_barrier(0);
    pipe_process_rhs<E, G, K, WV, FULL, HINTED, true>(a, L, sm, x, r0, p0, r1, p1, slab, sc, scn, ndots, Mc, N, 0, pair, R.hint,
                                                       &W, nblocks);
    return;
  }
  slab_load<E, G, K, WV, FULL>(a, A, lda, Mc, N, pair);
  __builtin_amdgcn_sched_barrier(0);
  STAMP(1);
  pipe_process_rhs<E, G, K, WV, FULL, HINTED>(a, L, sm, x, r0, p0, r1, p1, slab, sc, scn, ndots, Mc, N, 0, pair, R.hint, nullptr, 0, R.ttw);
  // further right-hand sides reuse the registers (a separate instantiation: with the loop present the
  // compiler hoists per-load address math out of it and the single-RHS kernel spills)
  if constexpr (BATCHED)
  for (int b = 1; b < R.nrhs; ++b) {
    const int64_t vo = (int64_t)b * R.vstride;
    pipe_load_small<E, C::EPT, C::NT, HINTED>(sm, x, r0, p0, r1, p1, v, dots, ndots, N, vo, b, R.hint);
    pipe_process_rhs<E, G, K, WV, FULL, HINTED>(a, L, sm, x, r0, p0, r1, p1, slab + (int64_t)b * R.slab_stride, sc + b,
                                                 scn + b, ndots, Mc, N, vo, pair, R.hint, nullptr, 0,
                                                 R.ttw ? R.ttw + (int64_t)b * R.tt_rows : nullptr);
  }
}

// K_R: v = sum of the slab rows (fixed order) for 16 columns per workgroup, the partial dots
// <p, v> and ||p||^2 for those columns, and the commit of the staged scalars.
// sum of this thread's partial rows for column jc: rows wy, wy+ny, wy+2ny, ...  Four independent loads per trip
// (none depends on anything but the kernel arguments, so they leave with the first instruction of the kernel);
// the order of the additions is fixed: s0 takes trips' rows 0 and 2, s1 rows 1 and 3, result s0 + s1.
template <typename E, int B>
__device__ static inline void slab_column_batches(const E* __restrict__ slab, int nwg, int64_t N, int64_t jc, int wy,
                                                  int ny, E& s0, E& s1) {
  for (int wgi = wy; wgi < nwg; wgi += B * ny) {
    E a[B];
#pragma unroll
    for (int q = 0; q < B; ++q) {
      const int row = wgi + q * ny;
      const int rc = row < nwg ? row : wgi;  // clamped address, masked below
      a[q] = slab[(int64_t)rc * N + jc];
    }
#pragma unroll
    for (int q = 0; q < B; ++q) {
      if (wgi + q * ny >= nwg) a[q] = elem<E>::zero();
      if (q & 1) s1 = elem<E>::add(s1, a[q]);
      else s0 = elem<E>::add(s0, a[q]);
    }
  }
}
template <typename E>
__device__ static inline E slab_column_sum(const E* __restrict__ slab, int nwg, int64_t N, int64_t jc, int wy, int ny) {
  E s0 = elem<E>::zero(), s1 = elem<E>::zero();
  // one batch of independent loads whenever it fits: 4 rows per thread at the headline shape (256 partial rows, 64
  // row groups), 8 at 512 partial rows (8192 x 4096 Float32); the order of the additions is the same either way
  if (nwg > 4 * ny) slab_column_batches<E, 8>(slab, nwg, N, jc, wy, ny, s0, s1);
  else slab_column_batches<E, 4>(slab, nwg, N, jc, wy, ny, s0, s1);
  return elem<E>::add(s0, s1);
}

// combine the row groups of a workgroup: first the 4 row groups inside each wave (lanes l, l+16, l+32, l+48 hold
// the same column), then one value per wave through LDS; valid in lanes 0..15 of wave 0.  Fixed order.
template <typename E>
__device__ static inline E slab_group_combine(E s, E (*sm)[16]) {
  const int cx = threadIdx.x % 16, w = threadIdx.x / 64, nw = blockDim.x / 64;
  float re = elem<E>::re(s), im = elem<E>::im(s);
  re = pair_sum16(re);
  if constexpr (elem<E>::cplx) im = pair_sum16(im);
  re = pair_sum32(re);
  if constexpr (elem<E>::cplx) im = pair_sum32(im);
  if ((threadIdx.x & 63) < 16) sm[w][cx] = elem<E>::make(re, im);
  __syncthreads();
  E t = elem<E>::zero();
  if (threadIdx.x < 16)
    for (int i = 0; i < nw; ++i) t = elem<E>::add(t, sm[i][cx]);
  return t;
}

template <typename E>
__global__ __launch_bounds__(1024) void cgnr_pipe_r_kernel(const E* __restrict__ slab, int nwg, int64_t N,
                                                          E* __restrict__ v, const E* p0, const E* p1,
                                                          double* __restrict__ dots, cgnr_scalars* __restrict__ sc,
                                                          const cgnr_scalars* __restrict__ scn, pipe_rhs_ptrs R) {
  const int b = blockIdx.y;  // right-hand side
  STAMP_R(0);
  slab += (int64_t)b * R.slab_stride;
  v += (int64_t)b * R.vstride;
  p0 += (int64_t)b * R.vstride;
  p1 += (int64_t)b * R.vstride;
  dots += (int64_t)b * 4 * gridDim.x;
  __shared__ E sm[16][16];
  const int cx = threadIdx.x % 16, wy = threadIdx.x / 16, ny = blockDim.x / 16;  // ny row groups
  const int64_t j = (int64_t)blockIdx.x * 16 + cx;
  const int64_t jc = j < N ? j : (N - 1);
  // everything this kernel reads is requested up front: the partial rows, both candidates for p (which one is
  // current is in the staged scalars) and the scalars themselves -- one memory round trip instead of three
  const E sum = slab_column_sum<E>(slab, nwg, N, jc, wy, ny);
  E pa = elem<E>::zero(), pb = elem<E>::zero();
  if (wy == 0) {
    pa = p0[jc];
    pb = p1[jc];
  }
  // CGLS form: this block's share of ||A p||^2 -- the row blocks blockIdx, blockIdx + gridDim, ... of K_A's ||t_w||^2 (requested
  // with everything else; fixed order)
  double ttb = 0.0;
  if (R.ttw && threadIdx.x == 0) {
    const double* tw = R.ttw + (int64_t)b * R.tt_rows;
    for (int row = blockIdx.x; row < R.tt_rows; row += gridDim.x) ttb += tw[row];
  }
  const cgnr_scalars Sn = scn[b];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    cgnr_scalars c = Sn;
    c.fresh = 0;
    sc[b] = c;
  }
  if (!Sn.fresh) return;
  STAMP_R(1);
  const E t = slab_group_combine<E>(sum, sm);
  STAMP_R(2);
  if (wy == 0) {
    double dre = 0.0, dim_ = 0.0, pp = 0.0;
    if (j < N) {
      v[j] = t;
      const E pj = Sn.cur ? pb : pa;
      if (!R.ttw) {
        dre = (double)elem<E>::re(pj) * (double)elem<E>::re(t) + (double)elem<E>::im(pj) * (double)elem<E>::im(t);
        dim_ = (double)elem<E>::re(pj) * (double)elem<E>::im(t) - (double)elem<E>::im(pj) * (double)elem<E>::re(t);
      }
      pp = (double)elem<E>::re(pj) * (double)elem<E>::re(pj) + (double)elem<E>::im(pj) * (double)elem<E>::im(pj);
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {  // lanes 0..15 of wave 0
      dre += __shfl_xor(dre, off, 64);
      dim_ += __shfl_xor(dim_, off, 64);
      pp += __shfl_xor(pp, off, 64);
    }
    if (cx == 0) {
      dots[4 * blockIdx.x] = R.ttw ? ttb : dre;
      dots[4 * blockIdx.x + 1] = dim_;
      dots[4 * blockIdx.x + 2] = pp;
    }
  }
  STAMP_R(3);
}

// K_F: apply a pending update and bring r, p back into the caller's vectors (single workgroup)
template <typename E, int EPT>
__global__ __launch_bounds__(FIN_THREADS) void cgnr_pipe_f_kernel(E* __restrict__ x, E* r0, E* p0, E* r1, E* p1,
                                                               const E* __restrict__ v,
                                                               const double* __restrict__ dots, int ndots,
                                                               cgnr_scalars* __restrict__ sc, int64_t N,
                                                               pipe_rhs_ptrs R, rls_mailbox_slot mb) {
  __shared__ double red[48];
  const int b = blockIdx.x;  // right-hand side
  sc += b;
  x += (int64_t)b * R.vstride;
  r0 += (int64_t)b * R.vstride;
  p0 += (int64_t)b * R.vstride;
  r1 += (int64_t)b * R.vstride;
  p1 += (int64_t)b * R.vstride;
  v += (int64_t)b * R.vstride;
  dots += (int64_t)b * 4 * ndots;
  const cgnr_scalars S = *sc;
  const int tid = threadIdx.x;
  if (!S.pending && S.cur == 0) {
    if (tid < 64) rls_mailbox_publish(mb, S, tid);  // (single right-hand side plans only arm the slot)
    return;
  }
  const E* rc = S.cur ? r1 : r0;
  const E* pc = S.cur ? p1 : p0;
  E pv[EPT], rv[EPT], vv[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = tid + (int64_t)e * FIN_THREADS;
    const int64_t ic = i < N ? i : (N - 1);
    pv[e] = pc[ic];
    rv[e] = rc[ic];
    vv[e] = v[ic];
  }
  double d0 = 0.0, d1 = 0.0, d2 = 0.0;
  if (tid < ndots) {
    d0 = dots[4 * tid];
    d1 = dots[4 * tid + 1];
    d2 = dots[4 * tid + 2];
  }
  cgnr_scalars Sn = S;
  if (S.pending && !S.done) {
    E pn[EPT], rn[EPT], al;
    cg_update_elems<E, EPT, FIN_THREADS>(S, d0, d1, d2, pv, rv, vv, N, red, pn, rn, al, Sn);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = tid + (int64_t)e * FIN_THREADS;
      if (i < N) {
        x[i] = elem<E>::fma(pv[e], al, x[i]);
        r0[i] = rn[e];
        p0[i] = pn[e];
      }
    }
  } else if (S.cur) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = tid + (int64_t)e * FIN_THREADS;
      if (i < N) {
        r0[i] = rv[e];
        p0[i] = pv[e];
      }
    }
  }
  Sn.pending = 0;
  Sn.cur = 0;
  Sn.fresh = 0;
  __syncthreads();
  if (tid == 0) *sc = Sn;
  if (tid < 64) rls_mailbox_publish(mb, Sn, tid);
}

// v[j] = sum_w slab[w][j] in a fixed order: 16 columns per workgroup, blockDim.x / 16 row groups per column -- the loads and the
// combine of cgnr_pipe_r_kernel (every partial row of a thread requested with the first instructions; round 5: this kernel summed
// 16 rows per thread in a two-load loop with 256 threads and took 4.9 us where K_R, which does more, takes 3.0)
template <typename E>
__global__ __launch_bounds__(1024) void slab_reduce_kernel(const E* __restrict__ slab, int nwg, int64_t N,
                                                           E* __restrict__ v, const int* __restrict__ skip) {
  __shared__ E sm[16][16];
  const int cx = threadIdx.x % 16, wy = threadIdx.x / 16, ny = blockDim.x / 16;
  const int64_t j = (int64_t)blockIdx.x * 16 + cx;
  const int64_t jc = j < N ? j : (N - 1);
  const E sum = slab_column_sum<E>(slab, nwg, N, jc, wy, ny);  // (requested before `skip` is looked at: it only gates the store)
  const bool off = skip && *skip;
  const E t = slab_group_combine<E>(sum, sm);
  if (threadIdx.x < 16 && j < N && !off) v[j] = t;
}

// ---- FISTA pipeline: iteration = K_A (previous gradient/prox/momentum update + one pass over A) + K_R --
// src/FISTA.jl:153-180 for the elements one thread owns, plus the NEXT iteration's Nesterov step (:144-148).
// Every workgroup runs it redundantly (same inputs, same summation order => identical scalars).
// NOMASK: the caller's ownership layout is not the strided one and its elements beyond N are zeros already (they stay
// zeros: the elementwise prox maps and projections map 0 to 0)
template <typename E, int EPT, int NT, bool NOMASK = false, bool LEADFREE = false>
__device__ static inline bool fista_update_elems(const fista_scalars& S, const E (&raw)[EPT], const E (&x0v)[EPT],
                                                 const E (&yv)[EPT], const E (&xk)[EPT], int64_t N, double* red,
                                                 E (&ri)[EPT], E (&xn)[EPT], E (&yn)[EPT], fista_scalars& Sn) {
  const int tid = threadIdx.x;
  const float rho = S.rho, thr = S.rho * S.lambda;  // prox!(reg, x, rho * lambda(reg))        :164
  double rn = 0.0, d = 0.0, zero = 0.0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = tid + (int64_t)e * NT;
    E r = elem<E>::sub(raw[e], x0v[e]);                                   // res .-= x0      :153
    E xv = elem<E>::sub(yv[e], elem<E>::scale(rho, r));                   // x .-= rho .* res :154
    xv = fista_proj_elem<E>(fista_prox_elem<E>(xv, S.reg_kind, thr), S.proj_kind);
    if (!NOMASK && i >= N) {
      r = elem<E>::zero();
      xv = elem<E>::zero();
    }
    ri[e] = r;
    xn[e] = xv;
    rn += (double)elem<E>::re(r) * (double)elem<E>::re(r) + (double)elem<E>::im(r) * (double)elem<E>::im(r);
    const E df = elem<E>::sub(xv, xk[e]);
    d += (double)elem<E>::re(r) * (double)elem<E>::re(df) + (double)elem<E>::im(r) * (double)elem<E>::im(df);
  }
  if constexpr (LEADFREE) block_sum3_nolead<NT / 64>(rn, d, zero, red);  // resident kernels: barriers of the exchange in between
  else block_sum3_n<NT / 64>(rn, d, zero, red);
  float theta = S.theta;
  if (S.restart && d > 0.0) theta = 1.f;                                  // gradient restart  :171-176
  const float theta_old = theta;                                          // :179
  theta = (1.f + sqrtf(1.f + 4.f * theta_old * theta_old)) / 2.f;         // :180
  const double res_norm = sqrt(rn);
  const float rel = (float)(res_norm / S.norm_x0);                        // :156
  const int done = (rel < S.rel_tol) || (S.iteration + 1 >= S.max_iter);  // :187-189
  const float c1 = (1.f - theta_old) / theta, c2 = (theta_old - 1.f) / theta + 1.f;
#pragma unroll
  for (int e = 0; e < EPT; ++e) yn[e] = elem<E>::add(elem<E>::scale(c1, xk[e]), elem<E>::scale(c2, xn[e]));
  RLS_FISTA_COPY(Sn, S);
  Sn.res_norm = res_norm;
  Sn.rel_res_norm = (double)rel;
  Sn.theta = theta;
  Sn.theta_old = theta_old;
  Sn.iteration = S.iteration + 1;
  Sn.done = done;
  return done != 0;
}

template <typename E, int G, int K, int WV, bool FULL, bool HINTED, bool MULTI = false>
__global__ __launch_bounds__(WV * 64) void fista_pipe_a_kernel(const E* __restrict__ A, int64_t lda, E* b0, E* b1,
                                                                const E* __restrict__ x0, E* __restrict__ res, E* y0,
                                                                E* y1, const E* __restrict__ res_raw,
                                                                E* __restrict__ slab,
                                                                const fista_scalars* __restrict__ sc,
                                                                fista_scalars* __restrict__ scn, int64_t Mc, int64_t N,
                                                                int pair, int hint, int nblocks) {
  using C = slab_cfg<E, G, K, WV>;
  constexpr int EPT = C::EPT;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  slab_lds<E, G, K, WV>& L = *reinterpret_cast<slab_lds<E, G, K, WV>*>(smem_raw);
  const int tid = threadIdx.x;
  const bool writer = blockIdx.x == 0;
  // small loads first, barrier, then the slab: see K_A of CGNR.  Both candidates of each ping-pong pair unless the
  // host passed the parity of the iteration count (HINTED: 16 loads per lane instead of 24)
  E raw[EPT], x0v[EPT], ya[EPT], yb[EPT], ba[EPT], bb[EPT];
  const E* yh = hint == 1 ? y1 : y0;
  const E* bh = hint == 1 ? b1 : b0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = tid + (int64_t)e * C::NT;
    const int64_t ic = i < N ? i : (N - 1);
    raw[e] = res_raw[ic];
    x0v[e] = x0[ic];
    if constexpr (HINTED) {
      ya[e] = yb[e] = yh[ic];
      ba[e] = bb[e] = bh[ic];
    } else {
      ya[e] = y0[ic];
      yb[e] = y1[ic];
      ba[e] = b0[ic];
      bb[e] = b1[ic];
    }
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  chunk<E, C::NV> a[K];
  slab_walk<E, G, K, WV, FULL> W;  // MULTI only
  if constexpr (MULTI) {
    W.init(A, lda, Mc, N, pair);
    W.aim(blockIdx.x);
#pragma unroll
    for (int k = 0; k < K; ++k) a[k] = W.load(k);
  } else {
    slab_load<E, G, K, WV, FULL>(a, A, lda, Mc, N, pair);
  }
  __builtin_amdgcn_sched_barrier(0);
  fista_scalars S;
  RLS_FISTA_COPY(S, *sc);
  if (S.done) {
    if (writer && tid == 0) {
      fista_scalars Sn;
      RLS_FISTA_COPY(Sn, S);
      Sn.fresh = 0;
      RLS_FISTA_COPY(*scn, Sn);
    }
    return;
  }
  E yv[EPT], xk[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = tid + (int64_t)e * C::NT;
    yv[e] = S.ycur ? yb[e] : ya[e];
    xk[e] = (S.iteration & 1) ? bb[e] : ba[e];  // state.x == buf[iteration & 1]
    if (i >= N) yv[e] = elem<E>::zero();
  }
  if constexpr (HINTED) {
    if (S.ycur != hint || (S.iteration & 1) != hint) {  // wrong hint (never with the host's bookkeeping): re-load, late
      const E* yc = S.ycur ? y1 : y0;
      const E* bc = (S.iteration & 1) ? b1 : b0;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int64_t i = tid + (int64_t)e * C::NT;
        const int64_t ic = i < N ? i : (N - 1);
        yv[e] = yc[ic];
        xk[e] = bc[ic];
        if (i >= N) yv[e] = elem<E>::zero();
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) inside the branch: nothing to wait for at the join otherwise
    }
  }
  fista_scalars Sn;
  if (S.pending) {
    E ri[EPT], xn[EPT], yn[EPT];
    const bool done = fista_update_elems<E, EPT, C::NT, false, true>(S, raw, x0v, yv, xk, N, L.red, ri, xn, yn, Sn);
    if (writer) {
      E* xw = (S.iteration & 1) ? b0 : b1;  // the reference's pointer swap: new x goes where x_{k-1} was
      E* yw = S.ycur ? y0 : y1;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int64_t i = tid + (int64_t)e * C::NT;
        if (i < N) {
          xw[i] = xn[e];
          res[i] = ri[e];
          if (!done) yw[i] = yn[e];
        }
      }
    }
    Sn.ycur = done ? S.ycur : 1 - S.ycur;
    Sn.pending = done ? 0 : 1;
    Sn.fresh = done ? 0 : 1;
    if (writer && tid == 0) RLS_FISTA_COPY(*scn, Sn);
    if (done) return;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = tid + e * C::NT;
      if (i < C::NMAX) L.xs[i] = yn[e];
    }
  } else {
    RLS_FISTA_COPY(Sn, S);
    Sn.pending = 1;
    Sn.fresh = 1;
    if (writer && tid == 0) RLS_FISTA_COPY(*scn, Sn);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = tid + e * C::NT;
      if (i < C::NMAX) L.xs[i] = yv[e];
    }
  }
  if constexpr (MULTI) slab_finish_multi<E, G, K, WV, FULL>(a, L, slab, W, nblocks);
  else slab_finish<E, G, K, WV, FULL>(a, L, slab, Mc, N, pair);
}

// K_R of FISTA: res_raw = sum of the slab rows (fixed order) + commit of the staged scalars
template <typename E>
__global__ __launch_bounds__(1024) void fista_pipe_r_kernel(const E* __restrict__ slab, int nwg, int64_t N,
                                                            E* __restrict__ res_raw, fista_scalars* __restrict__ sc,
                                                            const fista_scalars* __restrict__ scn) {
  __shared__ E sm[16][16];
  const int cx = threadIdx.x % 16, wy = threadIdx.x / 16, ny = blockDim.x / 16;
  const int64_t j = (int64_t)blockIdx.x * 16 + cx;
  const int64_t jc = j < N ? j : (N - 1);
  const E sum = slab_column_sum<E>(slab, nwg, N, jc, wy, ny);  // requested before the scalars are looked at
  fista_scalars Sn;
  RLS_FISTA_COPY(Sn, *scn);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    fista_scalars c;
    RLS_FISTA_COPY(c, Sn);
    c.fresh = 0;
    RLS_FISTA_COPY(*sc, c);
  }
  if (!Sn.fresh) return;
  const E t = slab_group_combine<E>(sum, sm);
  if (wy == 0 && j < N) res_raw[j] = t;
}

// K_F of FISTA: apply a pending update (single workgroup)
template <typename E, int EPT>
__global__ __launch_bounds__(FIN_THREADS) void fista_pipe_f_kernel(E* b0, E* b1, const E* __restrict__ x0,
                                                                    E* __restrict__ res, E* y0, E* y1,
                                                                    const E* __restrict__ res_raw,
                                                                    fista_scalars* __restrict__ sc, int64_t N, rls_mailbox_slot mb) {
  __shared__ double red[48];
  fista_scalars S;
  RLS_FISTA_COPY(S, *sc);
  if (!S.pending || S.done) {
    if (threadIdx.x < 64) rls_mailbox_publish(mb, S, (int)threadIdx.x);
    return;
  }
  const int tid = threadIdx.x;
  const E* yc = S.ycur ? y1 : y0;
  const E* xc = (S.iteration & 1) ? b1 : b0;
  E raw[EPT], x0v[EPT], yv[EPT], xk[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = tid + (int64_t)e * FIN_THREADS;
    const int64_t ic = i < N ? i : (N - 1);
    raw[e] = res_raw[ic];
    x0v[e] = x0[ic];
    yv[e] = yc[ic];
    xk[e] = xc[ic];
  }
  E ri[EPT], xn[EPT], yn[EPT];
  fista_scalars Sn;
  const bool done = fista_update_elems<E, EPT, FIN_THREADS>(S, raw, x0v, yv, xk, N, red, ri, xn, yn, Sn);
  E* xw = (S.iteration & 1) ? b0 : b1;
  E* yw = S.ycur ? y0 : y1;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = tid + (int64_t)e * FIN_THREADS;
    if (i < N) {
      xw[i] = xn[e];
      res[i] = ri[e];
      if (!done) yw[i] = yn[e];
    }
  }
  Sn.ycur = done ? S.ycur : 1 - S.ycur;
  Sn.pending = 0;
  Sn.fresh = 0;
  __syncthreads();
  if (tid == 0) RLS_FISTA_COPY(*sc, Sn);
  if (tid < 64) rls_mailbox_publish(mb, Sn, tid);
}

// ---- Gram-mode CGNR pipeline: ONE launch per iteration -----------------------------------------
// With an explicit AHA (the reference constructor's default for a dense Matrix, AHA = A' * A,
// src/CGNR.jl:49) the operator apply is one N x N GEMV.  A workgroup owns G*V ROWS of AHA (all columns,
// register slab as above), so its rows of v = AHA p are complete: no partial slab, no reduce kernel.
// Launch i reads the vectors / partial dots / scalars of parity q = i & 1 (written by launch i-1),
// applies the pending CG update in its prologue (every workgroup redundantly, workgroup 0 stores) and
// writes parity q ^ 1; nothing a launch reads is written by the same launch.
template <typename E, int G, int K, int WV>
struct gram_lds {
  E xs[slab_cfg<E, G, K, WV>::NMAX];
  E part[WV][G][elem<E>::vec];
  E tw[G * elem<E>::vec];
  double red[48];
};

// rows of AHA * xs owned by this workgroup (the first product of slab_finish): after the call L.part holds the
// per-wave partial sums, the caller adds them up for its row
template <typename E, int G, int K, int WV, bool FULL>
__device__ static inline void gram_rows(chunk<E, elem<E>::vec> (&a)[K], gram_lds<E, G, K, WV>& L, int64_t Mc, int64_t N,
                                        int pair) {
  using C = slab_cfg<E, G, K, WV>;
  constexpr int NV = C::NV;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane % G, s = lane / G, slot = w * C::S + s;
  const int64_t chunk_id = row_block_of(blockIdx.x, pair) * G + g;
  if constexpr (!FULL) {
    const bool row_ok = chunk_id < Mc;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k * C::CPR + slot >= N || !row_ok) a[k] = zero_chunk<E, NV>();
    }
  }
  __syncthreads();  // xs complete
  E acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = elem<E>::zero();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (k % 8 == 0) __builtin_amdgcn_sched_barrier(0);
    const E xe = L.xs[k * C::CPR + slot];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = elem<E>::fma_pk(a[k].e[i], xe, acc[i]);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int off = G; off < 64; off <<= 1) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float re = elem<E>::re(acc[i]) + __shfl_xor(elem<E>::re(acc[i]), off, 64);
      float im = 0.f;
      if constexpr (elem<E>::cplx) im = elem<E>::im(acc[i]) + __shfl_xor(elem<E>::im(acc[i]), off, 64);
      acc[i] = elem<E>::make(re, im);
    }
  }
  if (s == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) L.part[w][g][i] = acc[i];
  }
  __syncthreads();
}

// the same for a workgroup that walks several row blocks of AHA (more blocks than CUs: ComplexF32 with N in (2048, 4096] has 512 of
// them): `a` holds the block W was aimed at BEFORE its last aim(); with RELOAD every chunk is re-requested for the block W aims at
// now as soon as the product has used it -- the single product is the last reader of the slab registers here (section 4.1b of DESIGN)
template <typename E, int G, int K, int WV, bool FULL, bool RELOAD>
__device__ static __forceinline__ void gram_rows_walk(chunk<E, elem<E>::vec> (&a)[K], gram_lds<E, G, K, WV>& L, bool dead_rows,
                                                      const slab_walk<E, G, K, WV, FULL>& W) {
  using C = slab_cfg<E, G, K, WV>;
  constexpr int NV = C::NV;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, w = tid >> 6;
  const int g = lane % G, s = lane / G, slot = w * C::S + s;
  lds_barrier();  // xs complete / the previous block's reads of part done (LDS only: the loads in flight stay in flight)
  E acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = elem<E>::zero();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (k % (RELOAD ? 2 : 8) == 0) __builtin_amdgcn_sched_barrier(0);
    if constexpr (!FULL) {
      if (W.dead(k, dead_rows)) a[k] = zero_chunk<E, NV>();
    }
    const E xe = L.xs[k * C::CPR + slot];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = elem<E>::fma_pk(a[k].e[i], xe, acc[i]);
    if constexpr (RELOAD) a[k] = W.load(k);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int off = G; off < 64; off <<= 1) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float re = elem<E>::re(acc[i]) + __shfl_xor(elem<E>::re(acc[i]), off, 64);
      float im = 0.f;
      if constexpr (elem<E>::cplx) im = elem<E>::im(acc[i]) + __shfl_xor(elem<E>::im(acc[i]), off, 64);
      acc[i] = elem<E>::make(re, im);
    }
  }
  if (s == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) L.part[w][g][i] = acc[i];
  }
  lds_barrier();
}

// MULTI: gridDim.x < nblocks, every workgroup walks several row blocks of AHA (gram_rows_walk) and leaves ONE set of partial dots
template <typename E, int G, int K, int WV, bool FULL, bool MULTI = false>
__global__ __launch_bounds__(WV * 64) void cgnr_gram_kernel(const E* __restrict__ Gm, int64_t ldg, E* __restrict__ x,
                                                            const E* __restrict__ rc, const E* __restrict__ pc,
                                                            E* __restrict__ rn_out, E* __restrict__ pn_out,
                                                            const E* __restrict__ vc, E* __restrict__ vn,
                                                            const double* __restrict__ dc, double* __restrict__ dn,
                                                            int ndots, const cgnr_scalars* __restrict__ sc,
                                                            cgnr_scalars* __restrict__ scn, int64_t Mc, int64_t N,
                                                            int pair, int order_mode, int nblocks) {
  using C = slab_cfg<E, G, K, WV>;
  constexpr int NV = C::NV, EPT = C::EPT;
  __shared__ gram_lds<E, G, K, WV> L;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const bool writer = blockIdx.x == 0;
  // the small loads go out ahead of the slab (a CU's vector-memory path returns loads in issue order)
  // X_LATE (ComplexF32, 16 rows of 32 columns per workgroup: 8 owned elements of four vectors beside a 128-register slab spilled
  // 36 B per lane): x is not held across the slab load; workgroup 0 -- the only one that stores it -- reads it where it updates it
  constexpr bool X_LATE = elem<E>::cplx && G == 4 && K == 32;
  E pv[EPT], rv[EPT], vv[EPT], xv[X_LATE ? 1 : EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = tid + (int64_t)e * C::NT;
    const int64_t ic = i < N ? i : (N - 1);
    if constexpr (!X_LATE) xv[e] = writer ? x[ic] : elem<E>::zero();  // only workgroup 0 stores x
    pv[e] = pc[ic];
    rv[e] = rc[ic];
    vv[e] = vc[ic];
  }
  const int dtid = tid < ndots ? tid : 0;
  double d0 = dc[4 * dtid], d1 = dc[4 * dtid + 1], d2 = dc[4 * dtid + 2];
  if (order_mode == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    __builtin_amdgcn_s_barrier();
  }
  __builtin_amdgcn_sched_barrier(0);
  chunk<E, NV> a[K];
  slab_walk<E, G, K, WV, FULL> W;  // MULTI only
  if constexpr (MULTI) {
    W.init(Gm, ldg, Mc, N, pair);
    W.aim(blockIdx.x);
#pragma unroll
    for (int k = 0; k < K; ++k) a[k] = W.load(k);
  } else {
    slab_load<E, G, K, WV, FULL>(a, Gm, ldg, Mc, N, pair);
  }
  __builtin_amdgcn_sched_barrier(0);
  if (tid >= ndots) d0 = d1 = d2 = 0.0;
  const cgnr_scalars S = *sc;
  if (S.done) {  // no-op launch: carry the state over to the other parity
    if (writer) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int64_t i = tid + (int64_t)e * C::NT;
        if (i < N) {
          rn_out[i] = rv[e];
          pn_out[i] = pv[e];
          vn[i] = vv[e];
        }
      }
      if (tid == 0) *scn = S;
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = tid + (int64_t)e * C::NT;
    if (i >= N) pv[e] = elem<E>::zero();
  }
  cgnr_scalars Sn;
  if (S.pending) {
    E pn[EPT], rn[EPT], al;
    const bool done = cg_update_elems<E, EPT, C::NT, false, true>(S, d0, d1, d2, pv, rv, vv, N, L.red, pn, rn, al, Sn);
    Sn.pending = done ? 0 : 1;
    if (writer) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int64_t i = tid + (int64_t)e * C::NT;
        if (i < N) {
          if constexpr (X_LATE) x[i] = elem<E>::fma(pv[e], al, x[i]);
          else x[i] = elem<E>::fma(pv[e], al, xv[e]);
          rn_out[i] = rn[e];
          pn_out[i] = pn[e];
          if (done) vn[i] = vv[e];
        }
      }
      if (tid == 0) *scn = Sn;
    }
    if (done) return;  // uniform: every workgroup derived the same scalars
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = tid + e * C::NT;
      if (i < C::NMAX) L.xs[i] = pn[e];
    }
  } else {
    Sn = S;
    Sn.pending = 1;
    if (writer) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int64_t i = tid + (int64_t)e * C::NT;
        if (i < N) {
          rn_out[i] = rv[e];
          pn_out[i] = pv[e];
        }
      }
      if (tid == 0) *scn = Sn;
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = tid + e * C::NT;
      if (i < C::NMAX) L.xs[i] = pv[e];
    }
  }
  // thread t < G*NV owns row (first chunk of the block) * NV + t: its v entry and its term of the dots (added up over the blocks of a
  // walking workgroup in the order they are walked)
  double dre = 0.0, dim_ = 0.0, pp = 0.0;
  auto row_terms = [&](int64_t vb) {
    if (tid < G * NV) {
      const int gg = tid / NV, i = tid % NV;
      E sum = elem<E>::zero();
#pragma unroll
      for (int ww = 0; ww < WV; ++ww) sum = elem<E>::add(sum, L.part[ww][gg][i]);
      const int64_t row = (row_block_of(vb, pair) * G + gg) * NV + i;
      if (row < N) {
        vn[row] = sum;
        const E pj = L.xs[row];
        dre += (double)elem<E>::re(pj) * (double)elem<E>::re(sum) + (double)elem<E>::im(pj) * (double)elem<E>::im(sum);
        dim_ += (double)elem<E>::re(pj) * (double)elem<E>::im(sum) - (double)elem<E>::im(pj) * (double)elem<E>::re(sum);
        pp += (double)elem<E>::re(pj) * (double)elem<E>::re(pj) + (double)elem<E>::im(pj) * (double)elem<E>::im(pj);
      }
    }
  };
  if constexpr (MULTI) {
    int64_t vb = blockIdx.x;
    const int64_t gs = gridDim.x;
    bool dead_rows = !W.row_ok;
    for (; vb + gs < nblocks; vb += gs) {
      W.aim(vb + gs);
      gram_rows_walk<E, G, K, WV, FULL, true>(a, L, dead_rows, W);
      dead_rows = !W.row_ok;
      row_terms(vb);
    }
    gram_rows_walk<E, G, K, WV, FULL, false>(a, L, dead_rows, W);
    row_terms(vb);
  } else {
    gram_rows<E, G, K, WV, FULL>(a, L, Mc, N, pair);
    row_terms(blockIdx.x);
  }
  if (w == 0) {  // G*NV <= 16 lanes of wave 0 hold the terms; fixed-order butterfly
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      dre += __shfl_xor(dre, off, 64);
      dim_ += __shfl_xor(dim_, off, 64);
      pp += __shfl_xor(pp, off, 64);
    }
    if (lane == 0) {
      dn[4 * blockIdx.x] = dre;
      dn[4 * blockIdx.x + 1] = dim_;
      dn[4 * blockIdx.x + 2] = pp;
    }
  }
}

// finish: apply the pending update of parity q and bring x, r, p, v back into the caller's vectors
// (index 0); the scalars are written to both parities so that the next call starts at parity 0
template <typename E, int EPT>
__global__ __launch_bounds__(FIN_THREADS) void cgnr_gram_f_kernel(E* __restrict__ x, const E* rc, const E* pc, E* r0,
                                                                  E* p0, const E* vc, E* v0,
                                                                  const double* __restrict__ dc, int ndots,
                                                                  const cgnr_scalars* sc, cgnr_scalars* sc0,
                                                                  cgnr_scalars* sc1, int64_t N) {
  __shared__ double red[48];
  const int tid = threadIdx.x;
  const cgnr_scalars S = *sc;
  E pv[EPT], rv[EPT], vv[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = tid + (int64_t)e * FIN_THREADS;
    const int64_t ic = i < N ? i : (N - 1);
    pv[e] = pc[ic];
    rv[e] = rc[ic];
    vv[e] = vc[ic];
    if (i >= N) pv[e] = elem<E>::zero();
  }
  double d0 = 0.0, d1 = 0.0, d2 = 0.0;
  for (int t = tid; t < ndots; t += FIN_THREADS) {
    d0 += dc[4 * t];
    d1 += dc[4 * t + 1];
    d2 += dc[4 * t + 2];
  }
  cgnr_scalars Sn = S;
  if (S.pending && !S.done) {
    E pn[EPT], rn[EPT], al;
    cg_update_elems<E, EPT, FIN_THREADS>(S, d0, d1, d2, pv, rv, vv, N, red, pn, rn, al, Sn);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = tid + (int64_t)e * FIN_THREADS;
      if (i < N) {
        x[i] = elem<E>::fma(pv[e], al, x[i]);
        r0[i] = rn[e];
        p0[i] = pn[e];
        v0[i] = vv[e];
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = tid + (int64_t)e * FIN_THREADS;
      if (i < N) {
        r0[i] = rv[e];
        p0[i] = pv[e];
        v0[i] = vv[e];
      }
    }
  }
  Sn.pending = 0;
  Sn.cur = 0;
  Sn.fresh = 0;
  __syncthreads();
  if (tid == 0) {
    *sc0 = Sn;
    *sc1 = Sn;
  }
}


// ---- Gram-mode FISTA: the same one-launch scheme (src/FISTA.jl:139-185 with AHA explicit, :58) ----
// res_raw = AHA y exists in two parities; x / xold and y keep their own ping-pong (iteration parity, ycur).
template <typename E, int G, int K, int WV, bool FULL, bool HINTED, bool MULTI = false>
__global__ __launch_bounds__(WV * 64) void fista_gram_kernel(const E* __restrict__ Gm, int64_t ldg, E* b0, E* b1,
                                                             const E* __restrict__ x0, E* __restrict__ res, E* y0,
                                                             E* y1, const E* __restrict__ rr_cur,
                                                             E* __restrict__ rr_next,
                                                             const fista_scalars* __restrict__ sc,
                                                             fista_scalars* __restrict__ scn, int64_t Mc, int64_t N,
                                                             int pair, int hint, int nblocks) {
  using C = slab_cfg<E, G, K, WV>;
  constexpr int NV = C::NV, EPT = C::EPT;
  __shared__ gram_lds<E, G, K, WV> L;
  const int tid = threadIdx.x;
  const bool writer = blockIdx.x == 0;
  // the scalars go out FIRST: read after the slab loads they would be a vector load the compiler has to
  // wait for with vmcnt(0), i.e. behind the whole slab (measured: 23 us per launch instead of 8)
  fista_scalars S;
  RLS_FISTA_COPY(S, *sc);
  E raw[EPT], x0v[EPT], ya[EPT], yb[EPT], ba[EPT], bb[EPT];
  const E* yh = hint == 1 ? y1 : y0;  // HINTED: the host passed the parity of the iteration count (16 loads, not 24)
  const E* bh = hint == 1 ? b1 : b0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = tid + (int64_t)e * C::NT;
    const int64_t ic = i < N ? i : (N - 1);
    raw[e] = rr_cur[ic];
    x0v[e] = x0[ic];
    if constexpr (HINTED) {
      ya[e] = yb[e] = yh[ic];
      ba[e] = bb[e] = bh[ic];
    } else {
      ya[e] = y0[ic];
      yb[e] = y1[ic];
      ba[e] = b0[ic];
      bb[e] = b1[ic];
    }
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  chunk<E, NV> a[K];
  slab_walk<E, G, K, WV, FULL> W;  // MULTI only
  if constexpr (MULTI) {
    W.init(Gm, ldg, Mc, N, pair);
    W.aim(blockIdx.x);
#pragma unroll
    for (int k = 0; k < K; ++k) a[k] = W.load(k);
  } else {
    slab_load<E, G, K, WV, FULL>(a, Gm, ldg, Mc, N, pair);
  }
  __builtin_amdgcn_sched_barrier(0);
  // No early return below: every path reaches the row product, so the compiler cannot sink part of the
  // slab loads under a branch (it did: 23 us per launch instead of 8); `active` guards the stores instead.
  bool active = true;
  E yv[EPT], xk[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = tid + (int64_t)e * C::NT;
    yv[e] = S.ycur ? yb[e] : ya[e];
    xk[e] = (S.iteration & 1) ? bb[e] : ba[e];  // state.x == buf[iteration & 1]
    if (i >= N) yv[e] = elem<E>::zero();
  }
  if constexpr (HINTED) {
    if (S.ycur != hint || (S.iteration & 1) != hint) {  // wrong hint (never with the host's bookkeeping): re-load, late
      const E* yc = S.ycur ? y1 : y0;
      const E* bc = (S.iteration & 1) ? b1 : b0;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int64_t i = tid + (int64_t)e * C::NT;
        const int64_t ic = i < N ? i : (N - 1);
        yv[e] = yc[ic];
        xk[e] = bc[ic];
        if (i >= N) yv[e] = elem<E>::zero();
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) inside the branch
    }
  }
  fista_scalars Sn;
  if (S.done) {
    if (writer && tid == 0) RLS_FISTA_COPY(*scn, S);
    active = false;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = tid + e * C::NT;
      if (i < C::NMAX) L.xs[i] = yv[e];
    }
  } else if (S.pending) {
    E ri[EPT], xn[EPT], yn[EPT];
    const bool done = fista_update_elems<E, EPT, C::NT, false, true>(S, raw, x0v, yv, xk, N, L.red, ri, xn, yn, Sn);
    if (writer) {
      E* xw = (S.iteration & 1) ? b0 : b1;
      E* yw = S.ycur ? y0 : y1;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int64_t i = tid + (int64_t)e * C::NT;
        if (i < N) {
          xw[i] = xn[e];
          res[i] = ri[e];
          if (!done) yw[i] = yn[e];
        }
      }
    }
    Sn.ycur = done ? S.ycur : 1 - S.ycur;
    Sn.pending = done ? 0 : 1;
    Sn.fresh = 0;
    if (writer && tid == 0) RLS_FISTA_COPY(*scn, Sn);
    if (done) active = false;  // uniform: every workgroup derived the same scalars
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = tid + e * C::NT;
      if (i < C::NMAX) L.xs[i] = yn[e];
    }
  } else {
    RLS_FISTA_COPY(Sn, S);
    Sn.pending = 1;
    Sn.fresh = 0;
    if (writer && tid == 0) RLS_FISTA_COPY(*scn, Sn);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = tid + e * C::NT;
      if (i < C::NMAX) L.xs[i] = yv[e];
    }
  }
  auto row_out = [&](int64_t vb) {
    if (active && tid < G * NV) {
      const int gg = tid / NV, i = tid % NV;
      E sum = elem<E>::zero();
#pragma unroll
      for (int ww = 0; ww < WV; ++ww) sum = elem<E>::add(sum, L.part[ww][gg][i]);
      const int64_t row = (row_block_of(vb, pair) * G + gg) * NV + i;
      if (row < N) rr_next[row] = sum;
    }
  };
  if constexpr (MULTI) {
    int64_t vb = blockIdx.x;
    const int64_t gs = gridDim.x;
    bool dead_rows = !W.row_ok;
    for (; vb + gs < nblocks; vb += gs) {
      W.aim(vb + gs);
      gram_rows_walk<E, G, K, WV, FULL, true>(a, L, dead_rows, W);
      dead_rows = !W.row_ok;
      row_out(vb);
    }
    gram_rows_walk<E, G, K, WV, FULL, false>(a, L, dead_rows, W);
    row_out(vb);
  } else {
    gram_rows<E, G, K, WV, FULL>(a, L, Mc, N, pair);
    row_out(blockIdx.x);
  }
}

template <typename E, int EPT>
__global__ __launch_bounds__(FIN_THREADS) void fista_gram_f_kernel(E* b0, E* b1, const E* __restrict__ x0,
                                                                   E* __restrict__ res, E* y0, E* y1,
                                                                   const E* __restrict__ rr_cur,
                                                                   const fista_scalars* sc, fista_scalars* sc0,
                                                                   fista_scalars* sc1, int64_t N) {
  __shared__ double red[48];
  fista_scalars S;
  RLS_FISTA_COPY(S, *sc);
  const int tid = threadIdx.x;
  fista_scalars Sn;
  RLS_FISTA_COPY(Sn, S);
  if (S.pending && !S.done) {
    const E* yc = S.ycur ? y1 : y0;
    const E* xc = (S.iteration & 1) ? b1 : b0;
    E raw[EPT], x0v[EPT], yv[EPT], xk[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = tid + (int64_t)e * FIN_THREADS;
      const int64_t ic = i < N ? i : (N - 1);
      raw[e] = rr_cur[ic];
      x0v[e] = x0[ic];
      yv[e] = yc[ic];
      xk[e] = xc[ic];
    }
    E ri[EPT], xn[EPT], yn[EPT];
    const bool done = fista_update_elems<E, EPT, FIN_THREADS>(S, raw, x0v, yv, xk, N, red, ri, xn, yn, Sn);
    E* xw = (S.iteration & 1) ? b0 : b1;
    E* yw = S.ycur ? y0 : y1;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = tid + (int64_t)e * FIN_THREADS;
      if (i < N) {
        xw[i] = xn[e];
        res[i] = ri[e];
        if (!done) yw[i] = yn[e];
      }
    }
    Sn.ycur = done ? S.ycur : 1 - S.ycur;
  }
  Sn.pending = 0;
  Sn.fresh = 0;
  __syncthreads();
  if (tid == 0) {
    RLS_FISTA_COPY(*sc0, Sn);
    RLS_FISTA_COPY(*sc1, Sn);
  }
}


// ---- resident CGNR: a whole rls_cgnr_step call in ONE launch, A held in registers across iterations -------
// At the headline shape A is 64 MiB and the chip's register files hold 128 MiB: with one 512-thread workgroup per
// CU each workgroup keeps its 16-row slab of A in VGPRs for the WHOLE solve, so an iteration costs no pass over
// memory at all -- only the two products from registers and two grid-wide exchanges:
//   1. every workgroup publishes its partial row of v = A^H (A p) (write-through stores)   | grid barrier
//   2. workgroup j sums 64-byte column chunk j over all partial rows in a fixed order and
//      publishes that piece of v and its share of <p, v>, ||p||^2                          | grid barrier
//   3. every workgroup reads v and the partial dots and applies the CG update redundantly
//      (identical inputs, identical order => identical alpha, beta, done); r, p, x stay in registers.
// Inter-workgroup visibility follows the guide's rule for in-launch hand-offs: every handed-off byte is stored
// sc1 (write-through) and drained (s_waitcnt vmcnt(0)) by its storing wave, a workgroup barrier, then ONE lane
// adds to the (sharded, monotonic) arrival counter; consumers poll the shards with sc1 loads and read the
// payload with sc1 loads only after a workgroup barrier behind the poll.  Nothing depends on dispatch order
// or XCD placement; every spin is bounded (`spin_limit`), a timeout leaves x, r, p untouched and raises `fail`.
template <typename E, int G, int K, int WV>
struct resident_lds {
  slab_lds<E, G, K, WV> L;
  f4 rp[WV][4];  // per-wave sums of the four 16-byte pieces of a 64-byte column chunk
  int flag;
  float ored[2 * WV * 32];  // owner layout: per-wave sums of t_w, then every wave's own copy of t_w
  double tt;                // flat exchange: the grid sum of the riding scalar (resident_allreduce, SCAL)
};
// dynamic LDS of a resident kernel: its struct, or the 128 KiB staging area of the one-off slab transposition
template <typename E, int G, int K, int WV>
constexpr size_t resident_lds_bytes() {
  return sizeof(resident_lds<E, G, K, WV>) > 131072 || !owner_cfg_ok<E, G, K, WV>() ? sizeof(resident_lds<E, G, K, WV>) : 131072;
}

// workgroup j sums 64-byte column chunk j (and j + nwg, ...) of the partial rows in a fixed order, stores that piece
// of v write-through and hands every summed column to `per_column(j, sum)` (threads 0..CW-1 of wave 0)
template <typename E, int G, int K, int WV, bool FULL = true, typename F>
__device__ static inline void resident_reduce_chunks(resident_lds<E, G, K, WV>& R, __amdgpu_buffer_rsrc_t slab_rs, E* v,
                                                     int nwg, int64_t N, F&& per_column) {
  constexpr int CW = 64 / (int)sizeof(E);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nchunks = (int)((N + CW - 1) / CW);  // !FULL: the last chunk may be short (N is a multiple of the 16-byte piece)
  for (int ch = blockIdx.x; ch < nchunks; ch += nwg) {
    const int piece = lane >> 4, r16 = lane & 15;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool piece_ok = FULL || (int64_t)ch * 64 + piece * 16 < N * (int64_t)sizeof(E);
    for (int row0 = 0; row0 < nwg; row0 += 256) {  // two independent loads per trip (one trip at 256 rows)
      const int ra = row0 + w * 16 + r16, rb = ra + 128;
      const uint32_t col_off = piece_ok ? (uint32_t)ch * 64u + (uint32_t)piece * 16u : 0u;
      const f4 ta = sc1_load16(slab_rs, (uint32_t)(ra < nwg ? ra : 0) * (uint32_t)(N * sizeof(E)) + col_off);
      const f4 tb = sc1_load16(slab_rs, (uint32_t)(rb < nwg ? rb : 0) * (uint32_t)(N * sizeof(E)) + col_off);
      if (ra < nwg && piece_ok) acc += ta;
      if (rb < nwg && piece_ok) acc += tb;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // the 16 rows of this wave: DPP butterfly inside a row of 16 lanes
      float t = acc[q];
      t += dpp_f(t, 0xB1);
      t += dpp_f(t, 0x4E);
      t += dpp_f(t, 0x141);
      t += dpp_f(t, 0x140);
      acc[q] = t;
    }
    if (r16 == 0) R.rp[w][piece] = acc;
    __syncthreads();
    if (tid < CW) {
      E sum = elem<E>::zero();
#pragma unroll
      for (int ww = 0; ww < WV; ++ww) sum = elem<E>::add(sum, reinterpret_cast<const E*>(&R.rp[ww][0])[tid]);
      const int j = ch * CW + tid;
      if (FULL || j < N) {
        sc1_store_elem<E>(v + j, sum);
        per_column(j, sum);
      }
    }
    __syncthreads();  // rp is reused by the next chunk
  }
}

// ---- column-owner layout of the resident slab (round 3) ----------------------------------------------------------------
// The streaming kernels load the slab so that a lane holds NV rows x K columns (coalesced 16-byte pieces of a column); the
// first product then sums over a lane's own columns, but the second needs a sum ACROSS the G lanes that share a column: K
// values per lane through LDS exchange planes, 256 KiB of LDS traffic per iteration -- 2 of the ~3.9 us the products cost.
// A resident kernel loads its slab once per LAUNCH, so it can afford to re-arrange it once: after the transposition below
// thread t holds ALL G * NV rows of the EPT columns it also owns of p, r, x, v (the 16-byte ownership pieces of
// own_index<.., WIDE>).  Per iteration then
//   * t_w = A_w p: the multiplier p[c] is the thread's own register (no staging of p in LDS), G * NV row sums per thread,
//     reduced over the workgroup by a halving butterfly in registers (DPP / bpermute: ~110 instructions per wave) and one
//     small LDS round for the 8 waves;
//   * A_w^H t_w: every column's sum is complete inside its owner thread -- no exchange at all -- and is stored straight into
//     the partial row at the thread's own pieces.
// Same FMA count as before; what disappears is the LDS traffic and four workgroup barriers.  Applies where a thread's
// ownership pieces match the 16-round passes of the transposition (EPT / NV == K / 16): the ComplexF32 shapes and Float32
// N in (2048, 4096]; Float32 N <= 2048 (32 rows x 4 columns per thread in ONE piece) keeps the exchange-plane products.
template <typename E, int G, int K, int WV>
struct owner_cfg {
  using C = slab_cfg<E, G, K, WV>;
  static constexpr bool ok = owner_cfg_ok<E, G, K, WV>();
  static constexpr int RF = G * C::NV * (elem<E>::cplx ? 2 : 1);  // floats of t_w: 32 (16 complex rows / 32 real rows) or 16
};

// once per launch: a[k] (rows NV g.., column (k WV + w) S + s) -> a[16 q + j G + i] = rows NV i.. of owned column (piece q, j).
// Pass q moves the 16 rounds whose columns are piece q through LDS (128 KiB), in place: the 16 registers it empties are
// the 16 it fills.  Slot swizzle: a column's G slots are permuted by its owner's index so that the reads (stride 256 B
// between lanes) spread over G bank groups.
template <typename E, int G, int K, int WV, bool FULL>
__device__ static inline void owner_transpose(chunk<E, elem<E>::vec> (&a)[K], char* lds, int64_t Mc, int64_t N, int pair) {
  using C = slab_cfg<E, G, K, WV>;
  constexpr int NV = C::NV, S = C::S, CPR = C::CPR, NT = C::NT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane % G, s = lane / G, slot = w * S + s;
  if constexpr (!FULL) {
    const bool row_ok = row_block_of(blockIdx.x, pair) * G + g < Mc;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (k * CPR + slot >= N || !row_ok) a[k] = zero_chunk<E, NV>();
  }
  f4* st = reinterpret_cast<f4*>(lds);
#pragma unroll
  for (int q = 0; q < K / 16; ++q) {
    __syncthreads();  // the previous pass has been read
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int cl = kk * CPR + slot;              // column inside the piece
      const int owner = cl / NV;                   // its owner thread
      st[cl * G + (g ^ (owner % G))] = __builtin_bit_cast(f4, a[16 * q + kk]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int i = 0; i < G; ++i)
        a[16 * q + j * G + i] = __builtin_bit_cast(chunk<E, NV>, st[(tid * NV + j) * G + (i ^ (tid % G))]);
  }
  __syncthreads();
  (void)NT;
}

// Sum of v[0..NVAL) over the 64 lanes of a wave by halving: after the step that pairs a lane with its partner it keeps the half
// selected by one bit of its index, so after log2(NVAL) steps it holds ONE value, and the remaining steps are plain adds.
// Round 6: no ds_bpermute at all.  The four steps inside a row of 16 lanes are DPP (full rate): row_mirror (i <-> 15 - i, bit 3
// selects), row_half_mirror (i <-> 7 - i, bit 2), quad_perm [2,3,0,1] (bit 1), quad_perm [1,0,3,2] (bit 0); the two steps ACROSS
// rows are gfx950's v_permlane16_swap / v_permlane32_swap (VALU: the odd rows of the first operand change places with the even
// rows of the second; the upper half with the lower half) -- one swap and one add do a whole halving step: with a = the lower,
// b = the upper value, a' + b' is the pair sum of a in the even rows and of b in the odd ones.  (Rounds 3-5 did the steps of
// distance 4, 8, 16, 32 through ds_bpermute: four dependent LDS round trips per product.)  Returns the value (complete over the
// wave; lanes l and l + 32 hold the same one); *idx_out = its element index.  The order of the additions is a function of the
// lane index alone: bit-reproducible.
// (inline assembly with its own wait states: hipcc 7.2's builtin returns the first result's register for BOTH results, and an
// asm statement is invisible to the compiler's hazard recogniser -- tools/ubench/permlane_probe.hip)
__device__ static inline void permlane16_swap(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ static inline void permlane32_swap(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
template <int NVAL>
__device__ static inline float wave_reduce_scatter(float (&v)[NVAL], int lane, int* idx_out) {
  static_assert(NVAL == 32 || NVAL == 16, "16 or 32 values");
  int idx = 0;
  {
    constexpr int H = NVAL / 2;
    const bool up = lane & 8;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const float lo = v[j] + dpp_f(v[j], 0x140), hi = v[H + j] + dpp_f(v[H + j], 0x140);
      v[j] = up ? hi : lo;
    }
    idx += up ? H : 0;
  }
  {
    constexpr int H = NVAL / 4;
    const bool up = lane & 4;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const float lo = v[j] + dpp_f(v[j], 0x141), hi = v[H + j] + dpp_f(v[H + j], 0x141);
      v[j] = up ? hi : lo;
    }
    idx += up ? H : 0;
  }
  {
    constexpr int H = NVAL / 8;
    const bool up = lane & 2;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const float lo = v[j] + dpp_f(v[j], 0x4E), hi = v[H + j] + dpp_f(v[H + j], 0x4E);
      v[j] = up ? hi : lo;
    }
    idx += up ? H : 0;
  }
  {
    constexpr int H = NVAL / 16;
    const bool up = lane & 1;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const float lo = v[j] + dpp_f(v[j], 0xB1), hi = v[H + j] + dpp_f(v[H + j], 0xB1);
      v[j] = up ? hi : lo;
    }
    idx += up ? H : 0;
  }
  float r;
  if constexpr (NVAL == 32) {  // two values left: the even rows keep the pair sum of v[0], the odd rows that of v[1]
    float a = v[0], b = v[1];
    permlane16_swap(a, b);
    r = a + b;
    idx += (lane & 16) ? 1 : 0;
  } else {                     // one value left: rows 0 + 1 and rows 2 + 3
    float a = v[0], b = v[0];
    permlane16_swap(a, b);
    r = a + b;
  }
  {
    float a = r, b = r;
    permlane32_swap(a, b);
    r = a + b;
  }
  *idx_out = idx;
  return r;
}

// one application of the slab in the owner layout: partial row of A_w^H (A_w pin) -> slab[blockIdx] (write-through, the
// thread's own 16-byte pieces).  pin[]: the thread's owned elements of the input vector (zero beyond N).
// `red`: 2 x WV x RF floats of LDS scratch.
template <typename E, int G, int K, int WV, bool FULL>
__device__ static inline void owner_products(const chunk<E, elem<E>::vec> (&a)[K], const E (&pin)[slab_cfg<E, G, K, WV>::EPT],
                                             float* red, __amdgpu_buffer_rsrc_t slab_rs, int64_t N, bool l2rows = false,
                                             const __amdgpu_buffer_rsrc_t* tt_rs = nullptr) {
  using C = slab_cfg<E, G, K, WV>;
  constexpr int NV = C::NV, EPT = C::EPT, NT = C::NT, RF = owner_cfg<E, G, K, WV>::RF;
  constexpr bool CX = elem<E>::cplx;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // ---- t_w = A_w p: this thread's columns ----
  E acc[G * NV];
#pragma unroll
  for (int i = 0; i < G * NV; ++i) acc[i] = elem<E>::zero();
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int q = e / NV, j = e % NV;
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
      for (int r = 0; r < NV; ++r) acc[i * NV + r] = elem<E>::fma_pk(a[16 * q + j * G + i].e[r], pin[e], acc[i * NV + r]);
  }
  float v[RF];
#pragma unroll
  for (int i = 0; i < G * NV; ++i) {
    if constexpr (CX) {
      v[2 * i] = elem<E>::re(acc[i]);
      v[2 * i + 1] = elem<E>::im(acc[i]);
    } else {
      v[i] = elem<E>::re(acc[i]);
    }
  }
  int idx;
  const float part = wave_reduce_scatter<RF>(v, lane, &idx);
  // (no barrier in front: the previous reader of `red` is the previous application's second product, and an exchange with
  //  several workgroup barriers lies between)
  if (lane < 32) red[w * RF + idx] = part;   // lanes l and l + 32 hold the same element (RF = 16: l, l + 16, ... likewise)
  lds_barrier();
  // Every WAVE adds the WV per-wave sums up for itself (the order ww = 0, 1, ... of the version that had RF threads of the workgroup
  // do it: the same bits) and hands the RF values to its own lanes through a wave-private row of LDS.  That needs no workgroup
  // barrier -- the LDS operations of one wave complete in order -- where the shared copy needs a second one (round 5; full-size
  // instantiations only: two of the ragged ones, at 254-256 registers, spilled 8-12 bytes with it and keep the shared copy).
  const float* tw;
  float tsum = 0.f;  // wave 0, lanes < RF: one component of t_w
  if constexpr (FULL) {
    float* mine = red + (WV + w) * RF;
    if (lane < RF) {
      float sum = red[lane];
#pragma unroll
      for (int ww = 1; ww < WV; ++ww) sum += red[ww * RF + lane];
      mine[lane] = sum;
      tsum = sum;
    }
    __builtin_amdgcn_wave_barrier();  // (compiler only: the reads below stay behind the store above)
    tw = mine;
  } else {
    if (tid < RF) {
      float sum = red[tid];
#pragma unroll
      for (int ww = 1; ww < WV; ++ww) sum += red[ww * RF + tid];
      red[WV * RF + tid] = sum;
      tsum = sum;
    }
    lds_barrier();
    tw = red + WV * RF;
  }
  // ||t_w||^2 (Float64, fixed order) -> word blockIdx of the scalar row: the workgroup's share of ||A p||^2 = <p, A^H A p>, which
  // the exchange sums along with the partial rows (src/CGNR.jl:153-154's dot(p, v) without a reduction over p and v behind the
  // exchange: the CGLS form of alpha).  Stored at the scope of the partial rows; the caller's drain covers it.
  if (tt_rs) {
    if (w == 0) {
      const double sq = half_wave_sum((double)tsum * (double)tsum);
      if (lane == 0) {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        if (l2rows) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, sq), *tt_rs, (uint32_t)blockIdx.x * 8u, 0, 1);
        else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, sq), *tt_rs, (uint32_t)blockIdx.x * 8u, 0, 16);
      }
    }
  }
  // ---- A_w^H t_w: complete inside the owner thread ----
  E tr[G * NV];
#pragma unroll
  for (int i = 0; i < G * NV; ++i) {
    if constexpr (CX) tr[i] = elem<E>::make(tw[2 * i], tw[2 * i + 1]);
    else tr[i] = elem<E>::make(tw[i], 0.f);
  }
#pragma unroll
  for (int q = 0; q < EPT / NV; ++q) {
    chunk<E, NV> out;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      E sum = elem<E>::zero();
#pragma unroll
      for (int i = 0; i < G; ++i)
#pragma unroll
        for (int r = 0; r < NV; ++r) sum = elem<E>::fmac_pk(a[16 * q + j * G + i].e[r], tr[i * NV + r], sum);
      out.e[j] = sum;
    }
    const int o = q * NT * NV + tid * NV;
    if (FULL || o < N) {
      const uint32_t off = (uint32_t)blockIdx.x * (uint32_t)(N * sizeof(E)) + (uint32_t)(o * sizeof(E));
      // l2rows (uniform; resident_rows_at_l2 below): this row is read by members of this workgroup's group only, and they share
      // its XCD's L2 -- the store stops there (sc0) instead of travelling to the memory side (sc1)
      if (l2rows) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, out), slab_rs, off, 0, 1);
      else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, out), slab_rs, off, 0, 16);
    }
  }
}

// The two-level exchange's FIRST hop stays inside a group (the workgroups with equal blockIdx % RES_GROUPS), and under the
// dispatcher's round-robin placement a group is the set of workgroups on ONE XCD: they share its L2.  So the partial rows a
// workgroup hands to its group need not be written through to the memory side: stored with sc0 they stop in the L2 (the store's
// acknowledgement -- which the hand-off waits for -- comes from there), and the members' sc1 loads (which bypass the CU's L1 as
// before) find them there.  tools/ubench/grid_barrier, arithmetic stripped: 7.4 -> 5.2 us per exchange.  The placement is CHECKED,
// not assumed: every workgroup compares HW_REG_XCC_ID with blockIdx % RES_GROUPS and reports a mismatch in a word of the sync
// block (zeroed per launch) ahead of its first arrival; the launch's first exchange runs write-through, and behind its grid
// barrier every workgroup reads the word -- the same decision everywhere, for the rest of the launch (a kernel that stays in
// server mode keeps it: workgroups do not move).  Rows of one XCD are never read through another XCD's L2, and the
// end of the kernel writes the L2 back, so the next launch may decide differently.
__device__ static inline unsigned resident_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xfu;
}
__device__ static inline void resident_report_placement(resident_sync* sync) {
  if (threadIdx.x == 0 && resident_xcc_id() != (blockIdx.x % RES_GROUPS))
    __hip_atomic_store(&sync->gcnt[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (a spare word of group 0's counter line)
}
__device__ static inline bool resident_rows_at_l2(resident_sync* sync) {  // behind a grid barrier of this launch
  return uni((int)__hip_atomic_load(&sync->gcnt[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0;
}

// The grid-wide sum of the workgroups' partial rows (slab[nwg][N]: just stored write-through and drained by every storing
// wave, a workgroup barrier behind the drains), delivered to EVERY workgroup as the 16-byte pieces vv[] its threads own.
//   EXCH 1 (any grid): grid barrier | workgroup j sums 64-byte column chunk j over all nwg rows, publishes it to v (plus
//     whatever `publish` adds: CGNR's partial dots) | grid barrier | every workgroup reads v.
//   EXCH 2 (two levels; nwg a multiple of RES_GROUPS, the row a power-of-two number of 16-byte pieces per group member):
//     group barrier (the nwg / 8 workgroups with equal blockIdx % 8) | member l sums slice l of the row over its group's
//     rows and publishes that group-partial slice | ONE grid barrier | every workgroup reads the 8 group-partial vectors
//     and adds them in group order.  One full-grid barrier per exchange instead of two at the price of an 8x larger final
//     read (128 KiB per workgroup at the headline shape).  tools/ubench/grid_barrier.hip, arithmetic stripped, 256
//     workgroups, 16 KiB rows: EXCH 1 10.4 us, EXCH 2 7.4 us per exchange (groups of 32 strided by 8; 9.5 us with
//     contiguous groups, 8.4 / 8.5 us with 4 / 16 groups; the bare grid barrier 1.56 us).
// Everything handed over is sc1-stored and sc1-loaded in both variants, so which workgroups share an XCD changes speed
// only; the group partials alternate between two buffers (a fast group's members may publish exchange k + 1 while a slow
// group still reads exchange k; k + 2 cannot start before every workgroup has passed the grid barrier of k + 1).
// Summation orders are functions of (nwg, N) alone: bit-reproducible.  Returns false when a wait ran into its bound.
// DPPSUM: compile the headline arrangement's slice sum in DPP rows (see there); POGM with restart, at 255 registers, spills 8 bytes with the
// second code path and leaves it out
// SCAL: one Float64 scalar per workgroup rides along (word blockIdx of the scalar row `d_rs`, stored by the caller at the scope of
// its partial row and drained with it): *tt = their sum over the grid, the same bits in every thread of every workgroup.
//   EXCH 1: behind the first grid barrier the last wave of every workgroup adds all nwg words up itself (lane l: words l, l + 64, ...,
//     then the wave butterfly) and leaves the sum in LDS; the second grid barrier's workgroup barrier hands it to the other waves.
//   EXCH 2: behind the group barrier wave 0 of the group's member 0 adds the group's words (lane l: member l) and publishes the
//     group's sum behind the group partials; behind the grid barrier every lane loads group (lane & 7)'s sum and three DPP steps
//     inside the 8 lanes add them.  No workgroup barrier, no LDS, nothing but one 8-byte load on the critical path.
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__device__ static inline double sc1_load_f64(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 16));
}
template <typename E, int G, int K, int WV, int EXCH, bool FULL, bool DPPSUM = true, bool SCAL = false, typename PC, typename PUB>
__device__ static inline bool resident_allreduce(resident_lds<E, G, K, WV>& R, resident_sync* sync,
                                                 __amdgpu_buffer_rsrc_t slab_rs, E* v, int nwg, int64_t N, unsigned& epoch,
                                                 unsigned& xchg, unsigned spin_limit,
                                                 E (&vv)[slab_cfg<E, G, K, WV>::EPT], PC&& per_column, PUB&& publish,
                                                 const __amdgpu_buffer_rsrc_t* d_rs = nullptr, double* tt = nullptr) {
  using C = slab_cfg<E, G, K, WV>;
  constexpr int NV = C::NV, EPT = C::EPT, NT = C::NT;
  const int tid = threadIdx.x;
  if constexpr (EXCH == 1) {
    if (!grid_arrive_wait(sync->cnt, ++epoch, (unsigned)nwg, spin_limit, &R.flag)) return false;
    STAMP(11);
    if constexpr (SCAL) {
      if ((tid >> 6) == WV - 1) {
        const int lane = tid & 63;
        double s = 0.0;
        for (int row = lane; row < nwg; row += 64) s += sc1_load_f64(*d_rs, (uint32_t)row * 8u);
        s = wave_sum(s);
        if (lane == 0) R.tt = s;
      }
    }
    resident_reduce_chunks<E, G, K, WV, FULL>(R, slab_rs, v, nwg, N, per_column);
    publish();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAMP(12);
    if (!grid_arrive_wait(sync->cnt, ++epoch, (unsigned)nwg, spin_limit, &R.flag)) return false;
    STAMP(13);
    const __amdgpu_buffer_rsrc_t v_rs = sc1_rsrc(v);
#pragma unroll
    for (int q = 0; q < EPT / NV; ++q) {
      const int o = q * NT * NV + tid * NV;
      const bool ok = FULL || o < N;
      const chunk<E, NV> c = __builtin_bit_cast(chunk<E, NV>, sc1_load16(v_rs, (uint32_t)((ok ? o : 0) * sizeof(E))));
#pragma unroll
      for (int j = 0; j < NV; ++j) vv[q * NV + j] = ok ? c.e[j] : elem<E>::zero();
    }
    if constexpr (SCAL) *tt = R.tt;
    return true;
  } else {
    const unsigned per = (unsigned)nwg / RES_GROUPS, grp = blockIdx.x % RES_GROUPS, mem = blockIdx.x / RES_GROUPS;
    const uint32_t rowb = (uint32_t)(N * sizeof(E));
    const unsigned q = (rowb / 16u) / per;                   // 16-byte pieces of the row per group member: a power of two
    const unsigned lq = (unsigned)__builtin_ctz(q), nrg = (unsigned)NT >> lq;
    const unsigned par = xchg & 1u;
    ++xchg;
    char* xpart = reinterpret_cast<char*>(sync) + sizeof(resident_sync) + (size_t)par * RES_GROUPS * rowb;
    const __amdgpu_buffer_rsrc_t xp_rs = sc1_rsrc(xpart);
    if (!group_arrive_wait(sync->gcnt + grp * 32, per * xchg, spin_limit, &R.flag)) return false;
    STAMP(11);
    // the group sums of the riding scalar: [2 parities][RES_GROUPS] doubles behind the group partials
    const uint32_t gs_off = (2u - par) * RES_GROUPS * rowb + par * (RES_GROUPS * 8u);
    if constexpr (SCAL) {
      if (mem == 0u && tid < 64) {  // (per <= 32: the grid holds at most 256 workgroups)
        double s = (unsigned)tid < per ? sc1_load_f64(*d_rs, (grp + RES_GROUPS * (unsigned)tid) * 8u) : 0.0;
        s = half_wave_sum(s);
        if (tid == 0) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, s), xp_rs, gs_off + grp * 8u, 0, 16);
      }
    }
    if (DPPSUM && nrg == 16u && q == 32u && per == 32u) {
      // The headline arrangement (256 workgroups, 16 KiB rows: 32 pieces per member, 32 rows): the 16 row groups of a piece sit in
      // the 16 lanes of ONE DPP row, so their sum is four full-rate cross-lane additions per component -- no LDS hand-over, no
      // workgroup barrier in front of a 16-term sum that 32 threads ran through one LDS round trip after the other (round 5).
      // Wave w: pieces 4w .. 4w + 3, one per DPP row; lane & 15 = the row group (rows rg and rg + 16 of the group's 32).
      const unsigned lane = (unsigned)tid & 63u, rg = lane & 15u, piece = 4u * ((unsigned)tid >> 6) + (lane >> 4);
      const uint32_t col = (mem * 32u + piece) * 16u;
      const f4 ta = sc1_load16(slab_rs, (grp + RES_GROUPS * rg) * rowb + col);
      const f4 tb = sc1_load16(slab_rs, (grp + RES_GROUPS * (rg + 16u)) * rowb + col);
      f4 acc = ta + tb;
#pragma unroll
      for (int c = 0; c < 4; ++c) {   // fixed order: pairs, quads, half rows, the row
        float v = acc[c];
        v += dpp_f(v, 0xB1);
        v += dpp_f(v, 0x4E);
        v += dpp_f(v, 0x141);
        v += dpp_f(v, 0x140);
        acc[c] = v;
      }
      if (rg == 0u) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, acc), xp_rs, grp * rowb + (mem * 32u + piece) * 16u, 0, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave has stored: each drains its own, then the barrier
      __syncthreads();
    } else {
      const unsigned piece = (unsigned)tid & (q - 1u), rg = (unsigned)tid >> lq;
      const uint32_t col = (mem * q + piece) * 16u;
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      for (unsigned row = rg; row < per; row += 2u * nrg) {  // two loads in flight per trip (one trip at 32 rows, q = 32)
        const unsigned rb = row + nrg;
        const f4 ta = sc1_load16(slab_rs, (grp + RES_GROUPS * row) * rowb + col);
        const f4 tb = sc1_load16(slab_rs, (grp + RES_GROUPS * (rb < per ? rb : row)) * rowb + col);
        acc += ta;
        if (rb < per) acc += tb;
      }
      f4* ex = reinterpret_cast<f4*>(&R.L.xg[0][0]);  // the exchange planes of the second product are idle here
      ex[tid] = acc;                                  // [rg][piece]
      __syncthreads();
      if ((unsigned)tid < q) {
        f4 sum = ex[tid];
        for (unsigned g = 1; g < nrg; ++g) sum += ex[(g << lq) + tid];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, sum), xp_rs, grp * rowb + (mem * q + (unsigned)tid) * 16u, 0, 16);
      }
      // Threads tid < q have stored.  When q <= 64 that is wave 0 alone, the wave that arrives at the grid barrier below: its own
      // drain in front of its own arrival is program order -- no workgroup barrier; the other waves go straight to the barrier inside
      // grid_arrive_wait, and what they read next (the flag) is written by wave 0 only.  More pieces per member keep the barrier.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (q > 64u) __syncthreads();
    }
    STAMP(12);
    if (!grid_arrive_wait(sync->cnt, ++epoch, (unsigned)nwg, spin_limit, &R.flag)) return false;
    STAMP(13);
    if constexpr (SCAL) {
      double s = sc1_load_f64(xp_rs, gs_off + ((unsigned)tid & 7u) * 8u);
      s += dpp_d(s, 0xB1);
      s += dpp_d(s, 0x4E);
      s += dpp_d(s, 0x141);
      *tt = s;
    }
#pragma unroll
    for (int qq = 0; qq < EPT / NV; ++qq) {
      const int o = qq * NT * NV + tid * NV;
      const bool ok = FULL || o < N;
      const uint32_t off = (uint32_t)((ok ? o : 0) * sizeof(E));
      f4 t[RES_GROUPS];
#pragma unroll
      for (int g = 0; g < RES_GROUPS; ++g) t[g] = sc1_load16(xp_rs, (uint32_t)g * rowb + off);
      f4 sum = t[0];
#pragma unroll
      for (int g = 1; g < RES_GROUPS; ++g) sum += t[g];
      const chunk<E, NV> c = __builtin_bit_cast(chunk<E, NV>, sum);
#pragma unroll
      for (int j = 0; j < NV; ++j) vv[qq * NV + j] = ok ? c.e[j] : elem<E>::zero();
    }
    return true;
  }
}

// SPEC (server mode only): the kernel runs ONE iteration ahead of the command that asks for it.  Behind the status and write-back of
// command k it computes iteration k + 1 at once -- under the host's turnaround -- and only then listens; the next command finds its
// first iteration done (nothing of it published or written back before the command is there).  Told to leave instead, the kernel
// leaves WITHOUT a write-back: memory holds the state of command k, which is what the host was told.  A separate instantiation: the
// plain kernel's code is the SPEC = false text, token for token.
template <typename E, int G, int K, int WV, int BAR, bool FULL, bool SPEC = false>
__global__ __launch_bounds__(WV * 64) void cgnr_resident_kernel(const E* __restrict__ A, int64_t lda, E* x, E* xw, E* r, E* p,
                                                                 E* v, E* slab, double* dout, cgnr_scalars* sc,
                                                                 resident_sync* sync, int64_t Mc, int64_t N, int pair,
                                                                 int n_steps, unsigned spin_limit, rls_cg_start St) {
  using C = slab_cfg<E, G, K, WV>;
  constexpr int NV = C::NV, EPT = C::EPT, NT = C::NT;
  static_assert(EPT % NV == 0, "16-byte ownership layout");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  resident_lds<E, G, K, WV>& R = *reinterpret_cast<resident_lds<E, G, K, WV>*>(smem_raw);
  slab_lds<E, G, K, WV>& L = R.L;
  const int tid = threadIdx.x;
  const int nwg = gridDim.x;
  STAMP(0);  // (launch-level stamps 0..4, tools/stamps_resident.py: entry, slab in owner layout, loop entered, loop left, written back)
  if (St.enabled && St.skip && *St.skip) {  // the ADMM plan is done: this cg! is a no-op (uniform: every workgroup reads the flag)
    if (blockIdx.x == 0 && tid == 0) {
      cgnr_scalars Z = *sc;
      Z.iteration = 0;
      Z.max_iter = St.maxiter;
      Z.pending = 0;
      Z.cur = 0;
      Z.fresh = 0;
      Z.done = 1;
      *sc = Z;
      sync->completed = 1u;
    }
    return;
  }
  cgnr_scalars S = *sc;
  // this thread's elements of the length-N vectors, in 16-byte pieces (own_index<.., WIDE = true>)
  E pv[EPT], rv[EPT], xv[EPT];
  if constexpr (FULL) {
    load_owned_wide<E, EPT, NT>(pv, p, tid);
    load_owned_wide<E, EPT, NT>(rv, r, tid);
    load_owned_wide<E, EPT, NT>(xv, x, tid);
  } else {  // ragged N (a multiple of the 16-byte piece): zeros beyond N, and they stay zeros through the update
    load_owned_wide_masked<E, EPT, NT>(pv, p, tid, N);
    load_owned_wide_masked<E, EPT, NT>(rv, r, tid, N);
    load_owned_wide_masked<E, EPT, NT>(xv, x, tid, N);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  chunk<E, NV> a[K];
  slab_load<E, G, K, WV, FULL>(a, A, lda, Mc, N, pair);
  __builtin_amdgcn_sched_barrier(0);
  if (!St.enabled && (S.done || n_steps <= 0)) return;  // uniform
  constexpr bool OWN = owner_cfg<E, G, K, WV>::ok;  // the slab re-arranged once so that a thread holds whole columns
  if constexpr (OWN) owner_transpose<E, G, K, WV, FULL>(a, smem_raw, Mc, N, pair);
  STAMP(1);
  const __amdgpu_buffer_rsrc_t slab_rs = sc1_rsrc(slab), d_rs = sc1_rsrc(dout);  // dout: the scalar row (word blockIdx = ||t_w||^2)
  unsigned epoch = 0, xchg = 0;
  bool alive = true;
  // partial rows at L2 scope (resident_rows_at_l2): decided behind the launch's first exchange, which runs write-through
  constexpr bool L2ROWS = BAR == 2 && OWN;
  bool l2rows = false, placed = !L2ROWS;  // uniform
  if constexpr (L2ROWS) resident_report_placement(sync);
  if (St.enabled) {
    // ---- cg! entry (cg_pipe_start_kernel of solvers.hip, folded in): c = AHA x through the same two exchanges, then
    // r = b - (c + rho x), p = r and the scalars of the solve, redundantly in every workgroup ----------------------------
    if constexpr (OWN) {
      owner_products<E, G, K, WV, FULL>(a, xv, R.ored, slab_rs, N);
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) L.xs[(int)own_index<E, EPT, NT, true>(tid, e)] = xv[e];
      slab_finish<E, G, K, WV, FULL, true>(a, L, slab, Mc, N, pair);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    E cv[EPT];
    if (!resident_allreduce<E, G, K, WV, BAR, FULL>(R, sync, slab_rs, v, nwg, N, epoch, xchg, spin_limit, cv, [](int, E) {}, []() {})) {
      resident_give_up(sync, St.enabled ? St.poison : nullptr);
      return;
    }
    if (!placed) {
      l2rows = resident_rows_at_l2(sync);
      placed = true;
    }
    double rr = 0.0;
#pragma unroll
    for (int q = 0; q < EPT / NV; ++q) {
      const int o = q * NT * NV + tid * NV;
      const bool ok = FULL || o < N;
      const int oc = ok ? o : 0;
      chunk<E, NV> cc;
#pragma unroll
      for (int j = 0; j < NV; ++j) cc.e[j] = cv[q * NV + j];
      chunk<E, NV> bi;
      if (St.beta_y) {  // beta = beta_y + rho (z - u)   (src/ADMM.jl:236-241), stored with xold = x by workgroup 0
        bi = load_chunk<E, NV>(reinterpret_cast<const E*>(St.beta_y) + oc);
        const chunk<E, NV> zc = load_chunk<E, NV>(reinterpret_cast<const E*>(St.z) + oc);
#pragma unroll
        for (int j = 0; j < NV; ++j) bi.e[j] = elem<E>::add(bi.e[j], elem<E>::scale(St.rho_admm, zc.e[j]));
        const chunk<E, NV> uc = load_chunk<E, NV>(reinterpret_cast<const E*>(St.u) + oc);
#pragma unroll
        for (int j = 0; j < NV; ++j) bi.e[j] = elem<E>::add(bi.e[j], elem<E>::scale(-St.rho_admm, uc.e[j]));
        if (blockIdx.x == 0 && ok) {
          chunk<E, NV> xc;
#pragma unroll
          for (int j = 0; j < NV; ++j) xc.e[j] = xv[q * NV + j];
          *reinterpret_cast<f4*>(reinterpret_cast<E*>(St.beta) + o) = __builtin_bit_cast(f4, bi);
          *reinterpret_cast<f4*>(reinterpret_cast<E*>(St.xold) + o) = __builtin_bit_cast(f4, xc);
        }
      } else {
        bi = load_chunk<E, NV>(reinterpret_cast<const E*>(St.b) + oc);
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const E ci = elem<E>::add(cc.e[j], elem<E>::scale(St.rho, xv[q * NV + j]));
        E ri = elem<E>::sub(bi.e[j], ci);
        if (!ok) ri = elem<E>::zero();
        rv[q * NV + j] = ri;
        pv[q * NV + j] = ri;
        rr += (double)elem<E>::re(ri) * (double)elem<E>::re(ri) + (double)elem<E>::im(ri) * (double)elem<E>::im(ri);
      }
    }
    rr = block_sum_n<NT / 64>(rr, L.red);
    S.rr = rr;
    S.z0 = sqrt(rr);
    S.zeta = 0.0;
    S.alpha_re = S.alpha_im = S.beta_re = S.beta_im = 0.0;
    S.lambda = St.rho;
    S.rel_tol = St.reltol;
    S.iteration = 0;
    S.max_iter = St.maxiter;
    S.pending = 0;
    S.cur = 0;
    S.fresh = 0;
    S.done = (St.maxiter <= 0) || (rr == 0.0) || (1.0f <= St.reltol);
    if (S.done) n_steps = 0;  // uniform
  }
  // x is not part of the recurrence (only x += alpha p touches it) and only workgroup 0 writes it back: it lives in plan
  // scratch `xw` between iterations -- read behind the products, advanced by workgroup 0 -- instead of in 8 registers that
  // would be live across the products (the kernel sat at 256 VGPRs and spilled without this).  The caller's x is written
  // once, at the end, so a launch that gives up still leaves it untouched.
  auto store_owned = [&](E* dst, const E (&src)[EPT]) {
#pragma unroll
    for (int q = 0; q < EPT / NV; ++q) {
      chunk<E, NV> c;
#pragma unroll
      for (int j = 0; j < NV; ++j) c.e[j] = src[q * NV + j];
      const int64_t o = (int64_t)q * (NT * NV) + (int64_t)tid * NV;
      if (FULL || o < N) *reinterpret_cast<f4*>(dst + o) = __builtin_bit_cast(f4, c);
    }
  };
  // read back with L1-bypassing loads: the thread re-reads what IT stored an iteration earlier (drained long since), and an
  // L1 line of this CU must not stand in for it
  const __amdgpu_buffer_rsrc_t xw_rs = sc1_rsrc(xw);
  auto load_owned_sc1 = [&](E (&dst)[EPT], const E*) {
#pragma unroll
    for (int q = 0; q < EPT / NV; ++q) {
      const int o = q * NT * NV + tid * NV;
      const bool ok = FULL || o < N;
      const chunk<E, NV> c = __builtin_bit_cast(chunk<E, NV>, sc1_load16(xw_rs, (uint32_t)((ok ? o : 0) * sizeof(E))));
#pragma unroll
      for (int j = 0; j < NV; ++j) dst[q * NV + j] = ok ? c.e[j] : elem<E>::zero();
    }
  };
  if (blockIdx.x == 0) store_owned(xw, xv);
  rls_mailbox_slot srv_mb = St.srv_mb;
  unsigned srv_seq = St.srv_seq0;  // the command being served; the host's next one carries srv_seq + 1
  int credit = 0;      // SPEC: iterations of the current command that were computed ahead of it
  bool ahead = false;  // SPEC: the pass below runs ahead of its command
  STAMP(2);
  for (;;) {  // (server mode: one pass per command; otherwise one pass)
  for (int it = SPEC ? credit : 0; it < n_steps; ++it) {
    if (S.done) break;  // uniform (a command behind the one that reached the stopping test)
    STAMP(8);
#ifdef RLS_STAMPS
    if (it == 1) STAMP(5);   // (how long the first iterations of a launch take: tools/stamps_resident.py)
    if (it == 2) STAMP(6);
    if (it == 10) STAMP(7);
#endif
    // t_w = A_w p, partial v = A_w^H t_w -> this workgroup's partial row (write-through); ||t_w||^2 -> its word of the scalar row
    if constexpr (OWN) {
      owner_products<E, G, K, WV, FULL>(a, pv, R.ored, slab_rs, N, l2rows, &d_rs);
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) L.xs[(int)own_index<E, EPT, NT, true>(tid, e)] = pv[e];
      const double ttw = slab_finish<E, G, K, WV, FULL, true, true>(a, L, slab, Mc, N, pair);
      if (tid == 0) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, ttw), d_rs, (uint32_t)blockIdx.x * 8u, 0, 16);
    }
    STAMP(9);
    // ||p||^2 (lambda > 0 only; uniform): reduced here, under the flight of the partial row's stores -- not behind the exchange
    double pp = 0.0;
    if (S.lambda > 0.f) {
      if constexpr (!OWN) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) pv[e] = L.xs[(int)own_index<E, EPT, NT, true>(tid, e)];
      }
#pragma unroll
      for (int e = 0; e < EPT; ++e)
        pp += (double)elem<E>::re(pv[e]) * (double)elem<E>::re(pv[e]) + (double)elem<E>::im(pv[e]) * (double)elem<E>::im(pv[e]);
      pp = block_sum_nolead<NT / 64, 0>(pp, L.red);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its own stores
    __syncthreads();
    STAMP(10);
    E xq[EPT];  // x, requested here and consumed behind the exchange (every workgroup: no branch around a load)
    load_owned_sc1(xq, xw);
    // ---- v = the sum of the partial rows and ||A p||^2 = the sum of the ||t_w||^2, in every workgroup ----------------------
    E vv[EPT];
    double tt = 0.0;
    const bool ok_x = resident_allreduce<E, G, K, WV, BAR, FULL, true, true>(R, sync, slab_rs, v, nwg, N, epoch, xchg, spin_limit, vv,
                                                                             [](int, E) {}, []() {}, &d_rs, &tt);
    if (!ok_x) {
      alive = false;
      break;
    }
    if (!placed) {
      l2rows = resident_rows_at_l2(sync);
      placed = true;
    }
    // p of this iteration, back from its LDS copy (L.xs, staged for the products and untouched since): it need not occupy
    // registers across the products and the exchange
    if constexpr (!OWN) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) pv[e] = L.xs[(int)own_index<E, EPT, NT, true>(tid, e)];
    }
    E pn[EPT], rn[EPT];
    float al;
    cgnr_scalars Sn;
#ifdef RLS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(14);
#endif
    const bool done = cg_update_elems_tt<E, EPT, NT, true>(S, tt, pp, pv, rv, vv, N, L.red, pn, rn, al, Sn);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      xq[e] = elem<E>::make(fmaf(elem<E>::re(pv[e]), al, elem<E>::re(xq[e])), fmaf(elem<E>::im(pv[e]), al, elem<E>::im(xq[e])));
      rv[e] = rn[e];
      pv[e] = pn[e];
    }
    if (blockIdx.x == 0) {  // buffer stores: the 32-bit offsets of the loads above, no 64-bit address pairs kept across the loop
#pragma unroll
      for (int q = 0; q < EPT / NV; ++q) {
        chunk<E, NV> c;
#pragma unroll
        for (int j = 0; j < NV; ++j) c.e[j] = xq[q * NV + j];
        const int o = q * NT * NV + tid * NV;
        if (FULL || o < N) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, c), xw_rs, (uint32_t)(o * sizeof(E)), 0, 0);
      }
    }
    S = Sn;
    RLS_CGNR_UNIFORM(S);
    STAMP(15);
    if (done) break;  // uniform: every workgroup derived the same scalars
  }
  STAMP(3);
  if (!alive) {
    resident_give_up(sync, St.enabled ? St.poison : nullptr);
    if (St.srv_ctl && blockIdx.x == 0 && tid == 0) __hip_atomic_store(St.srv_ctl + 17, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;  // x, r, p and the scalars are untouched: the call was a no-op
  }
  if constexpr (SPEC) {
    if (ahead) {  // that pass ran ahead: its command first (uniform)
      ahead = false;
      const unsigned cmd = resident_listen(St.srv_ctl, srv_seq, St.srv_idle_us, sync, epoch, (unsigned)nwg, spin_limit, &R.flag, srv_mb, srv_seq - St.srv_seq0 + 1u);
      if (cmd == RLS_SRV_EXIT) return;  // (memory holds the state of the last command served: nothing of the pass ahead was stored)
      n_steps = (int)cmd;
      credit = 1;
      continue;
    }
  }
  if (blockIdx.x == 0) {
    S.pending = 0;
    S.cur = 0;
    S.fresh = 0;
    // server mode: the status of this command goes to the host FIRST -- the write-back below then runs under the host's turnaround.
    // (Whoever wants x, r, p from memory asks the kernel to leave first, rls_enter, and it leaves behind its write-back.)
    if (St.srv_ctl && tid < 64) rls_mailbox_publish(srv_mb, S, tid);
    E xf[EPT];
    load_owned_sc1(xf, xw);
    // buffer stores (a descriptor in SGPRs + a 32-bit lane offset): nothing per-lane has to survive the loop for them
    auto store_buf = [&](E* dst, const E (&src)[EPT]) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0xffffffff, 0x00020000);
#pragma unroll
      for (int q = 0; q < EPT / NV; ++q) {
        chunk<E, NV> c;
#pragma unroll
        for (int j = 0; j < NV; ++j) c.e[j] = src[q * NV + j];
        const int o = q * NT * NV + tid * NV;
        if (FULL || o < N) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, c), rs, (uint32_t)(o * sizeof(E)), 0, 0);
      }
    };
    store_buf(x, xf);
    store_buf(r, rv);
    store_buf(p, pv);
    if (tid == 0) {
      *sc = S;
      sync->completed = 1u;
    }
  }
  if (!St.srv_ctl) break;  // uniform
  if constexpr (SPEC) {
    credit = 0;
    if (!S.done) {  // uniform: one iteration ahead of the next command
      ahead = true;
      n_steps = 1;
      continue;
    }
  }
  // ---- server mode: listen for the next command (resident_listen, resident_sync.hpp) ----------------------------------------
  const unsigned cmd = resident_listen(St.srv_ctl, srv_seq, St.srv_idle_us, sync, epoch, (unsigned)nwg, spin_limit, &R.flag, srv_mb, srv_seq - St.srv_seq0 + 1u);
  if (cmd == RLS_SRV_EXIT) return;  // uniform (told to leave, left idle, or a wait ran out: the control block says which)
  n_steps = (int)cmd;
  }
}


// ---- resident Gram-mode CGNR: a whole rls_cgnr_step call (or a whole cg! solve) in ONE launch, AHA in registers --------
// With AHA explicit a workgroup's rows of v = AHA p are complete (no partial rows), so an iteration needs ONE grid-wide
// exchange instead of the two of cgnr_resident_kernel: every workgroup publishes its 8 (complex) / 16 (real) entries of
// v and its share of <p, v>, ||p||^2 (write-through), one barrier, then every workgroup reads v and the partial dots and
// applies the CG update redundantly.  v and the dots alternate between two parities: a workgroup that has passed
// barrier k publishes iteration k + 1 into the other buffer while a slower one still reads iteration k.  Same
// visibility protocol, bounded spins and fail-as-a-no-op behaviour as the matrix-free resident kernel; same element
// ownership and summation orders as cgnr_gram_kernel.
// SRV: the instantiation that can stay and listen (server mode, rls_cgnr_step_status); not built for Float32 K = 32, which
// has no registers to spare for it (it spilled 148-316 B per lane)
template <typename E, int K, int BAR, bool FULL, int SRV = 0>  // SRV: 0 one pass, 1 stays and listens, 2 ... and runs one iteration ahead
__global__ __launch_bounds__(512) void cgnr_gram_resident_kernel(const E* __restrict__ Gm, int64_t ldg, E* x, E* r, E* p,
                                                                  E* v0, E* v1, double* dots, cgnr_scalars* sc0,
                                                                  cgnr_scalars* sc1, resident_sync* sync, int64_t Mc,
                                                                  int64_t N, int pair, int n_steps, unsigned spin_limit,
                                                                  rls_cg_start St) {
  constexpr int G = 4, WV = 8;
  using C = slab_cfg<E, G, K, WV>;
  constexpr int NV = C::NV, EPT = C::EPT, NT = C::NT;
  __shared__ gram_lds<E, G, K, WV> L;
  __shared__ int flag;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nwg = gridDim.x;
  if (St.enabled && St.skip && *St.skip) {  // the ADMM plan is done: this cg! is a no-op (uniform: every workgroup reads the flag)
    if (blockIdx.x == 0 && tid == 0) {
      cgnr_scalars Z = *sc0;
      Z.iteration = 0;
      Z.max_iter = St.maxiter;
      Z.pending = 0;
      Z.cur = 0;
      Z.fresh = 0;
      Z.done = 1;
      *sc0 = Z;
      *sc1 = Z;
      sync->completed = 1u;
    }
    return;
  }
  cgnr_scalars S = *sc0;
  E pv[EPT], rv[EPT], xv[EPT], vv[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {  // FULL: N == EPT * NT, every index is valid; else clamped loads, zeroed tails
    const int64_t i = tid + (int64_t)e * NT;
    const int64_t ic = FULL || i < N ? i : (N - 1);
    pv[e] = p[ic];
    rv[e] = r[ic];
    xv[e] = x[ic];
    vv[e] = elem<E>::zero();
    if (!FULL && i >= N) pv[e] = rv[e] = xv[e] = elem<E>::zero();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  chunk<E, NV> a[K];
  slab_load<E, G, K, WV, FULL>(a, Gm, ldg, Mc, N, pair);
  __builtin_amdgcn_sched_barrier(0);
  if (!St.enabled && (S.done || n_steps <= 0)) return;  // uniform
  const __amdgpu_buffer_rsrc_t d_rs = sc1_rsrc(dots);
  unsigned epoch = 0;
  bool alive = true;
  if (St.enabled) {
    // ---- cg! entry: c = AHA x (one more product + exchange, parity-1 buffer), r = b - (c + rho x), p = r ------------------
    // (cg_pipe_start_kernel of solvers.hip, folded in: no separate operator apply, no start kernel, no reload of r and p)
#pragma unroll
    for (int e = 0; e < EPT; ++e) L.xs[tid + e * NT] = xv[e];
    gram_rows<E, G, K, WV, FULL>(a, L, Mc, N, pair);
    if (tid < G * NV) {
      const int gg = tid / NV, i = tid % NV;
      E sum = elem<E>::zero();
#pragma unroll
      for (int ww = 0; ww < WV; ++ww) sum = elem<E>::add(sum, L.part[ww][gg][i]);
      const int64_t row = (row_block_of(blockIdx.x, pair) * G + gg) * NV + i;
      if (FULL || row < N) sc1_store_elem<E>(v1 + row, sum);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!grid_arrive_wait(sync->cnt, ++epoch, (unsigned)nwg, spin_limit, &flag)) {
      resident_give_up(sync, St.enabled ? St.poison : nullptr);
      return;
    }
    const E* bb = reinterpret_cast<const E*>(St.b);
    const E* by = reinterpret_cast<const E*>(St.beta_y);
    const E* zz = reinterpret_cast<const E*>(St.z);
    const E* uu = reinterpret_cast<const E*>(St.u);
    double rr = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = tid + (int64_t)e * NT;
      const bool ok = FULL || i < N;
      const int64_t ic = ok ? i : (N - 1);
      const E ci = elem<E>::add(sc1_load_elem<E>(v1 + ic), elem<E>::scale(St.rho, xv[e]));
      E bi;
      if (by) {  // beta = beta_y + rho (z - u)   (src/ADMM.jl:236-241), stored with xold = x by workgroup 0
        bi = by[ic];
        bi = elem<E>::add(bi, elem<E>::scale(St.rho_admm, zz[ic]));
        bi = elem<E>::add(bi, elem<E>::scale(-St.rho_admm, uu[ic]));
        if (blockIdx.x == 0 && ok) {
          reinterpret_cast<E*>(St.beta)[i] = bi;
          reinterpret_cast<E*>(St.xold)[i] = xv[e];
        }
      } else {
        bi = bb[ic];
      }
      E ri = elem<E>::sub(bi, ci);
      if (!ok) ri = elem<E>::zero();
      rv[e] = ri;
      pv[e] = ri;
      rr += (double)elem<E>::re(ri) * (double)elem<E>::re(ri) + (double)elem<E>::im(ri) * (double)elem<E>::im(ri);
    }
    rr = block_sum_n<NT / 64>(rr, L.red);
    S.rr = rr;
    S.z0 = sqrt(rr);
    S.zeta = 0.0;
    S.alpha_re = S.alpha_im = S.beta_re = S.beta_im = 0.0;
    S.lambda = St.rho;
    S.rel_tol = St.reltol;
    S.iteration = 0;
    S.max_iter = St.maxiter;
    S.pending = 0;
    S.cur = 0;
    S.fresh = 0;
    S.done = (St.maxiter <= 0) || (rr == 0.0) || (1.0f <= St.reltol);
    if (S.done) n_steps = 0;  // uniform: nothing to iterate; the state below is written back as it stands
  }
  rls_mailbox_slot srv_mb = St.srv_mb;
  unsigned srv_seq = St.srv_seq0;  // server mode (rls_cgnr_step_status): the command being served
  unsigned itg = 0;                // iterations run by this launch, over all its commands: the parity of v and of the dots
  int credit = 0;                  // SRV == 2: iterations of the current command computed ahead of it (as cgnr_resident_kernel's SPEC)
  bool ahead = false;              // ... the pass below runs ahead of its command
  for (;;) {  // (server mode: one pass per command; otherwise one pass)
  for (int it = SRV ? credit : 0; it < n_steps; ++it) {
    if (SRV && S.done) break;  // uniform (a command behind the one that reached the stopping test)
    const int q = (int)(itg++ & 1u);
    E* vq = q ? v1 : v0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) L.xs[tid + e * NT] = pv[e];
    gram_rows<E, G, K, WV, FULL>(a, L, Mc, N, pair);
    double dre = 0.0, dim_ = 0.0, pp = 0.0;
    if (tid < G * NV) {
      const int gg = tid / NV, i = tid % NV;
      E sum = elem<E>::zero();
#pragma unroll
      for (int ww = 0; ww < WV; ++ww) sum = elem<E>::add(sum, L.part[ww][gg][i]);
      const int64_t row = (row_block_of(blockIdx.x, pair) * G + gg) * NV + i;
      if (FULL || row < N) {
        sc1_store_elem<E>(vq + row, sum);
        const E pj = L.xs[row];
        dre = (double)elem<E>::re(pj) * (double)elem<E>::re(sum) + (double)elem<E>::im(pj) * (double)elem<E>::im(sum);
        dim_ = (double)elem<E>::re(pj) * (double)elem<E>::im(sum) - (double)elem<E>::im(pj) * (double)elem<E>::re(sum);
        pp = (double)elem<E>::re(pj) * (double)elem<E>::re(pj) + (double)elem<E>::im(pj) * (double)elem<E>::im(pj);
      }
    }
    if (w == 0) {  // G*NV <= 16 lanes of wave 0 hold the terms; fixed-order butterfly
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        dre += __shfl_xor(dre, off, 64);
        dim_ += __shfl_xor(dim_, off, 64);
        pp += __shfl_xor(pp, off, 64);
      }
      if (lane < 3) {
        const double dv = lane == 0 ? dre : (lane == 1 ? dim_ : pp);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(dots + (size_t)q * 4 * nwg + 4 * blockIdx.x + lane),
                           __builtin_bit_cast(unsigned long long, dv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its own stores
    __syncthreads();
    if (!grid_arrive_wait(sync->cnt, ++epoch, (unsigned)nwg, spin_limit, &flag)) {
      alive = false;
      break;
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = tid + (int64_t)e * NT;
      vv[e] = sc1_load_elem<E>(vq + (FULL || i < N ? i : (N - 1)));
      if (!FULL && i >= N) vv[e] = elem<E>::zero();
    }
    double d0 = 0.0, d1 = 0.0, d2 = 0.0;
    {
      const int dt = tid < nwg ? tid : 0;
      const uint32_t base = (uint32_t)((size_t)q * 4 * nwg * sizeof(double));
      const f4 lo = sc1_load16(d_rs, base + (uint32_t)dt * 32u), hi = sc1_load16(d_rs, base + (uint32_t)dt * 32u + 16u);
      if (tid < nwg) {
        d0 = __builtin_bit_cast(double, __builtin_shufflevector(lo, lo, 0, 1));
        d1 = __builtin_bit_cast(double, __builtin_shufflevector(lo, lo, 2, 3));
        d2 = __builtin_bit_cast(double, __builtin_shufflevector(hi, hi, 0, 1));
      }
    }
    E pn[EPT], rn[EPT], al;
    cgnr_scalars Sn;
    const bool done = cg_update_elems<E, EPT, NT, FULL, true>(S, d0, d1, d2, pv, rv, vv, N, L.red, pn, rn, al, Sn);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      xv[e] = elem<E>::fma(pv[e], al, xv[e]);
      rv[e] = rn[e];
      pv[e] = pn[e];
    }
    S = Sn;
    if (done) break;  // uniform: every workgroup derived the same scalars
  }
  if (!alive) {
    resident_give_up(sync, St.enabled ? St.poison : nullptr);
    if (SRV && St.srv_ctl && blockIdx.x == 0 && tid == 0) __hip_atomic_store(St.srv_ctl + 17, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;  // x, r, p, v and the scalars are untouched: the call was a no-op
  }
  if constexpr (SRV != 0) {
    if (ahead) {  // that pass ran ahead: its command first (uniform); told to leave, memory holds the last command served
      ahead = false;
      const unsigned cmd = resident_listen(St.srv_ctl, srv_seq, St.srv_idle_us, sync, epoch, (unsigned)nwg, spin_limit, &flag, srv_mb,
                                           srv_seq - St.srv_seq0 + 1u);
      if (cmd == RLS_SRV_EXIT) return;
      n_steps = (int)cmd;
      credit = 1;
      continue;
    }
  }
  if (blockIdx.x == 0) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = tid + (int64_t)e * NT;
      if (FULL || i < N) {
        x[i] = xv[e];
        r[i] = rv[e];
        p[i] = pv[e];
        // the last v (a workgroup still reading parity 0 reads the same values).  A kernel that stays and listens must not leave
        // these lines DIRTY in this XCD's L2: v0 is an exchange buffer, a later command's write-through rows from other XCDs
        // would be overwritten whenever the stale lines are evicted
        // SRV == 2 with a pass ahead to follow: NOT here -- no grid barrier lies between this write-back and that pass, whose
        // iteration publishes its rows into the buffer of the other parity than the last one: v0 whenever the last v sits in v1,
        // and this store of the whole vector raced with the other workgroups' rows (one run in three gave another x).  Nothing
        // reads v between launches while nothing is pending (S.pending = 0 below).
        if constexpr (SRV != 0) {
          if (!(SRV == 2 && St.srv_ctl && !S.done)) sc1_store_elem<E>(v0 + i, vv[e]);
        } else {
          v0[i] = vv[e];
        }
      }
    }
    S.pending = 0;
    S.cur = 0;
    S.fresh = 0;
    if (tid == 0) {
      *sc0 = S;
      *sc1 = S;
      sync->completed = 1u;
    }
    if constexpr (SRV != 0) {
      if (St.srv_ctl && tid < 64) rls_mailbox_publish(srv_mb, S, tid);  // the status of this command, straight to the host
    }
  }
  if constexpr (SRV != 0) {
    if (!St.srv_ctl) break;  // uniform
    credit = 0;
    if (SRV == 2 && !S.done) {  // uniform: one iteration ahead of the next command, under the host's turnaround
      ahead = true;
      n_steps = 1;
      continue;
    }
    const unsigned cmd = resident_listen(St.srv_ctl, srv_seq, St.srv_idle_us, sync, epoch, (unsigned)nwg, spin_limit, &flag, srv_mb,
                                         srv_seq - St.srv_seq0 + 1u);
    if (cmd == RLS_SRV_EXIT) return;  // uniform
    n_steps = (int)cmd;
  } else {
    break;
  }
  }
}


// ---- resident FISTA: the same scheme for src/FISTA.jl:139-185 (BASELINE configs[1] has the headline shape) -------
// Per iteration: xs = y, partial rows of AHA y, exchange 1, chunk sums -> res_raw, exchange 2, then the gradient step,
// prox, restart test, theta and the next extrapolated point redundantly in every workgroup (fista_update_elems: its
// two scalar sums run over full vectors every workgroup holds, so no partial dots travel).
// SPEC (server mode only): one iteration ahead of the command that asks for it, as cgnr_resident_kernel's SPEC.  state.res is the
// caller's array (src/FISTA.jl:131 reads it): the pass ahead leaves its residual in plan scratch (the second half of raw_g) and workgroup
// 0 copies it behind the status of the command that asks for that iteration (kept in registers instead, the two-level instantiations
// spilled 12-36 B per lane).
template <typename E, int G, int K, int WV, int BAR, bool FULL, bool SPEC = false>
__global__ __launch_bounds__(WV * 64) void fista_resident_kernel(const E* __restrict__ A, int64_t lda, E* b0, E* b1,
                                                                  const E* __restrict__ x0, E* res, E* y0, E* y1,
                                                                  E* raw_g, E* slab, fista_scalars* sc,
                                                                  resident_sync* sync, int64_t Mc, int64_t N, int pair_flags,
                                                                  int n_steps, unsigned spin_limit, rls_srv_args Sv) {
  using C = slab_cfg<E, G, K, WV>;
  constexpr int NV = C::NV, EPT = C::EPT, NT = C::NT;
  static_assert(EPT % NV == 0, "16-byte ownership layout");
  const int pair = pair_flags & 1;
  const bool defer_on = (pair_flags & 2) == 0;  // rls_tune_set("fista_defer", 0): the measurement switch of DEFER below (uniform)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  resident_lds<E, G, K, WV>& R = *reinterpret_cast<resident_lds<E, G, K, WV>*>(smem_raw);
  slab_lds<E, G, K, WV>& L = R.L;
  const int tid = threadIdx.x;
  const int nwg = gridDim.x;
  fista_scalars S;
  RLS_FISTA_COPY(S, *sc);
  // Loop-carried in registers: y, x, xold (the recurrence).  NOT carried: x0 = A^H b (read-only: re-read every iteration
  // behind the products, so that it is not live while the slab's 128 registers and the product's temporaries are) and res
  // (output only: workgroup 0 stores it every iteration) -- 16 registers less across the products, no spill.
  // DEFER (round 5; the full-size column-owner instantiations on the two-level exchange): without gradient restart nothing of an
  // iteration depends on ||res|| but the stopping test (theta follows a data-independent recursion, src/FISTA.jl:179-180), so
  // the waves leave their partial ||res||^2 in LDS WITHOUT a barrier and every thread adds the eight up behind the NEXT
  // iteration's exchange, where a stop found late drops that iteration's exchange (nothing of it has been applied) -- as
  // fista_gram_resident_kernel does.  The registers for the second update path come from x_{k-1}: it is not part of the
  // recurrence (only the write-back wants it), so workgroup 0 keeps it in plan scratch (`raw_g`, idle under this exchange)
  // instead of every thread carrying it across the products.
  // (ComplexF32 only: the Float32 column-owner instantiation, N in (2048, 4096], spilled 12 B per lane with the second path)
  constexpr bool DEFER = owner_cfg<E, G, K, WV>::ok && FULL && BAR == 2 && elem<E>::cplx;
  E yv[EPT], xk[EPT], xp[EPT];
  if constexpr (FULL) {
    load_owned_wide<E, EPT, NT>(yv, S.ycur ? y1 : y0, tid);
    load_owned_wide<E, EPT, NT>(xk, (S.iteration & 1) ? b1 : b0, tid);  // state.x == buf[iteration & 1]
    load_owned_wide<E, EPT, NT>(xp, (S.iteration & 1) ? b0 : b1, tid);
  } else {
    load_owned_wide_masked<E, EPT, NT>(yv, S.ycur ? y1 : y0, tid, N);
    load_owned_wide_masked<E, EPT, NT>(xk, (S.iteration & 1) ? b1 : b0, tid, N);
    load_owned_wide_masked<E, EPT, NT>(xp, (S.iteration & 1) ? b0 : b1, tid, N);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  chunk<E, NV> a[K];
  slab_load<E, G, K, WV, FULL>(a, A, lda, Mc, N, pair);
  __builtin_amdgcn_sched_barrier(0);
  if (S.done || n_steps <= 0) return;  // uniform
  constexpr bool OWN = owner_cfg<E, G, K, WV>::ok;
  if constexpr (OWN) owner_transpose<E, G, K, WV, FULL>(a, smem_raw, Mc, N, pair);
  const __amdgpu_buffer_rsrc_t slab_rs = sc1_rsrc(slab);
  unsigned epoch = 0, xchg = 0;
  bool alive = true;
  // partial rows at L2 scope (resident_rows_at_l2); not in the masked instantiation, which has no register to spare (12 B spilled)
  constexpr bool L2ROWS = BAR == 2 && OWN && FULL;
  bool l2rows = false, placed = !L2ROWS;    // uniform
  if constexpr (L2ROWS) resident_report_placement(sync);
  int ycur = S.ycur;
  rls_mailbox_slot srv_mb = Sv.mb;
  unsigned srv_seq = Sv.seq0;  // server mode (rls_fista_step_status): the command being served
  const __amdgpu_buffer_rsrc_t xo_rs = sc1_rsrc(raw_g);
  auto stash_xold = [&](const E (&src)[EPT]) {  // workgroup 0: x_{k-1} of the write-back lives in plan scratch (DEFER)
#pragma unroll
    for (int q = 0; q < EPT / NV; ++q) {
      chunk<E, NV> c;
#pragma unroll
      for (int j = 0; j < NV; ++j) c.e[j] = src[q * NV + j];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, c), xo_rs, (uint32_t)((q * NT * NV + tid * NV) * sizeof(E)), 0, 0);
    }
  };
  if constexpr (DEFER) {
    if (blockIdx.x == 0) stash_xold(xp);
  }
  bool pend = false;   // the norm of the last applied iteration is still in LDS (uniform)
  unsigned itg = 0;    // iterations applied by this launch over all its commands: the parity of the norm slots
  auto resolve = [&](unsigned par) {  // src/FISTA.jl:156, :187-189 for that iteration; slots [24, 32) / [40, 48) of L.red
    double rn = 0.0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) rn += L.red[24 + 16 * par + i];
    const double res_norm = sqrt(rn);
    const float rel = (float)(res_norm / S.norm_x0);
    S.res_norm = res_norm;
    S.rel_res_norm = (double)rel;
    if (rel < S.rel_tol) S.done = 1;
    pend = false;
    RLS_FISTA_UNIFORM(S);
  };
  int credit = 0;            // SPEC: iterations of the current command that were computed ahead of it
  bool ahead = false;        // SPEC: the pass below runs ahead of its command
  bool res_parked = false;   // SPEC: state.res of the command's last iteration is still in plan scratch (uniform)
  for (;;) {  // (server mode: one pass per command; otherwise one pass)
  for (int it = SPEC ? credit : 0; it < n_steps; ++it) {
    if (S.done) break;  // uniform (a command behind the one that reached the stopping test)
    STAMP(8);  // (the slots of cgnr_resident_kernel: tools/stamps_resident.py fista)
    if constexpr (OWN) {
      owner_products<E, G, K, WV, FULL>(a, yv, R.ored, slab_rs, N, l2rows);
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) L.xs[(int)own_index<E, EPT, NT, true>(tid, e)] = yv[e];
      slab_finish<E, G, K, WV, FULL, true>(a, L, slab, Mc, N, pair);
    }
    STAMP(9);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAMP(10);
    E x0v[EPT];  // requested here, consumed behind the exchange
    if constexpr (FULL) load_owned_wide<E, EPT, NT>(x0v, x0, tid);
    else load_owned_wide_masked<E, EPT, NT>(x0v, x0, tid, N);
    E raw[EPT];
    if (!resident_allreduce<E, G, K, WV, BAR, FULL>(R, sync, slab_rs, raw_g, nwg, N, epoch, xchg, spin_limit, raw, [](int, E) {}, []() {})) {
      alive = false;
      break;
    }
    if (!placed) {
      l2rows = resident_rows_at_l2(sync);
      placed = true;
    }
    // y of this iteration, back from its LDS copy (L.xs): not carried in registers across the products and the exchange
    if constexpr (!OWN) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) yv[e] = L.xs[(int)own_index<E, EPT, NT, true>(tid, e)];
    }
    if constexpr (DEFER) {
      if (pend) {  // uniform: the previous iteration's stopping test, behind this exchange's barriers
        resolve((itg & 1u) ^ 1u);
        if (S.done) break;  // it had converged: this iteration's exchange is dropped, nothing of it was applied
      }
    }
#ifdef RLS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(14);
#endif
    E xn[EPT], yn[EPT], ri[EPT];
    bool done;
    if (DEFER && defer_on && !S.restart) {  // uniform
      const float rho = S.rho, thr = S.rho * S.lambda;  // prox!(reg, x, rho * lambda(reg))        :164
      double rn = 0.0;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const E r = elem<E>::sub(raw[e], x0v[e]);                             // res .-= x0      :153
        E xv = elem<E>::sub(yv[e], elem<E>::scale(rho, r));                   // x .-= rho .* res :154
        xv = fista_proj_elem<E>(fista_prox_elem<E>(xv, S.reg_kind, thr), S.proj_kind);
        ri[e] = r;
        xn[e] = xv;
        rn += (double)elem<E>::re(r) * (double)elem<E>::re(r) + (double)elem<E>::im(r) * (double)elem<E>::im(r);
      }
      rn = wave_sum(rn);
      if ((tid & 63) == 0) L.red[24 + 16 * (itg & 1u) + (tid >> 6)] = rn;  // read behind the next exchange, or the barrier at the end
      pend = true;
      const float theta_old = S.theta;                                        // :179
      const float theta = (1.f + sqrtf(1.f + 4.f * theta_old * theta_old)) / 2.f;  // :180
      const float c1 = (1.f - theta_old) / theta, c2 = (theta_old - 1.f) / theta + 1.f;
#pragma unroll
      for (int e = 0; e < EPT; ++e) yn[e] = elem<E>::add(elem<E>::scale(c1, xk[e]), elem<E>::scale(c2, xn[e]));
      S.theta = theta;
      S.theta_old = theta_old;
      S.iteration += 1;
      S.done = S.iteration >= S.max_iter;  // the relTol half of :187-189 follows with the norm
      done = S.done != 0;
    } else {
      fista_scalars Sn;
      done = fista_update_elems<E, EPT, NT, true, true>(S, raw, x0v, yv, xk, N, L.red, ri, xn, yn, Sn);
      RLS_FISTA_COPY(S, Sn);
    }
    ++itg;
    if constexpr (SPEC) res_parked = ahead;
    // (SPEC: workgroup 1 -- every workgroup holds the same residual -- so that workgroup 0, which the others wait for at the next
    //  exchange, has the write-back of x, x_{k-1}, y alone)
    if (blockIdx.x == (SPEC && nwg > 1 ? 1 : 0)) {  // state.res of this iteration (nothing reads it back: a launch that gives up later loses nothing)
      // (SPEC, the pass ahead: into plan scratch -- the caller's array changes with the command that asks for this iteration)
      const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(SPEC && ahead ? raw_g + N : res, 0, 0xffffffff, 0x00020000);
#pragma unroll
      for (int q = 0; q < EPT / NV; ++q) {
        chunk<E, NV> c3;
#pragma unroll
        for (int j = 0; j < NV; ++j) c3.e[j] = ri[q * NV + j];
        const int o = q * NT * NV + tid * NV;
        if (FULL || o < N) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, c3), res_rs, (uint32_t)(o * sizeof(E)), 0, 0);
      }
    }
    if constexpr (DEFER) {
      if (blockIdx.x == 0) stash_xold(xk);  // x_k becomes x_{k-1}
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      if constexpr (!DEFER) xp[e] = xk[e];
      xk[e] = xn[e];
      if (!done) yv[e] = yn[e];
    }
    if (!done) ycur ^= 1;
    RLS_FISTA_UNIFORM(S);
    STAMP(15);
    if (done) break;  // uniform
  }
  if constexpr (DEFER) {
    if (alive && pend) {  // uniform: the last applied iteration's norm and stopping test
      lds_barrier();
      resolve((itg - 1u) & 1u);
    }
  }
  if (!alive) {
    resident_give_up(sync, nullptr);
    if (Sv.ctl && blockIdx.x == 0 && tid == 0) __hip_atomic_store(Sv.ctl + 17, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  if constexpr (SPEC) {
    if (ahead) {  // that pass ran ahead: its command first (uniform); told to leave, memory holds the last command served
      ahead = false;
      const unsigned cmd = resident_listen(Sv.ctl, srv_seq, Sv.idle_us, sync, epoch, (unsigned)nwg, spin_limit, &R.flag, srv_mb, srv_seq - Sv.seq0 + 1u);
      if (cmd == RLS_SRV_EXIT) return;
      n_steps = (int)cmd;
      credit = 1;
      continue;
    }
  }
  if (blockIdx.x == 0) {
    if constexpr (DEFER) {  // x_{k-1} back from plan scratch (this thread's own stores, drained first)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < EPT / NV; ++q) {
        const chunk<E, NV> c = __builtin_bit_cast(chunk<E, NV>, sc1_load16(xo_rs, (uint32_t)((q * NT * NV + tid * NV) * sizeof(E))));
#pragma unroll
        for (int j = 0; j < NV; ++j) xp[q * NV + j] = c.e[j];
      }
    }
    S.ycur = ycur;
    S.pending = 0;
    S.fresh = 0;
    if (Sv.ctl && tid < 64) rls_mailbox_publish(srv_mb, S, tid);  // (server mode: status first, write-back under the host's turnaround)
    E* xw = (S.iteration & 1) ? b1 : b0;   // state.x == buf[iteration & 1] afterwards as well
    E* xo = (S.iteration & 1) ? b0 : b1;
    E* yw = ycur ? y1 : y0;
#pragma unroll
    for (int q = 0; q < EPT / NV; ++q) {
      chunk<E, NV> c0, c1, c2;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        c0.e[j] = xk[q * NV + j];
        c1.e[j] = xp[q * NV + j];
        c2.e[j] = yv[q * NV + j];
      }
      const int64_t o = (int64_t)q * (NT * NV) + (int64_t)tid * NV;
      if (FULL || o < N) {
        *reinterpret_cast<f4*>(xw + o) = __builtin_bit_cast(f4, c0);
        *reinterpret_cast<f4*>(xo + o) = __builtin_bit_cast(f4, c1);
        *reinterpret_cast<f4*>(yw + o) = __builtin_bit_cast(f4, c2);
      }
    }
    if (tid == 0) {
      RLS_FISTA_COPY(*sc, S);
      sync->completed = 1u;
    }
  }
  if constexpr (SPEC) {
    // uniform: the command's last iteration was the one computed ahead -- its residual, out of plan scratch (this thread's own stores,
    // drained first), into the caller's array; the next listen lies behind it
    if (res_parked && blockIdx.x == (nwg > 1 ? 1 : 0)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const __amdgpu_buffer_rsrc_t rp_rs = sc1_rsrc(raw_g + N);
      const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(res, 0, 0xffffffff, 0x00020000);
#pragma unroll
      for (int q = 0; q < EPT / NV; ++q) {
        const int o = q * NT * NV + tid * NV;
        if (FULL || o < N) {
          const f4 c = sc1_load16(rp_rs, (uint32_t)(o * sizeof(E)));
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, c), res_rs, (uint32_t)(o * sizeof(E)), 0, 0);
        }
      }
    }
    res_parked = false;
  }
  if (!Sv.ctl) break;  // uniform
  if constexpr (SPEC) {
    credit = 0;
    if (!S.done) {  // uniform: one iteration ahead of the next command
      ahead = true;
      n_steps = 1;
      continue;
    }
  }
  const unsigned cmd = resident_listen(Sv.ctl, srv_seq, Sv.idle_us, sync, epoch, (unsigned)nwg, spin_limit, &R.flag, srv_mb, srv_seq - Sv.seq0 + 1u);
  if (cmd == RLS_SRV_EXIT) return;  // uniform
  n_steps = (int)cmd;
  }
}

// ---- resident OptISTA / POGM: a whole block of iterations in ONE launch ------------------------------------
// Same skeleton as fista_resident_kernel on the column-owner layout: per iteration res_raw = AHA x from registers (one
// grid-wide all-reduce), then the elementwise half of iterate (src/OptISTA.jl:176-204 / src/POGM.jl:176-212: the bodies of
// optista_update_kernel / pogm_update_body in pgm.hip) redundantly in every workgroup.  The momentum coefficients depend on
// the iteration index only: the host computes them in Float32 as the reference does and hands the block's worth over as a
// kernel argument.  Loop-carried in registers: three vectors (x, y, z); zold / xold and res are written every iteration by
// workgroup 0 and never read (a launch that gives up loses nothing: x, y, z and the record are written once, at the end).
// POGM swaps its x / y references every iteration (:203): the launch leaves the operator input of the NEXT iteration in the
// buffer the host's reference `x` points to after as many swaps as iterations ran.
template <typename E, int G, int K, int WV, int BAR, bool FULL, int KIND>
__global__ __launch_bounds__(WV * 64) void pgm_resident_kernel(const E* __restrict__ A, int64_t lda, E* b0, E* b1, E* b2, E* b3, E* o0,
                                                                E* res, const E* __restrict__ x0, E* raw_g, E* slab,
                                                                pgm_state* st, rls_pgm_coefs CF, float norm_x0, float rel_tol,
                                                                int reg_kind, int proj_kind, resident_sync* sync, int64_t Mc,
                                                                int64_t N, int pair_flags, int n_steps, int first_it,
                                                                unsigned spin_limit) {
  using C = slab_cfg<E, G, K, WV>;
  constexpr int NV = C::NV, EPT = C::EPT, NT = C::NT;
  static_assert(owner_cfg<E, G, K, WV>::ok, "column-owner layout only");
  const int pair = pair_flags & 1;
  const bool defer_on = (pair_flags & 2) == 0;  // rls_tune_set("fista_defer", 0): measurement switch of DEFER below (uniform)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  resident_lds<E, G, K, WV>& R = *reinterpret_cast<resident_lds<E, G, K, WV>*>(smem_raw);
  const int tid = threadIdx.x;
  const int nwg = gridDim.x;
  int iteration = st->iteration;
  int done = st->done;
  float res_norm = st->res_norm;
  E xv[EPT], yv[EPT], zv[EPT];
  E wv[KIND == 2 ? EPT : 1];  // POGM's w (gradient restart), loop-carried like x, y, z
  // KIND 2: theta, sigma, gamma of src/POGM.jl:183-232 are loop-carried uniform scalars (the record's words 4..7)
  float theta = 1.f, theta_old = 1.f, sigma = 1.f, gamma = 1.f;
  if constexpr (KIND == 2) {
    const pogm_auto_state* ast = reinterpret_cast<const pogm_auto_state*>(st);
    theta = ast->theta;
    theta_old = ast->theta_old;
    sigma = ast->sigma;
    gamma = ast->gamma;
  }
  if constexpr (FULL) {
    load_owned_wide<E, EPT, NT>(xv, b0, tid);
    load_owned_wide<E, EPT, NT>(yv, b1, tid);
    load_owned_wide<E, EPT, NT>(zv, b2, tid);
    if constexpr (KIND == 2) load_owned_wide<E, EPT, NT>(wv, b3, tid);
  } else {
    load_owned_wide_masked<E, EPT, NT>(xv, b0, tid, N);
    load_owned_wide_masked<E, EPT, NT>(yv, b1, tid, N);
    load_owned_wide_masked<E, EPT, NT>(zv, b2, tid, N);
    if constexpr (KIND == 2) load_owned_wide_masked<E, EPT, NT>(wv, b3, tid, N);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  chunk<E, NV> a[K];
  slab_load<E, G, K, WV, FULL>(a, A, lda, Mc, N, pair);
  __builtin_amdgcn_sched_barrier(0);
  // uniform.  `iteration != first_it`: an earlier launch of this sequence was lost -- the coefficients of this one belong
  // to later iterations, so it must not run (the host finishes the sequence launch by launch, pgm.hip)
  if (done || n_steps <= 0 || iteration != first_it) return;
  owner_transpose<E, G, K, WV, FULL>(a, smem_raw, Mc, N, pair);
  const __amdgpu_buffer_rsrc_t slab_rs = sc1_rsrc(slab);
  const __amdgpu_buffer_rsrc_t o0_rs = __builtin_amdgcn_make_buffer_rsrc(o0, 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(res, 0, 0xffffffff, 0x00020000);
  auto store_buf = [&](__amdgpu_buffer_rsrc_t rs, const E (&src)[EPT]) {
#pragma unroll
    for (int q = 0; q < EPT / NV; ++q) {
      chunk<E, NV> c;
#pragma unroll
      for (int j = 0; j < NV; ++j) c.e[j] = src[q * NV + j];
      const int o = q * NT * NV + tid * NV;
      if (FULL || o < N) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, c), rs, (uint32_t)(o * sizeof(E)), 0, 0);
    }
  };
  unsigned epoch = 0, xchg = 0;
  bool alive = true;
  int ran = 0;
  constexpr bool L2ROWS = BAR == 2 && (FULL || KIND != 2);  // partial rows at L2 scope (resident_rows_at_l2); the masked restart instantiation spilled 8 B with it
  bool l2rows = false, placed = !L2ROWS;  // uniform
  if constexpr (L2ROWS) resident_report_placement(sync);
  // KIND 0 / 1 (no gradient restart): nothing of an iteration depends on ||res|| but the stopping test, so the waves leave their
  // partial sums in LDS without a barrier and every thread adds them up behind the NEXT iteration's exchange (a stop found there
  // drops that exchange: nothing of it has been applied) -- fista_resident_kernel's DEFER, round 5
  constexpr bool DEFER = KIND != 2;
  bool pend = false;  // uniform
  auto resolve = [&](unsigned par) {
    double rn = 0.0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) rn += R.L.red[24 + 16 * par + i];
    res_norm = uni((float)sqrt(rn));
    done = uni((int)(((double)res_norm / (double)norm_x0) < (double)rel_tol));
    pend = false;
  };
  for (int it = 0; it < n_steps; ++it) {
    owner_products<E, G, K, WV, FULL>(a, xv, R.ored, slab_rs, N, l2rows);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // x0: requested here, consumed behind the exchange -- except in the masked restart instantiation, which has no registers
    // to hold it across the exchange (12 B per lane of scratch otherwise) and asks for it afterwards
    constexpr bool X0_LATE = KIND == 2 && !FULL;
    E x0v[EPT];
    if constexpr (!X0_LATE) {
      if constexpr (FULL) load_owned_wide<E, EPT, NT>(x0v, x0, tid);
      else load_owned_wide_masked<E, EPT, NT>(x0v, x0, tid, N);
    }
    E raw[EPT];
    if (!resident_allreduce<E, G, K, WV, BAR, FULL, KIND != 2>(R, sync, slab_rs, raw_g, nwg, N, epoch, xchg, spin_limit, raw, [](int, E) {}, []() {})) {
      alive = false;
      break;
    }
    if (!placed) {
      l2rows = resident_rows_at_l2(sync);
      placed = true;
    }
    if constexpr (X0_LATE) load_owned_wide_masked<E, EPT, NT>(x0v, x0, tid, N);
    if constexpr (DEFER) {
      if (pend) {  // uniform: the previous iteration's stopping test, behind this exchange's barriers
        resolve(((unsigned)ran & 1u) ^ 1u);
        if (done) break;  // it had converged: this iteration's exchange is dropped, nothing of it was applied
      }
    }
    float c0, c1, c2, c3, c4, c5, c6 = 0.f, rg = 0.f, th = 1.f, gamma_n = 1.f;
    if constexpr (KIND == 2) {
      // the coefficients of this iteration from theta, sigma, gamma: Float32, one rounding per operation, the host's order
      // (pogm_auto_kernel, pgm.hip; src/POGM.jl:183-201)
      const float rho = CF.c[0][0], lam = CF.c[0][1];
      const bool last = iteration == (int)CF.c[0][3] - 1;
      const float t2 = f32_mul(f32_mul(last ? 8.f : 4.f, theta), theta);
      th = f32_add(1.f, sqrtf(f32_add(1.f, t2))) / 2.f;
      const float alpha = f32_sub(theta, 1.f) / th;
      const float beta = f32_mul(sigma, theta) / th;
      c3 = f32_add(f32_add(1.f, alpha), beta);
      gamma_n = f32_mul(rho, c3);
      c5 = f32_mul(rho, alpha) / gamma;
      c4 = -f32_add(beta, c5);
      c1 = f32_mul(gamma_n, lam);
      c2 = -alpha;
      c0 = rho;
      rg = rho / gamma_n;
    } else {
      c0 = CF.c[it][0]; c1 = CF.c[it][1]; c2 = CF.c[it][2]; c3 = CF.c[it][3]; c4 = CF.c[it][4]; c5 = CF.c[it][5];
      c6 = CF.c[it][6];
    }
    E ri[EPT], ov[EPT];
    double rn = 0.0, dwx = 0.0, dwz = 0.0, dwr = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = own_index<E, EPT, NT, true>(tid, e);
      E r = elem<E>::sub(raw[e], x0v[e]);
      if (!FULL && i >= N) r = elem<E>::zero();
      ri[e] = r;
      rn += (double)elem<E>::re(r) * (double)elem<E>::re(r) + (double)elem<E>::im(r) * (double)elem<E>::im(r);
      if constexpr (KIND == 0) {  // OptISTA: {step, thr, c_z, c_y, c_x, c_zn, c_zo}   (optista_update_kernel, pgm.hip)
        const E zo = zv[e], ztmp = yv[e], xi = xv[e];
        E yn = elem<E>::add(ztmp, elem<E>::scale(-c0, r));
        yn = fista_prox_elem<E>(yn, reg_kind, c1);
        E zn = elem<E>::add(elem<E>::scale(c2, ztmp), xi);
        zn = elem<E>::add(zn, elem<E>::scale(c3, yn));
        E xn = elem<E>::add(elem<E>::scale(c4, xi), elem<E>::scale(c5, zn));
        xn = elem<E>::add(xn, elem<E>::scale(c6, zo));
        if (!FULL && i >= N) yn = zn = xn = elem<E>::zero();
        ov[e] = zo;   // zold
        yv[e] = yn;
        zv[e] = zn;
        xv[e] = xn;
      } else {  // POGM: {rho, thr, c_y, c_x1, c_xo, c_z}   (pogm_update_body, pgm.hip)
        const E xo = xv[e], yp = yv[e];
        const E x1 = elem<E>::add(xo, elem<E>::scale(-c0, r));
        E xn = elem<E>::add(elem<E>::scale(c2, yp), elem<E>::scale(c3, x1));
        xn = elem<E>::add(xn, elem<E>::scale(c4, xo));
        xn = elem<E>::add(xn, elem<E>::scale(c5, zv[e]));
        E zn = xn;
        xn = fista_proj_elem<E>(fista_prox_elem<E>(xn, reg_kind, c1), proj_kind);
        E y1 = x1;
        if (!FULL && i >= N) xn = zn = y1 = elem<E>::zero();
        ov[e] = xo;   // xold
        zv[e] = zn;
        yv[e] = y1;   // the gradient point: the reference's y after its swap
        xv[e] = xn;   // the next operator input
        if constexpr (KIND == 2) {  // gradient restart, src/POGM.jl:218-232 (pogm_update_body<E, true>, pgm.hip)
          E wi = elem<E>::add(wv[e], y1);
          wi = elem<E>::add(wi, elem<E>::scale(rg, xn));
          wi = elem<E>::add(wi, elem<E>::scale(-rg, zn));
          dwx += (double)elem<E>::re(wi) * (double)elem<E>::re(xn) + (double)elem<E>::im(wi) * (double)elem<E>::im(xn);
          dwz += (double)elem<E>::re(wi) * (double)elem<E>::re(zn) + (double)elem<E>::im(wi) * (double)elem<E>::im(zn);
          dwr += (double)elem<E>::re(wi) * (double)elem<E>::re(r) + (double)elem<E>::im(wi) * (double)elem<E>::im(r);
          const E wn = elem<E>::add(elem<E>::scale(rg, zn), elem<E>::scale(-rg, xn));
          wv[e] = elem<E>::sub(wn, y1);
        }
      }
    }
    if (blockIdx.x == 0) {
      store_buf(o0_rs, ov);
      store_buf(res_rs, ri);
    }
    if constexpr (DEFER) {
      if (defer_on) {  // uniform
        rn = wave_sum(rn);
        if ((tid & 63) == 0) R.L.red[24 + 16 * ((unsigned)ran & 1u) + (tid >> 6)] = rn;  // read behind the next exchange, or at the end
        pend = true;
        iteration += 1;
        ran += 1;
        continue;
      }
    }
    rn = block_sum_nolead<NT / 64>(rn, R.L.red);
    res_norm = uni((float)sqrt(rn));
    if constexpr (KIND == 2) {
      block_sum3(dwx, dwz, dwr, R.L.red);
      const float crit = f32_sub(f32_sub((float)dwx, (float)dwz) / gamma_n, (float)dwr);   // :224
      const bool restart = crit < 0.f;
      theta_old = theta;
      theta = uni(restart ? 1.f : th);
      sigma = uni(restart ? 1.f : f32_mul(sigma, CF.c[0][2]));
      gamma = uni(gamma_n);
    }
    iteration += 1;
    ran += 1;
    done = uni((int)(((double)res_norm / (double)norm_x0) < (double)rel_tol));
    if (done) break;  // uniform: every workgroup derived the same scalar
  }
  if constexpr (DEFER) {
    if (alive && pend) {  // uniform: the last applied iteration's norm and stopping test
      lds_barrier();
      resolve(((unsigned)ran - 1u) & 1u);
    }
  }
  if (!alive) {
    resident_give_up(sync, nullptr);
    return;
  }
  if (blockIdx.x == 0) {
    // POGM: an odd number of iterations leaves the roles of the two buffers swapped (the host swaps its references as often)
    E* xd = (KIND >= 1 && (ran & 1)) ? b1 : b0;
    E* yd = (KIND >= 1 && (ran & 1)) ? b0 : b1;
    store_buf(__builtin_amdgcn_make_buffer_rsrc(xd, 0, 0xffffffff, 0x00020000), xv);
    store_buf(__builtin_amdgcn_make_buffer_rsrc(yd, 0, 0xffffffff, 0x00020000), yv);
    store_buf(__builtin_amdgcn_make_buffer_rsrc(b2, 0, 0xffffffff, 0x00020000), zv);
    if constexpr (KIND == 2) store_buf(__builtin_amdgcn_make_buffer_rsrc(b3, 0, 0xffffffff, 0x00020000), wv);
    if (tid == 0) {
      if constexpr (KIND == 2) {
        pogm_auto_state* ast = reinterpret_cast<pogm_auto_state*>(st);
        ast->theta = theta;
        ast->theta_old = theta_old;
        ast->sigma = sigma;
        ast->gamma = gamma;
      }
      st->iteration = iteration;
      st->done = done;
      st->res_norm = res_norm;
      sync->completed = 1u;
    }
  }
}

// ---- resident Gram-mode FISTA: as cgnr_gram_resident_kernel, for src/FISTA.jl:139-185 with AHA explicit ---------------
// Per iteration: xs = y, this workgroup's rows of AHA y published (two parities), ONE grid exchange, then the gradient
// step, prox, restart test, theta and the next extrapolated point redundantly in every workgroup.
// Without gradient restart nothing of an iteration depends on ||res|| but the stopping test (src/FISTA.jl:156,187-189; theta
// follows a data-independent recursion, :179-180): the waves leave their partial ||res||^2 in LDS WITHOUT a barrier and every
// thread adds the eight up behind the NEXT iteration's grid barrier, where a stop found late drops that iteration's exchange
// (nothing of it has been applied) -- the block reduction and its two barriers are off the critical path.
// SRV: the instantiation that can stay and listen (server mode, rls_fista_step_status); not built for Float32 K = 32 (it spilled)
template <typename E, int K, int BAR, bool FULL, int SRV = 0>  // SRV: 0 one pass, 1 stays and listens, 2 ... and runs one iteration ahead
__global__ __launch_bounds__(512) void fista_gram_resident_kernel(const E* __restrict__ Gm, int64_t ldg, E* b0, E* b1,
                                                                   const E* __restrict__ x0, E* res, E* y0, E* y1,
                                                                   E* rr0, E* rr1, fista_scalars* sc0, fista_scalars* sc1,
                                                                   resident_sync* sync, int64_t Mc, int64_t N, int pair,
                                                                   int n_steps, unsigned spin_limit, rls_srv_args Sv) {
  constexpr int G = 4, WV = 8;
  using C = slab_cfg<E, G, K, WV>;
  constexpr int NV = C::NV, EPT = C::EPT, NT = C::NT;
  __shared__ gram_lds<E, G, K, WV> L;
  __shared__ int flag;
  __shared__ double wpart[2][WV];  // per-wave partial ||res||^2 of an iteration, by the parity of its exchange
  const int tid = threadIdx.x;
  const int nwg = gridDim.x;
  fista_scalars S;
  RLS_FISTA_COPY(S, *sc0);
  constexpr bool DEFER = K != 32;  // (the 32-pieces-per-row slabs have no registers to spare for the second update path: 28 B spilled)
  bool pend = false;  // the norm of the last applied iteration is still in wpart (uniform)
  auto resolve = [&](unsigned par) {  // :156, :187-189 for that iteration
    double rn = 0.0;
#pragma unroll
    for (int i = 0; i < WV; ++i) rn += wpart[par][i];
    const double res_norm = sqrt(rn);
    const float rel = (float)(res_norm / S.norm_x0);
    S.res_norm = res_norm;
    S.rel_res_norm = (double)rel;
    if (rel < S.rel_tol) S.done = 1;
    pend = false;
  };
  E yv[EPT], xk[EPT], xp[EPT], x0v[EPT], ri[EPT];
  {
    const E* yc = S.ycur ? y1 : y0;
    const E* xc = (S.iteration & 1) ? b1 : b0;  // state.x == buf[iteration & 1]
    const E* xo = (S.iteration & 1) ? b0 : b1;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {  // FULL: N == EPT * NT; else clamped loads, zeroed tails
      const int64_t i = tid + (int64_t)e * NT;
      const int64_t ic = FULL || i < N ? i : (N - 1);
      yv[e] = yc[ic];
      xk[e] = xc[ic];
      xp[e] = xo[ic];
      x0v[e] = x0[ic];
      ri[e] = res[ic];
      if (!FULL && i >= N) yv[e] = xk[e] = xp[e] = x0v[e] = ri[e] = elem<E>::zero();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  chunk<E, NV> a[K];
  slab_load<E, G, K, WV, FULL>(a, Gm, ldg, Mc, N, pair);
  __builtin_amdgcn_sched_barrier(0);
  if (S.done || n_steps <= 0) return;  // uniform
  unsigned epoch = 0;
  bool alive = true;
  int ycur = S.ycur;
  rls_mailbox_slot srv_mb = Sv.mb;
  unsigned srv_seq = Sv.seq0;  // server mode (rls_fista_step_status): the command being served
  unsigned itg = 0;            // iterations run by this launch, over all its commands: the parity of the exchanged rows
  int credit = 0;              // SRV == 2: iterations of the current command computed ahead of it (as cgnr_gram_resident_kernel)
  bool ahead = false;          // ... the pass below runs ahead of its command
  for (;;) {  // (server mode: one pass per command; otherwise one pass)
  for (int it = SRV ? credit : 0; it < n_steps; ++it) {
    if (SRV && S.done) break;  // uniform (a command behind the one that reached the stopping test)
    const unsigned itn = itg++;
    E* rq = (itn & 1u) ? rr1 : rr0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) L.xs[tid + e * NT] = yv[e];
    gram_rows<E, G, K, WV, FULL>(a, L, Mc, N, pair);
    if (tid < G * NV) {
      const int gg = tid / NV, i = tid % NV;
      E sum = elem<E>::zero();
#pragma unroll
      for (int ww = 0; ww < WV; ++ww) sum = elem<E>::add(sum, L.part[ww][gg][i]);
      const int64_t row = (row_block_of(blockIdx.x, pair) * G + gg) * NV + i;
      if (FULL || row < N) sc1_store_elem<E>(rq + row, sum);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!grid_arrive_wait(sync->cnt, ++epoch, (unsigned)nwg, spin_limit, &flag)) {
      alive = false;
      break;
    }
    E raw[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = tid + (int64_t)e * NT;
      raw[e] = sc1_load_elem<E>(rq + (FULL || i < N ? i : (N - 1)));
      if (!FULL && i >= N) raw[e] = elem<E>::zero();
    }
    if (pend) {  // uniform: the previous iteration's stopping test, under the loads just requested
      resolve((itn & 1u) ^ 1u);
      if (S.done) break;  // it had converged: this iteration's exchange is dropped, nothing of it was applied
    }
    E xn[EPT], yn[EPT];
    bool done;
    if (!DEFER || S.restart) {  // uniform: the restart test needs its dot product now (:171-176)
      fista_scalars Sn;
      done = fista_update_elems<E, EPT, NT, false, true>(S, raw, x0v, yv, xk, N, L.red, ri, xn, yn, Sn);
      RLS_FISTA_COPY(S, Sn);
    } else {
      const float rho = S.rho, thr = S.rho * S.lambda;
      double rn = 0.0;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int64_t i = tid + (int64_t)e * NT;
        E r = elem<E>::sub(raw[e], x0v[e]);                                   // res .-= x0      :153
        E xv = elem<E>::sub(yv[e], elem<E>::scale(rho, r));                   // x .-= rho .* res :154
        xv = fista_proj_elem<E>(fista_prox_elem<E>(xv, S.reg_kind, thr), S.proj_kind);
        if (!FULL && i >= N) {
          r = elem<E>::zero();
          xv = elem<E>::zero();
        }
        ri[e] = r;
        xn[e] = xv;
        rn += (double)elem<E>::re(r) * (double)elem<E>::re(r) + (double)elem<E>::im(r) * (double)elem<E>::im(r);
      }
      rn = wave_sum(rn);
      if ((tid & 63) == 0) wpart[itn & 1u][tid >> 6] = rn;  // (read behind the next grid barrier or the barrier at the end)
      pend = true;
      const float theta_old = S.theta;                                        // :179
      const float theta = (1.f + sqrtf(1.f + 4.f * theta_old * theta_old)) / 2.f;  // :180
      const float c1 = (1.f - theta_old) / theta, c2 = (theta_old - 1.f) / theta + 1.f;
#pragma unroll
      for (int e = 0; e < EPT; ++e) yn[e] = elem<E>::add(elem<E>::scale(c1, xk[e]), elem<E>::scale(c2, xn[e]));
      S.theta = theta;
      S.theta_old = theta_old;
      S.iteration += 1;
      S.done = S.iteration >= S.max_iter;  // the relTol half of :187-189 follows with the norm
      done = S.done != 0;
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      xp[e] = xk[e];
      xk[e] = xn[e];
      if (!done) yv[e] = yn[e];
    }
    if (!done) ycur ^= 1;
    if (done) break;  // uniform
  }
  if (alive && pend) {  // uniform: the last applied iteration's norm and stopping test
    __syncthreads();
    resolve((itg - 1u) & 1u);
  }
  if (!alive) {
    resident_give_up(sync, nullptr);
    if (SRV && Sv.ctl && blockIdx.x == 0 && tid == 0) __hip_atomic_store(Sv.ctl + 17, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  if constexpr (SRV != 0) {
    if (ahead) {  // that pass ran ahead: its command first (uniform); told to leave, memory holds the last command served
      ahead = false;
      const unsigned cmd = resident_listen(Sv.ctl, srv_seq, Sv.idle_us, sync, epoch, (unsigned)nwg, spin_limit, &flag, srv_mb, srv_seq - Sv.seq0 + 1u);
      if (cmd == RLS_SRV_EXIT) return;
      n_steps = (int)cmd;
      credit = 1;
      continue;
    }
  }
  if (blockIdx.x == 0) {
    S.ycur = ycur;
    S.pending = 0;
    S.fresh = 0;
    if constexpr (SRV != 0) {
      if (Sv.ctl && tid < 64) rls_mailbox_publish(srv_mb, S, tid);  // (server mode: status first, write-back under the host's turnaround)
    }
    E* xw = (S.iteration & 1) ? b1 : b0;  // state.x == buf[iteration & 1] afterwards as well
    E* xo = (S.iteration & 1) ? b0 : b1;
    E* yw = ycur ? y1 : y0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = tid + (int64_t)e * NT;
      if (FULL || i < N) {
        xw[i] = xk[e];
        xo[i] = xp[e];
        yw[i] = yv[e];
        res[i] = ri[e];
      }
    }
    __syncthreads();
    if (tid == 0) {
      RLS_FISTA_COPY(*sc0, S);
      RLS_FISTA_COPY(*sc1, S);
      sync->completed = 1u;
    }
  }
  if constexpr (SRV != 0) {
    if (!Sv.ctl) break;  // uniform
    credit = 0;
    if (SRV == 2 && !S.done) {  // uniform: one iteration ahead of the next command, under the host's turnaround
      ahead = true;
      n_steps = 1;
      continue;
    }
    const unsigned cmd = resident_listen(Sv.ctl, srv_seq, Sv.idle_us, sync, epoch, (unsigned)nwg, spin_limit, &flag, srv_mb, srv_seq - Sv.seq0 + 1u);
    if (cmd == RLS_SRV_EXIT) return;  // uniform
    n_steps = (int)cmd;
  } else {
    break;
  }
  }
}

struct fused_cfg {
  int G, K, WV;
};

// (the measurement switches of this file -- slab_g, slab_order, resident_barrier, red_threads, slab_multi -- live in the context:
//  rls_tuning, rls_common.hpp; rls_tune_set)

// workgroups (= partial rows) of a slab launch over `nwg` row blocks: the blocks themselves while they fit the chip's CUs, otherwise
// one workgroup per CU (slab_finish_multi; K = 32 slabs only -- the smaller ones leave room for two workgroups per CU, which
// overlap each other's streams by themselves)
static int slab_grid(rls_ctx* ctx, int K, int nwg) {
  if (!ctx->tune.slab_multi || K != 32) return nwg;
  int cus = ctx->cus;  // (cached in the context: no shared table between the per-rank worker threads)
  if (cus <= 0) {
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus <= 0) return nwg;
    ctx->cus = cus;
  }
  // one workgroup per CU; workgroup b walks blocks b, b + cus, b + 2 cus, ... (the last round may be a partial one: its few
  // workgroups stream their extra block alone, at their CU's full rate)
  return nwg <= cus ? nwg : cus;
}

// candidate slab shapes, smallest column capacity first; NMAX = K * WV * (64 / G).  (Rounds 1-2 also instantiated 16-wave
// workgroups -- {8, 16, 16}, {4, 16, 16} -- for the measurement switch "slab_wv": at 128 VGPRs per lane every one of them
// spilled, none was ever selected by this table's default, and they were a quarter of the library's code: removed.)
static const fused_cfg kCfgs[] = {{8, 8, 8}, {8, 16, 8}, {8, 32, 8}, {4, 32, 8}};

template <typename E>
static bool pick_cfg(const rls_tuning& T, int64_t N, fused_cfg* c) {
  for (const fused_cfg& k : kCfgs) {
    if (T.slab_g && k.G != T.slab_g) continue;  // measurement override (rls_tune_set "slab_g"): set before the operator is created
    const int64_t nmax = (int64_t)k.K * k.WV * (64 / k.G);
    // LDS image (slab_lds): the exchange planes + the input vector + small scratch must fit in 160 KiB
    const int64_t planes = k.G == 8 ? 4 : (elem<E>::cplx && k.G == 4) ? 2 : k.G;  // slab_planes
    const int64_t lds = (planes * (nmax + 64 / (int64_t)sizeof(E)) + nmax) * (int64_t)sizeof(E) + 4096;
    if (N <= nmax && lds <= 160 * 1024) {
      *c = k;
      return true;
    }
  }
  return false;
}

template <typename E>
static bool fused_ok(const rls_tuning& T, int64_t M, int64_t N, const void* A, int64_t lda) {
  constexpr int V = elem<E>::vec;
  fused_cfg c;
  if (!(A && M > 0 && N > 0 && M % V == 0 && lda % V == 0 && (reinterpret_cast<uintptr_t>(A) % 16 == 0))) return false;
  if (!pick_cfg<E>(T, N, &c)) return false;
  // 32-bit lane offsets: (slots per round) * column stride + row offset must stay below 2^32
  const int64_t cpr = c.WV * (64 / c.G);
  return cpr * lda * (int64_t)sizeof(E) + (M / V) * 16 < (int64_t)0xffffffffll;
}

template <typename E>
static int64_t fused_nwg(const rls_tuning& T, int64_t M, int64_t N) {
  fused_cfg c;
  pick_cfg<E>(T, N, &c);
  const int64_t Mc = M / elem<E>::vec;
  return (Mc + c.G - 1) / c.G;
}

template <typename KernelT>
static void allow_big_lds(KernelT* k, size_t lds) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

template <typename E, int G, int K, int WV>
static int launch_slab(rls_ctx* ctx, const E* A, int64_t lda, const E* p, E* slab, int64_t M, int64_t N, int nwg,
                       const int* skip) {
  using C = slab_cfg<E, G, K, WV>;
  const int64_t Mc = M / C::NV;
  const int pair = slab_pairing(G, nwg);
  constexpr size_t lds = sizeof(slab_lds<E, G, K, WV>);
  static rls_device_once attr_once;  // per template instantiation and device
  if (auto once_ = attr_once.first(ctx->device)) {
    allow_big_lds(&normal_slab_kernel<E, G, K, WV, true>, lds);
    allow_big_lds(&normal_slab_kernel<E, G, K, WV, false>, lds);
    if constexpr (K == 32) {
      allow_big_lds(&normal_slab_multi_kernel<E, G, K, WV, true>, lds);
      allow_big_lds(&normal_slab_multi_kernel<E, G, K, WV, false>, lds);
    }
  }
  const bool full = N == C::NMAX && (int64_t)nwg * G == Mc;
  const int grid = slab_grid(ctx, K, nwg);
  if constexpr (K == 32) {
    if (grid < nwg) {  // several row blocks per workgroup: `grid` partial rows
      if (full)
        hipLaunchKernelGGL((normal_slab_multi_kernel<E, G, K, WV, true>), dim3(grid), dim3(C::NT), lds, ctx->stream, A, lda, p,
                           slab, Mc, N, pair, nwg, skip);
      else
        hipLaunchKernelGGL((normal_slab_multi_kernel<E, G, K, WV, false>), dim3(grid), dim3(C::NT), lds, ctx->stream, A, lda, p,
                           slab, Mc, N, pair, nwg, skip);
      return grid;
    }
  }
  if (full)
    hipLaunchKernelGGL((normal_slab_kernel<E, G, K, WV, true>), dim3(nwg), dim3(C::NT), lds, ctx->stream, A, lda, p,
                       slab, Mc, N, pair, skip);
  else
    hipLaunchKernelGGL((normal_slab_kernel<E, G, K, WV, false>), dim3(nwg), dim3(C::NT), lds, ctx->stream, A, lda, p,
                       slab, Mc, N, pair, skip);
  return nwg;
}

// rows: the workgroups of K_A (= its partial rows); rows < nwg is the MULTI instantiation, which keeps <p, v>
static pipe_rhs_ptrs rhs_of(const rls_cgnr_pipe& P, int nwg, int rows = 0) {
  pipe_rhs_ptrs R;
  R.nrhs = P.nrhs > 0 ? P.nrhs : 1;
  R.vstride = P.vstride;
  R.slab_stride = (int64_t)nwg * P.N;
  R.hint = P.cur_hint;
  R.ttw = (rows > 0 && rows == nwg) ? P.ttw : nullptr;
  R.tt_rows = rows;
  return R;
}

// `grid` < nwg: the MULTI instantiation (K = 32 only), `grid` workgroups walking the nwg row blocks
template <typename E, int G, int K, int WV>
static void launch_pipe_a(rls_ctx* ctx, const rls_cgnr_pipe& P, int nwg, int grid) {
  using C = slab_cfg<E, G, K, WV>;
  const int64_t Mc = P.M / C::NV;
  const int pair = slab_pairing(G, nwg);
  constexpr size_t lds = sizeof(slab_lds<E, G, K, WV>);
  // ComplexF32 with 16 rows x 32 columns per workgroup (N in (2048, 4096]): the instantiation that loads BOTH (r, p) candidates
  // holds 6 x 8 owned elements beside a 128-register slab and spilled 28-36 B per lane.  It is not instantiated: a launch that
  // does not know which pair is current (the first node of a graph chunk) runs the hinted kernel on a guess -- that kernel
  // checks the hint on the device and re-loads the right pair, late, when it was wrong.
  constexpr bool ALWAYS_HINTED = elem<E>::cplx && G == 4 && K == 32;
  static rls_device_once attr_once;
  if (auto once_ = attr_once.first(ctx->device)) {
    if constexpr (!ALWAYS_HINTED) {
      allow_big_lds(&cgnr_pipe_a_kernel<E, G, K, WV, true, false, false>, lds);
      allow_big_lds(&cgnr_pipe_a_kernel<E, G, K, WV, false, false, false>, lds);
    }
    allow_big_lds(&cgnr_pipe_a_kernel<E, G, K, WV, true, false, true>, lds);
    allow_big_lds(&cgnr_pipe_a_kernel<E, G, K, WV, false, false, true>, lds);
    if constexpr (K == 32) {
      if constexpr (!ALWAYS_HINTED) {
        allow_big_lds(&cgnr_pipe_a_kernel<E, G, K, WV, true, false, false, true>, lds);
        allow_big_lds(&cgnr_pipe_a_kernel<E, G, K, WV, false, false, false, true>, lds);
      }
      allow_big_lds(&cgnr_pipe_a_kernel<E, G, K, WV, true, false, true, true>, lds);
      allow_big_lds(&cgnr_pipe_a_kernel<E, G, K, WV, false, false, true, true>, lds);
    }
  }
  const bool full = P.N == C::NMAX && (int64_t)nwg * G == Mc;
  const bool batched = P.nrhs > 1;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  // single right-hand side: the hinted instantiation (its full-size form loads the vectors in 16-byte pieces)
  const bool aligned = al16(P.r0) && al16(P.p0) && al16(P.r1) && al16(P.p1) && al16(P.v);
  const bool hinted = !batched && P.cur_hint >= 0 && aligned;
  pipe_rhs_ptrs R = rhs_of(P, nwg, grid);
#define RLS_LAUNCH_A2(FULLV, BATCHV, HINTV, MULTIV)                                                                     \
  hipLaunchKernelGGL((cgnr_pipe_a_kernel<E, G, K, WV, FULLV, BATCHV, HINTV, MULTIV>), dim3(MULTIV ? grid : nwg), dim3(C::NT),  \
                     lds, ctx->stream, (const E*)P.A, P.lda, (E*)P.x, (E*)P.r0, (E*)P.p0, (E*)P.r1, (E*)P.p1, (const E*)P.v, \
                     (E*)P.slab, P.dots, P.ndots, P.sc, P.scn, Mc, P.N, pair, ctx->tune.slab_order, R, nwg)
#define RLS_LAUNCH_A(FULLV, BATCHV, HINTV)                      \
  do {                                                          \
    if constexpr (K == 32) {                                    \
      if (grid < nwg) {                                         \
        RLS_LAUNCH_A2(FULLV, BATCHV, HINTV, true);              \
        break;                                                  \
      }                                                         \
    }                                                           \
    RLS_LAUNCH_A2(FULLV, BATCHV, HINTV, false);                 \
  } while (0)
  // (a BATCHED = true instantiation -- the slab kernel looping over several right-hand sides on the VALU -- existed until
  //  round 3 as the fallback of the matrix-core batched path; it spilled up to 528 bytes per lane and was four times slower
  //  per solve-iteration than the skinny kernels: shapes those do not cover now run one plan per column)
  (void)batched;
  if constexpr (ALWAYS_HINTED) {
    if (R.hint < 0) R.hint = 0;
    if (full && aligned) RLS_LAUNCH_A(true, false, true);   // (the full-size hinted form loads 16-byte pieces)
    else RLS_LAUNCH_A(false, false, true);
  } else {
    if (full && hinted) RLS_LAUNCH_A(true, false, true);
    else if (hinted) RLS_LAUNCH_A(false, false, true);
    else if (full) RLS_LAUNCH_A(true, false, false);
    else RLS_LAUNCH_A(false, false, false);
  }
#undef RLS_LAUNCH_A
#undef RLS_LAUNCH_A2
}

#define RLS_FOR_EACH_CFG(X) X(8, 8, 8) X(8, 16, 8) X(8, 32, 8) X(4, 32, 8)

static int32_t launch_status(rls_ctx* ctx) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

template <typename E>
static int32_t normal_typed(rls_ctx* ctx, int64_t M, int64_t N, const E* A, int64_t lda, const E* p, E* v, E* slab,
                            const int* skip) {
  fused_cfg c;
  if (!pick_cfg<E>(ctx->tune, N, &c)) return rls_fail(ctx, RLS_E_UNSUPPORTED, "normal_fused: N too large for a register slab");
  const int nwg = (int)fused_nwg<E>(ctx->tune, M, N);
  int rows = nwg;  // partial rows the slab launch leaves
#define RLS_SLAB_CASE(GG, KK, WW) \
  if (c.G == GG && c.K == KK && c.WV == WW) rows = launch_slab<E, GG, KK, WW>(ctx, A, lda, p, slab, M, N, nwg, skip);
  RLS_FOR_EACH_CFG(RLS_SLAB_CASE)
#undef RLS_SLAB_CASE
  hipLaunchKernelGGL(slab_reduce_kernel<E>, dim3((unsigned)((N + 15) / 16)), dim3(ctx->tune.red_threads), 0, ctx->stream, slab, rows, N,
                     v, skip);
  return launch_status(ctx);
}

template <typename E>
static int32_t pipe_iteration_typed(rls_ctx* ctx, const rls_cgnr_pipe& P, int which = 3) {
  fused_cfg c;
  if (!pick_cfg<E>(ctx->tune, P.N, &c)) return rls_fail(ctx, RLS_E_UNSUPPORTED, "cgnr pipeline: N too large");
  const int nwg = (int)fused_nwg<E>(ctx->tune, P.M, P.N);
  const int rows = slab_grid(ctx, c.K, nwg);  // partial rows K_A leaves = its workgroups
#define RLS_PIPE_CASE(GG, KK, WW) \
  if (c.G == GG && c.K == KK && c.WV == WW) launch_pipe_a<E, GG, KK, WW>(ctx, P, nwg, rows);
  if (which & 1) {
    RLS_FOR_EACH_CFG(RLS_PIPE_CASE)
  }
#undef RLS_PIPE_CASE
  if (which & 2)
    hipLaunchKernelGGL(cgnr_pipe_r_kernel<E>, dim3((unsigned)P.ndots, (unsigned)(P.nrhs > 0 ? P.nrhs : 1)),
                       dim3(ctx->tune.red_threads), 0, ctx->stream, (const E*)P.slab, rows, P.N, (E*)P.v, (const E*)P.p0,
                       (const E*)P.p1, P.dots, P.sc, P.scn, rhs_of(P, nwg, rows));
  return launch_status(ctx);
}

template <typename E>
static int32_t pipe_finish_typed(rls_ctx* ctx, const rls_cgnr_pipe& P) {
  const int ept = (int)((P.N + FIN_THREADS - 1) / FIN_THREADS);
#define RLS_FIN_CASE(EE)                                                                                         \
  hipLaunchKernelGGL((cgnr_pipe_f_kernel<E, EE>), dim3((unsigned)(P.nrhs > 0 ? P.nrhs : 1)), dim3(FIN_THREADS), 0, \
                     ctx->stream, (E*)P.x, (E*)P.r0, (E*)P.p0, (E*)P.r1, (E*)P.p1, (const E*)P.v, P.dots, P.ndots, P.sc, \
                     P.N, rhs_of(P, 0), P.mb)
  if (ept <= 1) RLS_FIN_CASE(1);
  else if (ept <= 2) RLS_FIN_CASE(2);
  else RLS_FIN_CASE(4);
#undef RLS_FIN_CASE
  return launch_status(ctx);
}

template <typename E, int G, int K, int WV>
static void launch_fista_a(rls_ctx* ctx, const rls_fista_pipe& P, int nwg, int grid) {
  using C = slab_cfg<E, G, K, WV>;
  const int64_t Mc = P.M / C::NV;
  const int pair = slab_pairing(G, nwg);
  constexpr size_t lds = sizeof(slab_lds<E, G, K, WV>);
  static rls_device_once attr_once;
  if (auto once_ = attr_once.first(ctx->device)) {
    allow_big_lds(&fista_pipe_a_kernel<E, G, K, WV, true, false>, lds);
    allow_big_lds(&fista_pipe_a_kernel<E, G, K, WV, false, false>, lds);
    allow_big_lds(&fista_pipe_a_kernel<E, G, K, WV, true, true>, lds);
    allow_big_lds(&fista_pipe_a_kernel<E, G, K, WV, false, true>, lds);
    if constexpr (K == 32) {
      allow_big_lds(&fista_pipe_a_kernel<E, G, K, WV, true, false, true>, lds);
      allow_big_lds(&fista_pipe_a_kernel<E, G, K, WV, false, false, true>, lds);
      allow_big_lds(&fista_pipe_a_kernel<E, G, K, WV, true, true, true>, lds);
      allow_big_lds(&fista_pipe_a_kernel<E, G, K, WV, false, true, true>, lds);
    }
  }
  const bool full = P.N == C::NMAX && (int64_t)nwg * G == Mc;
#define RLS_LAUNCH_FA2(FULLV, HINTV, MULTIV)                                                                              \
  hipLaunchKernelGGL((fista_pipe_a_kernel<E, G, K, WV, FULLV, HINTV, MULTIV>), dim3(MULTIV ? grid : nwg), dim3(C::NT), lds,   \
                     ctx->stream, (const E*)P.A, P.lda, (E*)P.b0, (E*)P.b1, (const E*)P.x0, (E*)P.res, (E*)P.y0, (E*)P.y1, \
                     (const E*)P.res_raw, (E*)P.slab, P.sc, P.scn, Mc, P.N, pair, P.par_hint, nwg)
#define RLS_LAUNCH_FA(FULLV, HINTV)                 \
  do {                                              \
    if constexpr (K == 32) {                        \
      if (grid < nwg) {                             \
        RLS_LAUNCH_FA2(FULLV, HINTV, true);         \
        break;                                      \
      }                                             \
    }                                               \
    RLS_LAUNCH_FA2(FULLV, HINTV, false);            \
  } while (0)
  if (full && P.par_hint >= 0) RLS_LAUNCH_FA(true, true);
  else if (P.par_hint >= 0) RLS_LAUNCH_FA(false, true);
  else if (full) RLS_LAUNCH_FA(true, false);
  else RLS_LAUNCH_FA(false, false);
#undef RLS_LAUNCH_FA
#undef RLS_LAUNCH_FA2
}

template <typename E>
static int32_t fista_iteration_typed(rls_ctx* ctx, const rls_fista_pipe& P) {
  fused_cfg c;
  if (!pick_cfg<E>(ctx->tune, P.N, &c)) return rls_fail(ctx, RLS_E_UNSUPPORTED, "fista pipeline: N too large");
  const int nwg = (int)fused_nwg<E>(ctx->tune, P.M, P.N);
  const int rows = slab_grid(ctx, c.K, nwg);
#define RLS_FISTA_CASE(GG, KK, WW) \
  if (c.G == GG && c.K == KK && c.WV == WW) launch_fista_a<E, GG, KK, WW>(ctx, P, nwg, rows);
  RLS_FOR_EACH_CFG(RLS_FISTA_CASE)
#undef RLS_FISTA_CASE
  hipLaunchKernelGGL(fista_pipe_r_kernel<E>, dim3((unsigned)((P.N + 15) / 16)), dim3(ctx->tune.red_threads), 0, ctx->stream,
                     (const E*)P.slab, rows, P.N, (E*)P.res_raw, P.sc, P.scn);
  return launch_status(ctx);
}

template <typename E>
static int32_t fista_finish_typed(rls_ctx* ctx, const rls_fista_pipe& P) {
  const int ept = (int)((P.N + FIN_THREADS - 1) / FIN_THREADS);
#define RLS_FFIN_CASE(EE)                                                                                        \
  hipLaunchKernelGGL((fista_pipe_f_kernel<E, EE>), dim3(1), dim3(FIN_THREADS), 0, ctx->stream, (E*)P.b0, (E*)P.b1, \
                     (const E*)P.x0, (E*)P.res, (E*)P.y0, (E*)P.y1, (const E*)P.res_raw, P.sc, P.N, P.mb)
  if (ept <= 1) RLS_FFIN_CASE(1);
  else if (ept <= 2) RLS_FFIN_CASE(2);
  else RLS_FFIN_CASE(4);
#undef RLS_FFIN_CASE
  return launch_status(ctx);
}

// ---- Gram pipeline host side -------------------------------------------------------------------
template <typename E>
static bool gram_pick(int64_t N, int* K) {  // G = 4 (64-byte row pieces: 8 / 16 rows per workgroup), WV = 8
  const int64_t cpr = 8 * (64 / 4);
  for (int k : {8, 16, 32}) {
    if (N <= k * cpr) {
      *K = k;
      return true;
    }
  }
  return false;
}

// workgroups of a Gram pipeline launch over nwg row blocks (= partial dots it leaves): see slab_grid
template <typename E>
static int gram_grid(rls_ctx* ctx, int K, int nwg) {
  return elem<E>::cplx ? slab_grid(ctx, K, nwg) : nwg;
}

template <typename E, int K>
static void launch_gram(rls_ctx* ctx, const rls_gram_pipe& P, int q, int nwg) {
  using C = slab_cfg<E, 4, K, 8>;
  const int64_t Mc = P.N / C::NV;
  const int pair = slab_pairing(4, nwg);
  const bool full = P.N == C::NMAX && (int64_t)nwg * 4 == Mc;
  const int grid = gram_grid<E>(ctx, K, nwg);
#define RLS_LAUNCH_G2(FULLV, MULTIV)                                                                                         \
  hipLaunchKernelGGL((cgnr_gram_kernel<E, 4, K, 8, FULLV, MULTIV>), dim3(MULTIV ? grid : nwg), dim3(C::NT), 0, ctx->stream,   \
                     (const E*)P.G, P.ldg, (E*)P.x, (const E*)P.r[q], (const E*)P.p[q], (E*)P.r[q ^ 1], (E*)P.p[q ^ 1],        \
                     (const E*)P.v[q], (E*)P.v[q ^ 1], P.dots + (size_t)q * 4 * nwg, P.dots + (size_t)(q ^ 1) * 4 * nwg,       \
                     MULTIV ? grid : nwg, P.sc[q], P.sc[q ^ 1], Mc, P.N, pair, ctx->tune.slab_order, nwg)
  // (Float32: N <= 4096 is at most 256 blocks of 16 rows -- nothing to walk, the instantiations would be dead code)
#define RLS_LAUNCH_G(FULLV)               \
  do {                                    \
    if constexpr (K == 32 && elem<E>::cplx) { \
      if (grid < nwg) {                   \
        RLS_LAUNCH_G2(FULLV, true);       \
        break;                            \
      }                                   \
    }                                     \
    RLS_LAUNCH_G2(FULLV, false);          \
  } while (0)
  if (full) RLS_LAUNCH_G(true);
  else RLS_LAUNCH_G(false);
#undef RLS_LAUNCH_G
#undef RLS_LAUNCH_G2
}

template <typename E>
static int32_t gram_iteration_typed(rls_ctx* ctx, const rls_gram_pipe& P, int q) {
  int K = 0;
  if (!gram_pick<E>(P.N, &K)) return rls_fail(ctx, RLS_E_UNSUPPORTED, "gram pipeline: N too large");
  const int nwg = rls_gram_pipe_nwg(elem<E>::cplx ? RLS_C32 : RLS_F32, P.N);
  if (K == 8) launch_gram<E, 8>(ctx, P, q, nwg);
  else if (K == 16) launch_gram<E, 16>(ctx, P, q, nwg);
  else launch_gram<E, 32>(ctx, P, q, nwg);
  return launch_status(ctx);
}

template <typename E>
static int32_t gram_finish_typed(rls_ctx* ctx, const rls_gram_pipe& P, int q) {
  const int nwg = rls_gram_pipe_nwg(elem<E>::cplx ? RLS_C32 : RLS_F32, P.N);
  int Kp = 0;
  gram_pick<E>(P.N, &Kp);
  const int ndots = gram_grid<E>(ctx, Kp, nwg);  // what the last iteration's launch left (launch_gram)
  const int ept = (int)((P.N + FIN_THREADS - 1) / FIN_THREADS);
#define RLS_GFIN_CASE(EE)                                                                                           \
  hipLaunchKernelGGL((cgnr_gram_f_kernel<E, EE>), dim3(1), dim3(FIN_THREADS), 0, ctx->stream, (E*)P.x,               \
                     (const E*)P.r[q], (const E*)P.p[q], (E*)P.r[0], (E*)P.p[0], (const E*)P.v[q], (E*)P.v[0],       \
                     P.dots + (size_t)q * 4 * nwg, ndots, P.sc[q], P.sc[0], P.sc[1], P.N)
  if (ept <= 1) RLS_GFIN_CASE(1);
  else if (ept <= 2) RLS_GFIN_CASE(2);
  else RLS_GFIN_CASE(4);
#undef RLS_GFIN_CASE
  return launch_status(ctx);
}

template <typename E, int K>
static void launch_fista_gram(rls_ctx* ctx, const rls_fista_gram& P, int q, int nwg) {
  using C = slab_cfg<E, 4, K, 8>;
  const int64_t Mc = P.N / C::NV;
  const int pair = slab_pairing(4, nwg);
  const bool full = P.N == C::NMAX && (int64_t)nwg * 4 == Mc;
  const int grid = gram_grid<E>(ctx, K, nwg);
#define RLS_LAUNCH_FG2(FULLV, HINTV, MULTIV)                                                                                  \
  hipLaunchKernelGGL((fista_gram_kernel<E, 4, K, 8, FULLV, HINTV, MULTIV>), dim3(MULTIV ? grid : nwg), dim3(C::NT), 0,         \
                     ctx->stream, (const E*)P.G, P.ldg, (E*)P.b0, (E*)P.b1, (const E*)P.x0, (E*)P.res, (E*)P.y0, (E*)P.y1,    \
                     (const E*)P.rr[q], (E*)P.rr[q ^ 1], P.sc[q], P.sc[q ^ 1], Mc, P.N, pair, P.par_hint, nwg)
#define RLS_LAUNCH_FG(FULLV, HINTV)             \
  do {                                          \
    if constexpr (K == 32 && elem<E>::cplx) {   \
      if (grid < nwg) {                         \
        RLS_LAUNCH_FG2(FULLV, HINTV, true);     \
        break;                                  \
      }                                         \
    }                                           \
    RLS_LAUNCH_FG2(FULLV, HINTV, false);        \
  } while (0)
  if (full && P.par_hint >= 0) RLS_LAUNCH_FG(true, true);
  else if (P.par_hint >= 0) RLS_LAUNCH_FG(false, true);
  else if (full) RLS_LAUNCH_FG(true, false);
  else RLS_LAUNCH_FG(false, false);
#undef RLS_LAUNCH_FG
#undef RLS_LAUNCH_FG2
}

template <typename E>
static int32_t fista_gram_iteration_typed(rls_ctx* ctx, const rls_fista_gram& P, int q) {
  int K = 0;
  if (!gram_pick<E>(P.N, &K)) return rls_fail(ctx, RLS_E_UNSUPPORTED, "gram pipeline: N too large");
  const int nwg = rls_gram_pipe_nwg(elem<E>::cplx ? RLS_C32 : RLS_F32, P.N);
  if (K == 8) launch_fista_gram<E, 8>(ctx, P, q, nwg);
  else if (K == 16) launch_fista_gram<E, 16>(ctx, P, q, nwg);
  else launch_fista_gram<E, 32>(ctx, P, q, nwg);
  return launch_status(ctx);
}

template <typename E>
static int32_t fista_gram_finish_typed(rls_ctx* ctx, const rls_fista_gram& P, int q) {
  const int ept = (int)((P.N + FIN_THREADS - 1) / FIN_THREADS);
#define RLS_FGFIN_CASE(EE)                                                                                         \
  hipLaunchKernelGGL((fista_gram_f_kernel<E, EE>), dim3(1), dim3(FIN_THREADS), 0, ctx->stream, (E*)P.b0, (E*)P.b1,  \
                     (const E*)P.x0, (E*)P.res, (E*)P.y0, (E*)P.y1, (const E*)P.rr[q], P.sc[q], P.sc[0], P.sc[1], P.N)
  if (ept <= 1) RLS_FGFIN_CASE(1);
  else if (ept <= 2) RLS_FGFIN_CASE(2);
  else RLS_FGFIN_CASE(4);
#undef RLS_FGFIN_CASE
  return launch_status(ctx);
}


// ---- resident CGNR host side ------------------------------------------------------------------------
// the two-level exchange (resident_allreduce, EXCH 2): 8 equal groups, the row split into a power-of-two number of
// 16-byte pieces per group member, at most one piece per thread.  Other grids (ragged M) take the flat exchange.
template <typename E>
static bool resident_two_level_ok(const rls_tuning& T, int nwg, int64_t N, int nt) {
  if (T.resident_barrier != 2 || nwg % RES_GROUPS != 0) return false;
  const int64_t pieces = N * (int64_t)sizeof(E) / 16, per = nwg / RES_GROUPS;
  if (N * (int64_t)sizeof(E) % 16 != 0 || pieces % per != 0) return false;
  const int64_t q = pieces / per;
  return q >= 1 && q <= nt && (q & (q - 1)) == 0;
}

template <typename E, int G, int K, int WV>
static int32_t launch_resident(rls_ctx* ctx, const rls_cgnr_pipe& P, double* dout, void* sync, int nwg, int n_steps,
                               unsigned spin_limit, const rls_cg_start& St) {
  using C = slab_cfg<E, G, K, WV>;
  // (complex with 64-byte row pieces holds 8 owned elements of x, r, p, v per thread on top of the slab: spills)
  if constexpr ((K == 32 || K == 16) && WV == 8 && C::EPT % C::NV == 0 && !(elem<E>::cplx && G == 4)) {
    const int64_t Mc = P.M / C::NV;
    const int pair = (G == 4 && nwg % 16 == 0) ? 1 : 0;
    const bool full = P.N == C::NMAX && (int64_t)nwg * G == Mc;
    constexpr size_t lds = resident_lds_bytes<E, G, K, WV>();
    static rls_device_once attr_once;
    if (auto once_ = attr_once.first(ctx->device)) {
      allow_big_lds(&cgnr_resident_kernel<E, G, K, WV, 1, true>, lds);
      allow_big_lds(&cgnr_resident_kernel<E, G, K, WV, 2, true>, lds);
      allow_big_lds(&cgnr_resident_kernel<E, G, K, WV, 1, false>, lds);
      allow_big_lds(&cgnr_resident_kernel<E, G, K, WV, 2, false>, lds);
      allow_big_lds(&cgnr_resident_kernel<E, G, K, WV, 1, true, true>, lds);
      allow_big_lds(&cgnr_resident_kernel<E, G, K, WV, 2, true, true>, lds);
      allow_big_lds(&cgnr_resident_kernel<E, G, K, WV, 1, false, true>, lds);
      allow_big_lds(&cgnr_resident_kernel<E, G, K, WV, 2, false, true>, lds);
      }
#define RLS_LAUNCH_RES(BB, FF, SS)                                                                                      \
  hipLaunchKernelGGL((cgnr_resident_kernel<E, G, K, WV, BB, FF, SS>), dim3(nwg), dim3(C::NT), lds, ctx->stream, (const E*)P.A, \
                     P.lda, (E*)P.x, (E*)P.r1, (E*)P.r0, (E*)P.p0, (E*)P.v, (E*)P.slab, dout, P.sc, (resident_sync*)sync, Mc, P.N,  \
                     pair, n_steps, spin_limit, St)
    // a kernel that stays and listens runs one iteration ahead of its commands (SPEC) unless the context says otherwise
    const bool spec = St.srv_ctl != nullptr && ctx->tune.resident_ahead != 0;
    if (resident_two_level_ok<E>(ctx->tune, nwg, P.N, C::NT)) {
      if (spec) {
        if (full) RLS_LAUNCH_RES(2, true, true);
        else RLS_LAUNCH_RES(2, false, true);
      } else {
        if (full) RLS_LAUNCH_RES(2, true, false);
        else RLS_LAUNCH_RES(2, false, false);
      }
    } else {
      if (spec) {
        if (full) RLS_LAUNCH_RES(1, true, true);
        else RLS_LAUNCH_RES(1, false, true);
      } else {
        if (full) RLS_LAUNCH_RES(1, true, false);
        else RLS_LAUNCH_RES(1, false, false);
      }
    }
#undef RLS_LAUNCH_RES
    return launch_status(ctx);
  } else {
    return rls_fail(ctx, RLS_E_UNSUPPORTED, "resident CGNR: slab shape not instantiated");
  }
}

template <typename E>
static bool resident_ok_typed(const rls_tuning& T, int device, int64_t M, int64_t N, const void* A, int64_t lda) {
  if (!fused_ok<E>(T, M, N, A, lda)) return false;
  fused_cfg c;
  if (!pick_cfg<E>(T, N, &c) || (c.K != 32 && c.K != 16) || c.WV != 8 || (elem<E>::cplx && c.G == 4)) return false;
  if (((int64_t)c.K * c.WV * (64 / c.G) / (c.WV * 64)) % elem<E>::vec) return false;  // 16-byte ownership pieces: EPT % V == 0
  // the K = 32 slab shapes: N in (NMAX / 2, NMAX], N a multiple of the 16-byte piece; ragged M and N run the masked
  // instantiation (the full-size one has no clamps at all)
  const int64_t nwg = fused_nwg<E>(T, M, N);
  if (N % elem<E>::vec) return false;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return false;
  // one 512-thread workgroup (256 VGPRs per lane, ~148 KiB of LDS) per CU: the grid is resident iff it fits the CUs;
  // the arrival words of the barrier (resident_sync::cnt, per-workgroup flags in mode 0) hold 256 workgroups
  if (!(nwg <= cus && nwg <= 256 && nwg * N * (int64_t)sizeof(E) < (int64_t)0xffffffffll)) return false;
  // ... and the runtime must agree that a workgroup of this instantiation fits a CU at all (the occupancy query is
  // advisory upwards -- it can over-report by one -- but "0" is a firm no: e.g. a device with less LDS per CU)
  int blocks = 0;
#define RLS_OCC_CASE(GG, KK, WW)                                                                                        \
  if (c.G == GG && c.K == KK && c.WV == WW) {                                                                            \
    if constexpr ((KK == 32 || KK == 16) && WW == 8 && slab_cfg<E, GG, KK, WW>::EPT % elem<E>::vec == 0 &&               \
                  !(elem<E>::cplx && GG == 4)) {                                                                         \
      allow_big_lds(&cgnr_resident_kernel<E, GG, KK, WW, 1, false>, resident_lds_bytes<E, GG, KK, WW>());                \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, cgnr_resident_kernel<E, GG, KK, WW, 1, false>, WW * 64,  \
                                                       resident_lds_bytes<E, GG, KK, WW>()) != hipSuccess)               \
        blocks = 0;                                                                                                      \
    }                                                                                                                    \
  }
  RLS_FOR_EACH_CFG(RLS_OCC_CASE)
#undef RLS_OCC_CASE
  (void)hipGetLastError();
  return blocks >= 1;
}

template <typename E>
static int32_t resident_typed(rls_ctx* ctx, const rls_cgnr_pipe& P, double* dout, void* sync, int n_steps,
                              unsigned spin_limit, const rls_cg_start& St) {
  fused_cfg c;
  if (!pick_cfg<E>(ctx->tune, P.N, &c)) return rls_fail(ctx, RLS_E_UNSUPPORTED, "resident CGNR: N too large");
  const int nwg = (int)fused_nwg<E>(ctx->tune, P.M, P.N);
  int32_t st = RLS_E_UNSUPPORTED;
#define RLS_RES_CASE(GG, KK, WW) \
  if (c.G == GG && c.K == KK && c.WV == WW) st = launch_resident<E, GG, KK, WW>(ctx, P, dout, sync, nwg, n_steps, spin_limit, St);
  RLS_FOR_EACH_CFG(RLS_RES_CASE)
#undef RLS_RES_CASE
  return st;
}

template <typename E, int G, int K, int WV>
static int32_t launch_fista_resident(rls_ctx* ctx, const rls_fista_pipe& P, void* sync, int nwg, int n_steps,
                                     unsigned spin_limit, const rls_srv_args& Sv) {
  using C = slab_cfg<E, G, K, WV>;
  if constexpr ((K == 32 || K == 16) && WV == 8 && C::EPT % C::NV == 0 && !(elem<E>::cplx && G == 4)) {
    const int64_t Mc = P.M / C::NV;
    const int pair = (G == 4 && nwg % 16 == 0) ? 1 : 0;
    const bool full = P.N == C::NMAX && (int64_t)nwg * G == Mc;
    constexpr size_t lds = resident_lds_bytes<E, G, K, WV>();
    static rls_device_once attr_once;
    if (auto once_ = attr_once.first(ctx->device)) {
      allow_big_lds(&fista_resident_kernel<E, G, K, WV, 1, true>, lds);
      allow_big_lds(&fista_resident_kernel<E, G, K, WV, 2, true>, lds);
      allow_big_lds(&fista_resident_kernel<E, G, K, WV, 1, false>, lds);
      allow_big_lds(&fista_resident_kernel<E, G, K, WV, 2, false>, lds);
      allow_big_lds(&fista_resident_kernel<E, G, K, WV, 1, true, true>, lds);
      allow_big_lds(&fista_resident_kernel<E, G, K, WV, 2, true, true>, lds);
      allow_big_lds(&fista_resident_kernel<E, G, K, WV, 1, false, true>, lds);
      allow_big_lds(&fista_resident_kernel<E, G, K, WV, 2, false, true>, lds);
      }
#define RLS_LAUNCH_FRES(BB, FF, SS)                                                                                      \
  hipLaunchKernelGGL((fista_resident_kernel<E, G, K, WV, BB, FF, SS>), dim3(nwg), dim3(C::NT), lds, ctx->stream, (const E*)P.A, \
                     P.lda, (E*)P.b0, (E*)P.b1, (const E*)P.x0, (E*)P.res, (E*)P.y0, (E*)P.y1, (E*)P.res_raw, (E*)P.slab,  \
                     P.sc, (resident_sync*)sync, Mc, P.N, pair | (ctx->tune.fista_defer ? 0 : 2), n_steps, spin_limit, Sv)
    const bool spec = Sv.ctl != nullptr && ctx->tune.resident_ahead != 0;  // a kernel that stays and listens runs one iteration ahead of its commands
    if (resident_two_level_ok<E>(ctx->tune, nwg, P.N, C::NT)) {
      if (spec) {
        if (full) RLS_LAUNCH_FRES(2, true, true);
        else RLS_LAUNCH_FRES(2, false, true);
      } else {
        if (full) RLS_LAUNCH_FRES(2, true, false);
        else RLS_LAUNCH_FRES(2, false, false);
      }
    } else {
      if (spec) {
        if (full) RLS_LAUNCH_FRES(1, true, true);
        else RLS_LAUNCH_FRES(1, false, true);
      } else {
        if (full) RLS_LAUNCH_FRES(1, true, false);
        else RLS_LAUNCH_FRES(1, false, false);
      }
    }
#undef RLS_LAUNCH_FRES
    return launch_status(ctx);
  } else {
    return rls_fail(ctx, RLS_E_UNSUPPORTED, "resident FISTA: slab shape not instantiated");
  }
}

template <typename E>
static int32_t fista_resident_typed(rls_ctx* ctx, const rls_fista_pipe& P, void* sync, int n_steps, unsigned spin_limit,
                                    const rls_srv_args& Sv) {
  fused_cfg c;
  if (!pick_cfg<E>(ctx->tune, P.N, &c)) return rls_fail(ctx, RLS_E_UNSUPPORTED, "resident FISTA: N too large");
  const int nwg = (int)fused_nwg<E>(ctx->tune, P.M, P.N);
  int32_t st = RLS_E_UNSUPPORTED;
#define RLS_FRES_CASE(GG, KK, WW) \
  if (c.G == GG && c.K == KK && c.WV == WW) st = launch_fista_resident<E, GG, KK, WW>(ctx, P, sync, nwg, n_steps, spin_limit, Sv);
  RLS_FOR_EACH_CFG(RLS_FRES_CASE)
#undef RLS_FRES_CASE
  return st;
}
template <typename E, int K>
static int32_t launch_gram_resident(rls_ctx* ctx, const rls_gram_pipe& P, void* sync, int nwg, int n_steps,
                                    unsigned spin_limit, const rls_cg_start& St) {
  using C = slab_cfg<E, 4, K, 8>;
  const int64_t Mc = P.N / C::NV;
  const int pair = (nwg % 16 == 0) ? 1 : 0;
  const bool full = P.N == C::NMAX && (int64_t)nwg * 4 == Mc;
#define RLS_LAUNCH_GR(BB, FF)                                                                                          \
  hipLaunchKernelGGL((cgnr_gram_resident_kernel<E, K, BB, FF>), dim3(nwg), dim3(C::NT), 0, ctx->stream, (const E*)P.G,  \
                     P.ldg, (E*)P.x, (E*)P.r[0], (E*)P.p[0], (E*)P.v[0], (E*)P.v[1], P.dots, P.sc[0], P.sc[1],         \
                     (resident_sync*)sync, Mc, P.N, pair, n_steps, spin_limit, St)
  if constexpr (K != 32) {
    if (St.srv_ctl) {  // the instantiation that can stay and listen
#define RLS_LAUNCH_GRS(FF, SS)                                                                                            \
  hipLaunchKernelGGL((cgnr_gram_resident_kernel<E, K, 1, FF, SS>), dim3(nwg), dim3(C::NT), 0, ctx->stream, (const E*)P.G, \
                     P.ldg, (E*)P.x, (E*)P.r[0], (E*)P.p[0], (E*)P.v[0], (E*)P.v[1], P.dots, P.sc[0], P.sc[1],            \
                     (resident_sync*)sync, Mc, P.N, pair, n_steps, spin_limit, St)
      if (ctx->tune.resident_ahead) {  // one iteration ahead of its commands
        if (full) RLS_LAUNCH_GRS(true, 2);
        else RLS_LAUNCH_GRS(false, 2);
      } else {
        if (full) RLS_LAUNCH_GRS(true, 1);
        else RLS_LAUNCH_GRS(false, 1);
      }
#undef RLS_LAUNCH_GRS
      return launch_status(ctx);
    }
  } else if (St.srv_ctl) {
    return rls_fail(ctx, RLS_E_UNSUPPORTED, "resident Gram CGNR: no listening instantiation for this shape");
  }
  if (full) RLS_LAUNCH_GR(1, true);
  else RLS_LAUNCH_GR(1, false);
#undef RLS_LAUNCH_GR
  return launch_status(ctx);
}

// every Gram-pipeline shape whose workgroups (8 / 16 rows of AHA each) fit the CUs at one per CU: N <= 2048 CF32,
// N <= 4096 F32 on 256 CUs; ragged N runs the masked instantiation
template <typename E>
static bool gram_resident_ok_typed(int device, int64_t N) {
  int K = 0;
  if (!gram_pick<E>(N, &K)) return false;
  const int nwg = rls_gram_pipe_nwg(elem<E>::cplx ? RLS_C32 : RLS_F32, N);
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return false;
  if (!(nwg <= cus && nwg <= 256)) return false;
  int blocks = 0;
  hipError_t e = hipSuccess;
  if (K == 8) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, cgnr_gram_resident_kernel<E, 8, 1, false>, 512, 0);
  else if (K == 16) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, cgnr_gram_resident_kernel<E, 16, 1, false>, 512, 0);
  else if constexpr (!elem<E>::cplx)  // complex K = 32 would be N > 2048: more than 256 workgroups, never resident
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, cgnr_gram_resident_kernel<E, 32, 1, false>, 512, 0);
  (void)hipGetLastError();
  return e == hipSuccess && blocks >= 1;
}

template <typename E>
static int32_t gram_resident_typed(rls_ctx* ctx, const rls_gram_pipe& P, void* sync, int n_steps, unsigned spin_limit,
                                   const rls_cg_start& St) {
  int K = 0;
  if (!gram_pick<E>(P.N, &K)) return rls_fail(ctx, RLS_E_UNSUPPORTED, "resident Gram CGNR: N too large");
  const int nwg = rls_gram_pipe_nwg(elem<E>::cplx ? RLS_C32 : RLS_F32, P.N);
  if (K == 8) return launch_gram_resident<E, 8>(ctx, P, sync, nwg, n_steps, spin_limit, St);
  if (K == 16) return launch_gram_resident<E, 16>(ctx, P, sync, nwg, n_steps, spin_limit, St);
  if constexpr (!elem<E>::cplx) return launch_gram_resident<E, 32>(ctx, P, sync, nwg, n_steps, spin_limit, St);
  return rls_fail(ctx, RLS_E_UNSUPPORTED, "resident Gram CGNR: shape not resident");
}
template <typename E, int K>
static int32_t launch_fista_gram_resident(rls_ctx* ctx, const rls_fista_gram& P, void* sync, int nwg, int n_steps,
                                          unsigned spin_limit, const rls_srv_args& Sv) {
  using C = slab_cfg<E, 4, K, 8>;
  const int64_t Mc = P.N / C::NV;
  const int pair = (nwg % 16 == 0) ? 1 : 0;
  const bool full = P.N == C::NMAX && (int64_t)nwg * 4 == Mc;
#define RLS_LAUNCH_FGR(BB, FF, SS)                                                                                         \
  hipLaunchKernelGGL((fista_gram_resident_kernel<E, K, BB, FF, SS>), dim3(nwg), dim3(C::NT), 0, ctx->stream, (const E*)P.G, \
                     P.ldg, (E*)P.b0, (E*)P.b1, (const E*)P.x0, (E*)P.res, (E*)P.y0, (E*)P.y1, (E*)P.rr[0], (E*)P.rr[1], \
                     P.sc[0], P.sc[1], (resident_sync*)sync, Mc, P.N, pair, n_steps, spin_limit, Sv)
  if constexpr (K != 32) {
    if (Sv.ctl) {  // the listening instantiation (the host asks rls_gram_resident_server_ok first)
      if (ctx->tune.resident_ahead) {  // one iteration ahead of its commands
        if (full) RLS_LAUNCH_FGR(1, true, 2);
        else RLS_LAUNCH_FGR(1, false, 2);
      } else {
        if (full) RLS_LAUNCH_FGR(1, true, 1);
        else RLS_LAUNCH_FGR(1, false, 1);
      }
      return launch_status(ctx);
    }
  } else if (Sv.ctl) {
    return rls_fail(ctx, RLS_E_UNSUPPORTED, "resident Gram FISTA: no listening instantiation for this shape");
  }
  if (full) RLS_LAUNCH_FGR(1, true, 0);
  else RLS_LAUNCH_FGR(1, false, 0);
#undef RLS_LAUNCH_FGR
  return launch_status(ctx);
}

template <typename E>
static int32_t fista_gram_resident_typed(rls_ctx* ctx, const rls_fista_gram& P, void* sync, int n_steps,
                                         unsigned spin_limit, const rls_srv_args& Sv) {
  int K = 0;
  if (!gram_pick<E>(P.N, &K)) return rls_fail(ctx, RLS_E_UNSUPPORTED, "resident Gram FISTA: N too large");
  const int nwg = rls_gram_pipe_nwg(elem<E>::cplx ? RLS_C32 : RLS_F32, P.N);
  if (K == 8) return launch_fista_gram_resident<E, 8>(ctx, P, sync, nwg, n_steps, spin_limit, Sv);
  if (K == 16) return launch_fista_gram_resident<E, 16>(ctx, P, sync, nwg, n_steps, spin_limit, Sv);
  if constexpr (!elem<E>::cplx) return launch_fista_gram_resident<E, 32>(ctx, P, sync, nwg, n_steps, spin_limit, Sv);
  return rls_fail(ctx, RLS_E_UNSUPPORTED, "resident Gram FISTA: shape not resident");
}
template <typename E, int G, int K, int WV>
static int32_t launch_pgm_resident(rls_ctx* ctx, const rls_pgm_desc& D, const rls_pgm_coefs& CF, void* sync, int nwg, int n_steps,
                                   unsigned spin_limit) {
  using C = slab_cfg<E, G, K, WV>;
  if constexpr ((K == 32 || K == 16) && WV == 8 && owner_cfg_ok<E, G, K, WV>() && !(elem<E>::cplx && G == 4)) {
    const int64_t Mc = D.M / C::NV;
    const int pair = (G == 4 && nwg % 16 == 0) ? 1 : 0;
    const bool full = D.N == C::NMAX && (int64_t)nwg * G == Mc;
    constexpr size_t lds = resident_lds_bytes<E, G, K, WV>();
    static rls_device_once attr_once;
    if (auto once_ = attr_once.first(ctx->device)) {
#define RLS_PGM_ATTR(BB, FF, KK2) allow_big_lds(&pgm_resident_kernel<E, G, K, WV, BB, FF, KK2>, lds);
      RLS_PGM_ATTR(1, true, 0) RLS_PGM_ATTR(2, true, 0) RLS_PGM_ATTR(1, false, 0) RLS_PGM_ATTR(2, false, 0)
      RLS_PGM_ATTR(1, true, 1) RLS_PGM_ATTR(2, true, 1) RLS_PGM_ATTR(1, false, 1) RLS_PGM_ATTR(2, false, 1)
      RLS_PGM_ATTR(1, true, 2) RLS_PGM_ATTR(2, true, 2) RLS_PGM_ATTR(1, false, 2) RLS_PGM_ATTR(2, false, 2)
#undef RLS_PGM_ATTR
    }
#define RLS_LAUNCH_PGM(BB, FF, KK2)                                                                                              \
  hipLaunchKernelGGL((pgm_resident_kernel<E, G, K, WV, BB, FF, KK2>), dim3(nwg), dim3(C::NT), lds, ctx->stream, (const E*)D.A,    \
                     D.lda, (E*)D.v0, (E*)D.v1, (E*)D.v2, (E*)D.v3, (E*)D.o0, (E*)D.res, (const E*)D.x0, (E*)D.raw, (E*)D.slab, D.st, CF,  \
                     D.norm_x0, D.rel_tol, D.reg_kind, D.proj_kind, (resident_sync*)sync, Mc, D.N, pair | (ctx->tune.fista_defer ? 0 : 2), n_steps, D.first_it, spin_limit)
    const bool two = resident_two_level_ok<E>(ctx->tune, nwg, D.N, C::NT);
    if (D.kind == 0) {
      if (two) { if (full) RLS_LAUNCH_PGM(2, true, 0); else RLS_LAUNCH_PGM(2, false, 0); }
      else { if (full) RLS_LAUNCH_PGM(1, true, 0); else RLS_LAUNCH_PGM(1, false, 0); }
    } else if (D.kind == 1) {
      if (two) { if (full) RLS_LAUNCH_PGM(2, true, 1); else RLS_LAUNCH_PGM(2, false, 1); }
      else { if (full) RLS_LAUNCH_PGM(1, true, 1); else RLS_LAUNCH_PGM(1, false, 1); }
    } else {
      if (two) { if (full) RLS_LAUNCH_PGM(2, true, 2); else RLS_LAUNCH_PGM(2, false, 2); }
      else { if (full) RLS_LAUNCH_PGM(1, true, 2); else RLS_LAUNCH_PGM(1, false, 2); }
    }
#undef RLS_LAUNCH_PGM
    return launch_status(ctx);
  } else {
    return rls_fail(ctx, RLS_E_UNSUPPORTED, "resident OptISTA / POGM: slab shape not instantiated");
  }
}

template <typename E>
static int32_t pgm_resident_typed(rls_ctx* ctx, const rls_pgm_desc& D, const rls_pgm_coefs& CF, void* sync, int n_steps,
                                  unsigned spin_limit) {
  fused_cfg c;
  if (!pick_cfg<E>(ctx->tune, D.N, &c)) return rls_fail(ctx, RLS_E_UNSUPPORTED, "resident OptISTA / POGM: N too large");
  const int nwg = (int)fused_nwg<E>(ctx->tune, D.M, D.N);
  int32_t st = RLS_E_UNSUPPORTED;
#define RLS_PGM_CASE(GG, KK, WW) \
  if (c.G == GG && c.K == KK && c.WV == WW) st = launch_pgm_resident<E, GG, KK, WW>(ctx, D, CF, sync, nwg, n_steps, spin_limit);
  RLS_FOR_EACH_CFG(RLS_PGM_CASE)
#undef RLS_PGM_CASE
  return st;
}

template <typename E>
static bool pgm_resident_ok_typed(const rls_tuning& T, int device, int64_t M, int64_t N, const void* A, int64_t lda) {
  if (!resident_ok_typed<E>(T, device, M, N, A, lda)) return false;
  fused_cfg c;
  if (!pick_cfg<E>(T, N, &c)) return false;
  bool ok = false;
#define RLS_PGM_OK(GG, KK, WW) \
  if (c.G == GG && c.K == KK && c.WV == WW) ok = owner_cfg_ok<E, GG, KK, WW>() && (KK == 16 || KK == 32) && WW == 8 && !(elem<E>::cplx && GG == 4);
  RLS_FOR_EACH_CFG(RLS_PGM_OK)
#undef RLS_PGM_OK
  return ok;
}
}  // namespace

int32_t rls_fista_gram_iteration(rls_ctx* ctx, int32_t dtype, const rls_fista_gram& P, int parity) {
  if (dtype == RLS_F32) return fista_gram_iteration_typed<float>(ctx, P, parity & 1);
  return fista_gram_iteration_typed<float2>(ctx, P, parity & 1);
}
int32_t rls_fista_gram_finish(rls_ctx* ctx, int32_t dtype, const rls_fista_gram& P, int parity) {
  if (dtype == RLS_F32) return fista_gram_finish_typed<float>(ctx, P, parity & 1);
  return fista_gram_finish_typed<float2>(ctx, P, parity & 1);
}

int rls_gram_pipe_nwg(int32_t dtype, int64_t N) {
  const int V = dtype == RLS_C32 ? 2 : 4;
  const int64_t Mc = N / V;
  return (int)((Mc + 3) / 4);
}
bool rls_gram_pipe_ok(int32_t dtype, int64_t N, const void* G, int64_t ldg) {
  const int V = dtype == RLS_C32 ? 2 : 4;
  if (!G || N <= 0 || N % V || ldg % V || (reinterpret_cast<uintptr_t>(G) % 16)) return false;
  if (N > 32 * 128 || N > 4 * FIN_THREADS) return false;
  if (rls_gram_pipe_nwg(dtype, N) > 512) return false;  // the partial dots are summed one per thread (NT = 512)
  return 128 * ldg * (int64_t)(dtype == RLS_C32 ? 8 : 4) + (N / V) * 16 < (int64_t)0xffffffffll;
}
int32_t rls_gram_pipe_iteration(rls_ctx* ctx, int32_t dtype, const rls_gram_pipe& P, int parity) {
  if (dtype == RLS_F32) return gram_iteration_typed<float>(ctx, P, parity & 1);
  return gram_iteration_typed<float2>(ctx, P, parity & 1);
}
int32_t rls_gram_pipe_finish(rls_ctx* ctx, int32_t dtype, const rls_gram_pipe& P, int parity) {
  if (dtype == RLS_F32) return gram_finish_typed<float>(ctx, P, parity & 1);
  return gram_finish_typed<float2>(ctx, P, parity & 1);
}

int32_t rls_fista_gram_resident_launch(rls_ctx* ctx, int32_t dtype, const rls_fista_gram& P, void* sync, int n_steps,
                                       unsigned spin_limit, const rls_srv_args& Sv) {
  if (dtype == RLS_F32) return fista_gram_resident_typed<float>(ctx, P, sync, n_steps, spin_limit, Sv);
  return fista_gram_resident_typed<float2>(ctx, P, sync, n_steps, spin_limit, Sv);
}
bool rls_gram_resident_ok(rls_ctx* ctx, int32_t dtype, int64_t N, const void* G, int64_t ldg) {
  if (!rls_gram_pipe_ok(dtype, N, G, ldg)) return false;
  return dtype == RLS_F32 ? gram_resident_ok_typed<float>(ctx->device, N) : gram_resident_ok_typed<float2>(ctx->device, N);
}
// shapes whose resident Gram kernel has a listening (server mode) instantiation: every one but the 32-columns-per-row-piece slabs
bool rls_gram_resident_server_ok(int32_t dtype, int64_t N) {
  int K = 0;
  const bool ok = dtype == RLS_F32 ? gram_pick<float>(N, &K) : gram_pick<float2>(N, &K);
  return ok && K != 32;
}
int32_t rls_gram_resident_launch(rls_ctx* ctx, int32_t dtype, const rls_gram_pipe& P, void* sync, int n_steps,
                                 unsigned spin_limit, const rls_cg_start& St) {
  if (dtype == RLS_F32) return gram_resident_typed<float>(ctx, P, sync, n_steps, spin_limit, St);
  return gram_resident_typed<float2>(ctx, P, sync, n_steps, spin_limit, St);
}


int32_t rls_fista_pipe_iteration(rls_ctx* ctx, int32_t dtype, const rls_fista_pipe& P) {
  if (dtype == RLS_F32) return fista_iteration_typed<float>(ctx, P);
  return fista_iteration_typed<float2>(ctx, P);
}
int32_t rls_fista_pipe_finish(rls_ctx* ctx, int32_t dtype, const rls_fista_pipe& P) {
  if (dtype == RLS_F32) return fista_finish_typed<float>(ctx, P);
  return fista_finish_typed<float2>(ctx, P);
}

#ifdef RLS_STAMPS
extern "C" int32_t rls_debug_stamps(unsigned long long* out_h) {
  return (int32_t)hipMemcpyFromSymbol(out_h, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16 * 8);
}
#endif

int32_t rls_cgnr_pipe_iteration(rls_ctx* ctx, int32_t dtype, const rls_cgnr_pipe& P) {
  if (dtype == RLS_F32) return pipe_iteration_typed<float>(ctx, P);
  return pipe_iteration_typed<float2>(ctx, P);
}
// which: 1 = only the normal-operator kernel K_A, 2 = only the reduce kernel K_R (both are
// idempotent when repeated: K_A reads the committed scalars and K_R the staged ones)
int32_t rls_cgnr_pipe_launch(rls_ctx* ctx, int32_t dtype, const rls_cgnr_pipe& P, int which) {
  if (dtype == RLS_F32) return pipe_iteration_typed<float>(ctx, P, which);
  return pipe_iteration_typed<float2>(ctx, P, which);
}
int32_t rls_cgnr_pipe_finish(rls_ctx* ctx, int32_t dtype, const rls_cgnr_pipe& P) {
  if (dtype == RLS_F32) return pipe_finish_typed<float>(ctx, P);
  return pipe_finish_typed<float2>(ctx, P);
}

size_t rls_normal_fused_workspace(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda) {
  if (!ctx) return 0;
  if (dtype == RLS_F32) return fused_ok<float>(ctx->tune, M, N, A, lda) ? (size_t)fused_nwg<float>(ctx->tune, M, N) * N * 4 : 0;
  if (dtype == RLS_C32) return fused_ok<float2>(ctx->tune, M, N, A, lda) ? (size_t)fused_nwg<float2>(ctx->tune, M, N) * N * 8 : 0;
  return 0;
}

int32_t rls_launch_normal_fused(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda,
                                const void* p, void* v, void* slab, const int* skip) {
  RLS_CHECK_CTX(ctx);
  if (!A || !p || !v || !slab) return rls_fail(ctx, RLS_E_INVALID, "normal_fused: null pointer");
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F32)
    return normal_typed<float>(ctx, M, N, (const float*)A, lda, (const float*)p, (float*)v, (float*)slab, skip);
  return normal_typed<float2>(ctx, M, N, (const float2*)A, lda, (const float2*)p, (float2*)v, (float2*)slab, skip);
}

// resident CGNR (one launch per step call, A in registers across iterations)
size_t rls_cgnr_resident_sync_bytes() { return sizeof(resident_sync); }
size_t rls_resident_sync_alloc_bytes(int32_t dtype, int64_t N) {  // + [2 parities][RES_GROUPS][N] group-partial vectors
  // (+ [2 parities][RES_GROUPS] group sums of the scalar that rides along in the CGNR exchange)
  return sizeof(resident_sync) + (size_t)2 * RES_GROUPS * (size_t)N * rls_elem_size(dtype) + (size_t)2 * RES_GROUPS * sizeof(double);
}
size_t rls_resident_sync_clear_bytes() { return offsetof(resident_sync, failed); }
size_t rls_resident_sync_flags_offset() { return offsetof(resident_sync, fail); }
size_t rls_resident_sync_placement_offset() { return offsetof(resident_sync, gcnt) + sizeof(unsigned); }  // the word resident_report_placement sets
bool rls_cgnr_resident_ok(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda) {
  if (!ctx) return false;
  if (dtype == RLS_F32) return resident_ok_typed<float>(ctx->tune, ctx->device, M, N, A, lda);
  if (dtype == RLS_C32) return resident_ok_typed<float2>(ctx->tune, ctx->device, M, N, A, lda);
  return false;
}
int rls_cgnr_resident_nwg(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N) {
  return dtype == RLS_F32 ? (int)fused_nwg<float>(ctx->tune, M, N) : (int)fused_nwg<float2>(ctx->tune, M, N);
}
int32_t rls_cgnr_resident_launch(rls_ctx* ctx, int32_t dtype, const rls_cgnr_pipe& P, double* dout, void* sync,
                                 int n_steps, unsigned spin_limit, const rls_cg_start& St) {
  if (dtype == RLS_F32) return resident_typed<float>(ctx, P, dout, sync, n_steps, spin_limit, St);
  return resident_typed<float2>(ctx, P, dout, sync, n_steps, spin_limit, St);
}

int32_t rls_fista_resident_launch(rls_ctx* ctx, int32_t dtype, const rls_fista_pipe& P, void* sync, int n_steps,
                                  unsigned spin_limit, const rls_srv_args& Sv) {
  if (dtype == RLS_F32) return fista_resident_typed<float>(ctx, P, sync, n_steps, spin_limit, Sv);
  return fista_resident_typed<float2>(ctx, P, sync, n_steps, spin_limit, Sv);
}

bool rls_pgm_resident_ok(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda) {
  if (!ctx) return false;
  if (dtype == RLS_F32) return pgm_resident_ok_typed<float>(ctx->tune, ctx->device, M, N, A, lda);
  if (dtype == RLS_C32) return pgm_resident_ok_typed<float2>(ctx->tune, ctx->device, M, N, A, lda);
  return false;
}
int32_t rls_pgm_resident_launch(rls_ctx* ctx, int32_t dtype, const rls_pgm_desc& D, const rls_pgm_coefs& C, void* sync,
                                int n_steps, unsigned spin_limit) {
  if (n_steps > RLS_PGM_MAX_IT) return rls_fail(ctx, RLS_E_INVALID, "pgm_resident: more iterations than coefficient slots");
  if (dtype == RLS_F32) return pgm_resident_typed<float>(ctx, D, C, sync, n_steps, spin_limit);
  return pgm_resident_typed<float2>(ctx, D, C, sync, n_steps, spin_limit);
}
