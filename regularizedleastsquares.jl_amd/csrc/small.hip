// Small systems: a whole rls_cgnr_step call as ONE single-workgroup launch, A held in one CU's registers.
//
// The reference's own test and documentation sizes (test/testSolvers.jl:3-43, docs/src/literate/howto/gpu_acceleration.jl:12-23:
// 32 x 16; BASELINE configs[0]: 256 x 128 Float32) are launch-bound on the per-iteration kernels: two launches per iteration cost
// more than the arithmetic (8.9 us per iteration at 256 x 128).  Here M N s <= 128 KiB fits the register file of ONE CU, so one
// workgroup of 512 threads runs every iteration of a step call (src/CGNR.jl:143-178) with workgroup barriers only -- no grid
// exchange, no co-residency requirement, nothing to time out:
//   * thread (row block rb = tid / 16, column block cb = tid % 16) keeps the R x C tile A[rb R .. +R][cb C .. +C]: the 16 column
//     blocks of a row block are the 16 lanes of one DPP row, so t = A p is C FMAs per row and a 4-step DPP butterfly (every lane
//     of the row ends up with its R entries of t);
//   * v = A^H t: R FMAs per column, two cross-row shuffles inside the wave, the 8 waves' partials through LDS;
//   * wave 0 alone applies the CG update (alpha, x, r, beta, p; Float64 dots as a wave reduction) and leaves p in LDS: two
//     workgroup barriers per iteration.
// State (x, r, p, v, scalars) is read at entry and written back at the end, in the layout every other path uses, so step calls
// of this kernel and of the pipelines can follow each other.
#include "rls_common.hpp"
#include "resident_sync.hpp"

namespace {

constexpr int SM_NT = 512, SM_WV = SM_NT / 64, SM_RB = SM_NT / 16;  // 32 row blocks x 16 column blocks

template <typename E>
__device__ static inline E row16_sum(E v) {  // all-reduce over the 16 lanes of a DPP row, result in every lane, fixed order
  if constexpr (elem<E>::cplx) {
    float re = v.x, im = v.y;
    re += dpp_f(re, 0xB1); im += dpp_f(im, 0xB1);
    re += dpp_f(re, 0x4E); im += dpp_f(im, 0x4E);
    re += dpp_f(re, 0x141); im += dpp_f(im, 0x141);
    re += dpp_f(re, 0x140); im += dpp_f(im, 0x140);
    return make_float2(re, im);
  } else {
    v += dpp_f(v, 0xB1);
    v += dpp_f(v, 0x4E);
    v += dpp_f(v, 0x141);
    v += dpp_f(v, 0x140);
    return v;
  }
}
template <typename E>
__device__ static inline E shfl_xor_e(E v, int m) {
  if constexpr (elem<E>::cplx) return make_float2(__shfl_xor(v.x, m, 64), __shfl_xor(v.y, m, 64));
  else return __shfl_xor(v, m, 64);
}

// this thread's R x C tile of A (zero outside the matrix)
template <typename E, int R, int C>
__device__ static inline void small_tile_load(E (&a)[R][C], const E* __restrict__ A, int64_t lda, int M, int N, int vec16, int cb,
                                              int rb) {
  // A thread's R rows of a column are contiguous: with a 16-byte aligned A (vec16) they come in as 16-byte pieces -- R C s / 16
  // loads per lane instead of R C element loads that each touch a cache line of their own (a one-iteration call of the
  // 256 x 128 Float32 system spent ~8 of its ~11 us of kernel time on them).  Pieces that stick out of the matrix are read
  // element by element (ragged M only).
  constexpr int NVE = 16 / (int)sizeof(E);
  if (R % NVE == 0 && vec16) {
    constexpr int NQ = R % NVE == 0 ? R / NVE : 1;
#pragma unroll
    for (int j = 0; j < C; ++j) {
      const int col = cb * C + j;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int row0 = rb * R + q * NVE;
        const bool full = row0 + NVE <= M && col < N;
        const chunk<E, NVE> piece = load_chunk<E, NVE>(A + (full ? (int64_t)col * lda + row0 : 0));
#pragma unroll
        for (int k = 0; k < NVE; ++k) a[(q * NVE + k) % R][j] = full ? piece.e[k] : elem<E>::zero();
        if (!full && col < N && row0 < M) {
#pragma unroll
          for (int k = 0; k < NVE; ++k)
            if (row0 + k < M) a[(q * NVE + k) % R][j] = A[(int64_t)col * lda + row0 + k];
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < C; ++j) {
      const int col = cb * C + j;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int row = rb * R + i;
        const bool ok = row < M && col < N;
        const E val = A[(int64_t)(ok ? col : 0) * lda + (ok ? row : 0)];
        a[i][j] = ok ? val : elem<E>::zero();
      }
    }
  }
}

// one application of the normal operator to the vector in `ps`: the per-wave partial rows of A^H (A ps) in vpart, complete
// behind the workgroup barrier this ends with
template <typename E, int R, int C>
__device__ static inline void small_adjoint_partials(const E (&a)[R][C], const E (&t)[R], E (*vpart)[16 * C], int cb, int lane, int w);
template <typename E, int R, int C>
__device__ static inline void small_normal_partials(const E (&a)[R][C], const E* ps, E (*vpart)[16 * C], int cb, int lane, int w) {
  // ---- t = A p: this thread's R rows over its C columns, then over the 16 column blocks ---------------------------------------
  E pj[C];
#pragma unroll
  for (int j = 0; j < C; ++j) pj[j] = ps[cb * C + j];
  E t[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    E s = elem<E>::zero();
#pragma unroll
    for (int j = 0; j < C; ++j) s = elem<E>::fma(a[i][j], pj[j], s);
    t[i] = row16_sum<E>(s);
  }
  small_adjoint_partials<E, R, C>(a, t, vpart, cb, lane, w);
}

// the per-wave partial rows of A^H t (t: this thread's R rows, the same in the 16 lanes of its row block), complete behind the
// workgroup barrier this ends with
template <typename E, int R, int C>
__device__ static inline void small_adjoint_partials(const E (&a)[R][C], const E (&t)[R], E (*vpart)[16 * C], int cb, int lane, int w) {
  // ---- v = A^H t: this thread's C columns over its R rows, then over the 4 row blocks of the wave, then over the waves ---------
#pragma unroll
  for (int j = 0; j < C; ++j) {
    E s = elem<E>::zero();
#pragma unroll
    for (int i = 0; i < R; ++i) s = elem<E>::fmac(a[i][j], t[i], s);  // conj(a) t
    s = elem<E>::add(s, shfl_xor_e<E>(s, 16));
    s = elem<E>::add(s, shfl_xor_e<E>(s, 32));
    if (lane < 16) vpart[w][cb * C + j] = s;
  }
  __syncthreads();
}

// server mode of the single-workgroup kernels: ONE thread polls the control block (protocol of resident_listen, resident_sync.hpp:
// idle timeout, the "leaving" handshake, a bounded life) and leaves the command for the workgroup in LDS
__device__ static inline void small_listen(const rls_srv_args& srv, unsigned srv_seq, unsigned* cmd_out, unsigned* mbseq_out) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the write-back is out before anything else is announced)
  unsigned* ctl = srv.ctl;
  const unsigned long long t0 = wall_clock64(), idle = (unsigned long long)(srv.idle_us & 0x7fffffffu) * 100ull;  // (bit 31: RLS_SRV_AHEAD)
  unsigned n = RLS_SRV_EXIT;
  srv_head hd{srv_seq, 0u, 0u, 0u};  // (one 16-byte read of the control block's head per poll: srv_read_head, resident_sync.hpp)
  for (; srv_seq - srv.seq0 + 1u < 2048u;) {
    hd = srv_read_head(ctl);
    if (hd.seq != srv_seq) break;
    if (wall_clock64() - t0 > idle) {
      __hip_atomic_store(ctl + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      hd = srv_read_head(ctl);
      if (hd.seq != srv_seq) __hip_atomic_store(ctl + 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  if (hd.seq != srv_seq) n = hd.n;
  *cmd_out = n;
  *mbseq_out = hd.mbseq;
  if (n == RLS_SRV_EXIT) __hip_atomic_store(ctl + 17, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One workgroup per system of the group (blockIdx.x): K independent small problems -- each with its own A -- advance in ONE launch.
template <typename E, int R, int C>
__global__ __launch_bounds__(SM_NT) void cgnr_small_kernel(const rls_small_group G, int n_steps) {
  const rls_small& D = G.d[blockIdx.x];
  const E* __restrict__ A = (const E*)D.A;
  const int64_t lda = D.lda;
  const int M = (int)D.M, N = (int)D.N;
  E *x = (E*)D.x, *r = (E*)D.r, *p = (E*)D.p, *v = (E*)D.v;
  cgnr_scalars* sc = D.sc;
  const rls_mailbox_slot mb = D.mb;
  const int vec16 = (int)((reinterpret_cast<uintptr_t>(D.A) & 15) == 0 && (lda * (int64_t)sizeof(E)) % 16 == 0);
  constexpr int NP = 16 * C;              // padded vector length
  constexpr int EPT = (NP + 63) / 64;     // vector elements per lane of wave 0
  __shared__ E ps[NP];                    // p, zero beyond N
  __shared__ E vpart[SM_WV][NP];          // per-wave partial rows of v
  __shared__ int sdone;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int cb = lane & 15, rb = tid >> 4;
  // ---- A tile into registers (zero outside the matrix) -------------------------------------------------------------------------
  E a[R][C];
  small_tile_load<E, R, C>(a, A, lda, M, N, vec16, cb, rb);
  // ---- state: wave 0 owns the vectors (elements lane, lane + 64, ...) -----------------------------------------------------------
  E xv[EPT], rv[EPT], pv[EPT], vv[EPT];
  cgnr_scalars S;
  const bool init = D.b != nullptr;  // uniform: init! in this launch (src/CGNR.jl:107-130)
  if (init) {
    // r = A^H b: the thread's R entries of b are its t, then the second product alone
    const E* bb = (const E*)D.b;
    E t[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int row = rb * R + i;
      t[i] = row < M ? bb[row] : elem<E>::zero();
    }
    small_adjoint_partials<E, R, C>(a, t, vpart, cb, lane, w);
  }
  if (w == 0) {
    if (init) {
      double rr = 0.0;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int i = lane + 64 * e;
        E s = elem<E>::zero();
        if (i < NP) {
#pragma unroll
          for (int ww = 0; ww < SM_WV; ++ww) s = elem<E>::add(s, vpart[ww][i]);
        }
        if (i >= N) s = elem<E>::zero();
        xv[e] = elem<E>::zero();
        vv[e] = elem<E>::zero();
        rv[e] = s;
        pv[e] = s;
        if (i < NP) ps[i] = s;
        rr += (double)elem<E>::re(s) * (double)elem<E>::re(s) + (double)elem<E>::im(s) * (double)elem<E>::im(s);
      }
      rr = wave_sum(rr);
      S.rr = rr;                      // cgnr_init_kernel (solvers.hip)
      S.z0 = sqrt(rr);
      S.zeta = 0.0;
      S.alpha_re = S.alpha_im = S.beta_re = S.beta_im = 0.0;
      S.lambda = D.lambda;
      S.rel_tol = D.rel_tol;
      S.iteration = 0;
      S.max_iter = D.max_iter;
      const float ratio = (float)(sqrt(rr) / sqrt(rr));  // NaN when r == 0, as in the reference
      S.done = (ratio <= D.rel_tol) || (0 >= D.max_iter);
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int i = lane + 64 * e;
        const bool ok = i < N;
        xv[e] = ok ? x[i] : elem<E>::zero();
        rv[e] = ok ? r[i] : elem<E>::zero();
        pv[e] = ok ? p[i] : elem<E>::zero();
        vv[e] = ok ? v[i] : elem<E>::zero();
        if (i < NP) ps[i] = pv[e];
      }
      S.rr = sc->rr; S.z0 = sc->z0; S.zeta = sc->zeta;
      S.alpha_re = sc->alpha_re; S.alpha_im = sc->alpha_im; S.beta_re = sc->beta_re; S.beta_im = sc->beta_im;
      S.lambda = sc->lambda; S.rel_tol = sc->rel_tol;
      S.iteration = sc->iteration; S.max_iter = sc->max_iter; S.done = sc->done;
    }
    if (lane == 0) sdone = S.done;
  }
  __syncthreads();
  __shared__ unsigned srv_cmd, srv_mbseq;
  rls_mailbox_slot mbs = mb;
  unsigned srv_seq = D.srv.seq0;
  bool first = true;
  // server mode, RLS_SRV_AHEAD: behind the status and write-back of command k the workgroup computes iteration k + 1 at once -- under the
  // host's turnaround -- and only then listens (as the SPEC / SRV = 2 instantiations of the resident kernels, normal.hip): nothing of that
  // pass is published or stored before its command is there; told to leave, it leaves without a write-back
  const bool run_ahead = D.srv.ctl && (D.srv.idle_us >> 31);
  int credit = 0;      // iterations of the current command computed ahead of it
  bool ahead = false;  // the pass below runs ahead of its command
  for (;;) {  // (server mode: one pass per command; otherwise one pass)
  for (int it = credit; it < n_steps; ++it) {
    if (sdone) break;  // uniform
    small_normal_partials<E, R, C>(a, ps, vpart, cb, lane, w);
    // ---- the CG update, wave 0 alone (src/CGNR.jl:153-176) -----------------------------------------------------------------------
    if (w == 0) {
      double nre = 0.0, nim = 0.0, pp = 0.0;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int i = lane + 64 * e;
        E s = elem<E>::zero();
        if (i < NP) {
#pragma unroll
          for (int ww = 0; ww < SM_WV; ++ww) s = elem<E>::add(s, vpart[ww][i]);
        }
        vv[e] = s;
        nre += (double)elem<E>::re(pv[e]) * (double)elem<E>::re(s) + (double)elem<E>::im(pv[e]) * (double)elem<E>::im(s);
        if constexpr (elem<E>::cplx)
          nim += (double)elem<E>::re(pv[e]) * (double)elem<E>::im(s) - (double)elem<E>::im(pv[e]) * (double)elem<E>::re(s);
        pp += (double)elem<E>::re(pv[e]) * (double)elem<E>::re(pv[e]) + (double)elem<E>::im(pv[e]) * (double)elem<E>::im(pv[e]);
      }
      nre = wave_sum(nre);
      if constexpr (elem<E>::cplx) nim = wave_sum(nim);
      const float lambda = S.lambda;
      if (lambda > 0.f) pp = wave_sum(pp);
      const double zeta = S.rr;
      const dcomplex alpha = dc_div({zeta, 0.0}, {nre + (lambda > 0.f ? (double)lambda * pp : 0.0), nim});
      const E al = elem<E>::make((float)alpha.re, (float)alpha.im);
      const E na = elem<E>::make(-(float)alpha.re, -(float)alpha.im);
      double rr = 0.0;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        xv[e] = elem<E>::fma(pv[e], al, xv[e]);
        E ri = elem<E>::fma(vv[e], na, rv[e]);
        if (lambda > 0.f) ri = elem<E>::fma(elem<E>::scale(-lambda, pv[e]), al, ri);
        rv[e] = ri;
        rr += (double)elem<E>::re(ri) * (double)elem<E>::re(ri) + (double)elem<E>::im(ri) * (double)elem<E>::im(ri);
      }
      rr = wave_sum(rr);
      const double beta = rr / zeta;
      const float bf = (float)beta;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        pv[e] = elem<E>::add(elem<E>::scale(bf, pv[e]), rv[e]);
        const int i = lane + 64 * e;
        if (i < NP) ps[i] = pv[e];
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
      if (lane == 0) sdone = S.done;
    }
    __syncthreads();
  }
  credit = 0;
  if (!ahead && w == 0) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = lane + 64 * e;
      if (i < N) {
        x[i] = xv[e];
        r[i] = rv[e];
        p[i] = pv[e];
        v[i] = vv[e];
      }
    }
    if (lane == 0) {
      sc->rr = S.rr; sc->zeta = S.zeta;
      sc->alpha_re = S.alpha_re; sc->alpha_im = S.alpha_im; sc->beta_re = S.beta_re; sc->beta_im = S.beta_im;
      sc->iteration = S.iteration; sc->done = S.done;
      sc->pending = 0; sc->cur = 0; sc->fresh = 0;
      if (init && first) {
        sc->z0 = S.z0; sc->lambda = S.lambda; sc->rel_tol = S.rel_tol; sc->max_iter = S.max_iter;
      }
    }
    S.pending = 0; S.cur = 0; S.fresh = 0;
    rls_mailbox_publish(mbs, S, lane);
  }
  if (!D.srv.ctl) break;  // uniform
  if (run_ahead && !ahead && !sdone) {  // uniform (sdone: behind the loop's last barrier): one iteration ahead of the next command
    ahead = true;
    n_steps = 1;
    first = false;
    continue;
  }
  // ---- server mode: this workgroup stays and listens for the next step call (one CU; nothing else waits for it) ----------------
  if (w == 0 && lane == 0) small_listen(D.srv, srv_seq, &srv_cmd, &srv_mbseq);
  __syncthreads();
  const unsigned cmd = srv_cmd;
  if (cmd == RLS_SRV_EXIT) return;  // uniform (behind a pass ahead: memory holds the state of the last command served)
  n_steps = (int)cmd;
  mbs.seq = srv_mbseq;
  srv_seq += 1;
  first = false;
  if (ahead) {  // the command's first iteration is done
    ahead = false;
    credit = 1;
  }
  __syncthreads();  // (srv_cmd is read before the next pass can overwrite it)
  }
}

// ---- FISTA (src/FISTA.jl:139-185) on the same tile layout -----------------------------------------------------------------------
// A whole rls_fista_step call: per iteration res_raw = A^H (A y) from the registers, then -- wave 0 alone, as above -- the
// gradient step, prox, restart test, theta, the stopping test and the next extrapolated point (fista_update_elems of normal.hip
// for one wave).  State in and out in the pipeline's layout: x_k in buf[k & 1], x_{k-1} in the other buffer, the extrapolated
// point in y0 / y1 by `ycur`, nothing pending.
template <typename E, int R, int C>
__global__ __launch_bounds__(SM_NT) void fista_small_kernel(const E* __restrict__ A, int64_t lda, int M, int N, E* b0, E* b1,
                                                            const E* __restrict__ x0, E* res, E* y0, E* y1, fista_scalars* sc,
                                                            int n_steps, rls_mailbox_slot mb, int vec16, rls_srv_args srv) {
  constexpr int NP = 16 * C;
  constexpr int EPT = (NP + 63) / 64;
  __shared__ E ps[NP];             // the extrapolated point y, zero beyond N
  __shared__ E vpart[SM_WV][NP];
  __shared__ int sdone;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int cb = lane & 15, rb = tid >> 4;
  E a[R][C];
  small_tile_load<E, R, C>(a, A, lda, M, N, vec16, cb, rb);
  E xk[EPT], xo[EPT], yv[EPT], x0v[EPT], ri[EPT];
  fista_scalars S;
  int ran = 0;
  if (w == 0) {
    RLS_FISTA_COPY(S, *sc);
    const E* xc = (S.iteration & 1) ? b1 : b0;   // state.x == buf[iteration & 1]
    const E* xp = (S.iteration & 1) ? b0 : b1;
    const E* yc = S.ycur ? y1 : y0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = lane + 64 * e;
      const bool ok = i < N;
      xk[e] = ok ? xc[i] : elem<E>::zero();
      xo[e] = ok ? xp[i] : elem<E>::zero();
      yv[e] = ok ? yc[i] : elem<E>::zero();
      x0v[e] = ok ? x0[i] : elem<E>::zero();
      ri[e] = ok ? res[i] : elem<E>::zero();
      if (i < NP) ps[i] = yv[e];
    }
    if (lane == 0) sdone = S.done;
  }
  __syncthreads();
  __shared__ unsigned srv_cmd, srv_mbseq;
  rls_mailbox_slot mbs = mb;
  unsigned srv_seq = srv.seq0;
  const bool run_ahead = srv.ctl && (srv.idle_us >> 31);  // RLS_SRV_AHEAD: one iteration ahead of the next command (as cgnr_small_kernel)
  int credit = 0;
  bool ahead = false;
  for (;;) {  // (server mode: one pass per command; otherwise one pass)
  for (int it = credit; it < n_steps; ++it) {
    if (sdone) break;  // uniform
    small_normal_partials<E, R, C>(a, ps, vpart, cb, lane, w);
    if (w == 0) {
      const float rho = S.rho, thr = S.rho * S.lambda;  // prox!(reg, x, rho * lambda(reg))        :164
      E xn[EPT];
      double rn = 0.0, d = 0.0;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int i = lane + 64 * e;
        E raw = elem<E>::zero();
        if (i < NP) {
#pragma unroll
          for (int ww = 0; ww < SM_WV; ++ww) raw = elem<E>::add(raw, vpart[ww][i]);
        }
        E rr = elem<E>::sub(raw, x0v[e]);                                   // res .-= x0      :153
        E xv = elem<E>::sub(yv[e], elem<E>::scale(rho, rr));                // x .-= rho .* res :154
        xv = fista_proj_elem<E>(fista_prox_elem<E>(xv, S.reg_kind, thr), S.proj_kind);
        if (i >= N) {
          rr = elem<E>::zero();
          xv = elem<E>::zero();
        }
        ri[e] = rr;
        xn[e] = xv;
        rn += (double)elem<E>::re(rr) * (double)elem<E>::re(rr) + (double)elem<E>::im(rr) * (double)elem<E>::im(rr);
        const E df = elem<E>::sub(xv, xk[e]);
        d += (double)elem<E>::re(rr) * (double)elem<E>::re(df) + (double)elem<E>::im(rr) * (double)elem<E>::im(df);
      }
      rn = wave_sum(rn);
      d = wave_sum(d);
      float theta = S.theta;
      if (S.restart && d > 0.0) theta = 1.f;                                  // gradient restart  :171-176
      const float theta_old = theta;                                          // :179
      theta = (1.f + sqrtf(1.f + 4.f * theta_old * theta_old)) / 2.f;         // :180
      const double res_norm = sqrt(rn);
      const float rel = (float)(res_norm / S.norm_x0);                        // :156
      const int done = (rel < S.rel_tol) || (S.iteration + 1 >= S.max_iter);  // :187-189
      const float c1 = (1.f - theta_old) / theta, c2 = (theta_old - 1.f) / theta + 1.f;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const E yn = elem<E>::add(elem<E>::scale(c1, xk[e]), elem<E>::scale(c2, xn[e]));
        xo[e] = xk[e];
        xk[e] = xn[e];
        if (!done) {  // (a plan that stops keeps the point its last iteration was taken at, as the pipeline does)
          yv[e] = yn;
          const int i = lane + 64 * e;
          if (i < NP) ps[i] = yn;
        }
      }
      S.res_norm = res_norm;
      S.rel_res_norm = (double)rel;
      S.theta = theta;
      S.theta_old = theta_old;
      S.iteration += 1;
      S.done = done;
      if (!done) S.ycur = 1 - S.ycur;
      ran += 1;
      if (lane == 0) sdone = done;
    }
    __syncthreads();
  }
  credit = 0;
  if (!ahead && w == 0) {
    if (ran > 0) {
      E* xw = (S.iteration & 1) ? b1 : b0;
      E* xp = (S.iteration & 1) ? b0 : b1;
      E* yw = S.ycur ? y1 : y0;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int i = lane + 64 * e;
        if (i < N) {
          xw[i] = xk[e];
          xp[i] = xo[e];
          yw[i] = yv[e];
          res[i] = ri[e];
        }
      }
      S.pending = 0;
      S.fresh = 0;
      if (lane == 0) RLS_FISTA_COPY(*sc, S);
    }
    rls_mailbox_publish(mbs, S, lane);
  }
  if (!srv.ctl) break;  // uniform
  if (run_ahead && !ahead && !sdone) {  // uniform: one iteration ahead of the next command
    ahead = true;
    n_steps = 1;
    continue;
  }
  if (w == 0 && lane == 0) small_listen(srv, srv_seq, &srv_cmd, &srv_mbseq);
  __syncthreads();
  const unsigned cmd = srv_cmd;
  if (cmd == RLS_SRV_EXIT) return;  // uniform (behind a pass ahead: memory holds the state of the last command served)
  n_steps = (int)cmd;
  mbs.seq = srv_mbseq;
  srv_seq += 1;
  if (ahead) {  // the command's first iteration is done
    ahead = false;
    credit = 1;
  }
  __syncthreads();
  }
}

struct small_tile {
  int R, C;
};
// tiles (rows per row block x columns per column block); the matrix is padded to 32 R x 16 C.  R C s / 4 registers per lane.
template <typename E>
static bool small_pick(int64_t M, int64_t N, small_tile* t) {
  constexpr small_tile real_tiles[] = {{1, 1}, {2, 2}, {4, 2}, {4, 4}, {8, 4}, {8, 8}, {16, 4}};
  constexpr small_tile cplx_tiles[] = {{1, 1}, {2, 2}, {4, 2}, {4, 4}, {8, 4}};
  const small_tile* tiles = elem<E>::cplx ? cplx_tiles : real_tiles;
  const int n = elem<E>::cplx ? 5 : 7;
  for (int i = 0; i < n; ++i)
    if (M <= (int64_t)SM_RB * tiles[i].R && N <= (int64_t)16 * tiles[i].C) {
      *t = tiles[i];
      return true;
    }
  return false;
}

template <typename E, int R, int C>
static void small_launch(rls_ctx* ctx, const rls_small_group& G, int n_steps) {
  hipLaunchKernelGGL((cgnr_small_kernel<E, R, C>), dim3((unsigned)G.count), dim3(SM_NT), 0, ctx->stream, G, n_steps);
}

// one tile shape for the whole group: the smallest that holds its largest system
template <typename E>
static int32_t small_typed(rls_ctx* ctx, const rls_small_group& G, int n_steps) {
  int64_t Mx = 0, Nx = 0;
  for (int k = 0; k < G.count; ++k) {
    Mx = G.d[k].M > Mx ? G.d[k].M : Mx;
    Nx = G.d[k].N > Nx ? G.d[k].N : Nx;
  }
  small_tile t;
  if (G.count < 1 || G.count > RLS_SMALL_GROUP_MAX || !small_pick<E>(Mx, Nx, &t))
    return rls_fail(ctx, RLS_E_UNSUPPORTED, "small-system kernel: shape too large (or an empty / oversized group)");
#define SM_CASE(RR, CC)          \
  if (t.R == RR && t.C == CC) {  \
    small_launch<E, RR, CC>(ctx, G, n_steps); \
  } else
  SM_CASE(1, 1) SM_CASE(2, 2) SM_CASE(4, 2) SM_CASE(4, 4) SM_CASE(8, 4) {
    if constexpr (!elem<E>::cplx) {
      SM_CASE(8, 8) SM_CASE(16, 4) {}
    }
  }
#undef SM_CASE
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

}  // namespace

template <typename E>
static int32_t fista_small_typed(rls_ctx* ctx, const rls_fista_pipe& P, int n_steps, const rls_srv_args& Sv) {
  small_tile t;
  if (!small_pick<E>(P.M, P.N, &t)) return rls_fail(ctx, RLS_E_UNSUPPORTED, "small-system kernel: shape too large");
  const int vec16 = (int)((reinterpret_cast<uintptr_t>(P.A) & 15) == 0 && (P.lda * (int64_t)sizeof(E)) % 16 == 0);
#define SM_CASE(RR, CC)                                                                                                        \
  if (t.R == RR && t.C == CC) {                                                                                                \
    hipLaunchKernelGGL((fista_small_kernel<E, RR, CC>), dim3(1), dim3(SM_NT), 0, ctx->stream, (const E*)P.A, P.lda, (int)P.M,  \
                       (int)P.N, (E*)P.b0, (E*)P.b1, (const E*)P.x0, (E*)P.res, (E*)P.y0, (E*)P.y1, P.sc, n_steps, P.mb, vec16, Sv); \
  } else
  SM_CASE(1, 1) SM_CASE(2, 2) SM_CASE(4, 2) SM_CASE(4, 4) SM_CASE(8, 4) {
    if constexpr (!elem<E>::cplx) {
      SM_CASE(8, 8) SM_CASE(16, 4) {}
    }
  }
#undef SM_CASE
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

// a whole rls_fista_step call of a system rls_small_ok accepts (plan state in the pipeline's layout: P.b0 / b1, y0 / y1, sc)
int32_t rls_fista_small_launch(rls_ctx* ctx, int32_t dtype, const rls_fista_pipe& P, int n_steps, const rls_srv_args& Sv) {
  return dtype == RLS_F32 ? fista_small_typed<float>(ctx, P, n_steps, Sv) : fista_small_typed<float2>(ctx, P, n_steps, Sv);
}

bool rls_small_ok(int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda) {
  if (!A || M < 1 || N < 1 || lda < M) return false;
  small_tile t;
  return dtype == RLS_F32 ? small_pick<float>(M, N, &t) : small_pick<float2>(M, N, &t);
}

int32_t rls_small_group_launch(rls_ctx* ctx, int32_t dtype, const rls_small_group& G, int n_steps) {
  return dtype == RLS_F32 ? small_typed<float>(ctx, G, n_steps) : small_typed<float2>(ctx, G, n_steps);
}
int32_t rls_small_launch(rls_ctx* ctx, int32_t dtype, const rls_small& D, int n_steps) {
  rls_small_group G;
  G.count = 1;
  G.d[0] = D;
  return rls_small_group_launch(ctx, dtype, G, n_steps);
}
