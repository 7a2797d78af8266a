// Operator handle and the fused solver steps: CGNR (src/CGNR.jl:107-185), FISTA
// (src/FISTA.jl:110-189) and IterativeSolvers.cg! as ADMM calls it (src/ADMM.jl:244).
//
// One iteration = the normal-operator apply (two GEMVs, matrix-free; one in Gram mode) plus ONE
// single-workgroup "update" kernel that holds every scalar (alpha, beta, zeta, theta, residuals,
// iteration count, done flag) in device memory, so the host never synchronises inside the loop
// (the reference's generic GPU path pays 4-5 blocking scalar read-backs per iteration).
// Once `done` is set every later kernel of the plan exits at entry, which reproduces the
// reference's "stop at the first iteration where done() holds" exactly while letting the host
// enqueue iterations in hipGraph-captured chunks.
#include "rls_common.hpp"

#include <algorithm>
#include <mutex>
#include <vector>


// ---- allocation: the plans' scratch comes from the stream-ordered pool of the context named by the innermost
// rls_alloc_scope of the calling thread; the pinned status mirrors from the process-wide free list (rls_common.hpp) ----
template <typename T>
static hipError_t dmalloc(T** p, size_t bytes) { return rls_scoped_malloc(reinterpret_cast<void**>(p), bytes); }
static hipError_t dfree(void* p) { return rls_scoped_free(p); }
template <typename T>
static hipError_t hmalloc(T** p, size_t bytes) { return rls_pinned_alloc(reinterpret_cast<void**>(p), bytes); }
static void hfree(void* p) { rls_pinned_free(p); }
// the context a plan allocated from, if it still exists (plans may outlive their context in a garbage-collected host)
// (pointer AND generation id: a destroyed context's address can be handed out again to a new one, possibly on another device)
static rls_ctx* alloc_ctx_of(rls_ctx* ctx, uint64_t id) { return rls_ctx_alive(ctx, id) ? ctx : nullptr; }

// ---------------------------------------------------------------------------------------------
// operator
// ---------------------------------------------------------------------------------------------
struct rls_operator {
  rls_ctx* ctx;
  uint64_t ctx_id = 0;  // ctx->id at creation (alloc_ctx_of)
  int32_t dtype;
  int64_t M, N;
  const void* A;  // may be null (Gram-only operator)
  int64_t lda;
  const void* G;  // Gram matrix (N x N) or null
  int64_t ldg;
  void* t;        // length-M scratch for the matrix-free normal operator
  void* slab;     // per-workgroup partial-v slab of the fused one-pass normal operator (or null)
};

static int32_t op_normal(rls_operator* op, const void* p, void* v, const int* skip) {
  rls_ctx* ctx = op->ctx;
  if (op->G) return rls_launch_gemv(ctx, op->dtype, RLS_OP_N, op->N, op->N, 1.f, 0.f, op->G, op->ldg, p, 0.f, 0.f, v, skip);
  if (!op->A) return rls_fail(ctx, RLS_E_STATE, "operator has neither A nor a Gram matrix");
  if (op->slab && ctx->tune.fused_normal)
    return rls_launch_normal_fused(ctx, op->dtype, op->M, op->N, op->A, op->lda, p, v, op->slab, skip);
  RLS_TRY(rls_launch_gemv(ctx, op->dtype, RLS_OP_N, op->M, op->N, 1.f, 0.f, op->A, op->lda, p, 0.f, 0.f, op->t, skip));
  return rls_launch_gemv(ctx, op->dtype, RLS_OP_C, op->M, op->N, 1.f, 0.f, op->A, op->lda, op->t, 0.f, 0.f, v, skip);
}

// ---------------------------------------------------------------------------------------------
// hipGraph chunking shared by the three plans
// ---------------------------------------------------------------------------------------------
struct step_graph {
  hipGraphExec_t exec = nullptr;
  int steps = 0;
  void* x_bound = nullptr;  // cg plans: the solution vector whose address the captured kernels carry
  int mode = 0;  // which kernel sequence was captured
  bool failed = false;
  uint64_t epoch = 0;  // rls_ctx::tune_epoch at capture: launch shapes and kernel choices follow the context's switches
};

// `rewind(c)`: a capture that fails has already called enqueue_one c times WITHOUT any of those launches running; a
// caller whose callback keeps a position (parity, launch index) gets the chance to step it back before the eager retry.
template <typename F, typename R>
static int32_t run_steps(rls_ctx* ctx, step_graph* big, int n_steps, F&& enqueue_one, R&& rewind) {
  const int chunk = ctx->tune.graph_chunk;
  while (n_steps > 0) {
    if (ctx->tune.use_graph && chunk > 1 && n_steps >= chunk && !big->failed) {
      if (!big->exec || big->steps != chunk || big->epoch != ctx->tune_epoch) {
        if (big->exec) {
          hipGraphExecDestroy(big->exec);
          big->exec = nullptr;
        }
        hipGraph_t graph = nullptr;
        std::unique_lock<std::mutex> capture_lock(rls_capture_mutex());
        hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed);
        int32_t st = 0;
        int calls = 0;
        if (e == hipSuccess) {
          for (int i = 0; i < chunk && st == 0; ++i, ++calls) st = enqueue_one();
          e = hipStreamEndCapture(ctx->stream, &graph);
        }
        capture_lock.unlock();
        if (e != hipSuccess || st != 0 || !graph ||
            hipGraphInstantiate(&big->exec, graph, nullptr, nullptr, 0) != hipSuccess) {
          big->failed = true;  // fall back to eager launches (still the HIP kernels)
          big->exec = nullptr;
          (void)hipGetLastError();
          rewind(calls);
        } else {
          big->steps = chunk;
          big->epoch = ctx->tune_epoch;
        }
        if (graph) hipGraphDestroy(graph);
        if (big->failed) continue;
      }
      RLS_HIP(ctx, hipGraphLaunch(big->exec, ctx->stream));
      n_steps -= chunk;
    } else {
      RLS_TRY(enqueue_one());
      n_steps -= 1;
    }
  }
  return 0;
}

template <typename F>
static int32_t run_steps(rls_ctx* ctx, step_graph* big, int n_steps, F&& enqueue_one) {
  return run_steps(ctx, big, n_steps, enqueue_one, [](int) {});
}

// Which (r, p) pair launch k of a step call finds current (rls_cgnr_pipe::cur_hint): the call starts at pair 0
// with nothing pending, launch 0 flips nothing, every later launch flips.  Launches at a chunk boundary may be
// the first node of a REPLAYED graph, whose captured hint belongs to another position in the sequence, so they
// get "unknown"; so does everything when the chunk length is odd (replays would be out of phase).
static inline int pipe_cur_hint(const rls_ctx* ctx, int k) {
  const int chunk = ctx->tune.graph_chunk;
  if (ctx->tune.pipe_hint_mode == 1) return -1;
  if (ctx->tune.pipe_hint_mode == 2) return k == 0 ? 1 : (k & 1);  // the opposite of the bookkeeping below
  if (k == 0) return (ctx->tune.use_graph && chunk > 1) ? -1 : 0;
  if (ctx->tune.use_graph && chunk > 1 && (chunk % 2 != 0 || k % chunk == 0)) return -1;
  return (k - 1) & 1;
}

// ---------------------------------------------------------------------------------------------
// CGNR
// ---------------------------------------------------------------------------------------------
// server mode of a plan's resident kernel (rls_cg_start::srv_ctl / rls_srv_args): the control block in pinned host memory and what
// the host knows about the kernel it left listening.  rls_ctx::server points at the one that is alive on the context's stream.
struct srv_state {
  unsigned* ctl = nullptr;
  bool alive = false;   // a kernel was left listening (it may have left on its own since: ctl[17])
  bool fresh = false;   // ... and the plan's status mirror holds the status of its last command
  bool off = false;     // lives that served fewer than three commands, twice in a row (the caller touches the device between
  int served = 0, short_lives = 0;  // iterates: a listening kernel only stands in its way): per-iteration pipeline until init!
  unsigned seq = 0;
  bool* resident_used = nullptr;  // the plan's flag: "a status call must read the sync block's flags" (set when a life ends)
};

struct rls_cgnr {
  rls_operator* op;
  rls_ctx* actx;  // the context whose pool the plan's scratch came from (checked alive before it is used in destroy)
  uint64_t actx_id = 0;
  int device;
  void *x, *r, *p, *v;
  cgnr_scalars* sc;    // device
  cgnr_scalars* sc_h;  // pinned host
  step_graph graph;
  bool initialised;
  // fused pipeline (normal.hip): alternate (r, p) pair, partial dots, staged scalars
  void *r1, *p1;
  double* dots;
  double* ttw = nullptr;
  cgnr_scalars* scn;
  // batched plans: nrhs right-hand sides, columns ldv elements apart, own partial-row slab
  int nrhs;
  int64_t ldv;
  void* slab_b;
  // Gram-mode pipeline (normal.hip): one launch per iteration, second parity of v and the partial dots
  bool gram_pipe;
  void* v1;
  double* gdots;
  // batched plans on the matrix cores (skinny.hip): packed operands + row-split partials
  bool skinny;
  float *Ppack, *Tpack;
  void* Vpart;
  int splits;
  int half;  // operand-panel layout, fixed at creation (rls_skinny_half)
  // resident mode (normal.hip, cgnr_resident_kernel): arrival counters + flags, per-workgroup partial dots
  void* rsync;
  double* rdots;
  unsigned* rsync_h;  // pinned: {fail, completed, failed} (resident_sync), read with the status
  bool resident_used;
  rls_mailbox_slot mb_arm;  // step_status: the call's last kernel publishes the scalars (pipeline and small-system paths)
  srv_state srv;  // server mode of the resident kernel (rls_cgnr_step_status)
  bool mb_sent = false;     // ... and this call's path did take the slot
  bool gram_resident;  // Gram mode: AHA fits the register files (rls_gram_resident_ok)
  // a resident launch whose workgroups were not all on the chip in time is a no-op (normal.hip); the status call re-runs
  // what was lost on the per-iteration pipeline and the plan stays there
  bool resident_off;
  int fallbacks;        // resident launches lost and recovered so far
  long long requested;  // iterations asked for since init
  bool rsync_clean = false;  // the init kernel has just zeroed the arrival counters (resident_chain)
  bool small = false;  // the system fits one CU's registers: a step call is ONE single-workgroup launch (small.hip)
  // batched plan on an explicit Gram matrix, <= 8 ComplexF32 columns, AHA in the register files (gramk.hip): exchange scratch
  bool gramk = false;
  float* gk_vx = nullptr;
  void* gk_xx = nullptr;
  double* gk_dots = nullptr;
  // rls_cgnr_solve_queue_host: this problem's right-hand side on the device and the pinned staging of b and x (created on first use)
  void* q_b = nullptr;
  void* q_bh = nullptr;
  void* q_xh = nullptr;
};

static bool cgnr_use_gram_pipeline(const rls_cgnr* s) {
  return s->gram_pipe && s->op->G && s->op->ctx->tune.gram_pipeline;
}

static rls_gram_pipe cgnr_gram_desc(const rls_cgnr* s) {
  rls_gram_pipe P;
  P.G = s->op->G;
  P.ldg = s->op->ldg;
  P.N = s->op->N;
  P.x = s->x;
  P.r[0] = s->r;
  P.r[1] = s->r1;
  P.p[0] = s->p;
  P.p[1] = s->p1;
  P.v[0] = s->v;
  P.v[1] = s->v1;
  P.dots = s->gdots;
  P.sc[0] = s->sc;
  P.sc[1] = s->scn;
  return P;
}

static rls_skinny cgnr_skinny_desc(const rls_cgnr* s) {
  rls_skinny K;
  K.A = s->op->A;
  K.lda = s->op->lda;
  K.M = s->op->M;
  K.N = s->op->N;
  K.G = s->op->G;  // explicit AHA (src/CGNR.jl:49): every operator apply of the batched loop is ONE product over it
  K.ldg = s->op->ldg;
  K.nrhs = s->nrhs;
  K.half = s->half;
  K.ngroups = rls_skinny_groups(s->nrhs, s->half);
  K.splits = s->splits;
  K.X = s->x;
  K.R = s->r;
  K.P = s->p;
  K.V = s->v;
  K.ldv = s->ldv;
  K.Ppack = s->Ppack;
  K.Tpack = s->Tpack;
  K.Vpart = s->Vpart;
  K.ldvp = s->op->N;
  K.sc = s->sc;
  return K;
}

// small systems: single right-hand side, matrix-free, A in ONE CU's registers for the whole step call (no grid exchange, no
// co-residency requirement: always available)
static bool cgnr_use_small(const rls_cgnr* s) {
  return s->small && s->nrhs == 1 && !s->op->G && s->op->ctx->tune.small && s->op->ctx->tune.resident;
}

// batched Gram mode as ONE resident launch per step call (a call of one iteration -- the callback cadence -- is cheaper on
// the streaming kernels: a resident launch loads its rows of AHA and gathers x once per call)
static bool cgnr_use_gramk(const rls_cgnr* s, int n_steps) {
  return s->gramk && s->rsync && !s->resident_off && s->op->ctx->tune.resident &&
         (n_steps != 1 || s->op->ctx->tune.resident == 2);  // resident = 2 (tools, tests): single-iteration calls too
}

static rls_gramk cgnr_gramk_desc(const rls_cgnr* s) {
  rls_gramk D;
  D.G = s->op->G;
  D.ldg = s->op->ldg;
  D.N = s->op->N;
  D.nrhs = s->nrhs;
  D.X = s->x;
  D.R = s->r;
  D.P = s->p;
  D.V = s->v;
  D.ldv = s->ldv;
  D.sc = s->sc;
  D.Vx = s->gk_vx;
  D.Xx = s->gk_xx;
  D.dots = s->gk_dots;
  D.Ppack = s->half ? s->Ppack : nullptr;  // (<= 8 complex columns: the half layout, unless switched off by the tools)
  return D;
}

static bool cgnr_use_gram_resident(const rls_cgnr* s) {
  return s->gram_resident && s->rsync && !s->resident_off && s->nrhs == 1 && cgnr_use_gram_pipeline(s) &&
         s->op->ctx->tune.resident;
}

static bool cgnr_use_pipeline(const rls_cgnr* s) {
  const rls_ctx* ctx = s->op->ctx;
  return s->r1 && s->op->slab && !s->op->G && ctx->tune.fused_normal && ctx->tune.cgnr_pipeline;
}

// the whole step call as one launch: single right-hand side, matrix-free, A small enough to stay in the register
// files (one workgroup per CU), 16-byte aligned state vectors
static bool cgnr_use_resident(const rls_cgnr* s) {
  const rls_ctx* ctx = s->op->ctx;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return s->rsync && !s->resident_off && s->nrhs == 1 && cgnr_use_pipeline(s) && ctx->tune.resident && al16(s->x) &&
         al16(s->r) && al16(s->p) && al16(s->v);
}

// A resident kernel needs every one of its workgroups on a CU at the same time.  Other kernels only delay that, but
// two resident kernels running side by side (two contexts = two streams of this process) could each hold CUs the
// other is waiting for.  So resident launches on one device form ONE chain across all streams of the process.  A launch
// on the stream that issued the previous one is ordered by the stream itself (no event traffic at all: the common
// single-context case); only when the stream CHANGES is an event recorded on the previous stream -- now, i.e. behind
// everything it has queued since: conservative -- and waited for by the new one.  (Another PROCESS on the same device
// is not covered: its symptom is the bounded-wait timeout reported by rls_cgnr_get_status.)
static hipEvent_t g_resident_ev[64];
static resident_chain_state g_resident_chain;  // guarded by rls_capture_mutex(); cleared by rls_resident_forget (context teardown)
void rls_resident_forget(int device, hipStream_t stream) {
  resident_chain_forget(rls_capture_mutex(), g_resident_chain, device, (void*)stream);
}
// `clean` (nullable): the plan's init kernel has zeroed the counters itself and nothing has used them since -- the memset
// (a launch of its own: ~4 us on the stream between init! and the resident kernel of every solve) is skipped, once
template <typename F>
static int32_t resident_chain(rls_ctx* ctx, void* rsync, F&& launch, bool* clean = nullptr) {
  const int d = ctx->device < 64 ? ctx->device : 63;
  return resident_chain_step(
      rls_capture_mutex(), g_resident_chain, ctx->device, (void*)ctx->stream,
      [&](void* prev) -> int32_t {  // the chain changes streams: order this launch behind everything the previous stream has queued
        if (!g_resident_ev[d]) RLS_HIP(ctx, hipEventCreateWithFlags(&g_resident_ev[d], hipEventDisableTiming));
        RLS_HIP(ctx, hipEventRecord(g_resident_ev[d], (hipStream_t)prev));
        RLS_HIP(ctx, hipStreamWaitEvent(ctx->stream, g_resident_ev[d], 0));
        return 0;
      },
      [&]() -> int32_t {
        // arrival counters and the {fail, completed} words of THIS launch; the count of lost launches behind them is sticky
        if (clean && *clean && ctx->tune.resident_preclear) *clean = false;
        else RLS_HIP(ctx, hipMemsetAsync(rsync, 0, rls_resident_sync_clear_bytes(), ctx->stream));
        if (!ctx->tune.resident_l2_rows)  // measurement switch: pretend a workgroup is misplaced -- partial rows are written through
          RLS_HIP(ctx, hipMemsetAsync((char*)rsync + rls_resident_sync_placement_offset(), 1, 1, ctx->stream));
        return launch();
      });
}
static int32_t resident_chain_launch(rls_ctx* ctx, rls_cgnr* s, const rls_cgnr_pipe& P, int n_steps) {
  return resident_chain(ctx, s->rsync, [&]() {
    return rls_cgnr_resident_launch(ctx, s->op->dtype, P, s->rdots, s->rsync, n_steps, (unsigned)ctx->tune.resident_spin);
  }, &s->rsync_clean);
}
// the sync block of a plan (zeroed once: the sticky word starts at 0) and the pinned mirror of its three flag words
static hipError_t resident_alloc(rls_ctx* ctx, const rls_operator* op, void** rsync, unsigned** rsync_h) {
  hipError_t e = dmalloc(rsync, rls_resident_sync_alloc_bytes(op->dtype, op->N));
  if (e == hipSuccess) e = hipMemsetAsync(*rsync, 0, rls_cgnr_resident_sync_bytes(), ctx->stream);
  if (e == hipSuccess && rsync_h && !*rsync_h) {
    e = hmalloc(rsync_h, 4 * sizeof(unsigned));
    if (e == hipSuccess) memset(*rsync_h, 0, 4 * sizeof(unsigned));
  }
  return e;
}
// enqueue the read-back of {fail, completed, failed}; the caller synchronises (normally with its scalar read-back)
static int32_t resident_fetch_flags(rls_ctx* ctx, const void* rsync, unsigned* rsync_h) {
  return rls_fetch_add(ctx, (const char*)rsync + rls_resident_sync_flags_offset(), rsync_h, 3 * sizeof(unsigned));
}
// Launches lost since the last call (0 = none).  A lost launch changed nothing (x, r, p and the scalars are written back by
// workgroup 0 only after its last barrier), so the caller re-runs the missing iterations on the per-iteration pipeline.
// The plan stays off the resident kernels from here on, and a context that has lost two launches stops using them at all:
// whatever keeps the grid from being resident (another process on the device, a long kernel on another stream) would
// cost every later attempt its full wait bound.
static unsigned resident_lost(rls_ctx* ctx, void* rsync, unsigned* rsync_h, bool* off, int* fallbacks) {
  const unsigned lost = rsync_h[2];
  if (!lost) return 0;
  (void)hipMemsetAsync((char*)rsync + rls_resident_sync_flags_offset() + 2 * sizeof(unsigned), 0, sizeof(unsigned), ctx->stream);
  rsync_h[2] = 0;
  *off = true;
  *fallbacks += (int)lost;
  if (++ctx->resident_failures >= 2) ctx->tune.resident = 0;
  return lost;
}

static rls_cgnr_pipe cgnr_pipe_desc(const rls_cgnr* s) {
  rls_cgnr_pipe P;
  P.A = s->op->A;
  P.lda = s->op->lda;
  P.M = s->op->M;
  P.N = s->op->N;
  P.x = s->x;
  P.r0 = s->r;
  P.p0 = s->p;
  P.r1 = s->r1;
  P.p1 = s->p1;
  P.v = s->v;
  P.slab = s->slab_b ? s->slab_b : s->op->slab;
  P.dots = s->dots;
  P.ttw = s->ttw;
  P.ndots = (int)((s->op->N + 15) / 16);
  P.sc = s->sc;
  P.scn = s->scn;
  P.nrhs = s->nrhs;
  P.vstride = s->ldv;
  return P;
}

constexpr int UPD_THREADS = 1024;

// after r = A^H b:  x = 0, v = 0, p = r, z0 = ||r||, scalars reset          (src/CGNR.jl:108-126)
template <typename E>
__global__ __launch_bounds__(UPD_THREADS) void cgnr_init_kernel(E* __restrict__ x, const E* __restrict__ r,
                                                                E* __restrict__ p, E* __restrict__ v, int64_t n,
                                                                cgnr_scalars* sc, float lambda, float rel_tol,
                                                                int max_iter, unsigned* clear_words = nullptr,
                                                                int n_clear = 0) {
  __shared__ double sm[16];
  // the arrival counters of the plan's resident kernel (resident_chain: saves the memset launch ahead of it)
  for (int i = threadIdx.x; i < n_clear; i += UPD_THREADS) clear_words[i] = 0u;
  double rr = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += UPD_THREADS) {
    const E ri = r[i];
    x[i] = elem<E>::zero();
    v[i] = elem<E>::zero();
    p[i] = ri;
    rr += (double)elem<E>::re(ri) * (double)elem<E>::re(ri) + (double)elem<E>::im(ri) * (double)elem<E>::im(ri);
  }
  rr = block_sum_n<UPD_THREADS / 64>(rr, sm);  // (constant wave count: no dispatch-packet read)
  if (threadIdx.x == 0) {
    sc->rr = rr;
    sc->z0 = sqrt(rr);
    sc->zeta = 0.0;
    sc->alpha_re = sc->alpha_im = sc->beta_re = sc->beta_im = 0.0;
    sc->lambda = lambda;
    sc->rel_tol = rel_tol;
    sc->iteration = 0;
    sc->max_iter = max_iter;
    sc->pending = 0;
    sc->cur = 0;
    sc->fresh = 0;
    // done() evaluated before the first iteration: ||r||/z0 <= relTol || 0 >= min(iterations, N)
    const float ratio = (float)(sqrt(rr) / sqrt(rr));  // NaN when r == 0, as in the reference
    sc->done = (ratio <= rel_tol) || (0 >= max_iter);
  }
}

// the BLAS-1 part of one CGNR iteration, src/CGNR.jl:153-176, after v = AHA p
template <typename E>
__global__ __launch_bounds__(UPD_THREADS) void cgnr_update_kernel(E* __restrict__ x, E* __restrict__ r,
                                                                  E* __restrict__ p, const E* __restrict__ v,
                                                                  int64_t n, cgnr_scalars* sc) {
  if (sc->done) return;
  __shared__ double sm[16];
  const float lambda = sc->lambda;
  const double zeta = sc->rr;  // zeta = ||r||^2                                        :153
  // normvl = <p, v> (conjugating, complex-typed) and ||p||^2                           :154,158
  double nre = 0.0, nim = 0.0, pp = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += UPD_THREADS) {
    const E pi = p[i], vi = v[i];
    nre += (double)elem<E>::re(pi) * (double)elem<E>::re(vi) + (double)elem<E>::im(pi) * (double)elem<E>::im(vi);
    if constexpr (elem<E>::cplx)
      nim += (double)elem<E>::re(pi) * (double)elem<E>::im(vi) - (double)elem<E>::im(pi) * (double)elem<E>::re(vi);
    if (lambda > 0.f)
      pp += (double)elem<E>::re(pi) * (double)elem<E>::re(pi) + (double)elem<E>::im(pi) * (double)elem<E>::im(pi);
  }
  nre = block_sum(nre, sm);
  if constexpr (elem<E>::cplx) nim = block_sum(nim, sm);
  if (lambda > 0.f) pp = block_sum(pp, sm);
  // alpha = zeta / (normvl + lambda ||p||^2)                                            :156-161
  dcomplex den = {nre + (lambda > 0.f ? (double)lambda * pp : 0.0), nim};
  dcomplex alpha = dc_div({zeta, 0.0}, den);
  const E a = elem<E>::make((float)alpha.re, (float)alpha.im);
  const E na = elem<E>::make(-(float)alpha.re, -(float)alpha.im);
  // x += alpha p ; r -= alpha v ; r -= lambda alpha p                                   :163-169
  double rr = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += UPD_THREADS) {
    const E pi = p[i];
    x[i] = elem<E>::fma(pi, a, x[i]);
    E ri = elem<E>::fma(v[i], na, r[i]);
    if (lambda > 0.f) ri = elem<E>::fma(elem<E>::scale(-lambda, pi), a, ri);
    r[i] = ri;
    rr += (double)elem<E>::re(ri) * (double)elem<E>::re(ri) + (double)elem<E>::im(ri) * (double)elem<E>::im(ri);
  }
  rr = block_sum(rr, sm);
  // beta = <r, r> / zeta ; p = beta p + r                                               :171-174
  const double beta = rr / zeta;
  const float bf = (float)beta;
  for (int64_t i = threadIdx.x; i < n; i += UPD_THREADS) p[i] = elem<E>::add(elem<E>::scale(bf, p[i]), r[i]);
  if (threadIdx.x == 0) {
    sc->zeta = zeta;
    sc->rr = rr;
    sc->alpha_re = alpha.re;
    sc->alpha_im = alpha.im;
    sc->beta_re = beta;
    sc->beta_im = 0.0;
    const int it = sc->iteration + 1;
    sc->iteration = it;
    const float ratio = (float)(sqrt(rr) / sc->z0);
    sc->done = (ratio <= sc->rel_tol) || (it >= sc->max_iter);  // :181-185
  }
}

// The same update with the four vectors held in registers (n <= EPT * UPD_THREADS): every load is requested before
// anything is waited for -- one memory round trip instead of the seven dependent ones of the loop form above
// (scalars, three passes over the vectors) -- and the same per-thread summation order, so the same bits.
template <typename E, int EPT>
__global__ __launch_bounds__(UPD_THREADS) void cgnr_update_reg_kernel(E* __restrict__ x, E* __restrict__ r,
                                                                      E* __restrict__ p, const E* __restrict__ v,
                                                                      int64_t n, cgnr_scalars* sc) {
  __shared__ double sm[48];
  const cgnr_scalars S = *sc;
  E pv[EPT], vv[EPT], xv[EPT], rv[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = threadIdx.x + (int64_t)e * UPD_THREADS;
    const int64_t ic = i < n ? i : n - 1;
    pv[e] = p[ic];
    vv[e] = v[ic];
    xv[e] = x[ic];
    rv[e] = r[ic];
  }
  if (S.done) return;
  const float lambda = S.lambda;
  const double zeta = S.rr;
  double nre = 0.0, nim = 0.0, pp = 0.0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = threadIdx.x + (int64_t)e * UPD_THREADS;
    if (i >= n) pv[e] = vv[e] = rv[e] = elem<E>::zero();
    nre += (double)elem<E>::re(pv[e]) * (double)elem<E>::re(vv[e]) + (double)elem<E>::im(pv[e]) * (double)elem<E>::im(vv[e]);
    if constexpr (elem<E>::cplx)
      nim += (double)elem<E>::re(pv[e]) * (double)elem<E>::im(vv[e]) - (double)elem<E>::im(pv[e]) * (double)elem<E>::re(vv[e]);
    pp += (double)elem<E>::re(pv[e]) * (double)elem<E>::re(pv[e]) + (double)elem<E>::im(pv[e]) * (double)elem<E>::im(pv[e]);
  }
  block_sum3(nre, nim, pp, sm);
  const dcomplex den = {nre + (lambda > 0.f ? (double)lambda * pp : 0.0), nim};
  const dcomplex alpha = dc_div({zeta, 0.0}, den);
  const E a = elem<E>::make((float)alpha.re, (float)alpha.im);
  const E na = elem<E>::make(-(float)alpha.re, -(float)alpha.im);
  double rr = 0.0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    xv[e] = elem<E>::fma(pv[e], a, xv[e]);
    E ri = elem<E>::fma(vv[e], na, rv[e]);
    if (lambda > 0.f) ri = elem<E>::fma(elem<E>::scale(-lambda, pv[e]), a, ri);
    rv[e] = ri;
    rr += (double)elem<E>::re(ri) * (double)elem<E>::re(ri) + (double)elem<E>::im(ri) * (double)elem<E>::im(ri);
  }
  rr = block_sum(rr, sm);
  const double beta = rr / zeta;
  const float bf = (float)beta;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = threadIdx.x + (int64_t)e * UPD_THREADS;
    if (i < n) {
      x[i] = xv[e];
      r[i] = rv[e];
      p[i] = elem<E>::add(elem<E>::scale(bf, pv[e]), rv[e]);
    }
  }
  if (threadIdx.x == 0) {
    sc->zeta = zeta;
    sc->rr = rr;
    sc->alpha_re = alpha.re;
    sc->alpha_im = alpha.im;
    sc->beta_re = beta;
    sc->beta_im = 0.0;
    const int it = S.iteration + 1;
    sc->iteration = it;
    const float ratio = (float)(sqrt(rr) / S.z0);
    sc->done = (ratio <= S.rel_tol) || (it >= S.max_iter);  // :181-185
  }
}

template <typename E>
static void cgnr_launch_init(rls_cgnr* s, float lambda, float rel_tol, int max_iter) {
  rls_operator* op = s->op;
  const bool clr = s->rsync && s->nrhs == 1;
  hipLaunchKernelGGL(cgnr_init_kernel<E>, dim3(1), dim3(UPD_THREADS), 0, op->ctx->stream, (E*)s->x, (const E*)s->r,
                     (E*)s->p, (E*)s->v, op->N, s->sc, lambda, rel_tol, max_iter, clr ? (unsigned*)s->rsync : nullptr,
                     clr ? (int)(rls_resident_sync_clear_bytes() / sizeof(unsigned)) : 0);
  s->rsync_clean = clr;
}
template <typename E>
static void cgnr_launch_update(rls_cgnr* s) {
  rls_operator* op = s->op;
  const int64_t n = op->N;
#define RLS_UPD_REG(EE)                                                                                             \
  hipLaunchKernelGGL((cgnr_update_reg_kernel<E, EE>), dim3(1), dim3(UPD_THREADS), 0, op->ctx->stream, (E*)s->x, (E*)s->r, \
                     (E*)s->p, (const E*)s->v, n, s->sc)
  if (n <= UPD_THREADS) RLS_UPD_REG(1);
  else if (n <= 2 * UPD_THREADS) RLS_UPD_REG(2);
  else if (n <= 4 * UPD_THREADS) RLS_UPD_REG(4);
  else if (n <= 8 * UPD_THREADS) RLS_UPD_REG(8);
  else
    hipLaunchKernelGGL(cgnr_update_kernel<E>, dim3(1), dim3(UPD_THREADS), 0, op->ctx->stream, (E*)s->x, (E*)s->r,
                       (E*)s->p, (const E*)s->v, n, s->sc);
#undef RLS_UPD_REG
}

static int32_t launch_status(rls_ctx* ctx) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

static int32_t cgnr_enqueue_update(rls_cgnr* s) {
  if (s->op->dtype == RLS_F32)
    cgnr_launch_update<float>(s);
  else
    cgnr_launch_update<float2>(s);
  return launch_status(s->op->ctx);
}

static int32_t cgnr_enqueue_iteration(rls_cgnr* s) {
  RLS_TRY(op_normal(s->op, s->p, s->v, &s->sc->done));
  return cgnr_enqueue_update(s);
}

static int32_t cgnr_effective_iterations(rls_cgnr* s, int32_t iterations) {
  // iteration >= min(solver.iterations, size(AHA, 2))    src/CGNR.jl:185
  int64_t m = iterations < s->op->N ? iterations : s->op->N;
  return (int32_t)(m < 0 ? 0 : m);
}

// ---------------------------------------------------------------------------------------------
// FISTA
// ---------------------------------------------------------------------------------------------
struct rls_fista {
  rls_operator* op;
  rls_ctx* actx;
  uint64_t actx_id = 0;
  int device;
  void* buf[2];  // x / xold, swapped by iteration parity: state.x == buf[iteration & 1]
  void *x0, *res;
  void* y;       // extrapolated point (plan-owned), the GEMV input
  void *y1, *res_raw;  // fused pipeline: second extrapolated-point buffer, AHA y before "- x0"
  fista_scalars* scn;  // staged scalars (slab pipeline) / second parity (Gram pipeline)
  bool use_pipe;
  void* res_raw1;      // Gram pipeline: second parity of AHA y
  bool use_gram;
  fista_scalars* sc;
  fista_scalars* sc_h;
  step_graph graph;
  int32_t reg_kind, proj_kind;
  float lambda;
  int64_t l21_slices;
  bool initialised;
  // batched plan (rls_fista_create_batched): nrhs columns ldv elements apart, the K extrapolated points as an
  // MFMA operand panel, T = A Y and the partial rows of A^H T (skinny.hip)
  int nrhs = 1;
  int64_t ldv = 0;
  float *Ypack = nullptr, *Tpack = nullptr;
  void* Vpart = nullptr;
  int splits = 1;
  int half = 0;  // operand-panel layout, fixed at creation (rls_skinny_half)
  fista_scalars* scb_h = nullptr;  // pinned [nrhs]
  int enq = 0;           // iterations enqueued since init (== the device's count unless the plan stopped early)
  int graph_parity = 0;  // parity of `enq` the cached graph's buffer hints were captured with
  float theta0 = 1.f;    // theta given to the last init (rls_fista_set_start needs it)
  // resident mode (normal.hip, fista_resident_kernel)
  void* rsync = nullptr;
  unsigned* rsync_h = nullptr;
  bool resident_used = false;
  bool small = false;         // dense A that fits ONE CU's registers: whole step calls on fista_small_kernel (small.hip)
  srv_state srv;              // server mode of the resident kernel (rls_fista_step_status)
  rls_mailbox_slot mb_arm;    // as the cgnr plan's
  bool mb_sent = false;
  bool resident_off = false;  // a resident launch was lost: the plan stays on the per-iteration pipeline (cgnr plan, above)
  int fallbacks = 0;
  long long requested = 0;    // iterations asked for since init
  bool rsync_clean = false;   // the init kernel has just zeroed the arrival counters (resident_chain)
  // batched plan on an explicit Gram matrix, <= 8 ComplexF32 columns, AHA in the register files (gramk.hip): exchange scratch
  bool fgramk = false;
  int restart_b = 0;          // gradient restart asked for at init (the resident batched kernel does not carry it)
  float* fk_yx = nullptr;
  void* fk_xx = nullptr;
  double* fk_dots = nullptr;
  // TV regulariser (rls_fista_set_reg_tv): the FGP launch sits BETWEEN the two halves of the update (src/FISTA.jl:164), reading the
  // gradient step from tv_in and leaving prox_TV of it in tv_out (plan scratch: no pointer depends on the iteration's parity)
  int tv_ndims = 0, tv_ntv = 0, tv_iters = 10;
  int64_t tv_shape[4] = {1, 1, 1, 1};
  int32_t tv_dims[4] = {0, 0, 0, 0};
  void *tv_in = nullptr, *tv_out = nullptr;
  float rho_h = 0.f;  // rho of the last init (the prox threshold rho * lambda is a launch argument of the FGP kernel)
};

// batched launches: workgroup b = column b.  Vpart non-null: AHA y arrives as `S` partial rows per column and is
// summed here (fixed order); Yp non-null: the extrapolated point also goes into the operand panel Yp[g][n][j].
template <typename E>
struct fista_batch {
  int64_t ldv;
  const E* Vpart;
  int S, nrhs_pad;
  E* Yp;
  int half;  // operand-panel layout (rls_common.hpp, panel_col)
};
template <typename E>
__device__ static inline E fista_parts(const fista_batch<E>& B, int b, int64_t N, int64_t i) {
  E v = B.Vpart[(int64_t)b * N + i];
  for (int s = 1; s < B.S; ++s) v = elem<E>::add(v, B.Vpart[((int64_t)s * B.nrhs_pad + b) * N + i]);
  return v;
}

// x0 = A^H b is already in place.  x = x_init (zero), xold = 0, res = Inf, y = x   (src/FISTA.jl:110-129)
template <typename E>
__global__ __launch_bounds__(UPD_THREADS) void fista_init_kernel(E* __restrict__ b0, E* __restrict__ b1,
                                                                 E* __restrict__ x0, E* __restrict__ res,
                                                                 E* __restrict__ y, int64_t n, fista_scalars* sc,
                                                                 float rho, float theta, float rel_tol, int max_iter,
                                                                 int restart, int reg_kind, int proj_kind,
                                                                 float lambda, long long slices, fista_batch<E> Bt,
                                                                 unsigned* clear_words = nullptr, int n_clear = 0) {
  __shared__ double sm[16];
  for (int i = threadIdx.x; i < n_clear; i += UPD_THREADS) clear_words[i] = 0u;  // (cgnr_init_kernel)
  const int b = blockIdx.x;
  b0 += b * Bt.ldv;
  b1 += b * Bt.ldv;
  x0 += b * Bt.ldv;
  res += b * Bt.ldv;
  y += b * Bt.ldv;
  sc += b;
  const panel_col<E> yp = panel_column<E>(Bt.Yp, n, b, Bt.half);
  double nn = 0.0;
  const float inf = __builtin_huge_valf();
  for (int64_t i = threadIdx.x; i < n; i += UPD_THREADS) {
    E v;
    if (Bt.Vpart) {  // x0 = A^H b from the partial rows of the skinny product
      v = fista_parts<E>(Bt, b, n, i);
      x0[i] = v;
    } else {
      v = x0[i];
    }
    nn += (double)elem<E>::re(v) * (double)elem<E>::re(v) + (double)elem<E>::im(v) * (double)elem<E>::im(v);
    b0[i] = elem<E>::zero();
    b1[i] = elem<E>::zero();
    y[i] = elem<E>::zero();
    if (Bt.Yp) yp.put(i, elem<E>::zero());
    res[i] = elem<E>::make(inf, 0.f);
  }
  nn = block_sum_n<UPD_THREADS / 64>(nn, sm);
  if (threadIdx.x == 0) {
    sc->norm_x0 = sqrt(nn);
    sc->res_norm = (double)inf;
    sc->rel_res_norm = (double)inf;
    sc->rho = rho;
    sc->theta = theta;
    sc->theta_old = theta;
    sc->rel_tol = rel_tol;
    sc->lambda = lambda;
    sc->iteration = 0;
    sc->max_iter = max_iter;
    sc->done = (0 >= max_iter);
    sc->restart = restart;
    sc->reg_kind = reg_kind;
    sc->proj_kind = proj_kind;
    sc->l21_slices = slices;
    sc->pending = 0;
    sc->ycur = 0;
    sc->fresh = 0;
  }
}

// everything of src/FISTA.jl:153-180 after res = AHA y, plus the NEXT iteration's momentum step
// (:144-148) so that one iteration is GEMV, GEMV, this kernel.
// `phase` splits it around a prox that is a launch of its own (TV: the FGP kernel): 0 = everything; 1 = res, the gradient step
// into tv_in (not into x) and the residual norm; 2 = x = proj(tv_out), then the restart test, theta, `done` and the next y.
template <typename E>
__global__ __launch_bounds__(UPD_THREADS) void fista_update_kernel(E* __restrict__ b0, E* __restrict__ b1,
                                                                   const E* __restrict__ x0, E* __restrict__ res,
                                                                   E* __restrict__ y, int64_t n, fista_scalars* sc,
                                                                   fista_batch<E> Bt, int phase = 0, E* __restrict__ tv_in = nullptr,
                                                                   const E* __restrict__ tv_out = nullptr) {
  const int b = blockIdx.x;
  b0 += b * Bt.ldv;
  b1 += b * Bt.ldv;
  x0 += b * Bt.ldv;
  res += b * Bt.ldv;
  y += b * Bt.ldv;
  sc += b;
  const panel_col<E> yp = panel_column<E>(Bt.Yp, n, b, Bt.half);
  if (sc->done) return;  // a retired column keeps its panel entry (src/MultiThreading.jl:60-78)
  __shared__ double sm[16];
  const int it = sc->iteration;
  E* xnew = (it & 1) ? b0 : b1;  // after the reference's pointer swap: state.x      :144-146
  E* xold = (it & 1) ? b1 : b0;  // holds x_k
  const float rho = sc->rho;
  const int reg_kind = sc->reg_kind, proj_kind = sc->proj_kind;
  const float thr = rho * sc->lambda;  // prox!(reg, x, rho * lambda(reg))             :164
  double rn = 0.0;
  if (phase == 2) {  // the prox ran as a launch of its own: x = proj(prox), the residual norm is phase 1's
    for (int64_t i = threadIdx.x; i < n; i += UPD_THREADS) xnew[i] = fista_proj_elem<E>(tv_out[i], proj_kind);
    __syncthreads();
  } else {
  for (int64_t i = threadIdx.x; i < n; i += UPD_THREADS) {
    const E ri = elem<E>::sub(Bt.Vpart ? fista_parts<E>(Bt, b, n, i) : res[i], x0[i]);  // res .-= x0      :153
    res[i] = ri;
    rn += (double)elem<E>::re(ri) * (double)elem<E>::re(ri) + (double)elem<E>::im(ri) * (double)elem<E>::im(ri);
    E xi = elem<E>::sub(y[i], elem<E>::scale(rho, ri));             // x .-= rho .* res :154
    if (phase == 1) {
      tv_in[i] = xi;
      continue;
    }
    if (reg_kind != RLS_REG_L21) xi = fista_proj_elem<E>(fista_prox_elem<E>(xi, reg_kind, thr), proj_kind);
    xnew[i] = xi;
  }
  }
  if (reg_kind == RLS_REG_L21) {  // group soft-threshold needs the whole new x         ProxL21.jl:30-35
    __syncthreads();
    const int64_t slen = n / sc->l21_slices;
    for (int64_t g = threadIdx.x; g < slen; g += UPD_THREADS) {
      float s2 = 0.f;
      for (int64_t k = g; k < n; k += slen) s2 += elem<E>::abs2(xnew[k]);
      const float gn = sqrtf(s2);
      const float q = (gn - thr) / gn;
      const float fac = (q != q) ? q : fmaxf(q, 0.f);
      for (int64_t k = g; k < n; k += slen) xnew[k] = fista_proj_elem<E>(elem<E>::scale(fac, xnew[k]), proj_kind);
    }
    __syncthreads();
  }
  rn = block_sum(rn, sm);
  if (phase == 1) {
    if (threadIdx.x == 0) sc->res_norm = sqrt(rn);
    return;
  }
  float theta = sc->theta;
  if (sc->restart) {  // real(res . (x - xold)) > 0  => theta = 1                       :171-176
    double d = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += UPD_THREADS) {
      const E df = elem<E>::sub(xnew[i], xold[i]);
      const E ri = res[i];
      d += (double)elem<E>::re(ri) * (double)elem<E>::re(df) + (double)elem<E>::im(ri) * (double)elem<E>::im(df);
    }
    d = block_sum(d, sm);
    if (d > 0.0) theta = 1.f;
  }
  const float theta_old = theta;                                     // :179
  theta = (1.f + sqrtf(1.f + 4.f * theta_old * theta_old)) / 2.f;    // :180
  const double res_norm = phase == 2 ? sc->res_norm : sqrt(rn);
  const float rel = (float)(res_norm / sc->norm_x0);                 // :156
  const int done = (rel < sc->rel_tol) || (it + 1 >= sc->max_iter);  // :187-189
  if (!done) {
    // next iteration's Nesterov step, formed out of place in y                         :147-148
    const float c1 = (1.f - theta_old) / theta;
    const float c2 = (theta_old - 1.f) / theta + 1.f;
    for (int64_t i = threadIdx.x; i < n; i += UPD_THREADS) {
      const E yi = elem<E>::add(elem<E>::scale(c1, xold[i]), elem<E>::scale(c2, xnew[i]));
      y[i] = yi;
      if (Bt.Yp) yp.put(i, yi);
    }
  }
  if (threadIdx.x == 0) {
    sc->res_norm = res_norm;
    sc->rel_res_norm = (double)rel;
    sc->theta = theta;
    sc->theta_old = theta_old;
    sc->iteration = it + 1;
    sc->done = done;
  }
}

// fista_update_kernel with the vectors in registers (n <= EPT * UPD_THREADS; every regulariser but L21, whose
// group norms need the whole new x): all loads requested up front, the same per-thread summation order.
template <typename E, int EPT>
__global__ __launch_bounds__(UPD_THREADS) void fista_update_reg_kernel(E* __restrict__ b0, E* __restrict__ b1,
                                                                       const E* __restrict__ x0, E* __restrict__ res,
                                                                       E* __restrict__ y, int64_t n, fista_scalars* sc,
                                                                       fista_batch<E> Bt) {
  const int b = blockIdx.x;
  b0 += b * Bt.ldv;
  b1 += b * Bt.ldv;
  x0 += b * Bt.ldv;
  res += b * Bt.ldv;
  y += b * Bt.ldv;
  sc += b;
  const panel_col<E> yp = panel_column<E>(Bt.Yp, n, b, Bt.half);
  __shared__ double sm[16];
  const int done0 = sc->done, it = sc->iteration, reg_kind = sc->reg_kind, proj_kind = sc->proj_kind;
  const int restart = sc->restart, max_iter = sc->max_iter;
  const float rho = sc->rho, lam = sc->lambda, rel_tol = sc->rel_tol, theta0 = sc->theta;
  const double norm_x0 = sc->norm_x0;
  E* xnew = (it & 1) ? b0 : b1;  // after the reference's pointer swap: state.x      :144-146
  E* xold = (it & 1) ? b1 : b0;  // holds x_k
  E raw[EPT], x0v[EPT], yv[EPT], xo[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = threadIdx.x + (int64_t)e * UPD_THREADS;
    const int64_t ic = i < n ? i : n - 1;
    raw[e] = Bt.Vpart ? fista_parts<E>(Bt, b, n, ic) : res[ic];
    x0v[e] = x0[ic];
    yv[e] = y[ic];
    xo[e] = xold[ic];
  }
  if (done0) return;  // a retired column keeps its panel entry (src/MultiThreading.jl:60-78)
  const float thr = rho * lam;  // prox!(reg, x, rho * lambda(reg))             :164
  E ri[EPT], xn[EPT];
  double rn = 0.0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = threadIdx.x + (int64_t)e * UPD_THREADS;
    ri[e] = elem<E>::sub(raw[e], x0v[e]);                                      // res .-= x0      :153
    E xi = elem<E>::sub(yv[e], elem<E>::scale(rho, ri[e]));                    // x .-= rho .* res :154
    xn[e] = fista_proj_elem<E>(fista_prox_elem<E>(xi, reg_kind, thr), proj_kind);
    if (i < n) rn += (double)elem<E>::re(ri[e]) * (double)elem<E>::re(ri[e]) + (double)elem<E>::im(ri[e]) * (double)elem<E>::im(ri[e]);
  }
  rn = block_sum(rn, sm);
  float theta = theta0;
  if (restart) {  // real(res . (x - xold)) > 0  => theta = 1                       :171-176
    double d = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t i = threadIdx.x + (int64_t)e * UPD_THREADS;
      const E df = elem<E>::sub(xn[e], xo[e]);
      if (i < n) d += (double)elem<E>::re(ri[e]) * (double)elem<E>::re(df) + (double)elem<E>::im(ri[e]) * (double)elem<E>::im(df);
    }
    d = block_sum(d, sm);
    if (d > 0.0) theta = 1.f;
  }
  const float theta_old = theta;                                     // :179
  theta = (1.f + sqrtf(1.f + 4.f * theta_old * theta_old)) / 2.f;    // :180
  const double res_norm = sqrt(rn);
  const float rel = (float)(res_norm / norm_x0);                     // :156
  const int done = (rel < rel_tol) || (it + 1 >= max_iter);          // :187-189
  const float c1 = (1.f - theta_old) / theta;
  const float c2 = (theta_old - 1.f) / theta + 1.f;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = threadIdx.x + (int64_t)e * UPD_THREADS;
    if (i < n) {
      xnew[i] = xn[e];
      res[i] = ri[e];
      if (!done) {  // next iteration's Nesterov step, formed out of place in y                         :147-148
        const E yi = elem<E>::add(elem<E>::scale(c1, xo[e]), elem<E>::scale(c2, xn[e]));
        y[i] = yi;
        if (Bt.Yp) yp.put(i, yi);
      }
    }
  }
  if (threadIdx.x == 0) {
    sc->res_norm = res_norm;
    sc->rel_res_norm = (double)rel;
    sc->theta = theta;
    sc->theta_old = theta_old;
    sc->iteration = it + 1;
    sc->done = done;
  }
}

// launch of the update half of an unfused / batched FISTA iteration: the register form when it applies
template <typename E>
static int32_t fista_launch_update(rls_fista* s, unsigned nblocks, const fista_batch<E>& Bt) {
  rls_operator* op = s->op;
  const int64_t n = op->N;
  int32_t tv_status = 0;
#define RLS_FUPD_REG(EE)                                                                                            \
  hipLaunchKernelGGL((fista_update_reg_kernel<E, EE>), dim3(nblocks), dim3(UPD_THREADS), 0, op->ctx->stream,        \
                     (E*)s->buf[0], (E*)s->buf[1], (const E*)s->x0, (E*)s->res, (E*)s->y, n, s->sc, Bt)
  if (s->reg_kind == RLS_REG_TV) {  // gradient step | FGP launch (its own workgroup, skipped once `done`) | projection, theta, y
    hipLaunchKernelGGL(fista_update_kernel<E>, dim3(1), dim3(UPD_THREADS), 0, op->ctx->stream, (E*)s->buf[0], (E*)s->buf[1],
                       (const E*)s->x0, (E*)s->res, (E*)s->y, n, s->sc, Bt, 1, (E*)s->tv_in, (const E*)s->tv_out);
    // (the threshold rho * lambda, the geometry and iterationsTV are kernel ARGUMENTS: a cached graph of this sequence is dropped
    //  whenever one of them changes -- fista_drop_graph in rls_fista_set_reg / _set_reg_tv / fista_init_finish)
    tv_status = rls_tv_single_launch(op->ctx, op->dtype, s->tv_ndims, s->tv_shape, s->tv_ntv, s->tv_dims, s->tv_in, nullptr, s->tv_out,
                                     s->rho_h * s->lambda, s->tv_iters, &s->sc->done, 1, 0, 0);
    hipLaunchKernelGGL(fista_update_kernel<E>, dim3(1), dim3(UPD_THREADS), 0, op->ctx->stream, (E*)s->buf[0], (E*)s->buf[1],
                       (const E*)s->x0, (E*)s->res, (E*)s->y, n, s->sc, Bt, 2, (E*)s->tv_in, (const E*)s->tv_out);
  } else if (s->reg_kind != RLS_REG_L21 && n <= 4 * UPD_THREADS) {
    if (n <= UPD_THREADS) RLS_FUPD_REG(1);
    else if (n <= 2 * UPD_THREADS) RLS_FUPD_REG(2);
    else RLS_FUPD_REG(4);
  } else {
    hipLaunchKernelGGL(fista_update_kernel<E>, dim3(nblocks), dim3(UPD_THREADS), 0, op->ctx->stream, (E*)s->buf[0],
                       (E*)s->buf[1], (const E*)s->x0, (E*)s->res, (E*)s->y, n, s->sc, Bt);
  }
#undef RLS_FUPD_REG
  return tv_status;  // (a refused FGP launch would leave phase 2 reading a stale tv_out: the step reports it)
}

static bool fista_pipe_ok(const rls_fista* s) {
  const rls_ctx* ctx = s->op->ctx;
  return s->y1 && s->op->slab && !s->op->G && ctx->tune.fused_normal && ctx->tune.cgnr_pipeline &&
         (s->reg_kind == RLS_REG_NONE || s->reg_kind == RLS_REG_L1 || s->reg_kind == RLS_REG_L2);
}

static bool fista_gram_ok(const rls_fista* s) {
  const rls_ctx* ctx = s->op->ctx;
  return s->res_raw1 && s->op->G && ctx->tune.gram_pipeline &&
         (s->reg_kind == RLS_REG_NONE || s->reg_kind == RLS_REG_L1 || s->reg_kind == RLS_REG_L2);
}

static rls_fista_gram fista_gram_desc(const rls_fista* s) {
  rls_fista_gram P;
  P.G = s->op->G;
  P.ldg = s->op->ldg;
  P.N = s->op->N;
  P.b0 = s->buf[0];
  P.b1 = s->buf[1];
  P.x0 = s->x0;
  P.res = s->res;
  P.y0 = s->y;
  P.y1 = s->y1;
  P.rr[0] = s->res_raw;
  P.rr[1] = s->res_raw1;
  P.sc[0] = s->sc;
  P.sc[1] = s->scn;
  return P;
}

static rls_fista_pipe fista_pipe_desc(const rls_fista* s) {
  rls_fista_pipe P;
  P.A = s->op->A;
  P.lda = s->op->lda;
  P.M = s->op->M;
  P.N = s->op->N;
  P.b0 = s->buf[0];
  P.b1 = s->buf[1];
  P.x0 = s->x0;
  P.res = s->res;
  P.y0 = s->y;
  P.y1 = s->y1;
  P.res_raw = s->res_raw;
  P.slab = s->op->slab;
  P.sc = s->sc;
  P.scn = s->scn;
  return P;
}

static int32_t fista_enqueue_iteration(rls_fista* s) {
  rls_operator* op = s->op;
  RLS_TRY(op_normal(op, s->y, s->res, &s->sc->done));
  if (op->dtype == RLS_F32)
    RLS_TRY(fista_launch_update<float>(s, 1, fista_batch<float>{0, nullptr, 1, 0, nullptr, 0}));
  else
    RLS_TRY(fista_launch_update<float2>(s, 1, fista_batch<float2>{0, nullptr, 1, 0, nullptr, 0}));
  return launch_status(op->ctx);
}

// ---------------------------------------------------------------------------------------------
// cg!  (IterativeSolvers v0.9 semantics, restated; see oracle/rls_oracle.py::cg_inplace)
// ---------------------------------------------------------------------------------------------
struct cg_scalars {
  double residual, prev, tol;
  float rho, reltol;
  int iteration, maxiter, done, pad;
};

struct rls_cg {
  rls_operator* op;
  rls_ctx* actx;
  uint64_t actx_id = 0;
  int device;
  void *u, *r, *c;
  cg_scalars* sc;
  cg_scalars* sc_h;
  // fused pipeline (normal.hip): cg! on (AHA + rho I) is the CGNR recurrence with lambda = rho and the
  // start residual b - (AHA + rho I) x0, so it reuses the two-launch CGNR pipeline (p = u, v = c)
  void *r1, *p1;
  double* dots;
  cgnr_scalars *psc, *pscn, *psc_h;
  step_graph graph;
  bool used_pipeline;
  // Gram-mode pipeline: second parity of c (= v) and of the partial dots
  void* v1;
  double* gdots;
  // batched plan (rls_cg_create_batched): nrhs columns ldv elements apart, operand panel + partial rows of the
  // skinny matrix-core products (skinny.hip); sc / sc_h hold nrhs structs
  int nrhs = 1;
  int64_t ldv = 0;
  float *Ppack = nullptr, *Tpack = nullptr;
  int half = 0;  // operand-panel layout, fixed at creation (rls_skinny_half)
  void* Vpart = nullptr;
  int splits = 1;
  // resident mode: cg! on (AHA + rho I) IS the CGNR recurrence, so after the start kernel the whole inner solve runs as
  // ONE launch of cgnr_resident_kernel with A in registers (normal.hip)
  void* rsync = nullptr;
  double* rdots = nullptr;
  unsigned* rsync_h = nullptr;  // pinned {fail, completed, failed}
  bool resident_used = false;
  bool gram_resident = false;  // Gram mode with AHA small enough for the register files (rls_gram_resident_ok)
  bool resident_off = false;   // a resident launch was lost: the plan stays on the per-iteration pipeline
  int fallbacks = 0;
  // the last rls_cg_solve, kept so that rls_cg_get_status can repeat it on the pipeline if its resident launch was lost
  // (a lost launch is a no-op: x still holds the warm start)
  struct {
    void* x = nullptr;
    const void* b = nullptr;
    float rho = 0.f, reltol = 0.f;
    int32_t maxiter = 0;
    bool valid = false;
  } last;
};

static bool cg_use_gram_pipeline(const rls_cg* s) {
  return s->gdots && s->op->G && s->op->ctx->tune.gram_pipeline;
}

static rls_gram_pipe cg_gram_desc(const rls_cg* s, void* x) {
  rls_gram_pipe P;
  P.G = s->op->G;
  P.ldg = s->op->ldg;
  P.N = s->op->N;
  P.x = x;
  P.r[0] = s->r;
  P.r[1] = s->r1;
  P.p[0] = s->u;
  P.p[1] = s->p1;
  P.v[0] = s->c;
  P.v[1] = s->v1;
  P.dots = s->gdots;
  P.sc[0] = s->psc;
  P.sc[1] = s->pscn;
  return P;
}

static bool cg_use_pipeline(const rls_cg* s) {
  const rls_ctx* ctx = s->op->ctx;
  return s->r1 && s->op->slab && !s->op->G && ctx->tune.fused_normal && ctx->tune.cgnr_pipeline;
}

static rls_cgnr_pipe cg_pipe_desc(const rls_cg* s, void* x) {
  rls_cgnr_pipe P;
  P.A = s->op->A;
  P.lda = s->op->lda;
  P.M = s->op->M;
  P.N = s->op->N;
  P.x = x;
  P.r0 = s->r;
  P.p0 = s->u;
  P.r1 = s->r1;
  P.p1 = s->p1;
  P.v = s->c;
  P.slab = s->op->slab;
  P.dots = s->dots;
  P.ndots = (int)((s->op->N + 15) / 16);
  P.sc = s->psc;
  P.scn = s->pscn;
  return P;
}

// warm start of the pipeline: c = AHA x is in place; r = b - c - rho x, u = r, scalars reset.
// done at entry mirrors cg!: residual <= tol = reltol * residual (only for reltol >= 1) or maxiter == 0;
// r == 0 exactly is also final (the next alpha would be 0/0).
// ADMM plan (rls_admm_step): the start kernel also forms the right-hand side, b = beta_y + rho (z - u), keeps
// xold = x (src/ADMM.jl:236-243, identity regTrafo), and turns the whole x-update into no-ops once the plan's
// `done` flag is set.  All pointers null for a plain cg! call.
template <typename E>
struct admm_fuse {
  const E *beta_y, *z, *u;
  E *beta, *xold;
  float rho;
  const int* skip;
};
// Batched plans (shared A, rls_cg_create_batched): the per-column kernels below run one workgroup per column
// (blockIdx.x = column, state matrices N x K with leading dimension ldv, one scalar struct per column); the
// operator apply in front of them is the pair of skinny matrix-core products, which leaves AHA u as S partial rows
// per column (summed here in a fixed order) and reads its right operand from the MFMA panel the kernels write.
// All zero = the single-column plans.
template <typename E>
struct col_batch {
  int64_t ldv = 0;
  const E* Vpart = nullptr;
  int S = 1, nrhs_pad = 16;
  E* panel = nullptr;       // operand panel [group][n][16]: column b -> panel + (b >> 4) * n * 16 + (b & 15), stride 16
  int half = 0;             // ... or the (8 re | 8 im) layout of at most 8 complex columns (rls_common.hpp, panel_col)
  int skip_stride = 0;      // ints between the columns' skip flags
  int64_t log_stride = 0;   // floats between the columns' ADMM logs
};
template <typename E>
__device__ static inline E col_parts(const col_batch<E>& B, int b, int64_t N, int64_t i) {
  E v = B.Vpart[(int64_t)b * N + i];
  for (int s = 1; s < B.S; ++s) v = elem<E>::add(v, B.Vpart[((int64_t)s * B.nrhs_pad + b) * N + i]);
  return v;
}
// X (N x K) -> operand panel, ahead of the warm-start apply AHA x of every outer iteration
template <typename E>
__global__ __launch_bounds__(256) void pack_panel_kernel(const E* __restrict__ X, int64_t ldv, E* __restrict__ panel, int64_t n,
                                                         int half) {
  const int b = blockIdx.y;
  const panel_col<E> up = panel_column<E>(panel, n, b, half);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    up.put(i, X[(int64_t)b * ldv + i]);
}
template <typename E>
__device__ static inline E admm_rhs(const admm_fuse<E>& F, const E* b, const E* x, int64_t i) {
  if (!F.beta_y) return b[i];
  E bi = F.beta_y[i];
  bi = elem<E>::add(bi, elem<E>::scale(F.rho, F.z[i]));
  bi = elem<E>::add(bi, elem<E>::scale(-F.rho, F.u[i]));
  F.beta[i] = bi;
  F.xold[i] = x[i];
  return bi;
}

template <typename E>
__global__ __launch_bounds__(UPD_THREADS) void cg_pipe_start_kernel(const E* __restrict__ x, const E* b,
                                                                    E* __restrict__ u, E* __restrict__ r,
                                                                    const E* __restrict__ c, int64_t n,
                                                                    cgnr_scalars* sc, float rho, float reltol,
                                                                    int maxiter, admm_fuse<E> F) {
  __shared__ double sm[16];
  if (F.skip && *F.skip) {
    if (threadIdx.x == 0) {
      sc->iteration = 0;
      sc->max_iter = maxiter;
      sc->pending = 0;
      sc->cur = 0;
      sc->fresh = 0;
      sc->done = 1;
    }
    return;
  }
  double rr = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += UPD_THREADS) {
    const E ci = elem<E>::add(c[i], elem<E>::scale(rho, x[i]));
    const E ri = elem<E>::sub(admm_rhs<E>(F, b, x, i), ci);
    r[i] = ri;
    u[i] = ri;
    rr += (double)elem<E>::re(ri) * (double)elem<E>::re(ri) + (double)elem<E>::im(ri) * (double)elem<E>::im(ri);
  }
  rr = block_sum(rr, sm);
  if (threadIdx.x == 0) {
    sc->rr = rr;
    sc->z0 = sqrt(rr);
    sc->zeta = 0.0;
    sc->alpha_re = sc->alpha_im = sc->beta_re = sc->beta_im = 0.0;
    sc->lambda = rho;
    sc->rel_tol = reltol;
    sc->iteration = 0;
    sc->max_iter = maxiter;
    sc->pending = 0;
    sc->cur = 0;
    sc->fresh = 0;
    sc->done = (maxiter <= 0) || (rr == 0.0) || (1.0f <= reltol);
  }
}

// ADMM bookkeeping for an identity regTrafo, src/ADMM.jl:259-299 in ONE single-workgroup launch:
//   u_new = u + x - z ;  Delta = ||x-xold|| + ||z-zold|| + ||u_new-u|| ;  s = rho ||z-zold|| ;
//   eps_pri = max(||x||, ||z||) ; r = ||x-z|| ; eps_dua = rho ||u_new||
// (with Phi = I the reference's hijacked scratch sequence reduces to exactly these seven norms).
// out[0..5] = Delta, s/rho, eps_pri, r, eps_dua/rho, ||x-xold||   (floats; rho applied on the host)
template <typename E>
__global__ __launch_bounds__(UPD_THREADS) void admm_post_kernel(const E* __restrict__ x, const E* __restrict__ xold,
                                                                const E* __restrict__ z, const E* __restrict__ zold,
                                                                E* __restrict__ u, int64_t n, float* __restrict__ out) {
  __shared__ double sm[48];
  double dx = 0, dz = 0, du = 0, nx = 0, nz = 0, nxz = 0, nu = 0;
  auto sq = [](E a) { return (double)elem<E>::re(a) * (double)elem<E>::re(a) + (double)elem<E>::im(a) * (double)elem<E>::im(a); };
  for (int64_t i = threadIdx.x; i < n; i += UPD_THREADS) {
    const E xi = x[i], zi = z[i], ui = u[i];
    const E xz = elem<E>::sub(xi, zi);
    const E un = elem<E>::sub(elem<E>::add(ui, xi), zi);  // u += x ; u -= z   (:266-267)
    u[i] = un;
    dx += sq(elem<E>::sub(xi, xold[i]));
    dz += sq(elem<E>::sub(zi, zold[i]));
    du += sq(elem<E>::sub(un, ui));
    nx += sq(xi);
    nz += sq(zi);
    nxz += sq(xz);
    nu += sq(un);
  }
  block_sum3(dx, dz, du, sm);
  block_sum3(nx, nz, nxz, sm);
  nu = block_sum(nu, sm);
  if (threadIdx.x == 0) {
    out[0] = (float)sqrt(dx) + (float)sqrt(dz) + (float)sqrt(du);
    out[1] = (float)sqrt(dz);
    out[2] = fmaxf((float)sqrt(nx), (float)sqrt(nz));
    out[3] = (float)sqrt(nxz);
    out[4] = (float)sqrt(nu);
    out[5] = (float)sqrt(dx);
  }
}

// beta = beta_y + rho (z - u) ; xold = x      (src/ADMM.jl:236-243, identity regTrafo)
template <typename E>
__global__ void admm_pre_kernel(E* __restrict__ beta, const E* __restrict__ beta_y, const E* __restrict__ z,
                                const E* __restrict__ u, const E* __restrict__ x, E* __restrict__ xold, int64_t n,
                                float rho, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    E bi = accumulate ? beta[i] : beta_y[i];
    bi = elem<E>::add(bi, elem<E>::scale(rho, z[i]));
    bi = elem<E>::add(bi, elem<E>::scale(-rho, u[i]));
    beta[i] = bi;
    if (!accumulate) xold[i] = x[i];
  }
}

// c = AHA x is in place.  c += rho x ; r = b - c ; residual = ||r|| ; tol ; u = r (= r + beta*0)
template <typename E>
__global__ __launch_bounds__(UPD_THREADS) void cg_start_kernel(const E* __restrict__ x, const E* b,
                                                               E* __restrict__ u, E* __restrict__ r,
                                                               const E* __restrict__ c, int64_t n, cg_scalars* sc,
                                                               float rho, float reltol, int maxiter, admm_fuse<E> F,
                                                               col_batch<E> B) {
  __shared__ double sm[16];
  const int bq = blockIdx.x;
  {
    const int64_t o = (int64_t)bq * B.ldv;
    x += o; u += o; r += o; c += o;
    if (b) b += o;
    if (F.beta_y) { F.beta_y += o; F.z += o; F.u += o; F.beta += o; F.xold += o; }
    if (F.skip) F.skip += (int64_t)bq * B.skip_stride;
    sc += bq;
  }
  const panel_col<E> up = panel_column<E>(B.panel, n, bq, B.half);
  if (F.skip && *F.skip) {
    if (threadIdx.x == 0) {
      sc->iteration = 0;
      sc->maxiter = maxiter;
      sc->done = 1;
    }
    return;
  }
  double rr = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += UPD_THREADS) {
    const E cv = B.Vpart ? col_parts<E>(B, bq, n, i) : c[i];
    const E ci = elem<E>::add(cv, elem<E>::scale(rho, x[i]));
    const E ri = elem<E>::sub(admm_rhs<E>(F, b, x, i), ci);
    r[i] = ri;
    u[i] = ri;  // first iteration: beta = residual^2 / 1^2 multiplies u == 0
    if (B.panel) up.put(i, ri);
    rr += (double)elem<E>::re(ri) * (double)elem<E>::re(ri) + (double)elem<E>::im(ri) * (double)elem<E>::im(ri);
  }
  rr = block_sum(rr, sm);
  if (threadIdx.x == 0) {
    const double residual = sqrt(rr);
    const double tol = fmax((double)((float)reltol * (float)residual), 0.0);
    sc->residual = residual;
    sc->prev = 1.0;
    sc->tol = tol;
    sc->rho = rho;
    sc->reltol = reltol;
    sc->iteration = 0;
    sc->maxiter = maxiter;
    sc->done = (0 >= maxiter) || ((float)residual <= (float)tol);
  }
}

// after c = AHA u:  c += rho u ; alpha = residual^2 / <u, c> ; x += alpha u ; r -= alpha c ;
// residual = ||r|| ; then the next direction u = r + beta u
template <typename E>
__global__ __launch_bounds__(UPD_THREADS) void cg_update_kernel(E* __restrict__ x, E* __restrict__ u,
                                                                E* __restrict__ r, E* __restrict__ c, int64_t n,
                                                                cg_scalars* sc, col_batch<E> B) {
  const int bq = blockIdx.x;
  {
    const int64_t o = (int64_t)bq * B.ldv;
    x += o; u += o; r += o; c += o;
    sc += bq;
  }
  if (sc->done) return;
  const panel_col<E> up = panel_column<E>(B.panel, n, bq, B.half);
  __shared__ double sm[16];
  const float rho = sc->rho;
  double dre = 0.0, dim_ = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += UPD_THREADS) {
    const E ui = u[i];
    const E cv = B.Vpart ? col_parts<E>(B, bq, n, i) : c[i];
    const E ci = elem<E>::add(cv, elem<E>::scale(rho, ui));
    c[i] = ci;
    dre += (double)elem<E>::re(ui) * (double)elem<E>::re(ci) + (double)elem<E>::im(ui) * (double)elem<E>::im(ci);
    if constexpr (elem<E>::cplx)
      dim_ += (double)elem<E>::re(ui) * (double)elem<E>::im(ci) - (double)elem<E>::im(ui) * (double)elem<E>::re(ci);
  }
  dre = block_sum(dre, sm);
  if constexpr (elem<E>::cplx) dim_ = block_sum(dim_, sm);
  const double res = sc->residual;
  dcomplex alpha = dc_div({res * res, 0.0}, {dre, dim_});
  const E a = elem<E>::make((float)alpha.re, (float)alpha.im);
  const E na = elem<E>::make(-(float)alpha.re, -(float)alpha.im);
  double rr = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += UPD_THREADS) {
    x[i] = elem<E>::fma(a, u[i], x[i]);
    const E ri = elem<E>::fma(na, c[i], r[i]);
    r[i] = ri;
    rr += (double)elem<E>::re(ri) * (double)elem<E>::re(ri) + (double)elem<E>::im(ri) * (double)elem<E>::im(ri);
  }
  rr = block_sum(rr, sm);
  const double residual = sqrt(rr);
  const int it = sc->iteration + 1;
  const int done = (it >= sc->maxiter) || ((float)residual <= (float)sc->tol);
  if (!done) {
    const float beta = (float)(rr / (res * res));
    for (int64_t i = threadIdx.x; i < n; i += UPD_THREADS) {
      const E un = elem<E>::add(r[i], elem<E>::scale(beta, u[i]));
      u[i] = un;
      if (B.panel) up.put(i, un);
    }
  }
  if (threadIdx.x == 0) {
    sc->prev = res;
    sc->residual = residual;
    sc->iteration = it;
    sc->done = done;
  }
}

// ---------------------------------------------------------------------------------------------
// Gram matrix  G = A^H A  (setup GEMM, src/CGNR.jl:49).  LDS-tiled, f32 FMA; not on the
// per-iteration path.  64 x 64 output tile per workgroup, K streamed in 16-row panels.
// ---------------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(256) void gram_kernel(const E* __restrict__ A, int64_t lda, int64_t M, int64_t N,
                                                   E* __restrict__ G, int64_t ldg) {
  constexpr int T = 64, KP = 16;
  __shared__ E sa[KP][T + 1];
  __shared__ E sb[KP][T + 1];
  const int64_t i0 = (int64_t)blockIdx.x * T, j0 = (int64_t)blockIdx.y * T;
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;  // 16 x 16 threads, 4 x 4 outputs each
  E acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = elem<E>::zero();
  for (int64_t k0 = 0; k0 < M; k0 += KP) {
    for (int idx = threadIdx.x; idx < KP * T; idx += 256) {
      const int kk = idx % KP, cc = idx / KP;
      const int64_t k = k0 + kk;
      sa[kk][cc] = (k < M && i0 + cc < N) ? A[(i0 + cc) * lda + k] : elem<E>::zero();
      sb[kk][cc] = (k < M && j0 + cc < N) ? A[(j0 + cc) * lda + k] : elem<E>::zero();
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
      E av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = sa[kk][ty * 4 + a];
#pragma unroll
      for (int b = 0; b < 4; ++b) bv[b] = sb[kk][tx * 4 + b];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = elem<E>::fmac(av[a], bv[b], acc[a][b]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int64_t i = i0 + ty * 4 + a, j = j0 + tx * 4 + b;
      if (i < N && j < N) G[j * ldg + i] = acc[a][b];
    }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
template <typename S>
static int32_t alloc_scalars(rls_ctx* ctx, S** d, S** h, int n = 1) {
  RLS_HIP(ctx, dmalloc((void**)d, sizeof(S) * n));
  RLS_HIP(ctx, hipMemsetAsync(*d, 0, sizeof(S) * n, ctx->stream));
  RLS_HIP(ctx, hmalloc(h, sizeof(S) * n));
  memset(*h, 0, sizeof(S) * n);
  return 0;
}
template <typename S>
static int32_t fetch_scalars(rls_ctx* ctx, S* d, S* h) {
  static_assert(sizeof(S) % 4 == 0, "status structs are copied dword by dword");
  RLS_TRY(rls_fetch_add(ctx, d, h, sizeof(S)));
  return rls_fetch_wait(ctx);  // one publishing launch for everything queued (resident flags, logs), then the host sees it
}

// G = A^H A is Hermitian with a real diagonal (Julia's A'*A goes through herk).  The generic kernels compute
// both triangles independently, so the two images of an entry can differ in the last bit: copy the upper triangle
// over the lower one as its conjugate and clear the imaginary part of the diagonal (the 64 x 64 tile kernel is
// Hermitian by construction and does not need this).
template <typename E>
__global__ void hermitianize_kernel(E* __restrict__ G, int64_t ld, int64_t N) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // column
  const int64_t i = (int64_t)blockIdx.y * blockDim.y + threadIdx.y;  // row
  if (i >= N || j >= N || i < j) return;
  if (i == j) {
    G[i + j * ld] = elem<E>::make(elem<E>::re(G[i + j * ld]), 0.f);
  } else {  // i > j: lower triangle <- conj(upper)
    const E u = G[j + i * ld];
    G[i + j * ld] = elem<E>::make(elem<E>::re(u), -elem<E>::im(u));
  }
}

static int32_t gram_hermitianize(rls_ctx* ctx, int32_t dtype, int64_t N, void* G, int64_t ld) {
  const dim3 block(32, 8), grid((unsigned)((N + 31) / 32), (unsigned)((N + 7) / 8));
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(hermitianize_kernel<float>, grid, block, 0, ctx->stream, (float*)G, ld, N);
  else
    hipLaunchKernelGGL(hermitianize_kernel<float2>, grid, block, 0, ctx->stream, (float2*)G, ld, N);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// ADMM plan: whole outer iterations without host involvement (one regulariser, identity regTrafo)
// ---------------------------------------------------------------------------------------------
struct admm_scalars {
  int iteration, done, max_iter, pad;
  float rho, sigma_abs, rel_tol, pad2;
};
constexpr int ADMM_REC = 8;  // floats per log record: Delta, sk, eps_pri, rk, eps_dua, cg iterations, 0, 0

struct rls_admm {
  rls_cg* cg;
  rls_ctx* actx;
  uint64_t actx_id = 0;
  int device;
  rls_admm_params P;
  bool ready;
  admm_scalars *sc, *sc_h;
  float *log, *log_h;
  int log_cap;
  int enq;  // outer iterations enqueued since init (== device iteration unless the plan stopped early)
  int nrhs = 1;  // batched plans: sc / sc_h / log hold one entry per column
  int requested = 0;  // outer iterations asked for since init (capped at P.iterations)
  int fallbacks = 0;  // resident cg! launches lost and recovered (rls_admm_get_status)
};

// src/ADMM.jl:246-309 in ONE single-workgroup launch: projections on x, z = prox(x + u) (L1 / L2 inline; a TV prox
// has been written to `znew` by the FGP launch before this one), u += x - z, the seven norms (see admm_post_kernel),
// then `converged` / `done` (:324-330) decided HERE in Float32 exactly as the host would, and one log record.
template <typename E>
__global__ __launch_bounds__(UPD_THREADS) void admm_zu_kernel(E* __restrict__ x, const E* __restrict__ xold,
                                                              E* __restrict__ znew, const E* __restrict__ zold,
                                                              E* __restrict__ u, int64_t n, int reg_kind, float lam,
                                                              int proj_kind, int z_ready, admm_scalars* sc,
                                                              const int* cg_iterations, float* __restrict__ log,
                                                              col_batch<E> B, const cg_scalars* cgs) {
  __shared__ double sm[48];
  {
    const int bq = blockIdx.x;  // batched plans: one workgroup per column
    const int64_t o = (int64_t)bq * B.ldv;
    x += o; xold += o; znew += o; zold += o; u += o;
    sc += bq;
    log += (int64_t)bq * B.log_stride;
    if (cgs) cg_iterations = &cgs[bq].iteration;
  }
  if (sc->done) return;
  double dx = 0, dz = 0, du = 0, nx = 0, nz = 0, nxz = 0, nu = 0;
  auto sq = [](E a) { return (double)elem<E>::re(a) * (double)elem<E>::re(a) + (double)elem<E>::im(a) * (double)elem<E>::im(a); };
  // four elements per thread and trip, every load of a trip requested before the first is used (the obvious loop was one
  // dependent memory round trip per element: 8.5 us for N = 4096, most of it waiting)
  constexpr int U = 4;
  for (int64_t i0 = threadIdx.x; i0 < n; i0 += (int64_t)U * UPD_THREADS) {
    E xa[U], ua[U], za[U], xo[U], zo[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int64_t i = i0 + (int64_t)q * UPD_THREADS;
      const int64_t ic = i < n ? i : i0;  // clamped address, masked below
      xa[q] = x[ic];
      ua[q] = u[ic];
      za[q] = z_ready ? znew[ic] : elem<E>::zero();
      xo[q] = xold[ic];
      zo[q] = zold[ic];
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int64_t i = i0 + (int64_t)q * UPD_THREADS;
      if (i >= n) continue;
      E xi = xa[q];
      if (proj_kind != RLS_PROJ_NONE) {
        xi = fista_proj_elem<E>(xi, proj_kind);
        x[i] = xi;
      }
      const E ui = ua[q];
      E zi;
      if (z_ready) {
        zi = za[q];
      } else {
        zi = fista_prox_elem<E>(elem<E>::add(xi, ui), reg_kind, lam);
        znew[i] = zi;
      }
      const E xz = elem<E>::sub(xi, zi);
      const E un = elem<E>::sub(elem<E>::add(ui, xi), zi);  // u += x ; u -= z   (:266-267)
      u[i] = un;
      dx += sq(elem<E>::sub(xi, xo[q]));
      dz += sq(elem<E>::sub(zi, zo[q]));
      du += sq(elem<E>::sub(un, ui));
      nx += sq(xi);
      nz += sq(zi);
      nxz += sq(xz);
      nu += sq(un);
    }
  }
  block_sum3(dx, dz, du, sm);
  block_sum3(nx, nz, nxz, sm);
  nu = block_sum(nu, sm);
  if (threadIdx.x == 0) {
    // one rounding per operation, as the host's Float32 arithmetic (f32_mul / f32_add: never fused into an FMA)
    const float rho = sc->rho;
    const float delta = f32_add(f32_add((float)sqrt(dx), (float)sqrt(dz)), (float)sqrt(du));
    const float sk = f32_mul(rho, (float)sqrt(dz));
    const float eps_pri = fmaxf((float)sqrt(nx), (float)sqrt(nz));
    const float rk = (float)sqrt(nxz);
    const float eps_dua = f32_mul(rho, (float)sqrt(nu));
    const float lim_pri = f32_add(sc->sigma_abs, f32_mul(sc->rel_tol, eps_pri));
    const float lim_dua = f32_add(sc->sigma_abs, f32_mul(sc->rel_tol, eps_dua));
    const bool conv = rk < lim_pri && sk < lim_dua;
    const int it = sc->iteration;
    float* rec = log + (int64_t)it * ADMM_REC;
    rec[0] = delta;
    rec[1] = sk;
    rec[2] = eps_pri;
    rec[3] = rk;
    rec[4] = eps_dua;
    rec[5] = (float)*cg_iterations;
    rec[6] = rec[7] = 0.f;
    sc->iteration = it + 1;
    sc->done = conv || (it + 1 >= sc->max_iter);
  }
}

template <typename E>
__global__ void admm_reset_kernel(admm_scalars* sc, int max_iter, float rho, float sigma_abs, float rel_tol) {
  sc += blockIdx.x;
  sc->iteration = 0;
  sc->max_iter = max_iter;
  sc->done = max_iter <= 0;
  sc->rho = rho;
  sc->sigma_abs = sigma_abs;
  sc->rel_tol = rel_tol;
}

// ---- batched FISTA (shared A, matrix-core products) ----------------------------------------------
static rls_skinny fista_skinny_desc(const rls_fista* s) {
  rls_skinny K;
  K.A = s->op->A;
  K.lda = s->op->lda;
  K.M = s->op->M;
  K.N = s->op->N;
  K.G = s->op->G;  // explicit AHA (src/CGNR.jl:49): every operator apply of the batched loop is ONE product over it
  K.ldg = s->op->ldg;
  K.nrhs = s->nrhs;
  K.half = s->half;
  K.ngroups = rls_skinny_groups(s->nrhs, s->half);
  K.splits = s->splits;
  K.X = K.R = K.P = K.V = nullptr;
  K.ldv = s->ldv;
  K.Ppack = s->Ypack;  // the T kernel's right operand: the extrapolated points
  K.Tpack = s->Tpack;
  K.Vpart = s->Vpart;
  K.ldvp = s->op->N;
  K.sc = nullptr;
  return K;
}

template <typename E>
static fista_batch<E> fista_batch_desc(const rls_fista* s) {
  return fista_batch<E>{s->ldv, (const E*)s->Vpart, s->splits, rls_skinny_pad(s->nrhs, s->half), (E*)s->Ypack, s->half};
}

static int32_t fista_enqueue_batched(rls_fista* s) {
  rls_operator* op = s->op;
  rls_ctx* ctx = op->ctx;
  RLS_TRY(rls_skinny_launch(ctx, op->dtype, fista_skinny_desc(s), 1 | 2));
  if (op->dtype == RLS_F32)
    RLS_TRY(fista_launch_update<float>(s, (unsigned)s->nrhs, fista_batch_desc<float>(s)));
  else
    RLS_TRY(fista_launch_update<float2>(s, (unsigned)s->nrhs, fista_batch_desc<float2>(s)));
  return launch_status(ctx);
}

// type-erased admm_fuse (null beta_y = plain cg!)
struct admm_fuse_v {
  const void *beta_y = nullptr, *z = nullptr, *u = nullptr;
  void *beta = nullptr, *xold = nullptr;
  float rho = 0.f;
  const int* skip = nullptr;
  int* poison = nullptr;  // rls_cg_start::poison
};
template <typename E>
static admm_fuse<E> typed_fuse(const admm_fuse_v& V) {
  admm_fuse<E> F;
  F.beta_y = (const E*)V.beta_y;
  F.z = (const E*)V.z;
  F.u = (const E*)V.u;
  F.beta = (E*)V.beta;
  F.xold = (E*)V.xold;
  F.rho = V.rho;
  F.skip = V.skip;
  return F;
}

static int32_t cg_solve_impl(rls_cg* s, void* x, const void* b, float rho, int32_t maxiter, float reltol,
                             const admm_fuse_v& FV) {
  rls_operator* op = s->op;
  rls_ctx* ctx = op->ctx;
  const int64_t n = op->N;
  const admm_fuse<float> Ff = typed_fuse<float>(FV);
  const admm_fuse<float2> Fc = typed_fuse<float2>(FV);
  if (cg_use_gram_pipeline(s) && s->gram_resident && s->rsync && !s->resident_off && ctx->tune.resident && maxiter > 0) {
    // Gram mode, AHA in the register files: the whole cg! -- warm-start apply, r = b - (AHA + rho I) x (with beta formed on
    // the way for ADMM), every iteration -- is ONE launch of cgnr_gram_resident_kernel (normal.hip)
    s->used_pipeline = true;
    s->resident_used = true;
    const rls_gram_pipe P = cg_gram_desc(s, x);
    rls_cg_start St;
    St.enabled = 1;
    St.b = b;
    St.beta_y = FV.beta_y;
    St.z = FV.z;
    St.u = FV.u;
    St.beta = FV.beta;
    St.xold = FV.xold;
    St.rho_admm = FV.rho;
    St.rho = rho;
    St.reltol = reltol;
    St.maxiter = maxiter;
    St.skip = FV.skip;
    St.poison = FV.poison;
    return resident_chain(ctx, s->rsync, [&]() {
      return rls_gram_resident_launch(ctx, op->dtype, P, s->rsync, maxiter, (unsigned)ctx->tune.resident_spin, St);
    });
  }
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool resident_mf = cg_use_pipeline(s) && !cg_use_gram_pipeline(s) && s->rsync && !s->resident_off &&
                           ctx->tune.resident && maxiter > 0 &&
                           al16(x) && al16(s->r) && al16(s->u) && al16(s->c) && al16(b) && al16(FV.beta_y) && al16(FV.z) &&
                           al16(FV.u) && al16(FV.beta) && al16(FV.xold);
  // warm start: one operator apply for r = b - (AHA + rho I) x   (inside the resident launch where that runs)
  if (!resident_mf) RLS_TRY(op_normal(op, x, s->c, FV.skip));
  if (cg_use_gram_pipeline(s)) {
    s->used_pipeline = true;
    if (op->dtype == RLS_F32)
      hipLaunchKernelGGL(cg_pipe_start_kernel<float>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (const float*)x,
                         (const float*)b, (float*)s->u, (float*)s->r, (const float*)s->c, n, s->psc, rho, reltol,
                         maxiter, Ff);
    else
      hipLaunchKernelGGL(cg_pipe_start_kernel<float2>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (const float2*)x,
                         (const float2*)b, (float2*)s->u, (float2*)s->r, (const float2*)s->c, n, s->psc, rho, reltol,
                         maxiter, Fc);
    RLS_TRY(launch_status(ctx));
    const rls_gram_pipe P = cg_gram_desc(s, x);
    const int32_t dtype = op->dtype;
    if (s->graph.exec && (s->graph.x_bound != x || s->graph.mode != 3)) {  // the captured kernels carry x's address
      hipGraphExecDestroy(s->graph.exec);
      s->graph = step_graph();
    }
    s->graph.x_bound = x;
    s->graph.mode = 3;
    int parity = 0;
    auto one = [ctx, dtype, &P, &parity]() {
      const int32_t st = rls_gram_pipe_iteration(ctx, dtype, P, parity);
      parity ^= 1;
      return st;
    };
    if (ctx->tune.graph_chunk % 2) {
      for (int i = 0; i < maxiter; ++i) RLS_TRY(one());
    } else {
      RLS_TRY(run_steps(ctx, &s->graph, maxiter, one, [&parity](int c) { parity ^= c & 1; }));
    }
    return rls_gram_pipe_finish(ctx, dtype, P, maxiter & 1);
  }
  s->used_pipeline = cg_use_pipeline(s);
  if (s->used_pipeline) {
    if (resident_mf) {
      // matrix-free, A in the register files: warm-start apply, residual (ADMM's beta formed on the way) and every
      // iteration in ONE launch of cgnr_resident_kernel (normal.hip)
      s->resident_used = true;
      const rls_cgnr_pipe P = cg_pipe_desc(s, x);
      rls_cg_start St;
      St.enabled = 1;
      St.b = b;
      St.beta_y = FV.beta_y;
      St.z = FV.z;
      St.u = FV.u;
      St.beta = FV.beta;
      St.xold = FV.xold;
      St.rho_admm = FV.rho;
      St.rho = rho;
      St.reltol = reltol;
      St.maxiter = maxiter;
      St.skip = FV.skip;
      St.poison = FV.poison;
      return resident_chain(ctx, s->rsync, [&]() {
        return rls_cgnr_resident_launch(ctx, op->dtype, P, s->rdots, s->rsync, maxiter, (unsigned)ctx->tune.resident_spin, St);
      });
    }
    if (op->dtype == RLS_F32)
      hipLaunchKernelGGL(cg_pipe_start_kernel<float>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (const float*)x,
                         (const float*)b, (float*)s->u, (float*)s->r, (const float*)s->c, n, s->psc, rho, reltol,
                         maxiter, Ff);
    else
      hipLaunchKernelGGL(cg_pipe_start_kernel<float2>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (const float2*)x,
                         (const float2*)b, (float2*)s->u, (float2*)s->r, (const float2*)s->c, n, s->psc, rho, reltol,
                         maxiter, Fc);
    RLS_TRY(launch_status(ctx));
    rls_cgnr_pipe P = cg_pipe_desc(s, x);
    const int32_t dtype = op->dtype;
    if (s->graph.exec && (s->graph.x_bound != x || s->graph.mode != 1)) {  // the captured kernels carry x's address
      hipGraphExecDestroy(s->graph.exec);
      s->graph = step_graph();
    }
    s->graph.x_bound = x;
    s->graph.mode = 1;
    int k = 0;
    RLS_TRY(run_steps(ctx, &s->graph, maxiter, [ctx, dtype, &P, &k]() {
      P.cur_hint = pipe_cur_hint(ctx, k++);
      return rls_cgnr_pipe_iteration(ctx, dtype, P);
    }, [&k](int c) { k -= c; }));
    return rls_cgnr_pipe_finish(ctx, dtype, P);
  }
  if (op->dtype == RLS_F32)
    hipLaunchKernelGGL(cg_start_kernel<float>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (const float*)x,
                       (const float*)b, (float*)s->u, (float*)s->r, (const float*)s->c, n, s->sc, rho, reltol, maxiter,
                       Ff, col_batch<float>());
  else
    hipLaunchKernelGGL(cg_start_kernel<float2>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (const float2*)x,
                       (const float2*)b, (float2*)s->u, (float2*)s->r, (const float2*)s->c, n, s->sc, rho, reltol,
                       maxiter, Fc, col_batch<float2>());
  RLS_TRY(launch_status(ctx));
  for (int it = 0; it < maxiter; ++it) {
    RLS_TRY(op_normal(op, s->u, s->c, &s->sc->done));
    if (op->dtype == RLS_F32)
      hipLaunchKernelGGL(cg_update_kernel<float>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (float*)x, (float*)s->u,
                         (float*)s->r, (float*)s->c, n, s->sc, col_batch<float>());
    else
      hipLaunchKernelGGL(cg_update_kernel<float2>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (float2*)x,
                         (float2*)s->u, (float2*)s->r, (float2*)s->c, n, s->sc, col_batch<float2>());
    RLS_TRY(launch_status(ctx));
  }
  return 0;
}

// ---- batched ADMM (shared A): every column's outer iteration advances together --------------------------------
// Per outer iteration: X -> operand panel, T = A X and V = A^H T on the matrix cores (the warm-start apply of every
// column's cg!), the per-column start kernel (beta = beta_y + rho (z - u), r, u; next operand = u), iterations_cg x
// {T = A U, V = A^H T, per-column CG update}, the FGP launch with one workgroup per column for a TV term, and the
// per-column z / u / residual-norm / `done` kernel.  Columns retire independently (per-column device flags), exactly
// as K independent solves do (src/MultiThreading.jl:60-78).
static rls_skinny cg_skinny_desc(const rls_cg* s) {
  rls_skinny K;
  K.A = s->op->A;
  K.lda = s->op->lda;
  K.M = s->op->M;
  K.N = s->op->N;
  K.G = s->op->G;  // explicit AHA (src/CGNR.jl:49): every operator apply of the batched loop is ONE product over it
  K.ldg = s->op->ldg;
  K.nrhs = s->nrhs;
  K.half = s->half;
  K.ngroups = rls_skinny_groups(s->nrhs, s->half);
  K.splits = s->splits;
  K.X = K.R = K.P = K.V = nullptr;
  K.ldv = s->ldv;
  K.Ppack = s->Ppack;
  K.Tpack = s->Tpack;
  K.Vpart = s->Vpart;
  K.ldvp = s->op->N;
  K.sc = nullptr;
  return K;
}

template <typename E>
static int32_t admm_step_batched_typed(rls_admm* a, int32_t n_outer) {
  rls_cg* cg = a->cg;
  rls_ctx* ctx = cg->op->ctx;
  const rls_admm_params& P = a->P;
  const int32_t dtype = cg->op->dtype;
  const int64_t n = cg->op->N;
  const unsigned K = (unsigned)a->nrhs;
  const rls_skinny SK = cg_skinny_desc(cg);
  col_batch<E> B;
  B.ldv = cg->ldv;
  B.Vpart = (const E*)cg->Vpart;
  B.S = cg->splits;
  B.nrhs_pad = rls_skinny_pad(a->nrhs, cg->half);
  B.panel = (E*)cg->Ppack;
  B.half = cg->half;
  B.skip_stride = (int)(sizeof(admm_scalars) / sizeof(int));
  B.log_stride = (int64_t)ADMM_REC * a->log_cap;
  col_batch<E> Bz = B;  // the z / u kernel reads no partial rows and writes no panel
  Bz.Vpart = nullptr;
  Bz.panel = nullptr;
  for (int k = 0; k < n_outer && a->enq < P.iterations; ++k, ++a->enq) {
    E* zcur = (E*)((a->enq & 1) ? P.z1 : P.z0);
    E* znew = (E*)((a->enq & 1) ? P.z0 : P.z1);
    hipLaunchKernelGGL(pack_panel_kernel<E>, dim3((unsigned)((n + 255) / 256 < 64 ? (n + 255) / 256 : 64), K), dim3(256), 0,
                       ctx->stream, (const E*)P.x, cg->ldv, (E*)cg->Ppack, n, cg->half);
    RLS_TRY(rls_skinny_launch(ctx, dtype, SK, 1 | 2));  // AHA x of every column  (:244, warm start)
    admm_fuse<E> F;
    F.beta_y = (const E*)P.beta_y;
    F.z = zcur;
    F.u = (const E*)P.u;
    F.beta = (E*)P.beta;
    F.xold = (E*)P.xold;
    F.rho = P.rho;
    F.skip = &a->sc->done;
    hipLaunchKernelGGL(cg_start_kernel<E>, dim3(K), dim3(UPD_THREADS), 0, ctx->stream, (const E*)P.x, (const E*)P.beta,
                       (E*)cg->u, (E*)cg->r, (const E*)cg->c, n, cg->sc, P.rho, P.tol_inner, P.iterations_cg, F, B);
    for (int it = 0; it < P.iterations_cg; ++it) {
      RLS_TRY(rls_skinny_launch(ctx, dtype, SK, 1 | 2));
      hipLaunchKernelGGL(cg_update_kernel<E>, dim3(K), dim3(UPD_THREADS), 0, ctx->stream, (E*)P.x, (E*)cg->u, (E*)cg->r,
                         (E*)cg->c, n, cg->sc, B);
    }
    const int z_ready = P.reg_kind == RLS_REG_TV;
    if (z_ready)
      RLS_TRY(rls_tv_single_launch(ctx, dtype, P.tv_ndims, P.tv_shape, P.tv_ntv, P.tv_dims, P.x, P.u, znew, P.prox_lambda,
                                   P.tv_iterations, &a->sc->done, (int)K, cg->ldv, B.skip_stride));
    hipLaunchKernelGGL(admm_zu_kernel<E>, dim3(K), dim3(UPD_THREADS), 0, ctx->stream, (E*)P.x, (const E*)P.xold, znew,
                       (const E*)zcur, (E*)P.u, n, P.reg_kind, P.prox_lambda, P.proj_kind, z_ready, a->sc,
                       (const int*)nullptr, a->log, Bz, (const cg_scalars*)cg->sc);
    RLS_TRY(launch_status(ctx));
  }
  return 0;
}

static int32_t admm_step_batched(rls_admm* a, int32_t n_outer) {
  return a->cg->op->dtype == RLS_F32 ? admm_step_batched_typed<float>(a, n_outer) : admm_step_batched_typed<float2>(a, n_outer);
}

// ---- helpers of the row-sharded entry points (templates: C++ linkage) --------------------------------------------------
template <typename PlanT>
static int32_t rowsharded_check(rls_comm* comm, PlanT* const* plans, rls_operator* (*op_of)(PlanT*), bool (*single)(PlanT*)) {
  if (!comm || !plans) return RLS_E_INVALID;
  const int n = rls_comm_size(comm);
  for (int r = 0; r < n; ++r) {
    rls_ctx* cr = nullptr;
    RLS_TRY(rls_comm_ctx(comm, r, &cr));
    if (!plans[r]) return rls_fail(cr, RLS_E_INVALID, "rowsharded: null plan");
    rls_operator* op = op_of(plans[r]);
    if (op->ctx != cr) return rls_fail(cr, RLS_E_INVALID, "rowsharded: plan r must live on the communicator's context r");
    if (!single(plans[r]) || op->N != op_of(plans[0])->N || op->dtype != op_of(plans[0])->dtype)
      return rls_fail(cr, RLS_E_INVALID, "rowsharded: the shards must share N and the element type (single right-hand side)");
    if (!op->A) return rls_fail(cr, RLS_E_INVALID, "rowsharded: a row shard needs its matrix (no Gram-only operators)");
  }
  return 0;
}
static rls_operator* cgnr_op(rls_cgnr* s) { return s->op; }
static bool cgnr_single(rls_cgnr* s) { return s->nrhs == 1; }
static rls_operator* fista_op(rls_fista* s) { return s->op; }
static bool fista_single(rls_fista* s) { return s->nrhs == 1; }
static rls_operator* admm_op(rls_admm* a) { return a->cg->op; }
static bool admm_single(rls_admm* a) { return a->nrhs == 1 && !a->cg->Vpart; }

template <typename E>
static void admm_local_start(rls_admm* a, const admm_fuse_v& F) {
  rls_cg* cg = a->cg;
  rls_ctx* ctx = cg->op->ctx;
  const rls_admm_params& P = a->P;
  hipLaunchKernelGGL(cg_start_kernel<E>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (const E*)P.x, (const E*)P.beta, (E*)cg->u,
                     (E*)cg->r, (const E*)cg->c, cg->op->N, cg->sc, P.rho, P.tol_inner, P.iterations_cg, typed_fuse<E>(F),
                     col_batch<E>());
}
template <typename E>
static void admm_local_finish(rls_admm* a, void* zcur, void* znew) {
  rls_cg* cg = a->cg;
  rls_ctx* ctx = cg->op->ctx;
  const rls_admm_params& P = a->P;
  hipLaunchKernelGGL(admm_zu_kernel<E>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (E*)P.x, (const E*)P.xold, (E*)znew,
                     (const E*)zcur, (E*)P.u, cg->op->N, P.reg_kind, P.prox_lambda, P.proj_kind,
                     P.reg_kind == RLS_REG_TV ? 1 : 0, a->sc, &cg->sc->iteration, a->log, col_batch<E>(), nullptr);
}

extern "C" {

int32_t rls_operator_create(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda,
                            rls_operator** out) {
  RLS_CHECK_CTX(ctx);
  if (!out || !rls_dtype_ok(dtype) || M < 0 || N <= 0) return rls_fail(ctx, RLS_E_INVALID, "operator_create: bad argument");
  if (A && (M <= 0 || lda < M)) return rls_fail(ctx, RLS_E_INVALID, "operator_create: bad shape/lda");
  RLS_HIP(ctx, rls_enter(ctx));
  rls_alloc_scope alloc_scope(ctx);
  rls_operator* op = new rls_operator();
  op->ctx = ctx;
  op->ctx_id = ctx->id;
  op->dtype = dtype;
  op->M = M;
  op->N = N;
  op->A = A;
  op->lda = lda;
  op->G = nullptr;
  op->ldg = 0;
  op->t = nullptr;
  op->slab = nullptr;
  if (A) {
    hipError_t e = dmalloc(&op->t, (size_t)M * rls_elem_size(dtype));
    const size_t ws = rls_normal_fused_workspace(ctx, dtype, M, N, A, lda);
    if (e == hipSuccess && ws > 0) e = dmalloc(&op->slab, ws);
    if (e != hipSuccess) {
      if (op->t) dfree(op->t);
      delete op;
      return rls_fail(ctx, (int32_t)e, "operator_create: hipMalloc failed");
    }
  }
  *out = op;
  return 0;
}

int32_t rls_operator_set_gram(rls_operator* op, const void* AHA, int64_t ld) {
  if (!op) return RLS_E_INVALID;
  if (AHA && ld < op->N) return rls_fail(op->ctx, RLS_E_INVALID, "operator_set_gram: ld < N");
  op->G = AHA;
  op->ldg = ld;
  return 0;
}

int32_t rls_operator_destroy(rls_operator* op) {
  if (!op) return RLS_E_INVALID;
  rls_alloc_scope alloc_scope(alloc_ctx_of(op->ctx, op->ctx_id));
  if (op->slab) dfree(op->slab);
  if (op->t) dfree(op->t);  // hipFree resolves the owning device from the pointer
  delete op;
  return 0;
}

int32_t rls_operator_mul(rls_operator* op, const void* x, void* y) {
  if (!op) return RLS_E_INVALID;
  if (!op->A) return rls_fail(op->ctx, RLS_E_STATE, "operator_mul: operator has no forward matrix");
  return rls_launch_gemv(op->ctx, op->dtype, RLS_OP_N, op->M, op->N, 1.f, 0.f, op->A, op->lda, x, 0.f, 0.f, y, nullptr);
}
int32_t rls_operator_mul_adj(rls_operator* op, const void* y, void* x) {
  if (!op) return RLS_E_INVALID;
  if (!op->A) return rls_fail(op->ctx, RLS_E_STATE, "operator_mul_adj: operator has no forward matrix");
  return rls_launch_gemv(op->ctx, op->dtype, RLS_OP_C, op->M, op->N, 1.f, 0.f, op->A, op->lda, y, 0.f, 0.f, x, nullptr);
}
// the same with a device flag that turns the launches into no-ops (deferred OptISTA / POGM sequences)
int32_t rls_operator_mul_normal_skip(rls_operator* op, const void* p, void* v, const void* skip_d) {
  if (!op) return RLS_E_INVALID;
  rls_ctx* ctx = op->ctx;
  if (!p || !v) return rls_fail(ctx, RLS_E_INVALID, "operator_mul_normal_skip: null pointer");
  RLS_HIP(ctx, rls_enter(ctx));
  return op_normal(op, p, v, (const int*)skip_d);
}

int32_t rls_operator_mul_normal(rls_operator* op, const void* p, void* v) {
  if (!op || !p || !v) return RLS_E_INVALID;
  return op_normal(op, p, v, nullptr);
}

int32_t rls_gram(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda, void* G, int64_t ld) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || M <= 0 || N <= 0 || !A || !G || lda < M || ld < N)
    return rls_fail(ctx, RLS_E_INVALID, "gram: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  if (ctx->tune.batched_mfma && rls_gram_tiles_ok(M, N) && rls_skinny_ok(dtype, M, N, A, lda))
    return rls_gram_tiles(ctx, dtype, M, N, A, lda, G, ld);  // Hermitian 64 x 64 tiles, no scratch
  if (ctx->tune.batched_mfma && N <= 65535 * 16 && rls_skinny_ok(dtype, M, N, A, lda)) {
    // matrix cores: A^H T with T = A in 16-column panels (skinny.hip); M x N scratch for the panels
    void* panels = nullptr;
    rls_alloc_scope alloc_scope(ctx);
    RLS_HIP(ctx, dmalloc(&panels, (size_t)M * (size_t)N * rls_elem_size(dtype)));
    int32_t st = rls_skinny_gram(ctx, dtype, M, N, A, lda, G, ld, panels);
    if (st == 0) st = gram_hermitianize(ctx, dtype, N, G, ld);
    hipError_t e = rls_stream_wait(ctx->stream);  // setup path: the scratch is freed before returning
    dfree(panels);
    if (st == 0 && e != hipSuccess) st = rls_fail(ctx, (int32_t)e, hipGetErrorString(e));
    return st;
  }
  dim3 grid((unsigned)((N + 63) / 64), (unsigned)((N + 63) / 64));
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(gram_kernel<float>, grid, dim3(256), 0, ctx->stream, (const float*)A, lda, M, N, (float*)G, ld);
  else
    hipLaunchKernelGGL(gram_kernel<float2>, grid, dim3(256), 0, ctx->stream, (const float2*)A, lda, M, N, (float2*)G, ld);
  RLS_TRY(launch_status(ctx));
  return gram_hermitianize(ctx, dtype, N, G, ld);
}

// ---- CGNR -----------------------------------------------------------------------------------
static int32_t cgnr_create_impl(rls_operator* op, int32_t nrhs, void* x, void* r, void* p, void* v, int64_t ldv,
                                rls_cgnr** out) {
  if (!op) return RLS_E_INVALID;
  rls_ctx* ctx = op->ctx;
  if (!x || !r || !p || !v || !out || nrhs < 1 || ldv < op->N)
    return rls_fail(ctx, RLS_E_INVALID, "cgnr_create: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  // the columns share solver.AHA (src/MultiThreading.jl:30-48): an explicit Gram matrix when the operator has one -- the
  // reference constructors' default for a dense matrix, src/CGNR.jl:49 --, otherwise the two products over A
  const bool skinny = nrhs > 1 && ctx->tune.batched_mfma && rls_skinny_ok(op->dtype, op->M, op->N, op->A, op->lda) &&
                      (!op->G || rls_skinny_ok(op->dtype, op->N, op->N, op->G, op->ldg));
  if (nrhs > 1 && !skinny)
    return rls_fail(ctx, RLS_E_UNSUPPORTED, "batched CGNR runs on the matrix cores: A (and AHA, when explicit) with M, N multiples "
                                            "of 16 and 16-byte aligned columns (other shapes: one plan per column)");
  rls_alloc_scope alloc_scope(ctx);
  rls_cgnr* s = new rls_cgnr();
  s->actx = ctx;
  s->actx_id = ctx->id;
  s->skinny = skinny;
  s->gram_pipe = false;
  s->v1 = nullptr;
  s->gdots = nullptr;
  s->Ppack = s->Tpack = nullptr;
  s->Vpart = nullptr;
  s->splits = 1;
  s->op = op;
  s->device = ctx->device;
  s->x = x;
  s->r = r;
  s->p = p;
  s->v = v;
  s->initialised = false;
  s->r1 = s->p1 = nullptr;
  s->dots = nullptr;
  s->scn = nullptr;
  s->sc = nullptr;
  s->sc_h = nullptr;
  s->nrhs = nrhs;
  s->ldv = ldv;
  s->slab_b = nullptr;
  s->rsync = nullptr;
  s->rdots = nullptr;
  s->rsync_h = nullptr;
  s->resident_used = false;
  s->gram_resident = false;
  s->resident_off = false;
  s->fallbacks = 0;
  s->requested = 0;
  const size_t sb = sizeof(cgnr_scalars) * (size_t)nrhs;
  hipError_t e = dmalloc(&s->sc, sb);
  if (e == hipSuccess) e = hipMemsetAsync(s->sc, 0, sb, ctx->stream);
  if (e == hipSuccess) e = hmalloc(&s->sc_h, sb);
  if (e == hipSuccess) memset(s->sc_h, 0, sb);
  if (e == hipSuccess && op->slab) {  // scratch of the fused pipeline
    const size_t vb = (size_t)ldv * nrhs * rls_elem_size(op->dtype);
    const size_t nd = (size_t)((op->N + 15) / 16) * 4 * sizeof(double) * nrhs;
    e = dmalloc(&s->r1, vb);
    if (e == hipSuccess) e = dmalloc(&s->p1, vb);
    if (e == hipSuccess) e = hipMemsetAsync(s->r1, 0, vb, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->p1, 0, vb, ctx->stream);
    if (e == hipSuccess) e = dmalloc(&s->dots, nd);
    if (e == hipSuccess && op->A && !op->G) {  // matrix-free: K_A's ||t_w||^2 per row block (alpha = zeta / ||A p||^2)
      const size_t tb = (size_t)rls_cgnr_resident_nwg(op->ctx, op->dtype, op->M, op->N) * sizeof(double) * nrhs;
      e = dmalloc(&s->ttw, tb);
      if (e == hipSuccess) e = hipMemsetAsync(s->ttw, 0, tb, ctx->stream);
    }
    if (e == hipSuccess) e = dmalloc(&s->scn, sb);
    if (e == hipSuccess) e = hipMemsetAsync(s->dots, 0, nd, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->scn, 0, sb, ctx->stream);
    if (e == hipSuccess && nrhs > 1 && !skinny)
      e = dmalloc(&s->slab_b, rls_normal_fused_workspace(op->ctx, op->dtype, op->M, op->N, op->A, op->lda) * (size_t)nrhs);
  }
  s->small = nrhs == 1 && op->A && !op->G && rls_small_ok(op->dtype, op->M, op->N, op->A, op->lda);
  if (e == hipSuccess && s->small && !s->srv.ctl && hmalloc(&s->srv.ctl, 32 * sizeof(unsigned)) == hipSuccess) {
    memset(s->srv.ctl, 0, 32 * sizeof(unsigned));  // (the single-workgroup kernel can stay and listen as well: rls_cgnr_step_status)
    s->srv.resident_used = nullptr;                // (no co-residency requirement, nothing to give up: no flags to read)
  }
  if (e == hipSuccess && nrhs == 1 && op->slab && op->A && !op->G &&
      rls_cgnr_resident_ok(ctx, op->dtype, op->M, op->N, op->A, op->lda)) {
    const size_t db = (size_t)rls_cgnr_resident_nwg(op->ctx, op->dtype, op->M, op->N) * 4 * sizeof(double);
    e = resident_alloc(ctx, op, &s->rsync, &s->rsync_h);
    if (e == hipSuccess) e = dmalloc(&s->rdots, db);
    if (e == hipSuccess) e = hipMemsetAsync(s->rdots, 0, db, ctx->stream);
    if (e == hipSuccess && !s->srv.ctl) e = hmalloc(&s->srv.ctl, 32 * sizeof(unsigned));
    if (e == hipSuccess) memset(s->srv.ctl, 0, 32 * sizeof(unsigned));
    s->srv.resident_used = &s->resident_used;
  }
  if (e == hipSuccess && nrhs == 1 && op->G && rls_gram_pipe_ok(op->dtype, op->N, op->G, op->ldg)) {
    const size_t vb = (size_t)op->N * rls_elem_size(op->dtype);
    const size_t nd = (size_t)2 * rls_gram_pipe_nwg(op->dtype, op->N) * 4 * sizeof(double);
    if (!s->r1) e = dmalloc(&s->r1, vb);
    if (e == hipSuccess && !s->p1) e = dmalloc(&s->p1, vb);
    if (e == hipSuccess) e = dmalloc(&s->v1, vb);
    if (e == hipSuccess) e = dmalloc(&s->gdots, nd);
    if (e == hipSuccess && !s->scn) e = dmalloc(&s->scn, sb);
    if (e == hipSuccess) e = hipMemsetAsync(s->r1, 0, vb, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->p1, 0, vb, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->v1, 0, vb, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->gdots, 0, nd, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->scn, 0, sb, ctx->stream);
    s->gram_pipe = e == hipSuccess;
    if (e == hipSuccess && !s->rsync && rls_gram_resident_ok(ctx, op->dtype, op->N, op->G, op->ldg)) {
      e = resident_alloc(ctx, op, &s->rsync, &s->rsync_h);
      s->gram_resident = e == hipSuccess;
      if (e == hipSuccess && !s->srv.ctl && rls_gram_resident_server_ok(op->dtype, op->N) &&
          hmalloc(&s->srv.ctl, 32 * sizeof(unsigned)) == hipSuccess) {
        memset(s->srv.ctl, 0, 32 * sizeof(unsigned));
        s->srv.resident_used = &s->resident_used;
      }
    }
  }
  if (e == hipSuccess && skinny) {
    size_t pb, tb, vb;
    rls_skinny_sizes(op->ctx, op->dtype, op->M, op->N, nrhs, &pb, &tb, &vb, &s->splits);
    s->half = rls_skinny_half(op->ctx, op->dtype, nrhs);
    e = dmalloc(&s->Ppack, pb);
    if (e == hipSuccess) e = hipMemsetAsync(s->Ppack, 0, pb, ctx->stream);  // the padding columns of the last group stay zero
    if (e == hipSuccess) e = dmalloc(&s->Tpack, tb);
    if (e == hipSuccess) e = dmalloc(&s->Vpart, vb);
    if (e == hipSuccess && op->G && s->half && rls_gramk_resident_ok(ctx, op->dtype, op->N, nrhs, op->G, op->ldg)) {
      size_t xb, xxb, db;
      rls_gramk_sizes(op->N, &xb, &xxb, &db);
      e = resident_alloc(ctx, op, &s->rsync, &s->rsync_h);
      if (e == hipSuccess) e = dmalloc(&s->gk_vx, xb);
      if (e == hipSuccess) e = hipMemsetAsync(s->gk_vx, 0, xb, ctx->stream);  // rows >= N are read, never written
      if (e == hipSuccess) e = dmalloc(&s->gk_xx, xxb);
      if (e == hipSuccess) e = dmalloc(&s->gk_dots, db);
      if (e == hipSuccess) e = hipMemsetAsync(s->gk_dots, 0, db, ctx->stream);  // slots of absent workgroups add 0.0
      s->gramk = e == hipSuccess;
    }
  }
  if (e != hipSuccess) {
    rls_cgnr_destroy(s);
    return rls_fail(ctx, (int32_t)e, "cgnr_create: allocation failed");
  }
  *out = s;
  return 0;
}

int32_t rls_cgnr_create(rls_operator* op, void* x, void* r, void* p, void* v, rls_cgnr** out) {
  if (!op) return RLS_E_INVALID;
  return cgnr_create_impl(op, 1, x, r, p, v, op->N, out);
}

int32_t rls_cgnr_create_batched(rls_operator* op, int32_t nrhs, void* X, void* R, void* P, void* V, int64_t ldv,
                                rls_cgnr** out) {
  return cgnr_create_impl(op, nrhs, X, R, P, V, ldv, out);
}

int32_t rls_cgnr_destroy(rls_cgnr* s) {
  if (!s) return RLS_E_INVALID;
  if (rls_ctx_alive(s->actx, s->actx_id) && s->actx->server == &s->srv) rls_server_stop(s->actx);  // (a kernel of this plan left listening)
  hipSetDevice(s->device);
  rls_alloc_scope alloc_scope(alloc_ctx_of(s->actx, s->actx_id));
  if (s->graph.exec) hipGraphExecDestroy(s->graph.exec);
  if (s->r1) dfree(s->r1);
  if (s->p1) dfree(s->p1);
  if (s->dots) dfree(s->dots);
  if (s->ttw) dfree(s->ttw);
  if (s->scn) dfree(s->scn);
  if (s->slab_b) dfree(s->slab_b);
  if (s->v1) dfree(s->v1);
  if (s->gdots) dfree(s->gdots);
  if (s->Ppack) dfree(s->Ppack);
  if (s->Tpack) dfree(s->Tpack);
  if (s->Vpart) dfree(s->Vpart);
  if (s->gk_vx) dfree(s->gk_vx);
  if (s->gk_xx) dfree(s->gk_xx);
  if (s->gk_dots) dfree(s->gk_dots);
  if (s->q_b) dfree(s->q_b);
  if (s->q_bh) hfree(s->q_bh);
  if (s->q_xh) hfree(s->q_xh);
  if (s->rsync) dfree(s->rsync);
  if (s->rdots) dfree(s->rdots);
  if (s->rsync_h) hfree(s->rsync_h);
  if (s->srv.ctl) hfree(s->srv.ctl);
  if (s->sc) dfree(s->sc);
  if (s->sc_h) hfree(s->sc_h);
  delete s;
  return 0;
}

int32_t rls_cgnr_init_local_a(rls_cgnr* s, const void* b, float lambda, float rel_tol, int32_t iterations) {
  if (!s) return RLS_E_INVALID;
  rls_operator* op = s->op;
  rls_ctx* ctx = op->ctx;
  if (!b) return rls_fail(ctx, RLS_E_INVALID, "cgnr_init: null b");
  if (s->nrhs != 1) return rls_fail(ctx, RLS_E_STATE, "cgnr_init on a batched plan: use rls_cgnr_init_batched");
  RLS_HIP(ctx, rls_enter(ctx));
  // r = A^H b   (initCGNR, src/CGNR.jl:132) ; without A, b already is A^H b (:134)
  if (op->A)
    RLS_TRY(rls_launch_gemv(ctx, op->dtype, RLS_OP_C, op->M, op->N, 1.f, 0.f, op->A, op->lda, b, 0.f, 0.f, s->r, nullptr));
  else
    RLS_HIP(ctx, hipMemcpyAsync(s->r, b, (size_t)op->N * rls_elem_size(op->dtype), hipMemcpyDeviceToDevice, ctx->stream));
  s->sc_h->lambda = lambda;
  s->sc_h->rel_tol = rel_tol;
  s->sc_h->max_iter = cgnr_effective_iterations(s, iterations);
  return 0;
}

int32_t rls_cgnr_init_local_b(rls_cgnr* s) {
  if (!s) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  RLS_HIP(ctx, rls_enter(ctx));
  if (s->op->dtype == RLS_F32)
    cgnr_launch_init<float>(s, s->sc_h->lambda, s->sc_h->rel_tol, s->sc_h->max_iter);
  else
    cgnr_launch_init<float2>(s, s->sc_h->lambda, s->sc_h->rel_tol, s->sc_h->max_iter);
  s->initialised = true;
  s->requested = 0;
  s->srv.off = false;  // (a new solve: the caller's pattern between iterates is judged afresh)
  s->srv.short_lives = 0;
  return launch_status(ctx);
}

static int32_t cgnr_group(rls_cgnr* const* plans, const void* const* b, int32_t count, float lambda, float rel_tol, int32_t iterations,
                          int32_t n_steps);
int32_t rls_cgnr_init(rls_cgnr* s, const void* b, float lambda, float rel_tol, int32_t iterations) {
  // small systems: init! is the single-workgroup kernel's own (r = A^H b from the registers: ONE launch instead of a GEMV and an
  // init kernel, and the same bits whether a solve is init + steps or one fused launch, rls_cgnr_init_step_group)
  if (s && b && s->nrhs == 1 && s->op->A && cgnr_use_small(s)) return cgnr_group(&s, &b, 1, lambda, rel_tol, iterations, 0);
  RLS_TRY(rls_cgnr_init_local_a(s, b, lambda, rel_tol, iterations));
  return rls_cgnr_init_local_b(s);
}

int32_t rls_cgnr_init_batched(rls_cgnr* s, const void* B, int64_t ldb, float lambda, float rel_tol, int32_t iterations) {
  if (!s) return RLS_E_INVALID;
  rls_operator* op = s->op;
  rls_ctx* ctx = op->ctx;
  if (!B || !op->A || ldb < op->M) return rls_fail(ctx, RLS_E_INVALID, "cgnr_init_batched: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  const size_t es = rls_elem_size(op->dtype);
  const int max_iter = cgnr_effective_iterations(s, iterations);
  if (s->skinny) {
    // R = A^H B on the matrix cores (B packed as the T operand), then the per-column init
    RLS_TRY(rls_skinny_init(ctx, op->dtype, cgnr_skinny_desc(s), B, ldb, lambda, rel_tol, max_iter));
    s->sc_h->lambda = lambda;
    s->sc_h->rel_tol = rel_tol;
    s->sc_h->max_iter = max_iter;
    s->initialised = true;
    s->requested = 0;
    return 0;
  }
  if (!cgnr_use_pipeline(s)) return rls_fail(ctx, RLS_E_UNSUPPORTED, "cgnr_init_batched: fused pipeline not active");
  for (int b = 0; b < s->nrhs; ++b) {
    char* rb = (char*)s->r + (size_t)b * s->ldv * es;
    RLS_TRY(rls_launch_gemv(ctx, op->dtype, RLS_OP_C, op->M, op->N, 1.f, 0.f, op->A, op->lda,
                            (const char*)B + (size_t)b * ldb * es, 0.f, 0.f, rb, nullptr));
    char* xb = (char*)s->x + (size_t)b * s->ldv * es;
    char* pb = (char*)s->p + (size_t)b * s->ldv * es;
    char* vb = (char*)s->v + (size_t)b * s->ldv * es;
    if (op->dtype == RLS_F32)
      hipLaunchKernelGGL(cgnr_init_kernel<float>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (float*)xb,
                         (const float*)rb, (float*)pb, (float*)vb, op->N, s->sc + b, lambda, rel_tol, max_iter);
    else
      hipLaunchKernelGGL(cgnr_init_kernel<float2>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (float2*)xb,
                         (const float2*)rb, (float2*)pb, (float2*)vb, op->N, s->sc + b, lambda, rel_tol, max_iter);
  }
  s->sc_h->lambda = lambda;
  s->sc_h->rel_tol = rel_tol;
  s->sc_h->max_iter = max_iter;
  s->initialised = true;
  return launch_status(ctx);
}

static int32_t cgnr_step_impl(rls_cgnr* s, int32_t n_steps);
static void cgnr_status_out(const rls_cgnr* s, const cgnr_scalars& h, rls_cgnr_status* out);
int32_t rls_cgnr_get_status_batched(rls_cgnr* s, rls_cgnr_status* out) {
  if (!s || !out) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  if (!s->initialised) return rls_fail(ctx, RLS_E_STATE, "cgnr_get_status before cgnr_init");
  RLS_HIP(ctx, rls_enter(ctx));
  if (s->resident_used) RLS_TRY(resident_fetch_flags(ctx, s->rsync, s->rsync_h));
  RLS_TRY(rls_fetch_add(ctx, s->sc, s->sc_h, sizeof(cgnr_scalars) * (size_t)s->nrhs));
  RLS_TRY(rls_fetch_wait(ctx));
  if (s->resident_used && resident_lost(ctx, s->rsync, s->rsync_h, &s->resident_off, &s->fallbacks)) {
    // a lost resident launch changed nothing: the live columns are all at the same count (they advance in lockstep since
    // init; retired ones stay behind), so what is missing is requested - that count, re-run on the streaming kernels
    long long at = 0;
    bool live = false;
    for (int b = 0; b < s->nrhs; ++b) {
      if (s->sc_h[b].iteration > at) at = s->sc_h[b].iteration;
      live = live || !s->sc_h[b].done;
    }
    const long long missing = s->requested - at;
    if (live && missing > 0) {
      RLS_TRY(cgnr_step_impl(s, (int32_t)(missing > 0x7fffffff ? 0x7fffffff : missing)));
      RLS_TRY(rls_fetch_add(ctx, s->sc, s->sc_h, sizeof(cgnr_scalars) * (size_t)s->nrhs));
      RLS_TRY(rls_fetch_wait(ctx));
    }
  }
  s->resident_used = false;
  for (int b = 0; b < s->nrhs; ++b) {
    const cgnr_scalars& h = s->sc_h[b];
    out[b].iteration = h.iteration;
    out[b].done = h.done;
    out[b].alpha_re = (float)h.alpha_re;
    out[b].alpha_im = (float)h.alpha_im;
    out[b].beta_re = (float)h.beta_re;
    out[b].beta_im = (float)h.beta_im;
    out[b].zeta = (float)h.zeta;
    out[b].residual = (float)sqrt(h.rr);
    out[b].z0 = (float)h.z0;
    out[b].fallbacks = s->fallbacks;
  }
  return 0;
}

static int32_t cgnr_step_impl(rls_cgnr* s, int32_t n_steps) {
  rls_ctx* ctx = s->op->ctx;
  if (cgnr_use_small(s)) {
    if (n_steps == 0) return 0;
    rls_small D;
    D.A = s->op->A;
    D.lda = s->op->lda;
    D.M = s->op->M;
    D.N = s->op->N;
    D.x = s->x;
    D.r = s->r;
    D.p = s->p;
    D.v = s->v;
    D.sc = s->sc;
    D.mb = s->mb_arm;
    s->mb_sent = s->mb_arm.dst != nullptr;
    return rls_small_launch(ctx, s->op->dtype, D, n_steps);
  }
  if (s->skinny && cgnr_use_gramk(s, n_steps)) {
    if (n_steps == 0) return 0;
    const rls_gramk D = cgnr_gramk_desc(s);
    s->resident_used = true;
    return resident_chain(ctx, s->rsync, [&]() {
      return rls_gramk_resident_launch(ctx, D, s->rsync, n_steps, (unsigned)ctx->tune.resident_spin);
    }, &s->rsync_clean);
  }
  if (s->skinny) {
    const rls_skinny K = cgnr_skinny_desc(s);
    const int32_t dtype = s->op->dtype;
    if (s->graph.steps && s->graph.mode != 2) {
      hipGraphExecDestroy(s->graph.exec);
      s->graph = step_graph();
    }
    s->graph.mode = 2;
    return run_steps(ctx, &s->graph, n_steps, [ctx, dtype, &K]() { return rls_skinny_launch(ctx, dtype, K, 7); });
  }
  if (cgnr_use_gram_pipeline(s)) {
    // one launch per iteration; every call starts at parity 0 (the finish kernel brings the state back to
    // the caller's vectors), and a graph chunk holds an even number of launches, so the captured parities
    // are always the right ones
    const rls_gram_pipe P = cgnr_gram_desc(s);
    const int32_t dtype = s->op->dtype;
    if (cgnr_use_gram_resident(s)) {  // the whole call as ONE launch, AHA in registers, one grid exchange per iteration
      if (n_steps == 0) return 0;
      s->resident_used = true;
      return resident_chain(ctx, s->rsync, [&]() {
        return rls_gram_resident_launch(ctx, dtype, P, s->rsync, n_steps, (unsigned)ctx->tune.resident_spin);
      }, &s->rsync_clean);
    }
    if (s->graph.steps && s->graph.mode != 3) {
      hipGraphExecDestroy(s->graph.exec);
      s->graph = step_graph();
    }
    s->graph.mode = 3;
    int parity = 0;
    auto one = [ctx, dtype, &P, &parity]() {
      const int32_t st = rls_gram_pipe_iteration(ctx, dtype, P, parity);
      parity ^= 1;
      return st;
    };
    if (ctx->tune.graph_chunk % 2) {  // an odd chunk would replay the captured parities out of phase
      for (int i = 0; i < n_steps; ++i) RLS_TRY(one());
    } else {
      RLS_TRY(run_steps(ctx, &s->graph, n_steps, one, [&parity](int c) { parity ^= c & 1; }));
    }
    return rls_gram_pipe_finish(ctx, dtype, P, n_steps & 1);
  }
  // A resident launch pays for loading its slab of A (~10 us at the headline shape) before its first iteration: a call of
  // ONE iteration -- the reference's solve! loop with callbacks, one iterate per call -- is cheaper on the two-launch
  // pipeline (bench.py other_paths, iterate_per_call_cadence: 40 us against 52 us per call, host synchronisation included)
  if (cgnr_use_resident(s) && n_steps != 1) {
    // ONE launch for the whole call: A stays in the register files, iterations are separated by an in-kernel
    // grid-wide all-reduce (normal.hip).  The arrival counters and flags are zeroed ahead of every launch.
    if (n_steps == 0) return 0;
    const rls_cgnr_pipe P = cgnr_pipe_desc(s);
    s->resident_used = true;
    return resident_chain_launch(ctx, s, P, n_steps);
  }
  if (cgnr_use_pipeline(s)) {
    // iteration k = K_A (applies update k-1 in its prologue, then one pass over A) + K_R; the last
    // update of this call is applied by K_F, which also returns r, p to the caller's vectors
    rls_cgnr_pipe P = cgnr_pipe_desc(s);
    const int32_t dtype = s->op->dtype;
    if (s->graph.steps && s->graph.mode != 1) {  // graph captured for the other kernel sequence
      hipGraphExecDestroy(s->graph.exec);
      s->graph = step_graph();
    }
    s->graph.mode = 1;
    int k = 0;
    RLS_TRY(run_steps(ctx, &s->graph, n_steps, [ctx, dtype, &P, &k]() {
      P.cur_hint = pipe_cur_hint(ctx, k++);
      return rls_cgnr_pipe_iteration(ctx, dtype, P);
    }, [&k](int c) { k -= c; }));
    if (s->nrhs == 1) {
      P.mb = s->mb_arm;
      s->mb_sent = s->mb_arm.dst != nullptr;
    }
    return rls_cgnr_pipe_finish(ctx, dtype, P);
  }
  if (s->nrhs != 1) return rls_fail(ctx, RLS_E_UNSUPPORTED, "batched CGNR: fused pipeline switched off");
  if (s->graph.steps && s->graph.mode != 0) {
    hipGraphExecDestroy(s->graph.exec);
    s->graph = step_graph();
  }
  s->graph.mode = 0;
  return run_steps(ctx, &s->graph, n_steps, [s]() { return cgnr_enqueue_iteration(s); });
}

int32_t rls_cgnr_step(rls_cgnr* s, int32_t n_steps) {
  if (!s) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  if (!s->initialised) return rls_fail(ctx, RLS_E_STATE, "cgnr_step before cgnr_init");
  if (n_steps < 0) return rls_fail(ctx, RLS_E_INVALID, "cgnr_step: n_steps < 0");
  RLS_HIP(ctx, rls_enter(ctx));
  s->requested += n_steps;
  return cgnr_step_impl(s, n_steps);
}

// ---- K independent small systems in ONE launch (small.hip, rls_small_group) -----------------------------------------------------
// The distinct-A flavour of a multi-solve (docs/src/literate/howto/multi_threading.jl:8-17: one solver and one A per problem) for
// problems that each fit one CU's registers: every plan is an ordinary single-column plan on rls_cgnr_path 8; the group call
// advances them together, one workgroup per plan, optionally with their init! in the same launch.
static int32_t cgnr_group(rls_cgnr* const* plans, const void* const* b, int32_t count, float lambda, float rel_tol, int32_t iterations,
                          int32_t n_steps) {
  if (!plans || count < 1 || !plans[0]) return RLS_E_INVALID;
  rls_ctx* ctx = plans[0]->op->ctx;
  if (n_steps < 0) return rls_fail(ctx, RLS_E_INVALID, "cgnr group: n_steps < 0");
  const int32_t dtype = plans[0]->op->dtype;
  RLS_HIP(ctx, rls_enter(ctx));
  for (int32_t k = 0; k < count; ++k) {
    rls_cgnr* s = plans[k];
    if (!s || s->op->ctx != ctx || s->op->dtype != dtype) return rls_fail(ctx, RLS_E_INVALID, "cgnr group: plans of one context and one element type");
    if (!cgnr_use_small(s)) return rls_fail(ctx, RLS_E_UNSUPPORTED, "cgnr group: a plan is not on the small-system path (rls_cgnr_path 8)");
    if (!b && !s->initialised) return rls_fail(ctx, RLS_E_STATE, "cgnr group: step before init");
    if (b && !b[k]) return rls_fail(ctx, RLS_E_INVALID, "cgnr group: null right-hand side");
  }
  for (int32_t k0 = 0; k0 < count; k0 += RLS_SMALL_GROUP_MAX) {
    rls_small_group G;
    G.count = count - k0 < RLS_SMALL_GROUP_MAX ? count - k0 : RLS_SMALL_GROUP_MAX;
    for (int j = 0; j < G.count; ++j) {
      rls_cgnr* s = plans[k0 + j];
      rls_small& D = G.d[j];
      D.A = s->op->A;
      D.lda = s->op->lda;
      D.M = s->op->M;
      D.N = s->op->N;
      D.x = s->x;
      D.r = s->r;
      D.p = s->p;
      D.v = s->v;
      D.sc = s->sc;
      if (b) {
        D.b = b[k0 + j];
        D.lambda = lambda;
        D.rel_tol = rel_tol;
        D.max_iter = cgnr_effective_iterations(s, iterations);
        s->sc_h->lambda = lambda;
        s->sc_h->rel_tol = rel_tol;
        s->sc_h->max_iter = D.max_iter;
        s->initialised = true;
        s->requested = 0;
      }
      s->requested += n_steps;
    }
    if (b || n_steps > 0) RLS_TRY(rls_small_group_launch(ctx, dtype, G, n_steps));
  }
  return 0;
}

// the statuses of a group in ONE read-back (one publishing kernel + one host spin per 24 plans instead of one per plan)
int32_t rls_cgnr_get_status_group(rls_cgnr* const* plans, int32_t count, rls_cgnr_status* out) {
  if (!plans || !out || count < 1 || !plans[0]) return RLS_E_INVALID;
  rls_ctx* ctx = plans[0]->op->ctx;
  RLS_HIP(ctx, rls_enter(ctx));
  for (int32_t k = 0; k < count; ++k) {
    rls_cgnr* s = plans[k];
    if (!s || s->op->ctx != ctx || s->nrhs != 1) return rls_fail(ctx, RLS_E_INVALID, "cgnr status group: single-column plans of one context");
    if (!s->initialised) return rls_fail(ctx, RLS_E_STATE, "cgnr_get_status before cgnr_init");
    if (s->resident_used) return rls_fail(ctx, RLS_E_UNSUPPORTED, "cgnr status group: a plan has resident launches to account for (use rls_cgnr_get_status)");
  }
  for (int32_t k0 = 0; k0 < count; k0 += RLS_FETCH_MAX) {
    const int32_t k1 = k0 + RLS_FETCH_MAX < count ? k0 + RLS_FETCH_MAX : count;
    for (int32_t k = k0; k < k1; ++k) RLS_TRY(rls_fetch_add(ctx, plans[k]->sc, plans[k]->sc_h, sizeof(cgnr_scalars)));
    RLS_TRY(rls_fetch_wait(ctx));
  }
  for (int32_t k = 0; k < count; ++k) cgnr_status_out(plans[k], *plans[k]->sc_h, out + k);
  return 0;
}

// ---- the distinct-A multi-solve as a queue (docs/src/literate/howto/multi_threading.jl:8-17: a solver AND an operator per task) ----
// `count` independent problems, each an ordinary single right-hand-side plan on its own operator (any shape, any kernel path; one
// context, hence one stream): problem k's init! (r = A_k^H b_k, x = 0, p = r: src/CGNR.jl:107-130) and all of its iterations
// (:143-178) are enqueued behind problem k - 1's, nothing is waited for in between, and ONE read-back at the end brings every
// problem's status.  A resident launch that was lost (its grid not on the chip in time) is re-run on the per-iteration pipeline
// exactly as rls_cgnr_get_status does it, for that problem alone.  The register-resident kernels need the whole chip each, so the
// problems run one after the other on the device -- what the queue removes is the host between them: per problem, plan creation,
// upload, a synchronising status call and a download (measured: 210 us of host per 320 us of kernels, round 5).
static int32_t cgnr_queue_validate(rls_cgnr* const* plans, int32_t count, rls_ctx** ctx_out) {
  if (!plans || count < 1 || !plans[0]) return RLS_E_INVALID;
  rls_ctx* ctx = plans[0]->op->ctx;
  for (int32_t k = 0; k < count; ++k) {
    const rls_cgnr* s = plans[k];
    if (!s || s->op->ctx != ctx || s->nrhs != 1)
      return rls_fail(ctx, RLS_E_INVALID, "cgnr solve queue: single right-hand-side plans of one context");
    for (int32_t j = 0; j < k; ++j)
      if (plans[j] == s) return rls_fail(ctx, RLS_E_INVALID, "cgnr solve queue: a plan appears twice (one plan per problem)");
  }
  *ctx_out = ctx;
  return 0;
}

// every plan's scalars (and, after resident launches, its sync block's flags) in as few publishing launches as the mailbox holds
static int32_t cgnr_queue_statuses(rls_ctx* ctx, rls_cgnr* const* plans, int32_t count, rls_cgnr_status* out) {
  constexpr int32_t PER = RLS_FETCH_MAX / 2;  // two read-backs per plan at most
  for (int32_t k0 = 0; k0 < count; k0 += PER) {
    const int32_t k1 = k0 + PER < count ? k0 + PER : count;
    for (int32_t k = k0; k < k1; ++k) {
      rls_cgnr* s = plans[k];
      if (s->resident_used) RLS_TRY(resident_fetch_flags(ctx, s->rsync, s->rsync_h));
      RLS_TRY(rls_fetch_add(ctx, s->sc, s->sc_h, sizeof(cgnr_scalars)));
    }
    RLS_TRY(rls_fetch_wait(ctx));
  }
  for (int32_t k = 0; k < count; ++k) {
    rls_cgnr* s = plans[k];
    if (s->resident_used && resident_lost(ctx, s->rsync, s->rsync_h, &s->resident_off, &s->fallbacks)) {
      const long long missing = s->requested - (long long)s->sc_h->iteration;
      if (!s->sc_h->done && missing > 0) {
        RLS_TRY(cgnr_step_impl(s, (int32_t)(missing > 0x7fffffff ? 0x7fffffff : missing)));
        RLS_TRY(fetch_scalars(ctx, s->sc, s->sc_h));
      }
    }
    s->resident_used = false;
    if (out) cgnr_status_out(s, *s->sc_h, out + k);
  }
  return 0;
}

int32_t rls_cgnr_solve_queue(rls_cgnr* const* plans, const void* const* b, int32_t count, float lambda, float rel_tol,
                             int32_t iterations, rls_cgnr_status* out) {
  rls_ctx* ctx = nullptr;
  RLS_TRY(cgnr_queue_validate(plans, count, &ctx));
  if (!b || iterations < 0) return rls_fail(ctx, RLS_E_INVALID, "cgnr solve queue: bad argument");
  for (int32_t k = 0; k < count; ++k) {
    RLS_TRY(rls_cgnr_init(plans[k], b[k], lambda, rel_tol, iterations));
    RLS_TRY(rls_cgnr_step(plans[k], iterations));
  }
  return cgnr_queue_statuses(ctx, plans, count, out);
}

// the same with b and x in HOST memory (the shape of the reference's task: host arrays in, host array out): b_k is staged through
// pinned memory and uploaded on the stream ahead of problem k's init, x_k is downloaded into pinned memory behind its last
// iteration -- every copy asynchronous, one synchronisation for the whole queue -- and handed to x_h[k] at the end.
int32_t rls_cgnr_solve_queue_host(rls_cgnr* const* plans, const void* const* b_h, void* const* x_h, int32_t count, float lambda,
                                  float rel_tol, int32_t iterations, rls_cgnr_status* out) {
  rls_ctx* ctx = nullptr;
  RLS_TRY(cgnr_queue_validate(plans, count, &ctx));
  if (!b_h || !x_h || iterations < 0) return rls_fail(ctx, RLS_E_INVALID, "cgnr solve queue: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  for (int32_t k = 0; k < count; ++k) {
    rls_cgnr* s = plans[k];
    const rls_operator* op = s->op;
    const size_t es = rls_elem_size(op->dtype);
    const size_t bb = (size_t)(op->A ? op->M : op->N) * es, xb = (size_t)op->N * es;
    if (!b_h[k] || !x_h[k]) return rls_fail(ctx, RLS_E_INVALID, "cgnr solve queue: null buffer");
    if (!s->q_b) {
      rls_alloc_scope alloc_scope(ctx);
      RLS_HIP(ctx, dmalloc(&s->q_b, bb));
      RLS_HIP(ctx, hmalloc(&s->q_bh, bb));
      RLS_HIP(ctx, hmalloc(&s->q_xh, xb));
    }
    memcpy(s->q_bh, b_h[k], bb);
    RLS_HIP(ctx, hipMemcpyAsync(s->q_b, s->q_bh, bb, hipMemcpyHostToDevice, ctx->stream));
    RLS_TRY(rls_cgnr_init(s, s->q_b, lambda, rel_tol, iterations));
    RLS_TRY(rls_cgnr_step(s, iterations));
    RLS_HIP(ctx, hipMemcpyAsync(s->q_xh, s->x, xb, hipMemcpyDeviceToHost, ctx->stream));
  }
  RLS_TRY(cgnr_queue_statuses(ctx, plans, count, out));
  // a problem whose resident launch was lost has been re-run behind the queued download of its x: fetch that one again
  bool any = false;
  for (int32_t k = 0; k < count; ++k) {
    rls_cgnr* s = plans[k];
    if (s->resident_off && s->fallbacks > 0) {
      RLS_HIP(ctx, hipMemcpyAsync(s->q_xh, s->x, (size_t)s->op->N * rls_elem_size(s->op->dtype), hipMemcpyDeviceToHost, ctx->stream));
      any = true;
    }
  }
  if (any) RLS_HIP(ctx, rls_stream_wait(ctx->stream));
  for (int32_t k = 0; k < count; ++k) memcpy(x_h[k], plans[k]->q_xh, (size_t)plans[k]->op->N * rls_elem_size(plans[k]->op->dtype));
  return 0;
}

int32_t rls_cgnr_step_group(rls_cgnr* const* plans, int32_t count, int32_t n_steps) {
  return cgnr_group(plans, nullptr, count, 0.f, 0.f, 0, n_steps);
}

int32_t rls_cgnr_init_step_group(rls_cgnr* const* plans, const void* const* b, int32_t count, float lambda, float rel_tol,
                                 int32_t iterations, int32_t n_steps) {
  if (!b) return RLS_E_INVALID;
  return cgnr_group(plans, b, count, lambda, rel_tol, iterations, n_steps);
}

// measurement only: the two kernels of the fused pipeline timed separately.  Each is idempotent
// when repeated (K_A reads the committed scalars and writes the staged ones, K_R the reverse), so
// after one ordinary iteration the normal-operator kernel is launched n_steps times back to back
// between two hipEvents, then the reduce kernel likewise; the averages are what rocprofv3 reports
// for back-to-back dispatches.  The solver state afterwards is that of ONE more iteration.
int32_t rls_cgnr_step_profiled(rls_cgnr* s, int32_t n_steps, float* us_normal, float* us_reduce) {
  if (!s || !us_normal || !us_reduce || n_steps <= 0) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  if (!s->initialised) return rls_fail(ctx, RLS_E_STATE, "cgnr_step_profiled before cgnr_init");
  if (!cgnr_use_pipeline(s)) return rls_fail(ctx, RLS_E_UNSUPPORTED, "cgnr_step_profiled: fused pipeline not active");
  RLS_HIP(ctx, rls_enter(ctx));
  rls_cgnr_pipe P = cgnr_pipe_desc(s);
  const int32_t dtype = s->op->dtype;
  RLS_TRY(rls_cgnr_pipe_iteration(ctx, dtype, P));  // leaves an update pending: K_A then does full work
  P.cur_hint = 0;  // ... on pair 0, every time (the reduce kernel that would commit the flip is not run in between)
  hipEvent_t ev[3];
  for (auto& e : ev) RLS_HIP(ctx, hipEventCreate(&e));
  int32_t st = 0;
  RLS_HIP(ctx, hipEventRecord(ev[0], ctx->stream));
  for (int i = 0; i < n_steps && st == 0; ++i) st = rls_cgnr_pipe_launch(ctx, dtype, P, 1);
  RLS_HIP(ctx, hipEventRecord(ev[1], ctx->stream));
  for (int i = 0; i < n_steps && st == 0; ++i) st = rls_cgnr_pipe_launch(ctx, dtype, P, 2);
  RLS_HIP(ctx, hipEventRecord(ev[2], ctx->stream));
  RLS_HIP(ctx, rls_event_wait(ev[2]));
  float a = 0.f, r = 0.f;
  RLS_HIP(ctx, hipEventElapsedTime(&a, ev[0], ev[1]));
  RLS_HIP(ctx, hipEventElapsedTime(&r, ev[1], ev[2]));
  for (auto& e : ev) hipEventDestroy(e);
  if (st != 0) return st;
  RLS_TRY(rls_cgnr_pipe_finish(ctx, dtype, P));
  *us_normal = 1e3f * a / n_steps;
  *us_reduce = 1e3f * r / n_steps;
  return 0;
}

// ---- row-partitioned solvers through a communicator (comm.hip): the collective lives inside the library ----------------
// Every loop below is a list of PHASES run by rls_comm_run: each rank's worker thread walks the whole list for its rank
// (src/MultiThreading.jl:60-78's Threads.@threads, inside the library), meeting the others at a host barrier wherever
// one rank's stream must not wait on an event another rank has not recorded yet -- i.e. between "publish" and "collect"
// of an all-reduce round.  The host side of an iteration therefore costs one rank's launches, not n ranks'.
int32_t rls_cgnr_init_rowsharded(rls_comm* comm, rls_cgnr* const* plans, const void* const* b_parts, float lambda,
                                 float rel_tol, int32_t iterations) {
  RLS_TRY(rowsharded_check<rls_cgnr>(comm, plans, cgnr_op, cgnr_single));
  if (!b_parts) return RLS_E_INVALID;
  const int64_t N = plans[0]->op->N;
  const int32_t dtype = plans[0]->op->dtype;
  int round0 = 0;
  RLS_TRY(rls_comm_next_rounds(comm, 1, N, dtype, &round0));
  const std::vector<rls_comm_phase> phases = {
      {[&](int r, int) {  // r_g = A_g^H b_g
         RLS_TRY(rls_cgnr_init_local_a(plans[r], b_parts[r], lambda, rel_tol, iterations));
         return rls_comm_publish(comm, r, plans[r]->r, N, dtype, round0);
       }, true},
      {[&](int r, int) {  // r = sum_g r_g, then the replicated rest of initCGNR
         RLS_TRY(rls_comm_collect(comm, r, plans[r]->r, N, dtype, round0));
         return rls_cgnr_init_local_b(plans[r]);
       }, false}};
  return rls_comm_run(comm, phases, 1);
}

int32_t rls_cgnr_step_rowsharded(rls_comm* comm, rls_cgnr* const* plans, int32_t n_steps) {
  RLS_TRY(rowsharded_check<rls_cgnr>(comm, plans, cgnr_op, cgnr_single));
  if (n_steps < 0) return RLS_E_INVALID;
  const int64_t N = plans[0]->op->N;
  const int32_t dtype = plans[0]->op->dtype;
  int round0 = 0;
  RLS_TRY(rls_comm_next_rounds(comm, n_steps, N, dtype, &round0));
  const std::vector<rls_comm_phase> phases = {
      {[&](int r, int k) {  // t_g = A_g p, v_g = A_g^H t_g
         RLS_TRY(rls_cgnr_step_local_a(plans[r]));
         return rls_comm_publish(comm, r, plans[r]->v, N, dtype, round0 + k);
       }, true},
      {[&](int r, int k) {  // v = sum_g v_g, the replicated update.  No barrier behind it: a rank can run at most one round
         RLS_TRY(rls_comm_collect(comm, r, plans[r]->v, N, dtype, round0 + k));  // ahead, and the rounds alternate buffers
         return rls_cgnr_step_local_b(plans[r]);
       }, false}};
  return rls_comm_run(comm, phases, n_steps);
}

// FISTA on a row-partitioned A (src/FISTA.jl:110-185): x0 = sum_g A_g^H b_g at init, res_g = A_g^H A_g y per iteration;
// gradient step, prox, momentum and `done` replicated.  plans[r]: rls_fista_create on rank r's shard operator (+ set_reg).
int32_t rls_fista_init_rowsharded(rls_comm* comm, rls_fista* const* plans, const void* const* b_parts, float rho, float theta,
                                  float rel_tol, int32_t iterations, int32_t restart_gradient) {
  RLS_TRY(rowsharded_check<rls_fista>(comm, plans, fista_op, fista_single));
  if (!b_parts) return RLS_E_INVALID;
  const int64_t N = plans[0]->op->N;
  const int32_t dtype = plans[0]->op->dtype;
  int round0 = 0;
  RLS_TRY(rls_comm_next_rounds(comm, 1, N, dtype, &round0));
  const std::vector<rls_comm_phase> phases = {
      {[&](int r, int) {
         RLS_TRY(rls_fista_init_local_a(plans[r], b_parts[r]));
         return rls_comm_publish(comm, r, plans[r]->x0, N, dtype, round0);
       }, true},
      {[&](int r, int) {
         RLS_TRY(rls_comm_collect(comm, r, plans[r]->x0, N, dtype, round0));
         return rls_fista_init_local_b(plans[r], rho, theta, rel_tol, iterations, restart_gradient);
       }, false}};
  return rls_comm_run(comm, phases, 1);
}

int32_t rls_fista_step_rowsharded(rls_comm* comm, rls_fista* const* plans, int32_t n_steps) {
  RLS_TRY(rowsharded_check<rls_fista>(comm, plans, fista_op, fista_single));
  if (n_steps < 0) return RLS_E_INVALID;
  const int64_t N = plans[0]->op->N;
  const int32_t dtype = plans[0]->op->dtype;
  int round0 = 0;
  RLS_TRY(rls_comm_next_rounds(comm, n_steps, N, dtype, &round0));
  const std::vector<rls_comm_phase> phases = {
      {[&](int r, int k) {
         RLS_TRY(rls_fista_step_local_a(plans[r]));
         return rls_comm_publish(comm, r, plans[r]->res, N, dtype, round0 + k);
       }, true},
      {[&](int r, int k) {
         RLS_TRY(rls_comm_collect(comm, r, plans[r]->res, N, dtype, round0 + k));
         return rls_fista_step_local_b(plans[r]);
       }, false}};
  return rls_comm_run(comm, phases, n_steps);
}

int32_t rls_cgnr_path(rls_cgnr* s, int32_t* out) {
  if (!s || !out) return RLS_E_INVALID;
  *out = cgnr_use_small(s) ? 8 : s->skinny ? (cgnr_use_gramk(s, 0) ? 7 : s->op->G ? 6 : 3) : cgnr_use_gram_resident(s) ? 5 : cgnr_use_gram_pipeline(s) ? 2 : cgnr_use_resident(s) ? 4 : cgnr_use_pipeline(s) ? 1 : 0;
  return 0;
}

int32_t rls_cgnr_step_local_a(rls_cgnr* s) {
  if (!s) return RLS_E_INVALID;
  if (!s->initialised) return rls_fail(s->op->ctx, RLS_E_STATE, "cgnr_step before cgnr_init");
  RLS_HIP(s->op->ctx, hipSetDevice(s->op->ctx->device));
  return op_normal(s->op, s->p, s->v, &s->sc->done);
}
int32_t rls_cgnr_step_local_b(rls_cgnr* s) {
  if (!s) return RLS_E_INVALID;
  if (!s->initialised) return rls_fail(s->op->ctx, RLS_E_STATE, "cgnr_step before cgnr_init");
  RLS_HIP(s->op->ctx, hipSetDevice(s->op->ctx->device));
  return cgnr_enqueue_update(s);
}

// status read-back of a single-RHS plan; re-runs on the per-iteration pipeline whatever a lost resident launch left undone
static int32_t cgnr_fetch_status(rls_cgnr* s) {
  rls_ctx* ctx = s->op->ctx;
  if (s->resident_used) RLS_TRY(resident_fetch_flags(ctx, s->rsync, s->rsync_h));
  RLS_TRY(fetch_scalars(ctx, s->sc, s->sc_h));
  if (s->resident_used && resident_lost(ctx, s->rsync, s->rsync_h, &s->resident_off, &s->fallbacks)) {
    const long long missing = s->requested - (long long)s->sc_h->iteration;
    if (!s->sc_h->done && missing > 0) {
      RLS_TRY(cgnr_step_impl(s, (int32_t)(missing > 0x7fffffff ? 0x7fffffff : missing)));
      RLS_TRY(fetch_scalars(ctx, s->sc, s->sc_h));
    }
  }
  s->resident_used = false;  // every resident launch up to here is accounted for (the lost count is sticky until it is read)
  return 0;
}

int32_t rls_cgnr_get_status(rls_cgnr* s, rls_cgnr_status* out) {
  if (!s || !out) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  if (!s->initialised) return rls_fail(ctx, RLS_E_STATE, "cgnr_get_status before cgnr_init");
  if (ctx->server == &s->srv && s->srv.alive && s->srv.fresh) {  // a kernel left listening: the mirror holds its last command's status
    cgnr_status_out(s, *s->sc_h, out);
    return 0;
  }
  RLS_HIP(ctx, rls_enter(ctx));
  RLS_TRY(cgnr_fetch_status(s));
  cgnr_status_out(s, *s->sc_h, out);
  return 0;
}

static void cgnr_status_out(const rls_cgnr* s, const cgnr_scalars& h, rls_cgnr_status* out) {
  out->iteration = h.iteration;
  out->done = h.done;
  out->alpha_re = (float)h.alpha_re;
  out->alpha_im = (float)h.alpha_im;
  out->beta_re = (float)h.beta_re;
  out->beta_im = (float)h.beta_im;
  out->zeta = (float)h.zeta;
  out->residual = (float)sqrt(h.rr);
  out->z0 = (float)h.z0;
  out->fallbacks = s->fallbacks;
}

// ---- resident kernels in server mode (rls_cg_start::srv_ctl, rls_srv_args) ---------------------------------------------------------
// rls_cgnr_step_status / rls_fista_step_status on a plan whose A lives in the register files do not let the kernel end: the next
// call posts {n_steps, mailbox sequence, command sequence} into the pinned control block the kernel listens on and spins on the
// mailbox -- no launch, no load of A per call.  Everything else that wants the stream sends EXIT first (rls_enter -> rls_server_stop).
extern "C++" {
static bool server_usable(const rls_ctx* ctx, const srv_state* v) {
  return ctx->tune.resident_server && ctx->tune.status_mailbox && v->ctl && !v->off && (ctx->server == nullptr || ctx->server == v);
}
static bool cgnr_use_server(const rls_cgnr* s) {
  return server_usable(s->op->ctx, &s->srv) &&
         (cgnr_use_small(s) || cgnr_use_resident(s) || (cgnr_use_gram_resident(s) && s->nrhs == 1));
}

// the life of a listening kernel is over (it was told to leave, left idle, or gave up): bookkeeping, and the verdict on lives
// too short to pay for their launch
static void server_life_over(rls_ctx* ctx, srv_state* v) {
  const bool gave_up = v->ctl[17] == 2;
  if (v->alive) {
    if (v->served < 3) {
      if (++v->short_lives >= 2) v->off = true;
    } else {
      v->short_lives = 0;
    }
  }
  v->alive = false;
  v->fresh = false;
  if (gave_up && v->resident_used) *v->resident_used = true;  // a wait ran out in that life: the next status call reads the sync block's flags
  if (ctx->server == v) ctx->server = nullptr;
}

void rls_server_stop(rls_ctx* ctx) {
  srv_state* v = static_cast<srv_state*>(ctx->server);
  if (!v) return;
  if (v->alive) {
    volatile unsigned* ctl = v->ctl;
    if (!ctl[17]) {
      ctl[1] = RLS_SRV_EXIT;
      std::atomic_thread_fence(std::memory_order_release);
      ctl[0] = ++v->seq;
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned n = 0; !ctl[17]; ++n) {
        rls_cpu_relax();
        if ((n & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(500)) {
          (void)hipSetDevice(ctx->device);
          (void)hipStreamSynchronize(ctx->stream);  // (its idle timeout or its wait bounds end it at the latest)
          break;
        }
      }
    }
  }
  server_life_over(ctx, v);
}

// 0: the command was served (status in the mirror); 1: the kernel had left before it saw the command; 2: it gave up inside it
static int server_wait(rls_ctx* ctx, srv_state* v, unsigned mbseq) {
  volatile unsigned* mb = ctx->mb_h;
  volatile unsigned* ctl = v->ctl;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned n = 0;; ++n) {
    if (*mb == mbseq) break;
    const unsigned ex = ctl[17];
    if (ex) {  // (the status of a served command is published before the kernel can leave: look once more)
      std::atomic_thread_fence(std::memory_order_acquire);
      if (*mb == mbseq) break;
      return (int)ex;
    }
    rls_cpu_relax();
    if ((n & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
      (void)hipStreamSynchronize(ctx->stream);
      if (*mb == mbseq) break;
      return ctl[17] == 1 ? 1 : 2;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return 0;
}

// One command: posted to the listening kernel, or carried by a launch (`launch(args)`, through the resident chain).
// Returns 0 = served (mirror current), 1 = nothing ran and the plan should take its ordinary path, 2 = a launch gave up inside
// the command (the caller's lost-launch recovery re-runs it), < 0 = error.
template <typename L>
static int32_t server_command(rls_ctx* ctx, srv_state* v, void* mirror, int32_t n_steps, bool usable_again, L&& launch) {
  volatile unsigned* ctl = v->ctl;
  for (int attempt = 0; attempt < 2; ++attempt) {
    const unsigned mbseq = ++ctx->mb_seq;
    if (v->alive) {  // post the command: payload, then the sequence word
      ctl[1] = (unsigned)n_steps;
      ctl[2] = mbseq;
      std::atomic_thread_fence(std::memory_order_release);
      ctl[0] = ++v->seq;
      std::atomic_thread_fence(std::memory_order_seq_cst);
    } else {
      ctl[16] = ctl[17] = 0;
      ctl[0] = v->seq;
      rls_srv_args a;
      a.ctl = v->ctl;
      a.seq0 = v->seq;
      a.idle_us = (unsigned)(ctx->tune.resident_server_idle_us > 0 ? ctx->tune.resident_server_idle_us : 1);
      if (ctx->tune.resident_ahead) a.idle_us |= RLS_SRV_AHEAD;  // (read by the single-workgroup kernels; the resident launches below mask it)
      a.mb.dst = mirror;
      a.mb.seq_h = ctx->mb_h;
      a.mb.seq = mbseq;
      const int32_t st = launch(a);
      if (st != 0) return st < 0 ? st : -1;
      v->alive = true;
      v->served = 0;
      ctx->server = v;
    }
    const int r = server_wait(ctx, v, mbseq);
    if (r == 0) {
      v->served += 1;
      v->fresh = true;
      return 0;
    }
    server_life_over(ctx, v);
    if (r == 2) {
      if (v->resident_used) *v->resident_used = true;  // (the status call that follows reads the sync block's flags)
      return 2;
    }
    // r == 1: it had left (idle) before it saw the command -- nothing ran.  Once more with a launch, or -- this plan's lives
    // keep ending early -- on the ordinary path
    if (attempt == 1 || !usable_again || v->off) {
      v->off = true;
      return 1;
    }
  }
  return 1;
}

extern "C" int32_t rls_cgnr_step_status(rls_cgnr* s, int32_t n_steps, rls_cgnr_status* out);
static int32_t cgnr_step_status_server(rls_cgnr* s, int32_t n_steps, rls_cgnr_status* out) {
  rls_ctx* ctx = s->op->ctx;
  RLS_HIP(ctx, hipSetDevice(s->device));
  s->requested += n_steps;
  const int32_t r = server_command(ctx, &s->srv, s->sc_h, n_steps, true, [&](const rls_srv_args& a) {
    rls_cg_start St;
    St.srv_ctl = a.ctl;
    St.srv_seq0 = a.seq0;
    St.srv_idle_us = a.idle_us & ~RLS_SRV_AHEAD;
    St.srv_mb = a.mb;
    if (cgnr_use_small(s)) {  // the single-workgroup kernel: one CU stays, nothing to chain
      rls_small D;
      D.A = s->op->A;
      D.lda = s->op->lda;
      D.M = s->op->M;
      D.N = s->op->N;
      D.x = s->x;
      D.r = s->r;
      D.p = s->p;
      D.v = s->v;
      D.sc = s->sc;
      D.mb = a.mb;
      D.srv = a;
      return rls_small_launch(ctx, s->op->dtype, D, n_steps);
    }
    if (cgnr_use_gram_resident(s)) {  // AHA explicit, held in the register files
      const rls_gram_pipe G = cgnr_gram_desc(s);
      return resident_chain(ctx, s->rsync, [&]() {
        return rls_gram_resident_launch(ctx, s->op->dtype, G, s->rsync, n_steps, (unsigned)ctx->tune.resident_spin, St);
      }, &s->rsync_clean);
    }
    const rls_cgnr_pipe P = cgnr_pipe_desc(s);
    return resident_chain(ctx, s->rsync, [&]() {
      return rls_cgnr_resident_launch(ctx, s->op->dtype, P, s->rdots, s->rsync, n_steps, (unsigned)ctx->tune.resident_spin, St);
    }, &s->rsync_clean);
  });
  if (r < 0) return r;
  if (r == 0) {
    cgnr_status_out(s, *s->sc_h, out);
    return 0;
  }
  if (r == 1) {  // nothing ran: the ordinary path
    s->requested -= n_steps;
    return rls_cgnr_step_status(s, n_steps, out);
  }
  return rls_cgnr_get_status(s, out);  // (reads the sync block's flags: iterations a lost launch did not run are re-run here)
}
}  // extern "C++"

// One iterate per call is the reference's solve! loop with callbacks (src/RegularizedLeastSquares.jl:161-176): step and read-back
// as ONE entry point, and on the per-iteration pipeline and the small-system kernel the call's last kernel stores the scalars into
// the plan's pinned mirror itself -- no publishing launch behind it.
int32_t rls_cgnr_step_status(rls_cgnr* s, int32_t n_steps, rls_cgnr_status* out) {
  if (!s || !out) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  // (whole-solve calls gain nothing from a kernel that stays: server mode is for the per-iterate calls of a solve! loop with callbacks)
  if (s->initialised && n_steps > 0 && n_steps <= 8 && !s->resident_used && cgnr_use_server(s)) return cgnr_step_status_server(s, n_steps, out);
  if (s->initialised && s->nrhs == 1 && n_steps > 0 && !s->resident_used) {
    RLS_HIP(ctx, rls_enter(ctx));
    s->mb_arm = rls_mailbox_arm(ctx, s->sc_h);
  }
  s->mb_sent = false;
  const int32_t st = rls_cgnr_step(s, n_steps);
  const rls_mailbox_slot mb = s->mb_arm;
  s->mb_arm = rls_mailbox_slot();
  if (st != 0) return st;
  if (!s->mb_sent) return rls_cgnr_get_status(s, out);
  RLS_TRY(rls_mailbox_wait(ctx, mb.seq));
  cgnr_status_out(s, *s->sc_h, out);
  return 0;
}

// ---- FISTA ----------------------------------------------------------------------------------
int32_t rls_fista_create(rls_operator* op, void* x, void* x0, void* xold, void* res, rls_fista** out) {
  if (!op) return RLS_E_INVALID;
  rls_ctx* ctx = op->ctx;
  if (!x || !x0 || !xold || !res || !out) return rls_fail(ctx, RLS_E_INVALID, "fista_create: null pointer");
  RLS_HIP(ctx, rls_enter(ctx));
  rls_alloc_scope alloc_scope(ctx);
  rls_fista* s = new rls_fista();
  s->op = op;
  s->actx = ctx;
  s->actx_id = ctx->id;
  s->device = op->ctx->device;
  s->buf[0] = x;
  s->buf[1] = xold;
  s->x0 = x0;
  s->res = res;
  s->y = nullptr;
  s->reg_kind = RLS_REG_L1;
  s->proj_kind = RLS_PROJ_NONE;
  s->lambda = 0.f;
  s->l21_slices = 1;
  s->initialised = false;
  s->y1 = s->res_raw = nullptr;
  s->scn = nullptr;
  s->use_pipe = false;
  s->res_raw1 = nullptr;
  s->use_gram = false;
  s->small = op->A && !op->G && rls_small_ok(op->dtype, op->M, op->N, op->A, op->lda);
  if (s->small && hmalloc(&s->srv.ctl, 32 * sizeof(unsigned)) == hipSuccess) memset(s->srv.ctl, 0, 32 * sizeof(unsigned));  // (server mode of the single-workgroup kernel)
  const size_t vb = (size_t)op->N * rls_elem_size(op->dtype);
  hipError_t e = dmalloc(&s->y, vb);
  const bool gram = op->G && rls_gram_pipe_ok(op->dtype, op->N, op->G, op->ldg);
  if (e == hipSuccess && gram) {
    e = dmalloc(&s->res_raw1, vb);
    if (e == hipSuccess) e = hipMemsetAsync(s->res_raw1, 0, vb, ctx->stream);
  }
  if (e == hipSuccess && (op->slab || gram)) {
    e = dmalloc(&s->y1, vb);
    if (e == hipSuccess) e = dmalloc(&s->res_raw, 2 * vb);  // (second half: state.res of the iteration a listening kernel runs ahead, fista_resident_kernel's SPEC)
    if (e == hipSuccess) e = dmalloc(&s->scn, sizeof(fista_scalars));
    if (e == hipSuccess) e = hipMemsetAsync(s->y1, 0, vb, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->res_raw, 0, 2 * vb, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->scn, 0, sizeof(fista_scalars), ctx->stream);
  }
  if (e != hipSuccess) {
    if (s->y) dfree(s->y);
    if (s->y1) dfree(s->y1);
    if (s->res_raw) dfree(s->res_raw);
    if (s->res_raw1) dfree(s->res_raw1);
    if (s->scn) dfree(s->scn);
    delete s;
    return rls_fail(ctx, (int32_t)e, "fista_create: hipMalloc failed");
  }
  if ((op->slab && op->A && !op->G && rls_cgnr_resident_ok(ctx, op->dtype, op->M, op->N, op->A, op->lda)) ||
      (gram && rls_gram_resident_ok(ctx, op->dtype, op->N, op->G, op->ldg))) {
    if (resident_alloc(ctx, op, &s->rsync, &s->rsync_h) != hipSuccess) {
      if (s->rsync) dfree(s->rsync);
      s->rsync = nullptr;  // resident mode is an optimisation: without its scratch the pipeline runs
      (void)hipGetLastError();
    } else if ((!gram || rls_gram_resident_server_ok(op->dtype, op->N)) &&
               (s->srv.ctl || hmalloc(&s->srv.ctl, 32 * sizeof(unsigned)) == hipSuccess)) {  // (both resident kernels can stay and listen)
      memset(s->srv.ctl, 0, 32 * sizeof(unsigned));
      s->srv.resident_used = &s->resident_used;
    }
  }
  int32_t st = alloc_scalars(ctx, &s->sc, &s->sc_h);
  if (st != 0) {
    dfree(s->y);
    if (s->y1) dfree(s->y1);
    if (s->res_raw) dfree(s->res_raw);
    if (s->res_raw1) dfree(s->res_raw1);
    if (s->scn) dfree(s->scn);
    delete s;
    return st;
  }
  *out = s;
  return 0;
}

int32_t rls_fista_destroy(rls_fista* s) {
  if (!s) return RLS_E_INVALID;
  if (rls_ctx_alive(s->actx, s->actx_id) && s->actx->server == &s->srv) rls_server_stop(s->actx);  // (a kernel of this plan left listening)
  hipSetDevice(s->device);
  if (s->srv.ctl) hfree(s->srv.ctl);
  rls_alloc_scope alloc_scope(alloc_ctx_of(s->actx, s->actx_id));
  if (s->graph.exec) hipGraphExecDestroy(s->graph.exec);
  dfree(s->y);
  if (s->y1) dfree(s->y1);
  if (s->res_raw) dfree(s->res_raw);
  if (s->res_raw1) dfree(s->res_raw1);
  if (s->scn) dfree(s->scn);
  if (s->Ypack) dfree(s->Ypack);
  if (s->Tpack) dfree(s->Tpack);
  if (s->Vpart) dfree(s->Vpart);
  if (s->scb_h) hfree(s->scb_h);
  if (s->rsync) dfree(s->rsync);
  if (s->rsync_h) hfree(s->rsync_h);
  if (s->fk_yx) dfree(s->fk_yx);
  if (s->fk_xx) dfree(s->fk_xx);
  if (s->fk_dots) dfree(s->fk_dots);
  if (s->tv_in) dfree(s->tv_in);
  if (s->tv_out) dfree(s->tv_out);
  dfree(s->sc);
  hfree(s->sc_h);
  delete s;
  return 0;
}

// A cached graph of this plan's iteration holds the kernel SEQUENCE of its regulariser and, for TV, the threshold rho * lambda, the
// image geometry and iterationsTV as kernel arguments (every other regulariser reads rho and lambda from the scalars on the device):
// whatever changes one of them drops the graph, the next step call captures afresh.
static void fista_drop_graph(rls_fista* s) {
  if (s->graph.exec) hipGraphExecDestroy(s->graph.exec);
  s->graph = step_graph();
}

int32_t rls_fista_set_reg(rls_fista* s, int32_t reg_kind, float lambda, int64_t l21_slices, int32_t proj_kind) {
  if (!s) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  if (reg_kind == RLS_REG_TV)
    return rls_fail(ctx, RLS_E_UNSUPPORTED, "fused FISTA: TV prox is not fused; drive FISTA from the primitives");
  if (reg_kind < RLS_REG_NONE || reg_kind > RLS_REG_L21 || proj_kind < RLS_PROJ_NONE || proj_kind > RLS_PROJ_POSITIVE)
    return rls_fail(ctx, RLS_E_INVALID, "fista_set_reg: unknown kind");
  if (reg_kind == RLS_REG_L21 && (l21_slices <= 0 || s->op->N / l21_slices == 0))
    return rls_fail(ctx, RLS_E_INVALID, "fista_set_reg: slices must be in 1..N");
  if (s->reg_kind == RLS_REG_TV || reg_kind != s->reg_kind) fista_drop_graph(s);  // another kernel sequence (or TV's baked arguments)
  s->reg_kind = reg_kind;
  s->proj_kind = proj_kind;
  s->lambda = lambda;
  s->l21_slices = l21_slices > 0 ? l21_slices : 1;
  return 0;
}

// FISTA with prox!(::TVRegularization) (src/FISTA.jl:164 -> src/proximalMaps/ProxTV.jl:64-125): the FGP loop is one
// single-workgroup launch between the two halves of the update, for images that fit one workgroup (rls_tv_single_ok: 1-D / 2-D
// images of up to 8192 Float32 / 4096 ComplexF32 pixels, other geometries up to 2048); single right-hand side.  The plan then
// runs on the two-product path (operator apply, update half, FGP, update half); row-sharded plans (rls_fista_step_local_b,
// rls_fista_step_rowsharded) take the same three launches behind their all-reduce.
int32_t rls_fista_set_reg_tv(rls_fista* s, float lambda, int32_t ndims, const int64_t* shape, int32_t ntv, const int32_t* dims,
                             int32_t iterations_tv, int32_t proj_kind) {
  if (!s) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  if (proj_kind < RLS_PROJ_NONE || proj_kind > RLS_PROJ_POSITIVE || iterations_tv < 0 || ndims < 1 || ndims > 4 || ntv < 0 || ntv > 4 ||
      !shape || (ntv > 0 && !dims))
    return rls_fail(ctx, RLS_E_INVALID, "fista_set_reg_tv: bad argument");
  int64_t n = 1;
  for (int k = 0; k < ndims; ++k) n *= shape[k];
  if (n != s->op->N) return rls_fail(ctx, RLS_E_INVALID, "fista_set_reg_tv: prod(shape) != N");
  if (s->nrhs != 1) return rls_fail(ctx, RLS_E_UNSUPPORTED, "fista_set_reg_tv: single right-hand side plans only");
  if (!rls_tv_single_ok(s->op->ctx, s->op->dtype, ndims, shape, ntv, dims))
    return rls_fail(ctx, RLS_E_UNSUPPORTED, "fista_set_reg_tv: the image does not fit the single-workgroup FGP kernel; drive FISTA from the primitives");
  RLS_HIP(ctx, rls_enter(ctx));
  if (!s->tv_in) {
    rls_alloc_scope alloc_scope(ctx);
    const size_t bytes = (size_t)s->op->N * rls_elem_size(s->op->dtype);
    RLS_HIP(ctx, dmalloc(&s->tv_in, bytes));
    RLS_HIP(ctx, dmalloc(&s->tv_out, bytes));
  }
  // lambda, the geometry and iterationsTV are arguments of the captured FGP launch: a cached graph survives only an identical call
  bool same = s->reg_kind == RLS_REG_TV && s->lambda == lambda && s->tv_iters == iterations_tv && s->tv_ndims == ndims &&
              s->tv_ntv == ntv && s->proj_kind == proj_kind;
  for (int k = 0; k < 4 && same; ++k)
    same = s->tv_shape[k] == (k < ndims ? shape[k] : 1) && s->tv_dims[k] == (k < ntv ? dims[k] : 0);
  if (!same) fista_drop_graph(s);
  s->tv_ndims = ndims;
  s->tv_ntv = ntv;
  for (int k = 0; k < 4; ++k) {
    s->tv_shape[k] = k < ndims ? shape[k] : 1;
    s->tv_dims[k] = k < ntv ? dims[k] : 0;
  }
  s->tv_iters = iterations_tv;
  s->reg_kind = RLS_REG_TV;
  s->proj_kind = proj_kind;
  s->lambda = lambda;
  s->l21_slices = 1;
  return 0;
}

// x0 = A^H b (src/FISTA.jl:114); on a row shard this is the partial sum the caller all-reduces
int32_t rls_fista_init_local_a(rls_fista* s, const void* b) {
  if (!s) return RLS_E_INVALID;
  rls_operator* op = s->op;
  rls_ctx* ctx = op->ctx;
  if (!b) return rls_fail(ctx, RLS_E_INVALID, "fista_init: null b");
  RLS_HIP(ctx, rls_enter(ctx));
  if (op->A)
    RLS_TRY(rls_launch_gemv(ctx, op->dtype, RLS_OP_C, op->M, op->N, 1.f, 0.f, op->A, op->lda, b, 0.f, 0.f, s->x0, nullptr));
  else
    RLS_HIP(ctx, hipMemcpyAsync(s->x0, b, (size_t)op->N * rls_elem_size(op->dtype), hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}

static int32_t fista_init_finish(rls_fista* s, float rho, float theta, float rel_tol, int32_t iterations,
                                 int32_t restart_gradient, bool local) {
  rls_operator* op = s->op;
  rls_ctx* ctx = op->ctx;
  RLS_HIP(ctx, rls_enter(ctx));
  if (op->dtype == RLS_F32)
    hipLaunchKernelGGL(fista_init_kernel<float>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (float*)s->buf[0],
                       (float*)s->buf[1], (float*)s->x0, (float*)s->res, (float*)s->y, op->N, s->sc, rho, theta,
                       rel_tol, iterations, restart_gradient, s->reg_kind, s->proj_kind, s->lambda,
                       (long long)s->l21_slices, fista_batch<float>{0, nullptr, 1, 0, nullptr, 0}, (unsigned*)s->rsync,
                       s->rsync ? (int)(rls_resident_sync_clear_bytes() / sizeof(unsigned)) : 0);
  else
    hipLaunchKernelGGL(fista_init_kernel<float2>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (float2*)s->buf[0],
                       (float2*)s->buf[1], (float2*)s->x0, (float2*)s->res, (float2*)s->y, op->N, s->sc, rho,
                       theta, rel_tol, iterations, restart_gradient, s->reg_kind, s->proj_kind, s->lambda,
                       (long long)s->l21_slices, fista_batch<float2>{0, nullptr, 1, 0, nullptr, 0}, (unsigned*)s->rsync,
                       s->rsync ? (int)(rls_resident_sync_clear_bytes() / sizeof(unsigned)) : 0);
  s->rsync_clean = s->rsync != nullptr;
  s->enq = 0;
  s->requested = 0;
  s->theta0 = theta;
  if (s->reg_kind == RLS_REG_TV) {
    if (rho != s->rho_h) fista_drop_graph(s);  // rho * lambda is an argument of the captured FGP launch
    // the single-workgroup FGP kernel was checked when the regulariser was set; its limits are context tunables that may have moved
    if (!rls_tv_single_ok(op->ctx, op->dtype, s->tv_ndims, s->tv_shape, s->tv_ntv, s->tv_dims))
      return rls_fail(ctx, RLS_E_UNSUPPORTED, "fista_init: the TV image no longer fits the single-workgroup FGP kernel");
  }
  s->rho_h = rho;
  s->srv.off = false;  // (a new solve: the caller's pattern between iterates is judged afresh)
  s->srv.short_lives = 0;
  s->initialised = true;
  // row-sharded plans exchange `res` between the operator apply and the update: two-half iterations only
  const bool gram = !local && fista_gram_ok(s);
  const bool pipe = !local && !gram && fista_pipe_ok(s);
  if ((pipe != s->use_pipe || gram != s->use_gram) && s->graph.exec) {  // captured for another kernel sequence
    hipGraphExecDestroy(s->graph.exec);
    s->graph = step_graph();
  }
  s->use_pipe = pipe;
  s->use_gram = gram;
  return launch_status(ctx);
}

int32_t rls_fista_init(rls_fista* s, const void* b, float rho, float theta, float rel_tol, int32_t iterations,
                       int32_t restart_gradient) {
  RLS_TRY(rls_fista_init_local_a(s, b));
  return fista_init_finish(s, rho, theta, rel_tol, iterations, restart_gradient, false);
}

int32_t rls_fista_init_local_b(rls_fista* s, float rho, float theta, float rel_tol, int32_t iterations,
                               int32_t restart_gradient) {
  if (!s) return RLS_E_INVALID;
  return fista_init_finish(s, rho, theta, rel_tol, iterations, restart_gradient, true);
}

// one row-sharded iteration: res_partial = A_g^H A_g y ; [caller: all-reduce(res)] ; gradient step, prox, momentum
int32_t rls_fista_step_local_a(rls_fista* s) {
  if (!s) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  if (!s->initialised || s->use_pipe || s->use_gram) return rls_fail(ctx, RLS_E_STATE, "fista_step_local before fista_init_local_b");
  RLS_HIP(ctx, rls_enter(ctx));
  return op_normal(s->op, s->y, s->res, &s->sc->done);
}
int32_t rls_fista_step_local_b(rls_fista* s) {
  if (!s) return RLS_E_INVALID;
  rls_operator* op = s->op;
  rls_ctx* ctx = op->ctx;
  if (!s->initialised || s->use_pipe || s->use_gram) return rls_fail(ctx, RLS_E_STATE, "fista_step_local before fista_init_local_b");
  RLS_HIP(ctx, rls_enter(ctx));
  if (op->dtype == RLS_F32)
    RLS_TRY(fista_launch_update<float>(s, 1, fista_batch<float>{0, nullptr, 1, 0, nullptr, 0}));
  else
    RLS_TRY(fista_launch_update<float2>(s, 1, fista_batch<float2>{0, nullptr, 1, 0, nullptr, 0}));
  return launch_status(ctx);
}

// K right-hand sides sharing A (solve!(solver, B) of src/MultiThreading.jl:30-79 for FISTA): per-column scalars,
// per-column retirement, the two products as skinny GEMMs on the matrix cores.
int32_t rls_fista_create_batched(rls_operator* op, int32_t nrhs, void* x, void* x0, void* xold, void* res, int64_t ldv,
                                 rls_fista** out) {
  if (!op) return RLS_E_INVALID;
  rls_ctx* ctx = op->ctx;
  if (!x || !x0 || !xold || !res || !out || nrhs < 1 || ldv < op->N)
    return rls_fail(ctx, RLS_E_INVALID, "fista_create_batched: bad argument");
  if (!op->A || !ctx->tune.batched_mfma || !rls_skinny_ok(op->dtype, op->M, op->N, op->A, op->lda) ||
      (op->G && !rls_skinny_ok(op->dtype, op->N, op->N, op->G, op->ldg)))
    return rls_fail(ctx, RLS_E_UNSUPPORTED, "batched FISTA needs A (and AHA, when explicit) with 16-aligned M, N (matrix-core path)");
  RLS_HIP(ctx, rls_enter(ctx));
  rls_alloc_scope alloc_scope(ctx);
  rls_fista* s = new rls_fista();
  s->op = op;
  s->actx = ctx;
  s->actx_id = ctx->id;
  s->device = ctx->device;
  s->buf[0] = x;
  s->buf[1] = xold;
  s->x0 = x0;
  s->res = res;
  s->y = s->y1 = s->res_raw = s->res_raw1 = nullptr;
  s->scn = nullptr;
  s->use_pipe = s->use_gram = false;
  s->reg_kind = RLS_REG_L1;
  s->proj_kind = RLS_PROJ_NONE;
  s->lambda = 0.f;
  s->l21_slices = 1;
  s->initialised = false;
  s->nrhs = nrhs;
  s->ldv = ldv;
  s->sc = s->sc_h = nullptr;
  size_t pb, tb, vb;
  rls_skinny_sizes(op->ctx, op->dtype, op->M, op->N, nrhs, &pb, &tb, &vb, &s->splits);
  s->half = rls_skinny_half(op->ctx, op->dtype, nrhs);
  const size_t yb = (size_t)ldv * nrhs * rls_elem_size(op->dtype);
  hipError_t e = dmalloc(&s->y, yb);
  if (e == hipSuccess) e = dmalloc(&s->Ypack, pb);
  if (e == hipSuccess) e = hipMemsetAsync(s->Ypack, 0, pb, ctx->stream);  // the padding columns of the last group stay zero
  if (e == hipSuccess) e = dmalloc(&s->Tpack, tb);
  if (e == hipSuccess) e = dmalloc(&s->Vpart, vb);
  if (e == hipSuccess) e = dmalloc(&s->sc, sizeof(fista_scalars) * nrhs);
  if (e == hipSuccess) e = hipMemsetAsync(s->sc, 0, sizeof(fista_scalars) * nrhs, ctx->stream);
  if (e == hipSuccess) e = hmalloc(&s->scb_h, sizeof(fista_scalars) * nrhs);
  if (e == hipSuccess) e = hmalloc(&s->sc_h, sizeof(fista_scalars));
  const bool vec16 = ldv % 2 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(xold) | reinterpret_cast<uintptr_t>(res)) & 15) == 0;
  if (e == hipSuccess && op->G && s->half && vec16 && rls_fgramk_resident_ok(ctx, op->dtype, op->N, nrhs, op->G, op->ldg)) {
    size_t yxb, xxb, db;
    rls_fgramk_sizes(op->N, &yxb, &xxb, &db);
    e = resident_alloc(ctx, op, &s->rsync, &s->rsync_h);
    if (e == hipSuccess) e = dmalloc(&s->fk_yx, yxb);
    if (e == hipSuccess) e = hipMemsetAsync(s->fk_yx, 0, yxb, ctx->stream);  // rows >= N are read, never written
    if (e == hipSuccess) e = dmalloc(&s->fk_xx, xxb);
    if (e == hipSuccess) e = dmalloc(&s->fk_dots, db);
    if (e == hipSuccess) e = hipMemsetAsync(s->fk_dots, 0, db, ctx->stream);  // slots of absent workgroups add 0.0
    s->fgramk = e == hipSuccess;
  }
  if (e != hipSuccess) {
    if (s->rsync) dfree(s->rsync);
    if (s->rsync_h) hfree(s->rsync_h);
    if (s->fk_yx) dfree(s->fk_yx);
    if (s->fk_xx) dfree(s->fk_xx);
    if (s->fk_dots) dfree(s->fk_dots);
    if (s->y) dfree(s->y);
    if (s->Ypack) dfree(s->Ypack);
    if (s->Tpack) dfree(s->Tpack);
    if (s->Vpart) dfree(s->Vpart);
    if (s->sc) dfree(s->sc);
    if (s->scb_h) hfree(s->scb_h);
    if (s->sc_h) hfree(s->sc_h);
    delete s;
    return rls_fail(ctx, (int32_t)e, "fista_create_batched: allocation failed");
  }
  *out = s;
  return 0;
}

int32_t rls_fista_init_batched(rls_fista* s, const void* B, int64_t ldb, float rho, float theta, float rel_tol,
                               int32_t iterations, int32_t restart_gradient) {
  if (!s) return RLS_E_INVALID;
  rls_operator* op = s->op;
  rls_ctx* ctx = op->ctx;
  if (s->nrhs < 2 && !s->Ypack) return rls_fail(ctx, RLS_E_STATE, "fista_init_batched on a single-column plan");
  if (!B || ldb < op->M) return rls_fail(ctx, RLS_E_INVALID, "fista_init_batched: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  RLS_TRY(rls_skinny_atb(ctx, op->dtype, fista_skinny_desc(s), B, ldb));  // partial rows of A^H B   (src/FISTA.jl:114)
  // (the resident launch's arrival counters are zeroed by the init kernel: every workgroup writes the same zeros)
  const int n_clear = s->rsync ? (int)(rls_resident_sync_clear_bytes() / sizeof(unsigned)) : 0;
  if (op->dtype == RLS_F32)
    hipLaunchKernelGGL(fista_init_kernel<float>, dim3((unsigned)s->nrhs), dim3(UPD_THREADS), 0, ctx->stream,
                       (float*)s->buf[0], (float*)s->buf[1], (float*)s->x0, (float*)s->res, (float*)s->y, op->N, s->sc,
                       rho, theta, rel_tol, iterations, restart_gradient, s->reg_kind, s->proj_kind, s->lambda,
                       (long long)s->l21_slices, fista_batch_desc<float>(s), (unsigned*)s->rsync, n_clear);
  else
    hipLaunchKernelGGL(fista_init_kernel<float2>, dim3((unsigned)s->nrhs), dim3(UPD_THREADS), 0, ctx->stream,
                       (float2*)s->buf[0], (float2*)s->buf[1], (float2*)s->x0, (float2*)s->res, (float2*)s->y, op->N,
                       s->sc, rho, theta, rel_tol, iterations, restart_gradient, s->reg_kind, s->proj_kind, s->lambda,
                       (long long)s->l21_slices, fista_batch_desc<float2>(s), (unsigned*)s->rsync, n_clear);
  s->initialised = true;
  s->use_pipe = s->use_gram = false;
  s->restart_b = restart_gradient;
  s->rsync_clean = s->rsync != nullptr;
  s->requested = 0;
  s->enq = 0;
  s->resident_used = false;
  return launch_status(ctx);
}

static int32_t fista_step_impl(rls_fista* s, int32_t n_steps);
int32_t rls_fista_get_status_batched(rls_fista* s, rls_fista_status* out) {
  if (!s || !out) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  if (!s->initialised || !s->scb_h) return rls_fail(ctx, RLS_E_STATE, "fista_get_status_batched: not a batched, initialised plan");
  RLS_HIP(ctx, rls_enter(ctx));
  if (s->resident_used) RLS_TRY(resident_fetch_flags(ctx, s->rsync, s->rsync_h));
  RLS_TRY(rls_fetch_add(ctx, s->sc, s->scb_h, sizeof(fista_scalars) * (size_t)s->nrhs));
  RLS_TRY(rls_fetch_wait(ctx));
  if (s->resident_used && resident_lost(ctx, s->rsync, s->rsync_h, &s->resident_off, &s->fallbacks)) {
    // a lost resident launch changed nothing (rls_cgnr_get_status_batched): the live columns are in lockstep, what is missing
    // is re-run on the streaming kernels
    long long at = 0;
    bool live = false;
    for (int b = 0; b < s->nrhs; ++b) {
      if (s->scb_h[b].iteration > at) at = s->scb_h[b].iteration;
      live = live || !s->scb_h[b].done;
    }
    const long long missing = s->requested - at;
    if (live && missing > 0) {
      RLS_TRY(fista_step_impl(s, (int32_t)(missing > 0x7fffffff ? 0x7fffffff : missing)));
      RLS_TRY(rls_fetch_add(ctx, s->sc, s->scb_h, sizeof(fista_scalars) * (size_t)s->nrhs));
      RLS_TRY(rls_fetch_wait(ctx));
    }
  }
  s->resident_used = false;
  for (int b = 0; b < s->nrhs; ++b) {
    const fista_scalars& h = s->scb_h[b];
    out[b].iteration = h.iteration;
    out[b].done = h.done;
    out[b].theta = h.theta;
    out[b].theta_old = h.theta_old;
    out[b].rel_res_norm = (float)h.rel_res_norm;
    out[b].residual = (float)h.res_norm;
    out[b].norm_x0 = (float)h.norm_x0;
    out[b].fallbacks = s->fallbacks;
  }
  return 0;
}

int32_t rls_fista_set_start(rls_fista* s, const void* x_init, int64_t n) {
  if (!s) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  if (!s->initialised) return rls_fail(ctx, RLS_E_STATE, "fista_set_start before fista_init");
  if (!x_init) return rls_fail(ctx, RLS_E_INVALID, "fista_set_start: null pointer");
  if (n != s->op->N) return rls_fail(ctx, RLS_E_INVALID, "fista_set_start: x_init must have the solution's length N");
  if (s->nrhs != 1) return rls_fail(ctx, RLS_E_UNSUPPORTED, "fista_set_start on a batched plan");
  RLS_HIP(ctx, rls_enter(ctx));
  const size_t bytes = (size_t)s->op->N * rls_elem_size(s->op->dtype);
  // iteration 0: state.x == buf[0], xold == 0; the first extrapolated point (src/FISTA.jl:147-148 with
  // thetaold == theta) is ((theta - 1) / theta + 1) x0 -- x0 itself only for the default theta = 1
  RLS_HIP(ctx, hipMemcpyAsync(s->buf[0], x_init, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  const float c2 = (s->theta0 - 1.f) / s->theta0 + 1.f;
  return rls_lincomb(ctx, s->op->dtype, n, c2, 0.f, x_init, 0.f, 0.f, x_init, s->y);
}

static bool fista_use_resident(const rls_fista* s);
static bool fista_use_gram_resident(const rls_fista* s) {
  return s->nrhs == 1 && s->use_gram && s->rsync && !s->resident_off && s->op->ctx->tune.resident;
}

// small systems: the whole step call as a single-workgroup launch, A in ONE CU's registers (as cgnr_use_small), for the
// regularisers the fused updates cover elementwise.  The kernel keeps the plan's state in the pipeline's layout (y0 / y1 by
// `ycur`); a plan without the pipeline's second buffer (shapes the slab kernels do not take) hands it y0 twice, which is the
// two-GEMV path's layout.
// batched Gram mode as ONE resident launch per step call (cgnr_use_gramk): no gradient restart -- its theta is the one global
// scalar the distributed update would have to wait for -- and the elementwise regularisers
static bool fista_use_gramk(const rls_fista* s, int n_steps) {
  return s->fgramk && s->rsync && !s->resident_off && s->op->ctx->tune.resident && !s->restart_b &&
         (s->reg_kind == RLS_REG_NONE || s->reg_kind == RLS_REG_L1 || s->reg_kind == RLS_REG_L2) &&
         (n_steps != 1 || s->op->ctx->tune.resident == 2);
}

static bool fista_use_small(const rls_fista* s) {
  return s->small && s->nrhs == 1 && !s->use_gram && s->op->ctx->tune.small && s->op->ctx->tune.resident &&
         (s->reg_kind == RLS_REG_NONE || s->reg_kind == RLS_REG_L1 || s->reg_kind == RLS_REG_L2);
}

static int32_t fista_step_impl(rls_fista* s, int32_t n_steps) {
  rls_ctx* ctx = s->op->ctx;
  if (fista_use_small(s)) {
    if (n_steps == 0) return 0;
    rls_fista_pipe P = fista_pipe_desc(s);
    if (!P.y1) P.y1 = P.y0;
    P.mb = s->mb_arm;
    s->mb_sent = s->mb_arm.dst != nullptr;
    s->enq += n_steps;
    return rls_fista_small_launch(ctx, s->op->dtype, P, n_steps);
  }
  if (fista_use_gramk(s, n_steps)) {  // AHA explicit, <= 8 columns: the whole call as ONE launch (fista_gramk_resident_kernel)
    if (n_steps == 0) return 0;
    rls_fgramk D;
    D.G = s->op->G;
    D.ldg = s->op->ldg;
    D.N = s->op->N;
    D.nrhs = s->nrhs;
    D.b0 = s->buf[0];
    D.b1 = s->buf[1];
    D.x0 = s->x0;
    D.res = s->res;
    D.y = s->y;
    D.ldv = s->ldv;
    D.sc = s->sc;
    D.Yx = s->fk_yx;
    D.Xx = s->fk_xx;
    D.dots = s->fk_dots;
    D.Ypack = s->Ypack;
    s->resident_used = true;
    return resident_chain(ctx, s->rsync, [&]() {
      return rls_fgramk_resident_launch(ctx, D, s->rsync, n_steps, (unsigned)ctx->tune.resident_spin);
    }, &s->rsync_clean);
  }
  if (s->nrhs > 1) {  // K columns share A: T = A Y, V = A^H T on the matrix cores, then one workgroup per column
    return run_steps(ctx, &s->graph, n_steps, [s]() { return fista_enqueue_batched(s); });
  }
  if (s->use_gram) {
    // explicit AHA: one launch per iteration (two-parity state, see cgnr_gram_kernel); the finish kernel
    // applies the last update and leaves the scalars in both parities, so every call starts at parity 0
    rls_fista_gram P = fista_gram_desc(s);
    const int32_t dtype = s->op->dtype;
    if (fista_use_gram_resident(s)) {  // the whole call as ONE launch, AHA in registers (fista_gram_resident_kernel)
      if (n_steps == 0) return 0;
      s->resident_used = true;
      s->enq += n_steps;
      return resident_chain(ctx, s->rsync, [&]() {
        return rls_fista_gram_resident_launch(ctx, dtype, P, s->rsync, n_steps, (unsigned)ctx->tune.resident_spin);
      }, &s->rsync_clean);
    }
    const int it0 = s->enq;  // buffer hints as in the slab pipeline below
    if (s->graph.exec && s->graph_parity != (it0 & 1)) {
      hipGraphExecDestroy(s->graph.exec);
      s->graph = step_graph();
    }
    s->graph_parity = it0 & 1;
    int parity = 0, k = 0;
    auto one = [ctx, dtype, &P, &parity, &k, it0]() {
      const int hk = pipe_cur_hint(ctx, k);
      P.par_hint = hk < 0 ? -1 : ((it0 + (k > 0 ? k - 1 : 0)) & 1);
      if (hk >= 0 && ctx->tune.pipe_hint_mode == 2) P.par_hint ^= 1;
      ++k;
      const int32_t st = rls_fista_gram_iteration(ctx, dtype, P, parity);
      parity ^= 1;
      return st;
    };
    if (ctx->tune.graph_chunk % 2) {
      for (int i = 0; i < n_steps; ++i) RLS_TRY(one());
    } else {
      RLS_TRY(run_steps(ctx, &s->graph, n_steps, one, [&parity, &k](int c) { parity ^= c & 1; k -= c; }));
    }
    s->enq += n_steps;
    return rls_fista_gram_finish(ctx, dtype, P, n_steps & 1);
  }
  if (fista_use_resident(s) && n_steps != 1) {  // (a single iteration: the pipeline, as in rls_cgnr_step)
    // the whole call as ONE launch, A held in registers (normal.hip, fista_resident_kernel)
    if (n_steps == 0) return 0;
    const rls_fista_pipe P = fista_pipe_desc(s);
    s->resident_used = true;
    s->enq += n_steps;
    return resident_chain(ctx, s->rsync, [&]() {
      return rls_fista_resident_launch(ctx, s->op->dtype, P, s->rsync, n_steps, (unsigned)ctx->tune.resident_spin);
    }, &s->rsync_clean);
  }
  if (s->use_pipe) {
    // iteration k = K_A (applies the gradient/prox/momentum update k-1 in its prologue, then one pass
    // over A for AHA y) + K_R (sums the partial rows); the last update of this call is applied by K_F
    rls_fista_pipe P = fista_pipe_desc(s);
    const int32_t dtype = s->op->dtype;
    // buffer hints: launch k >= 1 of this call finds iteration count enq + k - 1 (launch 0 applies nothing); the
    // hints a cached graph carries belong to the parity of `enq` it was captured with
    const int it0 = s->enq;
    if (s->graph.exec && s->graph_parity != (it0 & 1)) {
      hipGraphExecDestroy(s->graph.exec);
      s->graph = step_graph();
    }
    s->graph_parity = it0 & 1;
    int k = 0;
    RLS_TRY(run_steps(ctx, &s->graph, n_steps, [ctx, dtype, &P, &k, it0]() {
      const int h = pipe_cur_hint(ctx, k);  // -1 where the position may be replayed out of sequence
      P.par_hint = h < 0 ? -1 : ((it0 + (k > 0 ? k - 1 : 0)) & 1);
      if (h >= 0 && ctx->tune.pipe_hint_mode == 2) P.par_hint ^= 1;  // tests: exercise the check-and-reload path
      ++k;
      return rls_fista_pipe_iteration(ctx, dtype, P);
    }, [&k](int c) { k -= c; }));
    s->enq += n_steps;
    P.mb = s->mb_arm;
    s->mb_sent = s->mb_arm.dst != nullptr;
    return rls_fista_pipe_finish(ctx, dtype, P);
  }
  return run_steps(ctx, &s->graph, n_steps, [s]() { return fista_enqueue_iteration(s); });
}

int32_t rls_fista_step(rls_fista* s, int32_t n_steps) {
  if (!s) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  if (!s->initialised) return rls_fail(ctx, RLS_E_STATE, "fista_step before fista_init");
  if (n_steps < 0) return rls_fail(ctx, RLS_E_INVALID, "fista_step: n_steps < 0");
  RLS_HIP(ctx, rls_enter(ctx));
  s->requested += n_steps;
  return fista_step_impl(s, n_steps);
}

static bool fista_use_resident(const rls_fista* s) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return s->nrhs == 1 && s->use_pipe && s->rsync && !s->resident_off && s->op->ctx->tune.resident && al16(s->buf[0]) &&
         al16(s->buf[1]) && al16(s->x0) && al16(s->res);
}

// status read-back of a single-column plan; re-runs on the per-iteration pipeline whatever a lost resident launch left
// undone (cgnr_fetch_status)
static int32_t fista_fetch_status(rls_fista* s) {
  rls_ctx* ctx = s->op->ctx;
  if (s->resident_used) RLS_TRY(resident_fetch_flags(ctx, s->rsync, s->rsync_h));
  RLS_TRY(fetch_scalars(ctx, s->sc, s->sc_h));
  if (s->resident_used && resident_lost(ctx, s->rsync, s->rsync_h, &s->resident_off, &s->fallbacks)) {
    const long long missing = s->requested - (long long)s->sc_h->iteration;
    s->enq = s->sc_h->iteration;  // the buffer-parity hints of the pipeline follow the device's count
    if (!s->sc_h->done && missing > 0) {
      RLS_TRY(fista_step_impl(s, (int32_t)(missing > 0x7fffffff ? 0x7fffffff : missing)));
      RLS_TRY(fetch_scalars(ctx, s->sc, s->sc_h));
    }
  }
  s->resident_used = false;
  return 0;
}

int32_t rls_fista_path(rls_fista* s, int32_t* out) {
  if (!s || !out) return RLS_E_INVALID;
  *out = fista_use_small(s) ? 8 : s->nrhs > 1 ? (fista_use_gramk(s, 0) ? 7 : 3) : fista_use_gram_resident(s) ? 5 : s->use_gram ? 2 : fista_use_resident(s) ? 4 : s->use_pipe ? 1 : 0;
  return 0;
}

static void fista_status_out(const rls_fista* s, const fista_scalars& h, rls_fista_status* out) {
  out->iteration = h.iteration;
  out->done = h.done;
  out->theta = h.theta;
  out->theta_old = h.theta_old;
  out->rel_res_norm = (float)h.rel_res_norm;
  out->residual = (float)h.res_norm;
  out->norm_x0 = (float)h.norm_x0;
  out->fallbacks = s->fallbacks;
}

int32_t rls_fista_get_status(rls_fista* s, rls_fista_status* out) {
  if (!s || !out) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  if (!s->initialised) return rls_fail(ctx, RLS_E_STATE, "fista_get_status before fista_init");
  if (ctx->server == &s->srv && s->srv.alive && s->srv.fresh) {  // a kernel left listening: the mirror holds its last command's status
    fista_status_out(s, *s->sc_h, out);
    return 0;
  }
  RLS_HIP(ctx, rls_enter(ctx));
  RLS_TRY(fista_fetch_status(s));
  fista_status_out(s, *s->sc_h, out);
  return 0;
}

int32_t rls_fista_step_status(rls_fista* s, int32_t n_steps, rls_fista_status* out) {  // as rls_cgnr_step_status
  if (!s || !out) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  if (s->initialised && n_steps > 0 && n_steps <= 8 && !s->resident_used && server_usable(ctx, &s->srv) &&
      (fista_use_small(s) || fista_use_gram_resident(s) || fista_use_resident(s))) {
    // the resident kernel in server mode (cgnr_step_status_server): posted to the kernel left listening, or carried by a launch
    RLS_HIP(ctx, hipSetDevice(s->device));
    s->requested += n_steps;
    s->enq += n_steps;
    const int32_t r = server_command(ctx, &s->srv, s->sc_h, n_steps, true, [&](const rls_srv_args& a) {
      rls_fista_pipe P = fista_pipe_desc(s);
      if (fista_use_small(s)) {  // the single-workgroup kernel: one CU stays, nothing to chain
        if (!P.y1) P.y1 = P.y0;
        P.mb = a.mb;
        return rls_fista_small_launch(ctx, s->op->dtype, P, n_steps, a);
      }
      rls_srv_args ar = a;  // (the resident kernels take the idle time as it is: the run-ahead choice is their instantiation)
      ar.idle_us &= ~RLS_SRV_AHEAD;
      if (fista_use_gram_resident(s)) {  // AHA explicit, in the register files (fista_gram_resident_kernel)
        const rls_fista_gram Pg = fista_gram_desc(s);
        return resident_chain(ctx, s->rsync, [&]() {
          return rls_fista_gram_resident_launch(ctx, s->op->dtype, Pg, s->rsync, n_steps, (unsigned)ctx->tune.resident_spin, ar);
        }, &s->rsync_clean);
      }
      return resident_chain(ctx, s->rsync, [&]() {
        return rls_fista_resident_launch(ctx, s->op->dtype, P, s->rsync, n_steps, (unsigned)ctx->tune.resident_spin, ar);
      }, &s->rsync_clean);
    });
    if (r < 0) return r;
    if (r == 0) {
      fista_status_out(s, *s->sc_h, out);
      return 0;
    }
    if (r == 1) {  // nothing ran: the ordinary path
      s->requested -= n_steps;
      s->enq -= n_steps;
      return rls_fista_step_status(s, n_steps, out);
    }
    return rls_fista_get_status(s, out);  // a launch gave up inside the command: the lost-launch recovery re-runs it
  }
  if (s->initialised && s->nrhs == 1 && n_steps > 0 && !s->resident_used) {
    RLS_HIP(ctx, rls_enter(ctx));
    s->mb_arm = rls_mailbox_arm(ctx, s->sc_h);
  }
  s->mb_sent = false;
  const int32_t st = rls_fista_step(s, n_steps);
  const rls_mailbox_slot mb = s->mb_arm;
  s->mb_arm = rls_mailbox_slot();
  if (st != 0) return st;
  if (!s->mb_sent) return rls_fista_get_status(s, out);
  RLS_TRY(rls_mailbox_wait(ctx, mb.seq));
  fista_status_out(s, *s->sc_h, out);
  return 0;
}

int32_t rls_fista_solution(rls_fista* s, void** x_out) {
  if (!s || !x_out) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  if (!s->initialised) return rls_fail(ctx, RLS_E_STATE, "fista_solution before fista_init");
  RLS_HIP(ctx, rls_enter(ctx));
  RLS_TRY(fista_fetch_status(s));
  *x_out = s->buf[s->sc_h->iteration & 1];
  return 0;
}

// ---- cg! ------------------------------------------------------------------------------------
int32_t rls_cg_create(rls_operator* op, void* u, void* r, void* c, rls_cg** out) {
  if (!op) return RLS_E_INVALID;
  rls_ctx* ctx = op->ctx;
  if (!u || !r || !c || !out) return rls_fail(ctx, RLS_E_INVALID, "cg_create: null pointer");
  RLS_HIP(ctx, rls_enter(ctx));
  rls_alloc_scope alloc_scope(ctx);
  rls_cg* s = new rls_cg();
  s->op = op;
  s->actx = ctx;
  s->actx_id = ctx->id;
  s->device = op->ctx->device;
  s->u = u;
  s->r = r;
  s->c = c;
  s->r1 = s->p1 = nullptr;
  s->dots = nullptr;
  s->psc = s->pscn = s->psc_h = nullptr;
  s->used_pipeline = false;
  s->v1 = nullptr;
  s->gdots = nullptr;
  int32_t st = alloc_scalars(ctx, &s->sc, &s->sc_h);
  if (st != 0) {
    delete s;
    return st;
  }
  if (op->G && rls_gram_pipe_ok(op->dtype, op->N, op->G, op->ldg)) {
    const size_t vb = (size_t)op->N * rls_elem_size(op->dtype);
    const size_t nd = (size_t)2 * rls_gram_pipe_nwg(op->dtype, op->N) * 4 * sizeof(double);
    hipError_t e = dmalloc(&s->r1, vb);
    if (e == hipSuccess) e = dmalloc(&s->p1, vb);
    if (e == hipSuccess) e = dmalloc(&s->v1, vb);
    if (e == hipSuccess) e = hipMemsetAsync(s->v1, 0, vb, ctx->stream);
    if (e == hipSuccess) e = dmalloc(&s->gdots, nd);
    if (e == hipSuccess) e = hipMemsetAsync(s->gdots, 0, nd, ctx->stream);
    if (e == hipSuccess) e = dmalloc(&s->pscn, sizeof(cgnr_scalars));
    if (e == hipSuccess) e = hipMemsetAsync(s->pscn, 0, sizeof(cgnr_scalars), ctx->stream);
    if (e != hipSuccess || alloc_scalars(ctx, &s->psc, &s->psc_h) != 0) {
      rls_cg_destroy(s);
      return rls_fail(ctx, (int32_t)e, "cg_create: hipMalloc failed");
    }
    if (rls_gram_resident_ok(ctx, op->dtype, op->N, op->G, op->ldg)) {
      if (resident_alloc(ctx, op, &s->rsync, &s->rsync_h) == hipSuccess) {
        s->gram_resident = true;
      } else {
        if (s->rsync) dfree(s->rsync);
        s->rsync = nullptr;  // an optimisation only: the one-launch-per-iteration pipeline runs without it
        (void)hipGetLastError();
      }
    }
  } else if (op->slab) {
    if (op->A && rls_cgnr_resident_ok(ctx, op->dtype, op->M, op->N, op->A, op->lda)) {
      const size_t db = (size_t)rls_cgnr_resident_nwg(op->ctx, op->dtype, op->M, op->N) * 4 * sizeof(double);
      if (resident_alloc(ctx, op, &s->rsync, &s->rsync_h) != hipSuccess || dmalloc(&s->rdots, db) != hipSuccess) {
        if (s->rsync) dfree(s->rsync);
        s->rsync = nullptr;  // an optimisation only: the two-launch pipeline runs without it
        (void)hipGetLastError();
      }
    }
    const size_t vb = (size_t)op->N * rls_elem_size(op->dtype);
    const size_t nd = (size_t)((op->N + 15) / 16) * 4 * sizeof(double);
    hipError_t e = dmalloc(&s->r1, vb);
    if (e == hipSuccess) e = dmalloc(&s->p1, vb);
    if (e == hipSuccess) e = dmalloc(&s->dots, nd);
    if (e == hipSuccess) e = hipMemsetAsync(s->dots, 0, nd, ctx->stream);
    if (e == hipSuccess) e = dmalloc(&s->pscn, sizeof(cgnr_scalars));
    if (e == hipSuccess) e = hipMemsetAsync(s->pscn, 0, sizeof(cgnr_scalars), ctx->stream);
    if (e != hipSuccess || alloc_scalars(ctx, &s->psc, &s->psc_h) != 0) {
      rls_cg_destroy(s);
      return rls_fail(ctx, (int32_t)e, "cg_create: hipMalloc failed");
    }
  }
  *out = s;
  return 0;
}

// K right-hand sides sharing A (solve!(solver::ADMM, B) with the shared-A scheduler; src/MultiThreading.jl:30-79 applies
// to every solver): U, R, C are N x nrhs scratch matrices (the CGStateVariables of every column, src/ADMM.jl:129)
int32_t rls_cg_create_batched(rls_operator* op, int32_t nrhs, void* U, void* R, void* Cm, int64_t ldv, rls_cg** out) {
  if (!op) return RLS_E_INVALID;
  rls_ctx* ctx = op->ctx;
  if (!U || !R || !Cm || !out || nrhs < 1 || ldv < op->N) return rls_fail(ctx, RLS_E_INVALID, "cg_create_batched: bad argument");
  if (!op->A || !rls_skinny_ok(op->dtype, op->M, op->N, op->A, op->lda) ||
      (op->G && !rls_skinny_ok(op->dtype, op->N, op->N, op->G, op->ldg)))
    return rls_fail(ctx, RLS_E_UNSUPPORTED, "cg_create_batched: needs A (and AHA, when explicit) with M, N multiples of 16");
  RLS_HIP(ctx, rls_enter(ctx));
  rls_alloc_scope alloc_scope(ctx);
  rls_cg* s = new rls_cg();
  s->op = op;
  s->actx = ctx;
  s->actx_id = ctx->id;
  s->device = ctx->device;
  s->u = U;
  s->r = R;
  s->c = Cm;
  s->r1 = s->p1 = nullptr;
  s->dots = nullptr;
  s->psc = s->pscn = s->psc_h = nullptr;
  s->used_pipeline = false;
  s->v1 = nullptr;
  s->gdots = nullptr;
  s->sc = s->sc_h = nullptr;
  s->nrhs = nrhs;
  s->ldv = ldv;
  size_t pb, tb, vb;
  rls_skinny_sizes(op->ctx, op->dtype, op->M, op->N, nrhs, &pb, &tb, &vb, &s->splits);
  s->half = rls_skinny_half(op->ctx, op->dtype, nrhs);
  hipError_t e = dmalloc(&s->Ppack, pb);
  if (e == hipSuccess) e = hipMemsetAsync(s->Ppack, 0, pb, ctx->stream);  // the padding columns of the last group stay zero
  if (e == hipSuccess) e = dmalloc(&s->Tpack, tb);
  if (e == hipSuccess) e = dmalloc(&s->Vpart, vb);
  if (e != hipSuccess || alloc_scalars(ctx, &s->sc, &s->sc_h, nrhs) != 0) {
    rls_cg_destroy(s);
    return rls_fail(ctx, (int32_t)e, "cg_create_batched: allocation failed");
  }
  *out = s;
  return 0;
}

int32_t rls_cg_destroy(rls_cg* s) {
  if (!s) return RLS_E_INVALID;
  hipSetDevice(s->device);
  rls_alloc_scope alloc_scope(alloc_ctx_of(s->actx, s->actx_id));
  if (s->graph.exec) hipGraphExecDestroy(s->graph.exec);
  if (s->Ppack) dfree(s->Ppack);
  if (s->Tpack) dfree(s->Tpack);
  if (s->Vpart) dfree(s->Vpart);
  if (s->rsync) dfree(s->rsync);
  if (s->rdots) dfree(s->rdots);
  if (s->rsync_h) hfree(s->rsync_h);
  if (s->r1) dfree(s->r1);
  if (s->p1) dfree(s->p1);
  if (s->dots) dfree(s->dots);
  if (s->v1) dfree(s->v1);
  if (s->gdots) dfree(s->gdots);
  if (s->psc) dfree(s->psc);
  if (s->pscn) dfree(s->pscn);
  if (s->psc_h) hfree(s->psc_h);
  if (s->sc) dfree(s->sc);
  if (s->sc_h) hfree(s->sc_h);
  delete s;
  return 0;
}

int32_t rls_cg_solve(rls_cg* s, void* x, const void* b, float rho, int32_t maxiter, float reltol) {
  if (!s) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  if (!x || !b || maxiter < 0) return rls_fail(ctx, RLS_E_INVALID, "cg_solve: bad argument");
  if (s->nrhs != 1) return rls_fail(ctx, RLS_E_STATE, "cg_solve on a batched plan: batched cg! runs inside rls_admm_step");
  RLS_HIP(ctx, rls_enter(ctx));
  s->last.x = x;
  s->last.b = b;
  s->last.rho = rho;
  s->last.reltol = reltol;
  s->last.maxiter = maxiter;
  s->last.valid = true;
  return cg_solve_impl(s, x, b, rho, maxiter, reltol, admm_fuse_v());
}

// row-sharded cg! (ADMM on a row-partitioned A): the operator apply and the update are separate calls with the
// caller's all-reduce of c between them; the unfused scalars / kernels, `done` on the device as usual
//   rls_cg_local_apply(s, x)      c_partial = A_g^H A_g x        (warm start; null = the direction u, skipped once done)
//   rls_cg_local_start(...)       r = b - (c + rho x), u = r, residual, tol
//   rls_cg_local_update(s, x)     alpha, x, r, residual, next direction
int32_t rls_cg_local_apply(rls_cg* s, const void* x) {
  if (!s) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  RLS_HIP(ctx, rls_enter(ctx));
  return x ? op_normal(s->op, x, s->c, nullptr) : op_normal(s->op, s->u, s->c, &s->sc->done);
}

int32_t rls_cg_local_start(rls_cg* s, const void* x, const void* b, float rho, int32_t maxiter, float reltol) {
  if (!s) return RLS_E_INVALID;
  rls_operator* op = s->op;
  rls_ctx* ctx = op->ctx;
  if (!x || !b || maxiter < 0) return rls_fail(ctx, RLS_E_INVALID, "cg_local_start: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  s->used_pipeline = false;
  if (op->dtype == RLS_F32)
    hipLaunchKernelGGL(cg_start_kernel<float>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (const float*)x,
                       (const float*)b, (float*)s->u, (float*)s->r, (const float*)s->c, op->N, s->sc, rho, reltol,
                       maxiter, typed_fuse<float>(admm_fuse_v()), col_batch<float>());
  else
    hipLaunchKernelGGL(cg_start_kernel<float2>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (const float2*)x,
                       (const float2*)b, (float2*)s->u, (float2*)s->r, (const float2*)s->c, op->N, s->sc, rho, reltol,
                       maxiter, typed_fuse<float2>(admm_fuse_v()), col_batch<float2>());
  return launch_status(ctx);
}

int32_t rls_cg_local_update(rls_cg* s, void* x) {
  if (!s) return RLS_E_INVALID;
  rls_operator* op = s->op;
  rls_ctx* ctx = op->ctx;
  if (!x) return rls_fail(ctx, RLS_E_INVALID, "cg_local_update: null x");
  RLS_HIP(ctx, rls_enter(ctx));
  if (op->dtype == RLS_F32)
    hipLaunchKernelGGL(cg_update_kernel<float>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (float*)x, (float*)s->u,
                       (float*)s->r, (float*)s->c, op->N, s->sc, col_batch<float>());
  else
    hipLaunchKernelGGL(cg_update_kernel<float2>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (float2*)x,
                       (float2*)s->u, (float2*)s->r, (float2*)s->c, op->N, s->sc, col_batch<float2>());
  return launch_status(ctx);
}

// ---------------------------------------------------------------------------------------------
// OptISTA / POGM (restart = :none) as resident launches (kernel: normal.hip, pgm_resident_kernel)
// ---------------------------------------------------------------------------------------------
// The solver state (x, y, z, zold / xold, res, x0 and the 4-word record) belongs to the caller, as with the per-iteration
// entry points rls_optista_update_async / rls_pogm_update_async; the plan owns what a resident launch needs on top: the
// arrival counters and the N-vector of the flat exchange.
struct rls_pgm {
  rls_operator* op;
  rls_ctx* actx;
  uint64_t actx_id = 0;
  int device;
  void* rsync = nullptr;
  unsigned* rsync_h = nullptr;
  void* raw = nullptr;
  bool resident_off = false;
  int fallbacks = 0;
};

int32_t rls_pgm_create(rls_operator* op, rls_pgm** out) {
  if (!op || !out) return RLS_E_INVALID;
  rls_ctx* ctx = op->ctx;
  *out = nullptr;
  RLS_HIP(ctx, rls_enter(ctx));
  if (!(op->slab && op->A && !op->G && rls_pgm_resident_ok(ctx, op->dtype, op->M, op->N, op->A, op->lda)))
    return RLS_E_UNSUPPORTED;  // (not an error state: the caller keeps its launch-per-iteration sequence)
  rls_alloc_scope alloc_scope(ctx);
  rls_pgm* s = new rls_pgm();
  s->op = op;
  s->actx = ctx;
  s->actx_id = ctx->id;
  s->device = ctx->device;
  hipError_t e = resident_alloc(ctx, op, &s->rsync, &s->rsync_h);
  if (e == hipSuccess) e = dmalloc(&s->raw, (size_t)op->N * rls_elem_size(op->dtype));
  if (e != hipSuccess) {
    if (s->rsync) dfree(s->rsync);
    if (s->rsync_h) hfree(s->rsync_h);
    if (s->raw) dfree(s->raw);
    delete s;
    (void)hipGetLastError();
    return rls_fail(ctx, (int32_t)e, "pgm_create: hipMalloc failed");
  }
  *out = s;
  return 0;
}

int32_t rls_pgm_destroy(rls_pgm* s) {
  if (!s) return RLS_E_INVALID;
  hipSetDevice(s->device);
  rls_alloc_scope alloc_scope(alloc_ctx_of(s->actx, s->actx_id));
  dfree(s->rsync);
  dfree(s->raw);
  hfree(s->rsync_h);
  delete s;
  return 0;
}

int32_t rls_pgm_step_resident(rls_pgm* s, int32_t kind, int32_t n_steps, int32_t first_iteration, const float* coefs, void* v0,
                              void* v1, void* v2, void* o0, void* res, const void* x0, int32_t reg_kind, int32_t proj_kind,
                              float norm_x0, float rel_tol, void* state_d) {
  if (!s) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if ((kind != 0 && kind != 1) || n_steps < 0 || n_steps > RLS_PGM_MAX_IT || first_iteration < 0 || !coefs || !v0 || !v1 || !v2 ||
      !o0 || !res || !x0 || !state_d || reg_kind < RLS_REG_NONE || reg_kind > RLS_REG_L2 || proj_kind < RLS_PROJ_NONE ||
      proj_kind > RLS_PROJ_POSITIVE || (kind == 0 && proj_kind != RLS_PROJ_NONE))
    return rls_fail(ctx, RLS_E_INVALID, "pgm_step_resident: bad argument");
  if (!(al16(v0) && al16(v1) && al16(v2) && al16(o0) && al16(res) && al16(x0)))
    return rls_fail(ctx, RLS_E_INVALID, "pgm_step_resident: vectors must be 16-byte aligned");
  if (s->resident_off || !ctx->tune.resident) return RLS_E_UNSUPPORTED;  // lost a launch earlier: per-iteration launches
  if (n_steps == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  rls_pgm_coefs C;
  memcpy(C.c, coefs, sizeof(float) * 8 * (size_t)n_steps);
  rls_pgm_desc D;
  D.A = s->op->A;
  D.lda = s->op->lda;
  D.M = s->op->M;
  D.N = s->op->N;
  D.kind = kind;
  D.v0 = v0;
  D.v1 = v1;
  D.v2 = v2;
  D.o0 = o0;
  D.res = res;
  D.x0 = x0;
  D.slab = s->op->slab;
  D.raw = s->raw;
  D.st = (pgm_state*)state_d;
  D.norm_x0 = norm_x0;
  D.rel_tol = rel_tol;
  D.reg_kind = reg_kind;
  D.proj_kind = proj_kind;
  D.first_it = first_iteration;
  return resident_chain(ctx, s->rsync, [&]() {
    return rls_pgm_resident_launch(ctx, s->op->dtype, D, C, s->rsync, n_steps, (unsigned)ctx->tune.resident_spin);
  });
}

// POGM with restart = :gradient as resident launches: theta, sigma, gamma live in the record (pogm_auto_state) and the kernel
// forms every iteration's coefficients from them (src/POGM.jl:183-232), so a block needs no table -- only rho, lambda, sigma_fac
// and the iteration count of the solve (the last iteration's theta rule, :185).
int32_t rls_pogm_step_resident_restart(rls_pgm* s, int32_t n_steps, int32_t first_iteration, float rho, float lambda, float sigma_fac,
                                       int32_t iterations, void* xbuf, void* ybuf, void* z, void* w, void* xold, void* res,
                                       const void* x0, int32_t reg_kind, int32_t proj_kind, float norm_x0, float rel_tol,
                                       void* state_d) {
  if (!s) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (n_steps < 0 || first_iteration < 0 || iterations < 1 || iterations >= (1 << 24) || !xbuf || !ybuf || !z || !w || !xold || !res ||
      !x0 || !state_d || reg_kind < RLS_REG_NONE || reg_kind > RLS_REG_L2 || proj_kind < RLS_PROJ_NONE || proj_kind > RLS_PROJ_POSITIVE)
    return rls_fail(ctx, RLS_E_INVALID, "pogm_step_resident_restart: bad argument");
  if (!(al16(xbuf) && al16(ybuf) && al16(z) && al16(w) && al16(xold) && al16(res) && al16(x0)))
    return rls_fail(ctx, RLS_E_INVALID, "pogm_step_resident_restart: vectors must be 16-byte aligned");
  if (s->resident_off || !ctx->tune.resident) return RLS_E_UNSUPPORTED;
  if (n_steps == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  rls_pgm_coefs C;
  C.c[0][0] = rho;
  C.c[0][1] = lambda;
  C.c[0][2] = sigma_fac;
  C.c[0][3] = (float)iterations;  // (exact: < 2^24)
  rls_pgm_desc D;
  D.A = s->op->A;
  D.lda = s->op->lda;
  D.M = s->op->M;
  D.N = s->op->N;
  D.kind = 2;
  D.v0 = xbuf;
  D.v1 = ybuf;
  D.v2 = z;
  D.v3 = w;
  D.o0 = xold;
  D.res = res;
  D.x0 = x0;
  D.slab = s->op->slab;
  D.raw = s->raw;
  D.st = (pgm_state*)state_d;
  D.norm_x0 = norm_x0;
  D.rel_tol = rel_tol;
  D.reg_kind = reg_kind;
  D.proj_kind = proj_kind;
  D.first_it = first_iteration;
  return resident_chain(ctx, s->rsync, [&]() {
    return rls_pgm_resident_launch(ctx, s->op->dtype, D, C, s->rsync, n_steps, (unsigned)ctx->tune.resident_spin);
  });
}

// after the launches of a sequence: how many of them gave up (bounded wait; they changed nothing).  Synchronises.
int32_t rls_pgm_lost(rls_pgm* s, int32_t* lost, int32_t* fallbacks_total) {
  if (!s || !lost) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  RLS_HIP(ctx, rls_enter(ctx));
  RLS_TRY(resident_fetch_flags(ctx, s->rsync, s->rsync_h));
  RLS_TRY(rls_fetch_wait(ctx));
  *lost = (int32_t)resident_lost(ctx, s->rsync, s->rsync_h, &s->resident_off, &s->fallbacks);
  if (fallbacks_total) *fallbacks_total = s->fallbacks;
  return 0;
}

int32_t rls_cg_path(rls_cg* s, int32_t* out) {
  if (!s || !out) return RLS_E_INVALID;
  const rls_ctx* ctx = s->op->ctx;
  const bool res = s->rsync && !s->resident_off && ctx->tune.resident;
  if (s->nrhs > 1 || s->Vpart) *out = 3;
  else if (cg_use_gram_pipeline(s)) *out = (res && s->gram_resident) ? 5 : 2;
  else if (cg_use_pipeline(s)) *out = res ? 4 : 1;
  else *out = 0;
  return 0;
}

int32_t rls_cg_get_status(rls_cg* s, rls_cg_status* out) {
  if (!s || !out) return RLS_E_INVALID;
  rls_ctx* ctx = s->op->ctx;
  RLS_HIP(ctx, rls_enter(ctx));
  out->fallbacks = s->fallbacks;
  if (s->resident_used) {
    // a lost resident launch left x at its warm start: repeat the solve on the per-iteration pipeline
    RLS_TRY(resident_fetch_flags(ctx, s->rsync, s->rsync_h));
    RLS_TRY(rls_fetch_wait(ctx));
    if (resident_lost(ctx, s->rsync, s->rsync_h, &s->resident_off, &s->fallbacks) && s->last.valid)
      RLS_TRY(cg_solve_impl(s, s->last.x, s->last.b, s->last.rho, s->last.maxiter, s->last.reltol, admm_fuse_v()));
    out->fallbacks = s->fallbacks;
  }
  if (s->used_pipeline) {
    RLS_TRY(fetch_scalars(ctx, s->psc, s->psc_h));
    out->iterations = s->psc_h->iteration;
    out->residual = (float)sqrt(s->psc_h->rr);
    out->tol = s->psc_h->rel_tol * (float)s->psc_h->z0;
    return 0;
  }
  RLS_TRY(fetch_scalars(ctx, s->sc, s->sc_h));
  out->iterations = s->sc_h->iteration;
  out->residual = (float)s->sc_h->residual;
  out->tol = (float)s->sc_h->tol;
  return 0;
}

// ---- ADMM fused elementwise steps (identity regTrafo) ----------------------------------------
int32_t rls_admm_pre(rls_ctx* ctx, int32_t dtype, int64_t n, void* beta, const void* beta_y, const void* z,
                     const void* u, const void* x, void* xold, float rho, int32_t accumulate) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || n < 0 || (n > 0 && (!beta || !beta_y || !z || !u || !x || !xold)))
    return rls_fail(ctx, RLS_E_INVALID, "admm_pre: bad argument");
  if (n == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  unsigned grid = (unsigned)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(admm_pre_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, (float*)beta, (const float*)beta_y,
                       (const float*)z, (const float*)u, (const float*)x, (float*)xold, n, rho, accumulate);
  else
    hipLaunchKernelGGL(admm_pre_kernel<float2>, dim3(grid), dim3(256), 0, ctx->stream, (float2*)beta,
                       (const float2*)beta_y, (const float2*)z, (const float2*)u, (const float2*)x, (float2*)xold, n,
                       rho, accumulate);
  return launch_status(ctx);
}

int32_t rls_admm_post(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, const void* xold, const void* z,
                      const void* zold, void* u, float* out_h) {
  RLS_CHECK_CTX(ctx);
  if (!rls_dtype_ok(dtype) || n <= 0 || !x || !xold || !z || !zold || !u || !out_h)
    return rls_fail(ctx, RLS_E_INVALID, "admm_post: bad argument");
  RLS_HIP(ctx, rls_enter(ctx));
  if (dtype == RLS_F32)
    hipLaunchKernelGGL(admm_post_kernel<float>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (const float*)x,
                       (const float*)xold, (const float*)z, (const float*)zold, (float*)u, n, ctx->res_d);
  else
    hipLaunchKernelGGL(admm_post_kernel<float2>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (const float2*)x,
                       (const float2*)xold, (const float2*)z, (const float2*)zold, (float2*)u, n, ctx->res_d);
  RLS_TRY(launch_status(ctx));
  RLS_HIP(ctx, hipMemcpyAsync(ctx->res_h, ctx->res_d, sizeof(float) * 6, hipMemcpyDeviceToHost, ctx->stream));
  RLS_HIP(ctx, rls_stream_wait(ctx->stream));
  for (int i = 0; i < 6; ++i) out_h[i] = ctx->res_h[i];
  return 0;
}

// ---- ADMM plan ------------------------------------------------------------------------------
int32_t rls_admm_create(rls_cg* cg, rls_admm** out) {
  if (!cg) return RLS_E_INVALID;
  rls_ctx* ctx = cg->op->ctx;
  if (!out) return rls_fail(ctx, RLS_E_INVALID, "admm_create: null out");
  RLS_HIP(ctx, rls_enter(ctx));
  rls_alloc_scope alloc_scope(ctx);
  rls_admm* a = new rls_admm();
  a->cg = cg;
  a->actx = ctx;
  a->actx_id = ctx->id;
  a->device = ctx->device;
  a->ready = false;
  a->log = a->log_h = nullptr;
  a->log_cap = 0;
  a->enq = 0;
  a->nrhs = cg->nrhs;
  const int32_t st = alloc_scalars(ctx, &a->sc, &a->sc_h, cg->nrhs);
  if (st != 0) {
    delete a;
    return st;
  }
  *out = a;
  return 0;
}

int32_t rls_admm_destroy(rls_admm* a) {
  if (!a) return RLS_E_INVALID;
  hipSetDevice(a->device);
  rls_alloc_scope alloc_scope(alloc_ctx_of(a->actx, a->actx_id));
  if (a->log) dfree(a->log);
  if (a->log_h) hfree(a->log_h);
  dfree(a->sc);
  hfree(a->sc_h);
  delete a;
  return 0;
}

int32_t rls_admm_init(rls_admm* a, const rls_admm_params* p) {
  if (!a) return RLS_E_INVALID;
  rls_ctx* ctx = a->cg->op->ctx;
  a->ready = false;
  if (!p || !p->x || !p->xold || !p->beta || !p->beta_y || !p->z0 || !p->z1 || !p->u || p->iterations < 0 ||
      p->iterations_cg < 0)
    return rls_fail(ctx, RLS_E_INVALID, "admm_init: bad argument");
  const int32_t dtype = a->cg->op->dtype;
  switch (p->reg_kind) {
    case RLS_REG_NONE:
    case RLS_REG_L1:
    case RLS_REG_L2:
      break;
    case RLS_REG_TV:
      if (p->proj_kind != RLS_PROJ_NONE || p->tv_iterations < 0 ||
          !rls_tv_single_ok(ctx, dtype, p->tv_ndims, p->tv_shape, p->tv_ntv, p->tv_dims))
        return rls_fail(ctx, RLS_E_UNSUPPORTED, "admm_init: TV prox does not fit the single-workgroup FGP kernel");
      {
        int64_t n = 1;
        for (int k = 0; k < p->tv_ndims; ++k) n *= p->tv_shape[k];
        if (n != a->cg->op->N) return rls_fail(ctx, RLS_E_INVALID, "admm_init: prod(shape) != N");
      }
      break;
    default:
      return rls_fail(ctx, RLS_E_UNSUPPORTED, "admm_init: regulariser not fused (use the per-call path)");
  }
  if (p->proj_kind != RLS_PROJ_NONE && p->proj_kind != RLS_PROJ_REAL && p->proj_kind != RLS_PROJ_POSITIVE)
    return rls_fail(ctx, RLS_E_INVALID, "admm_init: bad proj_kind");
  RLS_HIP(ctx, rls_enter(ctx));
  rls_alloc_scope alloc_scope(ctx);
  const int cap = p->iterations > 0 ? p->iterations : 1;
  if (cap > a->log_cap) {
    if (a->log) dfree(a->log);
    if (a->log_h) hfree(a->log_h);
    a->log = a->log_h = nullptr;
    a->log_cap = 0;
    RLS_HIP(ctx, dmalloc(&a->log, sizeof(float) * ADMM_REC * cap * a->nrhs));
    RLS_HIP(ctx, hmalloc(&a->log_h, sizeof(float) * ADMM_REC * cap * a->nrhs));
    a->log_cap = cap;
  }
  a->P = *p;
  a->enq = 0;
  a->requested = 0;
  hipLaunchKernelGGL(admm_reset_kernel<float>, dim3((unsigned)a->nrhs), dim3(1), 0, ctx->stream, a->sc, p->iterations,
                     p->rho, p->sigma_abs, p->rel_tol);
  RLS_TRY(launch_status(ctx));
  a->ready = true;
  return 0;
}

int32_t rls_admm_step(rls_admm* a, int32_t n_outer) {
  if (!a) return RLS_E_INVALID;
  rls_cg* cg = a->cg;
  rls_ctx* ctx = cg->op->ctx;
  if (!a->ready) return rls_fail(ctx, RLS_E_STATE, "admm_step before admm_init");
  if (n_outer < 0) return rls_fail(ctx, RLS_E_INVALID, "admm_step: n_outer < 0");
  RLS_HIP(ctx, rls_enter(ctx));
  const rls_admm_params& P = a->P;
  const int32_t dtype = cg->op->dtype;
  const int64_t n = cg->op->N;
  if (a->nrhs > 1 || cg->Vpart) return admm_step_batched(a, n_outer);
  a->requested = (int)std::min<long long>((long long)a->requested + n_outer, (long long)P.iterations);
  for (int k = 0; k < n_outer && a->enq < P.iterations; ++k, ++a->enq) {
    void* zcur = (a->enq & 1) ? P.z1 : P.z0;
    void* znew = (a->enq & 1) ? P.z0 : P.z1;
    admm_fuse_v F;
    F.beta_y = P.beta_y;
    F.z = zcur;
    F.u = P.u;
    F.beta = P.beta;
    F.xold = P.xold;
    F.rho = P.rho;
    F.skip = &a->sc->done;
    F.poison = &a->sc->done;  // a resident cg! that gives up sets done = 2: the z / u kernels behind it skip
    RLS_TRY(cg_solve_impl(cg, P.x, P.beta, P.rho, P.iterations_cg, P.tol_inner, F));  // :236-244
    const int* cg_it = cg->used_pipeline ? &cg->psc->iteration : &cg->sc->iteration;
    const int z_ready = P.reg_kind == RLS_REG_TV;
    if (z_ready)
      RLS_TRY(rls_tv_single_launch(ctx, dtype, P.tv_ndims, P.tv_shape, P.tv_ntv, P.tv_dims, P.x, P.u, znew,
                                   P.prox_lambda, P.tv_iterations, &a->sc->done));
    if (dtype == RLS_F32)
      hipLaunchKernelGGL(admm_zu_kernel<float>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (float*)P.x,
                         (const float*)P.xold, (float*)znew, (const float*)zcur, (float*)P.u, n, P.reg_kind,
                         P.prox_lambda, P.proj_kind, z_ready, a->sc, cg_it, a->log, col_batch<float>(), nullptr);
    else
      hipLaunchKernelGGL(admm_zu_kernel<float2>, dim3(1), dim3(UPD_THREADS), 0, ctx->stream, (float2*)P.x,
                         (const float2*)P.xold, (float2*)znew, (const float2*)zcur, (float2*)P.u, n, P.reg_kind,
                         P.prox_lambda, P.proj_kind, z_ready, a->sc, cg_it, a->log, col_batch<float2>(), nullptr);
    RLS_TRY(launch_status(ctx));
  }
  return 0;
}

// ---- ADMM on a row-partitioned A (src/ADMM.jl:191-330; one regulariser, identity regTrafo, vary_rho = :none) --------------
// plans[r]: rls_cg_create + rls_admm_create + rls_admm_init on rank r's shard operator, every rank with the same
// parameters (sigma_abs from the length of the WHOLE b, :214).  The x-update's cg! is the distributed part: its operator
// applies are per-shard products followed by ONE all-reduce of c each; it always runs its `iterations_cg` half-step
// pairs -- its own convergence and the plan's `done` are device flags replicated on every rank (identical inputs,
// identical order), so the collective count never depends on the data.  Everything else of the outer iteration is the
// single-GPU plan's kernels, replicated: beta = beta_y + rho (z - u) inside the start kernel, the FGP launch for a TV
// term, admm_zu_kernel (projections, z, u, the norms, `converged`).
int32_t rls_admm_init_rowsharded(rls_comm* comm, rls_admm* const* plans, const void* const* b_parts) {
  RLS_TRY(rowsharded_check<rls_admm>(comm, plans, admm_op, admm_single));
  if (!b_parts) return RLS_E_INVALID;
  const int n = rls_comm_size(comm);
  for (int r = 0; r < n; ++r)
    if (!plans[r]->ready || !b_parts[r]) return rls_fail(plans[r]->cg->op->ctx, RLS_E_STATE, "admm_init_rowsharded before admm_init");
  const int64_t N = plans[0]->cg->op->N;
  const int32_t dtype = plans[0]->cg->op->dtype;
  int round0 = 0;
  RLS_TRY(rls_comm_next_rounds(comm, 1, N, dtype, &round0));
  const std::vector<rls_comm_phase> phases = {
      {[&](int r, int) {  // beta_y = A^H b   (:198), per shard
         rls_operator* op = plans[r]->cg->op;
         RLS_HIP(op->ctx, hipSetDevice(op->ctx->device));
         RLS_TRY(rls_launch_gemv(op->ctx, dtype, RLS_OP_C, op->M, op->N, 1.f, 0.f, op->A, op->lda, b_parts[r], 0.f, 0.f,
                                 plans[r]->P.beta_y, nullptr));
         return rls_comm_publish(comm, r, plans[r]->P.beta_y, N, dtype, round0);
       }, true},
      {[&](int r, int) { return rls_comm_collect(comm, r, plans[r]->P.beta_y, N, dtype, round0); }, false}};
  return rls_comm_run(comm, phases, 1);
}

int32_t rls_admm_step_rowsharded(rls_comm* comm, rls_admm* const* plans, int32_t n_outer) {
  RLS_TRY(rowsharded_check<rls_admm>(comm, plans, admm_op, admm_single));
  if (n_outer < 0) return RLS_E_INVALID;
  const int n = rls_comm_size(comm);
  for (int r = 0; r < n; ++r) {
    if (!plans[r]->ready) return rls_fail(plans[r]->cg->op->ctx, RLS_E_STATE, "admm_step_rowsharded before admm_init");
    if (plans[r]->enq != plans[0]->enq || plans[r]->P.iterations != plans[0]->P.iterations ||
        plans[r]->P.iterations_cg != plans[0]->P.iterations_cg)
      return rls_fail(plans[r]->cg->op->ctx, RLS_E_INVALID, "admm_step_rowsharded: the ranks' plans are out of step");
  }
  const int64_t N = plans[0]->cg->op->N;
  const int32_t dtype = plans[0]->cg->op->dtype;
  const int icg = plans[0]->P.iterations_cg;
  const int todo = std::min<int>(n_outer, plans[0]->P.iterations - plans[0]->enq);
  if (todo <= 0) return 0;
  const int per_outer = icg + 1;  // all-reduce rounds of one outer iteration: the warm-start apply + one per inner step
  int round0 = 0;
  RLS_TRY(rls_comm_next_rounds(comm, todo * per_outer, N, dtype, &round0));
  auto zbufs = [&](rls_admm* a, void** zcur, void** znew) {
    *zcur = (a->enq & 1) ? a->P.z1 : a->P.z0;
    *znew = (a->enq & 1) ? a->P.z0 : a->P.z1;
  };
  std::vector<rls_comm_phase> phases;
  phases.push_back({[&](int r, int k) {  // c_g = A_g^H A_g x   (warm start of cg!, src/ADMM.jl:244)
                      rls_admm* a = plans[r];
                      rls_cg* cg = a->cg;
                      RLS_HIP(cg->op->ctx, hipSetDevice(cg->op->ctx->device));
                      cg->used_pipeline = false;
                      RLS_TRY(op_normal(cg->op, a->P.x, cg->c, &a->sc->done));
                      return rls_comm_publish(comm, r, cg->c, N, dtype, round0 + k * per_outer);
                    }, true});
  for (int j = 0; j <= icg; ++j) {
    phases.push_back({[&, j](int r, int k) {
                        rls_admm* a = plans[r];
                        rls_cg* cg = a->cg;
                        rls_ctx* ctx = cg->op->ctx;
                        RLS_HIP(ctx, rls_enter(ctx));
                        RLS_TRY(rls_comm_collect(comm, r, cg->c, N, dtype, round0 + k * per_outer + j));
                        void *zcur, *znew;
                        zbufs(a, &zcur, &znew);
                        if (j == 0) {  // beta = beta_y + rho (z - u), xold = x, r = beta - (c + rho x), u = r   (:236-243)
                          admm_fuse_v F;
                          F.beta_y = a->P.beta_y;
                          F.z = zcur;
                          F.u = a->P.u;
                          F.beta = a->P.beta;
                          F.xold = a->P.xold;
                          F.rho = a->P.rho;
                          F.skip = &a->sc->done;
                          if (dtype == RLS_F32) admm_local_start<float>(a, F);
                          else admm_local_start<float2>(a, F);
                        } else {
                          RLS_TRY(rls_cg_local_update(cg, a->P.x));
                        }
                        RLS_TRY(launch_status(ctx));
                        if (j < icg) {  // next inner step's operator apply on the direction u
                          RLS_TRY(op_normal(cg->op, cg->u, cg->c, &cg->sc->done));
                          return rls_comm_publish(comm, r, cg->c, N, dtype, round0 + k * per_outer + j + 1);
                        }
                        // the rest of the outer iteration, replicated (:246-309)
                        if (a->P.reg_kind == RLS_REG_TV)
                          RLS_TRY(rls_tv_single_launch(ctx, dtype, a->P.tv_ndims, a->P.tv_shape, a->P.tv_ntv, a->P.tv_dims, a->P.x,
                                                       a->P.u, znew, a->P.prox_lambda, a->P.tv_iterations, &a->sc->done));
                        if (dtype == RLS_F32) admm_local_finish<float>(a, zcur, znew);
                        else admm_local_finish<float2>(a, zcur, znew);
                        ++a->enq;
                        a->requested = a->enq;
                        return launch_status(ctx);
                      }, j < icg});
  }
  return rls_comm_run(comm, phases, todo);
}

// batched plans: out_h[nrhs]; log_h (nullable): nrhs blocks of log_records records, column after column
int32_t rls_admm_get_status_batched(rls_admm* a, rls_admm_status* out, float* log_h, int32_t log_records) {
  if (!a || !out) return RLS_E_INVALID;
  rls_ctx* ctx = a->cg->op->ctx;
  if (!a->ready) return rls_fail(ctx, RLS_E_STATE, "admm_get_status before admm_init");
  if (log_records < 0 || (log_records > 0 && !log_h)) return rls_fail(ctx, RLS_E_INVALID, "admm_get_status: bad log");
  RLS_HIP(ctx, rls_enter(ctx));
  const size_t stride = (size_t)ADMM_REC * a->log_cap;
  RLS_HIP(ctx, hipMemcpyAsync(a->log_h, a->log, sizeof(float) * stride * a->nrhs, hipMemcpyDeviceToHost, ctx->stream));
  RLS_HIP(ctx, hipMemcpyAsync(a->sc_h, a->sc, sizeof(admm_scalars) * a->nrhs, hipMemcpyDeviceToHost, ctx->stream));
  RLS_HIP(ctx, rls_stream_wait(ctx->stream));
  for (int b = 0; b < a->nrhs; ++b) {
    const int it = a->sc_h[b].iteration;
    out[b].iteration = it;
    out[b].done = a->sc_h[b].done;
    out[b].fallbacks = 0;
    out[b].delta = out[b].sk = out[b].eps_pri = out[b].rk = out[b].eps_dua = 0.f;
    out[b].cg_iterations = 0;
    if (it > 0 && it <= a->log_cap) {
      const float* rec = a->log_h + stride * b + (size_t)(it - 1) * ADMM_REC;
      out[b].delta = rec[0];
      out[b].sk = rec[1];
      out[b].eps_pri = rec[2];
      out[b].rk = rec[3];
      out[b].eps_dua = rec[4];
      out[b].cg_iterations = (int32_t)rec[5];
    }
    const int ncopy = it < log_records ? it : log_records;
    for (int i = 0; i < ncopy * ADMM_REC; ++i) log_h[(size_t)b * log_records * ADMM_REC + i] = a->log_h[stride * b + i];
  }
  return 0;
}

int32_t rls_admm_get_status(rls_admm* a, rls_admm_status* out, float* log_h, int32_t log_records) {
  if (!a || !out) return RLS_E_INVALID;
  rls_ctx* ctx = a->cg->op->ctx;
  if (!a->ready) return rls_fail(ctx, RLS_E_STATE, "admm_get_status before admm_init");
  if (a->nrhs != 1) return rls_fail(ctx, RLS_E_STATE, "admm_get_status on a batched plan: use rls_admm_get_status_batched");
  if (log_records < 0 || (log_records > 0 && !log_h)) return rls_fail(ctx, RLS_E_INVALID, "admm_get_status: bad log");
  RLS_HIP(ctx, rls_enter(ctx));
  rls_cg* cg = a->cg;
  int nrec = a->enq < a->log_cap ? a->enq : a->log_cap;
  if (nrec > 0) RLS_TRY(rls_fetch_add(ctx, a->log, a->log_h, sizeof(float) * ADMM_REC * nrec));
  if (cg->resident_used) RLS_TRY(resident_fetch_flags(ctx, cg->rsync, cg->rsync_h));
  RLS_TRY(fetch_scalars(ctx, a->sc, a->sc_h));  // synchronises the stream
  if (cg->resident_used && resident_lost(ctx, cg->rsync, cg->rsync_h, &cg->resident_off, &cg->fallbacks)) {
    // A resident cg! gave up: it was a no-op and poisoned `done` (= 2), so everything queued behind it skipped.  The
    // outer iteration it belonged to has changed nothing that its repetition does not rewrite (beta, xold), so clear the
    // poison and run the missing outer iterations again; the inner solves now take the per-iteration pipeline.
    a->fallbacks = cg->fallbacks;
    if (a->sc_h->done == 2) {
      const int it_done = a->sc_h->iteration;
      RLS_HIP(ctx, hipMemsetAsync(&a->sc->done, 0, sizeof(int), ctx->stream));
      a->enq = it_done;
      const int missing = a->requested - it_done;
      a->requested = it_done;
      if (missing > 0) RLS_TRY(rls_admm_step(a, missing));
      nrec = a->enq < a->log_cap ? a->enq : a->log_cap;
      if (nrec > 0) RLS_TRY(rls_fetch_add(ctx, a->log, a->log_h, sizeof(float) * ADMM_REC * nrec));
      RLS_TRY(fetch_scalars(ctx, a->sc, a->sc_h));
    }
  }
  const int it = a->sc_h->iteration;
  a->enq = it;  // a plan that stopped early continues (after a re-init only) from the device's count
  out->iteration = it;
  out->done = a->sc_h->done;
  out->fallbacks = a->fallbacks;
  out->delta = out->sk = out->eps_pri = out->rk = out->eps_dua = 0.f;
  out->cg_iterations = 0;
  if (it > 0 && it <= nrec) {
    const float* rec = a->log_h + (size_t)(it - 1) * ADMM_REC;
    out->delta = rec[0];
    out->sk = rec[1];
    out->eps_pri = rec[2];
    out->rk = rec[3];
    out->eps_dua = rec[4];
    out->cg_iterations = (int32_t)rec[5];
  }
  const int ncopy = it < log_records ? it : log_records;
  for (int i = 0; i < ncopy * ADMM_REC; ++i) log_h[i] = a->log_h[i];
  return 0;
}

int32_t rls_admm_step_status(rls_admm* a, int32_t n_outer, rls_admm_status* out, float* log_h, int32_t log_records) {
  RLS_TRY(rls_admm_step(a, n_outer));
  return rls_admm_get_status(a, out, log_h, log_records);
}

}  // extern "C"
