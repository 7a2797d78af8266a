/*
 * rls_mi355x.h -- C ABI of librls_mi355x.so: the MI355X (gfx950) backend for the iterative inner
 * loop of RegularizedLeastSquares.jl (CGNR / FISTA / ADMM on a dense operator).
 *
 * The reference has no FFI today: its backend boundary is Julia multiple dispatch on the array
 * type of A / b.  Every entry point below names the reference method(s) a Julia
 * extension would overload with a `ccall` to it (file:line relative to the reference tree).
 *
 * Conventions
 *   - plain C linkage; every function returns an int32 status: 0 = OK, >0 = hipError_t,
 *     <0 = RLS_E_*.  rls_last_error_string(ctx) describes the last failure on that context.
 *   - all array arguments are DEVICE pointers owned by the caller, unless the name says _h (host).
 *   - matrices are column-major with an explicit leading dimension `lda` in ELEMENTS
 *     (Julia `Matrix`); complex = interleaved (re, im) float pairs.
 *   - complex scalars cross the boundary as two floats (no struct-by-value).
 *   - work is enqueued on the context's stream and is asynchronous; only functions that return
 *     host scalars (rls_nrm2, rls_dotc, *_status, memcpy_d2h) synchronise.
 *   - the library is re-entrant: all mutable state lives in rls_ctx (one per host thread / GPU).
 */
#ifndef RLS_MI355X_H
#define RLS_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLS_ABI_VERSION 2

typedef struct rls_ctx rls_ctx;           /* device + stream + workspace                           */
typedef struct rls_operator rls_operator; /* dense forward operator A (+ optional Gram matrix)     */
typedef struct rls_cgnr rls_cgnr;         /* fused CGNR plan bound to caller-owned state vectors   */
typedef struct rls_fista rls_fista;       /* fused FISTA plan                                      */
typedef struct rls_cg rls_cg;             /* fused cg! plan (ADMM x-update)                        */

enum { RLS_F32 = 0, RLS_C32 = 1 };                 /* Float32, ComplexF32                          */
/* Float64 / ComplexF64: accepted by the rls_*_d entry points ONLY (the L1 protocol with double scalars, at the end of the
 * vector section); every other entry point -- the fused plans, the resident and matrix-core kernels -- is Float32 / ComplexF32
 * (north_star) and answers RLS_E_INVALID for these codes. */
enum { RLS_F64 = 2, RLS_C64 = 3 };
enum { RLS_OP_N = 0, RLS_OP_T = 1, RLS_OP_C = 2 }; /* A, transpose(A), adjoint(A)                  */
enum { RLS_NORMAL_MATRIXFREE = 0, RLS_NORMAL_GRAM = 1 };

enum {
  RLS_E_INVALID = -1,   /* bad argument (null pointer, negative size, unknown dtype/op)            */
  RLS_E_UNSUPPORTED = -2,
  RLS_E_WORKSPACE = -3, /* workspace too small / allocation failed                                */
  RLS_E_STATE = -4      /* call out of order (e.g. step before init)                              */
};

/* ---------------------------------------------------------------------------------------------
 * context / memory.   Julia side: the device vector type's constructor, finalizer, copyto!, Array().
 * ------------------------------------------------------------------------------------------- */
int32_t rls_abi_version(void);
int32_t rls_ctx_create(int32_t device, rls_ctx** out);
/* borrow an existing hipStream_t (e.g. torch's current stream); the ctx does not destroy it */
int32_t rls_ctx_create_on_stream(int32_t device, void* hip_stream, rls_ctx** out);
int32_t rls_ctx_destroy(rls_ctx* ctx);
int32_t rls_ctx_sync(rls_ctx* ctx);
void* rls_ctx_stream(rls_ctx* ctx);
const char* rls_last_error_string(rls_ctx* ctx);
int32_t rls_device_count(int32_t* out);
/* kernel-selection knobs for measurement sweeps and for forcing a path in the parity tests; not part of the
 * reference.  Per context: "gemvn_g", "gemvn_waves", "gemvt_cols" (0 = heuristic), "graph_chunk", "use_graph",
 * "fuse_level", "fused_normal" (one-pass normal operator), "cgnr_pipeline" (2-launch CGNR), "gram_pipeline"
 * (1-launch Gram-mode CGNR / cg), "batched_mfma" (matrix-core batched path and Gram GEMM), "pipe_hint_mode" (0: the
 * host tells the 2-launch pipeline which (r, p) buffer pair is current, 1: never, 2: deliberately wrong -- tests),
 * "resident" (1: a whole step call as ONE launch with A / AHA held in registers where the shape allows, 0: the per-iteration
 * pipelines), "resident_spin" (bound of the in-kernel waits, in polls), "resident_preclear" (1: a plan's init kernel zeroes its
 * resident kernel's arrival counters, saving the memset launch ahead of the first step call).  Process-wide (measurement only, set before the
 * plan is created): "slab_g", "slab_wv", "slab_order", "red_threads", "resident_barrier", "tv_fused_max_n",
 * "tv_fused_2d", "skinny_t_waves", "skinny_t_u", "skinny_v_waves", "skinny_v_u", "skinny_v_splits", "skinny_half"
 * (the (8 re | 8 im) operand layout for <= 8 complex right-hand sides), "skinny_t_roll", "skinny_v_roll", "skinny_g_roll" (rolling-window
 * depth of the batched kernels' load pipelines), "gram_lds_kib", "kaczmarz_nt".  Per context again: "small" (1: systems that
 * fit one CU's registers run a step call as a single-workgroup launch), "resident_server" (1: rls_cgnr_step_status / rls_fista_step_status leave the resident kernel listening for the
 * next call, see there), "resident_server_idle_us", "status_mailbox" (>= 1: status read-backs through a kernel
 * that stores into pinned host memory + a host spin; 2, the default: rls_*_step_status has the call's last kernel do that store where
 * it can; 0: hipMemcpyAsync + stream wait), "resident_l2_rows" (1, the default: the matrix-free resident kernels keep the partial
 * rows of their in-kernel all-reduce in the XCD's L2 whenever every workgroup sits on the XCD its group assumes -- checked in
 * every launch; 0: always written through to the memory side), "gemvt_reverse" (-1, the default: the transposed GEMV walks the
 * columns from the last one down exactly when A is larger than the Infinity Cache -- the second product of a two-GEMV normal
 * operator then starts on what the first one left in the cache; 0 / 1 force the direction; no bit of any result depends on it),
 * "fista_defer" (1, the default: the matrix-free resident FISTA / OptISTA / POGM kernels sum ||res||^2 off the critical path
 * where they can; 0: the block reduction in place -- the same bits either way).  Process-wide: "slab_multi" (1, the default: a shape
 * with more row blocks than the device has CUs runs its one-pass kernels as one workgroup per CU that walks several blocks, the
 * next one streaming in under the products of the current one; 0: one workgroup per block, in rounds -- the partial sums are
 * added in a different order, so results differ in the last bits between the two settings, each being reproducible). */
int32_t rls_tune_set(rls_ctx* ctx, const char* key, int32_t value);
/* Device memory is STREAM-ORDERED on the context's stream (a private hipMemPool per device; RLS_ALLOC=sync or a device without
 * memory pools: hipMalloc / hipFree): rls_free does not wait for the stream, the block is reused behind everything enqueued on
 * this context so far.  Memory that ANOTHER context's stream (or another library's) still uses must be synchronised by the
 * caller before it is freed.  rls_free with a handle that is not a live context -- a finalizer that runs after its context was
 * destroyed -- does not dereference the handle and frees synchronously. */
int32_t rls_malloc(rls_ctx* ctx, size_t bytes, void** out);  /* similar(b, dims...)  src/CGNR.jl:92-95 */
int32_t rls_free(rls_ctx* ctx, void* p);
int32_t rls_memcpy_h2d(rls_ctx* ctx, void* dst, const void* src_h, size_t bytes);
int32_t rls_memcpy_d2h(rls_ctx* ctx, void* dst_h, const void* src, size_t bytes); /* synchronises */
int32_t rls_memcpy_d2d(rls_ctx* ctx, void* dst, const void* src, size_t bytes);   /* copyto! src/CGNR.jl:126 */
int32_t rls_fill(rls_ctx* ctx, int32_t dtype, int64_t n, void* x, float re, float im); /* v .= c  src/CGNR.jl:108-115, src/FISTA.jl:121-123 */

/* timing helpers for the measurement harness (hipEvents on the ctx stream) */
int32_t rls_timer_start(rls_ctx* ctx);
int32_t rls_timer_stop_ms(rls_ctx* ctx, float* ms_out); /* synchronises */

/* ---------------------------------------------------------------------------------------------
 * BLAS-2:  y = alpha * op(A) * x + beta * y       (5-arg mul!)
 * replaces LinearAlgebra.mul!(y, A, x) / mul!(x, adjoint(A), y):  src/CGNR.jl:132,151
 * src/FISTA.jl:114,152  src/ADMM.jl:198  and the 5-arg form src/CGNR.jl:119, src/ADMM.jl:239-240
 * ------------------------------------------------------------------------------------------- */
int32_t rls_gemv(rls_ctx* ctx, int32_t dtype, int32_t op, int64_t M, int64_t N, float alpha_re, float alpha_im,
                 const void* A, int64_t lda, const void* x, float beta_re, float beta_im, void* y);

/* ---------------------------------------------------------------------------------------------
 * BLAS-1.   replaces norm / dot / rmul! / fused broadcasts: src/CGNR.jl:125,153-174,182
 * src/FISTA.jl:118,147-156,172  src/ADMM.jl:236-309
 * result_h: host float[2] (re, im); the *_dev forms write float[2] to device memory, no sync.
 * ------------------------------------------------------------------------------------------- */
int32_t rls_nrm2(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, float* result_h);
int32_t rls_nrm2_dev(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, float* result_d);
int32_t rls_dotc(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, const void* y, float* result_h); /* conj(x).y */
int32_t rls_dotc_dev(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, const void* y, float* result_d);
int32_t rls_asum(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, float* result_h); /* norm(x,1) with complex modulus: ProxL1.jl:29-32 */
int32_t rls_scal(rls_ctx* ctx, int32_t dtype, int64_t n, float a_re, float a_im, void* x);                   /* rmul!(x, a)   */
int32_t rls_axpy(rls_ctx* ctx, int32_t dtype, int64_t n, float a_re, float a_im, const void* x, void* y);    /* y .+= a .* x  */
int32_t rls_axpby(rls_ctx* ctx, int32_t dtype, int64_t n, float a_re, float a_im, const void* x, float b_re,
                  float b_im, void* y);                                                                      /* y = a x + b y */
/* z = a x + b y (out of place; z may alias x or y): src/ADMM.jl:282-284 */
int32_t rls_lincomb(rls_ctx* ctx, int32_t dtype, int64_t n, float a_re, float a_im, const void* x, float b_re,
                    float b_im, const void* y, void* z);

/* ---------------------------------------------------------------------------------------------
 * proximal maps (in place).   replaces prox!(reg, x, lambda)
 * ------------------------------------------------------------------------------------------- */
/* ---- the same protocol for Float64 / ComplexF64 arrays (dtype RLS_F64 / RLS_C64): double scalars in, double results out.
 * The reference's suites run every solver in Float32 AND Float64 (test/testSolvers.jl:242) and its prox tests in ComplexF64
 * (test/testProxMaps.jl:47,78,106): a double-precision caller runs the reference's own loops (src/CGNR.jl:143-178,
 * src/FISTA.jl:139-185, src/ADMM.jl:230-322) on these primitives.  Plain coalesced kernels, Float64 fixed-order reductions;
 * the reductions synchronise and return {re, im} (nrm2: {norm, 0}).  gemv_d: y = alpha op(A) x + beta y, op as rls_gemv. */
int32_t rls_fill_d(rls_ctx* ctx, int32_t dtype, int64_t n, void* x, double re, double im);
int32_t rls_scal_d(rls_ctx* ctx, int32_t dtype, int64_t n, double a_re, double a_im, void* x);
int32_t rls_axpy_d(rls_ctx* ctx, int32_t dtype, int64_t n, double a_re, double a_im, const void* x, void* y);
int32_t rls_lincomb_d(rls_ctx* ctx, int32_t dtype, int64_t n, double a_re, double a_im, const void* x, double b_re, double b_im,
                      const void* y, void* z); /* z = a x + b y (y may be z: axpby; b == 0 never reads y) */
int32_t rls_nrm2_d(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, double* result_h);
int32_t rls_asum_d(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, double* result_h);
int32_t rls_dotc_d(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, const void* y, double* result_h); /* conj(x).y */
int32_t rls_gemv_d(rls_ctx* ctx, int32_t dtype, int32_t op, int64_t M, int64_t N, double alpha_re, double alpha_im, const void* A,
                   int64_t lda, const void* x, double beta_re, double beta_im, void* y);
int32_t rls_prox_l1_d(rls_ctx* ctx, int32_t dtype, int64_t n, void* x, double lambda);       /* ProxL1.jl:18-22, eps(Float64) */
int32_t rls_prox_l2_d(rls_ctx* ctx, int32_t dtype, int64_t n, void* x, double lambda);       /* ProxL2.jl:18-21 */
int32_t rls_prox_l21_d(rls_ctx* ctx, int32_t dtype, int64_t n, int64_t slices, void* x, double lambda); /* ProxL21.jl:30-35 */
int32_t rls_prox_positive_d(rls_ctx* ctx, int32_t dtype, int64_t n, void* x);                /* ProxPositive.jl:16-20 */
int32_t rls_prox_real_d(rls_ctx* ctx, int32_t dtype, int64_t n, void* x);                    /* ProxReal.jl:16-19 */
/* proxTV! (FGP, ProxTV.jl:89-125) with the geometry arguments of rls_prox_tv_fgp; the workspace (three dual arrays + xTmp) comes
 * from the context's stream-ordered pool */
int32_t rls_prox_tv_fgp_d(rls_ctx* ctx, int32_t dtype, int32_t ndims, const int64_t* shape, int32_t ntv, const int32_t* dims, void* x,
                          double lambda, int32_t iterations);

/* the row-action solver in double precision: transpose(A) (the row-access layout of src/Kaczmarz.jl:391), rownorm² (src/Utils.jl:20-23,
 * device output), diag(w) A, and the sweep of rls_kaczmarz_sweep with double denominators (src/Kaczmarz.jl:283-308) */
int32_t rls_transpose_d(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda, void* At, int64_t ldat);
int32_t rls_rownorm2_d(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda, double* out_d);
int32_t rls_scale_rows_d(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* w, const void* A, int64_t lda, void* B, int64_t ldb);
int32_t rls_kaczmarz_sweep_d(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* At, int64_t ldat, int32_t nrhs, void* X,
                             int64_t ldx, const void* U, int64_t ldu, void* VL, int64_t ldvl, const int32_t* rows_d, const double* denom_d,
                             int32_t nused, double eps_w, int32_t n_sweeps);

int32_t rls_prox_l1(rls_ctx* ctx, int32_t dtype, int64_t n, void* x, float lambda);        /* src/proximalMaps/ProxL1.jl:18-22 */
int32_t rls_prox_l2(rls_ctx* ctx, int32_t dtype, int64_t n, void* x, float lambda);        /* src/proximalMaps/ProxL2.jl:18-21 */
int32_t rls_prox_l21(rls_ctx* ctx, int32_t dtype, int64_t n, int64_t slices, void* x, float lambda); /* ProxL21.jl:30-35; ext/..GPUArraysExt/ProxL21.jl:1-12 */
int32_t rls_prox_positive(rls_ctx* ctx, int32_t dtype, int64_t n, void* x);                /* ProxPositive.jl:16-20, Utils.jl:114-144 */
int32_t rls_prox_real(rls_ctx* ctx, int32_t dtype, int64_t n, void* x);                    /* ProxReal.jl:16-19 */
int32_t rls_norm_l21(rls_ctx* ctx, int32_t dtype, int64_t n, int64_t slices, const void* x, float lambda,
                     float* result_h);                                                     /* ProxL21.jl:42-46 */

/* TV.  shape[ndims] column-major extents; dims[ntv] 0-based dimensions differenced, in order.
 * Gradient layout = LinearOperatorCollection.GradientOp: per dim d a block of
 * prod(shape with shape[d]-1) forward differences g[i] = x[i] - x[i+e_d], blocks concatenated. */
int64_t rls_tv_grad_len(int32_t ndims, const int64_t* shape, int32_t ntv, const int32_t* dims);
/* g = alpha * grad(x) + beta * g           mul!(pq, grad, xTmp, 1/(8 lambda), 1)  ProxTV.jl:109 */
int32_t rls_tv_grad(rls_ctx* ctx, int32_t dtype, int32_t ndims, const int64_t* shape, int32_t ntv,
                    const int32_t* dims, const void* x, void* g, float alpha, float beta);
/* x = alpha * grad^T(g) + beta * x         mul!(xTmp, transpose(grad), rs, -lambda, 1)  ProxTV.jl:108,123 */
int32_t rls_tv_grad_t(rls_ctx* ctx, int32_t dtype, int32_t ndims, const int64_t* shape, int32_t ntv,
                      const int32_t* dims, const void* g, void* x, float alpha, float beta);
int32_t rls_tv_restrict(rls_ctx* ctx, int32_t dtype, int64_t n, void* pq);                /* tv_restrictMagnitude! ProxTV.jl:135-139; ext/..ProxTV.jl:1-8 */
int32_t rls_tv_lincomb(rls_ctx* ctx, int32_t dtype, int64_t n, void* rs, float t3, const void* pq, float t2,
                       const void* pqOld);                                                /* tv_linearcomb! ProxTV.jl:141-145; ext/..ProxTV.jl:10-17 */
/* whole FGP loop: proxTV!(x, lambda, p::TVParams; iterationsTV)  ProxTV.jl:89-125.
 * workspace: >= rls_prox_tv_workspace_bytes(...) bytes of device scratch (the TVParams buffers). */
size_t rls_prox_tv_workspace_bytes(int32_t dtype, int32_t ndims, const int64_t* shape, int32_t ntv,
                                   const int32_t* dims);
int32_t rls_prox_tv_fgp(rls_ctx* ctx, int32_t dtype, int32_t ndims, const int64_t* shape, int32_t ntv,
                        const int32_t* dims, void* x, float lambda, int32_t iterations, void* workspace,
                        size_t workspace_bytes);

/* ---------------------------------------------------------------------------------------------
 * operator handle: the backend's operator type for A (SURVEY 3.1 "two operator modes").
 * A' * A on it yields the lazy normal operator (matrix-free, two GEMVs) unless a Gram matrix is set.
 * ------------------------------------------------------------------------------------------- */
int32_t rls_operator_create(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda,
                            rls_operator** out);
int32_t rls_operator_set_gram(rls_operator* op, const void* AHA, int64_t ld); /* explicit AHA= keyword: src/CGNR.jl:46-49 */
int32_t rls_operator_destroy(rls_operator* op);
/* y = A x, x = A^H y, v = A^H A p (mul!(v, AHA, p): src/CGNR.jl:151, src/FISTA.jl:152, src/Utils.jl:278) */
int32_t rls_operator_mul(rls_operator* op, const void* x, void* y);
int32_t rls_operator_mul_adj(rls_operator* op, const void* y, void* x);
int32_t rls_operator_mul_normal(rls_operator* op, const void* p, void* v);
/* the same, as a no-op when the device int32 *skip_d is non-zero at launch time (nullable) */
int32_t rls_operator_mul_normal_skip(rls_operator* op, const void* p, void* v, const void* skip_d);
/* setup GEMM AHA = A' * A (src/CGNR.jl:49) on device; G is N x N column-major, ld >= N */
int32_t rls_gram(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda, void* G, int64_t ld);
/* Singular-value soft-thresholding.
 * rls_prox_nuclear: prox!(::NuclearRegularization, x, lambda) (src/proximalMaps/ProxNuclear.jl:26-31): x is the
 *   m x n column-major matrix reshape(x, svtShape); U, S, V = svd; prox!(L1, S, lambda); x = U diag(S) V'.
 * rls_prox_llr: proxLLRNonOverlapping! (src/proximalMaps/ProxLLR.jl:43-88): x = reshape(x, shape..., K), the image is
 *   circularly shifted by `shift` (the reference draws it with rand when randshift = true; pass zeros for
 *   randshift = false), cut into distinct blocks of `block` voxels (edge blocks zero-padded) and every
 *   prod(block) x K matrix is singular-value thresholded.  ndims <= 3; n = length(x).
 * One wave per matrix, one-sided Jacobi SVD in LDS; RLS_E_UNSUPPORTED when a matrix does not fit (150 KiB). */
int32_t rls_prox_nuclear(rls_ctx* ctx, int32_t dtype, int64_t m, int64_t n, void* x, float lambda);
int32_t rls_prox_llr(rls_ctx* ctx, int32_t dtype, int32_t ndims, const int64_t* shape, const int64_t* block,
                     const int64_t* shift, int64_t n, void* x, float lambda);
/* Fused elementwise half of one OptISTA iteration (src/OptISTA.jl:176-204), after res = AHA x:
 *   zold = z; z = y; res -= x0; y -= step res; prox!(reg, y, thr); z = c_z z + x + c_y y; x = c_x x + c_zn z + c_zo zold
 * with step = rho gamma, thr = rho gamma lambda, c_z = -1/gamma, c_y = 1/gamma, c_x = -beta, c_zn = 1 + alpha + beta,
 * c_zo = -alpha (host-side Float32 recurrences).  reg_kind: RLS_REG_NONE / L1 / L2.  *res_norm_h = ||res|| (synchronises). */
int32_t rls_optista_update(rls_ctx* ctx, int32_t dtype, int64_t n, void* res, const void* x0, void* x, void* y, void* z,
                           void* zold, float step, int32_t reg_kind, float thr, float c_z, float c_y, float c_x,
                           float c_zn, float c_zo, float* res_norm_h);
/* Fused elementwise half of one POGM iteration (src/POGM.jl:176-233), after res = AHA x.  On entry xbuf = x_k and
 * ybuf = y_{k-1}; on exit xbuf holds the gradient point x_k - rho res (the new y after the reference's swap, :203) and
 * ybuf the new x: the caller swaps its two references.  x_new = prox(c_y y + c_x1 (x - rho res) + c_xo x + c_z z),
 * z = the pre-prox value, xold = x_k.  restart != 0: the gradient-restart vector w is updated as in :218-232 with
 * rho_over_gamma.  out_h: float[4] = { ||res||, real<w,x>, real<w,z>, real<w,res> } (synchronises). */
int32_t rls_pogm_update(rls_ctx* ctx, int32_t dtype, int64_t n, void* res, const void* x0, void* xbuf, void* ybuf,
                        void* xold, void* z, void* w, float rho, float c_y, float c_x1, float c_xo, float c_z,
                        int32_t reg_kind, float thr, int32_t proj_kind, int32_t restart, float rho_over_gamma,
                        float* out_h);
/* Deferred forms: no host read-back.  state_d points at 4 device words {int32 iteration, int32 done, float ||res||,
 * pad}, zeroed by the caller before the first iteration; each launch is a no-op once `done` is set, increments
 * `iteration`, stores ||res|| and sets done = (||res|| / norm_x0 < rel_tol) -- the reference's stopping test
 * (src/OptISTA.jl:206-209, src/POGM.jl:234-237).  Pair with rls_operator_mul_normal_skip(op, x, res, &state->done).
 * POGM: restart = :none only. */
int32_t rls_optista_update_async(rls_ctx* ctx, int32_t dtype, int64_t n, void* res, const void* x0, void* x, void* y,
                                 void* z, void* zold, float step, int32_t reg_kind, float thr, float c_z, float c_y,
                                 float c_x, float c_zn, float c_zo, float norm_x0, float rel_tol, void* state_d);
int32_t rls_pogm_update_async(rls_ctx* ctx, int32_t dtype, int64_t n, void* res, const void* x0, void* xbuf, void* ybuf,
                              void* xold, void* z, float rho, float c_y, float c_x1, float c_xo, float c_z,
                              int32_t reg_kind, float thr, int32_t proj_kind, float norm_x0, float rel_tol,
                              void* state_d);
/* OptISTA / POGM (restart = :none) with a whole block of iterations in ONE launch (A in the register files, the elementwise
 * half of iterate redundantly in every workgroup -- the layout of rls_fista_step on its resident path).  The caller keeps
 * the solver state exactly as for rls_optista_update_async / rls_pogm_update_async (vectors, the 4-word record state_d);
 * the plan owns the launch's own scratch.  rls_pgm_create returns RLS_E_UNSUPPORTED (no error recorded) when the operator
 * does not fit the resident form (not a dense matrix held in the register files, another dtype): the caller stays on the
 * per-iteration entry points.
 *   kind 0 = OptISTA: v0, v1, v2 = x, y, z; o0 = zold; coefs row = {step, thr, c_z, c_y, c_x, c_zn, c_zo, 0}
 *   kind 1 = POGM:    v0, v1, v2 = xbuf, ybuf, z; o0 = xold; coefs row = {rho, thr, c_y, c_x1, c_xo, c_z, 0, 0}
 * (the float arguments of the *_update_async calls, one row of 8 per iteration, n_steps <= 48 rows).  POGM swaps the roles
 * of xbuf / ybuf once per iteration run, as the sequence of rls_pogm_update_async calls does: after an odd number of
 * iterations the current x is in ybuf.  first_iteration = the count state_d holds when this launch starts; a launch that
 * finds another count (an earlier launch of the sequence gave up) does nothing.  A launch that cannot get all its workgroups
 * resident within the bounded wait changes nothing either: rls_pgm_lost (synchronises) reports such launches, the record
 * shows fewer iterations than requested and the caller runs the rest through the per-iteration entry points; the plan
 * answers RLS_E_UNSUPPORTED from then on. */
typedef struct rls_pgm rls_pgm;
int32_t rls_pgm_create(rls_operator* op, rls_pgm** out);
int32_t rls_pgm_destroy(rls_pgm* plan);
int32_t rls_pgm_step_resident(rls_pgm* plan, int32_t kind, int32_t n_steps, int32_t first_iteration, const float* coefs, void* v0,
                              void* v1, void* v2, void* o0, void* res, const void* x0, int32_t reg_kind, int32_t proj_kind,
                              float norm_x0, float rel_tol, void* state_d);
int32_t rls_pgm_lost(rls_pgm* plan, int32_t* lost, int32_t* fallbacks_total);
/* POGM with restart = :gradient, deferred: theta, sigma, gamma live in the device record (8 words: int32 iteration,
 * int32 done, float ||res||, pad, float theta, theta_old, sigma, gamma) and every launch derives its coefficients
 * from them in Float32 with the host's operation order (src/POGM.jl:183-201), applies the update, evaluates the
 * restart criterion (:218-232) and the stopping test.  The caller sets theta, sigma, gamma before the first
 * iteration, alternates xbuf / ybuf between launches, and reads the record back after the last one. */
int32_t rls_pogm_update_auto(rls_ctx* ctx, int32_t dtype, int64_t n, void* res, const void* x0, void* xbuf, void* ybuf,
                             void* xold, void* z, void* w, float rho, float lambda, float sigma_fac, int32_t iterations,
                             int32_t reg_kind, int32_t proj_kind, float norm_x0, float rel_tol, void* state_d);
/* The same as resident launches of a rls_pgm plan (A in the register files, n_steps <= any count per launch: the coefficients
 * are formed in the kernel from the record's theta, sigma, gamma, so no table travels): xbuf / ybuf swap roles once per
 * iteration run as in rls_pgm_step_resident (kind 1); w is loop-carried beside x, y, z.  `iterations` = the solve's iteration
 * count MINUS the count the record started from (the last iteration's theta rule, src/POGM.jl:185, compares the record's
 * count); first_iteration, lost launches and RLS_E_UNSUPPORTED as for rls_pgm_step_resident. */
int32_t rls_pogm_step_resident_restart(rls_pgm* plan, int32_t n_steps, int32_t first_iteration, float rho, float lambda, float sigma_fac,
                                       int32_t iterations, void* xbuf, void* ybuf, void* z, void* w, void* xold, void* res,
                                       const void* x0, int32_t reg_kind, int32_t proj_kind, float norm_x0, float rel_tol,
                                       void* state_d);
/* At = transpose(A) (no conjugation): N x M column-major, leading dimension ldat >= N.  Row k of A becomes the
 * contiguous column k of At -- the "structure for row access" that the reference's row-action solvers ask for
 * (createLinearSolver(Kaczmarz, transpose(A_T)), src/Kaczmarz.jl:391, dot_with_matrix_row(::Transpose…)
 * src/Utils.jl:63-67, 82-86). */
int32_t rls_transpose(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda, void* At,
                      int64_t ldat);
/* Kaczmarz row sweeps: for i in usedIndices: iterate_row_index (src/Kaczmarz.jl:283-308) =
 *   tau = dot_with_matrix_row(A, x, row) (src/Utils.jl:55-88, non-conjugating);
 *   alpha = denom[i] * (u[row] - tau - eps_w * vl[row]);  kaczmarz_update!(A, x, row, alpha): x += alpha * conj(A[row,:])
 *   (src/Kaczmarz.jl:435-517, GPU ext Kaczmarz.jl:1-31);  vl[row] += alpha * eps_w.
 * At: transpose(A) as rls_transpose builds it.  X (N x nrhs, ldx), U (M x nrhs, ldu: the right-hand sides b),
 * VL (M x nrhs, ldvl): one independent solve per column (src/MultiThreading.jl:30-79), one workgroup each.
 * rows_d[i] (0-based row, = rowindex[usedIndices[i]]) and denom_d[i] in processing order, i < nused, rows distinct;
 * n_sweeps repeats the same order.  X and VL are updated in place; asynchronous on the context's stream.
 * RLS_E_UNSUPPORTED for N beyond the register-resident sweep (8192 chunks of 16 bytes = 16384 ComplexF32). */
int32_t rls_kaczmarz_sweep(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* At, int64_t ldat,
                           int32_t nrhs, void* X, int64_t ldx, const void* U, int64_t ldu, void* VL, int64_t ldvl,
                           const int32_t* rows_d, const float* denom_d, int32_t nused, float eps_w, int32_t n_sweeps);
/* squared row norms of A, out_d[m] = rownorm²(A, m) (src/Utils.jl:20-23) for all rows at once -- the
 * mapreduce(abs2, +, A, dims = 2) of ext/RegularizedLeastSquaresGPUArraysExt/NormalizedRegularization.jl:1-5
 * that normalize(::SystemMatrixBasedNormalization, A, b) (src/Regularization/NormalizedRegularization.jl:47-58)
 * sums; out_d: device float[M].  Synchronises (setup path). */
int32_t rls_rownorm2(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda, float* out_d);
/* B = diag(w) A: ProdOp(WeightingOp(w), A) (docs/src/literate/howto/normal_operator.jl:41-44, src/Utils.jl:23,102)
 * materialised, so that the weighted operator and its normal operator A^H W^H W A run on the same kernels as a
 * plain dense A.  w: device vector of length M, same dtype as A; B: M x N, leading dimension ldb (B may alias A). */
int32_t rls_scale_rows(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* w, const void* A, int64_t lda,
                       void* B, int64_t ldb);

/* ---------------------------------------------------------------------------------------------
 * fused CGNR.   replaces init!(::CGNR, ::CGNRState, b) src/CGNR.jl:107-130 and
 * iterate(::CGNR, ::CGNRState) src/CGNR.jl:143-178 + done() :181-185 for device state vectors.
 * x, r (= x0 in the reference), p (= pl), v (= vl): caller-owned length-N device vectors.
 * ------------------------------------------------------------------------------------------- */
typedef struct rls_cgnr_status {
  int32_t iteration; /* state.iteration                                                          */
  int32_t done;      /* converged || iteration >= min(iterations, N)   src/CGNR.jl:185           */
  float alpha_re, alpha_im; /* state.alphal (complex-typed for complex problems)                 */
  float beta_re, beta_im;
  float zeta;     /* ||r||^2 at the start of the last iteration                                  */
  float residual; /* ||r|| now: solverconvergence(state).residual  src/CGNR.jl:136               */
  float z0;       /* ||A^H b||                                                                   */
  int32_t fallbacks; /* resident launches that timed out (no-ops) and were re-run on the per-iteration pipeline
                      * by this or an earlier status call; 0 in normal operation                                  */
} rls_cgnr_status;

int32_t rls_cgnr_create(rls_operator* op, void* x, void* r, void* p, void* v, rls_cgnr** out);
int32_t rls_cgnr_destroy(rls_cgnr* s);
/* b: length M (or length N when the operator has only a Gram matrix: b must then be A^H b) */
int32_t rls_cgnr_init(rls_cgnr* s, const void* b, float lambda, float rel_tol, int32_t iterations);
/* enqueue n_steps iterations (no-ops once done); asynchronous, graph-replayed */
int32_t rls_cgnr_step(rls_cgnr* s, int32_t n_steps);
/* synchronises.  If a resident launch of this plan (rls_cgnr_path 4 / 5) could not get all its workgroups onto the chip
 * within its wait bound -- another process on the device, a long kernel on another stream -- that launch was a no-op;
 * this call then re-runs the missing iterations on the per-iteration pipeline before it reports (state and result are
 * those of an undisturbed run), counts the event in `fallbacks`, and the plan stays on the pipeline.  The reference's
 * solve! has no such failure mode (src/RegularizedLeastSquares.jl:103-117), so none is surfaced here either. */
int32_t rls_cgnr_get_status(rls_cgnr* s, rls_cgnr_status* out_h);
/* rls_cgnr_step(s, n_steps) followed by rls_cgnr_get_status(s, out_h) in one call: what one `iterate` of the reference's
 * solve! loop needs (advance, then `done` / the convergence record for the callbacks) with ONE host synchronisation.
 * On a plan whose A lives in the register files (path 4) the kernel of this call is left LISTENING on a control block in
 * pinned host memory ("server mode"): the next rls_cgnr_step_status posts its command there instead of launching -- no launch
 * and no load of A per call -- and reads the status from the mailbox the kernel publishes into.  The kernel leaves when
 * nothing arrives for "resident_server_idle_us" (300), and EVERY other entry point of this library that touches the context
 * (a download, init, another plan, destroy, rls_sync ...) asks it to leave first, so nothing the caller does through the
 * library waits behind it; work the caller puts on the same stream by other means waits for the idle timeout at most.
 * A caller that touches the device between iterates twice in a row is served by the per-iteration pipeline until the next
 * init.  rls_tune_set("resident_server", 0) switches the mode off.  rls_fista_step_status: the same.
 * AHEAD (round 6): a listening kernel computes ONE iteration ahead of the next command, under the host's turnaround; nothing of that
 * iteration is published or stored before a command asks for it, and a kernel told to leave instead leaves without a write-back
 * (memory holds the state of the last command served).  rls_tune_set("resident_ahead", 0) selects the kernels that do not.
 * rls_free (rls_mi355x.h:memory) does not ask a listening kernel to leave: the pooled free is ordered behind it on the stream.
 * VISIBILITY: while a kernel listens, rls_*_get_status / rls_*_step_status answer from the host mirror the kernel published
 * into; the state vectors x, r, p it wrote are current for every LATER call on this context (each asks the kernel to leave
 * first), but a consumer outside this context -- another rls context, another stream or library reading the caller-owned
 * buffers -- sees them only after such a call (rls_ctx_sync is enough) or after the idle timeout ("resident_server_idle_us",
 * clamped to [1, 10000]).  For that reason contexts created on a BORROWED stream (rls_ctx_create_on_stream) start with
 * "resident_server" = 0; rls_tune_set opts in. */
int32_t rls_cgnr_step_status(rls_cgnr* s, int32_t n_steps, rls_cgnr_status* out_h);
/* K independent SMALL systems -- each with its own A: the distinct-A flavour of a multi-solve, one solver per problem under
 * Threads.@threads in the reference (docs/src/literate/howto/multi_threading.jl:8-17) -- advanced together in ONE launch, one
 * workgroup per plan.  Every plan is an ordinary single right-hand-side plan on rls_cgnr_path 8 (same context, same element type;
 * shapes may differ: the group runs on the tile of its largest member, so a member's bits may differ from its solo run's in the last
 * place); status, solution and further rls_cgnr_step calls per plan as usual.  rls_cgnr_init_step_group also runs every plan's
 * init (r = A^H b[k], x = 0, p = r: src/CGNR.jl:107-130) inside the same launch: a whole solve of K problems is ONE launch. */
int32_t rls_cgnr_step_group(rls_cgnr* const* plans, int32_t count, int32_t n_steps);
/* the statuses of `count` single right-hand-side plans of one context in one read-back (out_h: count structs) */
int32_t rls_cgnr_get_status_group(rls_cgnr* const* plans, int32_t count, rls_cgnr_status* out_h);
int32_t rls_cgnr_init_step_group(rls_cgnr* const* plans, const void* const* b, int32_t count, float lambda, float rel_tol,
                                 int32_t iterations, int32_t n_steps);
/* The distinct-A multi-solve as a QUEUE (BASELINE configs[3], distinct-A flavour; the reference's shape is one solver and one
 * operator per task under Threads.@threads, docs/src/literate/howto/multi_threading.jl:8-17; per-problem semantics
 * src/CGNR.jl:107-185): `count` independent problems of any shape, plans[k] an ordinary single right-hand-side plan on its own
 * operator (all on one context).  Problem k's init! and all `iterations` iterations are enqueued behind problem k - 1's on the
 * context's stream -- no host wait in between -- and ONE read-back at the end fills out_h[count].  b[k]: device pointers
 * (length M_k, or N_k for a Gram-only operator).  A resident launch lost to a busy device is re-run on the per-iteration pipeline
 * for that problem alone, as rls_cgnr_get_status does.  x, r, p of problem k are in plan k's caller-owned vectors afterwards. */
int32_t rls_cgnr_solve_queue(rls_cgnr* const* plans, const void* const* b, int32_t count, float lambda, float rel_tol,
                             int32_t iterations, rls_cgnr_status* out_h);
/* the same with HOST buffers: b_h[k] is staged through pinned memory (the plan's own, created on first use) and uploaded on the
 * stream ahead of problem k, x_h[k] (length N_k) is filled from a pinned download queued behind its last iteration; every copy is
 * asynchronous and the call synchronises ONCE, at the end. */
int32_t rls_cgnr_solve_queue_host(rls_cgnr* const* plans, const void* const* b_h, void* const* x_h, int32_t count, float lambda,
                                  float rel_tol, int32_t iterations, rls_cgnr_status* out_h);
/* Batched plan (BASELINE config 4, shared-A flavour; semantics of solve!(solver, B; scheduler =
 * MultiThreadingState), src/MultiThreading.jl:30-79): nrhs independent CGNR solves that share ONE pass over A
 * per iteration.  X, R, P, V: caller-owned N x nrhs column-major device matrices, leading dimension ldv;
 * B: M x nrhs, leading dimension ldb.  Every column keeps its own scalars and its own `done` flag
 * (columns retire independently); rls_cgnr_step advances all of them.  RLS_E_UNSUPPORTED when the shape
 * does not run on the one-pass kernel (then use one plan per column). */
int32_t rls_cgnr_create_batched(rls_operator* op, int32_t nrhs, void* X, void* R, void* P, void* V, int64_t ldv,
                                rls_cgnr** out);
int32_t rls_cgnr_init_batched(rls_cgnr* s, const void* B, int64_t ldb, float lambda, float rel_tol, int32_t iterations);
int32_t rls_cgnr_get_status_batched(rls_cgnr* s, rls_cgnr_status* out_h /* [nrhs] */);
/* measurement harness only: n_steps iterations with hipEvents around each kernel of the fused
 * pipeline; average device microseconds per launch of the one-pass normal-operator kernel and of
 * the partial-sum reduce kernel.  RLS_E_UNSUPPORTED when the shape runs on the two-GEMV path. */
int32_t rls_cgnr_step_profiled(rls_cgnr* s, int32_t n_steps, float* us_normal, float* us_reduce);
/* Which kernel sequence the next rls_cgnr_step call of this plan takes (a query; measurement harness and tests):
 * 0 = two GEMVs + update kernel, 1 = one-pass slab pipeline (two launches per iteration), 2 = Gram-mode pipeline
 * (one launch per iteration), 3 = batched matrix-core kernels, 4 = resident (the whole call in ONE launch, A held in
 * registers across iterations; needs A <= the register files), 5 = resident Gram mode (the same with AHA explicit and
 * held in registers: one in-kernel grid exchange per iteration), 6 = batched on an explicit AHA (ONE matrix-core product
 * V = AHA P per iteration), 7 = the same as ONE resident launch per step call (<= 8 ComplexF32 columns, N <= 2048: AHA in
 * the register files, the operand panel in LDS; calls of a single iteration take path 6), 8 = small system (M N s <= ~128 KiB,
 * matrix-free, single right-hand side): the whole step call as ONE single-workgroup launch with A in one CU's registers. */
int32_t rls_cgnr_path(rls_cgnr* s, int32_t* out);

/* ---------------------------------------------------------------------------------------------
 * fused FISTA.   replaces init!/iterate(::FISTA, ::FISTAState) src/FISTA.jl:110-129,139-185.
 * reg_kind selects the proximal map applied with rho*lambda (src/FISTA.jl:164); proj_kind the
 * projection applied every iteration (:166-168).
 * ------------------------------------------------------------------------------------------- */
enum { RLS_REG_NONE = 0, RLS_REG_L1 = 1, RLS_REG_L2 = 2, RLS_REG_L21 = 3, RLS_REG_TV = 4 };
enum { RLS_PROJ_NONE = 0, RLS_PROJ_REAL = 1, RLS_PROJ_POSITIVE = 2 };

typedef struct rls_fista_status {
  int32_t iteration;
  int32_t done;
  float theta, theta_old;
  float rel_res_norm; /* ||res|| / ||x0||   src/FISTA.jl:156                                      */
  float residual;     /* ||res||: solverconvergence  src/FISTA.jl:131                             */
  float norm_x0;
  int32_t fallbacks; /* as rls_cgnr_status.fallbacks */
} rls_fista_status;

/* x, x0, xold, res: caller-owned length-N device vectors.  The plan swaps x/xold internally by
 * pointer as the reference does (:144-146); rls_fista_solution() returns the current x. */
int32_t rls_fista_create(rls_operator* op, void* x, void* x0, void* xold, void* res, rls_fista** out);
/* solve!(solver::FISTA, B::AbstractMatrix) (src/MultiThreading.jl:30-79): nrhs columns advance together and share
 * A -- T = A Y and V = A^H T as skinny GEMMs on the matrix cores, then one workgroup per column for
 * src/FISTA.jl:153-180 with that column's own scalars (theta, rel_res_norm, done: a column that is done stops
 * changing, the others go on).  x, x0, xold, res: N x nrhs, columns ldv elements apart; column j's current
 * solution is in x when its iteration count is even, in xold when odd (the pointer swap of :144-146).
 * RLS_E_UNSUPPORTED unless the operator is matrix-free with M, N multiples of 16.  rls_fista_set_reg,
 * rls_fista_step and rls_fista_destroy apply unchanged.
 * With an explicit AHA on the operator (the constructors' default for a dense matrix, src/FISTA.jl:58) every product is
 * ONE pass over AHA; with <= 8 ComplexF32 columns, N <= 2048 (a multiple of 16), no gradient restart and an elementwise
 * regulariser (none, L1, L2) a whole rls_fista_step call of more than one iteration is ONE resident launch -- AHA in the
 * register files, every workgroup advancing its own 8 rows of all columns, the rows of the next extrapolated point
 * the only exchange (rls_fista_path 7).  A launch that cannot get its grid onto the chip changes nothing;
 * rls_fista_get_status_batched re-runs what it left undone on the streaming kernels and counts it in `fallbacks`. */
int32_t rls_fista_create_batched(rls_operator* op, int32_t nrhs, void* x, void* x0, void* xold, void* res, int64_t ldv,
                                 rls_fista** out);
int32_t rls_fista_init_batched(rls_fista* s, const void* B, int64_t ldb, float rho, float theta, float rel_tol,
                               int32_t iterations, int32_t restart_gradient);
int32_t rls_fista_get_status_batched(rls_fista* s, rls_fista_status* out_h /* [nrhs] */);
int32_t rls_fista_destroy(rls_fista* s);
int32_t rls_fista_set_reg(rls_fista* s, int32_t reg_kind, float lambda, int64_t l21_slices, int32_t proj_kind);
/* prox!(::TVRegularization) inside the plan (src/FISTA.jl:164 -> src/proximalMaps/ProxTV.jl:64-125; shape / dims (0-based) /
 * iterations_tv as rls_prox_tv_fgp): the FGP loop is ONE single-workgroup launch between the two halves of the update, skipped
 * once the plan is done.  RLS_E_UNSUPPORTED when the image does not fit that kernel (1-D / 2-D images up to 8192 Float32 or 4096
 * ComplexF32 pixels, other geometries up to 2048) or on a batched plan -- the host then drives FISTA from the primitives.
 * The plan runs on the two-product path; row-sharded plans (rls_fista_step_local_b, rls_fista_step_rowsharded) take the same
 * launches behind their all-reduce.  proj_kind: the Positive / Real projection applied after the prox (:166-168). */
int32_t rls_fista_set_reg_tv(rls_fista* s, float lambda, int32_t ndims, const int64_t* shape, int32_t ntv, const int32_t* dims,
                             int32_t iterations_tv, int32_t proj_kind);
int32_t rls_fista_init(rls_fista* s, const void* b, float rho, float theta, float rel_tol, int32_t iterations,
                       int32_t restart_gradient);
/* optional warm start x0 != 0 (init!(solver, b; x0), src/FISTA.jl:110,120): call right after init.  x_init: n = N
 * elements (a scalar x0 is broadcast by the caller, as `state.x .= x0` does); RLS_E_INVALID on any other length */
int32_t rls_fista_set_start(rls_fista* s, const void* x_init, int64_t n);
int32_t rls_fista_step(rls_fista* s, int32_t n_steps);
/* which kernel sequence the next rls_fista_step call takes (the codes of rls_cgnr_path): 0 = two GEMVs + update kernel,
 * 1 = one-pass slab pipeline, 2 = Gram-mode pipeline, 3 = batched matrix-core kernels, 4 = resident (one launch per
 * call, A in registers), 5 = resident Gram mode (one launch per call, AHA in registers), 7 = batched resident Gram mode
 * (rls_fista_create_batched), 8 = small system (one single-workgroup
 * launch per call, A in one CU's registers: fista_small_kernel) */
int32_t rls_fista_path(rls_fista* s, int32_t* out);
int32_t rls_fista_get_status(rls_fista* s, rls_fista_status* out_h); /* synchronises; recovers lost resident launches as rls_cgnr_get_status */
int32_t rls_fista_step_status(rls_fista* s, int32_t n_steps, rls_fista_status* out_h); /* step + status, one synchronisation */
int32_t rls_fista_solution(rls_fista* s, void** x_out); /* device pointer currently holding state.x */

/* ---------------------------------------------------------------------------------------------
 * fused cg!   replaces IterativeSolvers.cg!(x, AHA + rho*I, b; maxiter, reltol, statevars)
 * call site src/ADMM.jl:244 (identity regTrafo: compositeAHA u = AHA u + rho u, :84,141-159).
 * u, r, c: the CGStateVariables scratch (src/ADMM.jl:129).
 * ------------------------------------------------------------------------------------------- */
typedef struct rls_cg_status {
  int32_t iterations; /* CG iterations actually performed                                        */
  float residual;     /* ||r|| at exit                                                            */
  float tol;          /* max(reltol * ||r0||, 0)                                                  */
  int32_t fallbacks;  /* as rls_cgnr_status.fallbacks                                                 */
} rls_cg_status;
int32_t rls_cg_create(rls_operator* op, void* u, void* r, void* c, rls_cg** out);
/* batched plan (shared A; solve!(solver::ADMM, B) with the shared-A scheduler, src/MultiThreading.jl:30-79): U, R, C are
 * N x nrhs scratch matrices, columns ldv elements apart.  Used through rls_admm_create / rls_admm_step, whose params then
 * point to N x nrhs matrices with the same ldv; every column keeps its own scalars, inner-iteration count and `done`.
 * RLS_E_UNSUPPORTED unless the operator is matrix-free with M, N multiples of 16. */
int32_t rls_cg_create_batched(rls_operator* op, int32_t nrhs, void* U, void* R, void* C, int64_t ldv, rls_cg** out);
int32_t rls_cg_destroy(rls_cg* s);
/* solves (AHA + rho I) x = b starting from x (warm start); asynchronous */
int32_t rls_cg_solve(rls_cg* s, void* x, const void* b, float rho, int32_t maxiter, float reltol);
/* synchronises.  When the solve ran as ONE resident launch (rls_cg_path 4 / 5) and that launch timed out, x still holds the
 * warm start: this call then repeats the solve on the per-iteration pipeline (`fallbacks`).  A caller that consumes x
 * asynchronously therefore calls this first whenever rls_cg_path reports 4 or 5; inside an ADMM plan (rls_admm_*) the
 * library does the equivalent itself. */
int32_t rls_cg_get_status(rls_cg* s, rls_cg_status* out_h);
/* which kernel sequence the next rls_cg_solve of this plan takes (the codes of rls_cgnr_path) */
int32_t rls_cg_path(rls_cg* s, int32_t* out);

/* ADMM elementwise steps for an identity regTrafo (the default opEye, src/ADMM.jl:84), fused:
 *   rls_admm_pre :  beta = (accumulate ? beta : beta_y) + rho (z - u) ; xold = x            src/ADMM.jl:236-243
 *   rls_admm_post:  u += x - z and the seven norms of the convergence bookkeeping src/ADMM.jl:265-299 in one
 *                   launch; out_h[6] = { Delta, ||z-zold||, eps_pri, r, ||u||, ||x-xold|| } (synchronises;
 *                   s = rho*out[1], eps_dua = rho*out[4]) */
int32_t rls_admm_pre(rls_ctx* ctx, int32_t dtype, int64_t n, void* beta, const void* beta_y, const void* z,
                     const void* u, const void* x, void* xold, float rho, int32_t accumulate);
int32_t rls_admm_post(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, const void* xold, const void* z,
                      const void* zold, void* u, float* out_h);

/* Whole ADMM outer iterations on the device for ONE regulariser with the identity regTrafo and vary_rho = :none
 * (src/ADMM.jl:230-322; `done`/`converged` :324-330 evaluated on the device in Float32, so no host read-back
 * between outer iterations).  Per outer iteration: the warm-started cg! of the plan `cg` on (AHA + rho I) with the
 * right-hand side beta = beta_y + rho (z - u) formed inside its start kernel (:236-244), projections on x
 * (:246-248), z = prox(x + u, lambda / (2 rho)) (:251-263; L1 / L2 inline, TV as one single-workgroup FGP launch),
 * u += x - z and the residual norms (:265-299) in one launch.  Once `done` is set every later launch of the plan is
 * a no-op.  rls_admm_init returns RLS_E_UNSUPPORTED when the regulariser cannot run inside the plan (callers then
 * use rls_admm_pre / rls_cg_solve / prox / rls_admm_post).  z0 holds z at init; z alternates between z0 and z1
 * (current z after k iterations: k odd -> z1).  All vectors are device pointers owned by the caller. */
typedef struct rls_admm rls_admm;
typedef struct rls_admm_params {
  void *x, *xold, *beta, *beta_y, *z0, *z1, *u;
  float rho;          /* fixed between two rls_admm_init calls */
  float sigma_abs;    /* sqrt(length(b)) * absTol   src/ADMM.jl:214 */
  float rel_tol;
  int32_t iterations, iterations_cg;
  float tol_inner;
  int32_t reg_kind;   /* RLS_REG_NONE (also for rho == 0, :260) / RLS_REG_L1 / RLS_REG_L2 / RLS_REG_TV */
  float prox_lambda;  /* lambda / (2 rho), formed by the caller in Float32 (:261) */
  int32_t proj_kind;  /* RLS_PROJ_*; must be RLS_PROJ_NONE with RLS_REG_TV */
  int32_t tv_ndims, tv_ntv, tv_iterations;
  int32_t tv_dims[4];
  int64_t tv_shape[4];
} rls_admm_params;
typedef struct rls_admm_status {
  int32_t iteration, done;
  float rk, sk, eps_pri, eps_dua, delta; /* of the last completed iteration */
  int32_t cg_iterations;
  int32_t fallbacks; /* resident cg! launches lost; the outer iterations behind them were re-run (rls_admm_get_status) */
} rls_admm_status;
int32_t rls_admm_create(rls_cg* cg, rls_admm** out);
int32_t rls_admm_destroy(rls_admm* a);
int32_t rls_admm_init(rls_admm* a, const rls_admm_params* p);
int32_t rls_admm_step(rls_admm* a, int32_t n_outer); /* asynchronous; stops enqueuing at `iterations` */
/* synchronises; log_h (nullable) receives min(iteration, log_records) records of 8 floats:
 * Delta, sk, eps_pri, rk, eps_dua, inner cg! iterations, 0, 0 */
int32_t rls_admm_get_status(rls_admm* a, rls_admm_status* out_h, float* log_h, int32_t log_records);
/* rls_admm_step(a, n_outer) followed by rls_admm_get_status in one call (one synchronisation) */
int32_t rls_admm_step_status(rls_admm* a, int32_t n_outer, rls_admm_status* out_h, float* log_h, int32_t log_records);
/* batched plans: out_h[nrhs]; log_h (nullable) = nrhs blocks of log_records records, column after column */
int32_t rls_admm_get_status_batched(rls_admm* a, rls_admm_status* out_h, float* log_h, int32_t log_records);

/* ---------------------------------------------------------------------------------------------
 * device pieces of the nested regularisation terms and of the plug-and-play input transforms
 *   rls_gather / rls_scatter   z = view(x, findall(mask)) and back          src/Regularization/MaskedRegularization.jl:27-31
 *   rls_stats                  out_h[5] = min, max, sum, sum of squares of the REAL parts, max |x| (modulus)
 *                              (minimum/maximum src/Transforms.jl:9, mean/std :39, maximum(abs.(x))
 *                              src/Regularization/ScaledRegularization.jl:55); synchronises
 *   rls_shift_scale            inverse = 0: x = (x - shift) / scale ; 1: x = x * scale + shift      src/Transforms.jl:11-16,41-46
 *   rls_clamp                  x = clamp(x, lo, hi)                                                 src/Transforms.jl:59
 *   rls_restore_outside        out[m] = orig[m] where orig < lo or orig >= hi                       src/Transforms.jl:53,62-66
 *   rls_complex_split / merge  real.(x), imag.(x) and back       src/Regularization/PlugAndPlayRegularization.jl:24-31
 * idx: device int32, 0-based.  Float32 vectors unless a dtype is given.
 * ------------------------------------------------------------------------------------------- */
int32_t rls_gather(rls_ctx* ctx, int32_t dtype, int64_t m, const int32_t* idx, const void* x, void* out);
int32_t rls_scatter(rls_ctx* ctx, int32_t dtype, int64_t m, const int32_t* idx, const void* in, void* x);
int32_t rls_stats(rls_ctx* ctx, int32_t dtype, int64_t n, const void* x, double* out_h);
int32_t rls_shift_scale(rls_ctx* ctx, int64_t n, float* x, float shift, float scale, int32_t inverse);
int32_t rls_clamp(rls_ctx* ctx, int64_t n, float* x, float lo, float hi);
int32_t rls_restore_outside(rls_ctx* ctx, int64_t n, float* out, const float* orig, float lo, float hi);
int32_t rls_complex_split(rls_ctx* ctx, int64_t n, const void* z, float* re, float* im);
int32_t rls_complex_merge(rls_ctx* ctx, int64_t n, const float* re, const float* im, void* z);

/* ---------------------------------------------------------------------------------------------
 * row-sharded operation (BASELINE config 5).  One process per GPU holds rows
 * [r*M/P, (r+1)*M/P) of A repacked contiguous; x, r, p, v and all scalars are replicated.
 * The plan is split at the one exchange step so the host can run the all-reduce (RCCL via
 * torch.distributed / a Julia RCCL binding) on the ctx stream between the two halves:
 *     rls_cgnr_step_local_a(s)  : t = A_g p ; v_partial = A_g^H t
 *     allreduce(sum, v, N elements)  -- caller, same stream
 *     rls_cgnr_step_local_b(s)  : scalars + vector updates on the replicated state
 * init is split the same way (r_partial = A_g^H b_g ; allreduce ; finish).
 * ------------------------------------------------------------------------------------------- */
int32_t rls_cgnr_init_local_a(rls_cgnr* s, const void* b_local, float lambda, float rel_tol, int32_t iterations);
int32_t rls_cgnr_init_local_b(rls_cgnr* s);
int32_t rls_cgnr_step_local_a(rls_cgnr* s);
int32_t rls_cgnr_step_local_b(rls_cgnr* s);
/* FISTA (src/FISTA.jl:110-185) on a row shard, split at the same exchange step: the caller all-reduces x0 (= A^H b)
 * between init_local_a and init_local_b, and `res` (= A^H A y) between step_local_a and step_local_b; the gradient
 * step, prox and momentum run replicated.  x0 and res are the vectors given to rls_fista_create. */
int32_t rls_fista_init_local_a(rls_fista* s, const void* b_local);
int32_t rls_fista_init_local_b(rls_fista* s, float rho, float theta, float rel_tol, int32_t iterations,
                               int32_t restart_gradient);
int32_t rls_fista_step_local_a(rls_fista* s);
int32_t rls_fista_step_local_b(rls_fista* s);
/* cg! (call site src/ADMM.jl:244) on a row shard: the caller all-reduces c (given to rls_cg_create) after every
 * rls_cg_local_apply.  apply(x) is the warm-start product AHA x; apply(NULL) is AHA u of one iteration and is
 * skipped on the device once the solve is done, like the update, so a fixed `maxiter` iterations can be enqueued. */
int32_t rls_cg_local_apply(rls_cg* s, const void* x);
int32_t rls_cg_local_start(rls_cg* s, const void* x, const void* b, float rho, int32_t maxiter, float reltol);
int32_t rls_cg_local_update(rls_cg* s, void* x);

/* ---------------------------------------------------------------------------------------------
 * Communicator: the exchange step of the row-partitioned mode inside the library, for hosts that drive several
 * GPUs from ONE process (the Julia extension: one task per GPU; fan-out site src/MultiThreading.jl:60-78; the
 * replicated init / iterate halves are src/CGNR.jl:107-130, :143-178).  Rank r owns context ctxs[r] (pass NULL to
 * have the communicator create one context per entry of devices[]; rls_comm_ctx returns them) and everything a rank
 * enqueues -- its local halves and its share of the collective -- goes to that context's stream.
 *   RLS_COMM_RCCL   : ncclAllReduce over xGMI, one group call per all-reduce; needs one distinct device per rank.
 *   RLS_COMM_DIRECT : one-shot direct-write all-reduce (peer stores into per-rank slots, stream events, the slots
 *                     summed in rank order: every rank adds the same numbers in the same order).  Ranks may share
 *                     a device, which is how the config-5 schedule runs on a one-GPU box.
 *   RLS_COMM_AUTO   : RCCL when the devices are distinct, DIRECT otherwise.
 * rls_comm_create probes hipDeviceCanAccessPeer between every pair of distinct devices.  A requested DIRECT transport whose
 * ranks have a device each but lack peer access somewhere is DROPPED to RCCL (rls_comm_transport tells; rls_comm_peer_access
 * reports the matrix: 1 can store into / same device, 0 cannot, -1 the query failed) -- never a fault at the first exchange.
 * ------------------------------------------------------------------------------------------- */
typedef struct rls_comm rls_comm;
enum { RLS_COMM_AUTO = 0, RLS_COMM_RCCL = 1, RLS_COMM_DIRECT = 2 };
int32_t rls_comm_create(int32_t nranks, const int32_t* devices, rls_ctx* const* ctxs, int32_t transport, rls_comm** out);
int32_t rls_comm_destroy(rls_comm* comm);
int32_t rls_comm_size(rls_comm* comm);
int32_t rls_comm_transport(rls_comm* comm);
/* out_matrix_h[nranks * nranks] (may be NULL), *out_requested_h = the transport rls_comm_create was asked for */
int32_t rls_comm_peer_access(rls_comm* comm, int32_t* out_matrix_h, int32_t* out_requested_h);
int32_t rls_comm_ctx(rls_comm* comm, int32_t rank, rls_ctx** out);
int32_t rls_comm_sync(rls_comm* comm); /* waits for every rank's stream */
/* in place: rank_bufs[r] (device pointer on rank r's device, n elements of dtype) <- sum over ranks; asynchronous */
int32_t rls_allreduce_sum(rls_comm* comm, void* const* rank_bufs, int64_t n, int32_t dtype);
/* CGNR on a row-partitioned A: plans[r] was created on rank r's context over that rank's row shard (repacked
 * contiguous), b_parts[r] is that rank's slice of b.  init = every rank's rls_cgnr_init_local_a, ONE all-reduce of
 * A^H b, every rank's rls_cgnr_init_local_b; a step = local_a (t_g = A_g p, v_g = A_g^H t_g), ONE all-reduce of v,
 * local_b (the replicated update).  Scalars need no collective: every rank reduces identical vectors in the same
 * order.  `done` is a device flag replicated on every rank, so the collective count never depends on the data. */
int32_t rls_cgnr_init_rowsharded(rls_comm* comm, rls_cgnr* const* plans, const void* const* b_parts, float lambda,
                                 float rel_tol, int32_t iterations);
int32_t rls_cgnr_step_rowsharded(rls_comm* comm, rls_cgnr* const* plans, int32_t n_steps);
/* The row-sharded calls fan out inside the library: one host worker thread per rank (the Threads.@threads of
 * src/MultiThreading.jl:60-78) walks that rank's share of the whole call, the ranks meeting at a host barrier between the
 * two halves of each all-reduce, so the host side of an iteration costs ONE rank's launches.  on = 0: the calling thread
 * drives every rank in turn (also: environment RLS_COMM_THREADS=0 at communicator creation).  Results are identical. */
int32_t rls_comm_set_threads(rls_comm* comm, int32_t on);
/* measurement only: seconds each rank's worker thread has spent enqueueing (inside the phases of the row-sharded calls,
 * host barriers and idle time excluded) since the previous call; out_h[nranks] */
int32_t rls_comm_debug_busy_seconds(rls_comm* comm, double* out_h);
/* FISTA on a row-partitioned A (src/FISTA.jl:110-185; call sites of the distributed step :114, :152): plans[r] =
 * rls_fista_create (+ rls_fista_set_reg) on rank r's shard operator.  init: x0 = sum_g A_g^H b_g (ONE all-reduce), then
 * every rank's rls_fista_init_local_b; a step: res_g = A_g^H A_g y, ONE all-reduce of res, the replicated gradient step /
 * prox / momentum / `done`.  Status and solution through any rank's plan (rls_fista_get_status, rls_fista_solution). */
int32_t rls_fista_init_rowsharded(rls_comm* comm, rls_fista* const* plans, const void* const* b_parts, float rho, float theta,
                                  float rel_tol, int32_t iterations, int32_t restart_gradient);
int32_t rls_fista_step_rowsharded(rls_comm* comm, rls_fista* const* plans, int32_t n_steps);
/* ADMM on a row-partitioned A (src/ADMM.jl:191-330, distributed step = the operator applies inside cg!, :244): plans[r] =
 * rls_cg_create + rls_admm_create + rls_admm_init on rank r's shard operator, the same parameters on every rank
 * (sigma_abs from the length of the whole b; the caller has set x, z0, u as init! does, :199-206).
 * init_rowsharded: beta_y = sum_g A_g^H b_g (ONE all-reduce, :198).  step_rowsharded: n_outer outer iterations, each
 * iterations_cg + 1 all-reduces of the cg! scratch c (the inner solve always runs its iterations_cg half-step pairs; its
 * convergence and the plan's `done` are replicated device flags, so the collective count never depends on the data);
 * z / u / prox / `converged` are the single-GPU plan's kernels, replicated.  Status through rls_admm_get_status. */
int32_t rls_admm_init_rowsharded(rls_comm* comm, rls_admm* const* plans, const void* const* b_parts);
int32_t rls_admm_step_rowsharded(rls_comm* comm, rls_admm* const* plans, int32_t n_outer);

#ifdef __cplusplus
}
#endif
#endif /* RLS_MI355X_H */
