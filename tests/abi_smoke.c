/* Plain-C driver of the drop-in boundary: no Python, no C++, no torch in the process -- exactly what a host that binds
 * include/rls_mi355x.h through an FFI (Julia ccall: INTEGRATION.md) executes.  create -> init -> step -> status for
 *   1. CGNR   (src/CGNR.jl:107-130, :143-178)        256 x 128 Float32, lambda = 1e-2, 10 iterations, every iterate checked
 *   2. FISTA  (src/FISTA.jl:110-129, :139-185) + L1  256 x 128 Float32, 25 iterations; + TV + Positive with the FGP prox inside
 *      the plan (rls_fista_set_reg_tv), 15 iterations
 *   3. CGNR at the headline shape 4096 x 2048 ComplexF32 (the resident one-launch path when the device offers it)
 *   4. row-partitioned CGNR through the library's communicator (rls_comm_*, rls_cgnr_*_rowsharded): 1, 2 and 4 ranks
 *      sharing device 0 (direct transport) and, where RCCL loads, the RCCL transport with one rank
 * against double-precision restatements of the same recurrences written out below (test code: the checker).
 * Build + run: tests/test_gpu_parity.py::test_plain_c_program_drives_the_abi (gcc, -m gpu).  Exit code 0 = all good. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rls_mi355x.h"

#define CHECK(expr)                                                                                         \
  do {                                                                                                      \
    int32_t st_ = (expr);                                                                                   \
    if (st_ != 0) {                                                                                         \
      fprintf(stderr, "FAIL %s:%d  %s -> %d (%s)\n", __FILE__, __LINE__, #expr, (int)st_, rls_last_error_string(g_ctx)); \
      exit(1);                                                                                              \
    }                                                                                                       \
  } while (0)
#define REQUIRE(cond, ...)                                   \
  do {                                                       \
    if (!(cond)) {                                           \
      fprintf(stderr, "FAIL %s:%d  ", __FILE__, __LINE__);   \
      fprintf(stderr, __VA_ARGS__);                          \
      fprintf(stderr, "\n");                                 \
      exit(1);                                               \
    }                                                        \
  } while (0)

static rls_ctx* g_ctx = NULL;

/* deterministic zero-mean pseudo-normal numbers (sum of 4 uniforms), no libc rand */
static uint64_t g_seed = 0x9E3779B97F4A7C15ull;
static double unif(void) {
  g_seed = g_seed * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(g_seed >> 11) / 9007199254740992.0;
}
static float gauss(void) { return (float)((unif() + unif() + unif() + unif() - 2.0) * 1.7320508); }

static void* dev_upload(rls_ctx* ctx, const void* h, size_t bytes) {
  void* d = NULL;
  CHECK(rls_malloc(ctx, bytes, &d));
  if (h) CHECK(rls_memcpy_h2d(ctx, d, h, bytes));
  return d;
}

static double rel_err_f(const float* got, const double* want, int64_t n) {
  double d = 0, s = 0;
  for (int64_t i = 0; i < n; ++i) {
    d += (got[i] - want[i]) * (got[i] - want[i]);
    s += want[i] * want[i];
  }
  return sqrt(d / (s > 0 ? s : 1));
}

/* ---- double-precision restatements (real case), column-major A ------------------------------------------- */
static void mul_n(const float* A, int64_t M, int64_t N, const double* x, double* y) {
  for (int64_t i = 0; i < M; ++i) y[i] = 0;
  for (int64_t j = 0; j < N; ++j)
    for (int64_t i = 0; i < M; ++i) y[i] += (double)A[i + j * M] * x[j];
}
static void mul_t(const float* A, int64_t M, int64_t N, const double* y, double* x) {
  for (int64_t j = 0; j < N; ++j) {
    double s = 0;
    for (int64_t i = 0; i < M; ++i) s += (double)A[i + j * M] * y[i];
    x[j] = s;
  }
}
static double dot(const double* a, const double* b, int64_t n) {
  double s = 0;
  for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

typedef struct {
  int64_t M, N;
  const float* A;
  double *x, *r, *p, *v, *t;
  double lambda;
} cgnr_ref;
static void cgnr_ref_init(cgnr_ref* c, const float* b) { /* src/CGNR.jl:107-130 */
  double* bd = (double*)malloc(sizeof(double) * c->M);
  for (int64_t i = 0; i < c->M; ++i) bd[i] = b[i];
  mul_t(c->A, c->M, c->N, bd, c->r);
  for (int64_t j = 0; j < c->N; ++j) {
    c->x[j] = 0;
    c->p[j] = c->r[j];
  }
  free(bd);
}
static void cgnr_ref_iterate(cgnr_ref* c) { /* src/CGNR.jl:151-176 */
  mul_n(c->A, c->M, c->N, c->p, c->t);
  mul_t(c->A, c->M, c->N, c->t, c->v);
  const double zeta = dot(c->r, c->r, c->N);
  double nv = dot(c->p, c->v, c->N);
  if (c->lambda > 0) nv += c->lambda * dot(c->p, c->p, c->N);
  const double alpha = zeta / nv;
  for (int64_t j = 0; j < c->N; ++j) {
    c->x[j] += alpha * c->p[j];
    c->r[j] -= alpha * c->v[j];
    if (c->lambda > 0) c->r[j] -= c->lambda * alpha * c->p[j];
  }
  const double beta = dot(c->r, c->r, c->N) / zeta;
  for (int64_t j = 0; j < c->N; ++j) c->p[j] = beta * c->p[j] + c->r[j];
}

static void test_cgnr_and_fista_small(void) {
  const int64_t M = 256, N = 128;
  float* A = (float*)malloc(sizeof(float) * M * N);
  float* b = (float*)malloc(sizeof(float) * M);
  double* xt = (double*)malloc(sizeof(double) * N);
  for (int64_t i = 0; i < M * N; ++i) A[i] = gauss();
  for (int64_t j = 0; j < N; ++j) xt[j] = gauss();
  {
    double* bd = (double*)malloc(sizeof(double) * M);
    mul_n(A, M, N, xt, bd);
    for (int64_t i = 0; i < M; ++i) b[i] = (float)bd[i];
    free(bd);
  }
  void* Ad = dev_upload(g_ctx, A, sizeof(float) * M * N);
  void* bd_ = dev_upload(g_ctx, b, sizeof(float) * M);
  rls_operator* op = NULL;
  CHECK(rls_operator_create(g_ctx, RLS_F32, M, N, Ad, M, &op));

  /* ---- CGNR ---- */
  void *x = dev_upload(g_ctx, NULL, 4 * N), *r = dev_upload(g_ctx, NULL, 4 * N), *p = dev_upload(g_ctx, NULL, 4 * N),
       *v = dev_upload(g_ctx, NULL, 4 * N);
  rls_cgnr* plan = NULL;
  CHECK(rls_cgnr_create(op, x, r, p, v, &plan));
  REQUIRE(rls_cgnr_step(plan, 1) == RLS_E_STATE, "step before init must be RLS_E_STATE");
  CHECK(rls_cgnr_init(plan, bd_, 1e-2f, 0.f, 10));
  cgnr_ref c = {M, N, A, NULL, NULL, NULL, NULL, NULL, 1e-2};
  c.x = (double*)calloc(N, 8); c.r = (double*)calloc(N, 8); c.p = (double*)calloc(N, 8); c.v = (double*)calloc(N, 8);
  c.t = (double*)calloc(M, 8);
  cgnr_ref_init(&c, b);
  float* xh = (float*)malloc(4 * N);
  rls_cgnr_status st;
  for (int it = 1; it <= 10; ++it) {
    CHECK(rls_cgnr_step(plan, 1));
    cgnr_ref_iterate(&c);
    CHECK(rls_cgnr_get_status(plan, &st));
    REQUIRE(st.iteration == it, "iteration %d != %d", st.iteration, it);
    CHECK(rls_memcpy_d2h(g_ctx, xh, x, 4 * N));
    const double e = rel_err_f(xh, c.x, N);
    REQUIRE(e < 1e-5, "CGNR iterate %d: relative error %.3e", it, e);
  }
  REQUIRE(st.done == 1, "done after 10 of 10 iterations");
  CHECK(rls_cgnr_step(plan, 3)); /* past the end: no-ops on the device */
  CHECK(rls_cgnr_get_status(plan, &st));
  REQUIRE(st.iteration == 10, "iteration stays at 10");
  printf("CGNR 256x128 f32: 10 iterates within 1e-5, residual %.3e\n", st.residual);
  CHECK(rls_cgnr_destroy(plan));

  /* ---- FISTA + L1 (src/FISTA.jl:139-185), rho from ||A||_F^2 (an upper bound of sigma_max^2) ---- */
  double fro2 = 0;
  for (int64_t i = 0; i < M * N; ++i) fro2 += (double)A[i] * A[i];
  const float rho = (float)(0.9 / fro2) * 8.f, lam = 5.0f;
  void *fx = dev_upload(g_ctx, NULL, 4 * N), *fx0 = dev_upload(g_ctx, NULL, 4 * N), *fxo = dev_upload(g_ctx, NULL, 4 * N),
       *fres = dev_upload(g_ctx, NULL, 4 * N);
  rls_fista* fp = NULL;
  CHECK(rls_fista_create(op, fx, fx0, fxo, fres, &fp));
  CHECK(rls_fista_set_reg(fp, RLS_REG_L1, lam, 1, RLS_PROJ_NONE));
  CHECK(rls_fista_init(fp, bd_, rho, 1.f, 0.f, 25, 0));
  CHECK(rls_fista_step(fp, 25));
  rls_fista_status fs;
  CHECK(rls_fista_get_status(fp, &fs));
  REQUIRE(fs.iteration == 25 && fs.done == 1, "FISTA iteration %d done %d", fs.iteration, fs.done);
  void* xs = NULL;
  CHECK(rls_fista_solution(fp, &xs));
  CHECK(rls_memcpy_d2h(g_ctx, xh, xs, 4 * N));
  { /* double restatement: x0 = A'b; loop: res = AHA y - x0; x = prox_l1(y - rho res, rho lam); theta; y = x + c (x - xold) */
    double *fx0d = (double*)calloc(N, 8), *y = (double*)calloc(N, 8), *xk = (double*)calloc(N, 8), *xo = (double*)calloc(N, 8),
           *res = (double*)calloc(N, 8), *t = (double*)calloc(M, 8), *bd = (double*)calloc(M, 8);
    for (int64_t i = 0; i < M; ++i) bd[i] = b[i];
    mul_t(A, M, N, bd, fx0d);
    double theta = 1, theta_old = 1;
    const double eps = 1.1920929e-07;
    for (int it = 0; it < 25; ++it) {
      for (int64_t j = 0; j < N; ++j) { /* :144-148 */
        const double xn = xk[j];
        y[j] = xk[j] * ((theta_old - 1) / theta + 1) + xo[j] * ((1 - theta_old) / theta);
        xo[j] = xn;
      }
      mul_n(A, M, N, y, t);
      mul_t(A, M, N, t, res);
      for (int64_t j = 0; j < N; ++j) {
        res[j] -= fx0d[j];
        const double u = y[j] - (double)rho * res[j], a = fabs(u), sh = a - (double)rho * lam > 0 ? a - (double)rho * lam : 0;
        xk[j] = sh * (u + eps) / (a + eps); /* ProxL1.jl:18-22 */
      }
      theta_old = theta;
      theta = (1 + sqrt(1 + 4 * theta_old * theta_old)) / 2;
    }
    const double e = rel_err_f(xh, xk, N);
    REQUIRE(e < 1e-5, "FISTA solution: relative error %.3e", e);
    printf("FISTA+L1 256x128 f32: 25 iterations within 1e-5 (%.2e), rel_res_norm %.3e\n", e, fs.rel_res_norm);
    free(fx0d); free(y); free(xk); free(xo); free(res); free(t); free(bd);
  }
  /* ---- FISTA + TV + Positive with the prox INSIDE the plan (rls_fista_set_reg_tv; src/FISTA.jl:164-168,
   *      src/proximalMaps/ProxTV.jl:89-125): x seen as a 1-D signal of N samples, 10 FGP iterations ---- */
  {
    const float lam_tv = 2.0f;
    const int64_t shape1[1] = {N};
    const int32_t dims1[1] = {0};
    CHECK(rls_fista_set_reg_tv(fp, lam_tv, 1, shape1, 1, dims1, 10, RLS_PROJ_POSITIVE));
    CHECK(rls_fista_init(fp, bd_, rho, 1.f, 0.f, 15, 0));
    CHECK(rls_fista_step(fp, 15));
    CHECK(rls_fista_get_status(fp, &fs));
    REQUIRE(fs.iteration == 15 && fs.done == 1, "FISTA + TV iteration %d done %d", fs.iteration, fs.done);
    CHECK(rls_fista_solution(fp, &xs));
    CHECK(rls_memcpy_d2h(g_ctx, xh, xs, 4 * N));
    double *fx0d = (double*)calloc(N, 8), *y = (double*)calloc(N, 8), *xk = (double*)calloc(N, 8), *xo = (double*)calloc(N, 8),
           *res = (double*)calloc(N, 8), *t = (double*)calloc(M, 8), *bd = (double*)calloc(M, 8), *u = (double*)calloc(N, 8),
           *pq = (double*)calloc(N, 8), *rs = (double*)calloc(N, 8), *po = (double*)calloc(N, 8), *xt = (double*)calloc(N, 8);
    for (int64_t i = 0; i < M; ++i) bd[i] = b[i];
    mul_t(A, M, N, bd, fx0d);
    double theta = 1, theta_old = 1;
    for (int it = 0; it < 15; ++it) {
      for (int64_t j = 0; j < N; ++j) {
        const double xn = xk[j];
        y[j] = xk[j] * ((theta_old - 1) / theta + 1) + xo[j] * ((1 - theta_old) / theta);
        xo[j] = xn;
      }
      mul_n(A, M, N, y, t);
      mul_t(A, M, N, t, res);
      for (int64_t j = 0; j < N; ++j) {
        res[j] -= fx0d[j];
        u[j] = y[j] - (double)rho * res[j];
      }
      /* prox_TV(u, rho * lam): FGP on the forward differences g[i] = x[i] - x[i + 1] (no boundary row), step 1 / (8 lam) */
      const double l = (double)rho * lam_tv;
      const int64_t ng = N - 1;
      for (int64_t i = 0; i < ng; ++i) pq[i] = rs[i] = po[i] = 0;
      double tt = 1;
      for (int k = 0; k < 10; ++k) {
        for (int64_t j = 0; j < N; ++j) xt[j] = u[j] - l * ((j < ng ? rs[j] : 0) - (j > 0 ? rs[j - 1] : 0)); /* x - l grad' rs */
        for (int64_t i = 0; i < ng; ++i) {
          double q = rs[i] + (xt[i] - xt[i + 1]) / (8 * l);
          q = q / (fabs(q) > 1 ? fabs(q) : 1);
          po[i] = pq[i];
          pq[i] = q;
        }
        const double to = tt;
        tt = (1 + sqrt(1 + 4 * to * to)) / 2;
        for (int64_t i = 0; i < ng; ++i) rs[i] = (1 + (to - 1) / tt) * pq[i] - ((to - 1) / tt) * po[i];
      }
      for (int64_t j = 0; j < N; ++j) {
        const double v = u[j] - l * ((j < ng ? pq[j] : 0) - (j > 0 ? pq[j - 1] : 0));
        xk[j] = v > 0 ? v : 0; /* PositiveRegularization after the prox (:166-168) */
      }
      theta_old = theta;
      theta = (1 + sqrt(1 + 4 * theta_old * theta_old)) / 2;
    }
    const double e = rel_err_f(xh, xk, N);
    REQUIRE(e < 1e-5, "FISTA + TV solution: relative error %.3e", e);
    printf("FISTA+TV+Positive 256x128 f32 (prox inside the plan): 15 iterations within 1e-5 (%.2e)\n", e);
    free(fx0d); free(y); free(xk); free(xo); free(res); free(t); free(bd); free(u); free(pq); free(rs); free(po); free(xt);
  }
  CHECK(rls_fista_destroy(fp));

  /* ---- the same CGNR problem row-partitioned over 1, 2 and 4 ranks on device 0 through rls_comm ---- */
  for (int nr = 1; nr <= 4; nr *= 2) {
    rls_comm* comm = NULL;
    int32_t devs[4] = {0, 0, 0, 0};
    CHECK(rls_comm_create(nr, devs, NULL, RLS_COMM_DIRECT, &comm));
    REQUIRE(rls_comm_size(comm) == nr && rls_comm_transport(comm) == RLS_COMM_DIRECT, "communicator shape");
    {
      int32_t peer[16], asked = -1; /* the probe of rls_comm_create: ranks sharing a device can always store into each other */
      CHECK(rls_comm_peer_access(comm, peer, &asked));
      REQUIRE(asked == RLS_COMM_DIRECT, "requested transport %d", asked);
      for (int q = 0; q < nr * nr; ++q) REQUIRE(peer[q] == 1, "peer matrix entry %d = %d", q, peer[q]);
    }
    rls_operator* ops[4];
    rls_cgnr* plans[4];
    void *xs_[4], *bparts[4], *As[4], *vecs[4][3];
    const int64_t rows = M / nr;
    for (int g = 0; g < nr; ++g) {
      rls_ctx* cg = NULL;
      CHECK(rls_comm_ctx(comm, g, &cg));
      float* Ash = (float*)malloc(4 * rows * N); /* repack rows [g rows, (g+1) rows) contiguous, lda = rows */
      for (int64_t j = 0; j < N; ++j) memcpy(Ash + j * rows, A + g * rows + j * M, 4 * rows);
      As[g] = dev_upload(cg, Ash, 4 * rows * N);
      free(Ash);
      bparts[g] = dev_upload(cg, b + g * rows, 4 * rows);
      xs_[g] = dev_upload(cg, NULL, 4 * N);
      for (int k = 0; k < 3; ++k) vecs[g][k] = dev_upload(cg, NULL, 4 * N);
      CHECK(rls_operator_create(cg, RLS_F32, rows, N, As[g], rows, &ops[g]));
      CHECK(rls_cgnr_create(ops[g], xs_[g], vecs[g][0], vecs[g][1], vecs[g][2], &plans[g]));
    }
    CHECK(rls_cgnr_init_rowsharded(comm, plans, (const void* const*)bparts, 1e-2f, 0.f, 10));
    CHECK(rls_cgnr_step_rowsharded(comm, plans, 10));
    CHECK(rls_comm_sync(comm));
    float* x0h = (float*)malloc(4 * N);
    for (int g = 0; g < nr; ++g) {
      rls_ctx* cg = NULL;
      CHECK(rls_comm_ctx(comm, g, &cg));
      CHECK(rls_memcpy_d2h(cg, xh, xs_[g], 4 * N));
      if (g == 0) memcpy(x0h, xh, 4 * N);
      REQUIRE(memcmp(x0h, xh, 4 * N) == 0, "rank %d of %d: replicated x differs from rank 0", g, nr);
      CHECK(rls_cgnr_get_status(plans[g], &st));
      REQUIRE(st.iteration == 10 && st.done, "rank %d iteration %d", g, st.iteration);
    }
    const double e = rel_err_f(x0h, c.x, N);
    REQUIRE(e < 1e-5, "row-sharded CGNR over %d ranks: relative error %.3e", nr, e);
    printf("row-sharded CGNR, %d rank(s) on device 0, direct transport: within 1e-5 (%.2e), replicas bit-identical\n", nr, e);
    free(x0h);
    for (int g = 0; g < nr; ++g) {
      rls_ctx* cg = NULL;
      CHECK(rls_comm_ctx(comm, g, &cg));
      CHECK(rls_cgnr_destroy(plans[g]));
      CHECK(rls_operator_destroy(ops[g]));
      rls_free(cg, As[g]); rls_free(cg, bparts[g]); rls_free(cg, xs_[g]);
      for (int k = 0; k < 3; ++k) rls_free(cg, vecs[g][k]);
    }
    CHECK(rls_comm_destroy(comm));
  }
  { /* RCCL transport, one rank (the box has one GPU): communicator creation + the world-size-1 collective */
    rls_comm* comm = NULL;
    int32_t dev0 = 0;
    rls_ctx* one = g_ctx;
    const int32_t rc = rls_comm_create(1, &dev0, &one, RLS_COMM_RCCL, &comm);
    if (rc == 0) {
      void* bufs[1] = {x};
      CHECK(rls_allreduce_sum(comm, bufs, N, RLS_F32));
      CHECK(rls_comm_sync(comm));
      CHECK(rls_comm_destroy(comm));
      printf("RCCL transport: communicator over 1 device created, all-reduce enqueued\n");
    } else {
      printf("RCCL transport not available here (%d: %s)\n", (int)rc, rls_last_error_string(g_ctx));
    }
  }
  CHECK(rls_operator_destroy(op));
  free(c.x); free(c.r); free(c.p); free(c.v); free(c.t); free(xh); free(A); free(b); free(xt);
}

static void test_headline_shape(void) {
  const int64_t M = 4096, N = 2048;
  float* A = (float*)malloc(8 * M * N); /* interleaved (re, im), column-major */
  for (int64_t i = 0; i < 2 * M * N; ++i) A[i] = gauss() * 0.70710678f;
  float* xt = (float*)malloc(8 * N);
  for (int64_t j = 0; j < 2 * N; ++j) xt[j] = gauss();
  void *Ad = dev_upload(g_ctx, A, 8 * M * N), *xtd = dev_upload(g_ctx, xt, 8 * N), *bd = dev_upload(g_ctx, NULL, 8 * M);
  rls_operator* op = NULL;
  CHECK(rls_operator_create(g_ctx, RLS_C32, M, N, Ad, M, &op));
  CHECK(rls_operator_mul(op, xtd, bd)); /* b = A x_true on the device */
  void *x = dev_upload(g_ctx, NULL, 8 * N), *r = dev_upload(g_ctx, NULL, 8 * N), *p = dev_upload(g_ctx, NULL, 8 * N),
       *v = dev_upload(g_ctx, NULL, 8 * N);
  rls_cgnr* plan = NULL;
  CHECK(rls_cgnr_create(op, x, r, p, v, &plan));
  CHECK(rls_cgnr_init(plan, bd, 0.f, 0.f, 32));
  int32_t path = -1;
  CHECK(rls_cgnr_path(plan, &path));
  rls_cgnr_status st;
  CHECK(rls_cgnr_get_status(plan, &st));
  const float z0 = st.z0;
  CHECK(rls_cgnr_step(plan, 32));
  CHECK(rls_cgnr_get_status(plan, &st));
  REQUIRE(st.iteration == 32 && st.done, "headline: iteration %d", st.iteration);
  float* xh = (float*)malloc(8 * N);
  CHECK(rls_memcpy_d2h(g_ctx, xh, x, 8 * N));
  double d = 0, s = 0;
  for (int64_t j = 0; j < 2 * N; ++j) {
    d += ((double)xh[j] - xt[j]) * ((double)xh[j] - xt[j]);
    s += (double)xt[j] * xt[j];
  }
  REQUIRE(sqrt(d / s) < 1e-4, "headline: x vs planted solution %.3e", sqrt(d / s));
  REQUIRE(st.residual < 1e-4f * z0, "headline: residual %.3e of z0 %.3e", st.residual, z0);
  printf("CGNR 4096x2048 c64, 32 iterations, kernel path %d: planted solution recovered to %.2e, ||r||/z0 = %.2e\n", (int)path,
         sqrt(d / s), st.residual / z0);
  CHECK(rls_cgnr_destroy(plan));
  CHECK(rls_operator_destroy(op));
  free(A); free(xt); free(xh);
}

int main(void) {
  int32_t ndev = 0;
  if (rls_device_count(&ndev) != 0 || ndev < 1) {
    fprintf(stderr, "no device\n");
    return 2;
  }
  REQUIRE(rls_abi_version() == RLS_ABI_VERSION, "ABI version");
  CHECK(rls_ctx_create(0, &g_ctx));
  test_cgnr_and_fista_small();
  test_headline_shape();
  CHECK(rls_ctx_sync(g_ctx));
  CHECK(rls_ctx_destroy(g_ctx));
  printf("abi_smoke OK\n");
  return 0;
}
