// CPU baseline (TEST / MEASUREMENT INFRASTRUCTURE ONLY -- nothing under regularizedleastsquares.jl_amd/ links or loads
// this): the CGNR iteration of RegularizedLeastSquares.jl restated in C++ with OpenMP, for the `cpu_baseline` leg of
// bench.py ("NumPy/OpenBLAS cgemv (and a C++/OpenMP variant)").  Follows /root/reference
// src/CGNR.jl:107-130 (init!: x0 = A' b, z0 = norm(x0), pl = x0) and :143-178 (iterate: vl = A'(A pl), zeta, alpha,
// x += alpha pl, x0 -= alpha vl [- lambda alpha pl], beta, pl = beta pl + x0) for a dense column-major ComplexF32 A.
// Both products stream A once each with every core (column panels per thread, see gemv_n / gemv_c); the upload of A
// into a NUMA-placed private copy is outside the timed region.  Float32 arithmetic, scalar reductions in double.
// Checked against oracle/rls_oracle.py in tests/test_oracle.py.  Build: oracle/Makefile -> oracle/_build/libcgnr_omp.so
#include <omp.h>

#include <chrono>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef std::complex<float> cf;

// Both products give every thread the same contiguous panel of COLUMNS (so a thread streams the same, first-touched,
// NUMA-local part of A in both): t = A p as per-thread partial M-vectors summed afterwards (rows split over threads),
// v = A^H t as independent dot products.
static void panel_of(int64_t N, int nt, int id, int64_t* lo, int64_t* hi) {
  const int64_t blk = (N + nt - 1) / nt;
  *lo = std::min<int64_t>(N, (int64_t)id * blk);
  *hi = std::min<int64_t>(N, *lo + blk);
}

static void gemv_n(const cf* A, int64_t M, int64_t N, const cf* p, cf* t, float* part /* [nt][2M] */) {  // t = A p
#pragma omp parallel
  {
    const int nt = omp_get_num_threads(), id = omp_get_thread_num();
    int64_t lo, hi;
    panel_of(N, nt, id, &lo, &hi);
    float* tr = part + (int64_t)id * 2 * M;
    for (int64_t i = 0; i < 2 * M; ++i) tr[i] = 0.f;
    for (int64_t j = lo; j < hi; ++j) {
      const float pr = p[j].real(), pi = p[j].imag();
      const float* a = reinterpret_cast<const float*>(A + j * M);
#pragma omp simd
      for (int64_t i = 0; i < M; ++i) {
        const float ar = a[2 * i], ai = a[2 * i + 1];
        tr[2 * i] += ar * pr - ai * pi;
        tr[2 * i + 1] += ar * pi + ai * pr;
      }
    }
#pragma omp barrier
    float* out = reinterpret_cast<float*>(t);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < 2 * M; ++i) {
      float s = 0.f;
      for (int k = 0; k < nt; ++k) s += part[(int64_t)k * 2 * M + i];
      out[i] = s;
    }
  }
}

static void gemv_c(const cf* A, int64_t M, int64_t N, const cf* t, cf* v) {  // v = A^H t
  const float* tr = reinterpret_cast<const float*>(t);
#pragma omp parallel
  {
    const int nt = omp_get_num_threads(), id = omp_get_thread_num();
    int64_t lo, hi;
    panel_of(N, nt, id, &lo, &hi);
    for (int64_t j = lo; j < hi; ++j) {
      const float* a = reinterpret_cast<const float*>(A + j * M);
      float sr = 0.f, si = 0.f;
#pragma omp simd reduction(+ : sr, si)
      for (int64_t i = 0; i < M; ++i) {
        const float ar = a[2 * i], ai = a[2 * i + 1], xr = tr[2 * i], xi = tr[2 * i + 1];
        sr += ar * xr + ai * xi;  // conj(a) * t
        si += ar * xi - ai * xr;
      }
      v[j] = cf(sr, si);
    }
  }
}

static double nrm2sq(const cf* x, int64_t n) {
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i) s += (double)x[i].real() * x[i].real() + (double)x[i].imag() * x[i].imag();
  return s;
}

extern "C" {

int cgnr_omp_max_threads(void) { return omp_get_max_threads(); }

// Runs `solves` back-to-back solves of `iterations` iterations each (init! included, as the GPU leg of bench.py does);
// x_out (length N, interleaved) receives the last solution; *seconds the wall time of everything.  Returns the
// iterations executed.
int64_t cgnr_omp_run(const float* A_, int64_t M, int64_t N, const float* b_, int iterations, int solves, float lambda,
                     int threads, float* x_out, double* seconds) {
  if (threads > 0) omp_set_num_threads(threads);
  const cf* b = reinterpret_cast<const cf*>(b_);
  // private copy of A, each column panel first touched by the thread that will stream it (NUMA placement)
  const int nt = omp_get_max_threads();
  cf* A = static_cast<cf*>(std::malloc(sizeof(cf) * (size_t)M * (size_t)N));
  if (!A) return -1;
#pragma omp parallel
  {
    int64_t lo, hi;
    panel_of(N, omp_get_num_threads(), omp_get_thread_num(), &lo, &hi);
    if (hi > lo) std::memcpy(A + lo * M, reinterpret_cast<const cf*>(A_) + lo * M, sizeof(cf) * (size_t)(hi - lo) * (size_t)M);
  }
  std::vector<float> part((size_t)nt * 2 * (size_t)M);
  std::vector<cf> x(N), r(N), p(N), v(N), t(M);
  int64_t done = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int s = 0; s < solves; ++s) {
    gemv_c(A, M, N, b, r.data());  // x0 = A' b            src/CGNR.jl:132
    for (int64_t i = 0; i < N; ++i) {
      x[i] = cf(0.f, 0.f);
      p[i] = r[i];
    }
    for (int it = 0; it < iterations; ++it) {
      gemv_n(A, M, N, p.data(), t.data(), part.data());  // vl = AHA pl   :151 (matrix-free: two products)
      gemv_c(A, M, N, t.data(), v.data());
      const double zeta = nrm2sq(r.data(), N);  // :153
      double dr = 0.0, di = 0.0;                // dot(pl, vl), first argument conjugated   :154
      for (int64_t i = 0; i < N; ++i) {
        dr += (double)p[i].real() * v[i].real() + (double)p[i].imag() * v[i].imag();
        di += (double)p[i].real() * v[i].imag() - (double)p[i].imag() * v[i].real();
      }
      if (lambda > 0.f) dr += (double)lambda * nrm2sq(p.data(), N);  // :158
      const double den = dr * dr + di * di;
      const cf alpha((float)(zeta * dr / den), (float)(-zeta * di / den));  // zeta / normvl
      for (int64_t i = 0; i < N; ++i) {
        x[i] += p[i] * alpha;                                   // :163
        r[i] += v[i] * (-alpha);                                // :164
        if (lambda > 0.f) r[i] += p[i] * (-lambda) * alpha;     // :166
      }
      const float beta = (float)(nrm2sq(r.data(), N) / zeta);   // :170
      for (int64_t i = 0; i < N; ++i) p[i] = p[i] * beta + r[i];  // :173-174
      ++done;
    }
  }
  *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::free(A);
  for (int64_t i = 0; i < N; ++i) {
    x_out[2 * i] = x[i].real();
    x_out[2 * i + 1] = x[i].imag();
  }
  return done;
}
}
