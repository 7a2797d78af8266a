"""Replays of the reference's own tests, and fixed-point certificates that do not depend on the restatement, shared by the
oracle tests (tests/test_oracle.py, CPU) and the device tests (tests/test_gpu_reference_suite.py, `-m gpu`).  Module-agnostic:
`mod` is either the oracle (oracle/rls_oracle.py) or the product's host mirror (rls_amd) -- same class names and keywords."""
import numpy as np


def convex_problem(seed, N=256):
    """test/testSolvers.jl:68-82: unitary N-point DFT, three spikes, a random half of the rows (NumPy's generator stands in for
    Julia's global RNG, which cannot be replayed; the suite's assertions are all rtol = 0.1)"""
    rng = np.random.default_rng(seed)
    jk = np.outer(np.arange(N), np.arange(N))
    F = np.exp(-2j * np.pi * jk / N) / np.sqrt(N)
    x = np.zeros(N)
    for _ in range(3):
        x[rng.integers(0, N)] = rng.random()
    b = np.fft.fft(x) / np.sqrt(N)
    idx = np.unique(rng.integers(0, N, N // 2))
    return F[idx, :], x, b[idx]


def convex_suite(mod, F, b, wrapA, wrapb, unwrap, default_rho, lam_t=np.float32, admm_scale=1e3):
    """every solve of test/testSolvers.jl:84-201 through `mod` (the oracle here; the product in tests/test_gpu_parity.py);
    returns {label: x_approx}.  `default_rho(A)` supplies the constructors' default 0.95 / power_iterations(AHA) where the
    module wants it explicitly (the oracle), or {} where the module has the default itself (the product)."""
    solve = getattr(mod, "solve", None) or mod.solve_
    out = {}
    lam0 = lam_t(1e-3)
    none = mod.NoNormalization() if hasattr(mod, "NoNormalization") else "none"
    meas = mod.MeasurementBasedNormalization() if hasattr(mod, "MeasurementBasedNormalization") else "measurement"
    scale_F = 1e3
    for name in ("POGM", "OptISTA", "FISTA", "ADMM"):
        S = getattr(mod, name)
        kw = {} if name == "ADMM" else default_rho(F)
        out[name] = unwrap(solve(S(wrapA(F), reg=mod.L1Regularization(lam0), iterations=200, normalizeReg=none, **kw), wrapb(b)))
        if name in ("POGM", "FISTA"):  # :101-113
            out[name + "+restart"] = unwrap(solve(S(wrapA(F), reg=mod.L1Regularization(lam0), iterations=200, normalizeReg=none,
                                                    restart="gradient", **kw), wrapb(b)))
        # :115-129 invariance to the maximum eigenvalue
        lam1 = lam_t(lam0 * len(b) / np.sum(np.abs(b)))
        sc = admm_scale if name == "ADMM" else scale_F
        kw = {} if name == "ADMM" else default_rho(F * sc)
        xs = unwrap(solve(S(wrapA(F * sc), reg=mod.L1Regularization(lam1), iterations=200, normalizeReg=meas, **kw), wrapb(b)))
        out[name + "+rescaled"] = xs * sc
    for rho, vary in ((1e6, "balance"), (1e-6, "balance"), (1e-6, "PnP")):  # :132-174
        out[f"ADMM rho={rho:g} {vary}"] = unwrap(solve(mod.ADMM(wrapA(F), reg=mod.L1Regularization(lam0), iterations=200,
                                                                normalizeReg=none, rho=rho, vary_rho=vary), wrapb(b)))
    lam2 = lam_t(2e-3)  # :177-201
    out["SplitBregman"] = unwrap(solve(mod.SplitBregman(wrapA(F), reg=mod.L1Regularization(lam2), iterations=5, iterationsInner=40,
                                                        rho=1.0, normalizeReg=none), wrapb(b)))
    lam3 = lam_t(lam2 * len(b) / np.sum(np.abs(b)))
    out["SplitBregman+measurement"] = unwrap(solve(mod.SplitBregman(wrapA(F), reg=mod.L1Regularization(lam3), iterations=5,
                                                                    iterationsInner=40, rho=1.0, normalizeReg=meas), wrapb(b)))
    return out




# ---- fixed-point certificates (independent of any restatement of the solvers) ---------------------------------------
def lasso_problem(seed, M=96, N=160, dt=np.complex128, density=8):
    """an underdetermined system with a sparse planted solution and sigma_max(A) far from 1 (so that a wrong power of rho in a
    prox threshold shows)"""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((M, N))
    x = np.zeros(N)
    if np.dtype(dt).kind == "c":
        A = A + 1j * rng.standard_normal((M, N))
        x = x.astype(np.complex128)
    A = A * 3.0 / np.sqrt(M)
    idx = rng.choice(N, density, replace=False)
    x[idx] = rng.standard_normal(density) + (1j * rng.standard_normal(density) if np.dtype(dt).kind == "c" else 0)
    b = A @ x + 0.01 * rng.standard_normal(M)
    return A.astype(dt), x.astype(dt), b.astype(dt)


def lasso_kkt_violation(A, b, x, lam, support_tol=1e-7):
    """optimality of x for  1/2 ||A x - b||^2 + lam ||x||_1  (complex modulus): with g = A^H (A x - b),
    g_i = -lam x_i / |x_i| on the support and |g_i| <= lam off it.  Returns the largest violation divided by lam (0 = optimal)
    and the support size.  Evaluated in float64 whatever produced x."""
    A = np.asarray(A).astype(np.complex128)
    x = np.asarray(x).astype(np.complex128)
    g = A.conj().T @ (A @ x - np.asarray(b).astype(np.complex128))
    on = np.abs(x) > support_tol * max(np.max(np.abs(x)), 1e-300)
    v_on = np.abs(g[on] + lam * x[on] / np.abs(x[on])) if on.any() else np.zeros(1)
    v_off = np.maximum(np.abs(g[~on]) - lam, 0) if (~on).any() else np.zeros(1)
    return float(max(v_on.max(), v_off.max()) / lam), int(on.sum())


def tv_duality_gap(x, u, lam, D):
    """certificate for u ~ argmin 1/2 ||u - x||^2 + lam ||D u||_1 (anisotropic TV, D = the stacked forward differences as a
    dense matrix): the dual  max_{|p|_inf <= 1}  1/2 ||x||^2 - 1/2 ||x - lam D' p||^2  bounds the primal optimum from below for
    EVERY feasible p, so  P(u) - D(p) >= P(u) - P* >= 0.  p is the feasible point whose primal image is closest to u (a
    bounded least-squares problem, SciPy).  Returns (gap / P(u), P(u))."""
    from scipy.optimize import lsq_linear

    x = np.asarray(x, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    res = lsq_linear(lam * D.T, x - u, bounds=(-1.0, 1.0), tol=1e-14, max_iter=2000)
    p = np.clip(res.x, -1.0, 1.0)
    primal = 0.5 * np.sum((u - x) ** 2) + lam * np.sum(np.abs(D @ u))
    dual = 0.5 * np.sum(x ** 2) - 0.5 * np.sum((x - lam * D.T @ p) ** 2)
    return float((primal - dual) / primal), float(primal)
