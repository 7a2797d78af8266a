#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the float64 oracle (oracle/rls_oracle.py).

The reference (Julia) cannot be executed in the build container and holds no golden vectors for
this path, so these fixtures pin the ORACLE against itself across refactors and give
the GPU tests fixed, seed-independent targets; the oracle in turn is pinned by the reference's exact
known-answer tests replayed in tests/test_oracle.py.  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import rls_oracle as O  # noqa: E402


def cgnr_case(M, N, dt, seed, lam, iters, name):
    A, xt, b = O.make_problem(M, N, dt, seed)
    dt64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    s = O.CGNR(A.astype(dt64), reg=O.L2Regularization(lam), iterations=iters, relTol=0.0)
    s.init(b.astype(dt64))
    xs, rs, ps, al, be = [], [], [], [], []
    for _ in range(iters):
        s.iterate()
        xs.append(s.x.copy()); rs.append(s.r.copy()); ps.append(s.p.copy()); al.append(s.alpha); be.append(s.beta)
    np.savez_compressed(os.path.join(HERE, name), A=A, b=b, lam=lam, x=np.array(xs), r=np.array(rs), p=np.array(ps),
                        alpha=np.array(al), beta=np.array(be))


def fista_case():
    A, xt, b = O.make_problem(64, 32, np.complex64, 2)
    A64, b64 = A.astype(np.complex128), b.astype(np.complex128)
    rho = 0.95 / np.linalg.norm(A64, 2) ** 2
    lam = 1e-2 * np.max(np.abs(A64.conj().T @ b64))
    out = {}
    for restart in ("none", "gradient"):
        s = O.FISTA(A64, reg=O.L1Regularization(lam), rho=rho, iterations=50, restart=restart)
        O.solve(s, b64)
        out["x_" + restart] = s.x.copy()
        out["rel_" + restart] = s.rel_res_norm
    np.savez_compressed(os.path.join(HERE, "fista_l1_64x32_c64.npz"), A=A, b=b, rho=rho, lam=lam, **out)


def admm_case():
    A, xt, b = O.make_problem(128, 64, np.float32, 3)
    s = O.ADMM(A.astype(np.float64), reg=O.TVRegularization(1e-2, shape=(8, 8)), rho=0.1, iterations=10,
               iterationsCG=10, tolInner=1e-5)
    O.solve(s, b.astype(np.float64))
    np.savez_compressed(os.path.join(HERE, "admm_tv_128x64_f32.npz"), A=A, b=b, x=s.x, rk=s.rk, sk=s.sk,
                        cg_iters=np.array(s.cg_iters))


def prox_cases():
    rng = np.random.default_rng(42)
    out = {}
    for tag, dt in (("f32", np.float32), ("c64", np.complex64)):
        n = 96
        x = rng.standard_normal(n) + (1j * rng.standard_normal(n) if np.dtype(dt).kind == "c" else 0)
        x = x.astype(dt)
        x[:4] = 0
        x[4:8] *= 1e-3
        out[f"x_{tag}"] = x
        out[f"l1_{tag}"] = O.prox_l1(x.copy(), 0.35)
        out[f"l2_{tag}"] = O.prox_l2(x.copy(), 0.35)
        out[f"l21_{tag}"] = O.prox_l21(x.copy(), 0.8, 8)
        out[f"pos_{tag}"] = O.prox_positive(x.copy())
        out[f"real_{tag}"] = O.prox_real(x.copy())
        out[f"tv_{tag}"] = O.prox_tv_fgp(x.astype(np.complex128 if np.dtype(dt).kind == "c" else np.float64).copy(),
                                         0.3, (12, 8), None, 10).astype(dt)
        out[f"tv1_{tag}"] = O.prox_tv_fgp(x.astype(np.complex128 if np.dtype(dt).kind == "c" else np.float64).copy(),
                                          0.3, (12, 8), (1,), 10).astype(dt)
    np.savez_compressed(os.path.join(HERE, "prox_cases.npz"), **out)


def next_tier_cases():
    """Kaczmarz, OptISTA, POGM (both restart modes), SplitBregman on one 96 x 40 ComplexF32 problem"""
    A, xt, b = O.make_problem(96, 40, np.complex64, 7)
    A64, b64 = A.astype(np.complex128), b.astype(np.complex128)
    rho = 0.95 / np.linalg.norm(A64, 2) ** 2
    lam = 1e-2 * np.max(np.abs(A64.conj().T @ b64))
    out = dict(A=A, b=b, rho=rho, lam=lam)
    k = O.Kaczmarz(A64, reg=O.L2Regularization(0.05), iterations=6)
    out["kaczmarz_x"] = O.solve(k, b64).copy()
    out["kaczmarz_vl"] = k.vl.copy()
    out["optista_x"] = O.solve(O.OptISTA(A64, reg=O.L1Regularization(lam), rho=rho, iterations=30), b64).copy()
    out["pogm_x"] = O.solve(O.POGM(A64, reg=O.L1Regularization(lam), rho=rho, iterations=30), b64).copy()
    out["pogm_restart_x"] = O.solve(O.POGM(A64, reg=O.L1Regularization(lam), rho=rho, iterations=30, restart="gradient"), b64).copy()
    sb = O.SplitBregman(A64, reg=O.L1Regularization(0.05), rho=0.5, iterations=3, iterationsInner=4, iterationsCG=10)
    out["splitbregman_x"] = O.solve(sb, b64).copy()
    out["splitbregman_cg_iters"] = np.array(sb.cg_iters)
    np.savez_compressed(os.path.join(HERE, "next_tier_96x40_c64.npz"), **out)


if __name__ == "__main__":
    next_tier_cases()
    cgnr_case(256, 128, np.float32, 1, 1e-2, 10, "cgnr_256x128_f32.npz")      # BASELINE config 1
    cgnr_case(64, 32, np.complex64, 1, 0.0, 10, "cgnr_64x32_c64.npz")
    fista_case()
    admm_case()
    prox_cases()
    print("golden fixtures written to", HERE)
