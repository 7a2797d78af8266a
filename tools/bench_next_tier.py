"""SURVEY 8f-1 solvers at the config-2 shape (4096 x 2048 CF32, L1): us per iteration, host-sequenced launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
M, N = 4096, 2048
A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx)
b = rls.DeviceVector.from_host((A @ np.ones(N, np.complex64)).astype(np.complex64), ctx)
rho = 0.95 / (np.sqrt(M) + np.sqrt(N)) ** 2
for name, T, kw, its in (("OptISTA", rls.OptISTA, dict(rho=rho), 50), ("POGM", rls.POGM, dict(rho=rho), 50),
                         ("POGM restart", rls.POGM, dict(rho=rho, restart="gradient"), 50),
                         ("SplitBregman", rls.SplitBregman, dict(rho=0.1, iterations=2, iterationsInner=5, iterationsCG=10), 10)):
    S = rls.createLinearSolver(T, Ad, reg=rls.L1Regularization(1e-2), **({"iterations": its} | kw) if "iterations" not in kw else kw)
    rls.solve_(S, b); ctx.sync()
    dts = []
    for _ in range(3):
        t0 = time.perf_counter(); rls.solve_(S, b); ctx.sync(); dts.append(time.perf_counter() - t0)
    n = its
    print(f"{name:14s}: {min(dts) / n * 1e6:8.1f} us per {'inner iteration (10 CG steps)' if T is rls.SplitBregman else 'iteration'}")
