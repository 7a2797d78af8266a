"""FISTA + L1 at 4096 x 2048 CF32 (config 2): matrix-free pipeline vs Gram mode (AHA = A'*A explicit)."""
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
G = Ad.gram()
rho = 0.95 / (np.sqrt(M) + np.sqrt(N)) ** 2
for name, kw in (("matrix-free", {}), ("gram", dict(AHA=G))):
    S = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=rho, iterations=50, **kw)
    for _ in range(20): rls.solve_(S, b)
    ctx.sync(); ctx.timer_start()
    for _ in range(20): rls.init_(S, b); ctx.lib.rls_fista_step(S.state._plan, 50)
    us = ctx.timer_stop_ms() * 1e3 / 1000
    print(f"FISTA+L1 {name:12s}: {us:6.2f} us/iteration ({1e6/us:7.0f} it/s)")
