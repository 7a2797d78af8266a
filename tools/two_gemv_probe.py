"""headline shape through the two-GEMV path (what shapes outside the slab limits run): us per CGNR iteration"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
M, N = 4096, 2048
A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx)
b = rls.DeviceVector.from_host((A @ np.ones(N, np.complex64)).astype(np.complex64), ctx)
for fused in (0, 1):
    ctx.tune(fused_normal=fused)
    S = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
    def run(n):
        for _ in range(n):
            rls.init_(S, b); ctx.lib.rls_cgnr_step(S.state._plan, 32)
    run(200); ctx.sync(); ctx.timer_start(); run(100); us = ctx.timer_stop_ms() * 1e3 / 3200
    print(f"fused_normal={fused}: {us:.2f} us/iteration ({1e6/us:.0f} it/s)")
