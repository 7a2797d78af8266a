"""headline CGNR iteration time for the two slab_order modes (small loads waited for before the slab / barrier only)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
M, N = 4096, 2048
A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx)
b = rls.DeviceVector.from_host((A @ np.ones(N, np.complex64)).astype(np.complex64), ctx)
for mode in (1, 0, 1, 0):
    ctx.tune(slab_order=mode)
    S = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
    def run(n):
        for _ in range(n):
            rls.init_(S, b); ctx.lib.rls_cgnr_step(S.state._plan, 32)
    run(200); ctx.sync(); ctx.timer_start(); run(100); us = ctx.timer_stop_ms() * 1e3 / 3200
    print(f"slab_order={mode}: {us:.2f} us/iteration ({1e6/us:.0f} it/s)")
