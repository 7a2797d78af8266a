"""A/B the CGNR pipeline slab configurations in one process (us per iteration from hipEvents)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
M, N = 4096, 2048
A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx)
b = rls.DeviceVector.from_host((A @ np.ones(N, np.complex64)).astype(np.complex64), ctx)
lib = ctx.lib
res = {}
for rnd in range(3):
    for wv in (8, 16):
        ctx.tune(slab_wv=wv)
        solver = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
        def run(n):
            for _ in range(n):
                rls.init_(solver, b); lib.rls_cgnr_step(solver.state._plan, 32)
        run(3); ctx.sync(); ctx.timer_start(); run(20); us = ctx.timer_stop_ms() * 1e3 / 640
        res.setdefault(wv, []).append(us)
        x = rls.solversolution(solver).to_host()
        del solver
for wv, v in res.items():
    print(f"slab_wv={wv}: {np.median(v):.2f} us/iter (min {min(v):.2f})  err vs ones {np.linalg.norm(x-1)/np.sqrt(N):.2e}")
