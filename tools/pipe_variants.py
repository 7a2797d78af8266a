"""A/B pipeline knobs in one process, interleaved rounds (us per CGNR iteration from hipEvents)"""
import os, sys, itertools
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
variants = [dict(slab_order=o, red_threads=t) for o in (0, 1) for t in (256, 512, 1024)]
res = {}
def run(n):
    for _ in range(n):
        rls.init_(solver, b); lib.rls_cgnr_step(solver.state._plan, 32)
solver = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
run(150); ctx.sync()
for rnd in range(4):
    for v in variants:
        ctx.tune(**v)
        solver = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)  # fresh plan: fresh hipGraph
        run(3); ctx.sync(); ctx.timer_start(); run(30); us = ctx.timer_stop_ms() * 1e3 / 960
        res.setdefault(tuple(v.items()), []).append(us)
        x = rls.solversolution(solver).to_host()
        assert np.linalg.norm(x - 1) / np.sqrt(N) < 1e-4
for k, v in res.items():
    print(dict(k), f"median {np.median(v):.2f} us/iter  min {min(v):.2f}")
