"""Gram-mode CGNR (AHA = A'*A explicit, the reference constructor's default for a dense Matrix, src/CGNR.jl:49):
per-iteration traffic N*N*s instead of 2*M*N*s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
M, N = 4096, 2048
A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx)
x = (np.random.default_rng(1).standard_normal(N) + 0j).astype(np.complex64)
b = rls.DeviceVector.from_host((A @ x).astype(np.complex64), ctx)
G = Ad.gram()
for name, kw in (("gram", dict(AHA=G)), ("matrix-free", {})):
    S = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0, **kw)
    xs = rls.solve_(S, b).to_host()
    lib, h = ctx.lib, ctx.handle
    def run(n):
        for _ in range(n):
            rls.init_(S, b); rls._lib.check(h, lib.rls_cgnr_step(S.state._plan, 32), "step")
    run(60); ctx.sync()  # the first long host wait of a process returns ~50 ms late, once (tools/stall_probe2.py): take it here
    run(5); ctx.sync(); ctx.timer_start(); run(40); us = ctx.timer_stop_ms() * 1e3 / (40 * 32)
    print(f"{name:12s}: {us:6.2f} us per iteration ({1e6/us:7.0f} it/s), err {np.linalg.norm(xs-x)/np.linalg.norm(x):.1e}")
