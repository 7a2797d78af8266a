"""solve!(solver::FISTA, B) with K columns sharing A on the matrix cores (rls_fista_*_batched), 4096 x 2048 CF32 + L1:
us per batched iteration and solve-iterations per second, against the single-column pipeline"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
M, N = 4096, 2048
A = make_A(M, N, 4); Ad = rls.DeviceMatrix.from_host(A, ctx)
rng = np.random.default_rng(5)
rho = 0.95 / (np.sqrt(M) + np.sqrt(N)) ** 2
lib, h = ctx.lib, ctx.handle
for K in tuple(int(k) for k in sys.argv[1].split(',')) if len(sys.argv) > 1 else (16, 32, 64):
    X = (rng.standard_normal((N, K)) + 1j * rng.standard_normal((N, K))).astype(np.complex64)
    Bd = rls.DeviceMatrix.from_host(np.asfortranarray((A @ X).astype(np.complex64)), ctx)
    S = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=rho, iterations=48, relTol=0.0)
    rls.solve_(S, Bd, scheduler=rls.BatchedState)
    st = S.state
    def run(n):
        for _ in range(n):
            rls._lib.check(h, lib.rls_fista_init_batched(st._plan, Bd.ptr, Bd.lda, rho, 1.0, 0.0, 48, 0), "init")
            rls._lib.check(h, lib.rls_fista_step(st._plan, 48), "step")
    run(3); ctx.sync(); ctx.timer_start(); run(10)
    us = ctx.timer_stop_ms() * 1e3 / (10 * 48)
    print(f"K={K:2d}: {us:7.2f} us per batched FISTA iteration = {us/K:5.2f} us per solve-iteration ({K*1e6/us:8.0f} solve-it/s)", flush=True)
