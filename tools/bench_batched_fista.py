"""Batched FISTA + L1 on the explicit Gram matrix (BASELINE configs[3]'s columns as FISTA states sharing solver.AHA,
src/MultiThreading.jl:30-48): microseconds per batched iteration of the resident launch (csrc/gramk.hip,
fista_gramk_resident_kernel, rls_fista_path 7) and of the streaming kernels (one skinny product + one update launch per
iteration, path 3).  usage: python3 tools/bench_batched_fista.py [K=8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch  # noqa
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
M, N = 4096, 2048
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
A = make_A(M, N, 4); Ad = rls.DeviceMatrix.from_host(A, ctx)
Gd = Ad.gram()
rng = np.random.default_rng(5)
X = (rng.standard_normal((N, K)) + 1j * rng.standard_normal((N, K))).astype(np.complex64)
B = np.asfortranarray((A @ X).astype(np.complex64))
Bd = rls.DeviceMatrix.from_host(B, ctx)
rho = float(0.9 / np.linalg.norm(A.astype(np.complex128), 2) ** 2)
lam = 1e-3 * float(np.abs(A.conj().T @ B[:, 0]).max())
lib, h = ctx.lib, ctx.handle
for resident in (1, 0):
    ctx.tune(resident=resident)
    for iters in (32, 64, 256):
        S = rls.createLinearSolver(rls.FISTA, Ad, AHA=Gd, reg=rls.L1Regularization(lam), rho=rho, iterations=iters, relTol=0.0)
        rls.solve_(S, Bd, scheduler=rls.BatchedState)
        st = S.state
        path = C.c_int32(-1)
        lib.rls_fista_path(st._plan, C.byref(path))
        def init():  # (rls.init_ would build a new plan: the same plan is re-initialised here, as a solver loop over frames would)
            rls._lib.check(h, lib.rls_fista_init_batched(st._plan, Bd.ptr, Bd.lda, rho, 1.0, 0.0, iters, 0), "rls_fista_init_batched")
        def run(n):
            for _ in range(n):
                init()
                st._step(iters)
        def run_init(n):
            for _ in range(n):
                init()
        run(3); ctx.sync()
        reps = 20
        ctx.timer_start(); run(reps); t = ctx.timer_stop_ms()
        stat = st.status()
        assert all(s_.fallbacks == 0 and s_.iteration == iters for s_ in stat), [(s_.iteration, s_.fallbacks) for s_ in stat]
        ctx.timer_start(); run_init(reps); t0 = ctx.timer_stop_ms()
        us, us_noinit = t * 1e3 / (reps * iters), (t - t0) * 1e3 / (reps * iters)
        print(f"K={K} resident={resident} path {path.value} {iters:4d}-iteration solves: {us:6.2f} us per batched iteration incl. init! "
              f"({us_noinit:6.2f} without; {K * 1e6 / us:8.0f} solve-it/s)", flush=True)
ctx.tune(resident=1)
