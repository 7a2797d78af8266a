"""BASELINE configs[0] (the reference's CPU-runnable case): CGNR 256 x 128 Float32, lambda = 1e-2, 10 iterations --
latency-bound on a GPU; us per iteration through the pipeline (hipGraph off: fewer than one chunk) and per solve"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
A = make_A(256, 128, 1, np.float32); Ad = rls.DeviceMatrix.from_host(A, ctx)
b = rls.DeviceVector.from_host((A @ np.ones(128, np.float32)).astype(np.float32), ctx)
for name, kw in (("matrix-free", {}), ("gram", dict(AHA=Ad.gram()))):
    S = rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(1e-2), iterations=10, relTol=0.0, **kw)
    rls.solve_(S, b)
    def run(n):
        for _ in range(n):
            rls.init_(S, b); ctx.lib.rls_cgnr_step(S.state._plan, 10)
    run(300); ctx.sync(); ctx.timer_start(); run(300); us = ctx.timer_stop_ms() * 1e3 / 300
    t0 = time.perf_counter()
    for _ in range(200): rls.solve_(S, b)
    ctx.sync(); wall = (time.perf_counter() - t0) / 200 * 1e6
    print(f"config 1 {name:12s}: {us/10:6.2f} us per iteration on the device ({us:6.1f} us per 10-iteration solve incl. init), "
          f"{wall:6.1f} us wall clock per solve_() from Python")
# in-kernel cost per iteration of the single-workgroup kernel: slope between step calls of 10 and of 110 iterations
S = rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(1e-2), iterations=128, relTol=0.0)
rls.solve_(S, b)
ts = {}
for n in (10, 110):
    def run(k):
        for _ in range(k):
            rls.init_(S, b); ctx.lib.rls_cgnr_step(S.state._plan, n)
    run(100); ctx.sync(); ctx.timer_start(); run(200); ts[n] = ctx.timer_stop_ms() * 1e3 / 200
print(f"config 1 matrix-free: step(10) {ts[10]:.1f} us, step(110) {ts[110]:.1f} us incl. init -> {(ts[110] - ts[10]) / 100:.2f} us per iteration in the kernel, "
      f"{ts[10] - 10 * (ts[110] - ts[10]) / 100:.1f} us fixed (init! = GEMV + init kernel, launch, load of A, write-back)")
# FISTA + L1 on the same system: the single-workgroup kernel (fista_small_kernel) against the slab pipeline (small = 0)
rho = float(0.9 / np.linalg.norm(A.astype(np.float64), 2) ** 2)
for small in (1, 0):
    ctx.tune(small=small)
    F = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=rho, iterations=50, relTol=0.0)
    rls.solve_(F, b)
    def runf(k):
        for _ in range(k):
            rls.init_(F, b); ctx.lib.rls_fista_step(F.state._plan, 50)
    runf(50); ctx.sync(); ctx.timer_start(); runf(100); us = ctx.timer_stop_ms() * 1e3 / 100
    t0 = time.perf_counter()
    for _ in range(100): rls.solve_(F, b)
    ctx.sync(); wall = (time.perf_counter() - t0) / 100 * 1e6
    print(f"FISTA + L1 256 x 128 Float32, 50 iterations, small = {small}: {us / 50:6.2f} us per iteration on the device incl. init, {wall:6.1f} us wall clock per solve_()")
ctx.tune(small=1)
