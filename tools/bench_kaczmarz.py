"""Kaczmarz row sweeps at 4096 x 2048 ComplexF32: row steps per second for one right-hand side
(latency-bound: one workgroup) and for K independent right-hand sides in one launch (one workgroup each)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from bench import make_A
from oracle import rls_oracle as O
ctx = rls.Context(0)
M, N = 4096, 2048
A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx)
rng = np.random.default_rng(5)
sweeps = 10
if len(sys.argv) > 1:
    ctx.tune(kaczmarz_nt=int(sys.argv[1]))
for K in (1, 256):
    X = (rng.standard_normal((N, K)) + 1j * rng.standard_normal((N, K))).astype(np.complex64)
    B = np.asfortranarray((A @ X).astype(np.complex64))
    S = rls.createLinearSolver(rls.Kaczmarz, Ad, reg=rls.L2Regularization(1e-3), iterations=sweeps)
    b = rls.DeviceMatrix.from_host(B, ctx) if K > 1 else rls.DeviceVector.from_host(B[:, 0], ctx)
    kw = dict(scheduler=rls.BatchedState) if K > 1 else {}
    rls.solve_(S, b, **kw); ctx.sync()
    dts = []
    for _ in range(5):
        t0 = time.perf_counter(); rls.solve_(S, b, **kw); ctx.sync(); dts.append(time.perf_counter() - t0)
    dt = min(dts)
    rows = sweeps * M
    print(f"K={K:4d}: {dt*1e3:8.2f} ms for {sweeps} sweeps = {dt/rows*1e6:6.3f} us per row step, "
          f"{K*rows/dt/1e6:8.2f} M row-updates/s, A stream {K*rows*N*8/dt/1e9:8.1f} GB/s", flush=True)
# CPU: the oracle's NumPy loop on the same matrix (one sweep)
ref = O.Kaczmarz(A, reg=O.L2Regularization(1e-3), iterations=1)
t0 = time.perf_counter(); O.solve(ref, (A @ X[:, 0]).astype(np.complex64)); dt = time.perf_counter() - t0
print(f"CPU (NumPy port, 1 thread): {dt/M*1e6:.2f} us per row step")
