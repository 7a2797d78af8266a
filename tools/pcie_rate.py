"""PCIe-inclusive rate (never the bench `value`): upload of A (pageable host buffer) + b, then one 32-iteration solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
M, N = 4096, 2048
A = make_A(M, N, 2)
x = (np.random.default_rng(1).standard_normal(N) + 0j).astype(np.complex64)
b = (A @ x).astype(np.complex64)
def once():
    t0 = time.perf_counter()
    Ad = rls.DeviceMatrix.from_host(A, ctx); bd = rls.DeviceVector.from_host(b, ctx)
    ctx.sync(); t1 = time.perf_counter()
    S = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
    rls.solve_(S, bd).to_host(); t2 = time.perf_counter()
    return t1 - t0, t2 - t1
once()
ups, sols = zip(*[once() for _ in range(5)])
up, sol = min(ups), min(sols)
print(f"upload A+b: {up*1e3:.2f} ms ({A.nbytes/up/1e9:.1f} GB/s), 32-iteration solve incl. plan creation + download: {sol*1e3:.2f} ms")
print(f"PCIe-inclusive: {32/(up+sol):.0f} iterations/s for a single 32-iteration solve; resident: {32/sol:.0f}")
