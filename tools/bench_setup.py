"""Setup-path timings: Gram GEMM A^H A on the matrix cores vs the plain tiled kernel, row norms,
row-weighted copy, power iterations -- 4096 x 2048 ComplexF32."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
M, N = 4096, 2048
A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx)
def timeit(f, reps=5):
    f(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): f()
    ctx.sync(); return (time.perf_counter() - t0) / reps * 1e3
flops = 8.0 * N * N * M
for mf in (1, 0):
    ctx.tune(batched_mfma=mf)
    ms = timeit(lambda: Ad.gram())
    print(f"gram (batched_mfma={mf}): {ms:8.3f} ms  = {flops / ms / 1e9:7.1f} TFLOP/s (incl. scratch malloc/free + sync)")
ctx.tune(batched_mfma=1)
G = Ad.gram().to_host(); ref = A.astype(np.complex128); ref = ref.conj().T @ ref
print("gram rel err", np.linalg.norm(G - ref) / np.linalg.norm(ref))
print(f"rownorm2: {timeit(lambda: Ad.rownorm2()):.3f} ms")
w = rls.DeviceVector.from_host(np.ones(M, np.complex64), ctx)
print(f"scale_rows: {timeit(lambda: (Ad.scale_rows(w), ctx.sync())):.3f} ms")
S = None
print(f"FISTA ctor incl. power_iterations: {timeit(lambda: rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-3)), reps=3):.3f} ms")
