"""Stress run of the resident kernels (cgnr_resident_kernel / fista_resident_kernel, csrc/normal.hip): many thousand
launches of the headline shape, every solve checked bit for bit against the first one and every launch's arrival /
timeout flags read back through the status call (a spin-limit timeout raises).  Usage: stress_resident.py [seconds]"""
import sys, os, math, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rls_amd as rls
from bench import make_A

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
ctx = rls.default_context(0)
lib = ctx.lib
M, N = 4096, 2048
A = make_A(M, N, 2)
rng = np.random.default_rng(1000)
xt = ((rng.standard_normal(N) + 1j * rng.standard_normal(N)) / math.sqrt(2)).astype(np.complex64)
b = (A @ xt).astype(np.complex64)
Ad, bd = rls.DeviceMatrix.from_host(A, ctx), rls.DeviceVector.from_host(b, ctx)
G = Ad.gram()
rho = 0.95 / (math.sqrt(M) + math.sqrt(N)) ** 2
cases = (("CGNR", lambda: rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)),
         ("FISTA+L1", lambda: rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), iterations=32, relTol=0.0, rho=rho)),
         ("CGNR, Gram mode", lambda: rls.createLinearSolver(rls.CGNR, Ad, AHA=G, iterations=32, relTol=0.0)),
         ("FISTA+L1, Gram mode", lambda: rls.createLinearSolver(rls.FISTA, Ad, AHA=G, reg=rls.L1Regularization(1e-2), iterations=32, relTol=0.0, rho=rho)))
for name, make in cases:
    S = make()
    ref = rls.solve_(S, bd).to_host().copy()
    import ctypes
    pth = ctypes.c_int32(-1)
    (lib.rls_cgnr_path if name.startswith("CGNR") else lib.rls_fista_path)(S.state._plan, ctypes.byref(pth))
    path = pth.value
    t0 = time.perf_counter()
    solves = bad = 0
    while time.perf_counter() - t0 < budget / len(cases):
        for _ in range(50):
            x = rls.solve_(S, bd)  # init! + 32 iterations in one resident launch + status read-back (raises on a timeout)
            solves += 1
        if not np.array_equal(x.to_host(), ref):
            bad += 1
    dt = time.perf_counter() - t0
    print(f"{name}: path {path}, {solves} solves ({solves * 32} iterations) in {dt:.1f} s, {bad} of {solves // 50} sampled results differ "
          f"from the first solve, no launch timed out", flush=True)
    assert bad == 0
