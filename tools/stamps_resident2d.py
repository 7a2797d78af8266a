"""diagnostic: phase timeline of the last iteration of cgnr_resident2d_kernel in workgroup 0 (needs the -DRLS_STAMPS build,
tools/build_stamps.sh -> tools/ubench/librls_stamps.so)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rls_amd as rls
import rls_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "tools", "ubench", "librls_stamps.so")
L._lib = None
from bench import make_A
ctx = rls.Context(0)
lib = ctx.lib
M, N = 4096, 2048
A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx)
rng = np.random.default_rng(3)
b = rls.DeviceVector.from_host((A @ (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)).astype(np.complex64), ctx)
S = rls.createLinearSolver(rls.CGNR, Ad, iterations=64, relTol=0.0)
names = ["iteration start", "product 1 done, rows of t handed to the store unit", "stores drained, workgroup barrier", "row group arrived (hop A)",
         "t_i summed (two halves in LDS, barrier)", "product 2 done (wave partials in LDS, barrier)", "v partial + dots out, drained",
         "grid arrived (hop B)", "v_j and the dots summed (barrier)", "alpha, x, r; ||r_j||^2 out", "row group arrived (hop C)",
         "beta, p, scalars (barrier): iteration end"]
for rep in range(3):
    rls.solve_(S, b); ctx.sync()
    buf = (C.c_ulonglong * 32)()
    lib.rls_debug_r2_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
    assert lib.rls_debug_r2_stamps(buf) == 0
    t = [buf[k] for k in range(12)]
    print(f"--- run {rep}: {(t[11] - t[0]) / 100:.2f} us for the iteration")
    for k in range(1, 12):
        print(f"  {(t[k] - t[k - 1]) / 100:6.2f} us  -> {names[k]}")
