"""diagnostic: phase timeline of cgnr_pipe_a_kernel (needs the -DRLS_STAMPS build in tools/ubench/librls_stamps.so)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import rls_amd as rls
import rls_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "tools", "ubench", "librls_stamps.so")
L._lib = None
from bench import make_A
ctx = rls.Context(0)
lib = ctx.lib
M, N = 4096, 2048
A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx)
b = rls.DeviceVector.from_host((A @ np.ones(N, np.complex64)).astype(np.complex64), ctx)
solver = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
for _ in range(5):
    rls.init_(solver, b); lib.rls_cgnr_step(solver.state._plan, 20); ctx.sync()
buf = (C.c_ulonglong * 128)()
lib.rls_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
print("status", lib.rls_debug_stamps(buf))
names = ["start", "loads issued", "scalars", "prologue start", "xs ready+barrier", "t_w done", "v partial in LDS", "end"]
t00 = buf[0]
for wg in range(7):
    t = [buf[wg * 16 + i] for i in range(8)]
    print(f"wg {wg*37+5}: start @{(t[0]-t00)*10:+d} ns  " + "  ".join(f"{names[i]} +{(t[i]-t[0])*10} ns" for i in range(1, 8)))
r = [buf[7 * 16 + i] for i in range(8)]
rn = ["start", "partials loaded", "LDS combine", "end"]
print("K_R before this K_A (wg 0): " + "  ".join(f"{rn[i]} @{(r[4+i]-t00)*10:+d} ns" for i in range(4)))
print("K_R after  this K_A (wg 0): " + "  ".join(f"{rn[i]} @{(r[i]-t00)*10:+d} ns" for i in range(4)))
