"""diagnostic: phase timeline of one iteration of cgnr_resident_kernel (needs the -DRLS_STAMPS build,
tools/build_stamps.sh -> tools/ubench/librls_stamps.so).  Stamps are those of the LAST iteration of the launch."""
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
b = rls.DeviceVector.from_host((A @ np.ones(N, np.complex64)).astype(np.complex64), ctx)
FISTA = len(sys.argv) > 1 and sys.argv[1] == "fista"   # the same slots in fista_resident_kernel (FISTA + L1)
if FISTA:
    solver = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=0.95 / (np.sqrt(M) + np.sqrt(N)) ** 2, iterations=32, relTol=0.0)
    for _ in range(5):
        rls.init_(solver, b); lib.rls_fista_step(solver.state._plan, 20); ctx.sync()
else:
    solver = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
    for _ in range(5):
        rls.init_(solver, b); lib.rls_cgnr_step(solver.state._plan, 20); ctx.sync()
buf = (C.c_ulonglong * 128)()
lib.rls_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
print("status", lib.rls_debug_stamps(buf))
names = ["iteration start", "products done (partial row stored)", "stores drained + wg barrier", "grid barrier 1 passed",
         "chunk reduced, v + dots stored, drained", "grid barrier 2 passed", "v + dots loaded", "CG update done"]
t00 = min(buf[wg * 16 + 8] for wg in range(7))
for wg in range(7):
    t = [buf[wg * 16 + 8 + i] for i in range(8)]
    print(f"wg {wg*37+5}: start @{(t[0]-t00)*10:+5d} ns  " + "  ".join(f"[{i}] +{(t[i]-t[0])*10}" for i in range(1, 8)))
print("phases: " + "; ".join(f"[{i}] {n}" for i, n in enumerate(names)))
if FISTA:
    sys.exit(0)
# launch level (one 20-iteration launch): entry -> slab loaded and re-arranged into the owner layout -> loop entered -> loop left
e00 = min(buf[wg * 16 + 0] for wg in range(7))
for wg in range(7):
    t = [buf[wg * 16 + i] for i in range(8)]
    print(f"wg {wg*37+5}: iteration 0 took {(t[5]-t[2])/100:.2f} us, iteration 1 {(t[6]-t[5])/100:.2f}, iterations 2..9 {(t[7]-t[6])/800:.2f} each, 10..19 {(t[3]-t[7])/1000:.2f} each")
    print(f"wg {wg*37+5}: entry @{(t[0]-e00)*10:+5d} ns   slab in owner layout +{(t[1]-t[0])/100:.2f} us   loop entered +{(t[2]-t[0])/100:.2f}   loop left +{(t[3]-t[0])/100:.2f}"
          f"   (20 iterations: {(t[3]-t[2])/100:.2f} us = {(t[3]-t[2])/2000:.3f} us each)")
ctx.timer_start(); lib.rls_cgnr_step(solver.state._plan, 0); ctx.timer_stop_ms()
best = 1e9
for _ in range(20):
    rls.init_(solver, b); ctx.sync()
    ctx.timer_start(); lib.rls_cgnr_step(solver.state._plan, 20); best = min(best, ctx.timer_stop_ms())
print(f"hipEvents around rls_cgnr_step(plan, 20): {best*1e3:.1f} us (stamped build)")
