"""diagnostic: phase timeline of cgnr_pipe_a_kernel where it walks two row blocks per workgroup (8192 x 4096 Float32, BASELINE configs[2];
needs the -DRLS_STAMPS build in tools/ubench/librls_stamps.so, tools/build_stamps.sh): us after the workgroup's start."""
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
M, N, dt = (int(sys.argv[1]), int(sys.argv[2]), np.complex64 if sys.argv[3] == "c" else np.float32) if len(sys.argv) > 3 else (8192, 4096, np.float32)
A = make_A(M, N, 2, dt); Ad = rls.DeviceMatrix.from_host(A, ctx)
b = rls.DeviceVector.from_host((A @ np.ones(N, dt)).astype(dt), ctx)
solver = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
for _ in range(5):
    rls.init_(solver, b); lib.rls_cgnr_step(solver.state._plan, 20); ctx.sync()
buf = (C.c_ulonglong * 128)()
lib.rls_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
print("status", lib.rls_debug_stamps(buf))
names = {0: "start", 2: "scalars", 3: "update done, xs in LDS", 8: "block 1: first product done", 9: "t_w known", 10: "second product done (block 2 requested)",
         11: "column sums", 12: "block 2: first product done", 13: "t_w known", 14: "second product done", 15: "column sums", 7: "end"}
t00 = buf[0]
for wg in range(7):
    t = [buf[wg * 16 + i] for i in range(16)]
    print(f"wg {wg*37+5}: start @{(t[0]-t00)*10:+d} ns")
    for i in (2, 3, 8, 9, 10, 11, 12, 13, 14, 15, 7):
        print(f"    {names[i]:45s} +{(t[i]-t[0])/100:.2f} us")
r = [buf[7 * 16 + i] for i in range(8)]
print("K_R after this K_A (wg 0): start @%+.2f us, end @%+.2f us" % ((r[0]-t00)/100, (r[3]-t00)/100))
