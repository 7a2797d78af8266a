"""ADMM + L1 at 4096 x 2048 CF32 (one column; 10 outer x 10 inner cg! iterations): us per outer iteration with the inner
cg! on the resident kernel (cg! entry folded into the launch) and on the two-launch pipeline."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rls_amd as rls
from bench import make_A
ctx = rls.default_context(0)
A = make_A(4096, 2048, 2); Ad = rls.DeviceMatrix.from_host(A)
b = rls.DeviceVector.from_host((A @ np.ones(2048, np.complex64)).astype(np.complex64))
for r in (0, 1, 0, 1):
    ctx.tune(resident=r)
    S = rls.createLinearSolver(rls.ADMM, Ad, reg=rls.L1Regularization(1e-2), rho=0.1, iterations=10, iterationsCG=10, tolInner=1e-5, absTol=0.0, relTol=0.0)
    rls.solve_(S, b); rls.solve_(S, b); ctx.sync()
    ctx.timer_start()
    for _ in range(5): rls.solve_(S, b)
    ev = ctx.timer_stop_ms()
    print(f"resident {r}: ADMM+L1 4096x2048 CF32 {ev * 1e3 / 50:.1f} us per outer iteration (inner cg! iterations {S.state.cg_iterations})", flush=True)
ctx.tune(resident=1)
