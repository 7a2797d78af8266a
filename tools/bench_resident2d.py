"""the 16 x 16 tile-grid resident CGNR kernel (resident2d.hip) against the row-slab one (normal.hip): same solve, solutions
compared, us per iteration from the slope between a 32- and a 288-iteration step call (hipEvents).
usage: bench_resident2d.py [M N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
M, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 2048)
A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx)
rng = np.random.default_rng(3)
xt = (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)
b = rls.DeviceVector.from_host((A @ xt).astype(np.complex64), ctx)
lib, h = ctx.lib, ctx.handle
out = {}
for two_d in (0, 1, 0, 1):
    ctx.tune(resident_2d=two_d)
    S = rls.createLinearSolver(rls.CGNR, Ad, iterations=320, relTol=0.0)
    x = rls.solve_(S, b).to_host()
    st = S.state
    plan = st._plan
    def run(n):
        rls.init_(S, b)
        ctx.sync(); ctx.timer_start()
        rls._lib.check(h, lib.rls_cgnr_step(plan, n), "step")
        return ctx.timer_stop_ms() * 1e3
    for n in (32, 288):
        run(n)
    t32 = min(run(32) for _ in range(5)); t288 = min(run(288) for _ in range(5))
    S32 = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
    x32 = rls.solve_(S32, b).to_host()
    out[two_d] = x32
    print(f"resident_2d={two_d}: {(t288 - t32) / 256:6.2f} us per iteration in the kernel (32: {t32:7.1f} us, 288: {t288:7.1f} us), "
          f"|x32 - x_true| / |x_true| = {np.linalg.norm(x32 - xt) / np.linalg.norm(xt):.2e}, fallbacks {S32.state.fallbacks if hasattr(S32.state, 'fallbacks') else '?'}", flush=True)
print("2-D vs row-slab after 32 iterations:", np.linalg.norm(out[1] - out[0]) / np.linalg.norm(out[0]))
