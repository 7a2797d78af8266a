"""OptISTA / POGM + L1 on resident launches at the BASELINE configs[1] shape, A/B of the deferred residual norm
(rls_tune_set("fista_defer", 0 / 1)) in one process: us per iteration by hipEvents around LONG solves (960 iterations = 20 blocks
of 48 enqueued back to back, one read-back at the end), so that the host's per-solve work does not show; same bits required."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rls_amd as rls  # noqa: E402
from bench import make_A  # noqa: E402

ctx = rls.default_context(0)
M, N = 4096, 2048
A = make_A(M, N, 2)
b = (A @ np.ones(N, np.complex64)).astype(np.complex64)
Ad, bd = rls.DeviceMatrix.from_host(A, ctx), rls.DeviceVector.from_host(b, ctx)
rho = 0.95 / (math.sqrt(M) + math.sqrt(N)) ** 2
its = int(sys.argv[1]) if len(sys.argv) > 1 else 960
for name in ("OptISTA", "POGM"):
    sols = {}
    for rep in range(2):
        for defer in (1, 0):
            ctx.tune(fista_defer=defer)
            S = rls.createLinearSolver(getattr(rls, name), Ad, reg=rls.L1Regularization(1e-2), rho=rho, iterations=its, relTol=0.0)
            x = rls.solve_(S, bd).to_host()
            if name in sols:
                assert np.array_equal(sols[name], x), f"{name}: fista_defer = {defer} changed the bits"
            sols[name] = x

            def run():
                rls.init_(S, bd)
                S._run(S.state)
            run(); ctx.sync()
            best = 1e9
            for _ in range(5):
                ctx.timer_start(); run(); best = min(best, ctx.timer_stop_ms())
            print(f"{name} fista_defer={defer}: {best * 1e3 / its:6.2f} us per iteration ({its}-iteration solves incl. init!), iterations {S.state.iteration}", flush=True)
ctx.tune(fista_defer=1)
