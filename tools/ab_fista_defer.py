"""A/B of fista_resident_kernel's deferred residual norm (rls_tune_set("fista_defer", 0 / 1)) in ONE process: FISTA + L1 at the
BASELINE configs[1] shape, us per iteration by hipEvents around back-to-back solves of `its` iterations (init! inside) and around
one long step call (in-kernel slope); both settings must give the SAME BITS (the norm only decides the stopping test)."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rls_amd as rls  # noqa: E402
from bench import make_A  # noqa: E402

ctx = rls.default_context(0)
lib = ctx.lib
M, N = 4096, 2048
A = make_A(M, N, 2)
rng = np.random.default_rng(1000)
xt = ((rng.standard_normal(N) + 1j * rng.standard_normal(N)) / math.sqrt(2)).astype(np.complex64)
b = (A @ xt).astype(np.complex64)
Ad, bd = rls.DeviceMatrix.from_host(A, ctx), rls.DeviceVector.from_host(b, ctx)
rho = 0.95 / (math.sqrt(M) + math.sqrt(N)) ** 2
sols = {}
for rep in range(2):
    for defer in (1, 0):
        ctx.tune(fista_defer=defer)
        for its in (32, 256):
            S = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=rho, iterations=its, relTol=0.0)
            x = rls.solve_(S, bd).to_host()
            key = its
            if key in sols:
                assert np.array_equal(sols[key], x), f"defer = {defer} changed the bits of the {its}-iteration solve"
            sols[key] = x

            def run(n):
                for _ in range(n):
                    rls.init_(S, bd)
                    lib.rls_fista_step(S.state._plan, its)
            run(10); ctx.sync()
            best = 1e9
            for _ in range(6):
                ctx.timer_start(); run(20); best = min(best, ctx.timer_stop_ms())
            print(f"fista_defer={defer} solves of {its:3d} iterations: {best * 1e3 / (20 * its):6.2f} us per iteration (incl. init!)", flush=True)
ctx.tune(fista_defer=1)
