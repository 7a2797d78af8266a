"""Resident CGNR (one launch per step call, A in registers) against the two-launch pipeline at the headline shape:
us per iteration by hipEvents around back-to-back solves of 32 iterations (init! inside, as in bench.py)."""
import sys, os, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rls_amd as rls
from bench import make_A

ctx = rls.default_context(0)
lib = ctx.lib
M, N = 4096, 2048
A = make_A(M, N, 2)
rng = np.random.default_rng(1000)
xt = ((rng.standard_normal(N) + 1j * rng.standard_normal(N)) / math.sqrt(2)).astype(np.complex64)
b = (A @ xt).astype(np.complex64)
Ad, bd = rls.DeviceMatrix.from_host(A, ctx), rls.DeviceVector.from_host(b, ctx)
S = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
for mode, bar in ((1, 2), (1, 1), (0, 2)):  # resident two-level exchange, resident flat exchange, two-launch pipeline
    ctx.tune(resident=mode, resident_barrier=bar)
    for seg in (32, 8):
        def run(nsolves):
            for _ in range(nsolves):
                rls.init_(S, bd)
                for _ in range(32 // seg):
                    lib.rls_cgnr_step(S.state._plan, seg)
        run(20); ctx.sync()
        best = 1e9
        for _ in range(5):
            ctx.timer_start(); run(50); best = min(best, ctx.timer_stop_ms())
        S.state._refresh(lib)
        print(f"resident={mode} barrier={bar} step calls of {seg:2d}: {best * 1e3 / (50 * 32):7.2f} us/iteration (incl. init! GEMV per 32)  it={S.state.iteration} res={S.state._residual:.3e}", flush=True)
