"""A/B in one process, interleaved rounds: the resident kernels' arrival counters zeroed by the init kernel (resident_preclear = 1)
against a memset launch ahead of the first step call of every solve (0).  us per iteration, headline shape, solves of 32 iterations."""
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
F = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=0.95 / (np.sqrt(M) + np.sqrt(N)) ** 2, iterations=32, relTol=0.0)
rls.solve_(S, bd); rls.solve_(F, bd)
def run_c(n):
    for _ in range(n):
        rls.init_(S, bd); lib.rls_cgnr_step(S.state._plan, 32)
def run_f(n):
    for _ in range(n):
        rls.init_(F, bd); lib.rls_fista_step(F.state._plan, 32)
res = {(k, v): [] for k in "cf" for v in (0, 1)}
for rnd in range(6):
    for v in (1, 0):
        ctx.tune(resident_preclear=v)
        for k, run in (("c", run_c), ("f", run_f)):
            run(10); ctx.sync(); ctx.timer_start(); run(50); res[(k, v)].append(ctx.timer_stop_ms() * 1e3 / (50 * 32))
for k, name in (("c", "CGNR"), ("f", "FISTA + L1")):
    for v in (1, 0):
        r = sorted(res[(k, v)])
        print(f"{name:11s} preclear={v}: min {r[0]:6.2f}  median {r[len(r)//2]:6.2f} us/iteration")
