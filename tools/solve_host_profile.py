"""where the wall clock of ONE whole solve!() goes on the host (4096 x 2048 CF32 CGNR, 32 iterations: 0.45 ms of kernels):
createLinearSolver, the first init_ (state vectors + plan creation), upload of b, a later init_ (plan reused), the step call,
the status read-back, the download of x -- each stage synchronised so that its own cost shows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rls_amd as rls
from bench import make_A

ctx = rls.default_context(0)
M, N = 4096, 2048
A = make_A(M, N, 2)
b = (A @ np.ones(N, np.complex64)).astype(np.complex64)
Ad = rls.DeviceMatrix.from_host(A, ctx)
ctx.sync()


def stage(name, fn, acc):
    t0 = time.perf_counter(); r = fn(); ctx.sync(); acc.setdefault(name, []).append(1e6 * (time.perf_counter() - t0)); return r


for rep in range(6):
    acc = {} if rep == 1 else acc if rep else {}
    S = stage("createLinearSolver", lambda: rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0), acc)
    bd = stage("upload b (hipMalloc + h2d)", lambda: rls.DeviceVector.from_host(b, ctx), acc)
    stage("first init_ (state vectors + plan)", lambda: rls.init_(S, bd), acc)
    stage("step(32) + status", lambda: S.state._step_status(ctx.lib, 32), acc)
    stage("second init_ (plan reused)", lambda: rls.init_(S, bd), acc)
    stage("step(32) + status (2)", lambda: S.state._step_status(ctx.lib, 32), acc)
    x = stage("download x", lambda: S.state.x.to_host(), acc)
    stage("free (solver, state, plan, vectors)", lambda: (S.__dict__.clear(), None)[1], acc)
    t0 = time.perf_counter(); rls.solve_(rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0), bd).to_host(); acc.setdefault("WHOLE solve_ (create + init + 32 it + download)", []).append(1e6 * (time.perf_counter() - t0))
for k, v in acc.items():
    print(f"{k:55s} median {sorted(v)[len(v) // 2]:9.1f} us   min {min(v):9.1f}")
