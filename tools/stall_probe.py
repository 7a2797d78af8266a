import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rls_amd as rls
from bench import make_A
torch.cuda.set_device(0)
ctx = rls.Context(0)
M, N = 4096, 2048
A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx)
b = rls.DeviceVector.from_host((A @ np.ones(N, np.complex64)).astype(np.complex64), ctx)
solver = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
rls.init_(solver, b); st = solver.state; lib, h = ctx.lib, ctx.handle
def T(label, fn):
    t0 = time.perf_counter(); fn(); print(f"{label}: {1e3*(time.perf_counter()-t0):.3f} ms", flush=True)
def run(n):
    for _ in range(n):
        rls.init_(solver, b); lib.rls_cgnr_step(st._plan, 32)
for rep in range(int(os.environ.get("REPS", 4))):
    print("--- rep", rep)
    T("enqueue 2 solves", lambda: run(2))
    T("ctx.sync", ctx.sync)
    T("torch.cuda.synchronize", torch.cuda.synchronize)
    T("enqueue 10 solves", lambda: run(10))
    ctx.timer_start()
    T("timer_stop(event sync)", lambda: ctx.timer_stop_ms())
    T("ctx.sync", ctx.sync)
    T("torch.cuda.synchronize", torch.cuda.synchronize)
    T("refresh", lambda: st._refresh(lib))
