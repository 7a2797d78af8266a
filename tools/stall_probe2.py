"""the first LONG host wait of a process returns ~50 ms late, once: wall clock vs hipEvents per repetition (why bench.py warms up outside W)"""
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
def run(n):
    for _ in range(n):
        rls.init_(solver, b); lib.rls_cgnr_step(st._plan, 32)
run(4); ctx.sync()
for rep in range(25):
    ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter(); ctx.timer_start()
    run(100)
    t1 = time.perf_counter()
    ev = ctx.timer_stop_ms()
    t2 = time.perf_counter()
    print(f"rep {rep:2d}: enqueue {1e3*(t1-t0):6.2f} ms  events {ev:7.2f} ms  wall {1e3*(t2-t0):7.2f} ms", flush=True)
