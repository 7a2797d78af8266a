"""host-side cost of the ctypes call sequence of one CGNR iteration (eager launches vs hipGraph replay)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
M, N = 4096, 2048
A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx)
b = rls.DeviceVector.from_host((A @ np.ones(N, np.complex64)).astype(np.complex64), ctx)
for use_graph in (1, 0):
  for pipe in (1, 0):
    ctx.tune(use_graph=use_graph, cgnr_pipeline=pipe)
    solver = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
    rls.init_(solver, b); st = solver.state; lib, h = ctx.lib, ctx.handle
    lib.rls_cgnr_step(st._plan, 32); ctx.sync()
    for rep in range(2):
        t0 = time.perf_counter(); 
        for _ in range(10):
            rls.init_(solver, b)
            t1 = time.perf_counter()
            lib.rls_cgnr_step(st._plan, 32)
        t2 = time.perf_counter(); ctx.sync(); t3 = time.perf_counter()
        print(f"graph={use_graph} pipe={pipe}: enqueue 10x(init+32 steps) {1e3*(t2-t0):.2f} ms, +sync {1e3*(t3-t0):.2f} ms -> {1e6*(t3-t0)/320:.1f} us/iter", flush=True)
