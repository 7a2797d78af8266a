"""The headline solve only, for the two counter passes bench.py runs as child processes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one each):
three resident launches of 32 iterations, then six iterations on the two-launch pipeline.  Eager launches, no hipGraph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
ctx.tune(use_graph=0)
M, N = 4096, 2048
A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx)
b = rls.DeviceVector.from_host((A @ np.ones(N, np.complex64)).astype(np.complex64), ctx)
solver = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
for _ in range(3):
    rls.init_(solver, b)
    ctx.lib.rls_cgnr_step(solver.state._plan, 32)
ctx.sync()
ctx.tune(resident=0)
rls.init_(solver, b)
ctx.lib.rls_cgnr_step(solver.state._plan, 6)
ctx.sync()
