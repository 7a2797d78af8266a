"""Minimal dispatch sequence for rocprofv3 --pmc passes: eager launches (no hipGraph), a handful of
CGNR iterations of the headline problem.  usage: rocprofv3 --pmc FETCH_SIZE ... -- python3 tools/pmc_probe.py"""
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
for _ in range(3):  # resident path: one launch per call of 32 iterations (where the device offers it)
    rls.init_(solver, b)
    ctx.lib.rls_cgnr_step(solver.state._plan, 32)
ctx.sync()
ctx.tune(resident=0)  # the two-launch pipeline: K_A + K_R per iteration
rls.init_(solver, b)
ctx.lib.rls_cgnr_step(solver.state._plan, 6)
ctx.sync()
S2 = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=0.95 / (np.sqrt(M) + np.sqrt(N)) ** 2, iterations=32)
ctx.tune(resident=1)
for _ in range(3):
    rls.init_(S2, b)
    ctx.lib.rls_fista_step(S2.state._plan, 32)
ctx.sync()
# Gram mode (AHA explicit): the resident Gram kernel, one launch per call of 32 iterations
G = Ad.gram()
S3 = rls.createLinearSolver(rls.CGNR, Ad, AHA=G, iterations=32, relTol=0.0)
for _ in range(3):
    rls.init_(S3, b)
    ctx.lib.rls_cgnr_step(S3.state._plan, 32)
ctx.sync()
# shared-A batched solves on the matrix cores: 8 right-hand sides (half operand layout) and 16
rng = np.random.default_rng(5)
for K in (8, 16):
    X = (rng.standard_normal((N, K)) + 1j * rng.standard_normal((N, K))).astype(np.complex64)
    Bd = rls.DeviceMatrix.from_host(np.asfortranarray((A @ X).astype(np.complex64)), ctx)
    S4 = rls.createLinearSolver(rls.CGNR, Ad, iterations=6, relTol=0.0)
    rls.solve_(S4, Bd, scheduler=rls.BatchedState)
    ctx.sync()
# the same on the reference's default operator (AHA explicit): the resident batched launch (8 columns) and the streaming product
for K, res in ((8, 1), (8, 0), (16, 1)):
    ctx.tune(resident=res)
    X = (rng.standard_normal((N, K)) + 1j * rng.standard_normal((N, K))).astype(np.complex64)
    Bd = rls.DeviceMatrix.from_host(np.asfortranarray((A @ X).astype(np.complex64)), ctx)
    S5 = rls.createLinearSolver(rls.CGNR, Ad, AHA=G, iterations=32 if res else 6, relTol=0.0)
    for _ in range(2):
        rls.solve_(S5, Bd, scheduler=rls.BatchedState)
    ctx.sync()
ctx.tune(resident=1)
# BASELINE configs[0] on the single-workgroup kernel
A1 = make_A(256, 128, 1, np.float32); A1d = rls.DeviceMatrix.from_host(A1, ctx)
b1 = rls.DeviceVector.from_host((A1 @ np.ones(128, np.float32)).astype(np.float32), ctx)
S6 = rls.createLinearSolver(rls.CGNR, A1d, reg=rls.L2Regularization(1e-2), iterations=10, relTol=0.0)
for _ in range(3):
    rls.solve_(S6, b1)
ctx.sync()
p = rls.DeviceVector.from_host(np.ones(N, np.complex64), ctx); t = rls.DeviceVector(M, np.complex64, ctx); v = rls.DeviceVector(N, np.complex64, ctx)
for _ in range(3):
    Ad.gemv_(0, p, t); Ad.gemv_(2, t, v)
ctx.sync()
print("pmc probe done", flush=True)
