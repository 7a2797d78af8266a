"""Gram mode (AHA explicit) for ComplexF32 with N in (2048, 4096]: more 8-row blocks of AHA than CUs.  CGNR and FISTA + L1 per iteration on
the one-launch-per-iteration Gram kernels with one workgroup per block in rounds (slab_multi=0) against one workgroup per CU walking its
blocks (default)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
for M, N, dt in ((4096, 4096, np.complex64), (4096, 3072, np.complex64), (4096, 2560, np.complex64)):
    A = make_A(M, N, 2, dt); Ad = rls.DeviceMatrix.from_host(A, ctx)
    b = rls.DeviceVector.from_host((A @ np.ones(N, dt)).astype(dt), ctx)
    G = Ad.gram()
    for multi in (0, 1):
        ctx.tune(slab_multi=multi)
        S = rls.createLinearSolver(rls.CGNR, Ad, AHA=G, iterations=32, relTol=0.0)
        for _ in range(5): rls.solve_(S, b)
        ctx.sync(); ctx.timer_start()
        for _ in range(20): rls.init_(S, b); ctx.lib.rls_cgnr_step(S.state._plan, 32)
        us = ctx.timer_stop_ms() * 1e3 / (20 * 32)
        F = rls.createLinearSolver(rls.FISTA, Ad, AHA=G, reg=rls.L1Regularization(1e-2), rho=0.95 / (np.sqrt(M) + np.sqrt(N)) ** 2, iterations=50)
        for _ in range(5): rls.solve_(F, b)
        ctx.sync(); ctx.timer_start()
        for _ in range(20): rls.init_(F, b); ctx.lib.rls_fista_step(F.state._plan, 50)
        usf = ctx.timer_stop_ms() * 1e3 / 1000
        print(f"Gram mode {M}x{N} {np.dtype(dt).name} slab_multi={multi}: CGNR {us:.2f} us/it, FISTA+L1 {usf:.2f} us/it; AHA {N*N*A.itemsize/2**20:.0f} MiB", flush=True)
        del S, F
    ctx.tune(slab_multi=1)
