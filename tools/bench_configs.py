"""Timing of BASELINE configs 2 and 3 (FISTA+L1 4096x2048 CF32; ADMM+TV 8192x4096 F32) -- parity-test cases, not
the bench line; prints us per (outer) iteration from hipEvents."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
args = sys.argv[1:]
for a in [a for a in args if "=" in a]:  # tuning switches, e.g. slab_g=2
    k, v = a.split("=")
    ctx.tune(**{k: int(v)})
which = [a for a in args if "=" not in a] or ["fista", "pgm", "admm"]
if "fista" in which:
    M, N = 4096, 2048
    A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx)
    b = rls.DeviceVector.from_host((A @ np.ones(N, np.complex64)).astype(np.complex64), ctx)
    S = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=0.95 / (np.sqrt(M) + np.sqrt(N)) ** 2, iterations=50)
    for _ in range(20): rls.solve_(S, b)
    ctx.sync(); ctx.timer_start()
    for _ in range(20): rls.init_(S, b); ctx.lib.rls_fista_step(S.state._plan, 50)
    us = ctx.timer_stop_ms() * 1e3 / 1000
    print(f"config 2: FISTA+L1 4096x2048 CF32: {us:.2f} us/iteration ({1e6/us:.0f} it/s)")
if "pgm" in which:   # OptISTA / POGM on the configs[1] problem, 48 iterations = one resident launch
    M, N = 4096, 2048
    A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx)
    b = rls.DeviceVector.from_host((A @ np.ones(N, np.complex64)).astype(np.complex64), ctx)
    for name in ("OptISTA", "POGM"):
        for res in (1, 0):
            ctx.tune(resident=res)
            S = rls.createLinearSolver(getattr(rls, name), Ad, reg=rls.L1Regularization(1e-2), rho=0.95 / (np.sqrt(M) + np.sqrt(N)) ** 2,
                                       iterations=48, relTol=0.0)
            for _ in range(5): rls.solve_(S, b)
            ctx.sync()
            evs, walls = [], []
            for _ in range(20):   # per-solve timings, MEDIAN reported: once per process a long host wait returns 50-75 ms late (the
                t0 = time.perf_counter(); ctx.timer_start()   # runtime's first interrupt-driven wait, tools/stall_probe2.py) -- averaged
                rls.init_(S, b); S._run(S.state)              # over 20 solves that one event reads as "104 us per iteration"
                evs.append(ctx.timer_stop_ms() * 1e3 / 48); walls.append((time.perf_counter() - t0) * 1e6 / 48)
            us, wall = sorted(evs)[10], sorted(walls)[10]
            print(f"{name} + L1 4096x2048 CF32, {'resident launches' if res else 'launch per iteration'}: {us:.2f} us/iteration "
                  f"(incl. init!; host wall {wall:.2f}; median of 20 solves, slowest {max(evs):.1f})")
    ctx.tune(resident=1)
if "tall" in which:   # more row blocks than CUs (the slab kernels walk several blocks per workgroup; A/B with slab_multi=0)
    for M, N, dt in ((8192, 4096, np.float32), (8192, 2048, np.complex64), (16384, 2048, np.complex64)):
        A = make_A(M, N, 2, dt); Ad = rls.DeviceMatrix.from_host(A, ctx)
        b = rls.DeviceVector.from_host((A @ np.ones(N, dt)).astype(dt), ctx)
        tag = f"{M}x{N} {'CF32' if dt == np.complex64 else 'F32'}"
        S = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
        for _ in range(5): rls.solve_(S, b)
        ctx.sync(); ctx.timer_start()
        for _ in range(20): rls.init_(S, b); ctx.lib.rls_cgnr_step(S.state._plan, 32)
        us = ctx.timer_stop_ms() * 1e3 / (20 * 32)
        print(f"CGNR {tag}: {us:.2f} us/iteration (32-iteration solves incl. init!)")
        S = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=0.95 / (np.sqrt(M) + np.sqrt(N)) ** 2, iterations=50)
        for _ in range(5): rls.solve_(S, b)
        ctx.sync(); ctx.timer_start()
        for _ in range(20): rls.init_(S, b); ctx.lib.rls_fista_step(S.state._plan, 50)
        us = ctx.timer_stop_ms() * 1e3 / 1000
        print(f"FISTA+L1 {tag}: {us:.2f} us/iteration (50-iteration solves incl. init!)")
        del S, Ad
if "admm" in which:
    M, N = 8192, 4096
    A = make_A(M, N, 3, np.float32); Ad = rls.DeviceMatrix.from_host(A, ctx)
    b = rls.DeviceVector.from_host((A @ np.ones(N, np.float32)).astype(np.float32), ctx)
    S = rls.createLinearSolver(rls.ADMM, Ad, reg=rls.TVRegularization(1e-2, shape=(64, 64)), rho=0.1, iterations=10, iterationsCG=10, tolInner=1e-5)
    rls.solve_(S, b); rls.solve_(S, b); ctx.sync()
    dts = []
    for _rep in range(3):  # min of 3: a sporadic ~50 ms host stall on the first long wait is not the solver's
        t0 = time.perf_counter()
        for _ in range(5): rls.solve_(S, b)
        ctx.sync(); dts.append(time.perf_counter() - t0)
    dt = min(dts)
    print(f"config 3: ADMM+TV 8192x4096 F32: {1e3*dt/50:.3f} ms/outer iteration (CG its {S.state.cg_iterations})")
    t0 = time.perf_counter(); G = Ad.gram(); ctx.sync(); tg = time.perf_counter() - t0
    S = rls.createLinearSolver(rls.ADMM, Ad, AHA=G, reg=rls.TVRegularization(1e-2, shape=(64, 64)), rho=0.1, iterations=10, iterationsCG=10, tolInner=1e-5)
    rls.solve_(S, b); rls.solve_(S, b); ctx.sync()
    dts = []
    for _rep in range(3):
        t0 = time.perf_counter()
        for _ in range(5): rls.solve_(S, b)
        ctx.sync(); dts.append(time.perf_counter() - t0)
    dt = min(dts)
    print(f"config 3 in Gram mode (AHA = A'*A explicit, setup {1e3*tg:.2f} ms): {1e3*dt/50:.3f} ms/outer iteration (CG its {S.state.cg_iterations})")
