"""us per iteration INSIDE fista_resident_kernel (FISTA + L1, BASELINE configs[1] shape) = (t(launch of 288) - t(launch of 32)) / 256, hipEvents,
best of several; library builds as arguments (each in a child process), as tools/ab_resident.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import rls_amd as rls
    import rls_amd._lib as L
    if sys.argv[2] != "-":
        L.LIB_PATH = sys.argv[2]; L._lib = None
    from bench import make_A
    ctx = rls.Context(0)
    lib = ctx.lib
    M, N = 4096, 2048
    A = make_A(M, N, 2)
    b = (A @ np.ones(N, np.complex64)).astype(np.complex64)
    Ad, bd = rls.DeviceMatrix.from_host(A, ctx), rls.DeviceVector.from_host(b, ctx)
    for restart in ("none", "gradient"):
        S = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=0.95 / (np.sqrt(M) + np.sqrt(N)) ** 2, iterations=1024, relTol=0.0, restart=restart)
        def t_of(n, reps=20):
            best = 1e9
            for _ in range(reps):
                rls.init_(S, bd); ctx.sync()
                ctx.timer_start(); lib.rls_fista_step(S.state._plan, n); best = min(best, ctx.timer_stop_ms())
            return best * 1e3
        t_of(32, 3)
        for rnd in range(2):
            t1, t2 = t_of(32), t_of(288)
            print(f"  restart={restart:8s} {(t2 - t1) / 256:7.3f} us/iteration in the kernel   (32: {t1:7.1f} us, 288: {t2:7.1f} us)", flush=True)
    sys.exit(0)
for l in (sys.argv[1:] or ["-"]):
    print(l, flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", l], check=False)
