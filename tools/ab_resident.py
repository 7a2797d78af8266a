"""A/B of the resident CGNR kernel between library builds on ONE box: us per iteration INSIDE the kernel =
(t(launch of n2 iterations) - t(launch of n1)) / (n2 - n1), hipEvents, best of several; each library in a child process.
usage: python tools/ab_resident.py [lib.so ...]   (default: the shipped library)   env AB_LAMBDA: L2 weight (default 0)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import math
    import numpy as np
    import rls_amd as rls
    import rls_amd._lib as L
    if sys.argv[2] != "-":
        L.LIB_PATH = sys.argv[2]
        L._lib = None
    from bench import make_A
    ctx = rls.Context(0)
    lib = ctx.lib
    M, N = 4096, 2048
    A = make_A(M, N, 2)
    rng = np.random.default_rng(1000)
    xt = ((rng.standard_normal(N) + 1j * rng.standard_normal(N)) / math.sqrt(2)).astype(np.complex64)
    b = (A @ xt).astype(np.complex64)
    Ad, bd = rls.DeviceMatrix.from_host(A, ctx), rls.DeviceVector.from_host(b, ctx)
    lam = float(os.environ.get("AB_LAMBDA", "0"))
    kw = dict(reg=rls.L2Regularization(lam)) if lam > 0 else {}
    S = rls.createLinearSolver(rls.CGNR, Ad, iterations=1024, relTol=0.0, **kw)
    def t_of(n, reps=30):
        best = 1e9
        for _ in range(reps):
            rls.init_(S, bd); ctx.sync()
            ctx.timer_start(); lib.rls_cgnr_step(S.state._plan, n); best = min(best, ctx.timer_stop_ms())
        return best * 1e3
    t_of(32, 5)
    for rnd in range(3):
        t1, t2 = t_of(32), t_of(288)
        S.state._refresh(lib)
        print(f"  {(t2 - t1) / 256:7.3f} us/iteration in the kernel   (32: {t1:7.1f} us, 288: {t2:7.1f} us)  it={S.state.iteration} res={S.state._residual:.6e}", flush=True)
    sys.exit(0)
libs = sys.argv[1:] or ["-"]
for rep in range(2):
    for l in libs:
        print(l, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", l], check=False)
