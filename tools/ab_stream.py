"""Streaming (resident = 0) CGNR pipeline at 4096 x 2048 ComplexF32 under tuning variants: us per iteration from the slope of two step calls.
usage: python tools/ab_stream.py [lib.so ...]   (each library build in a child process; env AB_SHAPE=M,N, AB_DTYPE=c|f)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import math
    import numpy as np
    import rls_amd as rls
    import rls_amd._lib as L
    if os.environ.get("AB_LIB"):
        L.LIB_PATH = os.environ["AB_LIB"]
        L._lib = None
    from bench import make_A
    ctx = rls.Context(0)
    lib = ctx.lib
    M, N = (int(v) for v in os.environ.get("AB_SHAPE", "4096,2048").split(","))
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        rc = lib.rls_tune_set(ctx.handle, k.encode(), int(v))
        assert rc == 0, (kv, rc)
    dt = np.float32 if os.environ.get("AB_DTYPE", "c") == "f" else np.complex64
    A = make_A(M, N, 2, dt)
    rng = np.random.default_rng(1000)
    xt = ((rng.standard_normal(N) + 1j * rng.standard_normal(N)) / math.sqrt(2)).astype(dt)
    b = (A @ xt).astype(dt)
    Ad, bd = rls.DeviceMatrix.from_host(A, ctx), rls.DeviceVector.from_host(b, ctx)
    S = rls.createLinearSolver(rls.CGNR, Ad, iterations=2048, relTol=0.0)
    def t_of(n, reps=20):
        best = 1e9
        for _ in range(reps):
            rls.init_(S, bd); ctx.sync()
            ctx.timer_start(); lib.rls_cgnr_step(S.state._plan, n); best = min(best, ctx.timer_stop_ms())
        return best * 1e3
    t_of(32, 3)
    for rnd in range(3):
        t1, t2 = t_of(32), t_of(288)
        S.state._refresh(lib)
        import ctypes
        path = ctypes.c_int32(-1)
        lib.rls_cgnr_path(S.state._plan, ctypes.byref(path))
        print(f"  {(t2 - t1) / 256:7.3f} us/iteration   (32: {t1:7.1f} us, 288: {t2:7.1f} us)  path={path.value} res={S.state._residual:.6e}", flush=True)
    sys.exit(0)
libs = sys.argv[1:] or [""]   # library builds to compare on this box ("" = the shipped one)
for rep in range(2):
    for l in libs:
        print(l or "shipped", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "resident=0"], check=False, env={**os.environ, "AB_LIB": l})
