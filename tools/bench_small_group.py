"""K independent small CGNR problems (each its own 64 x 32 ComplexF32 / 256 x 128 Float32 matrix, 10 iterations): solve_group_ (ONE
launch: init! + all iterations of all problems) against one solve_ per problem and against ConcurrentSolves (8 streams)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rls_amd as rls
ctx = rls.Context(0)
for dt, M, N in ((np.float32, 256, 128), (np.complex64, 64, 32)):
    for K in (8, 24, 64):
        rng = np.random.default_rng(K)
        cplx = np.dtype(dt).kind == "c"
        mk = lambda *sh: ((rng.standard_normal(sh) + 1j * rng.standard_normal(sh)) if cplx else rng.standard_normal(sh)).astype(dt)
        As = [np.asfortranarray(mk(M, N)) for _ in range(K)]
        bs = [(A @ mk(N)).astype(dt) for A in As]
        mats = [rls.DeviceMatrix.from_host(A, ctx) for A in As]
        rhs = [rls.DeviceVector.from_host(b, ctx) for b in bs]
        make = lambda Ad: rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(1e-2), iterations=10, relTol=0.0)
        group = [make(Ad) for Ad in mats]
        solo = [make(Ad) for Ad in mats]
        rls.solve_group_(group, rhs); [rls.solve_(s_, b) for s_, b in zip(solo, rhs)]; ctx.sync()
        def t(f, reps=20):
            best = 1e9
            for _ in range(reps):
                t0 = time.perf_counter(); f(); ctx.sync(); best = min(best, time.perf_counter() - t0)
            return best * 1e6
        tg = t(lambda: rls.solve_group_(group, rhs))
        ts = t(lambda: [rls.solve_(s_, b) for s_, b in zip(solo, rhs)])
        print(f"{np.dtype(dt).name} {M} x {N}, K = {K:2d}: one launch {tg:8.1f} us ({tg / K:6.1f} per problem), one solve_ per problem {ts:8.1f} us ({ts / K:6.1f} per problem)", flush=True)
