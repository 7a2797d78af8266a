"""Differential fuzz of the resident kernels on ragged shapes: random M, N (within the resident limits) and dtype; CGNR
and FISTA + L1 (matrix-free and Gram mode) through the resident launch against the per-iteration pipelines of the same
plan (validated against the float64 oracle by the parity tests).  usage: fuzz_resident.py [cases] [seed]"""
import sys, os, math, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import rls_amd as rls
import rls_oracle as O  # problem generator only

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = rls.default_context(0)
lib = ctx.lib
worst = 0.0
for c in range(cases):
    cplx = bool(rng.integers(2))
    dt = np.complex64 if cplx else np.float32
    V = 2 if cplx else 4
    N = int(rng.integers(130 if cplx else 260, 2048 if cplx else 4096)) // V * V
    M = max(N, int(rng.integers(N, 4096))) // V * V
    gram = bool(rng.integers(2))
    A, xt, b = O.make_problem(M, N, dt, int(rng.integers(1 << 30)))
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    kw = dict(AHA=Ad.gram()) if gram else {}
    lam = float(rng.choice([0.0, 1e-3, 1e-1]))
    its = int(rng.integers(3, 20))
    rho = 0.9 / (math.sqrt(M) + math.sqrt(N)) ** 2
    for name, make in (("cgnr", lambda: rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(lam), iterations=its, relTol=0.0, **kw)),
                       ("fista", lambda: rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(lam), rho=rho, iterations=its, relTol=0.0,
                                                                 restart="gradient" if c % 2 else "none", **kw)),
                       # ADMM: the inner cg! (warm start, entry folded into the resident launch) -- 3 outer x `its` inner iterations
                       ("admm", lambda: rls.createLinearSolver(rls.ADMM, Ad, reg=rls.L1Regularization(max(lam, 1e-3)), rho=0.1, iterations=3,
                                                                iterationsCG=min(its, 10), tolInner=1e-6, absTol=0.0, relTol=0.0, **kw))):
        S = make()
        got = {}
        for res in (1, 0):
            ctx.tune(resident=res)
            got[res] = rls.solve_(S, bd).to_host()
            if res == 1:
                pth = ctypes.c_int32(-1)
                if name != "admm":
                    (lib.rls_cgnr_path if name == "cgnr" else lib.rls_fista_path)(S.state._plan, ctypes.byref(pth))
        ctx.tune(resident=1)
        err = float(np.linalg.norm(got[1] - got[0]) / max(np.linalg.norm(got[0]), 1e-30))
        worst = max(worst, err)
        flag = "" if err < 2e-5 and np.isfinite(got[1]).all() else "   <-- MISMATCH"
        print(f"{c:3d} {name:5s} {'gram' if gram else 'free'} {np.dtype(dt).name:9s} {M:5d}x{N:<5d} lam={lam:<6g} its={its:2d} path={pth.value} "
              f"resident vs pipeline {err:.2e}{flag}", flush=True)
        assert not flag
print(f"worst relative difference {worst:.2e} over {cases} cases")
