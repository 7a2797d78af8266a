"""Cadence of the reference's solve! loop on a kernel that stays (rls_cgnr_step_status(plan, 1) per call) with and without the iteration
it computes ahead of the next command (rls_tune_set("resident_ahead")): us per call, and the iterates against a launch-per-call run.
usage: python tools/ab_ahead.py [gram] [fista]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
lib, L = ctx.lib, rls._lib
M, N = 4096, 2048
A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx)
rng = np.random.default_rng(3)
b = rls.DeviceVector.from_host((A @ (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)).astype(np.complex64), ctx)
n_it = 200


GRAM = "gram" in sys.argv[1:]    # AHA = A' * A explicit (the reference constructors' default for a dense matrix)
FISTA = "fista" in sys.argv[1:]  # FISTA + L1 instead of CGNR
Gd = Ad.gram() if GRAM else None
RHO = float(0.9 / np.linalg.norm(A.astype(np.complex128), 2) ** 2)


def run(ahead, server=1):
    ctx.tune(resident_server=server, resident_ahead=ahead)
    kw = dict(AHA=Gd) if GRAM else {}
    if FISTA:
        S = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=RHO, iterations=n_it, relTol=0.0, **kw)
    else:
        S = rls.createLinearSolver(rls.CGNR, Ad, iterations=n_it, relTol=0.0, **kw)
    st = L.FistaStatus() if FISTA else L.CgnrStatus()
    step = lib.rls_fista_step_status if FISTA else lib.rls_cgnr_step_status
    res = []
    def once(record):
        rls.init_(S, b)
        for _ in range(n_it):
            rc = step(S.state._plan, 1, C.byref(st))
            assert rc == 0, rc
            if record:
                res.append((st.iteration, st.residual))
    once(False); ctx.sync()
    best = float("inf")
    for _ in range(6):
        t0 = time.perf_counter(); once(False); best = min(best, time.perf_counter() - t0)
    once(True)
    S.state._refresh(lib)
    x = S.state.x.to_host()
    return 1e6 * best / n_it, res, x


base_t, base_res, base_x = run(0, 0)
print(f"launch per call        : {base_t:7.2f} us per iterate call")
ref = None
for ahead in (0, 1, 0, 1):
    t, res, x = run(ahead)
    if ref is None:
        ref = (res, x)
    same = res == ref[0] and np.array_equal(x, ref[1])
    print(f"server, ahead = {ahead}      : {t:7.2f} us per iterate call   statuses and x identical to the kernel that does not run ahead: {same}")
