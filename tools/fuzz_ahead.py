"""stress of the listening kernels that run one iteration ahead (resident_ahead = 1): many solves driven by one-iterate commands, random
command sizes and random pauses, downloads in between; every solve must give the bits of ONE launch of the same number of iterations.
usage: python tools/fuzz_ahead.py [rounds]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
lib, L = ctx.lib, rls._lib
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(7)
M, N = 4096, 2048
A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx); Gd = Ad.gram()
b = rls.DeviceVector.from_host((A @ (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)).astype(np.complex64), ctx)
As = np.asfortranarray(rng.standard_normal((256, 128)).astype(np.float32)); Asd = rls.DeviceMatrix.from_host(As, ctx)
bs = rls.DeviceVector.from_host((As @ rng.standard_normal(128).astype(np.float32)).astype(np.float32), ctx)
rho = float(0.9 / np.linalg.norm(A.astype(np.complex128), 2) ** 2)
iters = 24
cases = {
    "cgnr matrix-free": (lambda: rls.createLinearSolver(rls.CGNR, Ad, iterations=iters, relTol=0.0), b, False),
    "cgnr gram": (lambda: rls.createLinearSolver(rls.CGNR, Ad, AHA=Gd, iterations=iters, relTol=0.0), b, False),
    "fista matrix-free": (lambda: rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=rho, iterations=iters, relTol=0.0), b, True),
    "fista gram": (lambda: rls.createLinearSolver(rls.FISTA, Ad, AHA=Gd, reg=rls.L1Regularization(1e-2), rho=rho, iterations=iters, relTol=0.0), b, True),
    "cgnr small": (lambda: rls.createLinearSolver(rls.CGNR, Asd, iterations=iters, relTol=0.0), bs, False),
    "fista small": (lambda: rls.createLinearSolver(rls.FISTA, Asd, reg=rls.L1Regularization(1e-2), rho=1e-3, iterations=iters, relTol=0.0), bs, True),
}
bad = 0
AHEAD = int(os.environ.get("AHEAD", "1"))
ctx.tune(resident_ahead=AHEAD)
print(f"resident_ahead = {AHEAD}")
for name, (make, rhs, fista) in cases.items():
    S = make()
    x_once = rls.solve_(S, rhs).to_host()
    st = L.FistaStatus() if fista else L.CgnrStatus()
    step = lib.rls_fista_step_status if fista else lib.rls_cgnr_step_status
    fails = wrong = 0
    for r in range(rounds):
        rls.init_(S, rhs)
        done = 0
        while done < iters:
            n = int(rng.integers(1, 4)); n = min(n, iters - done)
            assert step(S.state._plan, n, C.byref(st)) == 0
            done += n
            assert st.iteration == done, (name, st.iteration, done)
            u = rng.random()
            if u < 0.05:
                time.sleep(1e-3)          # longer than the idle time: the kernel leaves on its own (behind its pass ahead)
            elif u < 0.10:
                S.state._refresh(lib); S.state.x.to_host()   # told to leave behind its pass ahead
        S.state._refresh(lib)
        xr = S.state.x.to_host()
        if not np.array_equal(xr, x_once):
            fails += 1   # (legitimate when the plan went over to the per-iteration pipeline after short kernel lives: other summation order)
            if np.linalg.norm(xr - x_once) > 2e-5 * np.linalg.norm(x_once):
                wrong += 1
    print(f"{name:18s}: {rounds} solves of {iters} iterations by commands of 1-3: {fails} not the bits of one launch, {wrong} WRONG (> 2e-5)", flush=True)
    bad += wrong
sys.exit(1 if bad else 0)
