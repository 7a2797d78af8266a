"""per-call cadence of the reference's solve! loop (one iterate + one done() check per iteration, src/RegularizedLeastSquares.jl:103-117):
rls_*_step_status(plan, 1) from Python, with the status mailbox (a kernel writing into pinned host memory + a host spin) and with
hipMemcpyAsync + stream wait.  usage: bench_cadence.py [mailbox=0|1 ...]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
lib, h, L = ctx.lib, ctx.handle, rls._lib
M, N = 4096, 2048
A = make_A(M, N, 2); Ad = rls.DeviceMatrix.from_host(A, ctx)
rng = np.random.default_rng(3)
b = rls.DeviceVector.from_host((A @ (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)).astype(np.complex64), ctx)
rho = float(0.9 / np.linalg.norm(A.astype(np.complex128), 2) ** 2)

def cadence(make, step_status, status_t, n_it):
    S = make()
    rls.solve_(S, b)
    st_ = status_t()
    plan = S.state._admm if hasattr(S.state, "_admm") and S.state._admm else S.state._plan
    def run():
        rls.init_(S, b)
        for _ in range(n_it):
            step_status(plan, st_)
    run(); ctx.sync()
    best = float("inf")
    for _ in range(8):
        t0 = time.perf_counter(); run(); best = min(best, time.perf_counter() - t0)
    assert st_.iteration == n_it, (st_.iteration, n_it)
    return 1e6 * best / n_it

# the floor of a one-kernel call: BASELINE configs[0] (256 x 128 Float32) on the small-system kernel (one launch, ~2 us of work)
from bench import make_A as _mk
As = np.asfortranarray(np.random.default_rng(5).standard_normal((256, 128)).astype(np.float32))
Asd = rls.DeviceMatrix.from_host(As, ctx)
bs = rls.DeviceVector.from_host((As @ np.random.default_rng(6).standard_normal(128).astype(np.float32)).astype(np.float32), ctx)
_b = b
for mb in (2, 1, 0):
    ctx.tune(status_mailbox=mb)
    b = bs
    c = cadence(lambda: rls.createLinearSolver(rls.CGNR, Asd, iterations=32, relTol=0.0),
                lambda p, st_: L.check(h, lib.rls_cgnr_step_status(p, 1, C.byref(st_)), "cgnr"), L.CgnrStatus, 32)
    print(f"status_mailbox={mb}: CGNR 256 x 128 Float32 (single-workgroup kernel) {c:6.1f} us per iterate call", flush=True)
b = _b
for mb in (2, 1, 0):
    ctx.tune(status_mailbox=mb)
    c = cadence(lambda: rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0),
                lambda p, st_: L.check(h, lib.rls_cgnr_step_status(p, 1, C.byref(st_)), "cgnr"), L.CgnrStatus, 32)
    f = cadence(lambda: rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=rho, iterations=32, relTol=0.0),
                lambda p, st_: L.check(h, lib.rls_fista_step_status(p, 1, C.byref(st_)), "fista"), L.FistaStatus, 32)
    a = cadence(lambda: rls.createLinearSolver(rls.ADMM, Ad, reg=rls.L1Regularization(1e-2), rho=0.1, iterations=8, iterationsCG=10,
                                               tolInner=1e-5, absTol=0.0, relTol=0.0),
                lambda p, st_: L.check(h, lib.rls_admm_step_status(p, 1, C.byref(st_), None, 0), "admm"), L.AdmmStatus, 8)
    print(f"status_mailbox={mb}: CGNR {c:6.1f} us per iterate call, FISTA + L1 {f:6.1f}, ADMM + L1 {a:6.1f} us per outer iteration", flush=True)
# the reference constructor's default operator (AHA = A' * A explicit): resident Gram kernel, with and without server mode
G = Ad.gram()
for srv in (1, 0):
    ctx.tune(status_mailbox=2, resident_server=srv)
    c = cadence(lambda: rls.createLinearSolver(rls.CGNR, Ad, AHA=G, iterations=32, relTol=0.0),
                lambda p, st_: L.check(h, lib.rls_cgnr_step_status(p, 1, C.byref(st_)), "cgnr"), L.CgnrStatus, 32)
    f = cadence(lambda: rls.createLinearSolver(rls.FISTA, Ad, AHA=G, reg=rls.L1Regularization(1e-2), rho=rho, iterations=32, relTol=0.0),
                lambda p, st_: L.check(h, lib.rls_fista_step_status(p, 1, C.byref(st_)), "fista"), L.FistaStatus, 32)
    a = cadence(lambda: rls.createLinearSolver(rls.ADMM, Ad, AHA=G, reg=rls.L1Regularization(1e-2), rho=0.1, iterations=8, iterationsCG=10,
                                               tolInner=1e-5, absTol=0.0, relTol=0.0),
                lambda p, st_: L.check(h, lib.rls_admm_step_status(p, 1, C.byref(st_), None, 0), "admm"), L.AdmmStatus, 8)
    print(f"on the explicit Gram matrix, resident_server={srv}: CGNR {c:6.1f} us per iterate call, FISTA + L1 {f:6.1f}, ADMM + L1 {a:6.1f} us per outer iteration", flush=True)
ctx.tune(resident_server=1)
