"""where do the gradient restarts of POGM fall?  oracle (float64 / float32) iterate by iterate, the device path iterate by
iterate (callbacks), the deferred sequence (resident = 0) and the resident launches: sigma after every iteration / at the end"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import rls_amd as rls
import rls_oracle as O
ctx = rls.Context(0)
M, N, dt, its = 4096, 2048, np.complex64, int(sys.argv[1]) if len(sys.argv) > 1 else 30
A, xt, b = O.make_problem(M, N, dt, 5)
A64, b64 = A.astype(np.complex128), b.astype(np.complex128)
rho = 0.95 / np.linalg.norm(A64, 2) ** 2
lam = 1e-2 * np.max(np.abs(A64.conj().T @ b64))
kw = dict(restart="gradient", sigma_fac=0.97)
for nm, (A_, b_) in (("oracle f64", (A64, b64)), ("oracle f32", (A, b))):
    r = O.POGM(A_, reg=O.L1Regularization(lam), rho=rho, iterations=its, relTol=0.0, **kw)
    r.init(b_)
    sig = []
    while r.iterate() is not None:
        sig.append(float(r.sigma))
    print(nm, "restarts at", [i for i, s in enumerate(sig) if s == 1.0], "theta", float(r.theta), "sigma", float(r.sigma))
Ad, bd = rls.DeviceMatrix.from_host(A, ctx), rls.DeviceVector.from_host(b, ctx)
sol = rls.createLinearSolver(rls.POGM, Ad, reg=rls.L1Regularization(lam), rho=rho, iterations=its, relTol=0.0, **kw)
sig = []
rls.solve_(sol, bd, callbacks=lambda s_, it: sig.append(s_.state.sigma))
print("device stepwise restarts at", [i - 1 for i, s in enumerate(sig) if s == 1.0 and i > 0], "theta", sol.state.theta, "sigma", sol.state.sigma)
for res in (0, 1):
    ctx.tune(resident=res)
    rls.solve_(sol, bd)
    import time
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(5):
        rls.solve_(sol, bd)
    ctx.sync(); dt_ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{dt_ms:.3f} ms per solve_ of {its} iterations = {1e3 * dt_ms / its:.1f} us per iteration;", end=" ")
    print("resident" if res else "sequence", "theta", sol.state.theta, "theta_old", sol.state.thetaold, "sigma", sol.state.sigma, "gamma", sol.state.gamma,
          "iteration", sol.state.iteration)
