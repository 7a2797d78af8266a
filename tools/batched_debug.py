import faulthandler, sys, os
faulthandler.enable(); faulthandler.dump_traceback_later(40, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.path.insert(0, "oracle")
import rls_oracle as O
import rls_amd as rls
def P(*a): print(*a, flush=True)
ctx = rls.default_context(0)
A, X, B = O.make_problem(96, 40, np.complex64, 23, n_rhs=3)
Ad = rls.DeviceMatrix.from_host(A)
S = rls.createLinearSolver(rls.CGNR, Ad, iterations=6, relTol=0.0)
P("init")
rls.init_(S, rls.DeviceMatrix.from_host(np.asfortranarray(B)), scheduler=rls.BatchedState)
P("state", type(S.state).__name__)
ctx.sync(); P("synced after init")
P([ (s.iteration, s.done, s.residual) for s in S.state.status()])
for k in range(8):
    r = rls.iterate(S); ctx.sync()
    P(k, r is not None, [ (s.iteration, s.done, round(s.residual,5)) for s in S.state.status()])
