import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
M, N, K = 4096, 2048, 8
A = make_A(M, N, 4); Ad = rls.DeviceMatrix.from_host(A, ctx)
rng = np.random.default_rng(5)
X = (rng.standard_normal((N, K)) + 1j * rng.standard_normal((N, K))).astype(np.complex64)
Bd = rls.DeviceMatrix.from_host(np.asfortranarray((A @ X).astype(np.complex64)), ctx)
G = Ad.gram(); ctx.sync(); print("gram ok", flush=True)
S = rls.createLinearSolver(rls.CGNR, Ad, AHA=G, iterations=2000, relTol=0.0)
rls.init_(S, Bd, scheduler=rls.BatchedState); ctx.sync(); print("init ok", flush=True)
st = S.state
for n in (2, 4, 8, 32):
    st._step(n); ctx.sync(); print("step", n, "ok", [s.iteration for s in st.status()], flush=True)
