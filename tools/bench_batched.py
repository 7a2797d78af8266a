"""BASELINE config 4, shared-A flavour: K CGNR solves sharing one pass over A per iteration (one GPU).
usage: bench_batched.py [K,K,...] [gram=1] [tune_key=value ...]   (gram=1: AHA = A' * A explicit, the reference's default)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
M, N = 4096, 2048
A = make_A(M, N, 4); Ad = rls.DeviceMatrix.from_host(A, ctx)
rng = np.random.default_rng(5)
Ks = tuple(int(k) for k in sys.argv[1].split(',')) if len(sys.argv) > 1 else (1, 2, 4, 8, 16, 32, 64)
gram = False
for kv in sys.argv[2:]:  # tuning switches, e.g. skinny_tu=0 skinny_tu_window=8
    k, v = kv.split('=')
    if k == "gram":
        gram = bool(int(v))
    else:
        ctx.tune(**{k: int(v)})
kw = dict(AHA=Ad.gram()) if gram else {}
for K in Ks:
    X = (rng.standard_normal((N, K)) + 1j * rng.standard_normal((N, K))).astype(np.complex64)
    B = np.asfortranarray((A @ X).astype(np.complex64))
    Bd = rls.DeviceMatrix.from_host(B, ctx)
    S = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0, **kw)
    lib, h = ctx.lib, ctx.handle
    if K > 1:
        xs = rls.solve_(S, Bd, scheduler=rls.BatchedState)
        st = S.state
        def run(n):
            for _ in range(n):
                rls._lib.check(h, lib.rls_cgnr_init_batched(st._plan, Bd.ptr, Bd.lda, 0.0, 0.0, 32), "init")
                rls._lib.check(h, lib.rls_cgnr_step(st._plan, 32), "step")
    else:
        bd = Bd.column(0)
        xs = [rls.solve_(S, bd)]
        def run(n):
            for _ in range(n):
                rls.init_(S, bd); rls._lib.check(h, lib.rls_cgnr_step(S.state._plan, 32), "step")
    err = max(np.linalg.norm(xs[j].to_host() - X[:, j]) / np.linalg.norm(X[:, j]) for j in range(K))
    run(3); ctx.sync(); ctx.timer_start()
    reps = 20
    run(reps)
    us = ctx.timer_stop_ms() * 1e3 / (reps * 32)
    print(("gram " if gram else "") + f"K={K:2d}: {us:7.2f} us per batched iteration = {us/K:6.2f} us per solve-iteration ({K*1e6/us:8.0f} solve-it/s), max rel err {err:.1e}", flush=True)
