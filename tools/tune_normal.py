"""time the one-pass normal operator (slab + reduce kernels) for the slab configurations"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from bench import make_A

M, N = int(os.environ.get("M", 4096)), int(os.environ.get("N", 2048))
dt = np.complex64 if os.environ.get("DT", "c") == "c" else np.float32
ctx = rls.Context(0)
A = make_A(M, N, 2, dt)
Ad = rls.DeviceMatrix.from_host(A, ctx)
rng = np.random.default_rng(0)
p_h = (rng.standard_normal(N) + (1j * rng.standard_normal(N) if dt == np.complex64 else 0)).astype(dt)
p = rls.DeviceVector.from_host(p_h, ctx)
v = rls.DeviceVector(N, dt, ctx)
A64 = A.astype(np.complex128)
want = A64.conj().T @ (A64 @ p_h)
reps = 200
for g in (8, 4):
    ctx.tune(slab_g=g)
    op = rls.OperatorHandle(Ad)
    for mode in (1, 0):
        ctx.tune(fused_normal=mode)
        for _ in range(10): op.mul_normal_(v, p)
        err = np.linalg.norm(v.to_host() - want) / np.linalg.norm(want)
        ctx.sync(); ctx.timer_start()
        for _ in range(reps): op.mul_normal_(v, p)
        us = ctx.timer_stop_ms() / reps * 1e3
        print(f"slab_g={g} fused={mode}: {us:.2f} us per normal-operator apply, rel err {err:.2e}", flush=True)
    del op
