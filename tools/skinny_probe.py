"""Per-kernel timing of the matrix-core batched path under tuning / diagnostic variants (rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
M, N = 4096, 2048
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
variants = [v for v in (sys.argv[2] if len(sys.argv) > 2 else "base").split(",")]
A = make_A(M, N, 4); Ad = rls.DeviceMatrix.from_host(A, ctx)
rng = np.random.default_rng(5)
X = (rng.standard_normal((N, K)) + 1j * rng.standard_normal((N, K))).astype(np.complex64)
B = np.asfortranarray((A @ X).astype(np.complex64))
Bd = rls.DeviceMatrix.from_host(B, ctx)
lib, h = ctx.lib, ctx.handle
for var in variants:
    kv = dict(x.split("=") for x in var.split("+") if "=" in x)
    for k, v in kv.items():
        ctx.tune(**{k: int(v)})
    S = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
    rls.solve_(S, Bd, scheduler=rls.BatchedState)
    st = S.state
    def run(n):
        for _ in range(n):
            rls._lib.check(h, lib.rls_cgnr_init_batched(st._plan, Bd.ptr, Bd.lda, 0.0, 0.0, 32), "init")
            rls._lib.check(h, lib.rls_cgnr_step(st._plan, 32), "step")
    run(3); ctx.sync(); ctx.timer_start()
    reps = 20
    run(reps)
    us = ctx.timer_stop_ms() * 1e3 / (reps * 32)
    print(f"{var:40s} K={K}: {us:7.2f} us per batched iteration", flush=True)
    for k in kv:
        ctx.tune(**{k: {"skinny_t_waves": 4, "skinny_v_waves": 4, "skinny_t_u": 4, "skinny_v_u": 1, "skinny_half": 1, "skinny_t_roll": 8, "skinny_v_roll": 2}.get(k, 0)})
    del S, st
