"""Shapes with more 16-row blocks than the chip has CUs (BASELINE configs[2]: 8192 x 4096 Float32 = 512 blocks): the one-pass normal
operator with one workgroup per block (rls_tune_set("slab_multi", 0): the blocks run in rounds, each round starting from nothing) against
one workgroup walking several blocks, the next one streaming in under the products of the current one (slab_multi = 1, the default).
us per normal-operator apply (slab + reduce kernels) and the error against float64, same process, interleaved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from bench import make_A

ctx = rls.Context(0)
reps = int(os.environ.get("REPS", 300))
shapes = [(8192, 4096, "f"), (16384, 4096, "f"), (8192, 2048, "c"), (16384, 2048, "c"), (12288, 2048, "c"), (8192 + 64, 4096, "f"),
          (8192, 4000, "f"), (4096, 2048, "c"),
          # ComplexF32 with N in (2048, 4096]: on the two-GEMV path until the exchange planes of that layout fitted the LDS (round 5)
          (4096, 4096, "c"), (2048, 4096, "c"), (3200, 3072, "c"), (8192, 4096, "c"),
          # ragged columns: the slab's column rounds past N re-read round 0
          (8192, 3072, "f"), (8192, 1536, "c"), (16384, 1280, "c")]
for M, N, t in shapes:
    dt = np.complex64 if t == "c" else np.float32
    A = make_A(M, N, 2, dt)
    Ad = rls.DeviceMatrix.from_host(A, ctx)
    rng = np.random.default_rng(0)
    p_h = (rng.standard_normal(N) + (1j * rng.standard_normal(N) if t == "c" else 0)).astype(dt)
    p = rls.DeviceVector.from_host(p_h, ctx)
    v = rls.DeviceVector(N, dt, ctx)
    A64 = A.astype(np.complex128 if t == "c" else np.float64)
    want = A64.conj().T @ (A64 @ p_h)
    out = {}
    for rnd in range(3):
        for multi in (0, 1):
            ctx.tune(slab_multi=multi)
            op = rls.OperatorHandle(Ad)
            for _ in range(10): op.mul_normal_(v, p)
            err = np.linalg.norm(v.to_host() - want) / np.linalg.norm(want)
            ctx.sync(); ctx.timer_start()
            for _ in range(reps): op.mul_normal_(v, p)
            us = ctx.timer_stop_ms() / reps * 1e3
            out.setdefault(multi, []).append((us, err))
            del op
    if (t == "c" and N > 2048) or N % 1024:
        ctx.tune(fused_normal=0)
        op = rls.OperatorHandle(Ad)
        for _ in range(10): op.mul_normal_(v, p)
        ctx.sync(); ctx.timer_start()
        for _ in range(reps): op.mul_normal_(v, p)
        print(f"{M}x{N} {t}32 two GEMVs: {ctx.timer_stop_ms() / reps * 1e3:.2f} us per apply", flush=True)
        ctx.tune(fused_normal=1)
        del op
    for multi in (0, 1):
        us = sorted(u for u, _ in out[multi])
        print(f"{M}x{N} {t}32 slab_multi={multi}: {us[len(us)//2]:.2f} us per apply (min {us[0]:.2f}), rel err {out[multi][0][1]:.2e}", flush=True)
ctx.tune(slab_multi=1)
