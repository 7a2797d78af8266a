#!/usr/bin/env python3
"""The config-5 shard (8192 x 8192 ComplexF32, 512 MiB: twice the Infinity Cache) runs the normal operator as two GEMVs; the
first one (t = A p) sweeps the columns upwards, so what it leaves in the cache are the LAST columns.  Does the second one
(v = A^H t) gain from walking the columns downwards (rls_tune_set("gemvt_reverse", 1))?  us per normal-operator apply, both ways,
for the shard and for shapes inside the cache (where it must not matter)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rls_amd as rls  # noqa: E402
from bench import make_A  # noqa: E402

ctx = rls.Context(0)
for (M, N, dt) in ((8192, 8192, np.complex64), (16384, 8192, np.complex64), (8192, 4096, np.float32), (4096, 2048, np.complex64)):
    A = make_A(M, N, 7, dt)
    Ad = rls.DeviceMatrix.from_host(A, ctx)
    op = rls.OperatorHandle(Ad)
    ctx.tune(fused_normal=0)
    p = rls.DeviceVector.from_host(np.ones(N, dt), ctx)
    v = rls.DeviceVector(N, dt, ctx)
    res = {}
    for rev in (0, 1, 0, 1):
        ctx.tune(gemvt_reverse=rev)
        for _ in range(5):
            op.mul_normal_(v, p)
        ctx.sync()
        best = 1e9
        for _ in range(5):
            ctx.timer_start()
            for _ in range(20):
                op.mul_normal_(v, p)
            best = min(best, ctx.timer_stop_ms() * 1e3 / 20)
        res.setdefault(rev, []).append(best)
    ctx.tune(gemvt_reverse=-1, fused_normal=1)
    by = 2.0 * M * N * np.dtype(dt).itemsize
    print(f"{M} x {N} {np.dtype(dt).name} ({by / 2 / 2**20:.0f} MiB): forward {min(res[0]):.1f} us ({by / min(res[0]) / 1e6:.2f} TB/s), "
          f"reverse {min(res[1]):.1f} us ({by / min(res[1]) / 1e6:.2f} TB/s)   runs {res}")
    del op, Ad, p, v
