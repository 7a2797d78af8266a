"""sweep of the single-workgroup FGP threshold (tv_fused_max_n) against the 2-launches-per-iteration graph path"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
ctx = rls.Context(0)
rng = np.random.default_rng(0)
for shape in ((16, 16), (32, 32), (64, 64), (128, 128), (256, 256)):
    n = shape[0] * shape[1]
    x = rng.standard_normal(n).astype(np.float32)
    xd = rls.DeviceVector.from_host(x, ctx)
    reg = rls.TVRegularization(0.1, shape=shape)
    for maxn in (1 << 30, 0):
        ctx.tune(tv_fused_max_n=min(maxn, 2**31 - 1))
        for _ in range(5): reg.prox_(xd, 0.1)
        ctx.sync(); ctx.timer_start()
        for _ in range(50): reg.prox_(xd, 0.1)
        us = ctx.timer_stop_ms() * 1e3 / 50
        print(f"{shape}: {'fused single-WG' if maxn else 'multi-launch':16s} {us:8.1f} us per prox (10 FGP iterations)")
