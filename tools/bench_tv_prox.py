"""TV prox (FGP, 10 iterations) of one image: the register-resident single-workgroup kernel against the chip-wide launch-per-step
sequence (tv_fused_2d = 0).  usage: bench_tv_prox.py [nx ny ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rls_amd as rls
ctx = rls.Context(0)
shapes = [(64, 64), (96, 80), (90, 91), (128, 64)]
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
for shape in shapes:
    n = shape[0] * shape[1]
    x = np.random.default_rng(n).standard_normal(n).astype(np.float32)
    out = []
    for fused in (1, 0):
        ctx.tune(tv_fused_2d=fused)
        xd = rls.DeviceVector.from_host(x, ctx)
        for _ in range(3):
            rls.prox_(rls.TVRegularization, xd, 0.3, shape=shape)
        ctx.sync(); ctx.timer_start()
        for _ in range(50):
            rls.prox_(rls.TVRegularization, xd, 0.3, shape=shape)
        out.append(ctx.timer_stop_ms() * 1e3 / 50)
    ctx.tune(tv_fused_2d=1)
    print(f"{shape[0]} x {shape[1]} Float32 ({n} pixels): register-resident kernel {out[0]:7.1f} us per prox, launch-per-step sequence {out[1]:7.1f} us")
