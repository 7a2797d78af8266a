"""prox_TV (FGP, 10 iterations) on small images: the register-resident 2-D single-workgroup kernel against the generic
LDS kernel / the 2-launches-per-iteration graph; checks each against the oracle (test infrastructure) first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from oracle import rls_oracle as orc
ctx = rls.Context(0)
rng = np.random.default_rng(0)
for dt in (np.float32, np.complex64):
    for shape in ((64, 64), (32, 32), (90, 70), (128, 64), (4096,), (17, 5)):
        n = int(np.prod(shape))
        if dt == np.complex64 and n > 4096:
            continue
        x = rng.standard_normal(n).astype(np.float32)
        if dt == np.complex64:
            x = (x + 1j * rng.standard_normal(n)).astype(np.complex64)
        lam = 0.1
        ref = orc.prox_tv_fgp(x.astype(np.complex128 if dt == np.complex64 else np.float64), lam, shape, iterationsTV=10)
        line = f"{np.dtype(dt).name:10s} {str(shape):12s}"
        for mode in (1, 0):
            ctx.tune(tv_fused_2d=mode)
            reg = rls.TVRegularization(lam, shape=shape)
            xd = rls.DeviceVector.from_host(x, ctx)
            reg.prox_(xd, lam)
            err = np.linalg.norm(xd.to_host() - ref) / np.linalg.norm(ref)
            for _ in range(5): reg.prox_(xd, lam)
            ctx.sync(); ctx.timer_start()
            for _ in range(50): reg.prox_(xd, lam)
            us = ctx.timer_stop_ms() * 1e3 / 50
            line += f" | 2d={mode}: {us:7.2f} us err {err:.1e}"
        print(line)
