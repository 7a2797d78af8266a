"""Singular-value thresholding prox maps at the reference's own test sizes (test/testProxMaps.jl:194-277)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from oracle import rls_oracle as O
ctx = rls.Context(0)
rng = np.random.default_rng(1)
for name, shape, bs, K in (("LLR 2-D 32x32x80, 4x4 blocks", (32, 32), (4, 4), 80), ("LLR 3-D 32x32x32x80, 4x4x4 blocks", (32, 32, 32), (4, 4, 4), 80),
                           ("LLR 2-D 256x256x16, 8x8 blocks", (256, 256), (8, 8), 16)):
    n = int(np.prod(shape)) * K
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    reg = rls.LLRRegularization(1.0, shape=shape, blockSize=bs, randshift=False)
    xd = rls.DeviceVector.from_host(x, ctx)
    rls.prox_(reg, xd); ctx.sync()
    ts = []
    for _ in range(3):
        xd.copy_from_host(x); ctx.sync(); t0 = time.perf_counter(); rls.prox_(reg, xd); ctx.sync(); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); O.prox_llr(x.copy(), 1.0, shape, bs); tc = time.perf_counter() - t0
    print(f"{name:36s}: GPU {min(ts)*1e3:8.3f} ms   NumPy/LAPACK (1 thread loop) {tc*1e3:9.1f} ms")
m = n_ = 256
x = (rng.standard_normal(m * n_) + 1j * rng.standard_normal(m * n_)).astype(np.complex64)
for shp in ((32, 32), (96, 96)):
    xx = x[: shp[0] * shp[1]].copy(); xd = rls.DeviceVector.from_host(xx, ctx)
    reg = rls.NuclearRegularization(1.0, svtShape=shp)
    rls.prox_(reg, xd); ctx.sync(); xd.copy_from_host(xx); ctx.sync()
    t0 = time.perf_counter(); rls.prox_(reg, xd); ctx.sync(); tg = time.perf_counter() - t0
    t0 = time.perf_counter(); O.prox_nuclear(xx.copy(), 1.0, shp); tc = time.perf_counter() - t0
    print(f"Nuclear {shp}: GPU {tg*1e3:.3f} ms (one workgroup), NumPy {tc*1e3:.2f} ms")
