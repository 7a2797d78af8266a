"""Sweep the GEMV kernel variants in ONE process, interleaved rounds (guide 5.4 rule 24)."""
import os, sys, json
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
p = rls.DeviceVector.from_host(np.ones(N, dt), ctx)
t = rls.DeviceVector(M, dt, ctx)
v = rls.DeviceVector(N, dt, ctx)
s = np.dtype(dt).itemsize
by = M * N * s + (M + N) * s
reps, rounds = 100, 5

def timeit(fn):
    for _ in range(5): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(reps): fn()
    return ctx.timer_stop_ms() / reps * 1e3

res = {}
variants_n = [(g, w) for g in (8, 16, 32, 64) for w in (4, 8, 16)]
variants_t = [1, 2, 4, 8]
for r in range(rounds):
    for g, w in variants_n:
        ctx.tune(gemvn_g=g, gemvn_waves=w)
        res.setdefault(("n", g, w), []).append(timeit(lambda: Ad.gemv_(0, p, t)))
    for c in variants_t:
        ctx.tune(gemvt_cols=c)
        res.setdefault(("t", c), []).append(timeit(lambda: Ad.gemv_(2, t, v)))
for k, vals in res.items():
    med = float(np.median(vals))
    print(k, f"median {med:.2f} us  min {min(vals):.2f} us  -> {by/med/1e3:.0f} GB/s (median)")
