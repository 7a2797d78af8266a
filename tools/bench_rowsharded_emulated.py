"""BASELINE config 5 (CGNR, 65536 x 8192 ComplexF32, row-partitioned, one all-reduce of A^H t per iteration) with all
its shards on ONE GPU: `--shards` contexts / streams / plans in one process, the exchange through the library's own
rls_allreduce_sum (direct transport).  Exercises the real collective schedule without an 8-GPU node and gives the
single-GPU cost of the whole problem: the per-shard kernels run one after another here, on a node they run side by side.
    python tools/bench_rowsharded_emulated.py [--shards 8] [--M 65536] [--N 8192] [--iters 16]"""
import argparse, math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rls_amd as rls
from rls_amd.multigpu import make_row_shard

ap = argparse.ArgumentParser()
ap.add_argument("--shards", type=int, default=8)
ap.add_argument("--M", type=int, default=65536)
ap.add_argument("--N", type=int, default=8192)
ap.add_argument("--iters", type=int, default=16)
a = ap.parse_args()
rng = np.random.default_rng(7)
x_true = ((rng.standard_normal(a.N) + 1j * rng.standard_normal(a.N)) / math.sqrt(2)).astype(np.complex64)
shards, parts = [], []
for r in range(a.shards):
    A, lo, hi = make_row_shard(a.M, a.N, r, a.shards)
    shards.append(A)
    parts.append((A @ x_true).astype(np.complex64))
s = rls.CommRowShardedCGNR(rls, shards, transport=2, iterations=a.iters, relTol=0.0)
del shards
s.init(parts); s.step(a.iters); s.sync()
best = 1e9
for _ in range(3):
    s.init(parts)
    s.sync()
    t0 = time.perf_counter()
    s.step(a.iters)
    s.sync()
    best = min(best, time.perf_counter() - t0)
st = s.status()
x = s.solution()
err = np.linalg.norm(x - x_true) / np.linalg.norm(x_true)
bytes_iter = 2 * a.M * a.N * 8
print(f"{a.shards} shard(s) of {a.M // a.shards} x {a.N} c64 on one GPU, direct-transport all-reduce: {1e6 * best / a.iters:.1f} us/iteration "
      f"({a.iters / best:.0f} it/s, {bytes_iter * a.iters / best / 1e12:.2f} TB/s algorithmic), iteration {st['iteration']}, "
      f"||x - x_true||/||x_true|| = {err:.2e}")
s.close()
