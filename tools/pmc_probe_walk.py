"""For `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (one counter per pass, nothing else; tools/pmc_walk_summary.py reads the csv):
CGNR on the two-launch pipeline, eager launches, at shapes with more row blocks than CUs -- 8192 x 4096 Float32 (BASELINE configs[2]:
512 blocks, 64-byte row chunks: two workgroups split every 128-byte line) and 8256 x 4096 (516 blocks: a ragged count) -- with the
row blocks walked (slab_multi = 1) and one workgroup per block (0).  Six iterations each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rls_amd as rls
from bench import make_A
ctx = rls.Context(0)
ctx.tune(use_graph=0, resident=0)
for M, N in ((8192, 4096), (8256, 4096)):
    A = make_A(M, N, 2, np.float32); Ad = rls.DeviceMatrix.from_host(A, ctx)
    b = rls.DeviceVector.from_host((A @ np.ones(N, np.float32)).astype(np.float32), ctx)
    for multi in (1, 0):
        ctx.tune(slab_multi=multi)
        S = rls.createLinearSolver(rls.CGNR, Ad, iterations=6, relTol=0.0)
        rls.solve_(S, b)
        ctx.sync()
        del S
    del Ad
ctx.tune(slab_multi=1)
