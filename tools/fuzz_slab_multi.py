"""Random shapes through the row-block-walking one-pass kernels (normal.hip, slab_finish_multi): M, N, dtype and the leading dimension
drawn at random among the shapes that have more row blocks than the chip has CUs; the normal-operator apply, 3 CGNR iterations and 3
FISTA + L1 iterations with slab_multi = 1 against float64 and against slab_multi = 0; ComplexF32 with N > 2048 also 3 CGNR iterations in
Gram mode (the row blocks of AHA walked).  usage: python tools/fuzz_slab_multi.py [cases=40] [seed=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import torch  # noqa
import rls_amd as rls
import rls_oracle as O   # a test tool: the oracle is the checker here, never the thing measured

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = rls.default_context(0)
rel = lambda a, b: float(np.linalg.norm(a.astype(np.complex128) - b.astype(np.complex128)) / np.linalg.norm(b.astype(np.complex128)))
worst = 0.0
for case in range(cases):
    cplx = bool(rng.integers(2))
    dt, hi = (np.complex64, np.complex128) if cplx else (np.float32, np.float64)
    V = 2 if cplx else 4                                    # elements per 16-byte row chunk
    N = int(rng.integers(1025, 4097))
    rows_per_block = (4 if N > 2048 else 8) * V             # 64-byte chunks for N in (2048, 4096], 128-byte ones below
    blocks = int(rng.integers(257, 900))
    M = (blocks * rows_per_block - int(rng.integers(0, rows_per_block // V)) * V)   # a multiple of V, last block possibly ragged
    if rng.integers(4) == 0: N = 4096 if N > 2048 else 2048                          # full-size columns now and then
    A = rng.standard_normal((M, N)).astype(np.float32)
    if cplx: A = (A + 1j * rng.standard_normal((M, N)).astype(np.float32)).astype(np.complex64)
    A = np.asfortranarray(A / np.float32(np.sqrt(M)))
    x = rng.standard_normal(N).astype(dt)
    b = (A @ x).astype(dt)
    A64 = A.astype(hi)
    want = A64.conj().T @ (A64 @ x.astype(hi))
    Ad = rls.DeviceMatrix.from_host(A, ctx)
    out = {}
    for multi in (1, 0):
        ctx.tune(slab_multi=multi)
        op = Ad.normal_operator()
        v = rls.DeviceVector(N, dt, ctx).fill_(np.nan)
        op.mul_(v, rls.DeviceVector.from_host(x, ctx))
        S = rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(1e-3), iterations=3, relTol=0.0)
        xc = rls.solve_(S, rls.DeviceVector.from_host(b, ctx)).to_host()
        F = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-3), rho=0.2, iterations=3)
        xf = rls.solve_(F, rls.DeviceVector.from_host(b, ctx)).to_host()
        xg = None
        if cplx and N > 2048 and N % 2 == 0:   # Gram mode: more 8-row blocks of AHA than CUs -- cgnr_gram_kernel walks them
            if multi == 1:
                Gd = Ad.gram()
            ctx.tune(resident=0)
            Sg = rls.createLinearSolver(rls.CGNR, Ad, AHA=Gd, reg=rls.L2Regularization(1e-3), iterations=3, relTol=0.0)
            xg = rls.solve_(Sg, rls.DeviceVector.from_host(b, ctx)).to_host()
            ctx.tune(resident=1)
        out[multi] = (v.to_host(), xc, xf, xg)
    ctx.tune(slab_multi=1)
    rc = O.CGNR(A64, reg=O.L2Regularization(1e-3), iterations=3, relTol=0.0); O.solve(rc, b.astype(hi))
    rf = O.FISTA(A64, reg=O.L1Regularization(1e-3), rho=0.2, iterations=3); O.solve(rf, b.astype(hi))
    e = (rel(out[1][0], want), rel(out[1][1], rc.x), rel(out[1][2], rf.x), rel(out[1][0], out[0][0]), rel(out[1][1], out[0][1]), rel(out[1][2], out[0][2]))
    if out[1][3] is not None:
        rg = O.CGNR(A64, reg=O.L2Regularization(1e-3), iterations=3, relTol=0.0, normal="gram"); O.solve(rg, b.astype(hi))
        eg = (rel(out[1][3], rg.x), rel(out[1][3], out[0][3]))
        assert eg[0] < 3e-5 and eg[1] < 3e-6, eg   # (AHA itself is formed in Float32: the Gram-mode bound of the parity gate)
        e = e + eg
    worst = max(worst, *e[:3])
    print(f"{case:3d} {'c32' if cplx else 'f32'} {M:6d} x {N:4d} ({blocks} blocks): apply {e[0]:.1e} cgnr {e[1]:.1e} fista {e[2]:.1e} | vs one block per workgroup {e[3]:.1e} {e[4]:.1e} {e[5]:.1e}" + (f" | Gram-mode cgnr {e[6]:.1e}, vs one block per workgroup {e[7]:.1e}" if len(e) > 6 else ""), flush=True)
    assert max(e[:3]) < 1e-5 and max(e[3:6]) < 3e-6, e
print(f"{cases} cases, worst relative error {worst:.2e}: OK")
