"""`-m gpu`: shapes with more row blocks than the chip has CUs (BASELINE configs[2], 8192 x 4096 Float32, is one: 512 blocks of 16
rows).  Their one-pass kernels run as ONE workgroup per CU that walks several row blocks, the next block streaming in under the
products of the current one (`slab_finish_multi`, normal.hip; `rls_tune_set("slab_multi", 0)` restores one workgroup per block).
Held here: the plain normal operator, CGNR (iterates of `src/CGNR.jl:151-174`) and FISTA + L1 (`src/FISTA.jl:139-189`) against the
float64 oracle on full-size, ragged-row, ragged-column and partial-last-round shapes of all four slab layouts, and the two
launch forms against each other."""
import numpy as np
import pytest

import rls_oracle as O
from conftest import parity_check as parity

pytestmark = pytest.mark.gpu
TOL = 1e-5


def rel(a, b):
    a = np.asarray(a).astype(np.complex128)
    b = np.asarray(b).astype(np.complex128)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def hi(dt):
    return np.complex128 if np.dtype(dt).kind == "c" else np.float64


# (dtype, M, N): the slab layout is chosen from N and the dtype; row blocks = ceil(M / rows per block) must exceed 256
SHAPES = [
    (np.float32, 8192, 4096),     # configs[2]: 16-row blocks (64-byte row chunks), 512 blocks, full size: two per workgroup
    (np.float32, 4200, 3000),     # same layout, 263 blocks (the last one half dead), ragged columns: 7 workgroups walk two
    (np.float32, 12288, 2048),    # 32-row blocks, 384 blocks: a partial second round
    (np.float32, 8400, 1500),     # same layout, ragged rows and columns
    (np.complex64, 8192, 2048),   # 16-row blocks of ComplexF32, 512 blocks, full size
    (np.complex64, 4202, 1500),   # ragged rows and columns
    (np.complex64, 12304, 2048),  # 769 blocks: three rounds, the third one a single block
    (np.complex64, 2200, 3072),   # 8-row blocks (N in (2048, 4096]), 275 blocks, ragged columns
    (np.complex64, 4096, 4096),   # the same layout at full size
]


def _problem(dt, M, N, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((M, N)).astype(np.float32)
    if np.dtype(dt).kind == "c":
        A = (A + 1j * rng.standard_normal((M, N)).astype(np.float32)).astype(np.complex64)
    A = np.asfortranarray(A / np.float32(np.sqrt(M)))
    xt = np.zeros(N, dt)
    idx = rng.choice(N, 20, replace=False)
    xt[idx] = rng.standard_normal(20) + (1j * rng.standard_normal(20) if np.dtype(dt).kind == "c" else 0)
    b = (A @ xt).astype(dt)
    return A, b


@pytest.mark.parametrize("dt,M,N", SHAPES)
def test_normal_operator_walking_row_blocks(rls, ctx, dt, M, N):
    A, _ = _problem(dt, M, N, M + N)
    rng = np.random.default_rng(1)
    p = rng.standard_normal(N).astype(np.float32).astype(dt)
    if np.dtype(dt).kind == "c":
        p = (p + 1j * rng.standard_normal(N)).astype(dt)
    A64 = A.astype(hi(dt))
    want = A64.conj().T @ (A64 @ p.astype(hi(dt)))
    Ad = rls.DeviceMatrix.from_host(A)
    pd = rls.DeviceVector.from_host(p)
    out = {}
    try:
        for multi in (1, 0):
            ctx.tune(slab_multi=multi)
            op = Ad.normal_operator()
            v = rls.DeviceVector(N, dt).fill_(np.nan)
            op.mul_(v, pd)
            out[multi] = v.to_host()
            assert rel(out[multi], want) < 3e-6, multi
            v2 = rls.DeviceVector(N, dt).fill_(np.nan)
            op.mul_(v2, pd)
            assert np.array_equal(v2.to_host(), out[multi]), multi   # fixed summation order: the same bits every time
    finally:
        ctx.tune(slab_multi=1)
    assert rel(out[1], out[0]) < 1e-6


@pytest.mark.parametrize("dt,M,N", SHAPES)
def test_cgnr_on_the_pipeline_walking_row_blocks(rls, ctx, dt, M, N):
    """x, and alpha / beta of the last iteration, after 1, 5 and 12 iterations (lambda > 0: the L2 term rides in the update)"""
    A, b = _problem(dt, M, N, 3 * M + N)
    lam = 1e-3
    A64, b64 = A.astype(hi(dt)), b.astype(hi(dt))
    Ad = rls.DeviceMatrix.from_host(A)
    got = {}
    try:
        for multi in (1, 0):
            ctx.tune(slab_multi=multi)
            for iters in (1, 5, 12):
                ref = O.CGNR(A64, reg=O.L2Regularization(lam), iterations=iters, relTol=0.0, normal="matrixfree")
                O.solve(ref, b64)
                sol = rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(lam), iterations=iters, relTol=0.0)
                x = rls.solve_(sol, rls.DeviceVector.from_host(b)).to_host()
                assert sol.state.iteration == iters
                assert rel(x, ref.x) < TOL, (multi, iters, rel(x, ref.x))
                sol.state._refresh(ctx.lib)
                assert abs(sol.state.alphal - ref.alpha) < 1e-5 * abs(ref.alpha), (multi, iters)
                if iters > 1:
                    assert abs(sol.state.betal - ref.beta) < 1e-4 * abs(ref.beta), (multi, iters)
                got[multi, iters] = x
    finally:
        ctx.tune(slab_multi=1)
    assert rel(got[1, 12], got[0, 12]) < 2e-6


@pytest.mark.parametrize("dt,M,N", SHAPES[:2] + SHAPES[4:8])
def test_fista_l1_on_the_pipeline_walking_row_blocks(rls, ctx, dt, M, N):
    A, b = _problem(dt, M, N, 5 * M + N)
    A64, b64 = A.astype(hi(dt)), b.astype(hi(dt))
    v = np.ones(N, hi(dt))
    for _ in range(40):   # largest eigenvalue of A^H A to a few digits: any rho below 1 / that is a valid step
        v = A64.conj().T @ (A64 @ v)
        ev = np.linalg.norm(v)
        v /= ev
    rho = 0.9 / ev
    lam = 1e-2 * float(np.max(np.abs(A64.conj().T @ b64)))
    iters = 25
    ref = O.FISTA(A64, reg=O.L1Regularization(lam), rho=rho, iterations=iters, restart="gradient")
    O.solve(ref, b64)
    Ad = rls.DeviceMatrix.from_host(A)
    got = {}
    try:
        for multi in (1, 0):
            ctx.tune(slab_multi=multi)
            sol = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(lam), rho=rho, iterations=iters, restart="gradient")
            got[multi] = rls.solve_(sol, rls.DeviceVector.from_host(b)).to_host()
            assert sol.state.iteration == iters
            assert rel(got[multi], ref.x) < TOL, (multi, rel(got[multi], ref.x))
            assert abs(sol.state.rel_res_norm - ref.rel_res_norm) < 1e-4 * ref.rel_res_norm + 1e-7
    finally:
        ctx.tune(slab_multi=1)
    assert rel(got[1], got[0]) < 2e-6


# Gram mode (AHA explicit, the reference constructors' default operator for a dense matrix): ComplexF32 with N in (2048, 4096] has
# more 8-row blocks of AHA than the chip has CUs -- cgnr_gram_kernel / fista_gram_kernel walk them (gram_rows_walk, normal.hip)
GRAM = [(np.complex64, 600, 4096), (np.complex64, 500, 3072), (np.complex64, 400, 2400)]


@pytest.mark.parametrize("dt,M,N", GRAM)
def test_gram_mode_pipeline_walking_row_blocks(rls, ctx, dt, M, N):
    """CGNR (x, alpha, beta) and FISTA + L1 on the explicit Gram matrix against the float64 oracle and against one workgroup per block"""
    A, b = _problem(dt, M, N, 11 * M + N)
    A64, b64 = A.astype(hi(dt)), b.astype(hi(dt))
    Ad = rls.DeviceMatrix.from_host(A)
    Gd = Ad.gram()
    lam = 1e-2
    v = np.ones(N, hi(dt))
    for _ in range(30):
        v = A64.conj().T @ (A64 @ v)
        ev = np.linalg.norm(v)
        v /= ev
    lam1 = 1e-2 * float(np.max(np.abs(A64.conj().T @ b64)))
    got = {}
    try:
        for multi in (1, 0):
            ctx.tune(slab_multi=multi, resident=0)
            for iters in (1, 4, 9):
                ref = O.CGNR(A64, reg=O.L2Regularization(lam), iterations=iters, relTol=0.0, normal="gram")
                O.solve(ref, b64)
                sol = rls.createLinearSolver(rls.CGNR, Ad, AHA=Gd, reg=rls.L2Regularization(lam), iterations=iters, relTol=0.0)
                x = rls.solve_(sol, rls.DeviceVector.from_host(b)).to_host()
                assert sol.state.iteration == iters
                assert rel(x, ref.x) < TOL, (multi, iters, rel(x, ref.x))
                sol.state._refresh(ctx.lib)
                assert abs(sol.state.alphal - ref.alpha) < 1e-5 * abs(ref.alpha), (multi, iters)
                got["cgnr", multi, iters] = x
            reff = O.FISTA(A64, reg=O.L1Regularization(lam1), rho=0.9 / ev, iterations=12, normal="gram", restart="gradient")
            O.solve(reff, b64)
            solf = rls.createLinearSolver(rls.FISTA, Ad, AHA=Gd, reg=rls.L1Regularization(lam1), rho=0.9 / ev, iterations=12, restart="gradient")
            xf = rls.solve_(solf, rls.DeviceVector.from_host(b)).to_host()
            # M < N: AHA has rank M, and it was formed in Float32 -- the gate is the usual one (1e-5 against float64, or twice what the
            # oracle itself loses when it runs in the working precision: conftest.parity_check)
            def f32_run():
                r32 = O.FISTA(A, reg=O.L1Regularization(np.float32(lam1)), rho=np.float32(0.9 / ev), iterations=12, normal="gram",
                              restart="gradient")
                O.solve(r32, b)
                return r32.x
            parity(f"gram walk fista {M}x{N} multi={multi}", xf, reff.x, f32_run)
            got["fista", multi] = xf
    finally:
        ctx.tune(slab_multi=1, resident=1)
    assert rel(got["cgnr", 1, 9], got["cgnr", 0, 9]) < 2e-6
    assert rel(got["fista", 1], got["fista", 0]) < 2e-6
