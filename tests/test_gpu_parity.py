"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Tolerances: 1e-5 relative l2 on iterates against the FLOAT64 oracle (BASELINE.json north_star);
where Float32 conditioning makes 1e-5 unattainable the gate is 2 x the error of the oracle's own Float32
run against its float64 run (`parity`, tests/conftest.py; both errors are logged per case and tabulated in
DESIGN.md section 3b); elementwise kernels 2e-6; reductions 1e-6.  Run on the GPU box with `pytest -m gpu`."""
import math
import os

import numpy as np
import pytest

import rls_oracle as O

pytestmark = pytest.mark.gpu

TOL_ITER = 1e-5
from conftest import parity_check as parity  # noqa: E402  (the gate: <= 1e-5 vs float64, or <= 2 x the Float32 oracle's own error)


def hi(dt):
    return np.complex128 if np.dtype(dt).kind == "c" else np.float64


def oracle_pair(make, *arrays):
    """make(*arrays) runs an oracle solve in the dtype of its inputs and returns the solution.  Returns the float64
    result and a thunk for the working-precision (Float32 = the reference's path) result."""
    x64 = make(*[a.astype(hi(a.dtype)) for a in arrays])
    return x64, (lambda: make(*arrays))


def rel(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    d = np.linalg.norm(a.astype(np.complex128) - b.astype(np.complex128))
    n = np.linalg.norm(b.astype(np.complex128))
    return d / n if n > 0 else d


def rnd(rng, n, dt):
    v = rng.standard_normal(n)
    if np.dtype(dt).kind == "c":
        v = (v + 1j * rng.standard_normal(n)) / math.sqrt(2)
    return v.astype(dt)


# ---------------------------------------------------------------------------------------------
# GEMV
# ---------------------------------------------------------------------------------------------
SHAPES = [(256, 128), (4096, 2048), (1000, 333), (37, 5), (1, 1), (64, 4097), (8192, 96), (130, 2050)]


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
@pytest.mark.parametrize("shape", SHAPES)
def test_gemv_all_ops(rls, ctx, dt, shape):
    M, N = shape
    rng = np.random.default_rng(M * 7919 + N)
    A = rnd(rng, M * N, dt).reshape(M, N)
    xN, xM = rnd(rng, N, dt), rnd(rng, M, dt)
    Ad = rls.DeviceMatrix.from_host(A)
    A64 = A.astype(np.complex128)
    # y = A x
    y = rls.DeviceVector(M, dt)
    Ad.gemv_(0, rls.DeviceVector.from_host(xN), y)
    assert rel(y.to_host(), A64 @ xN) < 2e-6
    # y = A^T x and y = A^H x
    for op, ref in ((1, A64.T @ xM), (2, A64.conj().T @ xM)):
        z = rls.DeviceVector(N, dt)
        Ad.gemv_(op, rls.DeviceVector.from_host(xM), z)
        assert rel(z.to_host(), ref) < 2e-6


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
def test_gemv_alpha_beta_and_lda(rls, ctx, dt):
    """5-arg mul! with a padded leading dimension that breaks 16-byte column alignment"""
    rng = np.random.default_rng(5)
    M, N, lda = 100, 37, 101
    A = rnd(rng, M * N, dt).reshape(M, N)
    Ad = rls.DeviceMatrix(M, N, dt, lda=lda)
    buf = np.zeros((lda, N), dtype=dt, order="F")
    buf[:M] = A
    import ctypes as C
    rls._lib.check(ctx.handle, ctx.lib.rls_memcpy_h2d(ctx.handle, Ad.ptr, buf.ctypes.data, buf.nbytes), "h2d")
    alpha, beta = (0.7 - 0.2j, -1.3 + 0.5j) if np.dtype(dt).kind == "c" else (0.7, -1.3)
    x, y0 = rnd(rng, N, dt), rnd(rng, M, dt)
    y = rls.DeviceVector.from_host(y0)
    Ad.gemv_(0, rls.DeviceVector.from_host(x), y, alpha, beta)
    assert rel(y.to_host(), alpha * (A.astype(np.complex128) @ x) + beta * y0) < 2e-6
    xm, z0 = rnd(rng, M, dt), rnd(rng, N, dt)
    z = rls.DeviceVector.from_host(z0)
    Ad.gemv_(2, rls.DeviceVector.from_host(xm), z, alpha, beta)
    assert rel(z.to_host(), alpha * (A.astype(np.complex128).conj().T @ xm) + beta * z0) < 2e-6


def test_gemv_beta_zero_ignores_nan(rls, ctx):
    """BLAS semantics: beta == 0 must not read y"""
    rng = np.random.default_rng(6)
    A = rnd(rng, 64 * 32, np.float32).reshape(64, 32)
    x = rnd(rng, 32, np.float32)
    y = rls.DeviceVector.from_host(np.full(64, np.nan, np.float32))
    rls.DeviceMatrix.from_host(A).gemv_(0, rls.DeviceVector.from_host(x), y)
    assert np.all(np.isfinite(y.to_host()))


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
def test_gemv_is_deterministic(rls, ctx, dt):
    rng = np.random.default_rng(8)
    A = rls.DeviceMatrix.from_host(rnd(rng, 2048 * 1024, dt).reshape(2048, 1024))
    x = rls.DeviceVector.from_host(rnd(rng, 1024, dt))
    t = rls.DeviceVector.from_host(rnd(rng, 2048, dt))
    y1, y2 = rls.DeviceVector(2048, dt), rls.DeviceVector(2048, dt)
    z1, z2 = rls.DeviceVector(1024, dt), rls.DeviceVector(1024, dt)
    A.gemv_(0, x, y1), A.gemv_(0, x, y2), A.gemv_(2, t, z1), A.gemv_(2, t, z2)
    assert np.array_equal(y1.to_host(), y2.to_host()) and np.array_equal(z1.to_host(), z2.to_host())


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
def test_gemv_tuning_variants_agree(rls, ctx, dt):
    """every kernel variant the heuristics can pick gives the same answer to rounding"""
    rng = np.random.default_rng(9)
    M, N = 1024, 768
    A = rnd(rng, M * N, dt).reshape(M, N)
    Ad = rls.DeviceMatrix.from_host(A)
    x, t = rnd(rng, N, dt), rnd(rng, M, dt)
    xd, td = rls.DeviceVector.from_host(x), rls.DeviceVector.from_host(t)
    refn, refc = A.astype(np.complex128) @ x, A.astype(np.complex128).conj().T @ t
    try:
        for g in (8, 16, 32, 64):
            for w in (4, 8, 16):
                ctx.tune(gemvn_g=g, gemvn_waves=w)
                y = rls.DeviceVector(M, dt)
                Ad.gemv_(0, xd, y)
                assert rel(y.to_host(), refn) < 2e-6, (g, w)
        for c in (1, 2, 4, 8):
            ctx.tune(gemvt_cols=c)
            z = rls.DeviceVector(N, dt)
            Ad.gemv_(2, td, z)
            assert rel(z.to_host(), refc) < 2e-6, c
    finally:
        ctx.tune(gemvn_g=0, gemvn_waves=0, gemvt_cols=0)


NORMAL_SHAPES = [(4096, 2048), (256, 128), (1000, 300), (8192, 4096), (64, 700), (2048, 1025), (12, 3)]


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
@pytest.mark.parametrize("shape", NORMAL_SHAPES)
def test_normal_operator_one_pass_vs_two_pass(rls, ctx, dt, shape):
    """mul!(v, AHA, p): the one-pass register-slab kernel, the two-GEMV path and float64 agree"""
    M, N = shape
    rng = np.random.default_rng(M + 3 * N)
    A = rnd(rng, M * N, dt).reshape(M, N)
    p = rnd(rng, N, dt)
    Ad = rls.DeviceMatrix.from_host(A)
    op = Ad.normal_operator()
    pd = rls.DeviceVector.from_host(p)
    A64 = A.astype(np.complex128)
    want = A64.conj().T @ (A64 @ p)
    out = {}
    try:
        for mode in (1, 0):
            ctx.tune(fused_normal=mode)
            v = rls.DeviceVector(N, dt).fill_(np.nan)
            op.mul_(v, pd)
            out[mode] = v.to_host()
            assert rel(out[mode], want) < 3e-6, mode
    finally:
        ctx.tune(fused_normal=1)
    v2 = rls.DeviceVector(N, dt)
    op.mul_(v2, pd)
    assert np.array_equal(v2.to_host(), out[1])  # deterministic


# ---------------------------------------------------------------------------------------------
# BLAS-1
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [np.float32, np.complex64])
@pytest.mark.parametrize("n", [1, 63, 2048, 100_003])
def test_blas1(rls, ctx, dt, n):
    rng = np.random.default_rng(n)
    x, y = rnd(rng, n, dt), rnd(rng, n, dt)
    xd, yd = rls.DeviceVector.from_host(x), rls.DeviceVector.from_host(y)
    x64, y64 = x.astype(np.complex128), y.astype(np.complex128)
    assert abs(xd.norm() - np.linalg.norm(x64)) <= 1e-6 * np.linalg.norm(x64)
    assert abs(xd.norm1() - np.sum(np.abs(x64))) <= 1e-6 * np.sum(np.abs(x64))
    d = xd.dot(yd)
    assert abs(d - np.vdot(x64, y64)) <= 1e-6 * np.linalg.norm(x64) * np.linalg.norm(y64)
    a = (0.3 - 1.1j) if np.dtype(dt).kind == "c" else 0.3
    b = (-0.4 + 0.2j) if np.dtype(dt).kind == "c" else -0.4
    assert rel(yd.copy().axpy_(a, xd).to_host(), y64 + a * x64) < 2e-7 * 4
    assert rel(yd.copy().axpby_(a, xd, b).to_host(), a * x64 + b * y64) < 2e-7 * 4
    assert rel(xd.copy().rmul_(a).to_host(), a * x64) < 2e-7 * 4
    z = rls.DeviceVector(n, dt).lincomb_(a, xd, b, yd)
    assert rel(z.to_host(), a * x64 + b * y64) < 2e-7 * 4
    assert np.array_equal(rls.DeviceVector(n, dt).fill_(2.5).to_host(), np.full(n, 2.5, dt))


def test_empty_vectors(rls, ctx):
    v = rls.DeviceVector(0, np.float32)
    assert v.norm() == 0.0 and v.to_host().size == 0
    v.fill_(1.0), v.rmul_(2.0)


# ---------------------------------------------------------------------------------------------
# prox maps
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [np.float32, np.complex64])
def test_prox_l1_l2_positive_real(rls, ctx, dt):
    rng = np.random.default_rng(11)
    n = 5000
    x = rnd(rng, n, dt)
    x[:7] = 0  # exact zeros
    x[7:14] *= 1e-3  # |x| < lambda
    lam = 0.35
    for Reg, ofn in ((rls.L1Regularization, O.prox_l1), (rls.L2Regularization, O.prox_l2)):
        got = rls.prox_(Reg, rls.DeviceVector.from_host(x), lam).to_host()
        want = ofn(x.copy(), lam)
        assert rel(got, want) < 2e-6, Reg.__name__
        got2 = rls.prox_(Reg(lam), rls.DeviceVector.from_host(x)).to_host()  # instance form uses lambda(reg)
        assert np.array_equal(got, got2)
    assert np.array_equal(rls.prox_(rls.PositiveRegularization, rls.DeviceVector.from_host(x)).to_host(),
                          O.prox_positive(x.copy()))
    assert np.array_equal(rls.prox_(rls.RealRegularization, rls.DeviceVector.from_host(x)).to_host(),
                          O.prox_real(x.copy()))


def test_prox_l2_closed_form(rls, ctx):
    """reference known answer: x / (1 + 2 lambda)   (test/testProxMaps.jl:13)"""
    x = np.zeros(256, np.float32)
    x[[3, 77, 200]] = [0.2, 0.9, 0.5]
    got = rls.prox_(rls.L2Regularization, rls.DeviceVector.from_host(x), 0.01).to_host()
    assert rel(got, x / (1 + 2 * 0.01)) < 1e-6


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
@pytest.mark.parametrize("n,slices", [(2048, 8), (1000, 7), (64, 64), (10, 3)])
def test_prox_l21(rls, ctx, dt, n, slices):
    rng = np.random.default_rng(n + slices)
    x = rnd(rng, n, dt)
    slen = n // slices
    x[0::slen] = 0  # one all-zero group: (0 - lam)/0 = -Inf -> clipped to 0
    lam = 0.8
    got = rls.prox_(rls.L21Regularization, rls.DeviceVector.from_host(x), lam, slices=slices).to_host()
    want = O.prox_l21(x.copy(), lam, slices)
    assert rel(got, want) < 2e-6
    nrm = rls.norm(rls.L21Regularization(lam, slices=slices), rls.DeviceVector.from_host(x))
    assert abs(nrm - O.norm_l21(x, lam, slices)) <= 2e-6 * abs(O.norm_l21(x, lam, slices))


def test_prox_l21_zero_group_lambda_zero_is_nan(rls, ctx):
    """0/0 = NaN propagates through max as in Julia (hard part 5)"""
    x = np.ones(8, np.float32)
    x[0::4] = 0
    got = rls.prox_(rls.L21Regularization, rls.DeviceVector.from_host(x), 0.0, slices=2).to_host()
    assert np.isnan(got[0]) and np.isnan(got[4]) and np.all(got[[1, 2, 3, 5, 6, 7]] == 1.0)


TV_CASES = [((16, 16), None), ((64, 64), None), ((8, 8), (1,)), ((8, 8), (2,)), ((5, 4, 3), None), ((300,), None),
            ((7, 9), (2, 1)), ((256, 256), None),
            ((96, 80), None), ((90, 91), None), ((128, 64), (1,))]   # 4097..8192 pixels: 8 per thread in the register-resident kernel (Float32)


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
@pytest.mark.parametrize("shape,dims", TV_CASES)
def test_tv_gradient_and_prox(rls, ctx, dt, shape, dims):
    rng = np.random.default_rng(int(np.prod(shape)))
    n = int(np.prod(shape))
    x = rnd(rng, n, dt)
    d0 = O._as_dims(shape, dims)
    G = rls.GradientOp(shape, dims)
    assert G.n_out == O.grad_len(shape, d0)
    xd = rls.DeviceVector.from_host(x)
    g = G.mul(xd)
    assert rel(g.to_host(), O.grad_apply(x, shape, d0)) < 1e-6
    gv = rnd(rng, G.n_out, dt)
    back = rls.DeviceVector(n, dt)
    G.mul_adj_(back, rls.DeviceVector.from_host(gv))
    assert rel(back.to_host(), O.grad_apply_t(gv, shape, d0)) < 1e-6
    lam = 0.3
    got = rls.prox_(rls.TVRegularization, rls.DeviceVector.from_host(x), lam, shape=shape, dims=dims).to_host()
    want = O.prox_tv_fgp(x.copy(), lam, shape, dims, 10)
    assert rel(got, want) < 1e-5


def test_tv_pieces_match_fused(rls, ctx):
    """the separately exported FGP pieces (the methods the Julia ext overloads one by one) compose
    to the fused kernel's result"""
    rng = np.random.default_rng(3)
    shape, lam = (12, 10), 0.25
    x = rnd(rng, 120, np.float32)
    fused = rls.prox_(rls.TVRegularization, rls.DeviceVector.from_host(x), lam, shape=shape).to_host()
    import ctypes as C
    G = rls.GradientOp(shape)
    lib, h = ctx.lib, ctx.handle
    xd = rls.DeviceVector.from_host(x)
    xt = rls.DeviceVector(120, np.float32)
    pq, rs, pqo = (rls.DeviceVector(G.n_out, np.float32).fill_(0) for _ in range(3))
    t = np.float32(1)
    for _ in range(10):
        pqt, pqo, pq = pqo, pq, rs
        xt.copy_from(xd)
        G.mul_adj_(xt, rs, -lam, 1.0)
        G.mul_(pq, xt, 1.0 / (8 * lam), 1.0)
        rls._lib.check(h, lib.rls_tv_restrict(h, 0, G.n_out, pq.ptr), "restrict")
        told = t
        t = (np.float32(1) + np.sqrt(np.float32(1) + np.float32(4) * told * told)) / np.float32(2)
        t2 = (told - 1) / t
        rs = pqt
        rls._lib.check(h, lib.rls_tv_lincomb(h, 0, G.n_out, rs.ptr, float(1 + t2), pq.ptr, float(t2), pqo.ptr), "lincomb")
    G.mul_adj_(xd, pq, -lam, 1.0)
    assert rel(xd.to_host(), fused) < 2e-6


# ---------------------------------------------------------------------------------------------
# solvers
# ---------------------------------------------------------------------------------------------
def _cgnr_pair(rls, M, N, dt, seed, lam, iters, mode="matrixfree"):
    A, xt, b = O.make_problem(M, N, dt, seed)
    dt64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    reg = O.L2Regularization(lam)
    ref = O.CGNR(A.astype(dt64), reg=reg, iterations=iters, relTol=0.0, normal=mode)
    Ad = rls.DeviceMatrix.from_host(A)
    kw = dict(AHA=Ad.gram()) if mode == "gram" else {}
    sol = rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(lam), iterations=iters, relTol=0.0, **kw)
    return ref, sol, b, dt64


@pytest.mark.parametrize("dt,M,N,lam", [(np.float32, 256, 128, 1e-2), (np.complex64, 64, 32, 0.0),
                                       (np.complex64, 4096, 2048, 0.0), (np.float32, 1000, 300, 0.5),
                                       # ComplexF32 N in (2048, 4096]: 16 rows x 32 columns per workgroup -- the slab pipeline's
                                       # hinted kernel only (an unhinted launch runs it on a guess, normal.hip launch_pipe_a)
                                       (np.complex64, 3200, 3072, 1e-3)])
def test_cgnr_iterates_match_oracle(rls, ctx, dt, M, N, lam):
    """per-iteration x, r, p and alpha, beta against the float64 oracle (SURVEY 8d parity gate)"""
    iters = 32 if M >= 1000 else 10
    ref, sol, b, dt64 = _cgnr_pair(rls, M, N, dt, 1, lam, iters)
    ref.init(b.astype(dt64))
    rls.init_(sol, rls.DeviceVector.from_host(b))
    checks = {1, 5, 10, 32}
    for it in range(1, iters + 1):
        assert ref.iterate() is not None
        assert rls.iterate(sol) is not None
        if it in checks:
            st = sol.state
            assert rel(st.x.to_host(), ref.x) < TOL_ITER, it
            # r and p shrink geometrically: compare them on the scale of the initial residual A^H b
            r0 = np.linalg.norm(ref.A.mul_adj(b.astype(dt64)))
            assert np.linalg.norm(st.pl.to_host() - ref.p) < TOL_ITER * r0, it
            assert np.linalg.norm(st.x0.to_host() - ref.r) < TOL_ITER * r0, it
            st._refresh(ctx.lib)
            assert abs(st.alphal - ref.alpha) < 1e-5 * abs(ref.alpha)
            assert abs(st.betal - ref.beta) < 1e-4 * abs(ref.beta)
    assert rls.iterate(sol) is None and ref.iterate() is None
    assert sol.state.iteration == iters
    if M >= 1000:  # the same solve as ONE step call (graph chunks: their first launch carries no buffer hint), both hint modes
        x_steps = sol.state.x.to_host()
        for mode in (0, 1, 2):  # host bookkeeping / always "unknown" / always wrong (the kernel re-loads the right pair)
            ctx.tune(pipe_hint_mode=mode, resident=0)
            try:
                x_once = rls.solve_(sol, rls.DeviceVector.from_host(b)).to_host()
            finally:
                ctx.tune(pipe_hint_mode=0, resident=1)
            assert rel(x_once, x_steps) < 1e-6, mode


def test_cgnr_gram_mode_and_float32_oracle(rls, ctx):
    ref, sol, b, dt64 = _cgnr_pair(rls, 512, 256, np.complex64, 3, 1e-3, 10, mode="gram")
    O.solve(ref, b.astype(dt64))
    x = rls.solve_(sol, rls.DeviceVector.from_host(b)).to_host()
    assert rel(x, ref.x) < TOL_ITER
    A, _, _ = O.make_problem(512, 256, np.complex64, 3)
    ref32 = O.CGNR(A, reg=O.L2Regularization(1e-3), iterations=10, relTol=0.0)
    O.solve(ref32, b)
    assert rel(x, ref32.x) < TOL_ITER


@pytest.mark.parametrize("pipe", [2, 1, 0])
@pytest.mark.parametrize("dt,M,N,lam,iters", [(np.complex64, 4096, 2048, 0.0, 32), (np.float32, 600, 256, 1e-2, 12),
                                              (np.complex64, 90, 46, 0.1, 9), (np.float32, 5000, 4096, 0.0, 6),
                                              (np.complex64, 1500, 1024, 1e-3, 12), (np.float32, 3000, 2048, 0.0, 10),
                                              # ComplexF32 N in (2048, 4096]: cgnr_gram_kernel<c32, 4, 32, 8> (32 rows of AHA per
                                              # lane beside 8 owned elements of four vectors: the one Gram kernel that spills)
                                              (np.complex64, 3200, 3072, 1e-3, 8)])
def test_cgnr_gram_pipeline_iterates(rls, ctx, dt, M, N, lam, iters, pipe):
    """Gram mode (AHA = A' * A explicit, the constructor default for a dense Matrix, src/CGNR.jl:49).  pipe 2: the
    resident kernel where AHA fits the register files (N <= 2048 CF32, N <= 4096 F32, ragged N included: the whole step
    call in one launch, one in-kernel grid exchange per iteration), the one-launch-per-iteration pipeline elsewhere; pipe 1:
    that pipeline everywhere; pipe 0: the unfused path.  Step-by-step iterates, a single n-step call and relTol
    retirement all agree with the float64 oracle in Gram mode."""
    ctx.tune(gram_pipeline=1 if pipe else 0, resident=1 if pipe == 2 else 0)
    try:
        ref, sol, b, dt64 = _cgnr_pair(rls, M, N, dt, 7, lam, iters, mode="gram")
        if pipe:
            import ctypes
            rls.init_(sol, rls.DeviceVector.from_host(b))
            pth = ctypes.c_int32(-1)
            ctx.lib.rls_cgnr_path(sol.state._plan, ctypes.byref(pth))
            fits = N <= (2048 if np.dtype(dt).kind == "c" else 4096)  # one workgroup (8 / 16 rows of AHA) per CU
            assert pth.value == (5 if pipe == 2 and fits else 2), pth.value
        ref32 = O.CGNR(ref.A.A.astype(dt), reg=O.L2Regularization(lam), iterations=iters, relTol=0.0, normal="gram")
        bd = rls.DeviceVector.from_host(b)
        ref.init(b.astype(dt64))
        ref32.init(b)
        rls.init_(sol, bd)
        r0 = np.linalg.norm(ref.A.mul_adj(b.astype(dt64)))
        tag = f"cgnr_gram_{M}x{N}_{np.dtype(dt).name}_pipe{pipe}"
        for it in range(1, iters + 1):
            assert ref.iterate() is not None and rls.iterate(sol) is not None and ref32.iterate() is not None
            if it in (1, 2, 5, iters):
                st = sol.state
                parity(f"{tag}_x_it{it}", st.x.to_host(), ref.x, ref32.x)
                parity(f"{tag}_p_it{it}", st.pl.to_host(), ref.p, ref32.p, scale=r0)
                parity(f"{tag}_r_it{it}", st.x0.to_host(), ref.r, ref32.r, scale=r0)
                parity(f"{tag}_v_it{it}", st.vl.to_host(), ref.v, ref32.v)
        assert rls.iterate(sol) is None and sol.state.iteration == iters
        x_steps = sol.state.x.to_host()
        x_once = rls.solve_(sol, bd).to_host()  # all iterations in one call (graph chunks + finish)
        assert rel(x_once, x_steps) < 1e-6
        # early retirement on relTol: same iteration count as the oracle (+-1 at the threshold)
        tol = 1e-3
        ref2 = O.CGNR(ref.A.A, reg=O.L2Regularization(lam), iterations=iters, relTol=tol, normal="gram")
        O.solve(ref2, b.astype(dt64))
        kw = dict(AHA=rls.DeviceMatrix.from_host(np.asfortranarray(ref.A.A.astype(dt))).gram())
        sol2 = rls.createLinearSolver(rls.CGNR, rls.DeviceMatrix.from_host(np.asfortranarray(ref.A.A.astype(dt))),
                                      reg=rls.L2Regularization(lam), iterations=iters, relTol=tol, **kw)
        x2 = rls.solve_(sol2, bd).to_host()
        assert abs(sol2.state.iteration - ref2.iteration) <= 1
        if sol2.state.iteration == ref2.iteration:
            ref2_32 = O.CGNR(ref.A.A.astype(dt), reg=O.L2Regularization(lam), iterations=ref2.iteration, relTol=0.0, normal="gram")
            parity(f"{tag}_reltol", x2, ref2.x, lambda: O.solve(ref2_32, b))
    finally:
        ctx.tune(gram_pipeline=1, resident=1)


def test_cgnr_callbacks_cadence_and_lstsq(rls, ctx):
    """reference known answers: callbacks fire iterations+1 times and solutions[end] == x_approx
    (test/testCallbacks.jl:6-16); CGNR converges to the least-squares solution."""
    A, xt, b = O.make_problem(32, 32, np.float32, 4)
    Ad = rls.DeviceMatrix.from_host(A)
    sol = rls.createLinearSolver(rls.CGNR, Ad, iterations=10, relTol=0.0)
    cb = rls.StoreSolutionCallback()
    count = []
    x = rls.solve_(sol, rls.DeviceVector.from_host(b), callbacks=[cb, lambda s, i: count.append(i)])
    assert len(cb.solutions) == 11 and count == list(range(11))
    assert np.array_equal(cb.solutions[-1], x.to_host())
    A2, x2, b2 = O.make_problem(256, 128, np.float32, 5)
    sol2 = rls.createLinearSolver(rls.CGNR, rls.DeviceMatrix.from_host(A2), iterations=128, relTol=0.0)
    xs = rls.solve_(sol2, rls.DeviceVector.from_host(b2)).to_host()
    xl = np.linalg.lstsq(A2.astype(np.float64), b2.astype(np.float64), rcond=None)[0]
    assert rel(xs, xl) < 1e-4


@pytest.mark.parametrize("solver,dt,M,N", [("cgnr", np.complex64, 1024, 512), ("cgnr", np.float32, 256, 128), ("cgnr", np.complex64, 32, 16),
                                           ("fista", np.complex64, 1024, 512), ("fista", np.float32, 300, 120)])
def test_step_status_published_by_the_last_kernel(rls, ctx, solver, dt, M, N):
    """One iterate per call (the reference's solve! loop, src/RegularizedLeastSquares.jl:161-176): rls_*_step_status has the
    call's last kernel store the scalars into pinned host memory.  The stream of statuses -- through the stopping test and for
    calls after it -- is the one step + get_status gives with the mailbox switched off (hipMemcpyAsync), field for field."""
    A, xt, b = O.make_problem(M, N, dt, 9)
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    A64 = A.astype(np.complex128 if np.dtype(dt).kind == "c" else np.float64)
    rho = float(0.9 / np.linalg.norm(A64, 2) ** 2)
    streams = []
    try:
        for mb in (2, 1, 0):
            ctx.tune(status_mailbox=mb)
            if solver == "cgnr":
                S = rls.createLinearSolver(rls.CGNR, Ad, iterations=24, relTol=1e-4)
            else:
                S = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=rho, iterations=24, relTol=1e-3)
            rls.init_(S, bd)
            rec = []
            for _ in range(28):  # past `done`: the finish kernel's nothing-to-apply path publishes too
                st = S.state._step_status(ctx.lib, 1)
                rec.append(tuple(getattr(st, f) for f, _ in st._fields_))
            assert rec[-1][1] == 1 and rec[-1][0] <= 24  # done, and the count stopped
            streams.append((rec, rls.solversolution(S).to_host().copy()))
    finally:
        ctx.tune(status_mailbox=2)
    assert streams[0][0] == streams[1][0] == streams[2][0]
    assert np.array_equal(streams[0][1], streams[1][1]) and np.array_equal(streams[0][1], streams[2][1])
    assert 1 < streams[0][0][-1][0]


def test_cgnr_reltol_stops_early_and_constraints(rls, ctx):
    A, xt, b = O.make_problem(128, 64, np.complex64, 6)
    ref = O.CGNR(A, reg=[O.PositiveRegularization()], iterations=64, relTol=1e-3)
    O.solve(ref, b)
    sol = rls.createLinearSolver(rls.CGNR, rls.DeviceMatrix.from_host(A), reg=[rls.PositiveRegularization()],
                                 iterations=64, relTol=1e-3)
    x = rls.solve_(sol, rls.DeviceVector.from_host(b)).to_host()
    assert sol.state.iteration == ref.iteration < 64
    assert rel(x, ref.x) < 1e-5 and np.all(x.imag == 0) and np.all(x.real >= 0)


def test_create_linear_solver_filters_unknown_kwargs(rls, ctx):
    A, _, _ = O.make_problem(16, 8, np.float32, 1)
    with pytest.warns(UserWarning, match="filtered out"):
        rls.createLinearSolver(rls.CGNR, rls.DeviceMatrix.from_host(A), iterations=3, shape=(2, 4))
    with pytest.raises(ValueError, match="additional regularization"):
        rls.createLinearSolver(rls.CGNR, rls.DeviceMatrix.from_host(A), reg=rls.L1Regularization(0.1))


@pytest.mark.parametrize("restart", ["none", "gradient"])
@pytest.mark.parametrize("dt,M,N", [(np.complex64, 64, 32), (np.complex64, 4096, 2048), (np.float32, 300, 120)])
def test_fista_l1_matches_oracle(rls, ctx, dt, M, N, restart):
    A, xt, b = O.make_problem(M, N, dt, 2)
    dt64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    A64, b64 = A.astype(dt64), b.astype(dt64)
    smax = np.linalg.norm(A64, 2)
    rho = 0.95 / smax ** 2
    lam = 1e-2 * np.max(np.abs(A64.conj().T @ b64))
    iters = 50
    ref = O.FISTA(A64, reg=O.L1Regularization(lam), rho=rho, iterations=iters, restart=restart)
    O.solve(ref, b64)
    sol = rls.createLinearSolver(rls.FISTA, rls.DeviceMatrix.from_host(A), reg=rls.L1Regularization(lam), rho=rho,
                                 iterations=iters, restart=restart)
    x = rls.solve_(sol, rls.DeviceVector.from_host(b)).to_host()
    assert sol.state.iteration == iters
    assert rel(x, ref.x) < TOL_ITER
    assert abs(sol.state.rel_res_norm - ref.rel_res_norm) < 1e-4 * ref.rel_res_norm + 1e-7


def test_fista_step_by_step_equals_batched_and_proj(rls, ctx):
    A, xt, b = O.make_problem(200, 80, np.complex64, 9)
    smax = np.linalg.norm(A.astype(np.complex128), 2)
    kw = dict(reg=[rls.L1Regularization(0.05), rls.PositiveRegularization()], rho=0.9 / smax ** 2, iterations=20)
    Ad = rls.DeviceMatrix.from_host(A)
    s1 = rls.createLinearSolver(rls.FISTA, Ad, **kw)
    x1 = rls.solve_(s1, rls.DeviceVector.from_host(b)).to_host()
    s2 = rls.createLinearSolver(rls.FISTA, Ad, **kw)
    seen = []
    x2 = rls.solve_(s2, rls.DeviceVector.from_host(b), callbacks=lambda s, i: seen.append(i)).to_host()
    assert seen == list(range(21)) and np.array_equal(x1, x2)
    ref = O.FISTA(A, reg=[O.L1Regularization(0.05), O.PositiveRegularization()], rho=0.9 / smax ** 2, iterations=20)
    O.solve(ref, b)
    assert rel(x1, ref.x) < TOL_ITER and np.all(x1.imag == 0) and np.all(x1.real >= 0)


@pytest.mark.parametrize("regname", ["l21", "l2", "tv"])
def test_fista_other_regs(rls, ctx, regname):
    A, xt, b = O.make_problem(160, 64, np.float32, 12)
    smax = np.linalg.norm(A.astype(np.float64), 2)
    rho = 0.9 / smax ** 2
    if regname == "l21":
        r_o, r_d = O.L21Regularization(2.0, slices=4), rls.L21Regularization(2.0, slices=4)
    elif regname == "l2":
        r_o, r_d = O.L2Regularization(0.5), rls.L2Regularization(0.5)
    else:
        r_o, r_d = O.TVRegularization(2.0, shape=(8, 8)), rls.TVRegularization(2.0, shape=(8, 8))
    x64, x32 = oracle_pair(lambda A_, b_: O.solve(O.FISTA(A_, reg=r_o, rho=rho, iterations=15), b_), A, b)
    sol = rls.createLinearSolver(rls.FISTA, rls.DeviceMatrix.from_host(A), reg=r_d, rho=rho, iterations=15)
    x = rls.solve_(sol, rls.DeviceVector.from_host(b)).to_host()
    parity(f"fista_{regname}_160x64_f32", x, x64, x32)


@pytest.mark.parametrize("dt,M,N,shape", [(np.float32, 128, 64, (8, 8)), (np.float32, 8192, 4096, (64, 64)),
                                         (np.complex64, 96, 36, (6, 6))])
def test_admm_tv_matches_oracle(rls, ctx, dt, M, N, shape):
    A, xt, b = O.make_problem(M, N, dt, 3)
    kw = dict(rho=0.1, iterations=10, iterationsCG=10, tolInner=1e-5)
    ref = O.ADMM(A, reg=O.TVRegularization(1e-2, shape=shape), **kw)  # Float32 oracle = the reference's path: counts
    O.solve(ref, b)
    ref64 = O.ADMM(A.astype(hi(dt)), reg=O.TVRegularization(1e-2, shape=shape), **kw)  # float64 oracle: the truth for x
    O.solve(ref64, b.astype(hi(dt)))
    sol = rls.createLinearSolver(rls.ADMM, rls.DeviceMatrix.from_host(A), reg=rls.TVRegularization(1e-2, shape=shape), **kw)
    x = rls.solve_(sol, rls.DeviceVector.from_host(b)).to_host()
    assert sol.state.iteration == ref.iteration
    assert sol.state.cg_iterations == ref.cg_iters
    parity(f"admm_tv_{M}x{N}_{np.dtype(dt).name}" + (" (BASELINE config 3)" if M == 8192 else ""), x, ref64.x, ref.x)
    assert np.allclose(sol.state.rk, ref.rk, rtol=2e-3, atol=1e-6) and np.allclose(sol.state.sk, ref.sk, rtol=2e-3, atol=1e-6)


@pytest.mark.parametrize("dt,M,N,kind", [(np.float32, 128, 64, "tv"), (np.complex64, 96, 36, "tv"), (np.float32, 120, 48, "l1"),
                                         (np.complex64, 90, 40, "l1pos"), (np.float32, 64, 32, "l2"),
                                         (np.float32, 100, 60, "tv1d"), (np.float32, 100, 63, "tv3d")])
def test_admm_device_plan_equals_per_call_path(rls, ctx, dt, M, N, kind):
    """rls_admm_step (whole outer iterations on the device, `done` decided there) against the per-call sequence of the
    same solver and against the oracle: same iteration count, inner cg! counts, residuals, solution -- step by step
    with callbacks and in one go"""
    A, xt, b = O.make_problem(M, N, dt, 21)
    def regs(R):
        if kind == "tv":
            sh = {64: (8, 8), 36: (6, 6)}[N]
            return R.TVRegularization(2e-2, shape=sh)
        if kind == "tv1d":
            return R.TVRegularization(2e-2, shape=(N,))
        if kind == "tv3d":
            return R.TVRegularization(2e-2, shape=(3, 7, 3))
        if kind == "l1":
            return R.L1Regularization(0.05)
        if kind == "l1pos":
            return [R.L1Regularization(0.05), R.PositiveRegularization()]
        return R.L2Regularization(0.3)
    kw = dict(rho=0.3, iterations=12, iterationsCG=6, tolInner=1e-4)
    ref = O.ADMM(A, reg=regs(O), **kw)
    O.solve(ref, b)
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    sol = rls.createLinearSolver(rls.ADMM, Ad, reg=regs(rls), **kw)
    x = rls.solve_(sol, bd).to_host()
    assert sol.state._plan_ok
    old = rls.createLinearSolver(rls.ADMM, Ad, reg=regs(rls), **kw)
    old.use_device_plan = False
    x_old = rls.solve_(old, bd).to_host()
    assert not old.state._plan_ok
    assert sol.state.iteration == old.state.iteration == ref.iteration
    assert sol.state.cg_iterations == old.state.cg_iterations == ref.cg_iters
    ref64 = O.ADMM(A.astype(hi(dt)), reg=regs(O), **kw)
    parity(f"admm_plan_{kind}_{M}x{N}_{np.dtype(dt).name}", x, O.solve(ref64, b.astype(hi(dt))), ref.x)
    assert rel(x, x_old) < 2e-6
    assert np.allclose(sol.state.rk, ref.rk, rtol=2e-3, atol=1e-6) and np.allclose(sol.state.sk, ref.sk, rtol=2e-3, atol=1e-6)
    assert np.allclose(sol.state.eps_pri, old.state.eps_pri, rtol=1e-5) and np.allclose(sol.state.eps_dua, old.state.eps_dua, rtol=1e-5)
    assert rel(sol.state.z[0].to_host(), old.state.z[0].to_host()) < 2e-6
    # u = sum of (x - z) over the iterations: a difference of nearly equal vectors, so the 2e-6 agreement of x and z
    # between the two device paths is ~10x larger relative to ||u|| (two device paths against each other, not the gate)
    assert rel(sol.state.u[0].to_host(), old.state.u[0].to_host()) < 2e-5
    # with callbacks: one plan iteration per host iteration, the same bits as the bulk run
    seen = []
    x_cb = rls.solve_(sol, bd, callbacks=lambda s, it: seen.append((it, s.state.iteration))).to_host()
    assert seen == [(k, k) for k in range(ref.iteration + 1)]
    assert np.array_equal(x_cb, x)


def test_admm_device_plan_stops_when_converged(rls, ctx):
    """`converged` (src/ADMM.jl:324-330) evaluated on the device: with loose tolerances the plan stops itself at the
    iteration the oracle stops at, the remaining enqueued launches are no-ops, and a second solve re-arms it"""
    A, xt, b = O.make_problem(150, 50, np.float32, 5)
    kw = dict(rho=1.0, iterations=60, iterationsCG=10, tolInner=1e-6, absTol=1e-3, relTol=2e-2)
    ref = O.ADMM(A, reg=O.L1Regularization(1e-3), **kw)
    O.solve(ref, b)
    assert 1 < ref.iteration < 60
    sol = rls.createLinearSolver(rls.ADMM, rls.DeviceMatrix.from_host(A), reg=rls.L1Regularization(1e-3), **kw)
    bd = rls.DeviceVector.from_host(b)
    # float64 truth with the stop iteration of the Float32 path (the stop test sits on a Float32 threshold)
    ref64 = O.ADMM(A.astype(np.float64), reg=O.L1Regularization(1e-3), **dict(kw, iterations=ref.iteration, absTol=0.0, relTol=0.0))
    x64 = O.solve(ref64, b.astype(np.float64))
    for _ in range(2):
        x = rls.solve_(sol, bd).to_host()
        assert sol.state._plan_ok and sol.state.iteration == ref.iteration
        assert sol.converged(sol.state)
        parity("admm_l1_converged_150x50_f32", x, x64, ref.x)
        assert len(sol.state.cg_iterations) == ref.iteration


@pytest.mark.parametrize("pipe", [2, 1, 0])
@pytest.mark.parametrize("dt,M,N,restart", [(np.complex64, 4096, 2048, "none"), (np.float32, 300, 120, "gradient"),
                                            (np.complex64, 70, 34, "none"), (np.float32, 2500, 2048, "gradient"),
                                            (np.complex64, 1100, 1024, "gradient")])
def test_fista_gram_mode_matches_oracle(rls, ctx, dt, M, N, restart, pipe):
    """FISTA(A; AHA = A'*A) (src/FISTA.jl:58, explicit Gram = the constructor default for a dense Matrix).  pipe 2: the
    resident kernel where AHA fits the register files (fista_gram_resident_kernel: one launch per step call), pipe 1: one
    launch per iteration, pipe 0: the unfused path; iterates step by step and in one call against the oracle"""
    ctx.tune(gram_pipeline=1 if pipe else 0, resident=1 if pipe == 2 else 0)
    try:
        A, xt, b = O.make_problem(M, N, dt, 9)
        dt64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
        A64, b64 = A.astype(dt64), b.astype(dt64)
        rho = 0.95 / np.linalg.norm(A64, 2) ** 2
        lam = 1e-2 * np.max(np.abs(A64.conj().T @ b64))
        its = 25
        oreg = [O.L1Regularization(lam), O.PositiveRegularization()] if restart == "gradient" else O.L1Regularization(lam)
        ref = O.FISTA(A64, reg=oreg, rho=rho, iterations=its, restart=restart, normal="gram")
        ref32 = O.FISTA(A, reg=oreg, rho=rho, iterations=its, restart=restart, normal="gram")
        tag = f"fista_gram_{M}x{N}_{np.dtype(dt).name}_pipe{pipe}"
        Ad = rls.DeviceMatrix.from_host(A)
        reg = [rls.L1Regularization(lam), rls.PositiveRegularization()] if restart == "gradient" else rls.L1Regularization(lam)
        sol = rls.createLinearSolver(rls.FISTA, Ad, AHA=Ad.gram(), reg=reg, rho=rho, iterations=its, restart=restart)
        bd = rls.DeviceVector.from_host(b)
        ref.init(b64)
        ref32.init(b)
        rls.init_(sol, bd)
        for it in range(1, its + 1):
            assert ref.iterate() is not None and rls.iterate(sol) is not None and ref32.iterate() is not None
            if it in (1, 2, 7, its):
                parity(f"{tag}_it{it}", sol.state.x.to_host(), ref.x, ref32.x)
        assert rls.iterate(sol) is None
        x_once = rls.solve_(sol, bd).to_host()
        parity(f"{tag}_once", x_once, ref.x, ref32.x)
        assert sol.state.iteration == its
        if pipe:
            import ctypes
            pth = ctypes.c_int32(-1)
            ctx.lib.rls_fista_path(sol.state._plan, ctypes.byref(pth))
            fits = N <= (2048 if np.dtype(dt).kind == "c" else 4096)  # one workgroup (8 / 16 rows of AHA) per CU
            assert pth.value == (5 if pipe == 2 and fits else 2), pth.value
    finally:
        ctx.tune(gram_pipeline=1, resident=1)


@pytest.mark.parametrize("dt,M,N,shape", [(np.float32, 8192, 4096, (64, 64)), (np.complex64, 96, 36, (6, 6))])
def test_admm_gram_mode_matches_oracle(rls, ctx, dt, M, N, shape):
    """ADMM(A; AHA = A'*A) -- the constructor default of the reference for a dense Matrix (src/ADMM.jl:82): cg! on the
    explicit Gram matrix runs through the one-launch-per-iteration Gram pipeline; config 3 shape included"""
    A, xt, b = O.make_problem(M, N, dt, 3)
    kw = dict(rho=0.1, iterations=6, iterationsCG=10, tolInner=1e-5)
    ref = O.ADMM(A, reg=O.TVRegularization(1e-2, shape=shape), normal="gram", **kw)
    O.solve(ref, b)
    ref64 = O.ADMM(A.astype(hi(dt)), reg=O.TVRegularization(1e-2, shape=shape), normal="gram", **kw)
    O.solve(ref64, b.astype(hi(dt)))
    Ad = rls.DeviceMatrix.from_host(A)
    sol = rls.createLinearSolver(rls.ADMM, Ad, AHA=Ad.gram(), reg=rls.TVRegularization(1e-2, shape=shape), **kw)
    x = rls.solve_(sol, rls.DeviceVector.from_host(b)).to_host()
    assert sol.state.iteration == ref.iteration and sol.state.cg_iterations == ref.cg_iters
    tag = f"admm_tv_gram_{M}x{N}_{np.dtype(dt).name}"
    parity(tag, x, ref64.x, ref.x)
    ctx.tune(gram_pipeline=0)
    try:
        sol2 = rls.createLinearSolver(rls.ADMM, Ad, AHA=Ad.gram(), reg=rls.TVRegularization(1e-2, shape=shape), **kw)
        parity(tag + "_nopipe", rls.solve_(sol2, rls.DeviceVector.from_host(b)).to_host(), ref64.x, ref.x)
    finally:
        ctx.tune(gram_pipeline=1)


@pytest.mark.parametrize("vary", ["none", "balance", "PnP"])
def test_admm_l1_vary_rho_and_gradient_trafo(rls, ctx, vary):
    A, xt, b = O.make_problem(100, 48, np.float32, 21)
    kw = dict(rho=0.5, iterations=8, vary_rho=vary)
    ref = O.ADMM(A, reg=O.L1Regularization(0.05), **kw)
    O.solve(ref, b)
    ref64 = O.ADMM(A.astype(np.float64), reg=O.L1Regularization(0.05), **kw)
    O.solve(ref64, b.astype(np.float64))
    sol = rls.createLinearSolver(rls.ADMM, rls.DeviceMatrix.from_host(A), reg=rls.L1Regularization(0.05), **kw)
    x = rls.solve_(sol, rls.DeviceVector.from_host(b)).to_host()
    parity(f"admm_l1_vary_{vary}_100x48_f32", x, ref64.x, ref.x)
    assert np.allclose(sol.state.rho, ref.rho)
    # TV as L1-of-gradient (src/ADMM.jl:74): regTrafo = GradientOp
    x64, x32 = oracle_pair(lambda A_, b_: O.solve(O.ADMM(A_, reg=O.L1Regularization(0.05), regTrafo=O.GradientTrafo((8, 6)),
                                                        rho=0.5, iterations=5), b_), A, b)
    sol2 = rls.createLinearSolver(rls.ADMM, rls.DeviceMatrix.from_host(A), reg=rls.L1Regularization(0.05),
                                  regTrafo=rls.GradientOp((8, 6)), rho=0.5, iterations=5)
    x2 = rls.solve_(sol2, rls.DeviceVector.from_host(b)).to_host()
    parity("admm_l1_gradient_trafo_100x48_f32", x2, x64, x32)


@pytest.mark.parametrize("scheduler", ["SequentialState", "MultiThreadingState"])
def test_matrix_rhs_equals_column_solves(rls, ctx, scheduler):
    """reference property: matrix solve == column-by-column solves, and a vector solve still works
    afterwards (test/testMultiThreading.jl:10-18)"""
    A, X, B = O.make_problem(96, 48, np.complex64, 7, n_rhs=4)
    Ad = rls.DeviceMatrix.from_host(A)
    S = rls.createLinearSolver(rls.CGNR, Ad, iterations=48)
    cols = [rls.solve_(S, rls.DeviceVector.from_host(B[:, j])).to_host() for j in range(4)]
    xs = rls.solve_(S, rls.DeviceMatrix.from_host(B), scheduler=getattr(rls, scheduler))
    for j in range(4):
        assert np.array_equal(xs[j].to_host(), cols[j])
        assert rel(cols[j], X[:, j]) < 1e-3
    again = rls.solve_(S, rls.DeviceVector.from_host(B[:, 0])).to_host()
    assert np.array_equal(again, cols[0])


def test_power_iterations(rls, ctx):
    A, _, _ = O.make_problem(300, 100, np.complex64, 31)
    rng = np.random.default_rng(0)
    v0 = rnd(rng, 100, np.complex64)
    want = O.power_iterations(O.NormalOp(O.DenseOp(A)), v0)
    got = rls.power_iterations(rls.DeviceMatrix.from_host(A).normal_operator(), rls.DeviceVector.from_host(v0))
    assert abs(got - want) < 1e-4 * want


def test_errors_are_loud(rls, ctx):
    A, _, b = O.make_problem(16, 8, np.float32, 1)
    Ad = rls.DeviceMatrix.from_host(A)
    with pytest.raises(ValueError, match="DimensionMismatch"):
        rls.solve_(rls.CGNR(Ad), rls.DeviceVector.from_host(b[:5]))
    with pytest.raises(TypeError):
        rls.DeviceVector.from_host(np.zeros(4, np.float16))   # (Float64 / ComplexF64 are element types since round 6: tests/test_gpu_float64.py)
    with pytest.raises(TypeError, match="element types differ"):   # ... and a Float32 right-hand side does not meet a Float64 operator
        rls.solve_(rls.createLinearSolver(rls.CGNR, rls.DeviceMatrix.from_host(A.astype(np.float64)), iterations=3),
                   rls.DeviceVector.from_host(b))
    with pytest.raises(TypeError, match="element types differ"):
        rls.DeviceVector.from_host(b).axpy_(1.0, rls.DeviceVector.from_host(b.astype(np.float64)))
    with pytest.raises(rls.RLSError):
        rls.prox_(rls.L21Regularization, rls.DeviceVector.from_host(np.ones(4, np.float32)), 0.1, slices=9)


def test_row_sharded_cgnr_single_rank_on_gpu(rls, ctx):
    """BASELINE config 5 control flow on one GPU (world = 1: the all-reduce is the identity): the split
    half-steps rls_cgnr_{init,step}_local_{a,b} on torch-owned state vectors equal the oracle"""
    import torch

    A, xt, b = O.make_problem(1024, 384, np.complex64, 13)
    ops = rls.multigpu.HipLocalOps(rls, A, torch.cuda.current_device())
    s = rls.RowShardedCGNR(ops, None, lam=1e-3, iterations=16, relTol=0.0)
    x = s.solve(b)
    ref = O.CGNR(A.astype(np.complex128), reg=O.L2Regularization(1e-3), iterations=16, relTol=0.0)
    O.solve(ref, b.astype(np.complex128))
    assert rel(x, ref.x) < TOL_ITER and ops.status()["iteration"] == 16
    ops.close()


class _TwoShards:
    """two row shards of one problem on ONE GPU behind the local-ops protocol, with a stand-in for torch.distributed
    whose all_reduce sums the two shards' tensors: exercises the partial products of the `local_a` halves and the
    replicated `local_b` halves exactly as two ranks would (same stream, so the order is the enqueue order)"""

    class Dist:
        class ReduceOp:
            SUM = "sum"

        @staticmethod
        def get_world_size():
            return 2

        @staticmethod
        def all_reduce(pair, op=None):
            total = pair[0] + pair[1]
            pair[0].copy_(total)
            pair[1].copy_(total)

    def __init__(self, a, b):
        self.shards = (a, b)

    def tensor(self, name):
        return tuple(s.tensor(name) for s in self.shards)

    def __getattr__(self, name):
        def call(*args):
            if name in ("init_a",):
                outs = [s.init_a(part) for s, part in zip(self.shards, args[0])]
            else:
                outs = [getattr(s, name)(*args) for s in self.shards]
            return outs[0]
        return call


@pytest.mark.parametrize("dt", [np.complex64, np.float32])
def test_row_sharded_fista_two_shards_on_gpu(rls, ctx, dt):
    """SURVEY 8e last row: FISTA on a row-partitioned A; x0 and res are all-reduced, everything else is replicated"""
    import torch

    M, N = 768, 256
    A, xt, b = O.make_problem(M, N, dt, 23)
    dt64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    rho = 0.9 / np.linalg.norm(A.astype(dt64), 2) ** 2
    lam = 0.05 * np.max(np.abs(A.conj().T @ b))
    dev = torch.cuda.current_device()
    lo = 388  # unequal shards, 4-row aligned
    mk = lambda rows: rls.multigpu.HipFistaOps(rls, np.asfortranarray(A[rows]), dev, reg=rls.L1Regularization(lam),
                                               proj=rls.PositiveRegularization() if dt == np.float32 else None)
    pair = _TwoShards(mk(slice(0, lo)), mk(slice(lo, M)))
    f = rls.RowShardedFISTA(pair, _TwoShards.Dist, rho=rho, iterations=30, relTol=0.0, restart="gradient")
    f.init((b[:lo], b[lo:]))
    f.step(30)
    xs = [s.solution() for s in pair.shards]
    assert np.array_equal(xs[0], xs[1])  # replicated state: bit-identical on both shards
    regs = [O.L1Regularization(lam)] + ([O.PositiveRegularization()] if dt == np.float32 else [])
    x64, x32 = oracle_pair(lambda A_, b_: O.solve(O.FISTA(A_, reg=regs, rho=rho, iterations=30, relTol=0.0, restart="gradient"), b_), A, b)
    parity(f"fista_rowsharded_2shards_768x256_{np.dtype(dt).name}", xs[0], x64, x32)
    assert all(s.status()["iteration"] == 30 for s in pair.shards)
    # world = 1 (no collective) through the same entry points
    one = rls.multigpu.HipFistaOps(rls, A, dev, reg=rls.L1Regularization(lam),
                                   proj=rls.PositiveRegularization() if dt == np.float32 else None)
    x1 = rls.RowShardedFISTA(one, None, rho=rho, iterations=30, relTol=0.0, restart="gradient").solve(b)
    parity(f"fista_rowsharded_1shard_768x256_{np.dtype(dt).name}", x1, x64, x32)
    for s in pair.shards + (one,):
        s.close()


@pytest.mark.parametrize("kind", ["l1", "tv"])
def test_row_sharded_admm_two_shards_on_gpu(rls, ctx, kind):
    """SURVEY 8e last row: ADMM on a row-partitioned A; only cg!'s operator applies are all-reduced"""
    import torch

    M, N = 640, 144
    A, xt, b = O.make_problem(M, N, np.float32, 29)
    dev = torch.cuda.current_device()
    lo = 300
    reg_d = rls.L1Regularization(0.05) if kind == "l1" else rls.TVRegularization(0.05, shape=(12, 12))
    reg_o = O.L1Regularization(0.05) if kind == "l1" else O.TVRegularization(0.05, shape=(12, 12))
    mk = lambda rows: rls.multigpu.HipAdmmOps(rls, np.asfortranarray(A[rows]), dev, reg=reg_d)
    pair = _TwoShards(mk(slice(0, lo)), mk(slice(lo, M)))
    kw = dict(rho=0.3, iterations=9, iterationsCG=6, tolInner=1e-4)
    a = rls.RowShardedADMM(pair, _TwoShards.Dist, lam=0.05, **kw)
    a.init((b[:lo], b[lo:]), M)
    while a.iterate() is not None:
        pass
    xs = [s.solution() for s in pair.shards]
    assert np.array_equal(xs[0], xs[1])
    ref = O.ADMM(A, reg=reg_o, **kw)
    O.solve(ref, b)
    assert a.iteration == ref.iteration and a.cg_iterations == ref.cg_iters
    ref64 = O.ADMM(A.astype(np.float64), reg=reg_o, **kw)
    parity(f"admm_rowsharded_2shards_{kind}_640x144_f32", xs[0], O.solve(ref64, b.astype(np.float64)), ref.x)
    assert np.allclose(a.rk, ref.rk, rtol=2e-3, atol=1e-6) and np.allclose(a.sk, ref.sk, rtol=2e-3, atol=1e-6)
    for s in pair.shards:
        s.close()


def test_multisolve_single_rank(rls, ctx):
    """BASELINE config 4 sharding at world = 1 (all columns local) equals column solves"""
    A, X, B = O.make_problem(128, 64, np.complex64, 17, n_rhs=5)
    Ad = rls.DeviceMatrix.from_host(A)
    ms = rls.MultiSolve(rls, lambda: rls.createLinearSolver(rls.CGNR, Ad, iterations=64))
    got = ms.solve(B)
    assert got.shape == (64, 5) and rel(got, X) < 1e-3
    # default: the local columns share A (BatchedState); the reference's scheduler gives the same columns
    seq = rls.MultiSolve(rls, lambda: rls.createLinearSolver(rls.CGNR, Ad, iterations=64), scheduler=rls.MultiThreadingState).solve(B)
    for j in range(5):  # both schedulers against the oracle, column by column
        x64, x32 = oracle_pair(lambda A_, b_: O.solve(O.CGNR(A_, iterations=64), b_), A, B[:, j])
        parity(f"multisolve_cgnr_batched_col{j}", got[:, j], x64, x32)
        parity(f"multisolve_cgnr_threads_col{j}", seq[:, j], x64, x32)
    rho = 0.9 / np.linalg.norm(A.astype(np.complex128), 2) ** 2
    mk = lambda: rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-3), rho=rho, iterations=30)
    f_b = rls.MultiSolve(rls, mk).solve(B)
    f_s = rls.MultiSolve(rls, mk, scheduler=rls.MultiThreadingState).solve(B)
    assert f_b.shape == (64, 5)
    for j in range(5):
        x64, x32 = oracle_pair(lambda A_, b_: O.solve(O.FISTA(A_, reg=O.L1Regularization(1e-3), rho=rho, iterations=30), b_), A, B[:, j])
        parity(f"multisolve_fista_batched_col{j}", f_b[:, j], x64, x32)
        parity(f"multisolve_fista_threads_col{j}", f_s[:, j], x64, x32)


@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("dt,M,N,K", [(np.complex64, 4096, 2048, 8), (np.float32, 512, 256, 3), (np.complex64, 96, 40, 5),
                                      (np.complex64, 272, 144, 20), (np.float32, 1040, 208, 33)])
def test_batched_matrix_rhs_shares_one_pass_over_A(rls, ctx, dt, M, N, K, mfma):
    ctx.tune(batched_mfma=mfma)
    try:
        _batched_case(rls, ctx, dt, M, N, K)
    finally:
        ctx.tune(batched_mfma=1)


def _batched_case(rls, ctx, dt, M, N, K):
    """BatchedState: K right-hand sides per pass over A == column-by-column solves (and the oracle),
    including columns that retire early (relTol) while others continue"""
    A, X, B = O.make_problem(M, N, dt, 23, n_rhs=K)
    B = np.asfortranarray(B)
    B[:, 0] *= 1e-3  # different scales: per-column scalars must stay independent
    Ad = rls.DeviceMatrix.from_host(A)
    iters = 12
    for relTol in (0.0, 1e-4):
        S = rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(1e-3), iterations=iters, relTol=relTol)
        xs = rls.solve_(S, rls.DeviceMatrix.from_host(B), scheduler=rls.BatchedState)
        its = [s_.iteration for s_ in S.state.status()] if isinstance(S.state, rls.BatchedState) else None
        dt64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
        for j in range(K):
            ref = O.CGNR(A.astype(dt64), reg=O.L2Regularization(1e-3), iterations=iters, relTol=relTol)
            O.solve(ref, B[:, j].astype(dt64))
            # Float32 bound: the oracle's Float32 run stopped at the float64 run's iteration (the relTol threshold is
            # itself a Float32-conditioned decision)
            x32 = lambda: O.solve(O.CGNR(A, reg=O.L2Regularization(1e-3), iterations=ref.iteration, relTol=0.0), np.ascontiguousarray(B[:, j]))
            if its is None or its[j] == ref.iteration:
                parity(f"batched_cgnr_{M}x{N}_{np.dtype(dt).name}_K{K}_reltol{relTol}_col{j}", xs[j].to_host(), ref.x, x32,
                       record=(j < 2))
            if its is not None:
                assert abs(its[j] - ref.iteration) <= (1 if relTol > 0 else 0), (its, j, ref.iteration)
    # a vector solve still works afterwards (src/MultiThreading.jl:39-43)
    S2 = rls.createLinearSolver(rls.CGNR, Ad, iterations=iters, relTol=0.0)
    rls.solve_(S2, rls.DeviceMatrix.from_host(B), scheduler=rls.BatchedState)
    v = rls.solve_(S2, rls.DeviceVector.from_host(B[:, 1])).to_host()
    x64, x32 = oracle_pair(lambda A_, b_: O.solve(O.CGNR(A_, iterations=iters, relTol=0.0), b_), A, np.ascontiguousarray(B[:, 1]))
    parity(f"batched_then_vector_{M}x{N}_{np.dtype(dt).name}", v, x64, x32)


@pytest.mark.parametrize("dt,M,N,K,resident", [(np.complex64, 4096, 2048, 8, 1), (np.complex64, 4096, 2048, 8, 0), (np.float32, 512, 256, 3, 1),
                                               (np.complex64, 272, 144, 20, 1), (np.complex64, 1040, 208, 7, 1), (np.complex64, 1040, 208, 7, 0),
                                               (np.complex64, 2000, 1936, 5, 1), (np.float32, 1040, 208, 33, 1)])
def test_batched_gram_mode_matrix_rhs(rls, ctx, dt, M, N, K, resident):
    """Matrix right-hand sides on the reference constructors' DEFAULT operator for a dense matrix, AHA = A' * A explicit
    (src/CGNR.jl:49, src/FISTA.jl:58, src/ADMM.jl:81): every column's state shares that one solver.AHA
    (src/MultiThreading.jl:30-48), so a batched iteration is ONE skinny product V = AHA P over N x N elements instead of two
    passes over A.  CGNR (with per-column relTol retirement), FISTA + L1 and ADMM + L1 columns against the float64
    oracle's Gram-mode solves.  resident = 1: <= 8 ComplexF32 columns with N <= 2048 run the whole step call as ONE launch
    (csrc/gramk.hip: AHA in the register files, the operand panel in LDS, path 7); resident = 0 and every other case: path 6."""
    ctx.tune(resident=resident)
    try:
        A, X, B = O.make_problem(M, N, dt, 29, n_rhs=K)
        B = np.asfortranarray(B)
        B[:, 0] *= 1e-3
        dt64 = hi(dt)
        Ad = rls.DeviceMatrix.from_host(A)
        Gd = Ad.gram()
        Bd = rls.DeviceMatrix.from_host(B)
        iters = 12
        tag = f"batched_gram_{M}x{N}_{np.dtype(dt).name}_K{K}_res{resident}"
        A64 = A.astype(dt64)
        G64, G32 = A64.conj().T @ A64, A.conj().T @ A   # formed ONCE for the K oracle solves (what normal="gram" forms per solver: the same product)
        for relTol in (0.0, 1e-4):
            S = rls.createLinearSolver(rls.CGNR, Ad, AHA=Gd, reg=rls.L2Regularization(1e-3), iterations=iters, relTol=relTol)
            xs = rls.solve_(S, Bd, scheduler=rls.BatchedState)
            assert isinstance(S.state, rls.BatchedState), "Gram-mode matrix solves must take the shared-AHA plan"
            fits = resident == 1 and np.dtype(dt).kind == "c" and K <= 8 and N <= 2048
            assert _cgnr_path(rls, S) == (7 if fits else 6), _cgnr_path(rls, S)
            stat = S.state.status()
            assert all(s_.fallbacks == 0 for s_ in stat)
            its = [s_.iteration for s_ in stat]
            for j in range(K):
                ref = O.CGNR(A64, AHA=G64, reg=O.L2Regularization(1e-3), iterations=iters, relTol=relTol)
                O.solve(ref, B[:, j].astype(dt64))
                x32 = lambda: O.solve(O.CGNR(A, AHA=G32, reg=O.L2Regularization(1e-3), iterations=ref.iteration, relTol=0.0),
                                      np.ascontiguousarray(B[:, j]))
                assert abs(its[j] - ref.iteration) <= (1 if relTol > 0 else 0), (its, j, ref.iteration)
                if its[j] == ref.iteration:
                    parity(f"{tag}_cgnr_reltol{relTol}_col{j}", xs[j].to_host(), ref.x, x32, record=(j < 2))
        rho = float(0.9 / np.linalg.norm(A.astype(dt64), 2) ** 2)
        lam = 1e-3 * float(np.abs(A.conj().T @ B[:, 1]).max())
        F = rls.createLinearSolver(rls.FISTA, Ad, AHA=Gd, reg=rls.L1Regularization(lam), rho=rho, iterations=15, relTol=0.0)
        fs = rls.solve_(F, Bd, scheduler=rls.BatchedState)
        assert isinstance(F.state, rls.FistaBatchedState)
        assert _fista_path(ctx, F) == (7 if fits else 3), _fista_path(ctx, F)
        for j in (0, 1, K - 1):
            x64, x32 = oracle_pair(lambda A_, b_: O.solve(O.FISTA(A_, reg=O.L1Regularization(lam), rho=rho, iterations=15, relTol=0.0,
                                                                 normal="gram"), b_), A, np.ascontiguousarray(B[:, j]))
            parity(f"{tag}_fista_col{j}", fs[j].to_host(), x64, x32, record=(j < 2))
        kw = dict(rho=0.1, iterations=4, iterationsCG=5, absTol=0.0, relTol=0.0, tolInner=1e-5)
        D = rls.createLinearSolver(rls.ADMM, Ad, AHA=Gd, reg=rls.L1Regularization(1e-3), **kw)
        ds = rls.solve_(D, Bd, scheduler=rls.BatchedState)
        assert isinstance(D.state, rls.AdmmBatchedState)
        for j in (0, K - 1):
            x64, x32 = oracle_pair(lambda A_, b_: O.solve(O.ADMM(A_, reg=O.L1Regularization(1e-3), normal="gram", **kw), b_), A,
                                   np.ascontiguousarray(B[:, j]))
            parity(f"{tag}_admm_col{j}", ds[j].to_host(), x64, x32, record=(j < 1))
    finally:
        ctx.tune(resident=1)


@pytest.mark.parametrize("M,N,K", [(4096, 2048, 8), (272, 144, 3), (1040, 208, 7)])
def test_batched_half_operand_panels_change_no_bit(rls, ctx, M, N, K):
    """Up to 8 ComplexF32 right-hand sides ride the matrix cores as (8 re | 8 im) operand columns -- two MFMAs per
    complex block instead of four (csrc/skinny.hip, H = true).  The four FMA chains per output element are the same
    chains in the same order as in the 16-column layout, so CGNR, FISTA and ADMM batched solves must agree with the
    full layout bit for bit (and a column's result must not depend on the layout its batch happened to get)."""
    A, X, B = O.make_problem(M, N, np.complex64, 41, n_rhs=K)
    B = np.asfortranarray(B)
    Ad = rls.DeviceMatrix.from_host(A)
    Bd = rls.DeviceMatrix.from_host(B)
    makers = {
        "cgnr": lambda: rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(1e-3), iterations=9, relTol=0.0),
        # explicit step size: the default is a power iteration from a random start (src/FISTA.jl:76), new per solver
        "fista": lambda: rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-3), iterations=9, relTol=0.0,
                                                rho=float(8.0 / np.linalg.norm(A) ** 2)),
        "admm": lambda: rls.createLinearSolver(rls.ADMM, Ad, reg=rls.L1Regularization(1e-3), rho=0.1, iterations=4,
                                               iterationsCG=5, absTol=0.0, relTol=0.0),
    }
    for name, make in makers.items():
        got = {}
        for half in (1, 0):
            ctx.tune(skinny_half=half)
            try:
                S = make()
                xs = rls.solve_(S, Bd, scheduler=rls.BatchedState)
                got[half] = np.stack([x.to_host() for x in xs], axis=1)
            finally:
                ctx.tune(skinny_half=1)
        assert np.isfinite(got[1]).all() and np.abs(got[1]).max() > 0
        assert np.array_equal(got[1], got[0]), (name, np.abs(got[1] - got[0]).max())


@pytest.mark.parametrize("name,kw", [("OptISTA", {}), ("POGM", {}), ("POGM", {"restart": "gradient"})])
@pytest.mark.parametrize("dt,M,N", [(np.complex64, 256, 96), (np.float32, 4096, 2048)])
def test_optista_pogm_match_oracle(rls, ctx, name, kw, dt, M, N):
    """OptISTA (src/OptISTA.jl:169-209) and POGM (src/POGM.jl:169-237) re-sequence the same
    device kernels; iterates against the float64 oracle"""
    A, xt, b = O.make_problem(M, N, dt, 31)
    dt64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    A64, b64 = A.astype(dt64), b.astype(dt64)
    rho = 0.95 / np.linalg.norm(A64, 2) ** 2
    lam = 1e-2 * np.max(np.abs(A64.conj().T @ b64))
    x64, x32 = oracle_pair(lambda A_, b_: O.solve(getattr(O, name)(A_, reg=O.L1Regularization(lam), rho=rho, iterations=30, **kw), b_), A, b)
    sol = rls.createLinearSolver(getattr(rls, name), rls.DeviceMatrix.from_host(A), reg=rls.L1Regularization(lam),
                                 rho=rho, iterations=30, **kw)
    x = rls.solve_(sol, rls.DeviceVector.from_host(b)).to_host()
    assert sol.state.iteration == 30
    parity(f"{name}{'_restart' if kw else ''}_{M}x{N}_{np.dtype(dt).name}", x, x64, x32)


@pytest.mark.parametrize("name,kw", [("OptISTA", {}), ("POGM", {"restart": "gradient"})])
def test_optista_pogm_generic_path_and_projection(rls, ctx, name, kw):
    """regularisers the fused update kernels do not cover (L21) run the primitive-by-primitive path; POGM with a
    Positive projection runs fused (src/POGM.jl:212-216)"""
    A, xt, b = O.make_problem(192, 64, np.complex64, 37)
    A64, b64 = A.astype(np.complex128), b.astype(np.complex128)
    rho = 0.95 / np.linalg.norm(A64, 2) ** 2
    lam = 1e-2 * np.max(np.abs(A64.conj().T @ b64))
    x64, x32 = oracle_pair(lambda A_, b_: O.solve(getattr(O, name)(A_, reg=O.L21Regularization(lam, slices=4), rho=rho, iterations=20, **kw), b_), A, b)
    sol = rls.createLinearSolver(getattr(rls, name), rls.DeviceMatrix.from_host(A), reg=rls.L21Regularization(lam, slices=4),
                                 rho=rho, iterations=20, **kw)
    parity(f"{name}_l21_generic_192x64_c64", rls.solve_(sol, rls.DeviceVector.from_host(b)).to_host(), x64, x32)
    if name == "POGM":
        x64, x32 = oracle_pair(lambda A_, b_: O.solve(O.POGM(A_, reg=[O.L1Regularization(lam), O.PositiveRegularization()], rho=rho,
                                                            iterations=20, **kw), b_), A, b)
        sol = rls.createLinearSolver(rls.POGM, rls.DeviceMatrix.from_host(A),
                                     reg=[rls.L1Regularization(lam), rls.PositiveRegularization()], rho=rho, iterations=20, **kw)
        parity("POGM_l1_positive_192x64_c64", rls.solve_(sol, rls.DeviceVector.from_host(b)).to_host(), x64, x32)


def test_split_bregman_matches_oracle(rls, ctx):
    """src/SplitBregman.jl:204-271, identity regTrafo and GradientOp regTrafo"""
    A, xt, b = O.make_problem(160, 64, np.float32, 33)
    kw = dict(rho=0.5, iterations=3, iterationsInner=4, iterationsCG=10)
    ref = O.SplitBregman(A.astype(np.float64), reg=O.L1Regularization(0.05), **kw)
    O.solve(ref, b.astype(np.float64))
    sol = rls.createLinearSolver(rls.SplitBregman, rls.DeviceMatrix.from_host(A), reg=rls.L1Regularization(0.05), **kw)
    x = rls.solve_(sol, rls.DeviceVector.from_host(b)).to_host()
    parity("splitbregman_l1_160x64_f32", x, ref.x, lambda: O.solve(O.SplitBregman(A, reg=O.L1Regularization(0.05), **kw), b))
    assert sol.state.iter_cnt == ref.iter_cnt
    ref2 = O.SplitBregman(A.astype(np.float64), reg=O.L1Regularization(0.05), regTrafo=O.GradientTrafo((8, 8)), **kw)
    O.solve(ref2, b.astype(np.float64))
    sol2 = rls.createLinearSolver(rls.SplitBregman, rls.DeviceMatrix.from_host(A), reg=rls.L1Regularization(0.05),
                                  regTrafo=rls.GradientOp((8, 8)), **kw)
    x2 = rls.solve_(sol2, rls.DeviceVector.from_host(b)).to_host()
    parity("splitbregman_l1_gradient_trafo_160x64_f32", x2, ref2.x,
           lambda: O.solve(O.SplitBregman(A, reg=O.L1Regularization(0.05), regTrafo=O.GradientTrafo((8, 8)), **kw), b))


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
def test_normalization_schemes(rls, ctx, dt):
    """MeasurementBased / SystemMatrixBased normalisation (src/Regularization/NormalizedRegularization.jl
    :40-84) -- the factor is computed on the device and scales lambda exactly as the oracle's restatement does"""
    A, xt, b = O.make_problem(192, 80, dt, 41)
    dt64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    A64, b64 = A.astype(dt64), b.astype(dt64)
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    f_sys = O.normalization_factor("systemmatrix", A64, None)
    assert abs(Ad.rownorm2().norm1() / A.shape[1] - f_sys) < 1e-5 * f_sys
    assert rel(Ad.rownorm2().to_host(), np.sum(np.abs(A64) ** 2, axis=1)) < 1e-6
    lam = 0.05
    for scheme, key, vec in ((rls.SystemMatrixBasedNormalization(), "systemmatrix", None),
                             (rls.MeasurementBasedNormalization(), "measurement", b64)):
        f = O.normalization_factor(key, A64, vec)
        S = rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(lam), normalizeReg=scheme, iterations=20)
        x = rls.solve_(S, bd).to_host()
        assert abs(S.L2.lam - lam * f) < 1e-5 * lam * f and rls.scalefactor(S.L2) == pytest.approx(f, rel=1e-5)
        x64, x32 = oracle_pair(lambda A_, b_: O.solve(O.CGNR(A_, reg=O.L2Regularization(lam * f), iterations=20), b_), A, b)
        parity(f"cgnr_normalized_{key}_192x80_{np.dtype(dt).name}", x, x64, x32)
        # solving again re-normalises from the unscaled lambda (NormalizedRegularization.jl:73), it does not compound
        rls.solve_(S, bd)
        assert abs(S.L2.lam - lam * f) < 1e-5 * lam * f
    # FISTA normalises with x0 = A^H b (src/FISTA.jl:128)
    rho = 0.95 / np.linalg.norm(A64, 2) ** 2
    f = O.normalization_factor("measurement", A64, A64.conj().T @ b64)
    S = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-3), normalizeReg=rls.MeasurementBasedNormalization(),
                               rho=rho, iterations=25)
    x = rls.solve_(S, bd).to_host()
    x64, x32 = oracle_pair(lambda A_, b_: O.solve(O.FISTA(A_, reg=O.L1Regularization(1e-3 * f), rho=rho, iterations=25), b_), A, b)
    parity(f"fista_normalized_192x80_{np.dtype(dt).name}", x, x64, x32)
    with pytest.raises(ValueError):
        rls.createLinearSolver(rls.CGNR, AHA=Ad.gram(), reg=rls.L2Regularization(lam),
                               normalizeReg=rls.SystemMatrixBasedNormalization())


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
def test_weighted_operator(rls, ctx, dt):
    """ProdOp(WeightingOp(w), A) and its normal operator A^H W^H W A
    (docs/src/literate/howto/normal_operator.jl:37-68): solve!(solver(WA; AHA = normalOperator(WA)), w .* b)"""
    A, xt, b = O.make_problem(4096 if np.dtype(dt).kind == "c" else 256, 2048 if np.dtype(dt).kind == "c" else 96, dt, 43)
    rng = np.random.default_rng(5)
    w = rng.uniform(0.5, 1.5, A.shape[0]).astype(np.float32)
    dt64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    WA64 = O.weighted_operator(w.astype(np.float64), A.astype(dt64))
    wb64 = w.astype(np.float64) * b.astype(dt64)
    WA = rls.ProdOp(rls.WeightingOp(rls.DeviceVector.from_host(w)), rls.DeviceMatrix.from_host(A))
    assert rel(WA.to_host(), WA64) < 1e-6
    S = rls.createLinearSolver(rls.CGNR, WA, AHA=rls.normalOperator(WA), reg=rls.L2Regularization(1e-4), iterations=16)
    x = rls.solve_(S, rls.DeviceVector.from_host((w * b).astype(dt))).to_host()
    ref = O.CGNR(WA64, reg=O.L2Regularization(1e-4), iterations=16)
    parity(f"cgnr_weighted_{A.shape[0]}x{A.shape[1]}_{np.dtype(dt).name}", x, O.solve(ref, wb64),
           lambda: O.solve(O.CGNR(O.weighted_operator(w, A), reg=O.L2Regularization(1e-4), iterations=16), (w * b).astype(dt)))
    f = O.normalization_factor("systemmatrix", WA64, None)  # weights^2 * rownorm², GPU ext NormalizedRegularization.jl:7-12
    assert abs(WA.rownorm2().norm1() / A.shape[1] - f) < 1e-5 * f


@pytest.mark.parametrize("dt,M,N", [(np.complex64, 1024, 512), (np.float32, 528, 272), (np.complex64, 100, 36),
                                    (np.complex64, 4096, 2048), (np.float32, 400, 192), (np.complex64, 48, 64)])
def test_gram_matrix_cores(rls, ctx, dt, M, N):
    """setup GEMM AHA = A' * A (src/CGNR.jl:49): matrix-core path for 16-aligned shapes, plain kernel otherwise"""
    A, _, _ = O.make_problem(M, N, dt, 47)
    G = rls.DeviceMatrix.from_host(A).gram().to_host()
    ref = A.astype(np.complex128 if np.dtype(dt).kind == "c" else np.float64)
    ref = ref.conj().T @ ref
    assert rel(G, ref) < 2e-6
    if N % 64 == 0 and M % 16 == 0:  # Hermitian tile kernel: symmetric bit for bit
        assert np.array_equal(G, G.conj().T)


# ---- Kaczmarz --------------------------------------------------------------------


def _one_row_update(rls, ctx, A, k, beta):
    """kaczmarz_update!(A, x, k, beta) through the sweep entry point: x = 0, denom = 1, u[k] = beta, eps_w = 0"""
    M, N = A.shape
    Ad = rls.DeviceMatrix.from_host(A)
    At = rls.DeviceMatrix(N, M, A.dtype, ctx)
    lib, h = ctx.lib, ctx.handle
    rls._lib.check(h, lib.rls_transpose(h, Ad.code, M, N, Ad.ptr, Ad.lda, At.ptr, At.lda), "rls_transpose")
    assert np.array_equal(At.to_host(), A.T)
    u = np.zeros(M, A.dtype)
    u[k] = beta
    x, ud, vl = rls.DeviceVector.from_host(np.zeros(N, A.dtype)), rls.DeviceVector.from_host(u), rls.DeviceVector.from_host(np.zeros(M, A.dtype))
    rows = rls.DeviceVector.from_host(np.array([k], np.int32).view(np.float32))
    den = rls.DeviceVector.from_host(np.ones(1, np.float32))
    rls._lib.check(h, lib.rls_kaczmarz_sweep(h, Ad.code, M, N, At.ptr, At.lda, 1, x.ptr, N, ud.ptr, M, vl.ptr, M, rows.ptr,
                                             den.ptr, 1, 0.0, 1), "rls_kaczmarz_sweep")
    return x.to_host()


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
@pytest.mark.parametrize("M,N", [(16, 127), (127, 16), (40, 2048), (8, 5000)])
def test_kaczmarz_update_known_answer(rls, ctx, dt, M, N):
    """test/testKaczmarz.jl:6-35: kaczmarz_update!(A, b, k, beta) == beta * conj(A[k, :]) (odd and aligned lengths)"""
    rng = np.random.default_rng(3)
    A = rng.random((M, N)).astype(dt)
    beta = dt(0.37)
    if np.dtype(dt).kind == "c":
        A = (A + 1j * rng.random((M, N))).astype(dt)
        beta = dt(0.37 - 0.81j)
    k = int(rng.integers(0, M))
    got = _one_row_update(rls, ctx, np.asfortranarray(A), k, beta)
    assert np.allclose(got, beta * np.conj(A[k, :]), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("dt,M,N,lam,its", [(np.complex64, 127, 16, 0.0, 5), (np.float32, 256, 96, 0.05, 4),
                                            (np.complex64, 96, 250, 0.1, 3), (np.complex64, 4096, 2048, 1e-2, 2)])
def test_kaczmarz_matches_oracle(rls, ctx, dt, M, N, lam, its):
    """src/Kaczmarz.jl:283-308: x and vl after `its` full sweeps against the float64 oracle"""
    A, xt, b = O.make_problem(M, N, dt, 51)
    A[min(3, M - 1), :] = 0  # a zero row is skipped by initkaczmarz (:376)
    dt64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    ref = O.Kaczmarz(A.astype(dt64), reg=O.L2Regularization(lam), iterations=its)
    xr = O.solve(ref, b.astype(dt64))
    S = rls.createLinearSolver(rls.Kaczmarz, rls.DeviceMatrix.from_host(A), reg=rls.L2Regularization(lam), iterations=its)
    assert len(S.rowindex) == M - 1
    x = rls.solve_(S, rls.DeviceVector.from_host(b)).to_host()
    assert S.state.iteration == its
    ref32 = O.Kaczmarz(A, reg=O.L2Regularization(lam), iterations=its)
    x32 = O.solve(ref32, b)
    parity(f"kaczmarz_{M}x{N}_{np.dtype(dt).name}_x", x, xr, x32)
    if np.linalg.norm(ref.vl) > 0:
        parity(f"kaczmarz_{M}x{N}_{np.dtype(dt).name}_vl", S.state.vl.to_host(), ref.vl, ref32.vl)
    else:
        assert np.linalg.norm(S.state.vl.to_host()) <= 1e-12
    res = rls.solverconvergence(S)["residual"]
    assert abs(res - np.linalg.norm(A.astype(dt64) @ xr - b.astype(dt64))) < 1e-3 * np.linalg.norm(b) + 1e-6
    # step-by-step with callbacks == the single multi-sweep launch (callback cadence 0..n, test/testCallbacks.jl:6-16)
    calls = []
    S2 = rls.createLinearSolver(rls.Kaczmarz, rls.DeviceMatrix.from_host(A), reg=rls.L2Regularization(lam), iterations=its)
    x2 = rls.solve_(S2, rls.DeviceVector.from_host(b), callbacks=lambda s_, it: calls.append(it)).to_host()
    assert calls == list(range(its + 1)) and rel(x2, x) < 1e-6


def test_kaczmarz_reference_test_suite(rls, ctx):
    """test/testKaczmarz.jl:37-131 replayed on the device (Float32 instead of Float64 operands)"""
    rng = np.random.default_rng(12345)
    M, N = 12, 8
    A = (rng.random((M, N)) + 1j * rng.random((M, N))).astype(np.complex64)
    x = (rng.random(N) + 1j * rng.random(N)).astype(np.complex64)
    b = (A.astype(np.complex128) @ x).astype(np.complex64)
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)

    def solve(Amat, iterations, bvec=bd, **kw):
        S = rls.createLinearSolver(rls.Kaczmarz, Amat, iterations=iterations, **kw)
        out = rls.solve_(S, bvec)
        return out.to_host()

    # Tikhonov matrix (:37-70)
    regm = rng.random(N).astype(np.float32) + 0.1
    x_matrix = solve(Ad, 100, reg=[rls.L2Regularization(regm)])
    As = rls.DeviceMatrix.from_host(np.asfortranarray(A * (1 / np.sqrt(regm))[None, :]).astype(np.complex64))
    x_approx = solve(As, 100, reg=[rls.L2Regularization(1.0)]) / np.sqrt(regm)
    assert rel(x_matrix, x_approx) < 1e-4
    lam = float(rng.random())
    assert rel(solve(Ad, 100, reg=[rls.L2Regularization(np.full(N, lam, np.float32))]),
               solve(Ad, 100, reg=[rls.L2Regularization(lam)])) < 1e-4
    # weighting matrix (:72-90): d * A  vs  ProdOp(WeightingOp(w), A)
    w = rng.random(M).astype(np.float32) + 0.1
    reg = rls.L2Regularization(float(rng.random()))
    dA = rls.DeviceMatrix.from_host(np.asfortranarray(w[:, None] * A).astype(np.complex64))
    wA = rls.ProdOp(rls.WeightingOp(rls.DeviceVector.from_host(w)), Ad)
    db = rls.DeviceVector.from_host((w * b).astype(np.complex64))
    assert rel(solve(wA, 200, bvec=db, reg=reg), solve(dA, 200, bvec=db, reg=reg)) < 1e-5
    # parameters (:94-131)
    assert rel(solve(Ad, 200), x) < 0.1
    assert rel(solve(Ad, 200, shuffleRows=True), x) < 0.1
    assert rel(solve(Ad, 2000, randomized=True), x) < 0.1
    with pytest.raises(NotImplementedError):
        rls.createLinearSolver(rls.Kaczmarz, Ad, greedy_randomized=True)
    for strategy in (rls.SystemMatrixBasedNormalization(), rls.MeasurementBasedNormalization()):
        xa = solve(Ad, 200, randomized=True, reg=rls.L2Regularization(0.1), normalizeReg=strategy)
        assert rel(xa, x) < 0.3
    # closed form: Kaczmarz on the extended system converges to the Tikhonov solution
    A64 = A.astype(np.complex128)
    xt = np.linalg.solve(A64.conj().T @ A64 + 0.3 * np.eye(N), A64.conj().T @ b.astype(np.complex128))
    assert rel(solve(Ad, 3000, reg=rls.L2Regularization(0.3)), xt) < 1e-4
    # projection applied after every sweep (:294-296)
    S = rls.createLinearSolver(rls.Kaczmarz, Ad, iterations=5, reg=[rls.L2Regularization(0.0), rls.RealRegularization()])
    ref = O.Kaczmarz(A64, reg=[O.L2Regularization(0.0), O.RealRegularization()], iterations=5)
    parity("kaczmarz_real_projection", rls.solve_(S, bd).to_host(), O.solve(ref, b.astype(np.complex128)),
           lambda: O.solve(O.Kaczmarz(A64.astype(np.complex64), reg=[O.L2Regularization(0.0), O.RealRegularization()], iterations=5),
                           b.astype(np.complex64)))


def test_kaczmarz_matrix_rhs_one_launch(rls, ctx):
    """columns of a matrix right-hand side: one workgroup per column in one launch == column-by-column solves"""
    A, X, B = O.make_problem(192, 128, np.complex64, 53, n_rhs=6)
    Ad = rls.DeviceMatrix.from_host(A)
    S = rls.createLinearSolver(rls.Kaczmarz, Ad, reg=rls.L2Regularization(0.02), iterations=6)
    xs = rls.solve_(S, rls.DeviceMatrix.from_host(np.asfortranarray(B)), scheduler=rls.BatchedState)
    S1 = rls.createLinearSolver(rls.Kaczmarz, Ad, reg=rls.L2Regularization(0.02), iterations=6)
    ys = rls.solve_(S1, rls.DeviceMatrix.from_host(np.asfortranarray(B)))  # MultiThreading semantics
    for j in range(6):
        one = rls.solve_(rls.createLinearSolver(rls.Kaczmarz, Ad, reg=rls.L2Regularization(0.02), iterations=6),
                         rls.DeviceVector.from_host(B[:, j])).to_host()
        assert np.array_equal(xs[j].to_host(), one) and np.array_equal(ys[j].to_host(), one)
    assert len(rls.solverconvergence(S)) == 6


# ---- BASELINE full sizes through size-independent properties -------------------------------------


def test_config4_all_64_columns_full_size(rls, ctx):
    """BASELINE config 4 (one A 4096 x 2048 CF32, 64 right-hand sides, 32 iterations, relTol = 0) on one GPU:
    every column converges to its planted solution, a sample of columns matches the float64 oracle, and the
    batched solve is linear in B (columns scaled / summed on the host give scaled / summed solutions)"""
    A, X, B = O.make_problem(4096, 2048, np.complex64, 4, n_rhs=64)
    Ad = rls.DeviceMatrix.from_host(A)
    S = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
    xs = rls.solve_(S, rls.DeviceMatrix.from_host(np.asfortranarray(B)), scheduler=rls.BatchedState)
    assert isinstance(S.state, rls.BatchedState) and len(xs) == 64
    got = np.stack([x.to_host() for x in xs], axis=1)
    errs = np.linalg.norm(got - X, axis=0) / np.linalg.norm(X, axis=0)
    assert errs.max() < 1e-4
    assert all(s_.iteration == 32 for s_ in S.state.status())
    A64 = A.astype(np.complex128)
    for j in (0, 17, 63):
        ref = O.CGNR(A64, iterations=32, relTol=0.0)
        parity(f"BASELINE config 4: batched CGNR 4096x2048 c64, 64 columns, column {j}", got[:, j],
               O.solve(ref, B[:, j].astype(np.complex128)), O.solve(O.CGNR(A, iterations=32, relTol=0.0), np.ascontiguousarray(B[:, j])))
    # linearity: column 0 <- 2 b0 - 0.5i b1 must give 2 x0 - 0.5i x1 (against the float64 solution of that column)
    B2 = np.asfortranarray(B[:, :16]).copy()
    B2[:, 0] = 2 * B[:, 0] - 0.5j * B[:, 1]
    S2 = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
    ys = rls.solve_(S2, rls.DeviceMatrix.from_host(B2), scheduler=rls.BatchedState)
    x64, x32 = oracle_pair(lambda A_, b_: O.solve(O.CGNR(A_, iterations=32, relTol=0.0), b_), A, np.ascontiguousarray(B2[:, 0]))
    parity("config 4 linearity column", ys[0].to_host(), x64, x32)
    assert rel(2 * got[:, 0] - 0.5j * got[:, 1], x64) < 1e-5
    assert np.array_equal(ys[5].to_host(), got[:, 5])  # the other columns are untouched, bit for bit


def test_config5_shard_size_properties(rls, ctx):
    """BASELINE config 5 per-GPU shard (8192 x 8192 ComplexF32 = 512 MiB, two-GEMV path): adjoint identity,
    linearity of the normal operator, monotone CGNR residual, and the iterates against a complex64 NumPy
    restatement (the float64 oracle would need 1 GiB and minutes)"""
    M = N = 8192
    rng = np.random.default_rng(500)
    A = np.empty((M, N), dtype=np.complex64, order="F")
    s_ = np.float32(1 / np.sqrt(2))
    for j0 in range(0, N, 1024):
        A[:, j0:j0 + 1024] = (rng.standard_normal((M, 1024), dtype=np.float32)
                              + 1j * rng.standard_normal((M, 1024), dtype=np.float32)) * s_
    Ad = rls.DeviceMatrix.from_host(A)
    x = (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)
    y = (rng.standard_normal(M) + 1j * rng.standard_normal(M)).astype(np.complex64)
    xd, yd = rls.DeviceVector.from_host(x), rls.DeviceVector.from_host(y)
    Ax = rls.DeviceVector(M, np.complex64, ctx)
    Ahy = rls.DeviceVector(N, np.complex64, ctx)
    Ad.mul_(Ax, xd)
    Ad.mul_adj_(Ahy, yd)
    lhs, rhs = np.vdot(y, Ax.to_host()), np.vdot(Ahy.to_host(), x)  # <y, A x> == <A^H y, x>
    assert abs(lhs - rhs) < 2e-5 * abs(lhs)
    assert rel(Ax.to_host(), A @ x) < 1e-5
    op = rls.OperatorHandle(Ad)
    z = (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)
    v1, v2, v3 = (rls.DeviceVector(N, np.complex64, ctx) for _ in range(3))
    op.mul_normal_(v1, xd)
    op.mul_normal_(v2, rls.DeviceVector.from_host(z))
    op.mul_normal_(v3, rls.DeviceVector.from_host((2 * x - 1j * z).astype(np.complex64)))
    assert rel(v3.to_host(), 2 * v1.to_host() - 1j * v2.to_host()) < 1e-5
    b = (A @ x).astype(np.complex64)
    res = []
    S = rls.createLinearSolver(rls.CGNR, Ad, iterations=8, relTol=0.0)
    got = rls.solve_(S, rls.DeviceVector.from_host(b),
                     callbacks=lambda s2, it: res.append(rls.solverconvergence(s2)["residual"])).to_host()
    assert all(b2 < a2 for a2, b2 in zip(res[1:], res[2:]))  # CG on a well-conditioned matrix: monotone here
    # float64 oracle with A kept in complex64 and widened panel by panel (a complex128 copy would be 1 GiB)
    ref64 = O.CGNR(_PanelF64Op(A), iterations=8, relTol=0.0)
    x64 = O.solve(ref64, b.astype(np.complex128))
    ref32 = O.CGNR(A, iterations=8, relTol=0.0)  # complex64 restatement = the reference's Float32 path
    parity("BASELINE config 5 shard: CGNR 8192x8192 c64 (square: ill-conditioned), 8 iterations", got, x64, O.solve(ref32, b))


class _PanelF64Op:
    """float64 forward / adjoint products of a complex64 (or float32) matrix, 512 columns at a time (oracle side)"""

    def __init__(self, A, panel=512):
        self.A, self.panel = A, panel
        self.dtype = np.dtype(hi(A.dtype))
        self.shape = A.shape

    def mul(self, x):
        y = np.zeros(self.shape[0], self.dtype)
        for j in range(0, self.shape[1], self.panel):
            y += self.A[:, j:j + self.panel].astype(self.dtype) @ x[j:j + self.panel]
        return y

    def mul_adj(self, y):
        out = np.empty(self.shape[1], self.dtype)
        for j in range(0, self.shape[1], self.panel):
            # conj(conj(y) P) = P' y without materialising conj(P) (the same bits: tests/test_oracle.py checks the identity)
            out[j:j + self.panel] = np.conj(np.conj(y) @ self.A[:, j:j + self.panel].astype(self.dtype))
        return out


def test_golden_next_tier_on_device(rls, ctx):
    """the committed float64 fixtures of the SURVEY 8f solvers (tests/golden/next_tier_96x40_c64.npz)"""
    import os
    from test_oracle import _next_tier_solutions
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "next_tier_96x40_c64.npz"))
    got = _next_tier_solutions(rls, g, wrap=lambda a: rls.DeviceMatrix.from_host(np.asfortranarray(a)),
                               vec=lambda a: rls.DeviceVector.from_host(a))
    got32 = _next_tier_solutions(O, g, wrap=lambda a: a.astype(np.complex64), vec=lambda a: a.astype(np.complex64))
    for k, v in got.items():
        parity(f"golden_next_tier_{k}", v.to_host(), g[k], np.asarray(got32[k]))


# ---- singular-value thresholding prox maps ------------------------------------------


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
@pytest.mark.parametrize("m,n", [(32, 32), (64, 20), (20, 64), (7, 3), (1, 9), (96, 96)])
def test_prox_nuclear_matches_svd(rls, ctx, dt, m, n):
    """prox!(::NuclearRegularization) (ProxNuclear.jl:26-31): one-sided Jacobi on the device vs LAPACK svd"""
    rng = np.random.default_rng(61)
    X = rng.standard_normal((m, n))
    if np.dtype(dt).kind == "c":
        X = X + 1j * rng.standard_normal((m, n))
    X[:, 0] *= 10  # spread the singular values
    x = X.reshape(-1, order="F").astype(dt)
    lam = 0.3 * np.linalg.svd(X, compute_uv=False)[min(m, n) // 2]
    ref = O.prox_nuclear(x.astype(np.complex128 if np.dtype(dt).kind == "c" else np.float64), lam, (m, n))
    xd = rls.DeviceVector.from_host(x)
    rls.prox_(rls.NuclearRegularization(lam, svtShape=(m, n)), xd)
    # Float32 bound: the oracle (LAPACK svd) in the working precision
    parity(f"prox_nuclear_{m}x{n}_{np.dtype(dt).name}", xd.to_host(), ref, lambda: O.prox_nuclear(x.copy(), lam, (m, n)), record=False)
    # lambda above the largest singular value: everything is thresholded away
    xd = rls.DeviceVector.from_host(x)
    rls.prox_(rls.NuclearRegularization, xd, 1e6, svtShape=(m, n))
    assert np.all(xd.to_host() == 0)


@pytest.mark.parametrize("dt,shape,bs,K,shift", [(np.complex64, (32, 32), (4, 4), 10, None), (np.float32, (30, 29), (4, 4), 6, (1, 3)),
                                                 (np.complex64, (16, 16), (4, 4), 80, (4, 2)), (np.float32, (8, 8, 8), (4, 4, 4), 6, None),
                                                 (np.complex64, (12,), (5,), 4, (2,)), (np.complex64, (8, 8, 4), (4, 4, 2), 40, (1, 1, 1))])
def test_prox_llr_matches_oracle(rls, ctx, dt, shape, bs, K, shift):
    """proxLLRNonOverlapping! (ProxLLR.jl:43-88): distinct blocks, shifted grids, cut edge blocks, wide (K > voxels per
    block) and tall matrices, 1-D / 2-D / 3-D images"""
    rng = np.random.default_rng(67)
    n = int(np.prod(shape)) * K
    x = rng.standard_normal(n)
    if np.dtype(dt).kind == "c":
        x = x + 1j * rng.standard_normal(n)
    x = x.astype(dt)
    lam = 1.5
    ref = O.prox_llr(x.astype(np.complex128 if np.dtype(dt).kind == "c" else np.float64), lam, shape, bs, shift)
    reg = rls.LLRRegularization(lam, shape=shape, blockSize=bs, randshift=False)
    xd = rls.DeviceVector.from_host(x)
    if shift is None:
        rls.prox_(reg, xd)
    else:
        reg._call(xd, lam, list(shift))
    parity(f"prox_llr_{shape}_{np.dtype(dt).name}", xd.to_host(), ref, lambda: O.prox_llr(x.copy(), lam, shape, bs, shift), record=False)


def test_prox_llr_overlapping_randshift_and_denoising(rls, ctx):
    """fully overlapping blocks (ProxLLR.jl:165-203) against the oracle; randshift is a valid shift of the grid;
    the low-rank + noise experiment of test/testProxMaps.jl:194-219"""
    rng = np.random.default_rng(71)
    shape, bs, K = (10, 9), (4, 3), 5
    x = (rng.standard_normal(int(np.prod(shape)) * K) + 1j * rng.standard_normal(int(np.prod(shape)) * K)).astype(np.complex64)
    ref = O.prox_llr_overlapping(x.astype(np.complex128), 0.8, shape, bs)
    xd = rls.DeviceVector.from_host(x)
    rls.prox_(rls.LLRRegularization(0.8, shape=shape, blockSize=bs, fullyOverlapping=True), xd)
    parity("prox_llr_overlapping", xd.to_host(), ref, lambda: O.prox_llr_overlapping(x.copy(), 0.8, shape, bs), record=False)
    reg = rls.LLRRegularization(0.8, shape=shape, blockSize=bs, randshift=True, seed=5)
    xd = rls.DeviceVector.from_host(x)
    rls.prox_(reg, xd)
    cands = [((a, b), O.prox_llr(x.astype(np.complex128), 0.8, shape, bs, (a, b))) for a in range(1, 5) for b in range(1, 4)]
    sh_best, c_best = min(cands, key=lambda c: rel(xd.to_host(), c[1]))
    parity("prox_llr_randshift", xd.to_host(), c_best, lambda: O.prox_llr(x.copy(), 0.8, shape, bs, sh_best), record=False)
    # denoising: rank-2 image series + noise
    shp, sigma = (32, 32, 20), 0.05
    base = sum(np.einsum("i,j,k->ijk", rng.random(32), rng.random(32), rng.random(20)) for _ in range(2))
    base = base / base.max()
    noisy = (base + sigma * rng.standard_normal(shp)).astype(np.float32).reshape(-1, order="F")
    xd = rls.DeviceVector.from_host(noisy)
    rls.prox_(rls.LLRRegularization, xd, 10 * sigma, shape=(32, 32), blockSize=(4, 4), randshift=False)
    assert np.linalg.norm(xd.to_host() - base.reshape(-1, order="F")) < np.linalg.norm(noisy - base.reshape(-1, order="F"))


# ---- nested regularisation terms, ProjectionRegularization, plug-and-play prior (device) -------------------------


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
def test_nested_regularization_terms_on_device(rls, ctx, dt):
    """Masked / Transformed / FixedScaled / FixedParameter / AutoScaled terms: the device composition (rls_gather,
    rls_scatter, rls_stats + the inner prox kernels) against the oracle's restatement of the reference files"""
    rng = np.random.default_rng(77)
    n = 3000
    x = rng.standard_normal(n).astype(np.float32)
    if dt == np.complex64:
        x = (x + 1j * rng.standard_normal(n)).astype(np.complex64)
    x64 = x.astype(np.complex128 if dt == np.complex64 else np.float64)
    dev = lambda a: rls.DeviceVector.from_host(a)
    mask = rng.random(n) < 0.4
    # docstring example of the reference (MaskedRegularization.jl:8-17)
    m4 = rls.MaskedRegularization(rls.PositiveRegularization(), [True, False, True, False])
    assert np.array_equal(rls.prox_(m4, dev(np.full(4, -1, np.float32))).to_host(), [0, -1, 0, -1])
    for inner_d, inner_o in ((rls.L1Regularization(0.3), O.L1Regularization(0.3)), (rls.PositiveRegularization(), O.PositiveRegularization())):
        got = rls.prox_(rls.MaskedRegularization(inner_d, mask), dev(x)).to_host()
        ref = O.MaskedRegularization(inner_o, mask).prox(x64.copy())
        assert rel(got, ref) < 2e-6
        assert np.array_equal(got[~mask], x[~mask])  # untouched outside the mask, bit for bit
    assert np.isclose(rls.norm(rls.MaskedRegularization(rls.L1Regularization(0.3), mask), dev(x)),
                      O.MaskedRegularization(O.L1Regularization(0.3), mask).norm(x64), rtol=1e-5)
    l1d, l1o = rls.L1Regularization(0.5), O.L1Regularization(0.5)
    fs = rls.FixedScaledRegularization(l1d, 4.0)
    assert rls.lam(fs) == 2.0 and rls.sink(fs) is l1d and rls.sinktype(fs) is rls.L1Regularization
    assert rel(rls.prox_(fs, dev(x)).to_host(), O.prox_l1(x64.copy(), 2.0)) < 2e-6
    assert rel(rls.prox_(fs, dev(x), 0.25).to_host(), O.prox_l1(x64.copy(), 0.25)) < 2e-6
    fp = rls.FixedParameterRegularization(l1d)
    assert rel(rls.prox_(fp, dev(x), 123.0).to_host(), O.prox_l1(x64.copy(), 0.5)) < 2e-6
    au_d, au_o = rls.AutoScaledRegularization(l1d), O.AutoScaledRegularization(l1o)
    for _ in range(2):  # first call fixes the factor (maximum(abs.(x))), the second uses lambda as given
        assert rel(rls.prox_(au_d, dev(x), 0.05).to_host(), au_o.prox(x64.copy(), 0.05)) < 3e-6
    assert np.isclose(au_d.factor, au_o.factor, rtol=1e-6) and np.isclose(rls.lam(au_d), O.reg_lambda(au_o), rtol=1e-6)
    # TransformedRegularization with a dense unitary transform held as a DeviceMatrix
    k = 96
    Q = np.linalg.qr(rng.standard_normal((k, k)) + (1j * rng.standard_normal((k, k)) if dt == np.complex64 else 0))[0]
    Qd = rls.DeviceMatrix.from_host(np.asfortranarray(Q.astype(dt)))
    w = x[:k]
    tr_d, tr_o = rls.TransformedRegularization(rls.L1Regularization(0.2), Qd), O.TransformedRegularization(O.L1Regularization(0.2), Q)
    assert rel(rls.prox_(tr_d, dev(w)).to_host(), tr_o.prox(w.astype(x64.dtype))) < 2e-5
    assert np.isclose(rls.norm(tr_d, dev(w)), tr_o.norm(w.astype(x64.dtype)), rtol=1e-5)
    # ... and with the finite-difference operator: soft-thresholding of the gradient
    tg = rls.TransformedRegularization(rls.L1Regularization(0.1), rls.GradientOp((12, 8)))
    tg_o = O.TransformedRegularization(O.L1Regularization(0.1), O.GradientTrafo((12, 8)))
    assert rel(rls.prox_(tg, dev(w)).to_host(), tg_o.prox(w.astype(x64.dtype))) < 2e-5
    # nested chain: a normalised, masked L2 term is still found as THE L2 term of CGNR (findsink, src/CGNR.jl:69)
    A, xt, b = O.make_problem(64, 32, dt, 3)
    nested = rls.FixedScaledRegularization(rls.L2Regularization(0.05), 2.0)
    S = rls.createLinearSolver(rls.CGNR, rls.DeviceMatrix.from_host(A), reg=[nested, rls.RealRegularization()], iterations=20)
    ref = O.CGNR(A, reg=[O.L2Regularization(0.1), O.RealRegularization()], iterations=20)
    assert rel(rls.solve_(S, dev(b)).to_host(), O.solve(ref, b)) < 5e-5


def test_projection_regularization_reference_test(rls, ctx):
    """testProj (test/testProxMaps.jl:153-164) with projFunc = x -> real(x) written for device vectors"""
    rng = np.random.default_rng(1234)
    x = (rng.standard_normal(1012) + 1j * rng.standard_normal(1012)).astype(np.complex64)
    proj = lambda v: rls.prox_(rls.RealRegularization, v.copy())
    xp = rls.prox_(rls.ProjectionRegularization, rls.DeviceVector.from_host(x), projFunc=proj).to_host()
    assert np.linalg.norm(xp - x.real) / np.linalg.norm(x.real) < 1e-4
    n_after = rls.norm(rls.ProjectionRegularization, rls.DeviceVector.from_host(xp), projFunc=proj)
    n_before = rls.norm(rls.ProjectionRegularization, rls.DeviceVector.from_host(x), projFunc=proj)
    assert n_after == 0.0 and n_before == float("inf")
    assert 0.5 * np.linalg.norm(x - xp) ** 2 + n_after <= n_before
    # as a solver constraint: sorted with the projections through its sink type
    A, xt, b = O.make_problem(48, 24, np.complex64, 4)
    S = rls.createLinearSolver(rls.FISTA, rls.DeviceMatrix.from_host(A), reg=[rls.L1Regularization(0.01), rls.ProjectionRegularization(proj)],
                               rho=0.5 / np.linalg.norm(A, 2) ** 2, iterations=12)
    ref = O.FISTA(A, reg=[O.L1Regularization(0.01), O.RealRegularization()], rho=0.5 / np.linalg.norm(A, 2) ** 2, iterations=12)
    assert rel(rls.solve_(S, rls.DeviceVector.from_host(b)).to_host(), O.solve(ref, b)) < 5e-5


def test_pnp_regularization_reference_tests_on_device(rls, ctx):
    """test/testRegularization.jl:1-79 on device vectors (constructor, compatibility with Kaczmarz and ADMM, prox on
    real / complex input, lambda clipping with the reference's warning) + the input transforms against the oracle"""
    ident = lambda v: v
    pnp = rls.PnPRegularization(ident, [2])
    assert pnp.lam == 1.0 and pnp.model is ident and pnp.shape == [2]
    assert pnp.input_transform is rls.MinMaxTransform and pnp.ignoreIm is False
    pnp = rls.PnPRegularization(0.1, model=ident, shape=[2], input_transform=lambda v: v, ignoreIm=True, sMtHeLsE=1)
    assert pnp.ignoreIm is True
    rng = np.random.default_rng(5)
    A = rng.random((3, 2)).astype(np.float32)
    xs = rng.random(2).astype(np.float32)
    for solver in (rls.Kaczmarz, rls.ADMM):  # "PnP Compatibility"
        S = rls.createLinearSolver(solver, rls.DeviceMatrix.from_host(np.asfortranarray(A)), iterations=2, reg=[rls.PnPRegularization(ident, [2])])
        assert np.all(np.isfinite(rls.solve_(S, rls.DeviceVector.from_host(A @ xs)).to_host()))
    zero = lambda v: v.similar().fill_(0)
    f32 = np.float32
    pnp = rls.PnPRegularization(0.1, model=zero, shape=[2], input_transform=rls.IdentityTransform)
    out = rls.prox_(pnp, rls.DeviceVector.from_host(np.array([1, 2], f32)), 0.1).to_host()
    assert np.allclose(out, [0.9, 1.8], rtol=2e-7)
    out = rls.prox_(pnp, rls.DeviceVector.from_host(np.array([1 + 1j, 2 + 2j], np.complex64)), 0.1).to_host()
    assert np.allclose(out.real, [0.9, 1.8], rtol=2e-7) and np.allclose(out.imag, [0.9, 1.8], rtol=2e-7)
    pnp_i = rls.PnPRegularization(0.1, model=zero, shape=[2], input_transform=rls.IdentityTransform, ignoreIm=True)
    out = rls.prox_(pnp_i, rls.DeviceVector.from_host(np.array([1 + 1j, 2 + 2j], np.complex64)), 0.1).to_host()
    assert np.allclose(out.real, [0.9, 1.8], rtol=2e-7) and np.array_equal(out.imag, [1.0, 2.0])
    with pytest.warns(UserWarning, match=r"was given λ with value 1.5. Valid range is \[0, 1\]. λ changed to temp"):
        out = rls.prox_(pnp, rls.DeviceVector.from_host(np.array([1, 2], f32)), 1.5).to_host()
    assert np.array_equal(out, [0.0, 0.0])
    with pytest.warns(UserWarning, match="-1.5"):
        out = rls.prox_(pnp, rls.DeviceVector.from_host(np.array([1, 2], f32)), -1.5).to_host()
    assert np.array_equal(out, [1.0, 2.0])
    # a non-trivial model (3-point smoothing, done with device BLAS-1 on shifted views is overkill: scale by 1/2)
    # and every input transform, against the oracle
    x = rng.standard_normal(5000).astype(f32) * 3 + 1
    half_d = lambda v: v.copy().rmul_(0.5)
    half_o = lambda a: 0.5 * a
    cases = [(rls.MinMaxTransform, O.MinMaxTransform), (rls.ZTransform, O.ZTransform), (rls.IdentityTransform, O.IdentityTransform),
             (lambda v: rls.ClampedScalingTransform(v, -2.0, 4.0), lambda a: O.ClampedScalingTransform(a, -2.0, 4.0))]
    for tf_d, tf_o in cases:
        got = rls.prox_(rls.PnPRegularization(0.3, model=half_d, shape=[50, 100], input_transform=tf_d), rls.DeviceVector.from_host(x), 0.3).to_host()
        ref = O.PnPRegularization(0.3, model=half_o, shape=[50, 100], input_transform=tf_o).prox(x.astype(np.float64), 0.3)
        assert rel(got, ref) < 5e-6
    st = rls.DeviceVector.from_host(x).stats()
    assert st[0] == x.min() and st[1] == x.max() and np.isclose(st[2], x.astype(np.float64).sum(), rtol=1e-12)
    assert np.isclose(st[3], (x.astype(np.float64) ** 2).sum(), rtol=1e-12) and st[4] == np.abs(x).max()


def test_library_is_reentrant_across_host_threads(rls, ctx):
    """SURVEY 8b "Threading": solve! may be entered concurrently from several host threads (src/MultiThreading.jl:71,
    docs multi_threading.jl:14-17), each with its own rls_ctx (own stream, own workspace).  Four threads run CGNR,
    FISTA (+L1), ADMM (+TV: FGP single-workgroup kernel, device plan) and a batched solve side by side, repeatedly;
    every result must equal, bit for bit, what the same solver produced alone."""
    import threading

    A, xt, b = O.make_problem(512, 256, np.complex64, 31)
    Ar, xr, br = O.make_problem(384, 144, np.float32, 32)
    _, X, B = O.make_problem(512, 256, np.complex64, 31, n_rhs=16)
    rho = 0.9 / np.linalg.norm(A.astype(np.complex128), 2) ** 2

    def job_cgnr(c):
        S = rls.createLinearSolver(rls.CGNR, rls.DeviceMatrix.from_host(A, c), reg=rls.L2Regularization(1e-3), iterations=24)
        return rls.solve_(S, rls.DeviceVector.from_host(b, c)).to_host()

    def job_fista(c):
        S = rls.createLinearSolver(rls.FISTA, rls.DeviceMatrix.from_host(A, c), reg=rls.L1Regularization(0.02), rho=rho, iterations=30)
        return rls.solve_(S, rls.DeviceVector.from_host(b, c)).to_host()

    def job_admm(c):
        S = rls.createLinearSolver(rls.ADMM, rls.DeviceMatrix.from_host(Ar, c), reg=rls.TVRegularization(0.02, shape=(12, 12)),
                                   rho=0.2, iterations=8, iterationsCG=6)
        return rls.solve_(S, rls.DeviceVector.from_host(br, c)).to_host()

    def job_batched(c):
        S = rls.createLinearSolver(rls.CGNR, rls.DeviceMatrix.from_host(A, c), iterations=16, relTol=0.0)
        xs = rls.solve_(S, rls.DeviceMatrix.from_host(np.asfortranarray(B), c), scheduler=rls.BatchedState)
        return np.stack([x.to_host() for x in xs], axis=1)

    jobs = [job_cgnr, job_fista, job_admm, job_batched]
    alone = [j(rls.Context(0)) for j in jobs]
    results, errors = [None] * len(jobs), []

    def worker(k):
        try:
            c = rls.Context(0)
            outs = [jobs[k](c) for _ in range(6)]
            results[k] = outs
        except Exception as e:  # surfaced below; a thread must not die silently
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    assert not errors, errors
    for k, outs in enumerate(results):
        assert outs is not None, f"job {k} did not finish"
        for o in outs:
            assert np.array_equal(o, alone[k]), f"job {k} differs under concurrency"


@pytest.mark.parametrize("dt,M,N,K,kind", [(np.complex64, 256, 128, 5, "l1"), (np.float32, 320, 96, 20, "l1pos"),
                                           (np.complex64, 4096, 2048, 16, "l1"), (np.float32, 128, 64, 3, "l21")])
def test_fista_batched_matrix_solve_equals_column_solves(rls, ctx, dt, M, N, K, kind):
    """solve!(solver::FISTA, B) with K columns sharing A on the matrix cores (rls_fista_*_batched) against K
    independent oracle solves and against the MultiThreadingState scheduler; per-column scalars and retirement
    (columns stop at different iterations under a loose relTol), gradient restart, padding columns (K % 16 != 0)"""
    A, X, B = O.make_problem(M, N, dt, 41, n_rhs=K)
    dt64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    A64 = A.astype(dt64)
    rho = 0.9 / np.linalg.norm(A64, 2) ** 2 if M < 1000 else 0.9 / (np.sqrt(M) + np.sqrt(N)) ** 2
    B = (B * (4.0 ** (np.arange(K) % 6))[None, :]).astype(dt)  # different scales against a fixed lambda: different stopping iterations
    lam = 0.02 * np.max(np.abs(A64.conj().T @ B[:, 0].astype(dt64)))
    its, relTol, restart = 25, (0.02 if M < 1000 else 0.0), "gradient" if kind == "l1pos" else "none"

    def regs(R):
        if kind == "l1":
            return R.L1Regularization(lam)
        if kind == "l1pos":
            return [R.L1Regularization(lam), R.PositiveRegularization()]
        return R.L21Regularization(lam, slices=4)

    Ad, Bd = rls.DeviceMatrix.from_host(A), rls.DeviceMatrix.from_host(np.asfortranarray(B))
    S = rls.createLinearSolver(rls.FISTA, Ad, reg=regs(rls), rho=rho, iterations=its, relTol=relTol, restart=restart)
    xs = rls.solve_(S, Bd, scheduler=rls.BatchedState)
    assert type(S.state).__name__ == "FistaBatchedState"
    stat = S.state.status()
    its_seen, refs = [], {}
    # the reference's own scheduler must give the same columns: both are gated against the oracle
    S2 = rls.createLinearSolver(rls.FISTA, Ad, reg=regs(rls), rho=rho, iterations=its, relTol=relTol, restart=restart)
    ys = rls.solve_(S2, Bd, scheduler=rls.MultiThreadingState)
    for j in range(K if M < 1000 else 3):
        ref = O.FISTA(A64, reg=regs(O), rho=rho, iterations=its, relTol=relTol, restart=restart)
        O.solve(ref, B[:, j].astype(dt64))
        assert stat[j].iteration == ref.iteration, (j, stat[j].iteration, ref.iteration)
        # Float32 bound: the oracle in the working precision, stopped at the same iteration
        x32 = lambda: O.solve(O.FISTA(A, reg=regs(O), rho=rho, iterations=ref.iteration, relTol=0.0, restart=restart), np.ascontiguousarray(B[:, j]))
        tag = f"fista_batched_{kind}_{M}x{N}_{np.dtype(dt).name}_K{K}_col{j}"
        parity(tag, xs[j].to_host(), ref.x, x32, record=(j < 2))
        parity(tag + "_threads", ys[j].to_host(), ref.x, x32, record=False)
        its_seen.append(ref.iteration)
        refs[j] = (ref.x, x32)
    if kind == "l1" and M < 1000:
        assert len(set(its_seen)) > 1  # the columns really did retire at different iterations
    # iterating with callbacks reaches the same result; a vector solve still works afterwards
    seen = []
    zs = rls.solve_(S, Bd, scheduler=rls.BatchedState, callbacks=lambda s_, it: seen.append(it))
    assert seen[0] == 0 and all(np.array_equal(z.to_host(), x.to_host()) for z, x in zip(zs, xs))
    x1 = rls.solve_(S, rls.DeviceVector.from_host(B[:, 1].copy())).to_host()
    parity("fista_vector_after_batched", x1, refs[1][0], refs[1][1], record=False)


@pytest.mark.parametrize("name", ["OptISTA", "POGM", "POGM-restart"])
def test_optista_pogm_deferred_run_equals_stepwise(rls, ctx, name):
    """without callbacks every iteration is enqueued at once and `rel_res_norm < relTol` is decided on the device
    (rls_*_update_async): same stopping iteration as the oracle, same bits as the iteration-by-iteration run, and the
    state (iteration, theta, rel_res_norm, x/y roles) is consistent afterwards so that a second solve works"""
    A, xt, b = O.make_problem(300, 100, np.complex64, 37)
    A64, b64 = A.astype(np.complex128), b.astype(np.complex128)
    rho = 0.9 / np.linalg.norm(A64, 2) ** 2
    lam = 1e-2 * np.max(np.abs(A64.conj().T @ b64))
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    extra = {}
    if name == "POGM-restart":  # theta / sigma / gamma and the restart decision on the device (rls_pogm_update_auto)
        name, extra = "POGM", dict(restart="gradient", sigma_fac=0.96)
    for relTol, its in ((0.0, 21), (3e-2, 60)):
        ref = getattr(O, name)(A64, reg=O.L1Regularization(lam), rho=rho, iterations=its, relTol=relTol, **extra)
        O.solve(ref, b64)
        sol = rls.createLinearSolver(getattr(rls, name), Ad, reg=rls.L1Regularization(lam), rho=rho, iterations=its, relTol=relTol,
                                     **extra)
        for _ in range(2):
            x = rls.solve_(sol, bd).to_host()
            assert sol.state.iteration == ref.iteration, (relTol, sol.state.iteration, ref.iteration)
            parity(f"{name}{'_restart' if extra else ''}_deferred_reltol{relTol}", x, ref.x,
                   lambda: O.solve(getattr(O, name)(A, reg=O.L1Regularization(lam), rho=rho, iterations=ref.iteration, relTol=0.0, **extra), b),
                   record=False)
            assert np.isclose(sol.state.rel_res_norm, ref.rel_res_norm, rtol=2e-3)
        if relTol > 0:
            assert 1 < ref.iteration < its
        th_deferred = sol.state.theta  # the theta recurrence is index-only arithmetic: identical on host and device
        seen = []
        x_cb = rls.solve_(sol, bd, callbacks=lambda s_, it: seen.append(it)).to_host()
        assert seen == list(range(ref.iteration + 1))
        if extra:  # two different kernels evaluate the same update: the compiler may fuse a*b + c*d either way round
            assert rel(x_cb, x) < 2e-6 and sol.state.theta == th_deferred
        elif getattr(sol, "_pgm", (None, None))[1] is not None:  # resident launches sum AHA x in another order
            assert rel(x_cb, x) < 2e-5
        else:
            assert np.array_equal(x_cb, x)


@pytest.mark.parametrize("dt,M,N,kind", [(np.float32, 128, 64, "tv"), (np.complex64, 120, 48, "l1"), (np.float32, 96, 40, "l1pos")])
def test_split_bregman_blocks_on_the_device_plan(rls, ctx, dt, M, N, kind):
    """SplitBregman (src/SplitBregman.jl:204-271): each block of inner iterations runs through rls_admm_step (prox
    threshold lambda / rho, block length iterationsInner, `converged` on the device), the Bregman update stays on
    the host: equal to the per-call sequence and to the oracle, in one go and iteration by iteration"""
    A, xt, b = O.make_problem(M, N, dt, 43)
    def regs(R):
        if kind == "tv":
            return R.TVRegularization(2e-2, shape=(8, 8))
        if kind == "l1":
            return R.L1Regularization(0.05)
        return [R.L1Regularization(0.05), R.PositiveRegularization()]
    kw = dict(rho=0.3, iterations=3, iterationsInner=4, iterationsCG=5, tolInner=1e-4)
    ref = O.SplitBregman(A, reg=regs(O), **kw)
    O.solve(ref, b)
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    sol = rls.createLinearSolver(rls.SplitBregman, Ad, reg=regs(rls), **kw)
    x = rls.solve_(sol, bd).to_host()
    assert sol.state._plan_ok
    old = rls.createLinearSolver(rls.SplitBregman, Ad, reg=regs(rls), **kw)
    old.use_device_plan = False
    x_old = rls.solve_(old, bd).to_host()
    assert not old.state._plan_ok
    assert (sol.state.iter_cnt, sol.state.iteration) == (old.state.iter_cnt, old.state.iteration) == (ref.iter_cnt, ref.iteration)
    ref64 = O.SplitBregman(A.astype(hi(dt)), reg=regs(O), **kw)
    parity(f"splitbregman_plan_{kind}_{M}x{N}", x, O.solve(ref64, b.astype(hi(dt))), ref.x, record=False)
    assert rel(x, x_old) < 3e-6
    assert np.allclose(sol.state.rk, old.state.rk, rtol=1e-4) and np.allclose(sol.state.sk, old.state.sk, rtol=1e-4)
    seen = []
    x_cb = rls.solve_(sol, bd, callbacks=lambda s_, it: seen.append(it)).to_host()
    assert seen == list(range(len(seen))) and len(seen) - 1 == kw["iterations"] * kw["iterationsInner"]
    assert np.array_equal(x_cb, x)


def test_random_shapes_against_oracle(rls, ctx):
    """shape fuzz across every alignment boundary of the kernels (rows per 16-byte chunk, 16-row slabs, 16-column MFMA
    tiles, columns per load round, N < one round, M < one slab, M or N = 1): GEMV N / C, the normal operator, a few
    CGNR and FISTA iterations, each against the float64 oracle.  Deterministic seed; run it under RLS_GUARD_ALLOC=1
    to turn any read past the end of a buffer into a fault."""
    rng = np.random.default_rng(20260101)
    shapes = [(1, 1), (2, 1), (1, 3), (3, 2), (15, 17), (16, 16), (17, 15), (33, 7), (64, 129), (130, 64), (255, 33),
              (256, 128), (257, 127), (48, 260), (500, 3), (3, 500)]
    shapes += [(int(rng.integers(1, 700)), int(rng.integers(1, 330))) for _ in range(22)]
    for k, (M, N) in enumerate(shapes):
        dt = np.complex64 if k % 2 == 0 else np.float32
        dt64 = np.complex128 if dt == np.complex64 else np.float64
        A = rng.standard_normal((M, N)).astype(np.float32)
        if dt == np.complex64:
            A = (A + 1j * rng.standard_normal((M, N))).astype(np.complex64)
        A = np.asfortranarray(A)
        x = rng.standard_normal(N).astype(dt)
        y = rng.standard_normal(M).astype(dt)
        A64 = A.astype(dt64)
        Ad = rls.DeviceMatrix.from_host(A)
        xd, yd = rls.DeviceVector.from_host(x), rls.DeviceVector.from_host(y)
        tag = (M, N, np.dtype(dt).name)
        # single products of random vectors can cancel: the Float32 bound is NumPy's own product in the working precision
        parity(f"fuzz_gemv_n_{tag}", Ad.mul_(rls.DeviceVector(M, dt, ctx), xd).to_host(), A64 @ x, lambda: A @ x.astype(dt), record=False)
        parity(f"fuzz_gemv_c_{tag}", Ad.mul_adj_(rls.DeviceVector(N, dt, ctx), yd).to_host(), A64.conj().T @ y,
               lambda: A.conj().T @ y.astype(dt), record=False)
        op = rls.OperatorHandle(Ad)
        parity(f"fuzz_normal_{tag}", op.mul_normal_(rls.DeviceVector(N, dt, ctx), xd).to_host(), A64.conj().T @ (A64 @ x),
               lambda: A.conj().T @ (A @ x.astype(dt)), record=False)
        b = (A64 @ rng.standard_normal(N)).astype(dt)
        its = min(4, N, max(M - 1, 1))  # CG past the rank of A only amplifies Float32 rounding
        ref = O.CGNR(A64, reg=O.L2Regularization(0.1), iterations=its, relTol=0.0)
        O.solve(ref, b.astype(dt64))
        S = rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(0.1), iterations=its, relTol=0.0)
        parity(f"fuzz_cgnr_{tag}", rls.solve_(S, rls.DeviceVector.from_host(b)).to_host(), ref.x,
               lambda: O.solve(O.CGNR(A, reg=O.L2Regularization(0.1), iterations=its, relTol=0.0), b), record=False)
        rho = 0.9 / max(np.linalg.norm(A64, 2) ** 2, 1e-30)
        reff = O.FISTA(A64, reg=O.L1Regularization(0.05), rho=rho, iterations=4, relTol=0.0)
        O.solve(reff, b.astype(dt64))
        Sf = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(0.05), rho=rho, iterations=4, relTol=0.0)
        got = rls.solve_(Sf, rls.DeviceVector.from_host(b)).to_host()
        parity(f"fuzz_fista_{tag}", got, reff.x, lambda: O.solve(O.FISTA(A, reg=O.L1Regularization(0.05), rho=rho, iterations=4, relTol=0.0), b),
               record=False, scale=max(np.linalg.norm(reff.x), 1e-3))


def test_random_shapes_other_paths_against_oracle(rls, ctx):
    """the same fuzz for the paths around the matrix-free loop: Gram GEMM + Gram-mode CGNR, the shared-A batched
    solve (matrix cores for multiples of 16, register slab / column plans otherwise), Kaczmarz sweeps, the
    elementwise prox maps and the TV prox on odd image shapes"""
    rng = np.random.default_rng(20260102)
    shapes = [(16, 16), (32, 48), (47, 31), (64, 64), (65, 63), (128, 16), (129, 17), (208, 112), (300, 37), (5, 9)]
    shapes += [(int(rng.integers(2, 400)), int(rng.integers(2, 200))) for _ in range(8)]
    for k, (M, N) in enumerate(shapes):
        dt = np.complex64 if k % 2 == 0 else np.float32
        dt64 = np.complex128 if dt == np.complex64 else np.float64
        A = rng.standard_normal((M, N)).astype(np.float32)
        if dt == np.complex64:
            A = (A + 1j * rng.standard_normal((M, N))).astype(np.complex64)
        A = np.asfortranarray(A)
        A64 = A.astype(dt64)
        Ad = rls.DeviceMatrix.from_host(A)
        tag = (M, N, np.dtype(dt).name)
        G = Ad.gram()
        Gh = G.to_host()
        parity(f"fuzz_gram_{tag}", Gh, A64.conj().T @ A64, lambda: A.conj().T @ A, record=False)
        assert np.array_equal(Gh, Gh.conj().T), tag  # Hermitian bit for bit
        b = (A64 @ rng.standard_normal(N)).astype(dt)
        its = min(4, N, max(M - 1, 1))
        ref = O.CGNR(A64, reg=O.L2Regularization(0.2), iterations=its, relTol=0.0, normal="gram")
        O.solve(ref, b.astype(dt64))
        S = rls.createLinearSolver(rls.CGNR, Ad, AHA=G, reg=rls.L2Regularization(0.2), iterations=its, relTol=0.0)
        parity(f"fuzz_cgnr_gram_{tag}", rls.solve_(S, rls.DeviceVector.from_host(b)).to_host(), ref.x,
               lambda: O.solve(O.CGNR(A, reg=O.L2Regularization(0.2), iterations=its, relTol=0.0, normal="gram"), b), record=False)
        K = int(rng.integers(2, 21))
        B = np.asfortranarray((A64 @ rng.standard_normal((N, K))).astype(dt))
        Sb = rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(0.2), iterations=its, relTol=0.0)
        xs = rls.solve_(Sb, rls.DeviceMatrix.from_host(B), scheduler=rls.BatchedState)
        for j in (0, K - 1):
            refj = O.CGNR(A64, reg=O.L2Regularization(0.2), iterations=its, relTol=0.0)
            parity(f"fuzz_batched_{tag}_{K}_{j}", xs[j].to_host(), O.solve(refj, B[:, j].astype(dt64)),
                   lambda: O.solve(O.CGNR(A, reg=O.L2Regularization(0.2), iterations=its, relTol=0.0), np.ascontiguousarray(B[:, j])), record=False)
        refk = O.Kaczmarz(A64, reg=O.L2Regularization(0.05), iterations=2)
        xk = O.solve(refk, b.astype(dt64))
        Sk = rls.createLinearSolver(rls.Kaczmarz, Ad, reg=rls.L2Regularization(0.05), iterations=2)
        parity(f"fuzz_kaczmarz_{tag}", rls.solve_(Sk, rls.DeviceVector.from_host(b)).to_host(), xk,
               lambda: O.solve(O.Kaczmarz(A, reg=O.L2Regularization(0.05), iterations=2), b), record=False)
        # prox maps on a vector of length M * N' (odd lengths included)
        n = int(rng.integers(1, 3000))
        v = rng.standard_normal(n).astype(np.float32)
        if dt == np.complex64:
            v = (v + 1j * rng.standard_normal(n)).astype(np.complex64)
        v64 = v.astype(dt64)
        assert rel(rls.prox_(rls.L1Regularization, rls.DeviceVector.from_host(v), 0.3).to_host(), O.prox_l1(v64.copy(), 0.3)) < 2e-6, (tag, n)
        assert rel(rls.prox_(rls.L2Regularization, rls.DeviceVector.from_host(v), 0.3).to_host(), O.prox_l2(v64.copy(), 0.3)) < 2e-6, (tag, n)
        assert np.array_equal(rls.prox_(rls.PositiveRegularization, rls.DeviceVector.from_host(v)).to_host(),
                              O.prox_positive(v.copy())), (tag, n)
        nx, ny = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        img = rng.standard_normal(nx * ny).astype(np.float32)
        if dt == np.complex64:
            img = (img + 1j * rng.standard_normal(nx * ny)).astype(np.complex64)
        if nx * ny > 1 and (nx > 1 or ny > 1):
            want = O.prox_tv_fgp(img.astype(dt64), 0.2, (nx, ny), None, 6)
            got = rls.prox_(rls.TVRegularization(0.2, shape=(nx, ny), iterationsTV=6), rls.DeviceVector.from_host(img), 0.2).to_host()
            assert rel(got, want) < 5e-6, (tag, nx, ny)


def test_random_shapes_solver_plans_against_oracle(rls, ctx):
    """fuzz of the device plans: ADMM (+L1 / +TV on an odd image shape), SplitBregman, OptISTA / POGM deferred runs and
    the batched FISTA over shapes that do and do not qualify for the fused paths"""
    rng = np.random.default_rng(20260103)
    cases = [(64, 48, (8, 6)), (130, 35, (7, 5)), (256, 128, (16, 8)), (96, 81, (9, 9)), (75, 20, (20,)), (513, 144, (12, 12))]
    for k, (M, N, shape) in enumerate(cases):
        dt = np.float32 if k % 2 == 0 else np.complex64
        dt64 = np.complex128 if dt == np.complex64 else np.float64
        A, xt, b = O.make_problem(M, N, dt, 100 + k)
        A64, b64 = A.astype(dt64), b.astype(dt64)
        Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
        tag = (M, N, np.dtype(dt).name)
        kw = dict(rho=0.25, iterations=4, iterationsCG=4, tolInner=1e-4)
        for mk in (lambda R: R.L1Regularization(0.03), lambda R: R.TVRegularization(0.03, shape=shape)):
            ref = O.ADMM(A, reg=mk(O), **kw)
            O.solve(ref, b)
            S = rls.createLinearSolver(rls.ADMM, Ad, reg=mk(rls), **kw)
            r64 = O.ADMM(A64, reg=mk(O), **kw)
            parity(f"fuzz_admm_{tag}", rls.solve_(S, bd).to_host(), O.solve(r64, b64), ref.x, record=False)
            assert S.state._plan_ok and S.state.iteration == ref.iteration, tag
        refb = O.SplitBregman(A, reg=O.L1Regularization(0.03), rho=0.25, iterations=2, iterationsInner=3, iterationsCG=4, tolInner=1e-4)
        O.solve(refb, b)
        Sb = rls.createLinearSolver(rls.SplitBregman, Ad, reg=rls.L1Regularization(0.03), rho=0.25, iterations=2, iterationsInner=3,
                                    iterationsCG=4, tolInner=1e-4)
        r64 = O.SplitBregman(A64, reg=O.L1Regularization(0.03), rho=0.25, iterations=2, iterationsInner=3, iterationsCG=4, tolInner=1e-4)
        parity(f"fuzz_splitbregman_{tag}", rls.solve_(Sb, bd).to_host(), O.solve(r64, b64), refb.x, record=False)
        rho = 0.9 / np.linalg.norm(A64, 2) ** 2
        lam = 1e-2 * np.max(np.abs(A64.conj().T @ b64))
        for name in ("OptISTA", "POGM"):
            refp = getattr(O, name)(A64, reg=O.L1Regularization(lam), rho=rho, iterations=9)
            O.solve(refp, b64)
            Sp = rls.createLinearSolver(getattr(rls, name), Ad, reg=rls.L1Regularization(lam), rho=rho, iterations=9)
            parity(f"fuzz_{name}_{tag}", rls.solve_(Sp, bd).to_host(), refp.x,
                   lambda: O.solve(getattr(O, name)(A, reg=O.L1Regularization(lam), rho=rho, iterations=9), b), record=False)
        K = int(rng.integers(2, 19))
        B = np.asfortranarray((A64 @ rng.standard_normal((N, K))).astype(dt))
        Sf = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(lam), rho=rho, iterations=8, relTol=0.0)
        xs = rls.solve_(Sf, rls.DeviceMatrix.from_host(B), scheduler=rls.BatchedState)
        for j in (0, K - 1):
            reff = O.FISTA(A64, reg=O.L1Regularization(lam), rho=rho, iterations=8, relTol=0.0)
            parity(f"fuzz_fista_batched_{tag}_{K}_{j}", xs[j].to_host(), O.solve(reff, B[:, j].astype(dt64)),
                   lambda: O.solve(O.FISTA(A, reg=O.L1Regularization(lam), rho=rho, iterations=8, relTol=0.0), np.ascontiguousarray(B[:, j])), record=False)


def test_plan_entry_points_reject_misuse(rls, ctx):
    """error behaviour of the plan-level C ABI: calls out of order return RLS_E_STATE, bad arguments RLS_E_INVALID,
    shapes / regularisers a fused plan does not cover RLS_E_UNSUPPORTED (so that the host falls back to the per-call
    sequence) -- each with a message in rls_last_error_string, never a crash and never a silent no-op"""
    import ctypes as C
    from rls_amd._lib import AdmmParams, AdmmStatus, FistaStatus
    lib, h = ctx.lib, ctx.handle
    A, _, b = O.make_problem(48, 24, np.float32, 2)   # N = 24: not a multiple of 16
    Ad = rls.DeviceMatrix.from_host(A)
    op = rls.OperatorHandle(Ad)
    v = [rls.DeviceVector(24, np.float32, ctx) for _ in range(8)]
    cg = C.c_void_p()
    assert lib.rls_cg_create(op.handle, v[0].ptr, v[1].ptr, v[2].ptr, C.byref(cg)) == 0
    plan = C.c_void_p()
    assert lib.rls_admm_create(cg, C.byref(plan)) == 0
    assert lib.rls_admm_step(plan, 1) == -4 and b"admm_init" in lib.rls_last_error_string(h)       # RLS_E_STATE
    st = AdmmStatus()
    assert lib.rls_admm_get_status(plan, C.byref(st), None, 0) == -4
    P = AdmmParams()
    assert lib.rls_admm_init(plan, C.byref(P)) == -1                                                # null vectors
    P.x, P.xold, P.beta, P.beta_y, P.z0, P.z1, P.u = (t.ptr for t in v[:7])
    P.rho, P.iterations, P.iterations_cg, P.reg_kind = 0.1, 3, 2, 3                                  # REG_L21: not fused
    assert lib.rls_admm_init(plan, C.byref(P)) == -2
    P.reg_kind, P.proj_kind = 4, 2                                                                   # TV + projection
    P.tv_ndims, P.tv_ntv, P.tv_iterations = 2, 2, 5
    P.tv_shape[0], P.tv_shape[1], P.tv_dims[0], P.tv_dims[1] = 6, 4, 0, 1
    assert lib.rls_admm_init(plan, C.byref(P)) == -2
    P.proj_kind = 0
    P.tv_shape[0] = 5                                                                                # 5 * 4 != N
    assert lib.rls_admm_init(plan, C.byref(P)) == -1
    P.tv_shape[0] = 6
    assert lib.rls_admm_init(plan, C.byref(P)) == 0 and lib.rls_admm_step(plan, -1) == -1
    assert lib.rls_admm_destroy(plan) == 0 and lib.rls_cg_destroy(cg) == 0
    # batched FISTA needs the matrix-core shape (M, N multiples of 16): the host then uses per-column plans
    fp = C.c_void_p()
    X = [rls.DeviceMatrix(24, 4, np.float32, ctx) for _ in range(4)]
    assert lib.rls_fista_create_batched(op.handle, 4, X[0].ptr, X[1].ptr, X[2].ptr, X[3].ptr, 24, C.byref(fp)) == -2
    S = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(0.01), rho=1e-3, iterations=3)
    B = np.asfortranarray(np.stack([b, 2 * b], axis=1))
    xs = rls.solve_(S, rls.DeviceMatrix.from_host(B), scheduler=rls.BatchedState)
    assert type(S.state).__name__ == "MultiThreadingState" and len(xs) == 2
    # single-column plan: the batched calls are refused
    f1 = C.c_void_p()
    assert lib.rls_fista_create(op.handle, v[0].ptr, v[1].ptr, v[2].ptr, v[3].ptr, C.byref(f1)) == 0
    fs = (FistaStatus * 1)()
    assert lib.rls_fista_get_status_batched(f1, fs) == -4
    assert lib.rls_fista_step_local_a(f1) == -4
    assert lib.rls_fista_destroy(f1) == 0
    # nested-term helpers
    d = (C.c_double * 5)()
    assert lib.rls_stats(h, 0, 0, v[0].ptr, d) == -1 and lib.rls_gather(h, 0, 4, None, v[0].ptr, v[1].ptr) == -1
    assert lib.rls_optista_update_async(h, 0, 24, v[0].ptr, v[1].ptr, v[2].ptr, v[3].ptr, v[4].ptr, v[5].ptr, 0.1, 1, 0.1,
                                        1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0, None) == -1


@pytest.mark.parametrize("mode", [1, 2])
def test_pipeline_pair_hint_is_only_a_hint(rls, ctx, mode):
    """the 2-launch CGNR / cg! pipeline is told by the host which (r, p) buffer pair is current so that it loads one
    pair instead of both; the kernel checks the hint against the device scalars: with the hint withheld (mode 1) or
    deliberately inverted (mode 2: every launch takes the check-and-reload path) the results are bit-identical"""
    A, xt, b = O.make_problem(4096, 2048, np.complex64, 2)
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    Ar, xr, br = O.make_problem(256, 128, np.float32, 3)
    Ard, brd = rls.DeviceMatrix.from_host(Ar), rls.DeviceVector.from_host(br)
    Gd = Ad.gram()

    def run():
        S = rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(1e-3), iterations=37, relTol=0.0)
        x1 = rls.solve_(S, bd).to_host()
        T = rls.createLinearSolver(rls.ADMM, Ard, reg=rls.L1Regularization(0.02), rho=0.3, iterations=4, iterationsCG=7, tolInner=1e-6)
        x2 = rls.solve_(T, brd).to_host()
        F = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=0.9 / (64 + 45.3) ** 2, iterations=35, relTol=0.0)
        x3 = rls.solve_(F, bd).to_host()
        rls.init_(F, bd)   # iteration by iteration: every call starts at another parity of the iteration count
        for _ in range(5):
            rls.iterate(F)
        x4 = F.state.x.to_host() if hasattr(F.state, "x") else None
        Fg = rls.createLinearSolver(rls.FISTA, Ad, AHA=Gd, reg=rls.L1Regularization(1e-2), rho=0.9 / (64 + 45.3) ** 2,
                                    iterations=35, relTol=0.0)   # the one-launch Gram-mode kernel takes the same hint
        x5 = rls.solve_(Fg, bd).to_host()
        rls.init_(Fg, bd)
        for _ in range(5):
            rls.iterate(Fg)
        x6 = Fg.state.x.to_host()
        return x1, x2, S.state.iteration, x3, x4, x5, x6

    want = run()
    ctx.tune(pipe_hint_mode=mode)
    try:
        got = run()
    finally:
        ctx.tune(pipe_hint_mode=0)
    assert got[2] == want[2] == 37
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(got[3], want[3]) and np.array_equal(got[4], want[4])
    assert np.array_equal(got[5], want[5]) and np.array_equal(got[6], want[6])


def test_random_shapes_svt_prox_and_batched_kaczmarz(rls, ctx):
    """fuzz of the singular-value thresholding prox maps (random image shapes, block sizes, shifts, series lengths;
    tall, wide and rank-deficient blocks) and of the batched Kaczmarz sweep (one workgroup per right-hand side)"""
    rng = np.random.default_rng(20260104)
    for k in range(14):
        dt = np.complex64 if k % 2 == 0 else np.float32
        dt64 = np.complex128 if dt == np.complex64 else np.float64
        nd = int(rng.integers(1, 4))
        shape = tuple(int(rng.integers(2, 12)) for _ in range(nd))
        bs = tuple(int(rng.integers(1, min(5, s) + 1)) for s in shape)
        K = int(rng.integers(1, 9))
        shift = tuple(int(rng.integers(0, b)) for b in bs)
        n = int(np.prod(shape)) * K
        x = rng.standard_normal(n)
        if dt == np.complex64:
            x = x + 1j * rng.standard_normal(n)
        if k % 5 == 0:  # rank-deficient blocks: repeat one frame
            X = x.reshape(shape + (K,), order="F")
            X[..., 1:] = X[..., :1]
            x = X.reshape(-1, order="F")
        x = x.astype(dt)
        ref = O.prox_llr(x.astype(dt64), 0.7, shape, bs, shift)
        reg = rls.LLRRegularization(0.7, shape=shape, blockSize=bs, randshift=False)
        xd = rls.DeviceVector.from_host(x)
        reg._call(xd, 0.7, list(shift))
        parity(f"fuzz_llr_{shape}_{bs}_{K}_{shift}_{np.dtype(dt).name}", xd.to_host(), ref, lambda: O.prox_llr(x.copy(), 0.7, shape, bs, shift),
               record=False)
        m, c = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        y = rng.standard_normal(m * c).astype(np.float32)
        if dt == np.complex64:
            y = (y + 1j * rng.standard_normal(m * c)).astype(np.complex64)
        refn = O.prox_nuclear(y.astype(dt64), 0.9, (m, c))
        got = rls.prox_(rls.NuclearRegularization, rls.DeviceVector.from_host(y), 0.9, svtShape=(m, c)).to_host()
        parity(f"fuzz_nuclear_{m}x{c}_{np.dtype(dt).name}", got, refn, lambda: O.prox_nuclear(y.copy(), 0.9, (m, c)), record=False,
               scale=max(np.linalg.norm(refn), 1e-2))
    for k in range(6):
        dt = np.complex64 if k % 2 == 0 else np.float32
        dt64 = np.complex128 if dt == np.complex64 else np.float64
        M, N, K = int(rng.integers(2, 200)), int(rng.integers(1, 300)), int(rng.integers(2, 12))
        A = rng.standard_normal((M, N)).astype(np.float32)
        if dt == np.complex64:
            A = (A + 1j * rng.standard_normal((M, N))).astype(np.complex64)
        A = np.asfortranarray(A)
        B = np.asfortranarray((A.astype(dt64) @ rng.standard_normal((N, K))).astype(dt))
        S = rls.createLinearSolver(rls.Kaczmarz, rls.DeviceMatrix.from_host(A), reg=rls.L2Regularization(0.02), iterations=2)
        xs = rls.solve_(S, rls.DeviceMatrix.from_host(B), scheduler=rls.BatchedState)
        for j in (0, K - 1):
            refk = O.Kaczmarz(A.astype(dt64), reg=O.L2Regularization(0.02), iterations=2)
            parity(f"fuzz_kaczmarz_batched_{M}x{N}_{K}_{j}", xs[j].to_host(), O.solve(refk, B[:, j].astype(dt64)),
                   lambda: O.solve(O.Kaczmarz(A, reg=O.L2Regularization(0.02), iterations=2), np.ascontiguousarray(B[:, j])), record=False)


# ---- resident CGNR: the whole step call in one launch, A in registers across iterations ----------------------


def _hold_cus(rls, ctx, n_workgroups, microseconds):
    """the CU-parking kernel of the co-tenancy tests: a TEST hook in its own shared object (csrc/test_hooks.hip), not a symbol
    of the product library"""
    import ctypes as C
    import os

    lib = C.CDLL(os.path.join(os.path.dirname(rls.LIB_PATH), "librls_test_hooks.so"))
    lib.rls_test_hold_cus.restype = C.c_int32
    lib.rls_test_hold_cus.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    ctx.lib.rls_ctx_stream.restype = C.c_void_p
    return lib.rls_test_hold_cus(ctx.lib.rls_ctx_stream(ctx.handle), 0, int(n_workgroups), int(microseconds))


def _resident_unavailable():
    """A resident kernel needs one workgroup per CU on up to 256 CUs.  On the device these tests are written for (MI355X: 256
    CUs) a plan that is not resident is a REGRESSION of the headline kernel's eligibility, not a reason to skip its parity gate;
    only a smaller device may skip."""
    import torch

    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if cus >= 256:
        pytest.fail(f"resident mode not available on a {cus}-CU device: the headline kernels' parity gate would not run")
    pytest.skip(f"resident kernels need 256 CUs; this device has {cus}")


def _fista_path(ctx, sol):
    import ctypes as C
    out = C.c_int32(-1)
    assert ctx.lib.rls_fista_path(sol.state._plan, C.byref(out)) == 0
    return out.value


def _cgnr_path(rls, sol):
    import ctypes as C
    out = C.c_int32(-1)
    assert rls.load().rls_cgnr_path(sol.state._plan, C.byref(out)) == 0
    return out.value


@pytest.mark.parametrize("dt,M,N", [(np.complex64, 4096, 2048), (np.float32, 2048, 4096), (np.complex64, 3000, 1502)])
def test_resident_partial_rows_at_l2_change_no_bit(rls, ctx, dt, M, N):
    """the matrix-free resident kernels keep the partial rows of their in-kernel all-reduce in the XCD's L2 when every workgroup
    sits on the XCD its group assumes (csrc/normal.hip, resident_rows_at_l2; checked per launch): same loads, same summation
    orders -- CGNR, FISTA + L1 and POGM + L1 solves must equal the write-through protocol (`resident_l2_rows = 0`) bit for bit,
    and the float64 oracle within the gate"""
    A, xt, b = O.make_problem(M, N, dt, 83)
    A64, b64 = A.astype(hi(dt)), b.astype(hi(dt))
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    rho = 0.95 / np.linalg.norm(A64, 2) ** 2
    lam = 1e-2 * float(np.max(np.abs(A64.conj().T @ b64)))
    makers = {
        "cgnr": lambda R, A_: (R.CGNR(A_, iterations=24, relTol=0.0) if R is O else R.createLinearSolver(R.CGNR, A_, iterations=24, relTol=0.0)),
        "fista": lambda R, A_: (R.FISTA(A_, reg=R.L1Regularization(lam), rho=rho, iterations=24, relTol=0.0) if R is O
                                else R.createLinearSolver(R.FISTA, A_, reg=R.L1Regularization(lam), rho=rho, iterations=24, relTol=0.0)),
        "pogm": lambda R, A_: (R.POGM(A_, reg=R.L1Regularization(lam), rho=rho, iterations=24, relTol=0.0) if R is O
                               else R.createLinearSolver(R.POGM, A_, reg=R.L1Regularization(lam), rho=rho, iterations=24, relTol=0.0)),
    }
    for name, make in makers.items():
        got = {}
        for l2 in (1, 0):
            ctx.tune(resident_l2_rows=l2)
            try:
                S = make(rls, Ad)
                got[l2] = rls.solve_(S, bd).to_host()
                got[(l2, "again")] = rls.solve_(S, bd).to_host()
            finally:
                ctx.tune(resident_l2_rows=1)
        assert np.array_equal(got[1], got[0]) and np.array_equal(got[1], got[(1, "again")]), (name, rel(got[1], got[0]))
        x64 = O.solve(make(O, A64), b64)
        parity(f"resident_l2_rows_{name}_{M}x{N}_{np.dtype(dt).name}", got[1], x64, lambda: O.solve(make(O, A), b), record=False)


@pytest.mark.parametrize("dt,M,N,lam", [(np.complex64, 4096, 2048, 0.0), (np.complex64, 4096, 2048, 1e-2), (np.float32, 4096, 4096, 0.0),
                                       (np.float32, 2048, 4096, 1e-3),
                                       # ragged M and N: the masked instantiation (N in (NMAX / 2, NMAX], fewer workgroups than CUs)
                                       (np.complex64, 4000, 2000, 1e-3), (np.complex64, 3000, 1502, 0.0), (np.float32, 4000, 2200, 0.0),
                                       (np.complex64, 2048, 1024, 0.0), (np.complex64, 1800, 900, 1e-3), (np.float32, 3000, 1600, 0.0)])
def test_cgnr_resident_kernel(rls, ctx, dt, M, N, lam):
    """rls_cgnr_step as ONE launch (cgnr_resident_kernel: A held in registers, two in-kernel grid exchanges per
    iteration): iterates against the float64 oracle at iterations 1 / 5 / 10 / 32, bit-identical run to run and
    identical whether the iterations are enqueued one launch each or all in one launch; relTol stops it at the
    oracle's iteration; the two-launch pipeline (resident = 0) agrees within the gate"""
    A, xt, b = O.make_problem(M, N, dt, 77)
    iters = 32
    ref = O.CGNR(A.astype(hi(dt)), reg=O.L2Regularization(lam), iterations=iters, relTol=0.0)
    ref32 = O.CGNR(A, reg=O.L2Regularization(lam), iterations=iters, relTol=0.0)
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    sol = rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(lam), iterations=iters, relTol=0.0)
    rls.init_(sol, bd)
    if _cgnr_path(rls, sol) != 4:
        _resident_unavailable()
    ref.init(b.astype(hi(dt)))
    ref32.init(b)
    r0 = np.linalg.norm(ref.A.mul_adj(b.astype(hi(dt))))
    tag = f"cgnr_resident_{M}x{N}_{np.dtype(dt).name}_lam{lam}"
    for it in range(1, iters + 1):
        assert ref.iterate() is not None and ref32.iterate() is not None and rls.iterate(sol) is not None
        if it in (1, 5, 10, 32):
            st = sol.state
            parity(f"{tag}_x_it{it}", st.x.to_host(), ref.x, ref32.x)
            parity(f"{tag}_r_it{it}", st.x0.to_host(), ref.r, ref32.r, scale=r0)
            parity(f"{tag}_p_it{it}", st.pl.to_host(), ref.p, ref32.p, scale=r0)
    assert rls.iterate(sol) is None and sol.state.iteration == iters
    # (a step call of ONE iteration -- what iterate() issues -- takes the two-launch pipeline since round 3: the gate above
    #  covered that path; the resident kernel proper is what the calls below reach)
    x_once = rls.solve_(sol, bd).to_host()      # all 32 iterations in ONE launch
    x_again = rls.solve_(sol, bd).to_host()
    parity(f"{tag}_one_launch", x_once, ref.x, ref32.x)
    rls.init_(sol, bd)
    for _ in range(4):                          # the same 32 iterations as four launches of 8
        assert ctx.lib.rls_cgnr_step(sol.state._plan, 8) == 0
    sol.state._refresh(ctx.lib)
    x_steps = sol.state.x.to_host()
    assert sol.state.iteration == iters
    assert np.array_equal(x_once, x_steps) and np.array_equal(x_once, x_again)
    # relTol: stops inside the launch, at the oracle's iteration
    tol = 1e-3
    ref2 = O.CGNR(A.astype(hi(dt)), reg=O.L2Regularization(lam), iterations=iters, relTol=tol)
    O.solve(ref2, b.astype(hi(dt)))
    sol2 = rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(lam), iterations=iters, relTol=tol)
    x2 = rls.solve_(sol2, bd).to_host()
    assert abs(sol2.state.iteration - ref2.iteration) <= 1  # (the square Float32 case never reaches the tolerance: both run out)
    if M > N:
        assert 1 < ref2.iteration < iters
    if sol2.state.iteration == ref2.iteration:
        parity(f"{tag}_reltol", x2, ref2.x, lambda: O.solve(O.CGNR(A, reg=O.L2Regularization(lam), iterations=ref2.iteration, relTol=0.0), b))
    # the two-launch pipeline of the same plan
    ctx.tune(resident=0)
    try:
        assert _cgnr_path(rls, sol) == 1
        x_pipe = rls.solve_(sol, bd).to_host()
    finally:
        ctx.tune(resident=1)
    parity(f"{tag}_pipeline", x_pipe, ref.x, ref32.x)


def _fresh_resident_ctx(ctx):
    """forget earlier timeouts (a context that lost two resident launches stops using the resident kernels)"""
    ctx.tune(resident=1)


def test_cgnr_resident_timeout_falls_back_to_the_pipeline(rls, ctx):
    """every in-kernel wait is bounded: with the bound forced to ONE poll workgroup 0 gives up and the launch changes
    nothing.  The next status read re-runs the lost iterations on the two-launch pipeline: no exception, the iteration
    count and the solution are those of an undisturbed run, `fallbacks` counts the event, and a lost launch stays
    visible (sticky) even when a later resident launch of the same plan succeeds before the status is read"""
    import ctypes as C
    A, xt, b = O.make_problem(4096, 2048, np.complex64, 78)
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    A64, b64 = A.astype(np.complex128), b.astype(np.complex128)
    ref = {n: O.solve(O.CGNR(A64, iterations=n, relTol=0.0), b64) for n in (4, 8)}
    _fresh_resident_ctx(ctx)
    try:
        for lost_then_ok in (False, True):
            sol = rls.createLinearSolver(rls.CGNR, Ad, iterations=8, relTol=0.0)
            rls.init_(sol, bd)
            if _cgnr_path(rls, sol) != 4:
                _resident_unavailable()
            ctx.tune(resident_spin=1)
            assert ctx.lib.rls_cgnr_step(sol.state._plan, 4) == 0       # lost: a no-op
            ctx.tune(resident_spin=100000)
            if lost_then_ok:
                assert ctx.lib.rls_cgnr_step(sol.state._plan, 4) == 0   # runs, and clears the per-launch flags
            st = sol.state._refresh(ctx.lib)
            want = 8 if lost_then_ok else 4
            assert st.iteration == want and st.fallbacks >= 1
            assert _cgnr_path(rls, sol) == 1                             # the plan stays on the pipeline
            parity(f"cgnr_resident_fallback_{want}", sol.state.x.to_host(), ref[want],
                   lambda: O.solve(O.CGNR(A, iterations=want, relTol=0.0), b))
            _fresh_resident_ctx(ctx)
    finally:
        ctx.tune(resident_spin=100000)
        _fresh_resident_ctx(ctx)
    sol = rls.createLinearSolver(rls.CGNR, Ad, iterations=8, relTol=0.0)
    x = rls.solve_(sol, bd).to_host()
    assert _cgnr_path(rls, sol) == 4 and sol.state._refresh(ctx.lib).fallbacks == 0
    parity("cgnr_resident_after_timeout", x, ref[8], lambda: O.solve(O.CGNR(A, iterations=8, relTol=0.0), b))


@pytest.mark.parametrize("dt,M,N,lam,iters", [(np.float32, 256, 128, 1e-2, 10), (np.float32, 32, 16, 1e-4, 16), (np.complex64, 64, 32, 0.0, 12),
                                              (np.complex64, 250, 61, 1e-3, 9), (np.float32, 37, 5, 0.0, 5), (np.float32, 500, 60, 1e-2, 8),
                                              (np.complex64, 1, 1, 0.0, 1)])
def test_cgnr_small_system_kernel(rls, ctx, dt, M, N, lam, iters):
    """Systems that fit ONE CU's register file (BASELINE configs[0] 256 x 128 Float32; the reference's own test and documentation
    sizes, test/testSolvers.jl:3-43, docs/src/literate/howto/gpu_acceleration.jl:12-23: 32 x 16) run a whole rls_cgnr_step call as a
    single-workgroup launch (csrc/small.hip, path 8).  Iterates against the float64 oracle step by step, one n-step call = n
    one-step calls bit for bit, relTol retirement, and agreement with the per-iteration pipeline (small = 0)."""
    A, xt, b = O.make_problem(M, N, dt, 61)
    dt64 = hi(dt)
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    reg = rls.L2Regularization(lam)
    ref = O.CGNR(A.astype(dt64), reg=O.L2Regularization(lam), iterations=iters, relTol=0.0)
    ref32 = O.CGNR(A, reg=O.L2Regularization(lam), iterations=iters, relTol=0.0)
    ref.init(b.astype(dt64)); ref32.init(b)
    sol = rls.createLinearSolver(rls.CGNR, Ad, reg=reg, iterations=iters, relTol=0.0)
    rls.init_(sol, bd)
    assert _cgnr_path(rls, sol) == 8
    tag = f"cgnr_small_{M}x{N}_{np.dtype(dt).name}"
    r0 = max(float(np.linalg.norm(ref.r)), 1e-30)
    n_it = min(iters, N)
    for it in range(1, n_it + 1):
        assert ref.iterate() is not None and ref32.iterate() is not None and rls.iterate(sol) is not None
        if it in (1, 2, 5, n_it):
            parity(f"{tag}_x_it{it}", sol.state.x.to_host(), ref.x, ref32.x)
            parity(f"{tag}_r_it{it}", sol.state.x0.to_host(), ref.r, ref32.r, scale=r0)
    assert rls.iterate(sol) is None and sol.state.iteration == n_it
    x_steps = sol.state.x.to_host()
    x_once = rls.solve_(sol, bd).to_host()  # all iterations in ONE launch
    assert np.array_equal(x_once, x_steps)
    ctx.tune(small=0)
    try:
        sol0 = rls.createLinearSolver(rls.CGNR, Ad, reg=reg, iterations=iters, relTol=0.0)
        x_pipe = rls.solve_(sol0, bd).to_host()
        assert _cgnr_path(rls, sol0) != 8
    finally:
        ctx.tune(small=1)
    assert rel(x_once, x_pipe) < 1e-5
    if N > 4:  # early retirement on relTol: the oracle's iteration count (+-1 at the threshold)
        ref2 = O.CGNR(A.astype(dt64), reg=O.L2Regularization(lam), iterations=iters, relTol=1e-2)
        O.solve(ref2, b.astype(dt64))
        sol2 = rls.createLinearSolver(rls.CGNR, Ad, reg=reg, iterations=iters, relTol=1e-2)
        rls.solve_(sol2, bd)
        assert abs(sol2.state.iteration - ref2.iteration) <= 1


@pytest.mark.parametrize("restart", ["none", "gradient"])
@pytest.mark.parametrize("dt,M,N,kind", [(np.float32, 256, 128, "l1"), (np.float32, 32, 16, "l1"), (np.complex64, 64, 32, "l1"),
                                         (np.complex64, 250, 61, "l1pos"), (np.float32, 37, 5, "l2"), (np.float32, 500, 60, "none")])
def test_fista_small_system_kernel(rls, ctx, dt, M, N, kind, restart):
    """FISTA (src/FISTA.jl:139-185) on systems that fit ONE CU's register file: a whole rls_fista_step call as a single-workgroup
    launch (fista_small_kernel, path 8).  Iterates against the float64 oracle step by step (x, xold, res, theta, the residual
    norm), one n-step call = n one-step calls bit for bit, the stopping test inside a launch, agreement with the slab pipeline
    (small = 0), and a second solve on the same plan."""
    import ctypes as C
    A, xt, b = O.make_problem(M, N, dt, 67)
    dt64 = hi(dt)
    A64, b64 = A.astype(dt64), b.astype(dt64)
    rho = 0.9 / np.linalg.norm(A64, 2) ** 2
    lam = 2e-2 * float(np.max(np.abs(A64.conj().T @ b64)))
    def regs(R):
        return {"l1": R.L1Regularization(lam), "l2": R.L2Regularization(lam), "none": None,
                "l1pos": [R.L1Regularization(lam), R.PositiveRegularization()]}[kind]
    iters = 25
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    ref = O.FISTA(A64, reg=regs(O), rho=rho, iterations=iters, relTol=0.0, restart=restart)
    ref32 = O.FISTA(A, reg=regs(O), rho=rho, iterations=iters, relTol=0.0, restart=restart)
    ref.init(b64); ref32.init(b)
    sol = rls.createLinearSolver(rls.FISTA, Ad, reg=regs(rls), rho=rho, iterations=iters, relTol=0.0, restart=restart)
    rls.init_(sol, bd)
    path = C.c_int32(-1)
    assert ctx.lib.rls_fista_path(sol.state._plan, C.byref(path)) == 0 and path.value == 8
    tag = f"fista_small_{M}x{N}_{np.dtype(dt).name}_{kind}_{restart}"
    for it in range(1, iters + 1):
        assert ref.iterate() is not None and ref32.iterate() is not None and rls.iterate(sol) is not None
        if it in (1, 2, 7, iters):
            st = sol.state
            parity(f"{tag}_x_it{it}", st.x.to_host(), ref.x, ref32.x, record=it == iters)
            parity(f"{tag}_xold_it{it}", st.xold.to_host(), ref.xold, ref32.xold, record=False)
            parity(f"{tag}_res_it{it}", st.res.to_host(), ref.res, ref32.res, scale=float(np.linalg.norm(ref.x0)), record=False)
            assert abs(st.rel_res_norm - ref.rel_res_norm) < 1e-4 * ref.rel_res_norm + 2e-6   # (a converged residual is Float32 noise)
    assert rls.iterate(sol) is None and sol.state.iteration == iters
    x_steps = sol.state.x.to_host()
    x_once = rls.solve_(sol, bd).to_host()   # all iterations in ONE launch
    assert np.array_equal(x_once, x_steps)
    assert np.array_equal(rls.solve_(sol, bd).to_host(), x_once)
    ctx.tune(small=0)
    try:
        sol0 = rls.createLinearSolver(rls.FISTA, Ad, reg=regs(rls), rho=rho, iterations=iters, relTol=0.0, restart=restart)
        x_pipe = rls.solve_(sol0, bd).to_host()
        assert ctx.lib.rls_fista_path(sol0.state._plan, C.byref(path)) == 0 and path.value != 8
    finally:
        ctx.tune(small=1)
    assert rel(x_once, x_pipe) < 1e-5
    # the stopping test inside the launch: the oracle's iteration count (+-1 at the threshold)
    rr = []
    probe = O.FISTA(A64, reg=regs(O), rho=rho, iterations=iters, relTol=0.0, restart=restart)
    probe.init(b64)
    while probe.iterate() is not None:
        rr.append(probe.rel_res_norm)
    tol = float(np.sqrt(rr[5] * rr[6])) if rr[6] < rr[5] else None
    if tol is not None and min(rr[:6]) > tol:
        ref2 = O.FISTA(A64, reg=regs(O), rho=rho, iterations=iters, relTol=tol, restart=restart)
        O.solve(ref2, b64)
        sol2 = rls.createLinearSolver(rls.FISTA, Ad, reg=regs(rls), rho=rho, iterations=iters, relTol=tol, restart=restart)
        x2 = rls.solve_(sol2, bd).to_host()
        assert sol2.state.iteration == ref2.iteration == 7
        parity(f"{tag}_reltol", x2, ref2.x, lambda: O.solve(O.FISTA(A, reg=regs(O), rho=rho, iterations=7, relTol=0.0, restart=restart), b), record=False)
        assert np.array_equal(rls.solve_(sol2, bd).to_host(), x2)


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
def test_small_problem_group_one_launch(rls, ctx, dt):
    """The reference's other multi-solve flavour -- one solver AND one matrix per problem (docs/src/literate/howto/multi_threading.jl:8-17)
    -- for problems that each fit one CU: solve_group_ runs init! and every iteration of all K problems as ONE launch, one workgroup
    per problem (rls_cgnr_init_step_group).  30 problems of differing shapes (two launches: 24 + 6) against the float64 oracle and
    against the solo solves; the group step entry point continues a started group; a group that does not qualify falls back."""
    import ctypes as C
    rng = np.random.default_rng(7)
    K = 30
    cplx = np.dtype(dt).kind == "c"   # (tall systems: a random square one is too ill-conditioned for a 1e-5 gate after 12 iterations)
    shapes = [(lambda N: (int(rng.integers(2 * N + 8, 129 if cplx else 257)), N))(int(rng.integers(4, 33 if cplx else 65))) for _ in range(K)]
    probs = [O.make_problem(M, N, dt, 200 + k) for k, (M, N) in enumerate(shapes)]
    lam, iters = 1e-2, 12
    mats = [rls.DeviceMatrix.from_host(A) for A, _, _ in probs]
    rhs = [rls.DeviceVector.from_host(b) for _, _, b in probs]
    make = lambda Ad: rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(lam), iterations=iters, relTol=0.0)
    group = [make(Ad) for Ad in mats]
    xs = rls.solve_group_(group, rhs)
    for k, ((A, xt, b), x) in enumerate(zip(probs, xs)):
        ref = O.CGNR(A.astype(hi(dt)), reg=O.L2Regularization(lam), iterations=iters, relTol=0.0)
        O.solve(ref, b.astype(hi(dt)))
        assert group[k].state.iteration == ref.iteration == min(iters, shapes[k][1])
        parity(f"small_group_{np.dtype(dt).name}_{k}", x.to_host(), ref.x,
               lambda A=A, b=b: O.solve(O.CGNR(A, reg=O.L2Regularization(lam), iterations=iters, relTol=0.0), b), record=k < 2)
        solo = rls.solve_(make(mats[k]), rhs[k]).to_host()
        assert rel(x.to_host(), solo) < 1e-4   # (the group runs on the tile of its largest member: another summation order)
    # a started group continues through the group step entry point: init + 5 iterations, then 7 more
    plans = (C.c_void_p * K)()
    bptr = (C.c_void_p * K)(*[b.ptr for b in rhs])
    again = [make(Ad) for Ad in mats]
    for k, (s_, b) in enumerate(zip(again, rhs)):
        s_._prepare(s_.state, b)
        plans[k] = s_.state._plan
    L = rls._lib
    L.check(ctx.handle, ctx.lib.rls_cgnr_init_step_group(plans, bptr, K, lam, 0.0, iters, 5), "init_step_group")
    L.check(ctx.handle, ctx.lib.rls_cgnr_step_group(plans, K, 7), "step_group")
    for k, s_ in enumerate(again):
        assert np.array_equal(s_.state.x.to_host(), xs[k].to_host()), k   # 5 + 7 in two launches == 12 in one
    # relTol differs between the solvers: not one launch -- the solves still come out
    odd = [make(mats[0]), rls.createLinearSolver(rls.CGNR, mats[1], reg=rls.L2Regularization(lam), iterations=iters, relTol=1e-3)]
    ys = rls.solve_group_(odd, rhs[:2])
    assert np.array_equal(ys[0].to_host(), rls.solve_(make(mats[0]), rhs[0]).to_host())


def test_distinct_operator_solve_queue(rls, ctx):
    """BASELINE configs[3], distinct-A flavour (docs/src/literate/howto/multi_threading.jl:8-17: a solver AND an operator per task):
    `count` problems of DIFFERENT shapes and kernel paths -- register-resident, two-launch pipeline, two-GEMV, small-system -- solved
    as ONE queue on the context's stream (rls_cgnr_solve_queue / _host through solve_group_): per problem against the float64 oracle
    (1e-5), against its solo solve (bit for bit: the same kernels, only the host between them is gone), iteration counts, a second
    call on the cached plans with other right-hand sides, host arrays in and out, and a lost resident launch inside the queue."""
    shapes = [(1024, 2048), (4096, 2048), (768, 640), (1001, 500), (96, 24), (1024, 2048)]   # (an odd M: the two-GEMV path)
    lam, iters = 1e-3, 14
    probs = [O.make_problem(M, N, np.complex64, 900 + k) for k, (M, N) in enumerate(shapes)]
    mats = [rls.DeviceMatrix.from_host(A, ctx) for A, _, _ in probs]
    make = lambda Ad: rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(lam), iterations=iters, relTol=0.0)
    group = [make(Ad) for Ad in mats]
    rhs = [rls.DeviceVector.from_host(b, ctx) for _, _, b in probs]
    xs = [x.to_host() for x in rls.solve_group_(group, rhs)]
    paths = [_cgnr_path(rls, s_) for s_ in group]
    assert 4 in paths and len(set(paths)) >= 3, paths   # the queue mixes kernel families
    for k, ((A, xt, b), x) in enumerate(zip(probs, xs)):
        ref = O.CGNR(A.astype(np.complex128), reg=O.L2Regularization(lam), iterations=iters, relTol=0.0)
        O.solve(ref, b.astype(np.complex128))
        assert group[k].state.iteration == ref.iteration == min(iters, shapes[k][1])
        parity(f"solve_queue_{k}_{shapes[k][0]}x{shapes[k][1]}", x, ref.x,
               lambda A=A, b=b: O.solve(O.CGNR(A, reg=O.L2Regularization(lam), iterations=iters, relTol=0.0), b), record=k < 3)
        assert np.array_equal(x, rls.solve_(make(mats[k]), rhs[k]).to_host()), k
    # host arrays in, host arrays out, on the SAME solvers (cached plans), other right-hand sides
    rng = np.random.default_rng(5)
    rhs2 = [(A @ (rng.standard_normal(A.shape[1]) + 1j * rng.standard_normal(A.shape[1]))).astype(np.complex64) for A, _, _ in probs]
    ys = rls.solve_group_(group, rhs2)
    assert all(isinstance(y, np.ndarray) for y in ys)
    for k, (A, _, _) in enumerate(probs):
        ref = O.CGNR(A.astype(np.complex128), reg=O.L2Regularization(lam), iterations=iters, relTol=0.0)
        O.solve(ref, rhs2[k].astype(np.complex128))
        assert rel(ys[k], ref.x) < 1e-5, (k, rel(ys[k], ref.x))
        assert np.array_equal(ys[k], rls.solve_(make(mats[k]), rls.DeviceVector.from_host(rhs2[k], ctx)).to_host()), k
    # constraints act at exit (src/CGNR.jl:145-147), also behind a queue with host arrays
    con = [rls.createLinearSolver(rls.CGNR, Ad, reg=[rls.L2Regularization(lam), rls.RealRegularization()], iterations=iters, relTol=0.0)
           for Ad in mats[:2]]
    zs = rls.solve_group_(con, rhs2[:2])
    assert all(np.all(z.imag == 0) for z in zs) and rel(zs[0].real, ys[0].real) < 1e-6
    # a resident launch lost inside the queue (wait bound of one poll): re-run on the pipeline for that problem, same result to 1e-5
    _fresh_resident_ctx(ctx)
    try:
        g2 = [make(Ad) for Ad in mats[:3]]
        for s_, b in zip(g2, rhs[:3]):
            s_._prepare(s_.state, b)
        ctx.tune(resident_spin=1)
        ws = rls.solve_group_(g2, [b for _, _, b in probs[:3]])
        ctx.tune(resident_spin=100000)
        assert sum(s_.state.fallbacks for s_ in g2) >= 1 and [s_.state.iteration for s_ in g2] == [iters] * 3
        for k in range(3):
            assert rel(ws[k], xs[k]) < 1e-5
    finally:
        ctx.tune(resident_spin=100000)
        _fresh_resident_ctx(ctx)


def test_batched_gram_resident_lost_launch_is_recovered(rls, ctx):
    """the batched resident launch (csrc/gramk.hip) under the same contract as the single-column ones: with the wait bound forced
    to one poll the launch gives up having changed nothing (only workgroup 0 writes the caller's state, after its last
    barrier), the status call re-runs the missing iterations on the streaming kernels (path 6), reports `fallbacks`, and the
    columns are those of an undisturbed solve"""
    M, N, K = 4096, 2048, 8
    A, X, B = O.make_problem(M, N, np.complex64, 31, n_rhs=K)
    B = np.asfortranarray(B)
    Ad = rls.DeviceMatrix.from_host(A)
    Gd, Bd = Ad.gram(), rls.DeviceMatrix.from_host(B)
    _fresh_resident_ctx(ctx)
    try:
        S = rls.createLinearSolver(rls.CGNR, Ad, AHA=Gd, iterations=10, relTol=0.0)
        rls.init_(S, Bd, scheduler=rls.BatchedState)
        assert _cgnr_path(rls, S) == 7
        ctx.tune(resident_spin=1)
        S.state._step(10)  # lost: a no-op
        ctx.tune(resident_spin=100000)
        stat = S.state.status()
        assert [s_.iteration for s_ in stat] == [10] * K and all(s_.fallbacks >= 1 for s_ in stat)
        assert _cgnr_path(rls, S) == 6  # the plan stays on the streaming kernels
        for j in (0, K - 1):
            x64, x32 = oracle_pair(lambda A_, b_: O.solve(O.CGNR(A_, iterations=10, relTol=0.0, normal="gram"), b_), A, np.ascontiguousarray(B[:, j]))
            parity(f"batched_gram_resident_recovered_col{j}", S.state.solutions()[j].to_host(), x64, x32)
    finally:
        ctx.tune(resident_spin=100000)
        _fresh_resident_ctx(ctx)


@pytest.mark.parametrize("M,N,K,kind", [(4096, 2048, 8, "l1")] + [(m, n, k, kind) for (m, n, k) in [(1040, 208, 7), (2000, 1936, 5), (320, 16, 2)]
                                                                  for kind in ("l1", "l2", "none")])
def test_batched_fista_gram_resident_launch(rls, ctx, M, N, K, kind):
    """batched FISTA on the explicit Gram matrix as ONE resident launch per step call (csrc/gramk.hip,
    fista_gramk_resident_kernel; rls_fista_path 7): every workgroup advances ITS 8 rows of every column and the rows of the
    next extrapolated point are what is exchanged.  Against the float64 oracle's Gram-mode FISTA per column
    (src/FISTA.jl:141-189, src/MultiThreading.jl:30-79), against the streaming kernels of the same plan (path 3), with
    per-column relTol retirement, and split into several step calls (same bits as one call)."""
    A, X, B = O.make_problem(M, N, np.complex64, 53, n_rhs=K)
    B = np.asfortranarray(B)
    B[:, 0] *= 1e-3
    A64 = A.astype(np.complex128)
    smooth = A64 @ (A64.conj().T @ (A64 @ (A64.conj().T @ B[:, K - 1].astype(np.complex128))))  # in the span of the leading singular vectors:
    B[:, K - 1] = (smooth / np.linalg.norm(smooth) * np.linalg.norm(B[:, 1])).astype(np.complex64)  # this column retires earlier
    Ad = rls.DeviceMatrix.from_host(A)
    Gd, Bd = Ad.gram(), rls.DeviceMatrix.from_host(B)
    rho = float(0.9 / np.linalg.norm(A64, 2) ** 2)
    lam = 1e-3 * float(np.abs(A64.conj().T @ B[:, 1]).max())
    reg = lambda R: {"l1": R.L1Regularization(lam), "l2": R.L2Regularization(lam), "none": R.L2Regularization(0.0)}[kind]
    iters = 15
    tag = f"batched_fista_gramk_{M}x{N}_K{K}_{kind}"
    _fresh_resident_ctx(ctx)
    try:
        for relTol in (0.0, 2e-2):
            F = rls.createLinearSolver(rls.FISTA, Ad, AHA=Gd, reg=reg(rls), rho=rho, iterations=iters, relTol=relTol)
            xs = [x.to_host() for x in rls.solve_(F, Bd, scheduler=rls.BatchedState)]
            assert _fista_path(ctx, F) == 7, _fista_path(ctx, F)
            stat = F.state.status()
            assert all(s_.fallbacks == 0 for s_ in stat)
            ctx.tune(resident=0)
            F0 = rls.createLinearSolver(rls.FISTA, Ad, AHA=Gd, reg=reg(rls), rho=rho, iterations=iters, relTol=relTol)
            x0s = [x.to_host() for x in rls.solve_(F0, Bd, scheduler=rls.BatchedState)]
            assert _fista_path(ctx, F0) == 3
            stat0 = F0.state.status()
            ctx.tune(resident=1)
            for j in range(K):
                ref = O.FISTA(A64, reg=reg(O), rho=rho, iterations=iters, relTol=relTol, normal="gram")
                O.solve(ref, B[:, j].astype(np.complex128))
                assert abs(stat[j].iteration - ref.iteration) <= (1 if relTol > 0 else 0), (j, stat[j].iteration, ref.iteration)
                assert abs(stat[j].iteration - stat0[j].iteration) <= (1 if relTol > 0 else 0)
                if stat[j].iteration == stat0[j].iteration:
                    assert rel(xs[j], x0s[j]) < 2e-5, (j, rel(xs[j], x0s[j]))
                    assert abs(stat[j].rel_res_norm - stat0[j].rel_res_norm) <= 1e-4 * abs(stat0[j].rel_res_norm) + 1e-12
                if stat[j].iteration == ref.iteration and j in (0, 1, K - 1):
                    x32 = lambda: O.solve(O.FISTA(A, reg=reg(O), rho=rho, iterations=ref.iteration, relTol=0.0, normal="gram"),
                                          np.ascontiguousarray(B[:, j]))
                    parity(f"{tag}_reltol{relTol}_col{j}", xs[j], ref.x, x32, record=(j < 2 and relTol == 0.0))
        # several step calls (4 + 7 + 4 iterations) leave the bits of one call
        F1 = rls.createLinearSolver(rls.FISTA, Ad, AHA=Gd, reg=reg(rls), rho=rho, iterations=iters, relTol=0.0)
        x_once = [x.to_host() for x in rls.solve_(F1, Bd, scheduler=rls.BatchedState)]
        rls.init_(F1, Bd, scheduler=rls.BatchedState)
        for n in (4, 7, 4):
            F1.state._step(n)
        st = F1.state.status()
        assert [s_.iteration for s_ in st] == [iters] * K and all(s_.fallbacks == 0 for s_ in st)
        for j, x in enumerate(F1.state.solutions()):
            assert np.array_equal(x.to_host(), x_once[j]), j
        # single-iteration calls take the streaming kernels (one launch of AHA's rows would not pay): the two paths hand the
        # plan's state to each other
        rls.init_(F1, Bd, scheduler=rls.BatchedState)
        for n in (1, 6, 1, 7):
            F1.state._step(n)
        st = F1.state.status()
        assert [s_.iteration for s_ in st] == [iters] * K and all(s_.fallbacks == 0 for s_ in st)
        for j, x in enumerate(F1.state.solutions()):
            assert rel(x.to_host(), x_once[j]) < 2e-5, (j, rel(x.to_host(), x_once[j]))
        # a lost launch (wait bound of one poll) changes nothing and is re-run on the streaming kernels
        if N >= 128:
            rls.init_(F1, Bd, scheduler=rls.BatchedState)
            ctx.tune(resident_spin=1)
            F1.state._step(iters)
            ctx.tune(resident_spin=100000)
            st = F1.state.status()
            assert [s_.iteration for s_ in st] == [iters] * K and all(s_.fallbacks >= 1 for s_ in st)
            assert _fista_path(ctx, F1) == 3
            for j, x in enumerate(F1.state.solutions()):
                assert rel(x.to_host(), x_once[j]) < 2e-5
    finally:
        ctx.tune(resident=1, resident_spin=100000)
        _fresh_resident_ctx(ctx)


def test_resident_solvers_survive_a_co_tenant(rls, ctx):
    """a kernel on ANOTHER stream sits on 64 whole CUs for longer than the resident kernels' wait bound while a resident
    CGNR step, a resident FISTA step and an ADMM plan (one resident cg! per outer iteration) are issued: the resident
    launches cannot get their 256 workgroups onto the chip and give up; the results still equal the oracle's, nothing
    raises, and every plan reports the fallback"""
    import ctypes as C
    A, xt, b = O.make_problem(4096, 2048, np.complex64, 79)
    A64, b64 = A.astype(np.complex128), b.astype(np.complex128)
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    other = rls.Context(0)  # own stream
    lam = 1e-2 * float(np.max(np.abs(A64.conj().T @ b64)))
    rho_f = 0.95 / np.linalg.norm(A64, 2) ** 2
    cases = {
        "cgnr": (lambda R, Am: R.CGNR(Am, iterations=8, relTol=0.0) if R is O else R.createLinearSolver(R.CGNR, Am, iterations=8, relTol=0.0)),
        "fista": (lambda R, Am: R.FISTA(Am, reg=R.L1Regularization(lam), rho=rho_f, iterations=8) if R is O
                  else R.createLinearSolver(R.FISTA, Am, reg=R.L1Regularization(lam), rho=rho_f, iterations=8)),
        "admm": (lambda R, Am: R.ADMM(Am, reg=R.L1Regularization(lam), rho=0.1, iterations=4, iterationsCG=5, tolInner=1e-5) if R is O
                 else R.createLinearSolver(R.ADMM, Am, reg=R.L1Regularization(lam), rho=0.1, iterations=4, iterationsCG=5, tolInner=1e-5)),
    }
    try:
        for name, make in cases.items():
            _fresh_resident_ctx(ctx)
            ctx.tune(resident_spin=20000)  # ~20 ms per wait
            ref = make(O, A64)
            O.solve(ref, b64)
            sol = make(rls, Ad)
            rls.init_(sol, bd)
            ctx.sync()
            assert _hold_cus(rls, other, 64, 400000) == 0   # 0.4 s on 64 CUs, other stream
            x = rls.solve_(sol, bd).to_host()
            other.sync()
            st = sol.state._refresh(ctx.lib) if name != "admm" else None
            if name == "admm":
                ast = rls._lib.AdmmStatus()
                assert ctx.lib.rls_admm_get_status(sol.state._admm, C.byref(ast), None, 0) == 0
                fallbacks, iteration = ast.fallbacks, ast.iteration
            else:
                fallbacks, iteration = st.fallbacks, st.iteration
            assert fallbacks >= 1, f"{name}: the co-tenant did not displace the resident launch"
            assert iteration == ref.iteration
            parity(f"co_tenant_{name}", x, ref.x, lambda: (lambda o: (O.solve(o, b), o.x)[1])(make(O, A)))
    finally:
        ctx.tune(resident_spin=100000)
        _fresh_resident_ctx(ctx)
        other.close()


@pytest.mark.parametrize("dt,M,N", [(np.complex64, 4096, 2048), (np.float32, 4000, 2200)])
def test_cgnr_resident_server_mode(rls, ctx, dt, M, N):
    """The reference's solve! loop with callbacks is one iterate + one `done` check per call (src/RegularizedLeastSquares.jl:161-176):
    rls_cgnr_step_status leaves the resident kernel LISTENING between calls (server mode, rls_cg_start::srv_ctl) -- the next call
    posts a command into pinned host memory instead of launching.  32 one-iterate calls back to back are the bits of ONE
    32-iteration launch, with the oracle's status stream; the stopping test ends the stream (later calls change nothing);
    anything else that touches the stream -- a download, a new init!, a destroy -- makes the kernel leave first; a kernel that
    left on its idle timeout is replaced; a caller that touches the device between iterates ends up on the per-iteration
    pipeline (same results within the gate); resident_server = 0 never listens."""
    import time
    A, xt, b = O.make_problem(M, N, dt, 91)
    A64, b64 = A.astype(hi(dt)), b.astype(hi(dt))
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    iters = 32
    ref = O.CGNR(A64, iterations=iters, relTol=0.0)
    ref.init(b64)
    res = []
    while ref.iterate() is not None:
        res.append(float(np.linalg.norm(ref.r)))
    sol = rls.createLinearSolver(rls.CGNR, Ad, iterations=iters, relTol=0.0)
    rls.init_(sol, bd)
    if _cgnr_path(rls, sol) != 4:
        _resident_unavailable()
    x_once = rls.solve_(sol, bd).to_host()            # ONE launch of 32 iterations
    rls.init_(sol, bd)
    seen = []
    while rls.iterate(sol) is not None:               # 32 calls: one launch, then 31 commands
        seen.append(sol.state.iteration)
    assert sol.state.iteration == iters and sol.state.fallbacks == 0
    plan = sol.state._plan
    st = sol.state._refresh(ctx.lib)                  # (answered from the mirror: the kernel is still listening)
    assert st.iteration == iters and abs(st.residual - res[-1]) < 1e-4 * res[0]
    assert np.array_equal(sol.state.x.to_host(), x_once)     # the download makes it leave first
    assert _cgnr_path(rls, sol) == 4
    # a pause longer than the idle timeout between calls: the kernel leaves on its own and is replaced
    ctx.tune(resident_server_idle_us=50)
    try:
        rls.init_(sol, bd)
        for k in range(iters):
            assert rls.iterate(sol) is not None
            if k in (3, 9):
                time.sleep(2e-3)
        assert rls.iterate(sol) is None and sol.state.iteration == iters
        parity(f"cgnr_server_idle_{M}x{N}", sol.state.x.to_host(), ref.x, lambda: O.solve(O.CGNR(A, iterations=iters, relTol=0.0), b), record=False)
    finally:
        ctx.tune(resident_server_idle_us=300)
    # a caller that reads x after every iterate: the listening kernel only stands in its way -- pipeline after two short lives
    rls.init_(sol, bd)
    k = 0
    while rls.iterate(sol) is not None:
        k += 1
        xk = sol.state.x.to_host()
        if k == 1:
            ref1 = O.CGNR(A64, iterations=1, relTol=0.0)
            O.solve(ref1, b64)
            assert rel(xk, ref1.x) < 1e-5
    assert k == iters
    parity(f"cgnr_server_downloads_{M}x{N}", sol.state.x.to_host(), ref.x, lambda: O.solve(O.CGNR(A, iterations=iters, relTol=0.0), b), record=False)
    # the stopping test inside a served command: the stream ends at the oracle's iteration, later commands change nothing
    tol = 1e-3
    ref2 = O.CGNR(A64, iterations=iters, relTol=tol)
    O.solve(ref2, b64)
    sol2 = rls.createLinearSolver(rls.CGNR, Ad, iterations=iters, relTol=tol)
    rls.init_(sol2, bd)
    n = 0
    while rls.iterate(sol2) is not None:
        n += 1
    assert abs(n - ref2.iteration) <= 1 and sol2.state.iteration == n
    import ctypes as C
    stt = rls._lib.CgnrStatus()
    for _ in range(3):
        assert ctx.lib.rls_cgnr_step_status(sol2.state._plan, 1, C.byref(stt)) == 0
        assert stt.iteration == n and stt.done == 1
    x2 = sol2.state.x.to_host()
    if n == ref2.iteration:
        parity(f"cgnr_server_reltol_{M}x{N}", x2, ref2.x, lambda: O.solve(O.CGNR(A, iterations=n, relTol=0.0), b), record=False)
    del sol2                                          # destroy with (possibly) a kernel still listening
    if M == 4096:
        # more calls than one kernel life serves (RLS_SRV_MAX_COMMANDS = 2048): it leaves with a command posted, the host
        # re-issues it with a launch -- still the bits of ONE launch
        n_long = 2300
        sol3 = rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(1e-3), iterations=n_long, relTol=0.0)
        x_long = rls.solve_(sol3, bd).to_host()
        it_long = sol3.state.iteration     # (Float32 CG reaches a residual of exactly 0 after a few hundred iterations and stops)
        rls.init_(sol3, bd)
        for _ in range(n_long):            # every call is a served command, iterating or not
            assert ctx.lib.rls_cgnr_step_status(sol3.state._plan, 1, C.byref(stt)) == 0
        assert stt.iteration == it_long and stt.fallbacks == 0
        assert np.array_equal(sol3.state.x.to_host(), x_long, equal_nan=True)
    # switched off: the per-iteration pipeline, as before
    ctx.tune(resident_server=0)
    try:
        rls.init_(sol, bd)
        while rls.iterate(sol) is not None:
            pass
        x_pipe = sol.state.x.to_host()
    finally:
        ctx.tune(resident_server=1)
    assert rel(x_pipe, x_once) < 2e-5 and not np.array_equal(x_pipe, x_once)


def test_frees_do_not_end_a_listening_kernel(rls, ctx):
    """rls_free while a kernel listens (a finalizer of the host's garbage collector, at any time): the pooled free is ordered behind the
    kernel on the stream and must not ask it to leave -- with a free after EVERY iterate call the kernel's lives would all be one
    command long, the plan would go over to the per-iteration pipeline after two of them, and the result would no longer be the bits of
    one launch"""
    import ctypes as C
    A, xt, b = O.make_problem(4096, 2048, np.complex64, 94)
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    iters = 24
    sol = rls.createLinearSolver(rls.CGNR, Ad, iterations=iters, relTol=0.0)
    rls.init_(sol, bd)
    if _cgnr_path(rls, sol) != 4:
        _resident_unavailable()
    x_once = rls.solve_(sol, bd).to_host()
    junk = [rls.DeviceVector(1024, np.float32, ctx) for _ in range(iters)]
    st = rls._lib.CgnrStatus()
    rls.init_(sol, bd)
    for k in range(iters):
        assert ctx.lib.rls_cgnr_step_status(sol.state._plan, 1, C.byref(st)) == 0 and st.iteration == k + 1
        junk.pop()   # -> rls_free
    assert st.fallbacks == 0
    assert np.array_equal(sol.state.x.to_host(), x_once)


@pytest.mark.parametrize("M,N,dt,path,gram", [(4096, 2048, np.complex64, 4, False), (2048, 2048, np.complex64, 5, True), (256, 128, np.float32, 8, False)])
def test_cgnr_server_runs_one_iteration_ahead(rls, ctx, M, N, dt, path, gram):
    """A listening kernel computes the NEXT iteration under the host's turnaround (the SPEC instantiations of cgnr_resident_kernel,
    rls_tune_set("resident_ahead")): nothing of it is published or written back before its command is there.  Commands of 1, 3, 2, ...
    iterates give the status stream and the bits of the kernel that does not run ahead (and the per-iteration pipeline's within
    rounding); a download between two commands (the kernel is told to
    leave AFTER it ran ahead) sees the state of the last command served, and the solve continues from there; the iteration limit and
    the stopping test end the stream where they do without it."""
    import ctypes as C
    A, xt, b = O.make_problem(M, N, dt, 92)
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    Gd = Ad.gram() if gram else None   # (the matrix-free resident kernel, the resident Gram kernel, the single-workgroup kernel)
    sizes = [1, 3, 2, 1, 1, 4, 1, 2, 1]
    iters = sum(sizes)
    lib = ctx.lib
    st = rls._lib.CgnrStatus()

    def stream(server, ahead_, peek_at=()):
        # (idle time at its maximum: a host hiccup between two calls must not end a kernel's life here -- two short lives in a row put
        #  the plan on the per-iteration pipeline, whose sums run in another order: the same iterates to rounding, not to the bit)
        ctx.tune(resident_server=server, resident_ahead=ahead_, resident_server_idle_us=10000)
        gc.collect()   # (an earlier solver freed by the cycle collector in the middle of this stream: rls_free -> rls_enter asks the listening
        #                kernel to leave, and two lives that short in a row put the plan on the per-iteration pipeline -- other rounding)
        sol = rls.createLinearSolver(rls.CGNR, Ad, AHA=Gd, reg=rls.L2Regularization(1e-3), iterations=iters, relTol=0.0)
        rls.init_(sol, bd)
        if _cgnr_path(rls, sol) != path:
            _resident_unavailable()
        out, peeks = [], []
        for k, n in enumerate(sizes):
            assert lib.rls_cgnr_step_status(sol.state._plan, n, C.byref(st)) == 0
            out.append((st.iteration, st.done, st.residual, st.alpha_re, st.beta_re))
            if k in peek_at:
                peeks.append(sol.state.x.to_host())      # the kernel leaves first -- behind the iteration it ran ahead
        assert lib.rls_cgnr_step_status(sol.state._plan, 1, C.byref(st)) == 0 and st.iteration == iters and st.done == 1   # the limit
        x_end = sol.state.x.to_host()
        return out, peeks, x_end, int(st.fallbacks)

    import gc
    gc.disable()
    try:
        pipe_out, pipe_peeks, pipe_x, _ = stream(0, 0, peek_at=(1, 4))   # a launch per command on the per-iteration pipeline
        ref_out, ref_peeks, ref_x, fb = stream(1, 0, peek_at=(1, 4))     # the listening kernel, never ahead
        assert fb == 0 and [o[:2] for o in ref_out] == [o[:2] for o in pipe_out] and rel(ref_x, pipe_x) < 2e-5
        out, peeks, x, fb = stream(1, 1)
        assert fb == 0 and out == ref_out and np.array_equal(x, ref_x)
        out, peeks, x, fb = stream(1, 1, peek_at=(1, 4))
        assert fb == 0 and out == ref_out and np.array_equal(x, ref_x)
        assert len(peeks) == 2 and all(np.array_equal(a, b_) for a, b_ in zip(peeks, ref_peeks))
        assert all(rel(a, b_) < 2e-5 for a, b_ in zip(peeks, pipe_peeks))
    finally:
        gc.enable()
        ctx.tune(resident_server=1, resident_ahead=1, resident_server_idle_us=300)


@pytest.mark.parametrize("M,N,dt,gram", [(4096, 2048, np.complex64, False), (4000, 2200, np.float32, False), (2048, 2048, np.complex64, True),
                                         (256, 128, np.float32, False)])
@pytest.mark.parametrize("restart", ["none", "gradient"])
def test_fista_server_runs_one_iteration_ahead(rls, ctx, M, N, dt, gram, restart):
    """as test_cgnr_server_runs_one_iteration_ahead for rls_fista_step_status on the kernels that run ahead: the matrix-free resident
    kernel (fista_resident_kernel, SPEC: full and ragged shape; the residual of the pass ahead waits in plan scratch), the resident Gram
    kernel (fista_gram_resident_kernel, SRV = 2) and the single-workgroup kernel -- status stream, x, x_{k-1} and state.res of the kernel that does
    not; a download of state.res between two commands sees the command's residual, not the one computed ahead"""
    import ctypes as C
    A, xt, b = O.make_problem(M, N, dt, 93)
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    Gd = Ad.gram() if gram else None
    rho = float(0.9 / np.linalg.norm(A.astype(hi(dt)), 2) ** 2)
    lam = 0.02 * float(np.max(np.abs(A.conj().T @ b)))
    sizes = [1, 2, 1, 1, 3, 1, 2, 1]
    iters = sum(sizes)
    lib = ctx.lib
    st = rls._lib.FistaStatus()

    def stream(ahead_, peek_at=()):
        ctx.tune(resident_server=1, resident_ahead=ahead_, resident_server_idle_us=10000)   # (as in the CGNR test above)
        gc.collect()
        sol = rls.createLinearSolver(rls.FISTA, Ad, AHA=Gd, reg=rls.L1Regularization(lam), rho=rho, iterations=iters, relTol=0.0, restart=restart)
        rls.init_(sol, bd)
        out, peeks = [], []
        for k, n in enumerate(sizes):
            assert lib.rls_fista_step_status(sol.state._plan, n, C.byref(st)) == 0
            out.append((st.iteration, st.done, st.theta, st.rel_res_norm, st.residual))
            if k in peek_at:
                peeks.append(sol.state.res.to_host())    # the kernel leaves first -- behind the iteration it ran ahead
        assert lib.rls_fista_step_status(sol.state._plan, 1, C.byref(st)) == 0 and st.iteration == iters and st.done == 1   # the limit
        sol.state._refresh(lib)
        return out, peeks, sol.state.x.to_host(), sol.state.xold.to_host(), sol.state.res.to_host(), int(st.fallbacks)

    import gc
    gc.disable()
    try:
        ref = stream(0, peek_at=(1, 4))
        assert ref[5] == 0 and ref[0][-1][0] == iters
        for peek in ((), (1, 4)):
            got = stream(1, peek_at=peek)
            assert got[5] == 0 and got[0] == ref[0]
            assert all(np.array_equal(a, b_) for a, b_ in zip(got[2:5], ref[2:5]))
            assert len(got[1]) == len(peek) and all(np.array_equal(a, b_) for a, b_ in zip(got[1], ref[1]))
        want = O.FISTA(A.astype(hi(dt)), AHA=(A.astype(hi(dt)).conj().T @ A.astype(hi(dt))) if gram else None, reg=O.L1Regularization(lam), rho=rho,
                       iterations=iters, relTol=0.0, restart=restart)
        O.solve(want, b.astype(hi(dt)))
        assert rel(ref[2], want.x) < 2e-5
    finally:
        gc.enable()
        ctx.tune(resident_server=1, resident_ahead=1, resident_server_idle_us=300)


@pytest.mark.parametrize("restart", ["none", "gradient"])
def test_fista_resident_server_mode(rls, ctx, restart):
    """rls_fista_step_status with the resident kernel left listening (as test_cgnr_resident_server_mode): 40 one-iterate calls
    back to back are the bits of ONE 40-iteration launch and the oracle's iterate; a download in between makes the kernel leave
    first; the stopping test ends the stream; resident_server = 0 is the per-iteration pipeline."""
    import ctypes as C
    M, N, dt = 4096, 2048, np.complex64
    A, xt, b = O.make_problem(M, N, dt, 95)
    A64, b64 = A.astype(np.complex128), b.astype(np.complex128)
    rho = 0.95 / np.linalg.norm(A64, 2) ** 2
    lam = 1e-2 * float(np.max(np.abs(A64.conj().T @ b64)))
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    iters = 40
    ref = O.FISTA(A64, reg=O.L1Regularization(lam), rho=rho, iterations=iters, relTol=0.0, restart=restart)
    O.solve(ref, b64)
    sol = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(lam), rho=rho, iterations=iters, relTol=0.0, restart=restart)
    rls.init_(sol, bd)
    path = C.c_int32(-1)
    assert ctx.lib.rls_fista_path(sol.state._plan, C.byref(path)) == 0
    if path.value != 4:
        _resident_unavailable()
    x_once = rls.solve_(sol, bd).to_host()
    rls.init_(sol, bd)
    k = 0
    while rls.iterate(sol) is not None:      # one launch, then 39 commands
        k += 1
    assert k == iters and sol.state.iteration == iters and sol.state.fallbacks == 0
    assert abs(sol.state.rel_res_norm - ref.rel_res_norm) < 1e-4 * ref.rel_res_norm + 1e-7
    x_srv = sol.state.x.to_host()
    assert np.array_equal(x_srv, x_once)
    parity(f"fista_server_{restart}", x_srv, ref.x,
           lambda: O.solve(O.FISTA(A, reg=O.L1Regularization(lam), rho=rho, iterations=iters, relTol=0.0, restart=restart), b), record=False)
    # downloads between iterates at 3, 4, 5: the kernel leaves, comes back, and the caller ends up where it should
    rls.init_(sol, bd)
    k = 0
    while rls.iterate(sol) is not None:
        k += 1
        if k in (3, 4, 5):
            sol.state.x.to_host()
    assert k == iters
    parity(f"fista_server_downloads_{restart}", sol.state.x.to_host(), ref.x,
           lambda: O.solve(O.FISTA(A, reg=O.L1Regularization(lam), rho=rho, iterations=iters, relTol=0.0, restart=restart), b), record=False)
    # the stopping test inside a served command
    probe = O.FISTA(A64, reg=O.L1Regularization(lam), rho=rho, iterations=iters, relTol=0.0, restart=restart)
    probe.init(b64)
    rr = []
    while probe.iterate() is not None:
        rr.append(probe.rel_res_norm)
    kk = next(i for i in range(20, 2, -1) if min(rr[:i]) > 1.001 * rr[i])
    tol = 1.0005 * rr[kk]
    sol2 = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(lam), rho=rho, iterations=iters, relTol=tol, restart=restart)
    rls.init_(sol2, bd)
    n = 0
    while rls.iterate(sol2) is not None:
        n += 1
    assert n == kk + 1 and sol2.state.iteration == n
    stt = rls._lib.FistaStatus()
    for _ in range(2):
        assert ctx.lib.rls_fista_step_status(sol2.state._plan, 1, C.byref(stt)) == 0 and stt.iteration == n and stt.done == 1
    ctx.tune(resident_server=0)
    try:
        rls.init_(sol, bd)
        while rls.iterate(sol) is not None:
            pass
        x_pipe = sol.state.x.to_host()
    finally:
        ctx.tune(resident_server=1)
    assert rel(x_pipe, x_once) < 2e-5


@pytest.mark.parametrize("dt,M,N,restart", [(np.complex64, 4096, 2048, "none"), (np.complex64, 2048, 1024, "gradient"), (np.float32, 3000, 1500, "none")])
def test_fista_gram_resident_server_mode(rls, ctx, dt, M, N, restart):
    """server mode on the reference constructor's default operator for FISTA (AHA = A' * A explicit, src/FISTA.jl:58; in the register
    files: rls_fista_path 5): 32 one-iterate calls back to back are the bits of ONE 32-iteration launch and the Gram-mode oracle's
    iterate; a download in between makes the kernel leave first; the stopping test ends the stream"""
    import ctypes as C
    A, xt, b = O.make_problem(M, N, dt, 96)
    dt64 = hi(dt)
    A64, b64 = A.astype(dt64), b.astype(dt64)
    rho = 0.95 / np.linalg.norm(A64, 2) ** 2
    lam = 1e-2 * float(np.max(np.abs(A64.conj().T @ b64)))
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    Gd = Ad.gram()
    iters = 32
    mk_ref = lambda A_, it=iters, tol=0.0: O.FISTA(A_, reg=O.L1Regularization(lam), rho=rho, iterations=it, relTol=tol, restart=restart, normal="gram")
    ref = mk_ref(A64)
    O.solve(ref, b64)
    sol = rls.createLinearSolver(rls.FISTA, Ad, AHA=Gd, reg=rls.L1Regularization(lam), rho=rho, iterations=iters, relTol=0.0, restart=restart)
    rls.init_(sol, bd)
    if _fista_path(ctx, sol) != 5:
        _resident_unavailable()
    x_once = rls.solve_(sol, bd).to_host()
    rls.init_(sol, bd)
    k = 0
    while rls.iterate(sol) is not None:      # one launch, then 31 commands
        k += 1
    assert k == iters and sol.state.iteration == iters and sol.state.fallbacks == 0
    assert abs(sol.state.rel_res_norm - ref.rel_res_norm) < 1e-4 * ref.rel_res_norm + 1e-7
    x_srv = sol.state.x.to_host()
    assert np.array_equal(x_srv, x_once)
    tag = f"fista_gram_server_{M}x{N}_{restart}"
    parity(tag, x_srv, ref.x, lambda: O.solve(mk_ref(A), b), record=False)
    rls.init_(sol, bd)
    k = 0
    while rls.iterate(sol) is not None:
        k += 1
        if k in (3, 4, 5):
            sol.state.x.to_host()
    assert k == iters
    parity(tag + "_downloads", sol.state.x.to_host(), ref.x, lambda: O.solve(mk_ref(A), b), record=False)
    # the stopping test inside a served command
    probe = mk_ref(A64)
    probe.init(b64)
    rr = []
    while probe.iterate() is not None:
        rr.append(probe.rel_res_norm)
    kk = next(i for i in range(20, 2, -1) if min(rr[:i]) > 1.001 * rr[i])
    sol2 = rls.createLinearSolver(rls.FISTA, Ad, AHA=Gd, reg=rls.L1Regularization(lam), rho=rho, iterations=iters, relTol=1.0005 * rr[kk],
                                  restart=restart)
    rls.init_(sol2, bd)
    n = 0
    while rls.iterate(sol2) is not None:
        n += 1
    assert n == kk + 1 and sol2.state.iteration == n
    x_stop = sol2.state.x.to_host()
    stt = rls._lib.FistaStatus()
    for _ in range(2):   # past the end: served, nothing changes
        assert ctx.lib.rls_fista_step_status(sol2.state._plan, 1, C.byref(stt)) == 0 and stt.iteration == n and stt.done == 1
    # the same stop inside ONE launch of all the iterations: without restart the kernel finds it one exchange late (the norm is
    # summed off the critical path) and drops that exchange -- same count, same bits
    x_one = rls.solve_(sol2, bd).to_host()
    assert sol2.state.iteration == n and np.array_equal(x_one, x_stop)
    ctx.tune(resident_server=0)
    try:
        rls.init_(sol, bd)
        while rls.iterate(sol) is not None:
            pass
        x_pipe = sol.state.x.to_host()
    finally:
        ctx.tune(resident_server=1)
    assert rel(x_pipe, x_once) < 2e-5


@pytest.mark.parametrize("dt,M,N", [(np.complex64, 4096, 2048), (np.float32, 3000, 1500)])
def test_cgnr_gram_resident_server_mode(rls, ctx, dt, M, N):
    """server mode on the reference constructor's default operator (AHA = A' * A explicit, held in the register files: rls_cgnr_path 5):
    32 one-iterate calls back to back are the bits of ONE 32-iteration launch and the Gram-mode oracle's iterate; a download in
    between makes the kernel leave first; the stopping test ends the stream"""
    import ctypes as C
    iters = 32
    ref, sol, b, dt64 = _cgnr_pair(rls, M, N, dt, 97, 1e-3, iters, mode="gram")
    O.solve(ref, b.astype(dt64))
    bd = rls.DeviceVector.from_host(b)
    rls.init_(sol, bd)
    if _cgnr_path(rls, sol) != 5:
        _resident_unavailable()
    x_once = rls.solve_(sol, bd).to_host()
    rls.init_(sol, bd)
    k = 0
    while rls.iterate(sol) is not None:
        k += 1
    assert k == iters and sol.state.iteration == iters and sol.state.fallbacks == 0
    x_srv = sol.state.x.to_host()
    assert np.array_equal(x_srv, x_once)
    ref32 = O.CGNR(ref.A.A.astype(dt), reg=O.L2Regularization(1e-3), iterations=iters, relTol=0.0, normal="gram")
    parity(f"cgnr_gram_server_{M}x{N}", x_srv, ref.x, lambda: O.solve(ref32, b), record=False)
    rls.init_(sol, bd)
    k = 0
    while rls.iterate(sol) is not None:
        k += 1
        if k in (3, 4, 5):
            sol.state.x.to_host()
    assert k == iters
    parity(f"cgnr_gram_server_downloads_{M}x{N}", sol.state.x.to_host(), ref.x, lambda: O.solve(ref32, b), record=False)
    stt = rls._lib.CgnrStatus()
    for _ in range(2):   # past the end: served, nothing changes
        assert ctx.lib.rls_cgnr_step_status(sol.state._plan, 1, C.byref(stt)) == 0 and stt.iteration == iters and stt.done == 1


def test_server_mode_with_four_contexts_at_once(rls, ctx):
    """four threads, each with its own context (= stream), each driving a resident CGNR plan one iterate per call: every listening
    kernel holds the whole chip, so they take turns (the resident chain orders the launches; a kernel that is kept waiting leaves on
    its idle timeout) -- nobody deadlocks, nobody gets a wrong iterate"""
    import threading
    M, N = 4096, 2048
    A, xt, b = O.make_problem(M, N, np.complex64, 99)
    S0 = rls.createLinearSolver(rls.CGNR, rls.DeviceMatrix.from_host(A), iterations=24, relTol=0.0)
    x_ref = rls.solve_(S0, rls.DeviceVector.from_host(b)).to_host()
    errs = []

    def worker(k):
        try:
            c = rls.Context(0)
            Ad, bd = rls.DeviceMatrix.from_host(A, c), rls.DeviceVector.from_host(b, c)
            S = rls.createLinearSolver(rls.CGNR, Ad, iterations=24, relTol=0.0)
            for rep in range(3):
                rls.init_(S, bd)
                n = 0
                while rls.iterate(S) is not None:
                    n += 1
                x = S.state.x.to_host()
                assert n == 24 and rel(x, x_ref) < 2e-5, (k, rep, n, rel(x, x_ref))
            del S
            c.close()
        except Exception as e:  # noqa: BLE001 -- surfaced after the join
            errs.append((k, repr(e)))

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(120)
    assert not any(t.is_alive() for t in ths), "a worker is stuck"
    assert not errs, errs


def test_cgnr_resident_server_survives_a_co_tenant(rls, ctx):
    """a listening launch that cannot get its 256 workgroups onto the chip gives up like every resident launch: the call
    re-runs its iterate on the per-iteration pipeline, reports the fallback, and the solve ends at the oracle's iterate"""
    A, xt, b = O.make_problem(4096, 2048, np.complex64, 93)
    A64, b64 = A.astype(np.complex128), b.astype(np.complex128)
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    other = rls.Context(0)
    try:
        _fresh_resident_ctx(ctx)
        ctx.tune(resident_spin=20000)
        ref = O.CGNR(A64, iterations=8, relTol=0.0)
        O.solve(ref, b64)
        sol = rls.createLinearSolver(rls.CGNR, Ad, iterations=8, relTol=0.0)
        rls.init_(sol, bd)
        ctx.sync()
        assert _hold_cus(rls, other, 64, 400000) == 0
        k = 0
        while rls.iterate(sol) is not None:
            k += 1
        other.sync()
        assert k == 8 and sol.state.iteration == 8
        assert sol.state._refresh(ctx.lib).fallbacks >= 1, "the co-tenant did not displace the listening launch"
        parity("co_tenant_cgnr_server", sol.state.x.to_host(), ref.x, lambda: O.solve(O.CGNR(A, iterations=8, relTol=0.0), b), record=False)
    finally:
        ctx.tune(resident_spin=100000)
        _fresh_resident_ctx(ctx)
        other.close()


@pytest.mark.parametrize("name", ["OptISTA", "POGM"])
@pytest.mark.parametrize("dt,M,N", [(np.complex64, 4096, 2048), (np.float32, 2048, 4096), (np.complex64, 4000, 2002), (np.float32, 4000, 2200)])
def test_optista_pogm_resident_launch(rls, ctx, name, dt, M, N):
    """SURVEY 8f-1 / BASELINE configs[1] shape: the remaining iterations of OptISTA / POGM as resident launches of up to 48
    iterations (pgm_resident_kernel through rls_pgm_step_resident).  Against the float64 oracle at 30, 49 (an odd count: POGM's
    x / y roles end swapped; two launches) and 100 iterations (three launches); the stopping test inside a launch stops at
    the oracle's iteration; solver state consistent afterwards (second solve, iterate-by-iterate run from the same state);
    run to run identical; and the launch-per-iteration sequence (resident = 0) passes the same gate"""
    A, xt, b = O.make_problem(M, N, dt, 5)
    A64, b64 = A.astype(hi(dt)), b.astype(hi(dt))
    rho = 0.95 / np.linalg.norm(A64, 2) ** 2
    lam = 1e-2 * np.max(np.abs(A64.conj().T @ b64))
    regs = lambda R: [R.L1Regularization(lam), R.PositiveRegularization()] if (name == "POGM" and N == 2002) else R.L1Regularization(lam)
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    tag = f"{name}_resident_{M}x{N}_{np.dtype(dt).name}"
    for its in (30, 49, 100):
        ref = getattr(O, name)(A64, reg=regs(O), rho=rho, iterations=its, relTol=0.0)
        O.solve(ref, b64)
        ref32 = lambda: O.solve(getattr(O, name)(A, reg=regs(O), rho=rho, iterations=its, relTol=0.0), b)
        sol = rls.createLinearSolver(getattr(rls, name), Ad, reg=regs(rls), rho=rho, iterations=its, relTol=0.0)
        x = rls.solve_(sol, bd).to_host()
        plan = sol._pgm[1]
        if plan is None:
            _resident_unavailable()
        assert not plan.off and plan.fallbacks == 0 and sol.state.iteration == its
        parity(f"{tag}_its{its}", x, ref.x, ref32, record=its == 30)
        assert abs(sol.state.rel_res_norm - ref.rel_res_norm) < 1e-4 * ref.rel_res_norm + 1e-7
        assert np.array_equal(rls.solve_(sol, bd).to_host(), x)
        if its == 49:
            ctx.tune(resident=0)
            try:
                x_seq = rls.solve_(sol, bd).to_host()
            finally:
                ctx.tune(resident=1)
            parity(f"{tag}_sequence", x_seq, ref.x, ref32, record=False)
            assert rel(x_seq, x) < 2e-5
            names = ("zold", "res", "y", "z") if name == "OptISTA" else ("xold", "res", "y", "z")   # every vector of the state
            seq = {v: getattr(sol.state, v).to_host() for v in names}
            assert np.array_equal(rls.solve_(sol, bd).to_host(), x) and not plan.off    # resident again
            for v in names:
                assert rel(getattr(sol.state, v).to_host(), seq[v]) < 1e-4, v
    # the stopping test inside a launch
    probe = getattr(O, name)(A64, reg=regs(O), rho=rho, iterations=90, relTol=0.0)
    probe.init(b64)
    rr = []
    while probe.iterate() is not None:
        rr.append(probe.rel_res_norm)
    k = next(k for k in range(60, 2, -1) if min(rr[:k]) > 1.001 * rr[k])   # a threshold first crossed at iteration k, with margin
    tol = 1.0005 * rr[k]
    ref = getattr(O, name)(A64, reg=regs(O), rho=rho, iterations=90, relTol=tol)
    O.solve(ref, b64)
    assert ref.iteration == k + 1
    sol = rls.createLinearSolver(getattr(rls, name), Ad, reg=regs(rls), rho=rho, iterations=90, relTol=tol)
    for _ in range(2):
        x = rls.solve_(sol, bd).to_host()
        assert sol.state.iteration == ref.iteration
        parity(f"{tag}_reltol", x, ref.x, lambda: O.solve(getattr(O, name)(A, reg=regs(O), rho=rho, iterations=ref.iteration, relTol=0.0), b),
               record=False)
        assert np.isclose(sol.state.rel_res_norm, ref.rel_res_norm, rtol=2e-3)
    # part of the iterations step by step (callbacks), then the state is what a resident run continues from
    sol = rls.createLinearSolver(getattr(rls, name), Ad, reg=regs(rls), rho=rho, iterations=30, relTol=0.0)
    rls.init_(sol, bd)
    for _ in range(7):
        assert rls.iterate(sol) is not None
    sol._run(sol.state)
    ref = getattr(O, name)(A64, reg=regs(O), rho=rho, iterations=30, relTol=0.0)
    O.solve(ref, b64)
    assert sol.state.iteration == 30
    parity(f"{tag}_continued", sol.state.x.to_host(), ref.x,
           lambda: O.solve(getattr(O, name)(A, reg=regs(O), rho=rho, iterations=30, relTol=0.0), b), record=False)


@pytest.mark.parametrize("dt,M,N", [(np.complex64, 4096, 2048), (np.float32, 4000, 2200)])
def test_pogm_gradient_restart_resident_launch(rls, ctx, dt, M, N):
    """POGM with restart = :gradient (src/POGM.jl:183-232) as resident launches (pgm_resident_kernel KIND 2 through
    rls_pogm_step_resident_restart): theta, sigma, gamma and the restart decision stay on the device and the kernel forms every
    iteration's coefficients itself.  Against the float64 oracle at 30, 49 (two launches, x / y roles swapped) and 100 iterations
    with sigma_fac < 1; the recurrence scalars equal those of the launch-per-iteration sequence (resident = 0) and of the oracle;
    the stopping test inside a launch; a run continued after iterate-by-iterate steps (the last iteration's theta rule counts
    from the solve's start, :185)."""
    A, xt, b = O.make_problem(M, N, dt, 5)
    A64, b64 = A.astype(hi(dt)), b.astype(hi(dt))
    rho = 0.95 / np.linalg.norm(A64, 2) ** 2
    lam = 1e-2 * np.max(np.abs(A64.conj().T @ b64))
    kw = dict(restart="gradient", sigma_fac=0.97)
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    tag = f"POGM_restart_resident_{M}x{N}_{np.dtype(dt).name}"
    for its in (30, 49, 100):
        ref = O.POGM(A64, reg=O.L1Regularization(lam), rho=rho, iterations=its, relTol=0.0, **kw)
        O.solve(ref, b64)
        ref32 = lambda: O.solve(O.POGM(A, reg=O.L1Regularization(lam), rho=rho, iterations=its, relTol=0.0, **kw), b)
        sol = rls.createLinearSolver(rls.POGM, Ad, reg=rls.L1Regularization(lam), rho=rho, iterations=its, relTol=0.0, **kw)
        x = rls.solve_(sol, bd).to_host()
        plan = sol._pgm[1]
        if plan is None:
            _resident_unavailable()
        assert not plan.off and plan.fallbacks == 0 and sol.state.iteration == its
        parity(f"{tag}_its{its}", x, ref.x, ref32, record=its == 30)
        assert abs(sol.state.rel_res_norm - ref.rel_res_norm) < 1e-4 * ref.rel_res_norm + 1e-7
        scal = (sol.state.theta, sol.state.thetaold, sol.state.sigma, sol.state.gamma)
        if its < 100:  # (once the iterate has converged the restart criterion is rounding noise: past ~60 iterations the Float32 and
            # Float64 oracles themselves restart at different iterations -- tools/debug_pogm_restart.py -- and only x is compared)
            assert np.allclose(scal, (ref.theta, ref.theta_old, ref.sigma, ref.gamma), rtol=1e-5), (scal, (ref.theta, ref.theta_old, ref.sigma, ref.gamma))
        assert np.array_equal(rls.solve_(sol, bd).to_host(), x)   # run to run identical
        if its == 49:
            ctx.tune(resident=0)
            try:
                x_seq = rls.solve_(sol, bd).to_host()
                scal_seq = (sol.state.theta, sol.state.thetaold, sol.state.sigma, sol.state.gamma)
                w_seq = sol.state.w.to_host()
            finally:
                ctx.tune(resident=1)
            assert rel(x_seq, x) < 2e-5 and scal_seq == scal
            assert np.array_equal(rls.solve_(sol, bd).to_host(), x) and not plan.off
            assert rel(sol.state.w.to_host(), w_seq) < 1e-4
    # the stopping test inside a launch
    probe = O.POGM(A64, reg=O.L1Regularization(lam), rho=rho, iterations=90, relTol=0.0, **kw)
    probe.init(b64)
    rr = []
    while probe.iterate() is not None:
        rr.append(probe.rel_res_norm)
    k = next(k for k in range(60, 2, -1) if min(rr[:k]) > 1.001 * rr[k])
    tol = 1.0005 * rr[k]
    ref = O.POGM(A64, reg=O.L1Regularization(lam), rho=rho, iterations=90, relTol=tol, **kw)
    O.solve(ref, b64)
    assert ref.iteration == k + 1
    sol = rls.createLinearSolver(rls.POGM, Ad, reg=rls.L1Regularization(lam), rho=rho, iterations=90, relTol=tol, **kw)
    for _ in range(2):
        x = rls.solve_(sol, bd).to_host()
        assert sol.state.iteration == ref.iteration
        parity(f"{tag}_reltol", x, ref.x,
               lambda: O.solve(O.POGM(A, reg=O.L1Regularization(lam), rho=rho, iterations=ref.iteration, relTol=0.0, **kw), b), record=False)
    # part of the iterations step by step (callbacks), then resident launches continue from that state
    sol = rls.createLinearSolver(rls.POGM, Ad, reg=rls.L1Regularization(lam), rho=rho, iterations=30, relTol=0.0, **kw)
    rls.init_(sol, bd)
    for _ in range(7):
        assert rls.iterate(sol) is not None
    sol._run(sol.state)
    ref = O.POGM(A64, reg=O.L1Regularization(lam), rho=rho, iterations=30, relTol=0.0, **kw)
    O.solve(ref, b64)
    assert sol.state.iteration == 30
    parity(f"{tag}_continued", sol.state.x.to_host(), ref.x,
           lambda: O.solve(O.POGM(A, reg=O.L1Regularization(lam), rho=rho, iterations=30, relTol=0.0, **kw), b), record=False)
    assert np.isclose(sol.state.theta, ref.theta, rtol=1e-6)


@pytest.mark.parametrize("name", ["OptISTA", "POGM", "POGM-restart"])
def test_optista_pogm_resident_lost_launch(rls, ctx, name):
    """a co-tenant holds 64 CUs for longer than the wait bound: the first resident launch gives up having changed nothing,
    the later launches of the sequence see a stale iteration count and do nothing, the host finishes launch by launch;
    the result equals the oracle's and the plan reports the fallback and retires"""
    A, xt, b = O.make_problem(4096, 2048, np.complex64, 83)
    A64, b64 = A.astype(np.complex128), b.astype(np.complex128)
    rho = 0.95 / np.linalg.norm(A64, 2) ** 2
    lam = 1e-2 * float(np.max(np.abs(A64.conj().T @ b64)))
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    other = rls.Context(0)
    its = 60   # two launches
    kw = {}
    if name == "POGM-restart":  # (restart = :gradient: rls_pogm_step_resident_restart, then rls_pogm_update_auto launch by launch)
        name, kw = "POGM", dict(restart="gradient", sigma_fac=0.97)
    try:
        _fresh_resident_ctx(ctx)
        ctx.tune(resident_spin=20000)
        ref = getattr(O, name)(A64, reg=O.L1Regularization(lam), rho=rho, iterations=its, relTol=0.0, **kw)
        O.solve(ref, b64)
        sol = rls.createLinearSolver(getattr(rls, name), Ad, reg=rls.L1Regularization(lam), rho=rho, iterations=its, relTol=0.0, **kw)
        rls.init_(sol, bd)
        ctx.sync()
        assert _hold_cus(rls, other, 64, 400000) == 0
        sol._run(sol.state)
        other.sync()
        plan = sol._pgm[1]
        if plan is None:
            _resident_unavailable()
        assert plan.off and plan.fallbacks >= 1, "the co-tenant did not displace the resident launch"
        assert sol.state.iteration == its
        parity(f"co_tenant_{name}{'_restart' if kw else ''}", sol.state.x.to_host(), ref.x,
               lambda: O.solve(getattr(O, name)(A, reg=O.L1Regularization(lam), rho=rho, iterations=its, relTol=0.0, **kw), b), record=False)
        x2 = rls.solve_(sol, bd).to_host()   # the retired plan: launch by launch
        assert rel(x2, sol.state.x.to_host()) == 0 and sol.state.iteration == its
    finally:
        ctx.tune(resident_spin=100000)
        _fresh_resident_ctx(ctx)
        other.close()


@pytest.mark.parametrize("dt,M,N,restart", [(np.complex64, 4096, 2048, "none"), (np.complex64, 4096, 2048, "gradient"),
                                            (np.float32, 4096, 4096, "gradient"),
                                            (np.complex64, 4000, 2002, "gradient"), (np.float32, 4000, 2200, "none")])  # ragged: masked instantiation
def test_fista_resident_kernel(rls, ctx, dt, M, N, restart):
    """BASELINE configs[1] shape through fista_resident_kernel (the whole rls_fista_step call in one launch): iterates
    against the float64 oracle step by step and in one call, bit-identical between the two and run to run, projection
    + gradient restart included; the two-launch pipeline (resident = 0) passes the same gate"""
    import ctypes as C
    A, xt, b = O.make_problem(M, N, dt, 2)
    A64, b64 = A.astype(hi(dt)), b.astype(hi(dt))
    rho = 0.95 / np.linalg.norm(A64, 2) ** 2 if M > N else 0.9 / (np.sqrt(M) + np.sqrt(N)) ** 2
    lam = 1e-2 * np.max(np.abs(A64.conj().T @ b64))
    its = 30
    regs = lambda R: [R.L1Regularization(lam), R.PositiveRegularization()] if restart == "gradient" else R.L1Regularization(lam)
    ref = O.FISTA(A64, reg=regs(O), rho=rho, iterations=its, restart=restart)
    ref32 = O.FISTA(A, reg=regs(O), rho=rho, iterations=its, restart=restart)
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    sol = rls.createLinearSolver(rls.FISTA, Ad, reg=regs(rls), rho=rho, iterations=its, restart=restart)
    rls.init_(sol, bd)
    path = C.c_int32(-1)
    assert ctx.lib.rls_fista_path(sol.state._plan, C.byref(path)) == 0
    if path.value != 4:
        _resident_unavailable()
    ref.init(b64)
    ref32.init(b)
    tag = f"fista_resident_{M}x{N}_{np.dtype(dt).name}_{restart}"
    for it in range(1, its + 1):
        assert ref.iterate() is not None and ref32.iterate() is not None and rls.iterate(sol) is not None
        if it in (1, 2, 7, its):
            parity(f"{tag}_it{it}", sol.state.x.to_host(), ref.x, ref32.x)
    assert rls.iterate(sol) is None and sol.state.iteration == its
    assert abs(sol.state.rel_res_norm - ref.rel_res_norm) < 1e-4 * ref.rel_res_norm + 1e-7
    # (iterate() = a step call of ONE iteration = the two-launch pipeline since round 3; the resident kernel proper below)
    x_once = rls.solve_(sol, bd).to_host()
    x_again = rls.solve_(sol, bd).to_host()
    parity(f"{tag}_one_launch", x_once, ref.x, ref32.x)
    rls.init_(sol, bd)
    for n in (its // 3, its // 3, its - 2 * (its // 3)):   # the same iterations as three resident launches
        assert ctx.lib.rls_fista_step(sol.state._plan, n) == 0
    sol.state._refresh(ctx.lib)
    x_steps = sol.state.x.to_host()
    assert sol.state.iteration == its
    assert np.array_equal(x_once, x_steps) and np.array_equal(x_once, x_again)
    seen = []
    x_cb = rls.solve_(sol, bd, callbacks=lambda s_, i: seen.append(i)).to_host()
    assert seen == list(range(its + 1))
    parity(f"{tag}_callbacks", x_cb, ref.x, ref32.x)
    ctx.tune(resident=0)
    try:
        assert ctx.lib.rls_fista_path(sol.state._plan, C.byref(path)) == 0 and path.value == 1
        x_pipe = rls.solve_(sol, bd).to_host()
    finally:
        ctx.tune(resident=1)
    parity(f"{tag}_pipeline", x_pipe, ref.x, ref32.x)


def test_plain_c_program_drives_the_abi(rls, tmp_path):
    """tests/abi_smoke.c: create -> init -> step -> status for CGNR and FISTA, the headline shape, and row-partitioned
    CGNR through the library's own communicator, from a plain-C process (gcc; no Python, C++ or torch inside it) --
    what a host binding include/rls_mi355x.h through its FFI executes"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.dirname(rls.LIB_PATH)
    exe = str(tmp_path / "abi_smoke")
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "abi_smoke.c"),
                    "-o", exe, "-L", pkg, "-lrls_mi355x", "-lm", f"-Wl,-rpath,{pkg}", "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_smoke OK" in out.stdout and "row-sharded CGNR, 4 rank(s)" in out.stdout


@pytest.mark.parametrize("theta", [1.0, 1.7])
def test_fista_warm_start_scalar_vector_and_theta(rls, ctx, theta):
    """init!(solver, b; x0, theta) (src/FISTA.jl:110-129): a scalar x0 is broadcast (`state.x .= x0`), a vector x0 is
    copied, the first extrapolated point is ((theta - 1) / theta + 1) x0, a wrong length is a DimensionMismatch"""
    A, xt, b = O.make_problem(192, 64, np.complex64, 91)
    A64, b64 = A.astype(np.complex128), b.astype(np.complex128)
    rho = 0.9 / np.linalg.norm(A64, 2) ** 2
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    rng = np.random.default_rng(3)
    xv = (rng.standard_normal(64) + 1j * rng.standard_normal(64)).astype(np.complex64)
    for x0 in (0.25, xv):
        ref = O.FISTA(A64, reg=O.L1Regularization(0.02), rho=rho, iterations=12)
        ref.init(b64, x0=x0 if np.ndim(x0) == 0 else x0.astype(np.complex128), theta=theta)
        while ref.iterate() is not None:
            pass
        sol = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(0.02), rho=rho, iterations=12)
        rls.init_(sol, bd, x0=x0, theta=theta)
        while rls.iterate(sol) is not None:
            pass
        parity(f"fista_warm_start_theta{theta}_{'scalar' if np.ndim(x0) == 0 else 'vector'}", sol.state.x.to_host(), ref.x,
               None, record=False)
    with pytest.raises(ValueError, match="DimensionMismatch"):
        rls.init_(sol, bd, x0=xv[:10])


@pytest.mark.parametrize("nshards", [2, 8])
def test_config5_schedule_through_the_library_communicator(rls, ctx, nshards):
    """BASELINE config 5's collective schedule on one GPU: `nshards` row shards of one tall complex A, one context
    (= one stream) and one CGNR plan per shard, the all-reduce of A^H t inside the library (rls_allreduce_sum, direct
    transport: ranks share the device).  Replicated state bit-identical on every rank, solution within the gate of the
    unsharded float64 oracle, same iteration count; unequal shards included"""
    M, N = 2048 + 64 * nshards, 512
    A, xt, b = O.make_problem(M, N, np.complex64, 97)
    cuts = [0] + [int(M * (k + 1) / nshards) // 4 * 4 for k in range(nshards - 1)] + [M]
    cuts[1] += 8  # unequal shards
    shards = [np.asfortranarray(A[cuts[k]:cuts[k + 1]]) for k in range(nshards)]
    parts = [b[cuts[k]:cuts[k + 1]] for k in range(nshards)]
    s = rls.CommRowShardedCGNR(rls, shards, transport=2, lam=1e-3, iterations=16, relTol=0.0)
    try:
        assert s.transport == 2
        x = s.solve(parts)
        xs = [s.solution(r) for r in range(nshards)]
        assert all(np.array_equal(xs[0], xr) for xr in xs[1:])
        assert all(s.status(r)["iteration"] == 16 for r in range(nshards))
        x64, x32 = oracle_pair(lambda A_, b_: O.solve(O.CGNR(A_, reg=O.L2Regularization(1e-3), iterations=16, relTol=0.0), b_), A, b)
        parity(f"config5_schedule_{nshards}_shards_one_gpu", x, x64, x32)
        # a second solve on the same communicator / plans (the double-buffered receive slots keep alternating)
        x2 = s.solve(parts)
        assert np.array_equal(x2, x)
    finally:
        s.close()


def _row_cuts(M, nshards, unequal=True):
    cuts = [0] + [int(M * (k + 1) / nshards) // 4 * 4 for k in range(nshards - 1)] + [M]
    if unequal and nshards > 1:
        cuts[1] += 8
    return cuts


@pytest.mark.parametrize("threads", [True, False])
@pytest.mark.parametrize("nshards", [2, 8])
def test_rowsharded_fista_through_the_library_communicator(rls, ctx, nshards, threads):
    """SURVEY 8e last row: FISTA on a row-partitioned A driven from ONE host process through the library
    (rls_fista_init_rowsharded / rls_fista_step_rowsharded, direct transport: the ranks share the device): replicated
    state bit-identical on every rank, iteration count and solution those of the unsharded oracle; with one host worker
    thread per rank and with the calling thread driving every rank the same bits"""
    M, N = 1536 + 64 * nshards, 384
    A, xt, b = O.make_problem(M, N, np.complex64, 131)
    A64, b64 = A.astype(np.complex128), b.astype(np.complex128)
    rho = 0.95 / np.linalg.norm(A64, 2) ** 2
    lam = 2e-2 * float(np.max(np.abs(A64.conj().T @ b64)))
    cuts = _row_cuts(M, nshards)
    shards = [np.asfortranarray(A[cuts[k]:cuts[k + 1]]) for k in range(nshards)]
    parts = [b[cuts[k]:cuts[k + 1]] for k in range(nshards)]
    regs = lambda R: [R.L1Regularization(lam), R.PositiveRegularization()]
    s = rls.CommRowShardedFISTA(rls, shards, reg=rls.L1Regularization(lam), proj=rls.PositiveRegularization(), transport=2,
                                rho=rho, iterations=20, restart="gradient", threads=threads)
    try:
        x = s.solve(parts)
        xs = [s.solution(r) for r in range(nshards)]
        assert all(np.array_equal(xs[0], xr) for xr in xs[1:])
        ref = O.FISTA(A64, reg=regs(O), rho=rho, iterations=20, restart="gradient")
        x64 = O.solve(ref, b64)
        assert all(s.status(r)["iteration"] == ref.iteration for r in range(nshards))
        parity(f"rowsharded_fista_comm_{nshards}_shards_threads{int(threads)}", x, x64,
               lambda: O.solve(O.FISTA(A, reg=regs(O), rho=rho, iterations=20, restart="gradient"), b))
        assert np.array_equal(s.solve(parts), x)   # a second solve on the same plans and communicator
    finally:
        s.close()


@pytest.mark.parametrize("kind,dt", [("l1", np.complex64), ("tv", np.float32)])
@pytest.mark.parametrize("nshards", [2, 8])
def test_rowsharded_admm_through_the_library_communicator(rls, ctx, nshards, kind, dt):
    """ADMM on a row-partitioned A from ONE host process through the library (rls_admm_init_rowsharded /
    rls_admm_step_rowsharded): iterations_cg + 1 all-reduces per outer iteration whatever the data; outer iteration count,
    inner cg! counts and solution those of the unsharded oracle; replicated state bit-identical on every rank"""
    N = 256
    M = 1024 + 64 * nshards
    A, xt, b = O.make_problem(M, N, dt, 137)
    hdt = hi(dt)
    regs = (lambda R: R.L1Regularization(0.05)) if kind == "l1" else (lambda R: R.TVRegularization(2e-2, shape=(16, 16)))
    kw = dict(rho=0.3, iterations=6, iterationsCG=5, tolInner=1e-4)
    cuts = _row_cuts(M, nshards)
    shards = [np.asfortranarray(A[cuts[k]:cuts[k + 1]]) for k in range(nshards)]
    parts = [b[cuts[k]:cuts[k + 1]] for k in range(nshards)]
    s = rls.CommRowShardedADMM(rls, shards, regs(rls), M, transport=2, **kw)
    try:
        x = s.solve(parts)
        xs = [s.solution(r) for r in range(nshards)]
        assert all(np.array_equal(xs[0], xr) for xr in xs[1:])
        ref = O.ADMM(A, reg=regs(O), **kw)   # Float32 oracle = the reference's path: the counts
        O.solve(ref, b)
        st = s.status()
        assert st["iteration"] == ref.iteration and st["cg_iterations"] == list(ref.cg_iters), (st, ref.iteration, ref.cg_iters)
        ref64 = O.ADMM(A.astype(hdt), reg=regs(O), **dict(kw, iterations=ref.iteration, absTol=0.0, relTol=0.0))
        parity(f"rowsharded_admm_comm_{kind}_{nshards}_shards", x, O.solve(ref64, b.astype(hdt)), ref.x)
    finally:
        s.close()


class _ShardedPanelF64Op:
    """float64 forward / adjoint products of a row-sharded complex64 matrix, 512 columns of one shard at a time (oracle side)"""

    def __init__(self, shards, panel=512):
        self.shards, self.panel = shards, panel
        self.dtype = np.dtype(hi(shards[0].dtype))
        self.shape = (sum(a.shape[0] for a in shards), shards[0].shape[1])

    def mul(self, x):
        return np.concatenate([_PanelF64Op(a, self.panel).mul(x) for a in self.shards])

    def mul_adj(self, y):
        out = np.zeros(self.shape[1], self.dtype)
        lo = 0
        for a in self.shards:
            out += _PanelF64Op(a, self.panel).mul_adj(y[lo:lo + a.shape[0]])
            lo += a.shape[0]
        return out


def test_config5_full_size_eight_shards_on_one_gpu(rls, ctx):
    """BASELINE configs[4] at FULL size under the gate: one 65536 x 8192 ComplexF32 CGNR (4 GiB of A) as 8 row shards of
    512 MiB on one GPU, the library's own communicator between them (rls_comm_*, direct transport, one host worker
    thread per rank), 3 iterations (the float64 oracle, applied shard- and panel-wise on the host, is what this test's time goes
    into: 8 iterations took 120 s, 4 took 52 s of the suite's 600 s budget), replicated state bit-identical across the 8 ranks"""
    from rls_amd.multigpu import make_row_shard
    M, N, nshards, its = 65536, 8192, 8, 3
    shards, cuts = [], [0]
    for r in range(nshards):
        A_r, lo, hi_ = make_row_shard(M, N, r, nshards)
        shards.append(A_r)
        cuts.append(hi_)
    rng = np.random.default_rng(7)
    x_true = ((rng.standard_normal(N) + 1j * rng.standard_normal(N)) / math.sqrt(2)).astype(np.complex64)
    parts = [(a @ x_true).astype(np.complex64) for a in shards]
    b = np.concatenate(parts)
    s = rls.CommRowShardedCGNR(rls, shards, transport=2, iterations=its, relTol=0.0)
    try:
        x = s.solve(parts)
        xs = [s.solution(r) for r in range(nshards)]
        assert all(np.array_equal(xs[0], xr) for xr in xs[1:])
        assert all(s.status(r)["iteration"] == its for r in range(nshards))
    finally:
        s.close()
    op64 = _ShardedPanelF64Op(shards)
    x64 = O.solve(O.CGNR(op64, iterations=its, relTol=0.0), b.astype(np.complex128))
    # the Float32 bound (needed only if the 1e-5 gate alone fails): the complex64 restatement on the concatenated matrix
    parity("BASELINE config 5 full size: CGNR 65536x8192 c64 as 8 shards on one GPU, 3 iterations", x, x64,
           lambda: O.solve(O.CGNR(np.concatenate(shards), iterations=its, relTol=0.0), b))


@pytest.mark.parametrize("dt,M,N,K,kind", [(np.float32, 256, 64, 5, "tv"), (np.complex64, 128, 48, 3, "l1"), (np.float32, 512, 256, 20, "l1pos"),
                                           (np.complex64, 4096, 2048, 8, "l1"), (np.float32, 320, 144, 4, "l2")])
def test_admm_batched_matrix_rhs(rls, ctx, dt, M, N, K, kind):
    """solve!(solver::ADMM, B) with the shared-A scheduler (src/MultiThreading.jl:30-79 applies to every solver): the K
    columns' cg! iterations share the passes over A (rls_cg_create_batched + rls_admm_step), prox / z / u / `done` per
    column.  Every column against its own oracle solve (iteration count, inner cg! counts, solution), columns that
    converge early retire while the others go on, and the reference's scheduler gives the same columns"""
    A, X, B = O.make_problem(M, N, dt, 61, n_rhs=K)
    B = np.asfortranarray(B * (3.0 ** (np.arange(K) % 4))[None, :]).astype(dt)  # different scales: different stopping iterations
    def regs(R):
        if kind == "tv":
            return R.TVRegularization(2e-2, shape=(8, 8))
        if kind == "l1":
            return R.L1Regularization(0.05)
        if kind == "l1pos":
            return [R.L1Regularization(0.05), R.PositiveRegularization()]
        return R.L2Regularization(0.3)
    loose = M < 1000
    kw = dict(rho=0.3, iterations=12 if loose else 4, iterationsCG=6, tolInner=1e-4, **(dict(absTol=1e-3, relTol=5e-2) if loose else {}))
    Ad, Bd = rls.DeviceMatrix.from_host(A), rls.DeviceMatrix.from_host(B)
    S = rls.createLinearSolver(rls.ADMM, Ad, reg=regs(rls), **kw)
    xs = rls.solve_(S, Bd, scheduler=rls.BatchedState)
    assert type(S.state).__name__ == "AdmmBatchedState"
    stat, cgits = S.state.status(), S.state.cg_iterations()
    seen = []
    for j in range(K if M < 1000 else 2):
        ref = O.ADMM(A, reg=regs(O), **kw)                 # Float32 oracle = the reference's path: counts
        O.solve(ref, np.ascontiguousarray(B[:, j]))
        assert stat[j].iteration == ref.iteration and cgits[j] == ref.cg_iters, (j, stat[j].iteration, ref.iteration, cgits[j], ref.cg_iters)
        ref64 = O.ADMM(A.astype(hi(dt)), reg=regs(O), **dict(kw, iterations=ref.iteration, absTol=0.0, relTol=0.0))
        parity(f"admm_batched_{kind}_{M}x{N}_{np.dtype(dt).name}_K{K}_col{j}", xs[j].to_host(), O.solve(ref64, B[:, j].astype(hi(dt))), ref.x,
               record=(j < 2))
        seen.append(ref.iteration)
    if loose and kind != "l2":
        assert len(set(seen)) > 1 or max(seen) < kw["iterations"]  # some column stopped early
    S2 = rls.createLinearSolver(rls.ADMM, Ad, reg=regs(rls), **kw)
    ys = rls.solve_(S2, Bd, scheduler=rls.MultiThreadingState)
    for j in range(K):
        assert rel(xs[j].to_host(), ys[j].to_host()) < 2e-5, j  # two device paths, each gated against the oracle above
    # a vector solve still works afterwards (src/MultiThreading.jl:39-43)
    v = rls.solve_(S, rls.DeviceVector.from_host(np.ascontiguousarray(B[:, 0]))).to_host()
    assert rel(v, ys[0].to_host()) < 2e-5


def test_concurrent_solves_with_distinct_matrices(rls, ctx):
    """config 4, distinct-A flavour on one GPU (docs/src/literate/howto/multi_threading.jl:8-17): 6 solvers with their own
    matrices on 3 contexts / streams driven from worker threads -- each result is bit-identical to solving that problem
    alone, and within the gate of its oracle (CGNR at the resident shape and FISTA at a small one, mixed)"""
    probs = []
    for k in range(6):
        M, N = (4096, 2048) if k % 2 == 0 else (192, 64)
        A, xt, b = O.make_problem(M, N, np.complex64, 200 + k)
        probs.append((A, b))
    def mk(Ad):
        if Ad.M == 4096:
            return rls.createLinearSolver(rls.CGNR, Ad, iterations=16, relTol=0.0)
        return rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(0.02), rho=0.9 / (np.sqrt(192) + np.sqrt(64)) ** 2, iterations=20)
    cs = rls.ConcurrentSolves(rls, n_streams=3)
    try:
        dA = cs.upload([p[0] for p in probs])
        xs = cs.solve(dA, [p[1] for p in probs], mk)
        xs2 = cs.solve(dA, [p[1] for p in probs], mk)
    finally:
        cs.close()
    for k, (A, b) in enumerate(probs):
        Ad = rls.DeviceMatrix.from_host(A)
        alone = rls.solve_(mk(Ad), rls.DeviceVector.from_host(b)).to_host()
        assert np.array_equal(xs[k], alone) and np.array_equal(xs2[k], alone), k
        if A.shape[0] == 4096:
            x64, x32 = oracle_pair(lambda A_, b_: O.solve(O.CGNR(A_, iterations=16, relTol=0.0), b_), A, b)
        else:
            x64, x32 = oracle_pair(lambda A_, b_: O.solve(O.FISTA(A_, reg=O.L1Regularization(0.02), rho=0.9 / (np.sqrt(192) + np.sqrt(64)) ** 2,
                                                                  iterations=20), b_), A, b)
        parity(f"concurrent_solves_problem{k}", xs[k], x64, x32, record=False)


@pytest.mark.parametrize("kind", ["l1", "tv"])
def test_admm_inner_cg_on_the_resident_kernel(rls, ctx, kind):
    """cg! on (AHA + rho I) is the CGNR recurrence: at a shape whose A fits the register files the whole
    inner solve of every ADMM outer iteration is ONE resident launch.  Same outer iteration count, same inner cg!
    counts and the same solution gate as the two-launch pipeline (resident = 0), which the oracle pins"""
    M, N = 4096, 2048
    A, xt, b = O.make_problem(M, N, np.complex64, 87)
    regs = (lambda R: R.L1Regularization(0.05)) if kind == "l1" else (lambda R: R.TVRegularization(2e-2, shape=(64, 32)))
    kw = dict(rho=0.3, iterations=5, iterationsCG=6, tolInner=1e-4)
    ref = O.ADMM(A, reg=regs(O), **kw)
    O.solve(ref, b)
    ref64 = O.ADMM(A.astype(np.complex128), reg=regs(O), **kw)
    O.solve(ref64, b.astype(np.complex128))
    Ad, bd = rls.DeviceMatrix.from_host(A), rls.DeviceVector.from_host(b)
    out = {}
    for res_on in (1, 0):
        ctx.tune(resident=res_on)
        try:
            sol = rls.createLinearSolver(rls.ADMM, Ad, reg=regs(rls), **kw)
            out[res_on] = rls.solve_(sol, bd).to_host()
            assert sol.state._plan_ok and sol.state.iteration == ref.iteration and sol.state.cg_iterations == ref.cg_iters
            again = rls.solve_(sol, bd).to_host()
            assert np.array_equal(again, out[res_on])
        finally:
            ctx.tune(resident=1)
        parity(f"admm_{kind}_4096x2048_c64_resident{res_on}", out[res_on], ref64.x, ref.x)
