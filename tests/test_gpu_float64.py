"""Float64 / ComplexF64 element types (round 6; the reference runs every solver in Float32 AND Float64, test/testSolvers.jl:242,
and its prox tests in ComplexF64, test/testProxMaps.jl:47,78,106).  The tuned path (fused plans, resident kernels, matrix cores) is
Float32 / ComplexF32 by SURVEY 8a; double-precision arrays get the L1 protocol with double scalars (rls_*_d, csrc/f64.hip) and the
reference's own loops of every solver on those primitives (Kaczmarz: the sweep kernel in double precision).  Bar: 1e-12 against the
float64 oracle (the same arithmetic in another summation order)."""
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rls_oracle as O  # noqa: E402  (the checker)

pytestmark = pytest.mark.gpu
DT = [np.float64, np.complex128]


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


def draw(rng, dt, *shape):
    v = rng.standard_normal(shape)
    if np.dtype(dt).kind == "c":
        v = (v + 1j * rng.standard_normal(shape)) / math.sqrt(2)
    return v.astype(dt)


@pytest.fixture(scope="module")
def rls():
    import rls_amd
    return rls_amd


@pytest.fixture(scope="module")
def ctx(rls):
    return rls.default_context(0)


@pytest.mark.parametrize("dt", DT)
def test_l1_protocol_in_double_precision(rls, ctx, dt):
    """mul!(y, A, x) / mul!(x, A', y) / 5-arg mul!, dot, norm, norm(x, 1), rmul!, the fused broadcasts, fill: against NumPy"""
    rng = np.random.default_rng(11)
    M, N = 1037, 259   # ragged on purpose
    A, x, y = draw(rng, dt, M, N), draw(rng, dt, N), draw(rng, dt, M)
    Ad = rls.DeviceMatrix.from_host(np.asfortranarray(A), ctx)
    xd, yd = rls.DeviceVector.from_host(x, ctx), rls.DeviceVector.from_host(y, ctx)
    al, be = (0.7 - 0.2j, -0.3 + 0.5j) if np.dtype(dt).kind == "c" else (0.7, -0.3)
    assert rel(Ad.mul_(yd.copy(), xd, al, be).to_host(), al * (A @ x) + be * y) < 1e-14
    assert rel(Ad.mul_adj_(xd.copy(), yd, al, be).to_host(), al * (A.conj().T @ y) + be * x) < 1e-14
    assert rel(Ad.mul_transpose_(xd.copy(), yd).to_host(), A.T @ y) < 1e-14
    assert rel((Ad @ xd).to_host(), A @ x) < 1e-14
    z = draw(rng, dt, 100003)
    w = draw(rng, dt, 100003)
    zd, wd = rls.DeviceVector.from_host(z, ctx), rls.DeviceVector.from_host(w, ctx)
    assert abs(zd.norm() - np.linalg.norm(z)) < 1e-13 * np.linalg.norm(z)
    assert abs(zd.norm1() - np.sum(np.abs(z))) < 1e-13 * np.sum(np.abs(z))
    assert abs(zd.dot(wd) - np.vdot(z, w)) < 1e-12 * abs(np.vdot(z, w)) + 1e-9
    assert rel(zd.copy().rmul_(al).to_host(), al * z) < 1e-15
    assert rel(zd.copy().axpy_(al, wd).to_host(), z + al * w) < 1e-15
    assert rel(zd.copy().axpby_(al, wd, be).to_host(), al * w + be * z) < 1e-15
    assert rel(zd.similar().lincomb_(al, zd, be, wd).to_host(), al * z + be * w) < 1e-15
    assert np.all(zd.similar().fill_(al).to_host() == np.dtype(dt).type(al))
    assert zd.norm() == zd.norm()   # run to run: the same bits


@pytest.mark.parametrize("dt", DT)
def test_prox_maps_in_double_precision(rls, ctx, dt):
    """test/testProxMaps.jl's ComplexF64 cases: L1 (:47), L2 closed form (:13), L21 (:78), Positive / Real (:139-164), TV (:106-128)"""
    rng = np.random.default_rng(5)
    x = draw(rng, dt, 4096)
    up = lambda v: rls.DeviceVector.from_host(v, ctx)
    lam = 0.37
    assert rel(rls.prox_(rls.L1Regularization, up(x), lam).to_host(), O.prox_l1(x.copy(), lam)) < 1e-14
    assert rel(rls.prox_(rls.L2Regularization, up(x), lam).to_host(), x / (1 + 2 * lam)) < 1e-15
    assert rel(rls.prox_(rls.L21Regularization, up(x), lam, slices=16).to_host(), O.prox_l21(x.copy(), lam, 16)) < 1e-14
    got = rls.prox_(rls.PositiveRegularization, up(x)).to_host()
    assert np.array_equal(got, np.maximum(x.real, 0).astype(dt))
    got = rls.prox_(rls.RealRegularization, up(x)).to_host()
    assert np.array_equal(got, x.real.astype(dt))
    for shape, dims in (((64, 64), None), ((4096,), None), ((16, 16, 16), None), ((64, 64), 1), ((8, 8, 8, 8), (1, 3))):
        want = O.prox_tv_fgp(x.copy(), 0.3, shape, dims, 10)
        got = rls.prox_(rls.TVRegularization, up(x), 0.3, shape=shape, dims=dims).to_host()
        assert rel(got, want) < 1e-12, (shape, dims, rel(got, want))


@pytest.mark.parametrize("dt", DT)
def test_cgnr_fista_admm_in_double_precision(rls, ctx, dt):
    """the three solvers of SURVEY 8a on a double-precision operator: iterates of the reference's loops (src/CGNR.jl:143-178,
    src/FISTA.jl:139-185, src/ADMM.jl:230-322) against the float64 oracle at 1e-12; callbacks cadence; explicit AHA"""
    M, N = 384, 160
    A, xt, b = O.make_problem(M, N, dt, 77)
    Ad, bd = rls.DeviceMatrix.from_host(A, ctx), rls.DeviceVector.from_host(b, ctx)
    # CGNR with an L2 weight, and at lambda = 0 with relTol retirement
    for lam, relTol in ((1e-2, 0.0), (0.0, 1e-9)):
        ref = O.CGNR(A, reg=O.L2Regularization(lam), iterations=25, relTol=relTol)
        O.solve(ref, b)
        S = rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(lam), iterations=25, relTol=relTol)
        seen = []
        x = rls.solve_(S, bd, callbacks=[lambda sv, i: seen.append(i)]).to_host()
        assert x.dtype == np.dtype(dt) and S.state.iteration == ref.iteration and seen == list(range(ref.iteration + 1))
        assert rel(x, ref.x) < 1e-12, rel(x, ref.x)
        assert abs(S.state.alphal - ref.alpha) <= 1e-10 * abs(ref.alpha)
    # the constructors' default for a dense matrix: AHA explicit (here formed on the host in double precision)
    G = np.asfortranarray(A.conj().T @ A)
    Sg = rls.createLinearSolver(rls.CGNR, Ad, AHA=rls.DeviceMatrix.from_host(G, ctx), iterations=20, relTol=0.0)
    refg = O.CGNR(A, AHA=G, iterations=20, relTol=0.0)
    O.solve(refg, b)
    assert rel(rls.solve_(Sg, bd).to_host(), refg.x) < 1e-11
    # FISTA + L1 / TV (+ Positive), explicit rho
    rho = 0.9 / np.linalg.norm(A, 2) ** 2
    lam1 = 0.02 * float(np.max(np.abs(A.conj().T @ b)))
    for regs in (lambda R: R.L1Regularization(lam1), lambda R: [R.TVRegularization(lam1, shape=(16, 10)), R.PositiveRegularization()]):
        for restart in ("none", "gradient"):
            ref = O.FISTA(A, reg=regs(O), rho=rho, iterations=30, relTol=0.0, restart=restart)
            O.solve(ref, b)
            S = rls.createLinearSolver(rls.FISTA, Ad, reg=regs(rls), rho=rho, iterations=30, relTol=0.0, restart=restart)
            x = rls.solve_(S, bd).to_host()
            assert rel(x, ref.x) < 1e-12, (restart, rel(x, ref.x))
    # ADMM + L1 and + TV
    for regs in (lambda R: R.L1Regularization(0.05), lambda R: R.TVRegularization(0.02, shape=(16, 10))):
        kw = dict(rho=0.3, iterations=8, iterationsCG=6, tolInner=1e-8, absTol=0.0, relTol=0.0)
        ref = O.ADMM(A, reg=regs(O), **kw)
        O.solve(ref, b)
        S = rls.createLinearSolver(rls.ADMM, Ad, reg=regs(rls), **kw)
        x = rls.solve_(S, bd).to_host()
        assert S.state.iteration == ref.iteration and S.state.cg_iterations == ref.cg_iters
        assert rel(x, ref.x) < 1e-11, rel(x, ref.x)


@pytest.mark.parametrize("dt", DT)
def test_optista_pogm_splitbregman_kaczmarz_in_double_precision(rls, ctx, dt):
    """the other four solvers of linearSolverList() on a double-precision operator (src/OptISTA.jl:169-209, src/POGM.jl:169-237,
    src/SplitBregman.jl:204-271, src/Kaczmarz.jl:283-308): iterates against the float64 oracle"""
    M, N = 384, 160
    A, xt, b = O.make_problem(M, N, dt, 78)
    Ad, bd = rls.DeviceMatrix.from_host(A, ctx), rls.DeviceVector.from_host(b, ctx)
    rho = 0.9 / np.linalg.norm(A, 2) ** 2
    lam1 = 0.02 * float(np.max(np.abs(A.conj().T @ b)))
    ref = O.OptISTA(A, reg=O.L1Regularization(lam1), rho=rho, iterations=30, relTol=0.0)
    O.solve(ref, b)
    S = rls.createLinearSolver(rls.OptISTA, Ad, reg=rls.L1Regularization(lam1), rho=rho, iterations=30, relTol=0.0)
    assert rel(rls.solve_(S, bd).to_host(), ref.x) < 1e-12
    for restart in ("none", "gradient"):
        for regs in (lambda R: R.L1Regularization(lam1), lambda R: [R.L2Regularization(lam1), R.PositiveRegularization()]):
            ref = O.POGM(A, reg=regs(O), rho=rho, iterations=30, relTol=0.0, restart=restart)
            O.solve(ref, b)
            S = rls.createLinearSolver(rls.POGM, Ad, reg=regs(rls), rho=rho, iterations=30, relTol=0.0, restart=restart)
            x = rls.solve_(S, bd).to_host()
            assert x.dtype == np.dtype(dt) and rel(x, ref.x) < 1e-12, (restart, rel(x, ref.x))
    for regs in (lambda R: R.L1Regularization(0.05), lambda R: R.TVRegularization(0.02, shape=(16, 10))):
        kw = dict(rho=0.3, iterations=3, iterationsInner=4, iterationsCG=6, tolInner=1e-8, absTol=0.0, relTol=0.0)
        ref = O.SplitBregman(A, reg=regs(O), **kw)
        O.solve(ref, b)
        S = rls.createLinearSolver(rls.SplitBregman, Ad, reg=regs(rls), **kw)
        x = rls.solve_(S, bd).to_host()
        assert rel(x, ref.x) < 1e-11, rel(x, ref.x)
    # Kaczmarz: x and vl after full sweeps; a Tikhonov vector; a matrix right-hand side (one workgroup per column); L1 between sweeps
    for lam in (0.0, 1e-2):
        ref = O.Kaczmarz(A, reg=O.L2Regularization(lam), iterations=3)
        O.solve(ref, b)
        S = rls.createLinearSolver(rls.Kaczmarz, Ad, reg=rls.L2Regularization(lam), iterations=3)
        x = rls.solve_(S, bd).to_host()
        assert x.dtype == np.dtype(dt) and rel(x, ref.x) < 1e-12, rel(x, ref.x)
        assert rel(S.state.vl.to_host(), ref.vl) < 1e-11 or lam == 0.0
    lv = np.linspace(0.5, 2.0, N)
    ref = O.Kaczmarz(A, reg=O.L2Regularization(lv), iterations=2)
    S = rls.createLinearSolver(rls.Kaczmarz, Ad, reg=rls.L2Regularization(lv), iterations=2)
    assert rel(rls.solve_(S, bd).to_host(), O.solve(ref, b)) < 1e-12
    ref = O.Kaczmarz(A, reg=[O.L2Regularization(1e-2), O.L1Regularization(1e-3)], iterations=3)
    S = rls.createLinearSolver(rls.Kaczmarz, Ad, reg=[rls.L2Regularization(1e-2), rls.L1Regularization(1e-3)], iterations=3)
    assert rel(rls.solve_(S, bd).to_host(), O.solve(ref, b)) < 1e-12
    B = np.asfortranarray(np.stack([b, 2 * b - 1, b[::-1]], axis=1))
    S = rls.createLinearSolver(rls.Kaczmarz, Ad, reg=rls.L2Regularization(1e-2), iterations=2)
    X = rls.solve_(S, rls.DeviceMatrix.from_host(B, ctx))
    X = X.to_host() if hasattr(X, "to_host") else np.stack([c.to_host() for c in X], axis=1)
    for j in range(B.shape[1]):
        ref = O.Kaczmarz(A, reg=O.L2Regularization(1e-2), iterations=2)
        assert rel(X[:, j], O.solve(ref, B[:, j].copy())) < 1e-12


@pytest.mark.parametrize("dt", DT)
def test_matrix_right_hand_side_in_double_precision(rls, ctx, dt):
    """solve!(solver, B) with a double-precision matrix B (src/MultiThreading.jl:30-79): independent per-column states on the
    primitives (the shared-A batched plans are Float32 / ComplexF32)"""
    M, N, K = 200, 96, 3
    A, _, _ = O.make_problem(M, N, dt, 31)
    rng = np.random.default_rng(32)
    B = np.asfortranarray(draw(rng, dt, M, K))
    B[:, 1] = A @ draw(rng, dt, N) * 1e-3   # (a consistent column of another scale)
    Ad, Bd = rls.DeviceMatrix.from_host(A, ctx), rls.DeviceMatrix.from_host(B, ctx)
    for mk_ref, mk in ((lambda: O.CGNR(A, reg=O.L2Regularization(1e-3), iterations=20, relTol=0.0),
                        lambda: rls.createLinearSolver(rls.CGNR, Ad, reg=rls.L2Regularization(1e-3), iterations=20, relTol=0.0)),
                       (lambda: O.FISTA(A, reg=O.L1Regularization(0.05), rho=0.9 / np.linalg.norm(A, 2) ** 2, iterations=20, relTol=0.0),
                        lambda: rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(0.05), rho=0.9 / np.linalg.norm(A, 2) ** 2,
                                                       iterations=20, relTol=0.0))):
        X = rls.solve_(mk(), Bd)
        X = X.to_host() if hasattr(X, "to_host") else np.stack([c.to_host() for c in X], axis=1)
        assert X.dtype == np.dtype(dt) and X.shape == (N, K)
        for j in range(K):
            assert rel(X[:, j], O.solve(mk_ref(), B[:, j].copy())) < 1e-10, j


def test_reference_small_systems_float64_arm(rls, ctx):
    """test/testSolvers.jl:242 `for elType in [Float32, Float64]`: the 3 x 2 `rand` systems of :3-65 in Float64 / ComplexF64 for every
    solver of linearSolverList() (`x_approx ≈ x rtol = 0.1`), with A, with a complex A, and with AHA alone (:44-56)"""
    rng = np.random.default_rng(12345)
    A = np.asfortranarray(rng.random((3, 2)))
    x = rng.random(2)
    Ac = np.asfortranarray(rng.random((3, 2)) + 1j * rng.random((3, 2)))
    xc = rng.random(2) + 1j * rng.random(2)
    for S in rls.linearSolverList():
        sol = rls.createLinearSolver(S, rls.DeviceMatrix.from_host(A, ctx), iterations=200)
        assert rel(rls.solve_(sol, rls.DeviceVector.from_host(A @ x, ctx)).to_host(), x) < 0.1, S.__name__
        sol = rls.createLinearSolver(S, rls.DeviceMatrix.from_host(Ac, ctx), iterations=100)
        assert rel(rls.solve_(sol, rls.DeviceVector.from_host(Ac @ xc, ctx)).to_host(), xc) < 0.1, S.__name__
        if S is rls.Kaczmarz:
            continue   # (no AHA-only constructor: src/Kaczmarz.jl:76)
        AHA = np.asfortranarray(Ac.conj().T @ Ac)
        sol = rls.createLinearSolver(S, None, AHA=rls.DeviceMatrix.from_host(AHA, ctx), iterations=100)
        assert rel(rls.solve_(sol, rls.DeviceVector.from_host(AHA @ xc, ctx)).to_host(), xc) < 0.1, S.__name__
    # the fused entry points refuse double-precision codes instead of misreading the memory
    v = rls.DeviceVector.from_host(np.ones(8), ctx)
    r = (__import__("ctypes").c_float * 2)()
    assert ctx.lib.rls_nrm2(ctx.handle, v.code, v.n, v.ptr, r) != 0
    assert ctx.lib.rls_kaczmarz_sweep(ctx.handle, v.code, 2, 2, v.ptr, 2, 1, v.ptr, 2, v.ptr, 2, v.ptr, 2, v.ptr, v.ptr, 1, 0.0, 1) != 0
