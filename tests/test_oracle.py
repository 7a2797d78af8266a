"""CPU tests of the oracle (oracle/rls_oracle.py): the reference's own exact known-answer tests for
this path replayed in NumPy, closed forms, and the committed golden fixtures."""
import os

import numpy as np
import pytest

import rls_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b))


# ---- reference known answers ------------------------------------------------------------------
def test_l2_prox_closed_form():
    """test/testProxMaps.jl:13: x / (1 + 2 lambda)"""
    rng = np.random.default_rng(1234)
    x = np.zeros(256)
    x[rng.integers(0, 256, 5)] = rng.random(5)
    lam = 0.01
    got = O.prox_l2(x.copy(), lam)
    assert rel(got, x / (1 + 2 * lam)) < 1e-12
    assert 0.5 * np.linalg.norm(x - got) ** 2 + lam * np.linalg.norm(got) ** 2 <= lam * np.linalg.norm(x) ** 2


def test_positive_and_real_projection():
    """test/testProxMaps.jl:139-164: explicit reference projection"""
    rng = np.random.default_rng(1234)
    x = rng.standard_normal(256) + 1j * rng.standard_normal(256)
    want = np.maximum(x.real, 0)
    got = O.prox_positive(x.copy())
    assert np.array_equal(got, want.astype(complex))
    assert np.array_equal(O.prox_real(x.copy()), x.real.astype(complex))
    xr = rng.standard_normal(64)
    assert np.array_equal(O.prox_positive(xr.copy()), np.maximum(xr, 0))
    assert np.array_equal(O.prox_real(xr.copy()), xr)


def test_l1_prox_denoises_and_decreases_objective():
    """test/testProxMaps.jl:17-39 (statistical assertions)"""
    rng = np.random.default_rng(1234)
    N, sigma = 256, 0.03
    x = np.zeros(N)
    x[rng.integers(0, N, 5)] = (1 - 2 * sigma) * rng.random(5) + 2 * sigma
    s = np.sum(np.abs(x)) / N * sigma
    noisy = x + s / np.sqrt(2) * (rng.standard_normal(N) + 1j * rng.standard_normal(N))
    den = O.prox_l1(noisy.copy(), 2 * s)
    assert np.linalg.norm(x - den) <= np.linalg.norm(x - noisy)
    assert np.linalg.norm(x - den) / np.linalg.norm(x) < 0.1
    assert 0.5 * np.linalg.norm(noisy - den) ** 2 + O.norm_l1(den, 2 * s) <= O.norm_l1(noisy, 2 * s)


def test_l1_prox_exact_values():
    x = np.array([3.0, -0.5, 0.0, 1e-4, 2.0 + 2.0j], dtype=np.complex128)
    got = O.prox_l1(x.copy(), 1.0)
    assert abs(got[0] - 2.0) < 1e-12 and got[1] == 0 and got[2] == 0 and got[3] == 0
    assert abs(abs(got[4]) - (abs(x[4]) - 1.0)) < 1e-12 and abs(np.angle(got[4]) - np.pi / 4) < 1e-12


def test_l21_prox_matches_definition_and_edge_cases():
    rng = np.random.default_rng(0)
    n, slices = 40, 5
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    slen = n // slices
    want = x.copy()
    for i in range(slen):
        g = np.linalg.norm(x[i::slen])
        want[i::slen] *= max((g - 0.7) / g, 0)
    assert rel(O.prox_l21(x.copy(), 0.7, slices), want) < 1e-12
    # ragged tail joins its group (x[i:sliceLength:end] runs to the end of x)
    y = rng.standard_normal(10)
    want = y.copy()
    for i in range(3):
        g = np.linalg.norm(y[i::3])
        want[i::3] *= max((g - 0.2) / g, 0)
    assert rel(O.prox_l21(y.copy(), 0.2, 3), want) < 1e-12
    # all-zero group: 0 for lam > 0, NaN for lam == 0 (Julia's max propagates NaN)
    z = np.ones(8)
    z[0::4] = 0
    assert np.all(O.prox_l21(z.copy(), 0.5, 2)[0::4] == 0)
    assert np.all(np.isnan(O.prox_l21(z.copy(), 0.0, 2)[0::4]))


def test_gradient_operator_is_adjoint_pair_and_matches_differences():
    rng = np.random.default_rng(3)
    for shape, dims in (((6, 5), (0, 1)), ((4, 3, 5), (0, 1, 2)), ((7,), (0,)), ((6, 5), (1,))):
        n = int(np.prod(shape))
        x = rng.standard_normal(n)
        g = O.grad_apply(x, shape, dims)
        assert g.shape[0] == O.grad_len(shape, dims)
        y = rng.standard_normal(g.shape[0])
        assert abs(np.dot(g, y) - np.dot(x, O.grad_apply_t(y, shape, dims))) < 1e-10
    img = np.arange(12.0).reshape(4, 3, order="F")
    g = O.grad_apply(img.reshape(-1, order="F"), (4, 3), (0, 1))
    assert np.allclose(g[:9], -1.0) and np.allclose(g[9:], -4.0)  # x[i] - x[i+1]; no boundary row


def test_tv_prox_directional_equals_per_column_and_denoises():
    """test/testProxMaps.jl:106-136 (batched 2-D call == per-column 1-D calls) and :75-103"""
    rng = np.random.default_rng(5)
    N = 24
    x = np.zeros((N, N))
    for _ in range(4):
        x[:, rng.integers(0, N):] += rng.standard_normal()
    noisy = x + 0.05 * rng.standard_normal((N, N))
    a = O.prox_tv_fgp(noisy.reshape(-1, order="F").copy(), 0.1, (N, N), (1,), 10).reshape(N, N, order="F")
    b = noisy.copy()
    for j in range(N):
        b[:, j] = O.prox_tv_fgp(noisy[:, j].copy(), 0.1, (N,), (1,), 10)
    assert rel(a, b) < 1e-12
    piece = np.zeros((N, N))
    piece[5:, 9:] += 1.0
    piece[14:, 3:] -= 0.7
    noisy = piece + 0.05 * rng.standard_normal((N, N))
    den = O.prox_tv_fgp(noisy.reshape(-1, order="F").copy(), 0.1, (N, N), None, 20).reshape(N, N, order="F")
    assert np.linalg.norm(den - piece) < np.linalg.norm(noisy - piece)


def test_tv_fgp_equals_dense_matrix_fgp():
    """independent check: the same FGP recursion written with an explicit gradient matrix"""
    rng = np.random.default_rng(9)
    shape, lam, iters = (5, 4), 0.3, 10
    n = 20
    D = np.stack([O.grad_apply(e, shape, (0, 1)) for e in np.eye(n)], axis=1)
    x = rng.standard_normal(n)
    pq = np.zeros(D.shape[0]); rs = pq.copy(); pqo = pq.copy(); t = 1.0
    for _ in range(iters):
        pqo = pq
        xt = x - lam * D.T @ rs
        pq = rs + D @ xt / (8 * lam)
        pq = pq / np.maximum(1, np.abs(pq))
        to = t; t = (1 + np.sqrt(1 + 4 * to * to)) / 2
        rs = (1 + (to - 1) / t) * pq - ((to - 1) / t) * pqo
    want = x - lam * D.T @ pq
    assert rel(O.prox_tv_fgp(x.copy(), lam, shape, None, iters), want) < 1e-12


def test_callback_cadence_and_last_solution():
    """test/testCallbacks.jl:6-16: iterations + 1 callbacks, solutions[end] == x_approx"""
    A, x, b = O.make_problem(32, 32, np.float64, 1)
    s = O.CGNR(A, iterations=10, relTol=0.0)
    seen, sols = [], []
    out = O.solve(s, b, callbacks=[lambda sv, i: seen.append(i), lambda sv, i: sols.append(sv.solution().copy())])
    assert seen == list(range(11)) and np.array_equal(sols[-1], out)
    assert np.linalg.norm(sols[0] - x) > np.linalg.norm(sols[-1] - x)


def test_matrix_solve_equals_column_solves():
    """test/testMultiThreading.jl:10-18, docs/.../multi_threading.jl:40 (==)"""
    A, X, B = O.make_problem(12, 6, np.complex64, 3, n_rhs=4)
    s = O.CGNR(A, iterations=100)
    cols = np.stack([O.solve(O.CGNR(A, iterations=100), B[:, j]).copy() for j in range(4)], axis=1)
    assert np.array_equal(O.solve(s, B), cols)
    assert rel(cols, X) < 0.1


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64, np.complex128])
def test_solvers_converge_on_small_systems(dt):
    """test/testSolvers.jl:3-43: random 3x2-style systems, rtol 0.1 (here 12x6, zero-mean entries)"""
    A, x, b = O.make_problem(12, 6, dt, 11)
    smax = np.linalg.norm(A.astype(np.complex128), 2)
    assert rel(O.solve(O.CGNR(A, iterations=100), b), x) < 0.1
    assert rel(O.solve(O.FISTA(A, reg=O.L1Regularization(1e-6), rho=0.95 / smax ** 2, iterations=200), b), x) < 0.1
    assert rel(O.solve(O.ADMM(A, reg=O.L1Regularization(1e-6), iterations=100), b), x) < 0.1
    AHA = A.conj().T @ A
    assert rel(O.solve(O.CGNR(None, AHA=AHA, iterations=100), A.conj().T @ b), x) < 0.1  # AHA-only (:45-65)


@pytest.mark.parametrize("dt", [np.float32, np.complex128])
def test_next_tier_solvers_converge(dt):
    """test/testSolvers.jl:3-43 style for OptISTA / POGM / SplitBregman"""
    A, x, b = O.make_problem(24, 12, dt, 13)
    rho = 0.95 / np.linalg.norm(A.astype(np.complex128), 2) ** 2
    assert rel(O.solve(O.OptISTA(A, reg=O.L1Regularization(1e-6), rho=rho, iterations=300), b), x) < 0.1
    assert rel(O.solve(O.POGM(A, reg=O.L1Regularization(1e-6), rho=rho, iterations=300), b), x) < 0.1
    assert rel(O.solve(O.POGM(A, reg=O.L1Regularization(1e-6), rho=rho, iterations=300, restart="gradient"), b), x) < 0.1
    assert rel(O.solve(O.SplitBregman(A, reg=O.L1Regularization(1e-6), iterations=20), b), x) < 0.1


def test_kaczmarz_reference_known_answers():
    """test/testKaczmarz.jl:37-131 on the oracle (Float64, as the reference's own test runs it) + the closed form
    that pins the restatement independently: the sweeps converge to (A^H A + lambda I)^-1 A^H b"""
    rng = np.random.default_rng(12345)
    M, N = 12, 8
    A = rng.random((M, N)) + 1j * rng.random((M, N))
    x = rng.random(N) + 1j * rng.random(N)
    b = A @ x
    regm = rng.random(N)
    x_matrix = O.solve(O.Kaczmarz(A, iterations=100, reg=[O.L2Regularization(regm)]), b)
    x_approx = O.solve(O.Kaczmarz(A * (1 / np.sqrt(regm))[None, :], iterations=100, reg=[O.L2Regularization(1.0)]), b) / np.sqrt(regm)
    assert rel(x_matrix, x_approx) < 1e-10
    lam = rng.random()
    assert np.allclose(O.solve(O.Kaczmarz(A, iterations=100, reg=[O.L2Regularization(lam)]), b),
                       O.solve(O.Kaczmarz(A, iterations=100, reg=[O.L2Regularization(np.full(N, lam))]), b))
    w = rng.random(M)
    assert np.allclose(O.solve(O.Kaczmarz(O.weighted_operator(w, A), iterations=200, reg=O.L2Regularization(0.3)), w * b),
                       O.solve(O.Kaczmarz(np.diag(w) @ A, iterations=200, reg=O.L2Regularization(0.3)), w * b))
    assert rel(O.solve(O.Kaczmarz(A, iterations=200), b), x) < 0.1
    perm = np.random.default_rng(1).permutation(M)
    assert rel(O.solve(O.Kaczmarz(A, iterations=200, order_fn=lambda it: perm), b), x) < 0.1
    xt = np.linalg.solve(A.conj().T @ A + 0.3 * np.eye(N), A.conj().T @ b)
    assert rel(O.solve(O.Kaczmarz(A, iterations=3000, reg=O.L2Regularization(0.3)), b), xt) < 1e-10
    # zero rows are skipped (initkaczmarz, src/Kaczmarz.jl:372-383)
    A0 = A.copy()
    A0[2] = 0
    den, idx = O.init_kaczmarz(A0, 0.5)
    assert 2 not in idx and len(idx) == M - 1 and np.allclose(den, 1 / (np.sum(np.abs(A0[idx]) ** 2, axis=1) + 0.5))


def test_svt_prox_maps_reference_tests():
    """test/testProxMaps.jl:167-247 (Nuclear, LLR, LLR fully overlapping): low-rank + noise, the prox must beat
    the noisy input in the regularised objective and bring the estimate closer to the truth"""
    rng = np.random.default_rng(3)
    N, rank, sigma = 32, 2, 0.05
    x = sum(np.outer(rng.random(N), rng.random(N)) for _ in range(rank))
    x = x / x.max()
    noisy = (x + sigma * rng.standard_normal(x.shape)).reshape(-1, order="F")
    est = O.prox_nuclear(noisy.copy(), 5 * sigma, (N, N))
    nuc = lambda v: np.linalg.svd(v.reshape(N, N, order="F"), compute_uv=False).sum()
    assert 0.5 * np.linalg.norm(noisy - est) ** 2 + 5 * sigma * nuc(est) <= 5 * sigma * nuc(noisy)
    assert np.linalg.norm(est - x.reshape(-1, order="F")) < np.linalg.norm(noisy - x.reshape(-1, order="F"))
    # closed form: thresholding a rank-1 matrix s u v' gives (s - lam) u v'
    u, v = rng.standard_normal(6), rng.standard_normal(5)
    X1 = 3.0 * np.outer(u / np.linalg.norm(u), v / np.linalg.norm(v))
    assert np.allclose(O.prox_nuclear(X1.reshape(-1, order="F").copy(), 1.0, (6, 5)).reshape(6, 5, order="F"), X1 * (2.0 / 3.0))
    # LLR: every distinct block is thresholded on its own; the shifted grid only moves the blocks
    shape, bs, K = (8, 6), (4, 3), 5
    img = rng.standard_normal(shape + (K,)) + 1j * rng.standard_normal(shape + (K,))
    flat = img.reshape(-1, order="F")
    out = O.prox_llr(flat.copy(), 0.7, shape, bs).reshape(shape + (K,), order="F")
    blk = img[4:8, 3:6].reshape(12, K, order="F")
    assert np.allclose(out[4:8, 3:6].reshape(12, K, order="F"), O._svt(blk, 0.7))
    rolled = np.roll(img, (1, 2), axis=(0, 1)).reshape(-1, order="F")
    assert np.allclose(np.roll(O.prox_llr(rolled.copy(), 0.7, shape, bs).reshape(shape + (K,), order="F"), (-1, -2), axis=(0, 1)),
                       O.prox_llr(flat.copy(), 0.7, shape, bs, shift=(1, 2)).reshape(shape + (K,), order="F"))
    # fully overlapping blocks denoise at least as well as one grid (test/testProxMaps.jl:222-247 only logs this)
    ov = O.prox_llr_overlapping(flat.copy(), 0.7, shape, bs)
    assert ov.shape == flat.shape and np.all(np.isfinite(ov))


def test_normalization_factors():
    """src/Regularization/NormalizedRegularization.jl:40-58"""
    A = np.array([[3.0, 4.0], [0.0, 2.0], [1.0, 0.0]])
    b = np.array([1.0, -2.0, 3.0j])
    assert O.normalization_factor("none", A, b) is None
    assert O.normalization_factor("measurement", A, b) == pytest.approx(2.0)
    assert O.normalization_factor("measurement", A, None) == 1.0
    assert O.normalization_factor("systemmatrix", A, None) == pytest.approx((25 + 4 + 1) / 2)
    w = np.array([2.0, 1.0, 3.0])
    WA = O.weighted_operator(w, A)  # GPU ext NormalizedRegularization.jl:7-12: weights^2 * rownorm²
    assert O.normalization_factor("systemmatrix", WA, None) == pytest.approx((4 * 25 + 4 + 9) / 2)
    with pytest.raises(ValueError):
        O.normalization_factor("systemmatrix", None, b)


def test_cgnr_closed_forms():
    A, x, b = O.make_problem(64, 32, np.complex128, 3)
    for lam in (0.0, 0.5):
        s = O.CGNR(A, reg=O.L2Regularization(lam), iterations=32, relTol=0.0)
        want = np.linalg.solve(A.conj().T @ A + lam * np.eye(32), A.conj().T @ b)
        assert rel(O.solve(s, b), want) < 1e-8
    with pytest.raises(ValueError):
        O.CGNR(A, reg=O.L1Regularization(0.1))


def test_cg_inplace_solves_spd_system_and_counts_iterations():
    rng = np.random.default_rng(2)
    B = rng.standard_normal((20, 20))
    Aop = B.T @ B + 0.5 * np.eye(20)
    b = rng.standard_normal(20)
    x = np.zeros(20)
    it = O.cg_inplace(x, lambda v: Aop @ v, b, maxiter=200, reltol=1e-12)
    assert rel(x, np.linalg.solve(Aop, b)) < 1e-8 and 0 < it <= 200
    # warm start: the tolerance is relative to the INITIAL residual b - A x0 (cg! semantics), so a
    # nearly exact x0 still iterates on rounding noise but must stay a solution
    x2 = np.linalg.solve(Aop, b) + 1e-6 * rng.standard_normal(20)
    it2 = O.cg_inplace(x2, lambda v: Aop @ v, b, maxiter=10, reltol=1e-3)
    assert it2 <= 10 and rel(x2, np.linalg.solve(Aop, b)) < 1e-6


def test_cg_inplace_iterates_equal_an_independent_cg():
    """`cg!` is third-party (IterativeSolvers, not in the reference tree): the restatement is checked, iterate by
    iterate, against SciPy's conjugate-gradient implementation (an independent code of the same recurrence) on a
    Hermitian positive definite system with a warm start"""
    from scipy.sparse.linalg import cg as scipy_cg, LinearOperator
    rng = np.random.default_rng(6)
    B = rng.standard_normal((30, 30)) + 1j * rng.standard_normal((30, 30))
    Aop = B.conj().T @ B + 0.3 * np.eye(30)
    b = rng.standard_normal(30) + 1j * rng.standard_normal(30)
    x0 = 0.1 * (rng.standard_normal(30) + 1j * rng.standard_normal(30))
    for k in (1, 2, 5, 9):
        x = x0.copy()
        it = O.cg_inplace(x, lambda v: Aop @ v, b, maxiter=k, reltol=0.0)
        assert it == k
        xs, _ = scipy_cg(LinearOperator((30, 30), matvec=lambda v: Aop @ v, dtype=np.complex128), b, x0=x0.copy(), maxiter=k,
                         rtol=0.0, atol=0.0)
        assert rel(x, xs) < 1e-10, k


def test_power_iterations_estimates_top_eigenvalue():
    A, _, _ = O.make_problem(80, 30, np.complex128, 4)
    lam = O.power_iterations(O.NormalOp(O.DenseOp(A)), np.ones(30, dtype=np.complex128), rtol=1e-6, maxiter=500)
    assert abs(lam - np.linalg.norm(A, 2) ** 2) / lam < 1e-3


def test_float32_oracle_tracks_float64_oracle():
    """the parity bar (1e-5) is only meaningful on well-conditioned inputs: SURVEY 7, hard part 4"""
    A, x, b = O.make_problem(256, 128, np.complex64, 1)
    s32 = O.CGNR(A, iterations=10, relTol=0.0); O.solve(s32, b)
    s64 = O.CGNR(A.astype(np.complex128), iterations=10, relTol=0.0); O.solve(s64, b.astype(np.complex128))
    assert rel(s32.x, s64.x) < 5e-6


# ---- golden fixtures --------------------------------------------------------------------------
@pytest.mark.parametrize("name,dt64", [("cgnr_256x128_f32.npz", np.float64), ("cgnr_64x32_c64.npz", np.complex128)])
def test_golden_cgnr(name, dt64):
    g = np.load(os.path.join(GOLD, name))
    s = O.CGNR(g["A"].astype(dt64), reg=O.L2Regularization(float(g["lam"])), iterations=len(g["x"]), relTol=0.0)
    s.init(g["b"].astype(dt64))
    for k in range(len(g["x"])):
        s.iterate()
        assert rel(s.x, g["x"][k]) < 1e-12 and rel(s.p, g["p"][k]) < 1e-10
        assert abs(s.alpha - g["alpha"][k]) < 1e-12 * abs(g["alpha"][k])


def _next_tier_solutions(mod, g, wrap=lambda a: a, vec=lambda a: a):
    """the five solver runs of tests/golden/next_tier_96x40_c64.npz through module `mod` (oracle or product)"""
    A, b = wrap(g["A"]), vec(g["b"])
    rho, lam = float(g["rho"]), float(g["lam"])
    out = {}
    solve = getattr(mod, "solve", None) or mod.solve_
    k = mod.Kaczmarz(A, reg=mod.L2Regularization(0.05), iterations=6)
    out["kaczmarz_x"] = solve(k, b)
    out["optista_x"] = solve(mod.OptISTA(A, reg=mod.L1Regularization(lam), rho=rho, iterations=30), b)
    out["pogm_x"] = solve(mod.POGM(A, reg=mod.L1Regularization(lam), rho=rho, iterations=30), b)
    out["pogm_restart_x"] = solve(mod.POGM(A, reg=mod.L1Regularization(lam), rho=rho, iterations=30, restart="gradient"), b)
    out["splitbregman_x"] = solve(mod.SplitBregman(A, reg=mod.L1Regularization(0.05), rho=0.5, iterations=3, iterationsInner=4,
                                                   iterationsCG=10), b)
    return out


def test_golden_next_tier():
    g = np.load(os.path.join(GOLD, "next_tier_96x40_c64.npz"))
    got = _next_tier_solutions(O, g, wrap=lambda a: a.astype(np.complex128), vec=lambda a: a.astype(np.complex128))
    for k, v in got.items():
        assert rel(v, g[k]) < 1e-12, k


def test_golden_fista_admm_prox():
    g = np.load(os.path.join(GOLD, "fista_l1_64x32_c64.npz"))
    for restart in ("none", "gradient"):
        s = O.FISTA(g["A"].astype(np.complex128), reg=O.L1Regularization(float(g["lam"])), rho=float(g["rho"]),
                    iterations=50, restart=restart)
        O.solve(s, g["b"].astype(np.complex128))
        assert rel(s.x, g["x_" + restart]) < 1e-12
    g = np.load(os.path.join(GOLD, "admm_tv_128x64_f32.npz"))
    s = O.ADMM(g["A"].astype(np.float64), reg=O.TVRegularization(1e-2, shape=(8, 8)), rho=0.1, iterations=10)
    O.solve(s, g["b"].astype(np.float64))
    assert rel(s.x, g["x"]) < 1e-12 and list(g["cg_iters"]) == s.cg_iters
    g = np.load(os.path.join(GOLD, "prox_cases.npz"))
    for tag in ("f32", "c64"):
        x = g[f"x_{tag}"]
        assert np.array_equal(O.prox_l1(x.copy(), 0.35), g[f"l1_{tag}"])
        assert np.array_equal(O.prox_l21(x.copy(), 0.8, 8), g[f"l21_{tag}"])
        assert np.array_equal(O.prox_positive(x.copy()), g[f"pos_{tag}"])


# ---- nested regularisation terms, ProjectionRegularization, plug-and-play prior ----------------------------------


def test_pnp_reference_known_answers():
    """test/testRegularization.jl:1-79 replayed on the restatement: constructor defaults, prox on real and complex
    input (ignoreIm on / off), clipping of lambda to [0, 1] with the reference's warning text"""
    model = lambda x: x
    pnp = O.PnPRegularization(model, [2])
    assert pnp.lam == 1.0 and pnp.model is model and pnp.shape == [2]
    assert pnp.input_transform is O.MinMaxTransform and pnp.ignoreIm is False
    pnp = O.PnPRegularization(0.1, model=model, shape=[2], input_transform=lambda x: x, ignoreIm=True, sMtHeLsE=1)
    assert pnp.ignoreIm is True
    zero = lambda x: np.zeros_like(x)
    pnp = O.PnPRegularization(0.1, model=zero, shape=[2], input_transform=O.IdentityTransform)
    assert np.array_equal(pnp.prox(np.array([1.0, 2.0]), 0.1), [0.9, 1.8])
    out = pnp.prox(np.array([1.0 + 1.0j, 2.0 + 2.0j]), 0.1)
    assert np.array_equal(out.real, [0.9, 1.8]) and np.array_equal(out.imag, [0.9, 1.8])
    pnp_i = O.PnPRegularization(0.1, model=zero, shape=[2], input_transform=O.IdentityTransform, ignoreIm=True)
    out = pnp_i.prox(np.array([1.0 + 1.0j, 2.0 + 2.0j]), 0.1)
    assert np.array_equal(out.real, [0.9, 1.8]) and np.array_equal(out.imag, [1.0, 2.0])
    assert np.array_equal(pnp.prox(np.array([1.0, 2.0]), 1.5), [0.0, 0.0])
    assert "was given λ with value 1.5. Valid range is [0, 1]. λ changed to temp" in pnp.warnings[-1]
    assert np.array_equal(pnp.prox(np.array([1.0, 2.0]), -1.5), [1.0, 2.0])
    assert "-1.5" in pnp.warnings[-1]


def test_nested_terms_reference_examples():
    """docstring example of MaskedRegularization (MaskedRegularization.jl:8-17); testProj (test/testProxMaps.jl:153-164);
    lambda algebra of the scaled terms (ScaledRegularization.jl:23,35,49,62-70); transforms are exact inverses"""
    x = np.full(4, -1.0)
    O.MaskedRegularization(O.PositiveRegularization(), [True, False, True, False]).prox(x)
    assert np.array_equal(x, [0.0, -1.0, 0.0, -1.0])
    rng = np.random.default_rng(1234)
    z = rng.standard_normal(1012) + 1j * rng.standard_normal(1012)
    pr = O.ProjectionRegularization(lambda v: v.real.astype(v.dtype))
    zp = pr.prox(z.copy())
    assert np.linalg.norm(zp - z.real) / np.linalg.norm(z.real) < 1e-4
    assert 0.5 * np.linalg.norm(z - zp) ** 2 + pr.norm(zp) <= pr.norm(z) and pr.norm(z) == np.inf and pr.norm(zp) == 0.0
    l1 = O.L1Regularization(0.5)
    fs = O.FixedScaledRegularization(l1, 4.0)
    assert O.reg_lambda(fs) == 2.0 and O.sink(fs) is l1
    v = np.array([3.0, -1.0, 0.5])
    assert np.array_equal(fs.prox(v.copy()), O.prox_l1(v.copy(), 2.0))           # no lambda given: the nested one
    assert np.array_equal(fs.prox(v.copy(), 0.25), O.prox_l1(v.copy(), 0.25))     # a given lambda passes through
    fp = O.FixedParameterRegularization(l1)
    assert np.array_equal(fp.prox(v.copy(), 123.0), O.prox_l1(v.copy(), 0.5))     # discards what it is given
    au = O.AutoScaledRegularization(l1)
    assert np.allclose(au.prox(v.copy(), 0.1), O.prox_l1(v.copy(), 0.3), rtol=1e-14) and au.factor == 3.0
    assert np.array_equal(au.prox(v.copy(), 0.1), O.prox_l1(v.copy(), 0.1)) and O.reg_lambda(au) == 1.5
    Q, _ = np.linalg.qr(rng.standard_normal((12, 12)))
    tr = O.TransformedRegularization(l1, Q)
    w = rng.standard_normal(12)
    assert np.allclose(tr.prox(w.copy(), 0.2), Q.T @ O.prox_l1(Q @ w, 0.2))
    assert np.isclose(tr.norm(w, 0.2), 0.2 * np.abs(Q @ w).sum())
    for tf in (O.MinMaxTransform(w), O.ZTransform(w), O.IdentityTransform(w), O.ClampedScalingTransform(w, -0.5, 0.7)):
        assert np.allclose(tf.inverse_transform(tf.transform(w.copy())), w)
    assert O._is_projection(O.MaskedRegularization(O.PositiveRegularization(), [True])) and not O._is_projection(fs)


def test_openmp_cgnr_baseline_matches_the_oracle():
    """oracle/cgnr_omp.cpp (the C++/OpenMP `cpu_baseline` leg of bench.py) restates src/CGNR.jl:107-178;
    it must agree with the NumPy restatement, which the reference's own known answers pin"""
    import ctypes as C
    import subprocess

    odir = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    subprocess.run(["make", "-C", odir], check=True, capture_output=True)
    lib = C.CDLL(os.path.join(odir, "_build", "libcgnr_omp.so"))
    lib.cgnr_omp_run.restype = C.c_int64
    lib.cgnr_omp_run.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p,
                                 C.POINTER(C.c_double)]
    for (M, N, lam, its, threads) in ((256, 128, 0.0, 10, 1), (300, 77, 1e-2, 12, 3), (64, 32, 0.5, 5, 2)):
        A, xt, b = O.make_problem(M, N, np.complex64, 11)
        A = np.asfortranarray(A)
        x = np.zeros(N, np.complex64)
        sec = C.c_double()
        n = lib.cgnr_omp_run(A.ctypes.data, M, N, b.ctypes.data, its, 2, lam, threads, x.ctypes.data, C.byref(sec))
        assert n == 2 * its and sec.value > 0
        ref = O.CGNR(A.astype(np.complex128), reg=O.L2Regularization(lam), iterations=its, relTol=0.0)
        assert rel(x, O.solve(ref, b.astype(np.complex128))) < 1e-5


def test_adjoint_product_forms_are_bit_identical():
    """the oracle forms A' y through a cached conjugate transpose (or, for very large matrices and in the GPU tests' panel-wise
    float64 operator, as conj(conj(y) A)): both must give the bits of the plain expression A.conj().T @ y"""
    rng = np.random.default_rng(5)
    for dt in (np.complex128, np.complex64, np.float32, np.float64):
        A = rng.standard_normal((257, 96))
        y = rng.standard_normal(257)
        if np.dtype(dt).kind == "c":
            A = A + 1j * rng.standard_normal((257, 96))
            y = y + 1j * rng.standard_normal(257)
        A, y = np.asfortranarray(A.astype(dt)), y.astype(dt)
        want = A.conj().T @ y
        assert np.array_equal(O.DenseOp(A).mul_adj(y), want)
        assert np.array_equal(np.conj(np.conj(y) @ A), want)


# ---- the reference's convex suite: test/testSolvers.jl:67-201 (shared with the device tests: tests/reference_suite.py) ----
from reference_suite import convex_problem, convex_suite, lasso_kkt_violation, lasso_problem, tv_duality_gap  # noqa: E402


def oracle_default_rho(A):
    """rho = 0.95 / power_iterations(AHA) (src/FISTA.jl:63, src/Utils.jl:262-287) from a seeded randn start"""
    rng = np.random.default_rng(77)
    b0 = rng.standard_normal(A.shape[1]) + 1j * rng.standard_normal(A.shape[1])
    return {"rho": 0.95 / O.power_iterations(O.NormalOp(O.DenseOp(A)), b0.astype(A.dtype))}


@pytest.mark.parametrize("seed", [12345, 7])
@pytest.mark.parametrize("dt", [np.complex128, np.complex64])
def test_reference_convex_suite_on_the_oracle(seed, dt):
    """test/testSolvers.jl:67-201 replayed statement by statement: POGM / OptISTA / FISTA / ADMM with L1Regularization(1e-3),
    200 iterations, gradient restart, the `F .* 1e3` + MeasurementBasedNormalization invariance, ADMM vary_rho :balance from
    rho = 1e6 and 1e-6 and :PnP from 1e-6, SplitBregman plain and measurement-normalised; every `@test x ≈ x_approx rtol = 0.1`.
    The reference runs it in ComplexF64 (x is a Float64 vector: the `elType` argument only types lambda); complex64 is the
    device's arithmetic.  One case does not survive Float32 and is rescaled by 30 instead of 1e3 there: ADMM on `F .* 1e3` with
    the default rho = 0.1 has `cg!` solve (1e6 F'F + 0.1 I) x = beta, condition number 1e7 -- the residual's rounding error
    (1e6 |x| eps) is the size of the right-hand side's null-space part 0.1 (z - u), so Float32 `cg!` cannot see it
    (relative error 0.65-0.77 in the complex64 oracle, 1e-3 in complex128)."""
    F, x, b = convex_problem(seed)
    got = convex_suite(O, F.astype(dt), b.astype(dt), lambda a: a, lambda v: v, lambda v: np.array(v), oracle_default_rho,
                       admm_scale=1e3 if dt == np.complex128 else 30.0)
    assert len(got) == 15
    for label, xa in got.items():
        assert rel(xa, x) < 0.1, (label, rel(xa, x))


# ---- fixed points that pin the prox scalings without reference to the restatement ----------------------------------------
@pytest.mark.parametrize("dt", [np.complex128, np.float64])
def test_fixed_points_pin_the_prox_thresholds(dt):
    """What each solver converges to says which threshold it hands to prox!, whatever the restatement does in between:
      * FISTA / POGM / OptISTA call prox!(reg, x, rho * lambda) after a step of length rho (src/FISTA.jl:157,164;
        src/POGM.jl:212 with gamma; src/OptISTA.jl:188 with rho * gamma): fixed point = the LASSO optimum for lambda --
        with sigma_max(A)^2 = 30 a missing or doubled rho would move the threshold by that factor;
      * ADMM calls prox!(reg, z, lambda / (2 rho)) (src/ADMM.jl:261): its fixed point is the LASSO optimum for lambda / 2;
      * SplitBregman calls prox!(reg, z, lambda / rho) (src/SplitBregman.jl:236): one outer iteration's fixed point is the
        LASSO optimum for lambda.
    KKT: g = A'(A x - b), g_i = -lambda x_i / |x_i| on the support, |g_i| <= lambda off it."""
    A, xt, b = lasso_problem(5, dt=dt)
    smax2 = np.linalg.norm(A, 2) ** 2
    assert smax2 > 20
    lam = 0.2 * np.max(np.abs(A.conj().T @ b))
    rho = 0.95 / smax2
    for S, its, tol in ((O.FISTA, 6000, 1e-7), (O.POGM, 6000, 1e-7), (O.OptISTA, 6000, 1e-3)):
        s = S(A, reg=O.L1Regularization(lam), rho=rho, iterations=its, relTol=0.0)
        x = O.solve(s, b)
        v, nnz = lasso_kkt_violation(A, b, x, lam)
        assert v < tol and 0 < nnz < 40, (S.__name__, v, nnz)
        assert lasso_kkt_violation(A, b, x, lam * rho)[0] > 1 and lasso_kkt_violation(A, b, x, lam / 2)[0] > 0.5  # the test has teeth
    for rho_admm in (0.3, 4.0):
        s = O.ADMM(A, reg=O.L1Regularization(lam), rho=rho_admm, iterations=3000, iterationsCG=200, tolInner=1e-12, absTol=0.0,
                   relTol=0.0)
        x = O.solve(s, b)
        v, nnz = lasso_kkt_violation(A, b, s.z[0], lam / 2)
        assert v < 1e-5 and 0 < nnz < 60, ("ADMM", rho_admm, v, nnz)
        assert rel(x, s.z[0]) < 1e-6 and lasso_kkt_violation(A, b, s.z[0], lam)[0] > 0.4
        s = O.SplitBregman(A, reg=O.L1Regularization(lam), rho=rho_admm, iterations=1, iterationsInner=3000, iterationsCG=200,
                           tolInner=1e-12, absTol=0.0, relTol=0.0)
        s.init(b)
        for _ in range(2999):  # stop short of the Bregman update, which resets z to x (src/SplitBregman.jl:253-263)
            s.iterate()
        v, nnz = lasso_kkt_violation(A, b, s.z[0], lam)
        assert v < 1e-5 and 0 < nnz < 40, ("SplitBregman", rho_admm, v, nnz)


def test_tv_prox_duality_gap():
    """prox!(::TVRegularization) (FGP, src/proximalMaps/ProxTV.jl:82-125) against a certificate that involves none of its
    internals: the duality gap of its output for  1/2 ||u - x||^2 + lambda ||grad u||_1.  Pins lambda's place in the primal
    update (:107, :123) and that the 1 / (8 lambda) step (:95) converges -- for 1-D, 2-D and directional TV."""
    rng = np.random.default_rng(2)
    for shape, dims, lam in (((40,), None, 0.4), ((12, 9), None, 0.25), ((12, 9), (1,), 0.5), ((6, 5, 4), None, 0.15)):
        n = int(np.prod(shape))
        x = np.cumsum(rng.standard_normal(n)) * 0.3 + rng.standard_normal(n)
        d0 = O._as_dims(shape, dims)
        D = np.stack([O.grad_apply(e, shape, d0) for e in np.eye(n)], axis=1)
        u = O.prox_tv_fgp(x.copy(), lam, shape, dims, 4000)
        gap, primal = tv_duality_gap(x, u, lam, D)
        assert -1e-12 < gap < 1e-6, (shape, dims, gap)
        u10 = O.prox_tv_fgp(x.copy(), lam, shape, dims, 10)  # the default iteration count: not converged, but on its way
        gap10, _ = tv_duality_gap(x, u10, lam, D)
        assert gap < gap10 < 0.2, (shape, dims, gap10)
        # a wrong lambda in the primal update would be optimal for another lambda: the certificate must reject that
        assert tv_duality_gap(x, O.prox_tv_fgp(x.copy(), 2 * lam, shape, dims, 4000), lam, D)[0] > 1e-3


def test_preconditioned_cg_iterates_equal_an_independent_pcg():
    """`cg!(...; Pl = solver.precon)` (call site src/ADMM.jl:244; IterativeSolvers is third-party and not in the tree): the
    preconditioned restatement against SciPy's conjugate-gradient code with M = the same preconditioner, iterate by iterate,
    with a warm start; and Pl = identity reproduces the unpreconditioned recurrence"""
    from scipy.sparse.linalg import cg as scipy_cg, LinearOperator
    rng = np.random.default_rng(8)
    n = 40
    B = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    d = 10.0 ** rng.uniform(-1.5, 1.5, n)
    Aop = (B.conj().T @ B) * np.sqrt(d)[:, None] * np.sqrt(d)[None, :] + np.diag(d)   # badly scaled, Hermitian positive definite
    dinv = 1.0 / np.real(np.diag(Aop))
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    x0 = 0.1 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    lin = lambda f: LinearOperator((n, n), matvec=f, dtype=np.complex128)
    for k in (1, 2, 5, 9):
        x = x0.copy()
        assert O.cg_inplace(x, lambda v: Aop @ v, b, maxiter=k, reltol=0.0, Pl=lambda r: dinv * r) == k
        xs, _ = scipy_cg(lin(lambda v: Aop @ v), b, x0=x0.copy(), maxiter=k, rtol=0.0, atol=0.0, M=lin(lambda r: dinv * r))
        assert rel(x, xs) < 1e-9, k
        x1, x2 = x0.copy(), x0.copy()
        O.cg_inplace(x1, lambda v: Aop @ v, b, maxiter=k, reltol=0.0, Pl=lambda r: r.copy())
        O.cg_inplace(x2, lambda v: Aop @ v, b, maxiter=k, reltol=0.0)
        assert rel(x1, x2) < 1e-12
    # the point of it: Jacobi scaling converges where plain CG has not, at the same iteration count
    xp, xu = np.zeros(n, complex), np.zeros(n, complex)
    O.cg_inplace(xp, lambda v: Aop @ v, b, maxiter=25, reltol=0.0, Pl=lambda r: dinv * r)
    O.cg_inplace(xu, lambda v: Aop @ v, b, maxiter=25, reltol=0.0)
    xt = np.linalg.solve(Aop, b)
    assert rel(xp, xt) < 0.2 * rel(xu, xt)
