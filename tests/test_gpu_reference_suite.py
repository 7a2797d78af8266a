"""`-m gpu`: the reference's own convex solver suite (test/testSolvers.jl:67-201) through the product's `createLinearSolver`
and the C ABI, fixed-point certificates that pin the prox thresholds on the device without reference to the oracle's
restatement, and the committed golden fixtures (tests/golden/*.npz) held against the HIP path directly."""
import os

import math

import numpy as np
import pytest

import rls_oracle as O
from conftest import parity_check as parity
from reference_suite import convex_problem, convex_suite, lasso_kkt_violation, lasso_problem, tv_duality_gap

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a = np.asarray(a).astype(np.complex128)
    b = np.asarray(b).astype(np.complex128)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


class ViaCreateLinearSolver:
    """`mod.FISTA(A, **kw)` -> `rls.createLinearSolver(rls.FISTA, A; kw...)`: the suite goes through the reference's front door"""

    def __init__(self, rls):
        self.rls = rls

    def __getattr__(self, name):
        obj = getattr(self.rls, name)
        if isinstance(obj, type) and issubclass(obj, self.rls.AbstractLinearSolver):
            return lambda A, **kw: self.rls.createLinearSolver(obj, A, **kw)
        return obj


@pytest.mark.parametrize("seed", [12345, 7])
def test_reference_convex_suite_on_device(rls, ctx, seed):
    """test/testSolvers.jl:67-201, every statement, in ComplexF32 on the device: POGM / OptISTA / FISTA / ADMM with
    L1Regularization(1e-3), the constructors' DEFAULT rho (power_iterations on the device, src/FISTA.jl:63), gradient restart, the
    `F .* 1e3` + MeasurementBasedNormalization invariance, ADMM `vary_rho` :balance / :PnP from 1e6 / 1e-6, SplitBregman plain
    and measurement-normalised; `@test x ≈ x_approx rtol = 0.1`.  Beside the reference's assertion every case is held against
    the float64 oracle's run of the same case (the iterates of 200 iterations; Float32 noise bound from the complex64 oracle)."""
    F, x, b = convex_problem(seed)
    Fc, bc = F.astype(np.complex64), b.astype(np.complex64)
    wrapA = lambda a: rls.DeviceMatrix.from_host(np.asfortranarray(a.astype(np.complex64)), ctx)
    wrapb = lambda v: rls.DeviceVector.from_host(v.astype(np.complex64), ctx)
    got = convex_suite(ViaCreateLinearSolver(rls), Fc, bc, wrapA, wrapb, lambda v: v.to_host().copy(), lambda A: {},
                       admm_scale=30.0)
    assert len(got) == 15
    for label, xa in got.items():
        assert np.all(np.isfinite(xa)), label
        assert rel(xa, x) < 0.1, (label, rel(xa, x))   # the reference's own assertion
    # the same cases on the oracle with the same (Float32-typed) lambdas; rho: the oracle's power iteration from its own start
    # vector differs from the device's in the 4th digit, which moves a 200-iteration iterate by far less than the gate below
    from test_oracle import oracle_default_rho
    ref = convex_suite(O, F.astype(np.complex128), b.astype(np.complex128), lambda a: a, lambda v: v, lambda v: np.array(v),
                       oracle_default_rho, admm_scale=30.0)
    for label, xa in got.items():
        assert rel(xa, ref[label]) < 2e-3, (label, rel(xa, ref[label]))


@pytest.mark.parametrize("dt", [np.complex64, np.float32])
def test_fixed_points_pin_the_prox_thresholds_on_device(rls, ctx, dt):
    """the LASSO optimality conditions of what the DEVICE solvers converge to (evaluated in float64 on the host): FISTA, POGM,
    OptISTA -> lambda (`rho * lambda` handed to prox!, src/FISTA.jl:164), ADMM -> lambda / 2 (`lambda / (2 rho)`,
    src/ADMM.jl:261), one SplitBregman block -> lambda (`lambda / rho`, src/SplitBregman.jl:236); sigma_max(A)^2 ~ 30 and
    rho_ADMM in {0.3, 4} so that a misplaced rho moves the threshold by a large factor.  No oracle solver is involved."""
    A, xt, b = lasso_problem(5, dt=dt)
    A64 = A.astype(np.complex128 if np.dtype(dt).kind == "c" else np.float64)
    smax2 = np.linalg.norm(A64, 2) ** 2
    lam = float(0.2 * np.max(np.abs(A64.conj().T @ b)))
    rho = 0.95 / smax2
    Ad = rls.DeviceMatrix.from_host(np.asfortranarray(A), ctx)
    bd = rls.DeviceVector.from_host(b, ctx)
    # (OptISTA's LAST iterate is only O(1 / iterations^2)-optimal and its theta recursion runs in Float32: 1.0e-3 on the float64 oracle,
    #  2.0e-3 .. 2.5e-3 on the device depending on the summation order of the operator apply -- a misplaced rho reads > 1)
    for S, its, tol in ((rls.FISTA, 4000, 2e-4), (rls.POGM, 4000, 2e-4), (rls.OptISTA, 4000, 4e-3)):
        s = rls.createLinearSolver(S, Ad, reg=rls.L1Regularization(lam), rho=rho, iterations=its, relTol=0.0)
        x = rls.solve_(s, bd).to_host()
        v, nnz = lasso_kkt_violation(A64, b, x, lam, support_tol=1e-5)
        assert v < tol and 0 < nnz < 40, (S.__name__, v, nnz)
        assert lasso_kkt_violation(A64, b, x, lam * rho, 1e-5)[0] > 1 and lasso_kkt_violation(A64, b, x, lam / 2, 1e-5)[0] > 0.5
    for rho_admm in (0.3, 4.0):
        s = rls.createLinearSolver(rls.ADMM, Ad, reg=rls.L1Regularization(lam), rho=rho_admm, iterations=3000, iterationsCG=100,
                                   tolInner=1e-7, absTol=0.0, relTol=0.0)
        x = rls.solve_(s, bd).to_host()
        # x = z at the fixed point up to the cg! tolerance; threshold x's rounding-level entries as z's prox would
        xs = np.where(np.abs(x) > 1e-4 * np.max(np.abs(x)), x, 0)
        v, nnz = lasso_kkt_violation(A64, b, xs, lam / 2, support_tol=1e-5)
        assert v < 2e-3 and 0 < nnz < 60, ("ADMM", rho_admm, v, nnz)
        assert lasso_kkt_violation(A64, b, xs, lam, 1e-5)[0] > 0.4


def test_tv_prox_duality_gap_on_device(rls, ctx):
    """prox!(::TVRegularization) on the device (the register-resident FGP kernel, the LDS-resident generic stencil and the
    two-launch sequence, depending on the geometry) certified by the duality gap of 1/2 ||u - x||^2 + lambda ||grad u||_1,
    which involves neither the oracle's FGP nor its gradient operator's sign convention beyond |D|."""
    rng = np.random.default_rng(2)
    for shape, dims, lam in (((40,), None, 0.4), ((12, 9), None, 0.25), ((12, 9), (1,), 0.5), ((6, 5, 4), None, 0.15),
                             ((64, 48), None, 0.3)):
        n = int(np.prod(shape))
        x = (np.cumsum(rng.standard_normal(n)) * 0.3 + rng.standard_normal(n)).astype(np.float32)
        d0 = O._as_dims(shape, dims)
        if n <= 512:
            D = np.stack([O.grad_apply(e, shape, d0) for e in np.eye(n)], axis=1)
        kw = dict(shape=shape, iterationsTV=3000, dims=dims)  # dims are 1-based, as in the reference API
        u = rls.prox_(rls.TVRegularization, rls.DeviceVector.from_host(x, ctx), lam, **kw).to_host()
        if n <= 512:
            gap, primal = tv_duality_gap(x, u, lam, D)
            assert -1e-5 < gap < 2e-5, (shape, dims, gap)
            u2 = rls.prox_(rls.TVRegularization, rls.DeviceVector.from_host(x, ctx), 2 * lam, **kw).to_host()
            assert tv_duality_gap(x, u2, lam, D)[0] > 1e-3
        else:  # too large for the dense certificate: the primal objective must not exceed the float64 oracle's optimum
            P = lambda w: 0.5 * np.sum((w - x.astype(np.float64)) ** 2) + lam * np.sum(np.abs(O.grad_apply(w, shape, d0)))
            uo = O.prox_tv_fgp(x.astype(np.float64), lam, shape, dims, 3000)
            assert P(u.astype(np.float64)) <= P(uo) * (1 + 1e-5)


# ---- the committed golden fixtures, held against the HIP path -------------------------------------------------------------
@pytest.mark.parametrize("name", ["cgnr_256x128_f32.npz", "cgnr_64x32_c64.npz"])
def test_golden_cgnr_iterates_on_device(rls, ctx, name):
    """tests/golden/cgnr_*.npz (BASELINE configs[0] and a 64 x 32 ComplexF32 case): x, r, p, alpha, beta of every iteration"""
    g = np.load(os.path.join(GOLD, name))
    A, b = g["A"], g["b"]
    n_it = len(g["x"])
    s = rls.createLinearSolver(rls.CGNR, rls.DeviceMatrix.from_host(np.asfortranarray(A), ctx),
                               reg=rls.L2Regularization(float(g["lam"])), iterations=n_it, relTol=0.0)
    s32 = O.CGNR(A, reg=O.L2Regularization(float(g["lam"])), iterations=n_it, relTol=0.0)
    s32.init(b)
    rls.init_(s, rls.DeviceVector.from_host(b, ctx))
    z0 = np.linalg.norm(g["r"][0])
    for k in range(n_it):
        assert s.iterate() is not None
        s32.iterate()
        st = s.state
        parity(f"golden_{name}_x{k}", st.x.to_host(), g["x"][k], s32.x.copy())
        parity(f"golden_{name}_r{k}", st.x0.to_host(), g["r"][k], s32.r.copy(), scale=z0)
        parity(f"golden_{name}_p{k}", st.pl.to_host(), g["p"][k], s32.p.copy(), scale=z0)
        st._refresh(ctx.lib)
        assert abs(st.alphal - g["alpha"][k]) <= 2e-5 * abs(g["alpha"][k])
        assert abs(st.betal - g["beta"][k]) <= 1e-4 * abs(g["beta"][k])
    assert s.iterate() is None


def test_golden_fista_admm_prox_on_device(rls, ctx):
    """tests/golden/fista_l1_64x32_c64.npz (both restart modes), admm_tv_128x64_f32.npz (solution AND the inner cg! iteration
    counts), prox_cases.npz (L1 / L2 / L21 / Positive / Real / TV incl. the zero and tiny entries) against the device"""
    g = np.load(os.path.join(GOLD, "fista_l1_64x32_c64.npz"))
    Ad = rls.DeviceMatrix.from_host(np.asfortranarray(g["A"]), ctx)
    for restart in ("none", "gradient"):
        s = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(float(g["lam"])), rho=float(g["rho"]), iterations=50,
                                   restart=restart)
        x = rls.solve_(s, rls.DeviceVector.from_host(g["b"], ctx)).to_host()
        o = O.FISTA(g["A"], reg=O.L1Regularization(float(g["lam"])), rho=float(g["rho"]), iterations=50, restart=restart)
        parity(f"golden_fista_{restart}", x, g["x_" + restart], np.array(O.solve(o, g["b"])))
    g = np.load(os.path.join(GOLD, "admm_tv_128x64_f32.npz"))
    kw = dict(rho=0.1, iterations=10, iterationsCG=10, tolInner=1e-5)
    s = rls.createLinearSolver(rls.ADMM, rls.DeviceMatrix.from_host(np.asfortranarray(g["A"]), ctx),
                               reg=rls.TVRegularization(1e-2, shape=(8, 8)), **kw)
    x = rls.solve_(s, rls.DeviceVector.from_host(g["b"], ctx)).to_host()
    o = O.ADMM(g["A"], reg=O.TVRegularization(1e-2, shape=(8, 8)), **kw)
    parity("golden_admm_tv", x, g["x"], np.array(O.solve(o, g["b"])))
    assert s.state.iteration == 10
    assert list(s.state.cg_iterations) == list(g["cg_iters"])
    g = np.load(os.path.join(GOLD, "prox_cases.npz"))
    for tag in ("f32", "c64"):
        x = g[f"x_{tag}"]
        dev = lambda: rls.DeviceVector.from_host(x.copy(), ctx)
        cases = (("l1", rls.prox_(rls.L1Regularization, dev(), 0.35)), ("l2", rls.prox_(rls.L2Regularization, dev(), 0.35)),
                 ("l21", rls.prox_(rls.L21Regularization, dev(), 0.8, slices=8)),
                 ("pos", rls.prox_(rls.PositiveRegularization, dev())), ("real", rls.prox_(rls.RealRegularization, dev())),
                 ("tv", rls.prox_(rls.TVRegularization, dev(), 0.3, shape=(12, 8))),
                 ("tv1", rls.prox_(rls.TVRegularization, dev(), 0.3, shape=(12, 8), dims=(1,))))
        for key, out in cases:
            want = g[f"{key}_{tag}"]
            got = out.to_host()
            if key in ("pos", "real"):
                assert np.array_equal(got, want.astype(got.dtype)), (key, tag)
            else:
                assert rel(got, want) < 2e-6, (key, tag, rel(got, want))


def test_fista_tv_plan_reused_with_other_parameters(rls, ctx):
    """One solver, two solves with iterations >= 2 graph chunks and DIFFERENT rho and lambda (a lambda sweep on one solver;
    MeasurementBasedNormalization does the same per b): the TV threshold rho * lambda, the image geometry and iterationsTV are
    arguments of the plan's captured FGP launch, so a plan that kept its graph across the change would replay the OLD threshold
    for 16 of every 16 + k iterations (review finding, round 5).  The second solve must equal a fresh solver's, bit for bit.
    Also L1 -> TV -> L1 on one plan (another kernel sequence).  src/FISTA.jl:164, src/proximalMaps/ProxTV.jl:64-68."""
    shape = (16, 16)
    N = 256
    A, xt, b = O.make_problem(3 * N, N, np.float32, 977)
    rho = 0.9 / np.linalg.norm(A.astype(np.float64), 2) ** 2
    lam = 0.02 * float(np.max(np.abs(A.astype(np.float64).T @ b)))
    Ad, bd = rls.DeviceMatrix.from_host(A, ctx), rls.DeviceVector.from_host(b, ctx)
    s = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.TVRegularization(lam, shape=shape), rho=rho, iterations=40, relTol=0.0)
    x1 = rls.solve_(s, bd).to_host()
    assert s.state._plan and s.state.iteration == 40
    for rho2, lam2, itv in ((0.5 * rho, 3.0 * lam, 10), (rho, lam, 4)):
        s.reg = rls.TVRegularization(lam2, shape=shape, iterationsTV=itv)
        s.state.rho = rho2
        x2 = rls.solve_(s, bd).to_host()
        f = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.TVRegularization(lam2, shape=shape, iterationsTV=itv), rho=rho2, iterations=40,
                                   relTol=0.0)
        xf = rls.solve_(f, bd).to_host()
        assert np.array_equal(x2, xf), (rho2, lam2, itv, rel(x2, xf))
        o = O.FISTA(A.astype(np.float64), reg=O.TVRegularization(lam2, shape=shape, iterationsTV=itv), rho=rho2, iterations=40, relTol=0.0)
        assert rel(x2, np.array(O.solve(o, b.astype(np.float64)))) < 1e-5
    assert not np.array_equal(x1, x2)
    # the regulariser KIND changes on one plan: L1's one-launch update, then TV's three launches, then L1 again
    s.reg = rls.L1Regularization(lam)
    xa = rls.solve_(s, bd).to_host()
    fa = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(lam), rho=rho, iterations=40, relTol=0.0)
    assert np.array_equal(xa, rls.solve_(fa, bd).to_host())
    s.reg = rls.TVRegularization(lam, shape=shape)
    assert np.array_equal(rls.solve_(s, bd).to_host(), x1)


# ---- the N > 1 path of bench.py, rehearsed on the one GPU of this box ---------------------------------------------------
@pytest.mark.parametrize("n", [2, 8])
def test_bench_multi_gpu_path_rehearsed_on_one_gpu(n):
    """`python bench.py --gpus N --rehearse`: the launcher, N ranks (all on device 0, gloo control plane), the headline leg with its
    max-over-ranks timing and solution check, `config4_batched` (matrix-free and on the explicit Gram matrix), `config5_rowsharded`
    (torch.distributed ranks + the one-process host through rls_comm_* on the direct transport) -- so that the driver's first
    8-GPU run is not the first execution of that code.  Semantics: src/MultiThreading.jl:30-79 (independent columns per GPU),
    SURVEY 8e.  Asserts ONE JSON line with n_gpus = N, per-rank rates, and no `error` anywhere in the two legs."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--rehearse", "--steps", "64", "--warmup", "32",
                        "--c5-rows", "8192"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == 64 and d["unit"] == "iterations/s" and d["value"] > 0 and "rehearsal" in d
    assert d["solution_check"]["rel_err_vs_float64_cgnr_same_iteration"] <= 1e-5
    assert d["n1_same_workload_value"] > 0

    def errors(o, path=""):
        if isinstance(o, dict):
            return [f"{path}/{k}: {v}" for k, v in o.items() if k == "error"] + [e for k, v in o.items() for e in errors(v, f"{path}/{k}")]
        return []

    c4, c5 = d["config4_batched"], d["config5_rowsharded"]
    assert not errors(c4) and not errors(c5), errors(c4) + errors(c5)
    assert len(c4["per_rank_solve_iterations_per_s_hip_events"]) == n and c4["value"] > 0 and c4["gram_mode"]["value"] > 0
    assert c5["value"] > 0 and c5["collective"]["world_size_seen_by_the_collective"] == n and np.isfinite(c5["residual"])
    direct = c5["one_process_host"]["direct"]
    assert direct["ranks"] == n and direct["iterations_per_s"] > 0 and np.isfinite(direct["residual"])
    assert direct["peer_probe"]["transport_in_use"] == 2 and direct["peer_probe"]["all_pairs"]
    assert "skipped" in c5["one_process_host"]["rccl"]


@pytest.mark.parametrize("where", ["config5:1", "config4_gram:0"])
def test_bench_multi_gpu_leg_failure_on_one_rank_keeps_the_line(where):
    """One rank failing inside the setup of a leg behind the headline (RLS_BENCH_FAIL=<leg>:<rank>) must not leave the other ranks
    in that leg's collectives nor cost the line: the ranks agree at the end of the setup (an all-reduced flag), skip the leg TOGETHER,
    the line prints with `error` naming the owner's rank, the legs after it still run, and the job exits 0.  (The first 8-GPU run of
    the driver is the first run of this code on more than one device; semantics src/MultiThreading.jl:30-79.)"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RLS_BENCH_FAIL=where, RLS_BENCH_DIST_TIMEOUT_S="60")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "64", "--warmup", "32",
                        "--c5-rows", "8192"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "distributed_error" not in d
    leg, owner = where.split(":")
    c4, c5 = d["config4_batched"], d["config5_rowsharded"]
    if leg == "config5":
        assert f"failed on rank {owner}" in c5["error"] and "injected failure" in r.stderr
        assert c4["value"] > 0 and c4["gram_mode"]["value"] > 0
    else:
        assert f"failed on rank {owner}" in c4["gram_mode"]["error"]
        assert c4["value"] > 0 and c5["value"] > 0 and np.isfinite(c5["residual"])  # the legs behind the failed one still ran


# ---- FISTA with a TV regulariser inside the plan, single GPU and row-sharded (SURVEY 8e last row) -----------------------------
@pytest.mark.parametrize("dt,shape,dims,proj", [(np.float32, (16, 16), None, "positive"), (np.complex64, (16, 16), None, "real"),
                                                (np.float32, (256,), None, None), (np.float32, (8, 8, 4), None, None),
                                                (np.float32, (16, 16), (2,), None)])
@pytest.mark.parametrize("restart", ["none", "gradient"])
def test_fista_tv_inside_the_plan(rls, ctx, dt, shape, dims, proj, restart):
    """`prox!(reg, x, rho * lambda)` with reg::TVRegularization (src/FISTA.jl:164 -> ProxTV.jl:64-125) followed by the
    projection (:166-168): the plan runs operator apply | gradient step | ONE FGP launch | projection, restart test, theta, y
    (rls_fista_set_reg_tv) instead of the primitive-by-primitive sequence; both against the float64 oracle, and each other."""
    N = int(np.prod(shape))
    A, xt, b = O.make_problem(3 * N, N, dt, 211)
    h64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    rho = 0.9 / np.linalg.norm(A.astype(h64), 2) ** 2
    lam = 0.02 * float(np.max(np.abs(A.astype(h64).conj().T @ b)))
    regs = lambda R: [R.TVRegularization(lam, shape=shape, dims=dims)] + (
        [R.PositiveRegularization()] if proj == "positive" else [R.RealRegularization()] if proj == "real" else [])
    kw = dict(rho=rho, iterations=25, relTol=0.0, restart=restart)
    x64, x32 = (lambda f: (f(A.astype(h64), b.astype(h64)), lambda: f(A, b)))(lambda A_, b_: np.array(O.solve(O.FISTA(A_, reg=regs(O), **kw), b_)))
    Ad = rls.DeviceMatrix.from_host(np.asfortranarray(A), ctx)
    s = rls.createLinearSolver(rls.FISTA, Ad, reg=regs(rls), **kw)
    x = rls.solve_(s, rls.DeviceVector.from_host(b, ctx)).to_host()
    assert s.state._plan, "the TV regulariser must run inside the plan for an image that fits one workgroup"
    assert s.state.iteration == 25
    parity(f"fista_tv_plan_{np.dtype(dt).name}_{shape}_{dims}_{proj}_{restart}", x, x64, x32)
    # the primitive-by-primitive sequence (the path of images too large for the FGP launch) must agree
    s2 = rls.createLinearSolver(rls.FISTA, Ad, reg=regs(rls), **kw)
    s2._tv_unfused = True
    x2 = rls.solve_(s2, rls.DeviceVector.from_host(b, ctx)).to_host()
    assert not s2.state._plan
    assert rel(x2, x) < 5e-6
    # callbacks cadence through the plan: one iterate at a time gives the same bits as the enqueued solve
    s3 = rls.createLinearSolver(rls.FISTA, Ad, reg=regs(rls), **kw)
    seen = []
    x3 = rls.solve_(s3, rls.DeviceVector.from_host(b, ctx), callbacks=[lambda sv, i: seen.append(i)]).to_host()
    assert seen == list(range(26)) and np.array_equal(x3, x)
    # early stop: the launches behind `done` (the FGP one included) are no-ops
    probe = O.FISTA(A.astype(h64), reg=regs(O), **kw)
    probe.init(b.astype(h64))
    rels = []
    while probe.iterate() is not None:
        rels.append(float(probe.rel_res_norm))
    k = 3   # a threshold first crossed at iteration k + 1 (the relative residual falls steeply over the first iterations)
    assert rels[k] < 0.95 * min(rels[:k])
    tol = float(np.sqrt(rels[k] * min(rels[:k])))
    s4 = rls.createLinearSolver(rls.FISTA, Ad, reg=regs(rls), rho=rho, iterations=25, relTol=tol, restart=restart)
    x4 = rls.solve_(s4, rls.DeviceVector.from_host(b, ctx)).to_host()
    o4 = O.FISTA(A.astype(h64), reg=regs(O), rho=rho, iterations=25, relTol=tol, restart=restart)
    x4o = np.array(O.solve(o4, b.astype(h64)))
    assert o4.iteration == k + 1 and s4.state.iteration == o4.iteration and rel(x4, x4o) < 1e-5


def test_fista_tv_image_too_large_for_the_plan_falls_back(rls, ctx):
    """a 3-D image of 4096 pixels does not fit the single-workgroup FGP kernel: rls_fista_set_reg_tv refuses and the solver runs
    from the primitives (same result as the oracle)"""
    shape = (16, 16, 16)
    N = 4096
    A, xt, b = O.make_problem(N + 512, N, np.float32, 5)
    # (an upper bound of sigma_max^2 from the Frobenius norm of a Gram block would do; the exact 2-norm of a 4608 x 4096 matrix was
    #  half of this test's 19 s: sigma_max of a Gaussian matrix is sqrt(M) + sqrt(N) to a per cent)
    rho = 0.9 / (math.sqrt(N + 512) + math.sqrt(N)) ** 2 / 1.05
    lam = 0.02 * float(np.max(np.abs(A.astype(np.float64).T @ b)))
    s = rls.createLinearSolver(rls.FISTA, rls.DeviceMatrix.from_host(np.asfortranarray(A), ctx), reg=rls.TVRegularization(lam, shape=shape),
                               rho=rho, iterations=6, relTol=0.0)
    x = rls.solve_(s, rls.DeviceVector.from_host(b, ctx)).to_host()
    assert not s.state._plan and s._tv_unfused
    x64 = np.array(O.solve(O.FISTA(A.astype(np.float64), reg=O.TVRegularization(lam, shape=shape), rho=rho, iterations=6, relTol=0.0), b.astype(np.float64)))
    parity("fista_tv_unfused_16x16x16", x, x64, lambda: np.array(O.solve(O.FISTA(A, reg=O.TVRegularization(lam, shape=shape), rho=rho, iterations=6, relTol=0.0), b)))


@pytest.mark.parametrize("nshards,proj", [(2, "positive"), (8, None)])
def test_rowsharded_fista_tv_through_the_library_communicator(rls, ctx, nshards, proj):
    """the config-5 pattern for FISTA with TV + Positive (SURVEY 8e last row; src/FISTA.jl:114,152,164-168): the operator apply is
    the only distributed step, the gradient step, the FGP launch and the projection are replicated on every rank"""
    from test_gpu_parity import _row_cuts

    shape, N = (32, 16), 512
    M = 1536 + 64 * nshards
    A, xt, b = O.make_problem(M, N, np.float32, 139)
    A64, b64 = A.astype(np.float64), b.astype(np.float64)
    rho = 0.95 / np.linalg.norm(A64, 2) ** 2
    lam = 2e-2 * float(np.max(np.abs(A64.T @ b64)))
    cuts = _row_cuts(M, nshards)
    shards = [np.asfortranarray(A[cuts[k]:cuts[k + 1]]) for k in range(nshards)]
    parts = [b[cuts[k]:cuts[k + 1]] for k in range(nshards)]
    regs = lambda R: [R.TVRegularization(lam, shape=shape)] + ([R.PositiveRegularization()] if proj else [])
    s = rls.CommRowShardedFISTA(rls, shards, reg=rls.TVRegularization(lam, shape=shape), proj=rls.PositiveRegularization() if proj else None,
                                transport=2, rho=rho, iterations=20, restart="gradient")
    try:
        x = s.solve(parts)
        xs = [s.solution(r) for r in range(nshards)]
        assert all(np.array_equal(xs[0], xr) for xr in xs[1:])   # replicated state: the same bits on every rank
        ref = O.FISTA(A64, reg=regs(O), rho=rho, iterations=20, restart="gradient")
        x64 = np.array(O.solve(ref, b64))
        assert all(s.status(r)["iteration"] == ref.iteration for r in range(nshards))
        parity(f"rowsharded_fista_tv_comm_{nshards}_shards", x, x64,
               lambda: np.array(O.solve(O.FISTA(A, reg=regs(O), rho=rho, iterations=20, restart="gradient"), b)))
    finally:
        s.close()
    with pytest.raises(NotImplementedError, match="does not fit"):
        rls.CommRowShardedFISTA(rls, [np.asfortranarray(np.zeros((64, 4096), np.float32))] * 2, reg=rls.TVRegularization(0.1, shape=(16, 16, 16)),
                                transport=2)


def test_row_sharded_fista_tv_two_shards_torch_ops(rls, ctx):
    """the same through the one-process-per-GPU host's local ops (HipFistaOps: torch tensors, the collective stood in for)"""
    import torch
    from test_gpu_parity import _TwoShards

    shape, N, M = (16, 16), 256, 768
    A, xt, b = O.make_problem(M, N, np.complex64, 23)
    rho = 0.9 / np.linalg.norm(A.astype(np.complex128), 2) ** 2
    lam = 0.05 * float(np.max(np.abs(A.conj().T @ b)))
    dev = torch.cuda.current_device()
    lo = 388
    mk = lambda rows: rls.multigpu.HipFistaOps(rls, np.asfortranarray(A[rows]), dev, reg=rls.TVRegularization(lam, shape=shape),
                                               proj=rls.RealRegularization())
    pair = _TwoShards(mk(slice(0, lo)), mk(slice(lo, M)))
    f = rls.RowShardedFISTA(pair, _TwoShards.Dist, rho=rho, iterations=20, relTol=0.0)
    f.init((b[:lo], b[lo:]))
    f.step(20)
    xs = [s.solution() for s in pair.shards]
    assert np.array_equal(xs[0], xs[1])
    regs = [O.TVRegularization(lam, shape=shape), O.RealRegularization()]
    x64 = np.array(O.solve(O.FISTA(A.astype(np.complex128), reg=regs, rho=rho, iterations=20, relTol=0.0), b.astype(np.complex128)))
    parity("fista_tv_rowsharded_2shards_torch_ops", xs[0], x64, lambda: np.array(O.solve(O.FISTA(A, reg=regs, rho=rho, iterations=20, relTol=0.0), b)))
    for s in pair.shards:
        s.close()


@pytest.mark.parametrize("solver", ["ADMM", "SplitBregman"])
@pytest.mark.parametrize("dt", [np.complex64, np.float32])
def test_admm_inner_cg_with_a_preconditioner(rls, ctx, solver, dt):
    """the `precon` keyword (src/ADMM.jl:82, src/SplitBregman.jl:84) reaches the inner cg! as `Pl` (:244 / :218): a Jacobi
    preconditioner on a column-scaled operator, against the oracle's preconditioned cg! (pinned to SciPy's PCG on the CPU) --
    iterates, inner iteration counts -- and it must beat the unpreconditioned solve at the same iteration budget"""
    M, N = 512, 192
    A, xt, b = O.make_problem(M, N, dt, 71)
    h64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    scale = (10.0 ** np.random.default_rng(3).uniform(-1, 1, N)).astype(np.float32)
    A = np.asfortranarray(A * scale[None, :]).astype(dt)
    b = (A.astype(h64) @ (xt / scale)).astype(dt)
    rho = 0.05
    d = (np.sum(np.abs(A.astype(h64)) ** 2, axis=0) + rho).astype(np.float64)   # diag(A'A) + rho
    kw = dict(rho=rho, iterations=6, iterationsCG=6, tolInner=1e-6, absTol=0.0, relTol=0.0)
    extra = dict(iterationsInner=3) if solver == "SplitBregman" else {}
    kw["iterations"] = 2 if solver == "SplitBregman" else 6
    ref = getattr(O, solver)(A.astype(h64), reg=O.L1Regularization(1e-3), precon=lambda r: r / d, **kw, **extra)
    x64 = np.array(O.solve(ref, b.astype(h64)))
    Ad = rls.DeviceMatrix.from_host(A, ctx)
    pre = rls.DiagonalPreconditioner(rls.DeviceVector.from_host(d.astype(dt), ctx))
    s = rls.createLinearSolver(getattr(rls, solver), Ad, reg=rls.L1Regularization(1e-3), precon=pre, **kw, **extra)
    x = rls.solve_(s, rls.DeviceVector.from_host(b, ctx)).to_host()
    parity(f"{solver}_precon_{np.dtype(dt).name}", x, x64,
           lambda: np.array(O.solve(getattr(O, solver)(A, reg=O.L1Regularization(1e-3), precon=lambda r: (r / d).astype(dt), **kw, **extra), b)))
    if solver == "ADMM":
        assert list(s.state.cg_iterations) == list(ref.cg_iters)
    plain = rls.createLinearSolver(getattr(rls, solver), Ad, reg=rls.L1Regularization(1e-3), **kw, **extra)
    xp = rls.solve_(plain, rls.DeviceVector.from_host(b, ctx)).to_host()
    want = xt / scale
    assert rel(x, want) < 0.7 * rel(xp, want)
    with pytest.raises(TypeError, match="ldiv_"):
        rls.createLinearSolver(rls.ADMM, Ad, precon=object())


@pytest.mark.parametrize("workload", ["rowsharded", "config4"])
def test_bench_other_multi_gpu_workloads_rehearsed(workload):
    """the two non-default N > 1 lines (`--workload rowsharded`: BASELINE configs[4], strong scaling, one all-reduce per iteration
    through torch.distributed; `--workload config4`: configs[3], 8 right-hand sides per GPU) through the same one-GPU rehearsal"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse", "--workload", workload, "--steps", "64",
                        "--warmup", "32", "--c5-rows", "8192"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["steps"] == 64
    if workload == "rowsharded":
        assert d["scaling"] == "strong" and d["config"]["collective"]["world_size_seen_by_the_collective"] == 2 and np.isfinite(d["residual"])
        assert d["config"]["rows_per_gpu"] == 4096
    else:
        assert d["scaling"] == "weak" and len(d["per_rank_solve_iterations_per_s_hip_events"]) == 2 and d["n1_same_workload_value"] > 0


# ---- test/testSolvers.jl:3-65 and :222-239 exactly as the reference runs them: 3 x 2 systems with `rand` entries ----------------
@pytest.mark.parametrize("seed", [12345, 1, 2])
def test_reference_small_system_suite_on_device(rls, ctx, seed, capsys):
    """testRealLinearSolver / testComplexLinearSolver / testComplexLinearAHASolver (A = rand(3, 2), every solver of
    linearSolverList(), iterations = 200 / 100, the constructors' defaults otherwise, `x_approx ≈ x rtol = 0.1`) and
    testVerboseSolvers (verbose = true, 3 iterations: must not throw) through createLinearSolver on the device"""
    rng = np.random.default_rng(seed)
    solvers = [s for s in rls.linearSolverList()]
    assert {s.__name__ for s in solvers} >= {"CGNR", "Kaczmarz", "FISTA", "OptISTA", "POGM", "ADMM", "SplitBregman"}
    # real (:3-22)
    A = rng.random((3, 2)).astype(np.float32)
    x = rng.random(2).astype(np.float32)
    b = A @ x
    for S in solvers:
        sol = rls.createLinearSolver(S, rls.DeviceMatrix.from_host(np.asfortranarray(A), ctx), iterations=200)
        xa = rls.solve_(sol, rls.DeviceVector.from_host(b, ctx)).to_host()
        assert rel(xa, x) < 0.1, ("real", S.__name__, rel(xa, x))
    # complex (:24-43)
    Ac = (rng.random((3, 2)) + 1j * rng.random((3, 2))).astype(np.complex64)
    xc = (rng.random(2) + 1j * rng.random(2)).astype(np.complex64)
    bc = Ac @ xc
    for S in solvers:
        sol = rls.createLinearSolver(S, rls.DeviceMatrix.from_host(np.asfortranarray(Ac), ctx), iterations=100)
        xa = rls.solve_(sol, rls.DeviceVector.from_host(bc, ctx)).to_host()
        assert rel(xa, xc) < 0.1, ("complex", S.__name__, rel(xa, xc))
    # AHA only (:45-65): solver(nothing; AHA = A'A), b = AHA x
    AHA = (Ac.conj().T @ Ac).astype(np.complex64)
    bh = AHA @ xc
    for S in solvers:
        if S.__name__ == "Kaczmarz":
            continue   # filtered out by the reference as well (:51)
        sol = rls.createLinearSolver(S, None, AHA=rls.DeviceMatrix.from_host(np.asfortranarray(AHA), ctx), iterations=100)
        xa = rls.solve_(sol, rls.DeviceVector.from_host(bh, ctx)).to_host()
        assert rel(xa, xc) < 0.1, ("AHA", S.__name__, rel(xa, xc))
    # verbose (:222-239)
    for name in ("ADMM", "FISTA", "POGM", "OptISTA", "SplitBregman"):
        sol = rls.createLinearSolver(getattr(rls, name), rls.DeviceMatrix.from_host(np.asfortranarray(A), ctx), iterations=3, verbose=True)
        rls.solve_(sol, rls.DeviceVector.from_host(b, ctx))
    capsys.readouterr()


def test_fista_resident_deferred_norm_changes_no_bit(rls, ctx):
    """round 5, lever (d): `fista_resident_kernel` sums ||res||^2 off the critical path (behind the NEXT iteration's exchange) when
    there is no gradient restart; a stop found there drops that iteration's exchange.  Against `fista_defer = 0` (the block
    reduction in place): the same bits of x, xold, res, the same iteration count and residual norm -- for a full solve, for a
    relTol that stops in the middle of a launch, and for step calls of 1 / 3 / 7 iterations (server mode included)."""
    import ctypes as C
    M, N = 4096, 2048
    A, xt, b = O.make_problem(M, N, np.complex64, 77)
    Ad, bd = rls.DeviceMatrix.from_host(np.asfortranarray(A), ctx), rls.DeviceVector.from_host(b, ctx)
    rho = 0.9 / (np.sqrt(M) + np.sqrt(N)) ** 2
    lam = 0.02 * float(np.max(np.abs(A.conj().T @ b)))
    path = C.c_int32(-1)

    def solve(defer, **kw):
        ctx.tune(fista_defer=defer)
        try:
            s = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(lam), rho=rho, **kw)
            x = rls.solve_(s, bd).to_host()
            ctx.lib.rls_fista_path(s.state._plan, C.byref(path))
            return x, s.state.xold.to_host(), s.state.res.to_host(), s.state.iteration, s.state.rel_res_norm
        finally:
            ctx.tune(fista_defer=1)

    # the oracle says where a relTol crossing lies
    o = O.FISTA(A.astype(np.complex128), reg=O.L1Regularization(lam), rho=rho, iterations=40, relTol=0.0)
    o.init(b.astype(np.complex128))
    rels = []
    while o.iterate() is not None:
        rels.append(float(o.rel_res_norm))
    k = max(i for i in range(3, 13) if rels[i] < 0.97 * min(rels[:i]))   # the last clear crossing among the first iterations
    tol = float(np.sqrt(rels[k] * min(rels[:k])))
    for kw in (dict(iterations=40, relTol=0.0), dict(iterations=40, relTol=tol), dict(iterations=1, relTol=0.0)):
        a, b_ = solve(1, **kw), solve(0, **kw)
        if torch_device_has_256_cus():
            assert path.value == 4, path.value
        assert a[3] == b_[3] and (kw["relTol"] == 0.0 or a[3] == k + 1), (kw, a[3], b_[3])
        assert np.array_equal(a[0], b_[0]) and np.array_equal(a[1], b_[1]) and np.array_equal(a[2], b_[2]), kw
        assert a[4] == b_[4]
    x64 = np.array(O.solve(O.FISTA(A.astype(np.complex128), reg=O.L1Regularization(lam), rho=rho, iterations=40, relTol=tol), b.astype(np.complex128)))
    parity("fista_resident_deferred_norm_reltol_stop", solve(1, iterations=40, relTol=tol)[0], x64,
           lambda: np.array(O.solve(O.FISTA(A, reg=O.L1Regularization(lam), rho=rho, iterations=40, relTol=tol), b)))
    # iterate-by-iterate with callbacks (server mode) against the enqueued solve: the same bits
    s = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(lam), rho=rho, iterations=40, relTol=tol)
    seen = []
    xs = rls.solve_(s, bd, callbacks=[lambda sv, i: seen.append(i)]).to_host()
    assert seen == list(range(k + 2)) and np.array_equal(xs, solve(1, iterations=40, relTol=tol)[0])


def torch_device_has_256_cus():
    import torch
    return torch.cuda.get_device_properties(0).multi_processor_count >= 256
