"""CPU tests of the host-side logic that needs no device: kwarg filtering, partitioning, TV geometry."""
import warnings

import numpy as np
import pytest


def test_filter_kwargs_warns_like_the_reference(rls):
    from rls_amd.solvers import _filter_kwargs

    with pytest.warns(UserWarning, match="filtered out: shape"):
        kept = _filter_kwargs(rls.CGNR, True, dict(iterations=3, shape=(2, 2)))
    assert kept == {"iterations": 3}
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert _filter_kwargs(rls.FISTA, False, dict(rho=0.1, bogus=1)) == {"rho": 0.1}
    assert rls.linearSolverList() == [rls.CGNR, rls.Kaczmarz, rls.FISTA, rls.OptISTA, rls.POGM, rls.ADMM, rls.SplitBregman]
    # isapplicable / applicableSolverList (src/RegularizedLeastSquares.jl:223-265)
    l1, l2, pos = rls.L1Regularization(0.1), rls.L2Regularization(0.1), rls.PositiveRegularization()
    assert rls.isapplicable(rls.FISTA, [l1]) and rls.isapplicable(rls.FISTA, l1) and not rls.isapplicable(rls.FISTA, [l1, l2])
    assert rls.isapplicable(rls.FISTA, [l1, pos])
    assert rls.isapplicable(rls.Kaczmarz, [l2]) and rls.isapplicable(rls.Kaczmarz, [l2, l1]) and not rls.isapplicable(rls.Kaczmarz, [l1])
    assert rls.isapplicable(rls.ADMM, [l1, l2]) and not rls.isapplicable(rls.CGNR, [l2])  # the Krylov category has no rule
    assert rls.applicableSolverList([l1]) == [rls.FISTA, rls.OptISTA, rls.POGM, rls.ADMM, rls.SplitBregman]
    assert rls.isapplicable(rls.FISTA, None, None, [l1]) and issubclass(rls.SplitBregman, rls.AbstractPrimalDualSolver)


def test_shard_columns_covers_every_column_once(rls):
    for n, w in ((64, 8), (10, 4), (3, 8), (0, 2)):
        seen = []
        for r in range(w):
            seen += list(rls.shard_columns(n, w, r))
        assert seen == list(range(n))
    assert list(rls.shard_columns(64, 8, 3)) == list(range(24, 32))  # columns 8k..8k+7 on GPU k
    with pytest.raises(ValueError):
        rls.shard_columns(4, 2, 2)


def test_shard_rows_is_aligned_and_complete(rls):
    for M, w in ((65536, 8), (1000, 3), (10, 4)):
        blocks = [rls.shard_rows(M, w, r) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == M
        for (lo, hi), (lo2, _) in zip(blocks, blocks[1:]):
            assert hi == lo2 and lo % 4 == 0
    assert rls.shard_rows(65536, 8, 2) == (16384, 24576)


def test_tv_geometry_and_regularization_plumbing(rls):
    from rls_amd.regularization import _tv_geometry, lam, normalize

    shape, d0, _, _ = _tv_geometry((8, 6), None)
    assert shape == (8, 6) and d0 == (0, 1)
    assert _tv_geometry((8, 6), 2)[1] == (1,) and _tv_geometry((8, 6), (2, 1))[1] == (1, 0)
    assert lam(rls.L1Regularization(0.25)) == 0.25 and lam(rls.PositiveRegularization()) is None
    regs = [rls.L2Regularization(1.0)]
    out = normalize(rls.NoNormalization(), regs)
    assert len(out) == len(regs) and all(a is b for a, b in zip(out, regs))
    # MeasurementBasedNormalization without b: factor one (NormalizedRegularization.jl:44); projections are never scaled
    nr = normalize(rls.MeasurementBasedNormalization(), [rls.L1Regularization(0.5), rls.PositiveRegularization()])
    assert isinstance(nr[0], rls.L1Regularization) and nr[0].lam == 0.5 and rls.scalefactor(nr[0]) == 1.0
    assert isinstance(nr[1], rls.PositiveRegularization)
    scaled = rls.NormalizedRegularization(rls.L1Regularization(0.5), 4.0)
    assert scaled.lam == 2.0 and rls.innerreg(scaled).lam == 0.5
    assert rls.NormalizedRegularization(scaled, 3.0).lam == 1.5  # update, not compound (:73)
    with pytest.raises(ValueError):
        normalize(rls.SystemMatrixBasedNormalization(), regs)
    assert rls.TVRegularization(0.1, shape=(4, 4)).iterationsTV == 10  # ctor default, ProxTV.jl:39


def test_dtype_guard(rls):
    from rls_amd.arrays import dtype_code

    assert dtype_code(np.float32) == 0 and dtype_code(np.complex64) == 1
    assert dtype_code(np.float64) == 2 and dtype_code(np.complex128) == 3   # round 6: the L1 protocol with double scalars (rls_*_d)
    for bad in (np.float16, np.int32, np.complex256 if hasattr(np, "complex256") else np.int8):
        with pytest.raises(TypeError, match="Float32 / ComplexF32"):
            dtype_code(bad)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` from a plain shell (no RANK / WORLD_SIZE) must start torch.distributed.run itself, as a
    CHILD process and before torch or HIP is touched, relay the child's exit code, and stay launchable under an external
    torchrun (the first 8-GPU node would otherwise have recorded a usage message)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench, "--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-launch"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = d["launch"]
    assert d["torch_imported"] is False  # the parent decides before importing torch (nothing of it may initialise the GPU)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    tail = cmd[cmd.index(bench):]
    assert tail == [bench, "--gpus", "8", "--steps", "20", "--warmup", "5"]  # the ranks get the same arguments, minus --dry-launch
    # a rank (or N = 1) does not launch anything
    for extra_env, args in (({"WORLD_SIZE": "8", "RANK": "3"}, ["--gpus", "8"]), ({}, ["--gpus", "1"]), ({}, [])):
        r = subprocess.run([sys.executable, bench, *args, "--dry-launch"], env={**env, **extra_env}, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["launch"] is None
    # exit code of the ranks comes back: without a GPU every rank stops with "bench.py needs an MI355X"
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "4", "--warmup", "1"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0
    assert "needs an MI355X" in (r.stderr + r.stdout)
