"""world_size-2 gloo tests (CPU) of the N > 1 control flow: row-sharded CGNR's collective schedule and
the column sharding of matrix solves.  The rank-local arithmetic is supplied by the oracle through
the same local-ops protocol the GPU implementation (multigpu.HipLocalOps) satisfies; what is under
test is the distributed logic of RowShardedCGNR / shard_rows / shard_columns."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleLocalOps:
    """NumPy stand-in for HipLocalOps: same half-steps, same replicated state, torch CPU tensors"""

    def __init__(self, A_local):
        import torch
        import rls_oracle as O

        self.O, self.torch = O, torch
        self.A = np.asarray(A_local)
        n = self.A.shape[1]
        self.t = {k: torch.zeros(n, dtype=torch.complex128) for k in ("x", "r", "p", "v")}

    def _np(self, k):
        return self.t[k].numpy()

    def init_a(self, b_local, lam, rel_tol, iterations):
        self.lam, self.rel_tol, self.max_iter = lam, rel_tol, min(iterations, self.A.shape[1])
        self._np("r")[:] = self.A.conj().T @ b_local

    def init_b(self):
        r = self._np("r")
        self._np("x")[:] = 0
        self._np("v")[:] = 0
        self._np("p")[:] = r
        self.rr = float(np.vdot(r, r).real)
        self.z0 = np.sqrt(self.rr)
        self.iteration = 0
        self.done = self.max_iter <= 0

    def step_a(self):
        if self.done:
            return
        p = self._np("p")
        self._np("v")[:] = self.A.conj().T @ (self.A @ p)

    def step_b(self):
        if self.done:
            return
        x, r, p, v = (self._np(k) for k in ("x", "r", "p", "v"))
        zeta = self.rr
        alpha = zeta / (np.vdot(p, v) + self.lam * np.vdot(p, p).real)
        x += alpha * p
        r -= alpha * v
        r -= self.lam * alpha * p
        self.rr = float(np.vdot(r, r).real)
        p *= self.rr / zeta
        p += r
        self.iteration += 1
        self.done = np.sqrt(self.rr) / self.z0 <= self.rel_tol or self.iteration >= self.max_iter

    def tensor(self, name):
        return self.t[name]

    def solution(self):
        return self._np("x").copy()

    def sync(self):
        pass


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import rls_amd as rls
    import rls_oracle as O

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        M, N = 96, 24
        A, xt, b = O.make_problem(M, N, np.complex128, 5)
        lo, hi = rls.shard_rows(M, world, rank, align=2)
        ops = OracleLocalOps(A[lo:hi])
        s = rls.RowShardedCGNR(ops, dist, lam=0.1, iterations=12, relTol=0.0)
        x = s.solve(b[lo:hi])
        ref = O.CGNR(A, reg=O.L2Regularization(0.1), iterations=12, relTol=0.0)
        O.solve(ref, b)
        err = float(np.linalg.norm(x - ref.x) / np.linalg.norm(ref.x))
        # replicated state must be bit-identical across ranks (no scalar collectives needed)
        gathered = [None] * world
        dist.all_gather_object(gathered, (x.tobytes(), ops.iteration, ops.rr))
        same = all(g == gathered[0] for g in gathered)
        cols = list(rls.shard_columns(7, world, rank))
        q.put((rank, err, same, ops.iteration, cols))
    finally:
        dist.destroy_process_group()


def test_row_sharded_cgnr_two_ranks_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, same, it, cols in res:
        assert err < 1e-10, err      # sharded == unsharded CGNR
        assert same and it == 12
    assert res[0][4] + res[1][4] == list(range(7))


def test_row_sharded_single_rank_equals_plain_cgnr():
    sys.path.insert(0, ROOT)
    import rls_amd as rls
    import rls_oracle as O

    A, xt, b = O.make_problem(40, 10, np.complex128, 8)
    s = rls.RowShardedCGNR(OracleLocalOps(A), None, lam=0.0, iterations=10, relTol=0.0)
    ref = O.CGNR(A, iterations=10, relTol=0.0)
    assert np.linalg.norm(s.solve(b) - O.solve(ref, b)) < 1e-10 * np.linalg.norm(ref.x)


# ---------------------------------------------------------------------------------------------
# row-sharded FISTA / ADMM: same protocol idea, NumPy stand-ins for HipFistaOps / HipAdmmOps
# ---------------------------------------------------------------------------------------------


class OracleFistaOps:
    """half-steps of src/FISTA.jl:139-185 in float64 (mirrors rls_fista_*_local_a/b)"""

    def __init__(self, A_local, reg, proj=()):
        import torch

        self.A, self.reg, self.proj = np.asarray(A_local), reg, tuple(proj)
        n = self.A.shape[1]
        self.t = {k: torch.zeros(n, dtype=torch.complex128) for k in ("x", "x0", "xold", "res", "y")}

    def _np(self, k):
        return self.t[k].numpy()

    def init_a(self, b_local):
        self._np("x0")[:] = self.A.conj().T @ b_local

    def init_b(self, rho, theta, rel_tol, iterations, restart_gradient):
        self.rho, self.theta, self.theta_old = rho, theta, theta
        self.rel_tol, self.max_iter, self.restart = rel_tol, iterations, restart_gradient
        self.norm_x0 = np.linalg.norm(self._np("x0"))
        for k in ("x", "xold"):
            self._np(k)[:] = 0
        self._np("y")[:] = 0  # extrapolated point of iteration 1 = x
        self.rel_res_norm, self.iteration = np.inf, 0

    @property
    def done(self):
        return self.rel_res_norm < self.rel_tol or self.iteration >= self.max_iter

    def step_a(self):
        if self.done:
            return
        y = self._np("y")
        self._np("res")[:] = self.A.conj().T @ (self.A @ y)

    def step_b(self):
        if self.done:
            return
        x, xold, res, y, x0 = (self._np(k) for k in ("x", "xold", "res", "y", "x0"))
        res -= x0
        xn = y - self.rho * res
        self.rel_res_norm = np.linalg.norm(res) / self.norm_x0
        self.reg.prox(xn, self.rho * self.reg.lam)
        for pr in self.proj:   # src/FISTA.jl:166-168: the projections follow the prox
            pr.prox(xn)
        if self.restart and np.real(np.vdot(res, xn - x)) > 0:
            self.theta = 1.0
        self.theta_old = self.theta
        self.theta = (1 + np.sqrt(1 + 4 * self.theta_old ** 2)) / 2
        xold[:] = x
        x[:] = xn
        # next extrapolated point (:147-150 of the NEXT iteration)
        y[:] = (1 - self.theta_old) / self.theta * xold + ((self.theta_old - 1) / self.theta + 1) * x
        self.iteration += 1

    def tensor(self, name):
        return self.t[name]

    def solution(self):
        return self._np("x").copy()


class OracleAdmmOps:
    """steps of src/ADMM.jl:191-309 + cg! in float64 (mirrors HipAdmmOps)"""

    def __init__(self, A_local, reg):
        import torch

        self.A, self.reg = np.asarray(A_local), reg
        n = self.A.shape[1]
        names = ("x", "xold", "beta", "beta_y", "z", "zold", "u", "cg_u", "cg_r", "cg_c")
        self.t = {k: torch.zeros(n, dtype=torch.complex128) for k in names}

    def _np(self, k):
        return self.t[k].numpy()

    def init_a(self, b_local):
        self._np("beta_y")[:] = self.A.conj().T @ b_local

    def init_b(self):
        for k in ("x", "z", "u"):
            self._np(k)[:] = 0

    def pre(self, rho):
        self._np("beta")[:] = self._np("beta_y") + rho * (self._np("z") - self._np("u"))
        self._np("xold")[:] = self._np("x")

    def apply_x(self):
        self._np("cg_c")[:] = self.A.conj().T @ (self.A @ self._np("x"))

    def cg_start(self, rho, maxiter, reltol):
        x, b, u, r, c = (self._np(k) for k in ("x", "beta", "cg_u", "cg_r", "cg_c"))
        self.cg_rho = rho
        r[:] = b - (c + rho * x)
        u[:] = r
        self.residual, self.prev = np.linalg.norm(r), 1.0
        self.tol = max(reltol * self.residual, 0.0)
        self.cg_it, self.cg_max = 0, maxiter
        self.cg_done = maxiter <= 0 or self.residual <= self.tol

    def apply_u(self):
        if self.cg_done:
            return
        self._np("cg_c")[:] = self.A.conj().T @ (self.A @ self._np("cg_u"))

    def cg_update(self):
        if self.cg_done:
            return
        x, u, r, c = (self._np(k) for k in ("x", "cg_u", "cg_r", "cg_c"))
        c += self.cg_rho * u
        alpha = self.residual ** 2 / np.vdot(u, c)
        x += alpha * u
        r -= alpha * c
        self.prev, self.residual = self.residual, np.linalg.norm(r)
        self.cg_it += 1
        self.cg_done = self.cg_it >= self.cg_max or self.residual <= self.tol
        if not self.cg_done:
            u[:] = r + (self.residual ** 2 / self.prev ** 2) * u

    def cg_iterations(self):
        return self.cg_it

    def post(self, prox_lambda):
        self.t["z"], self.t["zold"] = self.t["zold"], self.t["z"]
        x, xold, z, zold, u = (self._np(k) for k in ("x", "xold", "z", "zold", "u"))
        z[:] = x + u
        if prox_lambda is not None:
            self.reg.prox(z, prox_lambda)
        un = u + x - z
        nrm = np.linalg.norm
        rec = [nrm(x - xold) + nrm(z - zold) + nrm(un - u), nrm(z - zold), max(nrm(x), nrm(z)), nrm(x - z), nrm(un),
               nrm(x - xold)]
        u[:] = un
        return rec

    def tensor(self, name):
        return self.t[name]

    def solution(self):
        return self._np("x").copy()


def _worker_fista_admm(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import rls_amd as rls
    import rls_oracle as O

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        M, N = 90, 20
        A, xt, b = O.make_problem(M, N, np.complex128, 11)
        lo, hi = rls.shard_rows(M, world, rank, align=2)
        rho = 0.9 / np.linalg.norm(A, 2) ** 2
        f = rls.RowShardedFISTA(OracleFistaOps(A[lo:hi], O.L1Regularization(0.3)), dist, rho=rho, iterations=15, relTol=0.0,
                                restart="gradient")
        xf = f.solve(b[lo:hi])
        ref = O.FISTA(A, reg=O.L1Regularization(0.3), rho=rho, iterations=15, relTol=0.0, restart="gradient")
        O.solve(ref, b)
        err_f = float(np.linalg.norm(xf - ref.x) / np.linalg.norm(ref.x))
        # TV + Positive (SURVEY 8e last row; src/FISTA.jl:164-168): prox and projection are replicated, only A^H A y is summed
        tv = lambda: O.TVRegularization(0.2, shape=(5, 4))
        ft = rls.RowShardedFISTA(OracleFistaOps(A[lo:hi], tv(), proj=[O.PositiveRegularization()]), dist, rho=rho, iterations=15,
                                 relTol=0.0)
        xt_ = ft.solve(b[lo:hi])
        reft = O.FISTA(A, reg=[tv(), O.PositiveRegularization()], rho=rho, iterations=15, relTol=0.0)
        O.solve(reft, b)
        err_f = max(err_f, float(np.linalg.norm(xt_ - reft.x) / np.linalg.norm(reft.x)))
        kw = dict(rho=0.4, iterations=8, iterationsCG=5, tolInner=1e-3)
        a = rls.RowShardedADMM(OracleAdmmOps(A[lo:hi], O.L1Regularization(0.2)), dist, lam=0.2, **kw)
        xa = a.solve(b[lo:hi], M)
        refa = O.ADMM(A, reg=O.L1Regularization(0.2), **kw)
        O.solve(refa, b)
        err_a = float(np.linalg.norm(xa - refa.x) / np.linalg.norm(refa.x))
        gathered = [None] * world
        dist.all_gather_object(gathered, (xf.tobytes(), xa.tobytes(), a.iteration, tuple(a.cg_iterations)))
        same = all(g == gathered[0] for g in gathered)
        q.put((rank, err_f, err_a, same, a.iteration == refa.iteration, a.cg_iterations == refa.cg_iters))
    finally:
        dist.destroy_process_group()


def test_row_sharded_fista_and_admm_two_ranks_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_fista_admm, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err_f, err_a, same, it_ok, cg_ok in res:
        assert err_f < 1e-10, err_f   # sharded == unsharded FISTA
        assert err_a < 1e-6, err_a    # oracle ADMM computes in the dtype of A with Float32-typed tolerances
        assert same and it_ok and cg_ok


# ---------------------------------------------------------------------------------------------
# bench.py --rehearse: the gloo proxy the rehearsal puts in front of torch.distributed
# ---------------------------------------------------------------------------------------------


def _staged_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        d = bench.HostStagedDist(dist)
        v = torch.tensor([1.0 + 2.0j, -3.0j], dtype=torch.complex64) * (rank + 1)   # the row-sharded vectors are complex
        d.all_reduce(v, op=d.ReduceOp.SUM)
        t = torch.tensor([float(rank)], dtype=torch.float64)
        d.all_reduce(t, op=d.ReduceOp.MAX)
        outs = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        d.all_gather(outs, torch.tensor([10.0 + rank], dtype=torch.float64))
        d.barrier()
        q.put((rank, v.numpy().tolist(), float(t.item()), [float(o.item()) for o in outs], d.get_world_size(), d.get_backend()))
    finally:
        d.destroy_process_group()


def test_rehearsal_proxy_over_gloo_two_ranks():
    """`bench.py --gpus N --rehearse` (the N > 1 code path on one GPU) replaces torch.distributed by HostStagedDist: the calls
    the bench makes -- all_reduce SUM of complex vectors, MAX of the elapsed time, all_gather of the per-rank rates, barrier --
    over gloo with world size 2; and the flag reaches the ranks the parent launches"""
    import json
    import subprocess
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_staged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, v, mx, outs, ws, backend in res:
        assert v == [3.0 + 6.0j, -9.0j] and mx == 1.0 and outs == [10.0, 11.0] and ws == 2 and backend.startswith("gloo")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--rehearse", "--c5-rows", "8192", "--dry-launch"],
                       env=env, capture_output=True, text=True, timeout=120)
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert "--rehearse" in cmd and cmd[cmd.index("--c5-rows") + 1] == "8192" and "--nproc-per-node=8" in cmd
