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
