"""Multi-GPU operation, one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).

Two modes, both new designs (the reference has no multi-process code):

  * shard_columns / MultiSolve  -- BASELINE config 4: the independent right-hand sides of a matrix
    solve (src/MultiThreading.jl:30-79 semantics) are sharded one contiguous block per rank.  No
    data-path collective; only the final gather of the N x K solution.
  * RowShardedCGNR              -- BASELINE config 5: one tall A row-partitioned, rank g holding rows
    [g*M/P, (g+1)*M/P) repacked contiguous.  x, r, p, v and every scalar are replicated; the only
    exchange is ONE all-reduce(sum) of the N-vector A_g^H t_g per iteration (plus one at init for
    A^H b).  Scalars stay consistent without communication because every rank sums identical
    all-reduced vectors in the same order.

The distributed control flow is written against a small "local ops" protocol so that it can be
exercised on CPU with the gloo backend (tests/test_multigpu_gloo.py supply a NumPy implementation of
the protocol); the product implementation below, HipLocalOps, drives the C ABI and has no fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import time
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ._lib import CgnrStatus, check


# --------------------------------------------------------------------------------------------
# partitioning helpers (pure functions)
# --------------------------------------------------------------------------------------------


def shard_columns(n_cols: int, world: int, rank: int) -> range:
    """Columns k..k+per-1 on rank r = k // per ('columns 8k..8k+7 on GPU k').
    The first n_cols % world ranks take one extra column."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(n_cols, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def shard_rows(M: int, world: int, rank: int, align: int = 4) -> Tuple[int, int]:
    """Row block [lo, hi) of rank `rank`; block boundaries are multiples of `align` rows so that every
    shard keeps 16-byte aligned columns (4 Float32 / 2 ComplexF32 per 16 bytes)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    per = -(-M // world)
    per = -(-per // align) * align
    lo = min(rank * per, M)
    hi = min(lo + per, M)
    return lo, hi


# --------------------------------------------------------------------------------------------
# config 4: independent solves, sharded by column
# --------------------------------------------------------------------------------------------


class MultiSolve:
    """solve!(solver, B; scheduler = MultiThreadingState) across ranks: every rank owns a replica of A
    (or its own A) and the columns shard_columns() gives it."""

    def __init__(self, rls, solver_factory, dist=None):
        self.rls = rls
        self.solver_factory = solver_factory
        self.dist = dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1

    def solve(self, B_host: np.ndarray, ctx=None) -> Optional[np.ndarray]:
        """B_host: M x K on every rank (only the local columns are uploaded).  Returns the N x K
        solution on rank 0 (None elsewhere)."""
        rls = self.rls
        cols = shard_columns(B_host.shape[1], self.world, self.rank)
        solver = self.solver_factory()
        local = None
        if len(cols):
            Bd = rls.DeviceMatrix.from_host(np.asfortranarray(B_host[:, cols.start:cols.stop]), ctx)
            xs = rls.solve_(solver, Bd, scheduler=rls.MultiThreadingState)
            local = np.stack([x.to_host() for x in xs], axis=1)
        if self.dist is None:
            return local
        gathered = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object((cols.start, local), gathered, dst=0)
        if self.rank != 0:
            return None
        parts = [g for g in gathered if g[1] is not None]
        parts.sort(key=lambda t: t[0])
        return np.concatenate([p[1] for p in parts], axis=1)


# --------------------------------------------------------------------------------------------
# config 5: row-sharded CGNR
# --------------------------------------------------------------------------------------------


class HipLocalOps:
    """The rank-local half-steps of row-sharded CGNR on the GPU (include/rls_mi355x.h:
    rls_cgnr_init_local_a/b, rls_cgnr_step_local_a/b).  State vectors live in torch CUDA tensors so
    that torch.distributed can all-reduce them in place; the rls context borrows torch's current
    stream, which orders the kernels and the collective without extra events."""

    def __init__(self, rls, A_local: np.ndarray, device: int):
        import torch

        self.rls, self.torch = rls, torch
        torch.cuda.set_device(device)
        self.ctx = rls.Context(device, stream=torch.cuda.current_stream().cuda_stream)
        self.A = rls.DeviceMatrix.from_host(A_local, self.ctx)
        self.op = rls.OperatorHandle(self.A)
        n = self.A.N
        cplx = self.A.dtype.kind == "c"
        tdt = torch.complex64 if cplx else torch.float32
        self.t = {k: torch.zeros(n, dtype=tdt, device=f"cuda:{device}") for k in ("x", "r", "p", "v")}
        lib, h = self.ctx.lib, self.ctx.handle
        plan = C.c_void_p()
        check(h, lib.rls_cgnr_create(self.op.handle, self.t["x"].data_ptr(), self.t["r"].data_ptr(),
                                     self.t["p"].data_ptr(), self.t["v"].data_ptr(), C.byref(plan)), "rls_cgnr_create")
        self.plan = plan
        self._b = None

    def init_a(self, b_local: np.ndarray, lam: float, rel_tol: float, iterations: int):
        self._b = self.rls.DeviceVector.from_host(b_local, self.ctx)
        check(self.ctx.handle, self.ctx.lib.rls_cgnr_init_local_a(self.plan, self._b.ptr, lam, rel_tol, iterations),
              "rls_cgnr_init_local_a")

    def init_b(self):
        check(self.ctx.handle, self.ctx.lib.rls_cgnr_init_local_b(self.plan), "rls_cgnr_init_local_b")

    def step_a(self):
        check(self.ctx.handle, self.ctx.lib.rls_cgnr_step_local_a(self.plan), "rls_cgnr_step_local_a")

    def step_b(self):
        check(self.ctx.handle, self.ctx.lib.rls_cgnr_step_local_b(self.plan), "rls_cgnr_step_local_b")

    def tensor(self, name):
        return self.t[name]

    def status(self):
        st = CgnrStatus()
        check(self.ctx.handle, self.ctx.lib.rls_cgnr_get_status(self.plan, C.byref(st)), "rls_cgnr_get_status")
        return {"iteration": st.iteration, "done": bool(st.done), "residual": st.residual, "z0": st.z0}

    def solution(self) -> np.ndarray:
        self.ctx.sync()
        return self.t["x"].cpu().numpy()

    def sync(self):
        self.ctx.sync()

    def close(self):
        if self.plan:
            self.ctx.lib.rls_cgnr_destroy(self.plan)
            self.plan = None


class RowShardedCGNR:
    """CGNR on a row-partitioned A.  `ops` implements the local-ops protocol (HipLocalOps in the
    product).  Per iteration: step_a (t_g = A_g p, v_g = A_g^H t_g), all-reduce(v), step_b."""

    def __init__(self, ops, dist=None, lam: float = 0.0, iterations: int = 10, relTol: float = float(np.finfo(np.float32).eps)):
        self.ops, self.dist = ops, dist
        self.lam, self.iterations, self.relTol = float(lam), int(iterations), float(relTol)

    def _allreduce(self, name):
        if self.dist is not None and self.dist.get_world_size() > 1:
            self.dist.all_reduce(self.ops.tensor(name), op=self.dist.ReduceOp.SUM)

    def init(self, b_local):
        """r = sum_g A_g^H b_g (one all-reduce), then the replicated init (src/CGNR.jl:107-130)"""
        self.ops.init_a(b_local, self.lam, self.relTol, self.iterations)
        self._allreduce("r")
        self.ops.init_b()

    def step(self, n: int = 1):
        for _ in range(n):
            self.ops.step_a()
            self._allreduce("v")
            self.ops.step_b()

    def solve(self, b_local):
        self.init(b_local)
        # every rank holds identical scalars, so `done` flips on the same iteration everywhere; the
        # steps past it are no-ops on the device, and the collective count stays matched
        self.step(min(self.iterations, self.ops.tensor("x").shape[0]))
        return self.ops.solution()


def make_row_shard(M: int, N: int, rank: int, world: int, dtype=np.complex64, seed0: int = 500):
    """C5 data: rank r generates rows of its block with seed seed0 + r, contiguous lda."""
    lo, hi = shard_rows(M, world, rank, align=4)
    rng = np.random.default_rng(seed0 + rank)
    m = hi - lo
    if np.dtype(dtype).kind == "c":
        A = np.empty((m, N), dtype=np.complex64, order="F")
        s = np.float32(1 / math.sqrt(2))
        A.real = rng.standard_normal((N, m), dtype=np.float32).T * s
        A.imag = rng.standard_normal((N, m), dtype=np.float32).T * s
    else:
        A = np.asfortranarray(rng.standard_normal((N, m), dtype=np.float32).T)
    return A, lo, hi


def bench_rowsharded(rls, ctx, dist, rank, world, K, W, M=65536, N=8192):
    """BASELINE config 5 measurement: iterations/s of one 65536 x 8192 ComplexF32 CGNR, row-sharded
    over `world` GPUs (strong scaling of one problem; not the default bench line)."""
    import torch

    A, lo, hi = make_row_shard(M, N, rank, world)
    rng = np.random.default_rng(7)
    x_true = ((rng.standard_normal(N) + 1j * rng.standard_normal(N)) / math.sqrt(2)).astype(np.complex64)
    b_local = (A @ x_true).astype(np.complex64)
    ops = HipLocalOps(rls, A, torch.cuda.current_device())
    seg = 32
    solver = RowShardedCGNR(ops, dist, iterations=seg, relTol=0.0)

    def run(n):
        while n > 0:
            m = min(n, seg)
            solver.init(b_local)
            solver.step(m)
            n -= m

    run(W)
    ops.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    run(K)
    ops.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    st = ops.status()
    s = 8
    bytes_iter = 2 * M * N * s + (16 * N + 2 * M) * s
    return {"metric": "CGNR iterations/sec, row-sharded 65536x8192 CF32 (BASELINE config 5)", "value": K / elapsed,
            "unit": "iterations/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "c64", "data": "synthetic",
            "config": {"workload": f"CGNR {M}x{N} ComplexF32 row-partitioned over {world} GPU(s), one all-reduce of "
                                   f"A^H t ({N * s} B) per iteration", "rows_per_gpu": hi - lo},
            "roofline": {"bound": "hbm", "achieved": bytes_iter * K / elapsed / 1e9, "peak": 8000.0 * world,
                         "unit": "GB/s", "frac": bytes_iter * K / elapsed / 1e9 / (8000.0 * world), "traffic": None},
            "residual": st["residual"]}
