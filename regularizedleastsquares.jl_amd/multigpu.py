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
  * RowShardedFISTA / RowShardedADMM -- the same partitioning for the other two solvers of the hot path
    (last row): the operator apply is the only distributed step (one all-reduce of A_g^H A_g y
    per FISTA iteration, of A_g^H A_g u per inner cg! iteration of ADMM); prox, momentum, z/u updates and
    all scalars run replicated.

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

from ._lib import (PROJ_NONE, PROJ_POSITIVE, PROJ_REAL, REG_L1, REG_L2, REG_L21, REG_NONE, REG_TV, AdmmStatus, CgnrStatus, CgStatus,
                   FistaStatus, check)
from .arrays import Context


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

    def __init__(self, rls, solver_factory, dist=None, scheduler=None):
        self.rls = rls
        self.solver_factory = solver_factory
        self.dist = dist
        # the local columns advance together through one pass over A per product where the solver has a batched plan
        # (CGNR, FISTA; matrix cores), and column by column otherwise -- same results either way
        self.scheduler = scheduler if scheduler is not None else rls.BatchedState
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
            xs = rls.solve_(solver, Bd, scheduler=self.scheduler)
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


class ConcurrentSolves:
    """Config 4, distinct-A flavour inside one GPU: several independent solvers, each with its OWN matrix, solved side
    by side -- the reference spawns one Julia task per solver (docs/src/literate/howto/multi_threading.jl:8-17); here
    every worker thread owns a context (= a HIP stream of its own), so uploads, setup kernels and the small kernels
    of one solve overlap with the large kernels of another.  The library is re-entrant (all mutable state lives in
    the rls_ctx) and ctypes releases the GIL inside every call.  Kernels that need the whole chip (the one-pass slab
    kernels, the resident kernels) still run one after another: they are chained on the device."""

    def __init__(self, rls, n_streams: int = 8, device: int = 0):
        self.rls = rls
        self.ctxs = [rls.Context(device) for _ in range(int(n_streams))]

    def upload(self, matrices):
        """A_k -> DeviceMatrix on context k % n_streams (residency before a timed region)"""
        rls = self.rls
        return [rls.DeviceMatrix.from_host(A, self.ctxs[k % len(self.ctxs)]) for k, A in enumerate(matrices)]

    def solve(self, device_matrices, rhs, make_solver):
        """rhs[k]: host vector for problem k; make_solver(Ad) -> solver.  Returns the host solutions in problem order."""
        import threading

        rls, n = self.rls, len(device_matrices)
        out, errs = [None] * n, []

        def worker(slot):
            # a context (its stream, its graph captures) is driven by ONE thread at a time: worker `slot` takes exactly
            # the problems that live on context `slot`
            for k in range(n):
                Ad = device_matrices[k]
                if Ad.ctx is not self.ctxs[slot]:
                    continue
                try:
                    solver = make_solver(Ad)
                    x = rls.solve_(solver, rls.DeviceVector.from_host(rhs[k], Ad.ctx))
                    out[k] = x.to_host()
                except Exception as e:  # surfaced after the join
                    errs.append((k, e))

        threads = [threading.Thread(target=worker, args=(s_,)) for s_ in range(len(self.ctxs))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errs:
            raise errs[0][1]
        return out

    def solve_queue(self, device_matrices, rhs, make_solver):
        """The same job as ONE queue per context (rls_cgnr_solve_queue_host through solve_group_): the solvers -- and with them the
        plans, their state vectors and the pinned staging of b and x -- are created once per operator and cached, problem k's upload,
        init! and iterations are enqueued behind problem k - 1's, and each context synchronises ONCE.  rhs[k]: host vectors; returns
        host solutions in problem order.  Contexts beyond the first only matter when the matrices were uploaded to several."""
        rls, n = self.rls, len(device_matrices)
        cache = self.__dict__.setdefault("_solvers", {})
        out = [None] * n
        for c in self.ctxs:
            ks = [k for k in range(n) if device_matrices[k].ctx is c]
            if not ks:
                continue
            solvers = []
            for k in ks:
                key = id(device_matrices[k])
                if key not in cache:
                    cache[key] = (make_solver(device_matrices[k]), device_matrices[k])  # (keeps the matrix alive with its solver)
                solvers.append(cache[key][0])
            xs = rls.solve_group_(solvers, [np.asarray(rhs[k]) for k in ks])
            for k, x in zip(ks, xs):
                out[k] = x
        return out

    def close(self):
        self.__dict__.pop("_solvers", None)
        for c in self.ctxs:
            c.close()
        self.ctxs = []


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


class _BorrowedContext(Context):
    """a Context object around an rls_ctx the communicator owns (never destroyed from here)"""

    def __init__(self, lib, handle, device):  # noqa: super().__init__ would create a new rls_ctx
        self.lib, self.handle, self.device = lib, handle, device

    def close(self):
        self.handle = None


COMM_AUTO, COMM_RCCL, COMM_DIRECT = 0, 1, 2


def comm_peer_report(lib, comm, n):
    """rls_comm_peer_access as a dict: the hipDeviceCanAccessPeer matrix rls_comm_create probed (1 can store into / same
    device, 0 cannot, -1 query failed), the transport that was asked for and the one in use (a requested direct transport is
    dropped to RCCL when some pair of distinct devices has no peer access)"""
    mat = (C.c_int32 * (n * n))()
    req = C.c_int32(-1)
    st = lib.rls_comm_peer_access(comm, mat, C.byref(req))
    if st != 0:
        return {"error": f"rls_comm_peer_access status {st}"}
    rows = [[int(mat[r * n + t]) for t in range(n)] for r in range(n)]
    used = int(lib.rls_comm_transport(comm))
    return {"peer_access": rows, "all_pairs": all(v == 1 for row in rows for v in row), "transport_requested": int(req.value),
            "transport_in_use": used, "dropped_to_rccl": bool(req.value == COMM_DIRECT and used == COMM_RCCL)}



class CommRowShardedCGNR:
    """Config 5 driven from ONE host process through the library's own communicator (include/rls_mi355x.h:
    rls_comm_create, rls_cgnr_init_rowsharded, rls_cgnr_step_rowsharded): rank r owns a context, its row shard of A
    (repacked contiguous) and a CGNR plan; the all-reduce of A^H t runs inside the library (RCCL between distinct
    devices, the one-shot direct-write transport when ranks share a device -- the emulation of an 8-GPU node on one
    GPU).  This is the call sequence the Julia host issues (one process, one task per GPU)."""

    def __init__(self, rls, shards, devices=None, transport=COMM_AUTO, lam=0.0, iterations=10,
                 relTol=float(np.finfo(np.float32).eps)):
        self.rls = rls
        lib = rls.load()
        n = len(shards)
        devices = list(devices) if devices is not None else [0] * n
        devs = (C.c_int32 * n)(*devices)
        comm = C.c_void_p()
        st = lib.rls_comm_create(n, devs, None, int(transport), C.byref(comm))
        if st != 0:
            raise rls.RLSError(f"rls_comm_create failed with status {st}")
        self.lib, self.comm, self.n = lib, comm, n
        self.lam, self.iterations, self.relTol = float(lam), int(iterations), float(relTol)
        self.ctxs, self.A, self.ops, self.vecs, self.plans = [], [], [], [], []
        for r in range(n):
            h = C.c_void_p()
            check(None, lib.rls_comm_ctx(comm, r, C.byref(h)), "rls_comm_ctx")
            ctx = _BorrowedContext(lib, h, devices[r])
            Ar = rls.DeviceMatrix.from_host(np.asfortranarray(shards[r]), ctx)
            op = rls.OperatorHandle(Ar)
            v = [rls.DeviceVector(Ar.N, Ar.dtype, ctx) for _ in range(4)]  # x, r, p, v
            plan = C.c_void_p()
            check(h, lib.rls_cgnr_create(op.handle, v[0].ptr, v[1].ptr, v[2].ptr, v[3].ptr, C.byref(plan)), "rls_cgnr_create")
            self.ctxs.append(ctx); self.A.append(Ar); self.ops.append(op); self.vecs.append(v); self.plans.append(plan)
        self._plans_c = (C.c_void_p * n)(*[p.value for p in self.plans])
        self._b = None

    @property
    def transport(self):
        return self.lib.rls_comm_transport(self.comm)

    def peer_report(self):
        return comm_peer_report(self.lib, self.comm, self.n)

    def init(self, b_parts):
        self._b = [self.rls.DeviceVector.from_host(np.ascontiguousarray(bp), c) for bp, c in zip(b_parts, self.ctxs)]
        ptrs = (C.c_void_p * self.n)(*[b.ptr for b in self._b])
        check(self.ctxs[0].handle, self.lib.rls_cgnr_init_rowsharded(self.comm, self._plans_c, ptrs, self.lam, self.relTol, self.iterations),
              "rls_cgnr_init_rowsharded")

    def step(self, n=1):
        check(self.ctxs[0].handle, self.lib.rls_cgnr_step_rowsharded(self.comm, self._plans_c, int(n)), "rls_cgnr_step_rowsharded")

    def sync(self):
        check(self.ctxs[0].handle, self.lib.rls_comm_sync(self.comm), "rls_comm_sync")

    def solution(self, rank=0) -> np.ndarray:
        self.sync()
        return self.vecs[rank][0].to_host()

    def status(self, rank=0):
        st = CgnrStatus()
        check(self.ctxs[rank].handle, self.lib.rls_cgnr_get_status(self.plans[rank], C.byref(st)), "rls_cgnr_get_status")
        return {"iteration": st.iteration, "done": bool(st.done), "residual": st.residual, "z0": st.z0}

    def solve(self, b_parts):
        self.init(b_parts)
        self.step(min(self.iterations, self.A[0].N))
        return self.solution()

    def close(self):
        if self.comm:
            self.sync()
            for p in self.plans:
                self.lib.rls_cgnr_destroy(p)
            self.plans, self._b, self.vecs, self.ops, self.A = [], None, [], [], []
            self.lib.rls_comm_destroy(self.comm)
            self.comm = None
            for c in self.ctxs:
                c.close()


class _CommHost:
    """shared plumbing of the single-process row-sharded hosts: the communicator, one borrowed context, one shard operator
    and a set of named device vectors per rank"""

    def __init__(self, rls, shards, names, devices=None, transport=COMM_AUTO, threads=True):
        self.rls = rls
        lib = rls.load()
        n = len(shards)
        devices = list(devices) if devices is not None else [0] * n
        devs = (C.c_int32 * n)(*devices)
        comm = C.c_void_p()
        st = lib.rls_comm_create(n, devs, None, int(transport), C.byref(comm))
        if st != 0:
            raise rls.RLSError(f"rls_comm_create failed with status {st}")
        self.lib, self.comm, self.n = lib, comm, n
        check(None, lib.rls_comm_set_threads(comm, 1 if threads else 0), "rls_comm_set_threads")
        self.ctxs, self.A, self.ops, self.v = [], [], [], []
        for r in range(n):
            h = C.c_void_p()
            check(None, lib.rls_comm_ctx(comm, r, C.byref(h)), "rls_comm_ctx")
            ctx = _BorrowedContext(lib, h, devices[r])
            Ar = rls.DeviceMatrix.from_host(np.asfortranarray(shards[r]), ctx)
            self.ctxs.append(ctx)
            self.A.append(Ar)
            self.ops.append(rls.OperatorHandle(Ar))
            self.v.append({k: rls.DeviceVector(Ar.N, Ar.dtype, ctx) for k in names})
        self.plans = []
        self._b = None

    @property
    def transport(self):
        return self.lib.rls_comm_transport(self.comm)

    def _upload_b(self, b_parts):
        self._b = [self.rls.DeviceVector.from_host(np.ascontiguousarray(bp), c) for bp, c in zip(b_parts, self.ctxs)]
        return (C.c_void_p * self.n)(*[b.ptr for b in self._b])

    def sync(self):
        check(self.ctxs[0].handle, self.lib.rls_comm_sync(self.comm), "rls_comm_sync")

    def _destroy_plans(self):
        raise NotImplementedError

    def close(self):
        if self.comm:
            self.sync()
            self._destroy_plans()
            self.plans, self._b, self.v, self.ops, self.A = [], None, [], [], []
            self.lib.rls_comm_destroy(self.comm)
            self.comm = None
            for c in self.ctxs:
                c.close()


class CommRowShardedFISTA(_CommHost):
    """FISTA on a row-partitioned A from ONE host process through the library (rls_fista_init_rowsharded /
    rls_fista_step_rowsharded): what the Julia host issues for SURVEY 8e's last row (src/FISTA.jl:114,152)."""

    def __init__(self, rls, shards, reg=None, proj=None, devices=None, transport=COMM_AUTO, rho=1.0, theta=1.0, iterations=50,
                 relTol=float(np.finfo(np.float32).eps), restart="none", threads=True):
        super().__init__(rls, shards, ("x", "x0", "xold", "res"), devices, transport, threads)
        self.rho, self.theta, self.iterations, self.relTol, self.restart = float(rho), float(theta), int(iterations), float(relTol), restart
        try:
            _reg_codes(rls, reg, proj)  # (validates; proj: one projection term or None)
            for r in range(self.n):
                v, h = self.v[r], self.ctxs[r].handle
                plan = C.c_void_p()
                check(h, self.lib.rls_fista_create(self.ops[r].handle, v["x"].ptr, v["x0"].ptr, v["xold"].ptr, v["res"].ptr, C.byref(plan)),
                      "rls_fista_create")
                self.plans.append(plan)
                _fista_plan_set_reg(rls, self.lib, h, plan, reg, proj)
        except BaseException:
            self.close()   # a refused regulariser (e.g. a TV image beyond the single-workgroup FGP launch) must not leak the
            raise          # communicator, its worker threads, the contexts and the plans created so far
        self._plans_c = (C.c_void_p * self.n)(*[p.value for p in self.plans])

    def init(self, b_parts):
        ptrs = self._upload_b(b_parts)
        check(self.ctxs[0].handle, self.lib.rls_fista_init_rowsharded(self.comm, self._plans_c, ptrs, self.rho, self.theta, self.relTol,
                                                                      self.iterations, int(self.restart == "gradient")),
              "rls_fista_init_rowsharded")

    def step(self, n=1):
        check(self.ctxs[0].handle, self.lib.rls_fista_step_rowsharded(self.comm, self._plans_c, int(n)), "rls_fista_step_rowsharded")

    def status(self, rank=0):
        st = FistaStatus()
        check(self.ctxs[rank].handle, self.lib.rls_fista_get_status(self.plans[rank], C.byref(st)), "rls_fista_get_status")
        return {"iteration": st.iteration, "done": bool(st.done), "rel_res_norm": st.rel_res_norm, "residual": st.residual}

    def solution(self, rank=0) -> np.ndarray:
        self.sync()
        p = C.c_void_p()
        check(self.ctxs[rank].handle, self.lib.rls_fista_solution(self.plans[rank], C.byref(p)), "rls_fista_solution")
        for k in ("x", "xold"):  # the plan swaps x / xold by pointer (src/FISTA.jl:144-146)
            if self.v[rank][k].ptr == p.value:
                return self.v[rank][k].to_host()
        raise RuntimeError("rls_fista_solution returned a foreign pointer")

    def solve(self, b_parts):
        self.init(b_parts)
        self.step(self.iterations)
        return self.solution()

    def _destroy_plans(self):
        for p in self.plans:
            self.lib.rls_fista_destroy(p)


class CommRowShardedADMM(_CommHost):
    """ADMM (one regulariser: L1 / L2 / TV, identity regTrafo, vary_rho = :none) on a row-partitioned A from ONE host
    process: the single-GPU device plan's kernels replicated per rank, the operator applies of the inner cg! per shard
    with one all-reduce each (rls_admm_init_rowsharded / rls_admm_step_rowsharded; src/ADMM.jl:198,244)."""

    NAMES = ("x", "xold", "beta", "beta_y", "z0", "z1", "u", "cg_u", "cg_r", "cg_c")

    def __init__(self, rls, shards, reg, M_total, proj=None, devices=None, transport=COMM_AUTO, rho=0.1, iterations=10,
                 iterationsCG=10, absTol=float(np.finfo(np.float32).eps), relTol=float(np.finfo(np.float32).eps), tolInner=1e-5,
                 threads=True):
        super().__init__(rls, shards, self.NAMES, devices, transport, threads)
        from .solvers import ADMM
        self.iterations = int(iterations)
        # the parameter block of the device plan, filled exactly as the single-GPU host fills it
        host = ADMM(self.A[0], reg=([reg] + list(proj or ())), rho=rho, iterations=iterations, iterationsCG=iterationsCG,
                    absTol=absTol, relTol=relTol, tolInner=tolInner)
        self._host = host
        self.cgs = []
        f32 = np.float32
        for r in range(self.n):
            v, h = self.v[r], self.ctxs[r].handle
            cg = C.c_void_p()
            check(h, self.lib.rls_cg_create(self.ops[r].handle, v["cg_u"].ptr, v["cg_r"].ptr, v["cg_c"].ptr, C.byref(cg)), "rls_cg_create")
            plan = C.c_void_p()
            check(h, self.lib.rls_admm_create(cg, C.byref(plan)), "rls_admm_create")
            self.cgs.append(cg)
            self.plans.append(plan)
        self._plans_c = (C.c_void_p * self.n)(*[p.value for p in self.plans])
        self._sigma_abs = f32(np.sqrt(f32(M_total))) * f32(absTol)   # sqrt(length(b)) of the WHOLE b   (src/ADMM.jl:214)
        self._rho, self._relTol, self._tolInner, self._icg = f32(rho), f32(relTol), f32(tolInner), int(iterationsCG)

    def _params(self, r):
        from types import SimpleNamespace
        v = self.v[r]
        st = SimpleNamespace(rho=np.array([self._rho], np.float32), x=v["x"], xold=v["xold"], beta=v["beta"], beta_y=v["beta_y"],
                             _zbufs=(v["z0"], v["z1"]), u=[v["u"]], sigma_abs=self._sigma_abs, relTol=self._relTol,
                             tolInner=self._tolInner)
        P = self._host._plan_params(st)
        if P is None:
            raise self.rls.RLSError("CommRowShardedADMM: this regulariser is not covered by the device plan")
        return P

    def init(self, b_parts):
        ptrs = self._upload_b(b_parts)
        for r in range(self.n):
            for k in ("x", "z0", "z1", "u"):   # x0 = 0, z = Phi x, u = 0   (src/ADMM.jl:199-206)
                self.v[r][k].fill_(0)
            P = self._params(r)
            check(self.ctxs[r].handle, self.lib.rls_admm_init(self.plans[r], C.byref(P)), "rls_admm_init")
        check(self.ctxs[0].handle, self.lib.rls_admm_init_rowsharded(self.comm, self._plans_c, ptrs), "rls_admm_init_rowsharded")

    def step(self, n=1):
        check(self.ctxs[0].handle, self.lib.rls_admm_step_rowsharded(self.comm, self._plans_c, int(n)), "rls_admm_step_rowsharded")

    def status(self, rank=0):
        st = AdmmStatus()
        cap = max(self.iterations, 1)
        log = (C.c_float * (8 * cap))()
        check(self.ctxs[rank].handle, self.lib.rls_admm_get_status(self.plans[rank], C.byref(st), log, cap), "rls_admm_get_status")
        return {"iteration": st.iteration, "done": bool(st.done), "rk": st.rk, "sk": st.sk,
                "cg_iterations": [int(log[8 * i + 5]) for i in range(st.iteration)]}

    def solution(self, rank=0) -> np.ndarray:
        self.sync()
        return self.v[rank]["x"].to_host()

    def solve(self, b_parts):
        self.init(b_parts)
        self.step(self.iterations)
        return self.solution()

    def _destroy_plans(self):
        for p in self.plans:
            self.lib.rls_admm_destroy(p)
        for c in self.cgs:
            self.lib.rls_cg_destroy(c)


# --------------------------------------------------------------------------------------------
# row-sharded FISTA and ADMM (same partitioning, same single exchange step per operator apply)
# --------------------------------------------------------------------------------------------


def _torch_vectors(torch, names, n, dtype, device):
    tdt = torch.complex64 if np.dtype(dtype).kind == "c" else torch.float32
    return {k: torch.zeros(n, dtype=tdt, device=f"cuda:{device}") for k in names}


def _reg_codes(rls, reg, proj):
    """(reg_kind, lambda, slices, proj_kind) of the fused elementwise update kernels"""
    if reg is None:
        kind, lam, slices = REG_NONE, 0.0, 1
    elif type(reg) is rls.L1Regularization:
        kind, lam, slices = REG_L1, reg.lam, 1
    elif type(reg) is rls.L2Regularization and getattr(reg, "lam_vector", None) is None:
        kind, lam, slices = REG_L2, reg.lam, 1
    elif type(reg) is rls.L21Regularization:
        kind, lam, slices = REG_L21, reg.lam, reg.slices
    elif type(reg) is rls.TVRegularization:
        kind, lam, slices = REG_TV, reg.lam, 1   # rls_fista_set_reg_tv (the FGP launch of the replicated update half)
    else:
        raise NotImplementedError("row-sharded FISTA: L1 / L2 / L21 / TV regularisation")
    pk = PROJ_NONE
    if proj is not None:
        if isinstance(proj, rls.PositiveRegularization):
            pk = PROJ_POSITIVE
        elif isinstance(proj, rls.RealRegularization):
            pk = PROJ_REAL
        else:
            raise NotImplementedError("row-sharded FISTA: Positive / Real projections")
    return kind, float(lam), int(slices), pk


def _fista_plan_set_reg(rls, lib, h, plan, reg, proj):
    """rls_fista_set_reg / rls_fista_set_reg_tv for one rank's plan (src/FISTA.jl:164-168: prox, then the projection)"""
    kind, lam, slices, pk = _reg_codes(rls, reg, proj)
    if kind == REG_TV:
        from .regularization import _tv_geometry
        shape, d0, cs, cd = _tv_geometry(reg.shape, reg.dims)
        st = lib.rls_fista_set_reg_tv(plan, lam, len(shape), cs, len(d0), cd, reg.iterationsTV, pk)
        if st == -2:
            raise NotImplementedError("row-sharded FISTA + TV: the image does not fit the single-workgroup FGP kernel "
                                      "(1-D / 2-D up to 8192 Float32 / 4096 ComplexF32 pixels, other geometries up to 2048)")
        check(h, st, "rls_fista_set_reg_tv")
    else:
        check(h, lib.rls_fista_set_reg(plan, kind, lam, slices, pk), "rls_fista_set_reg")


class HipFistaOps:
    """Rank-local half-steps of row-sharded FISTA on the GPU (rls_fista_init_local_a/b, rls_fista_step_local_a/b)."""

    def __init__(self, rls, A_local: np.ndarray, device: int, reg=None, proj=None):
        import torch

        self.rls, self.torch = rls, torch
        torch.cuda.set_device(device)
        self.ctx = rls.Context(device, stream=torch.cuda.current_stream().cuda_stream)
        self.A = rls.DeviceMatrix.from_host(A_local, self.ctx)
        self.op = rls.OperatorHandle(self.A)
        self.t = _torch_vectors(torch, ("x", "x0", "xold", "res"), self.A.N, self.A.dtype, device)
        lib, h = self.ctx.lib, self.ctx.handle
        plan = C.c_void_p()
        check(h, lib.rls_fista_create(self.op.handle, self.t["x"].data_ptr(), self.t["x0"].data_ptr(),
                                      self.t["xold"].data_ptr(), self.t["res"].data_ptr(), C.byref(plan)), "rls_fista_create")
        self.plan = plan
        _fista_plan_set_reg(rls, lib, h, plan, reg, proj)
        self._b = None

    def init_a(self, b_local: np.ndarray):
        self._b = self.rls.DeviceVector.from_host(b_local, self.ctx)
        check(self.ctx.handle, self.ctx.lib.rls_fista_init_local_a(self.plan, self._b.ptr), "rls_fista_init_local_a")

    def init_b(self, rho, theta, rel_tol, iterations, restart_gradient):
        check(self.ctx.handle, self.ctx.lib.rls_fista_init_local_b(self.plan, float(rho), float(theta), float(rel_tol),
                                                                   int(iterations), int(bool(restart_gradient))),
              "rls_fista_init_local_b")

    def step_a(self):
        check(self.ctx.handle, self.ctx.lib.rls_fista_step_local_a(self.plan), "rls_fista_step_local_a")

    def step_b(self):
        check(self.ctx.handle, self.ctx.lib.rls_fista_step_local_b(self.plan), "rls_fista_step_local_b")

    def tensor(self, name):
        return self.t[name]

    def status(self):
        st = FistaStatus()
        check(self.ctx.handle, self.ctx.lib.rls_fista_get_status(self.plan, C.byref(st)), "rls_fista_get_status")
        return {"iteration": st.iteration, "done": bool(st.done), "rel_res_norm": st.rel_res_norm, "residual": st.residual}

    def solution(self) -> np.ndarray:
        p = C.c_void_p()
        check(self.ctx.handle, self.ctx.lib.rls_fista_solution(self.plan, C.byref(p)), "rls_fista_solution")
        for t in (self.t["x"], self.t["xold"]):  # the plan swaps x / xold by pointer (src/FISTA.jl:144-146)
            if t.data_ptr() == p.value:
                return t.cpu().numpy()
        raise RuntimeError("rls_fista_solution returned a foreign pointer")

    def sync(self):
        self.ctx.sync()

    def close(self):
        if self.plan:
            self.ctx.lib.rls_fista_destroy(self.plan)
            self.plan = None


class RowShardedFISTA:
    """FISTA (src/FISTA.jl:110-185) on a row-partitioned A.  Per iteration: step_a (res_g = A_g^H A_g y),
    all-reduce(res), step_b (gradient step, prox, momentum -- replicated).  `done` flips on the same iteration on
    every rank (identical scalars), the steps past it are no-ops on the device, the collective count stays matched."""

    def __init__(self, ops, dist=None, rho: float = 1.0, theta: float = 1.0, iterations: int = 50,
                 relTol: float = float(np.finfo(np.float32).eps), restart: str = "none"):
        if restart not in ("none", "gradient"):
            raise ValueError("restart must be 'none' or 'gradient'")
        self.ops, self.dist = ops, dist
        self.rho, self.theta, self.iterations, self.relTol = float(rho), float(theta), int(iterations), float(relTol)
        self.restart = restart

    def _allreduce(self, name):
        if self.dist is not None and self.dist.get_world_size() > 1:
            self.dist.all_reduce(self.ops.tensor(name), op=self.dist.ReduceOp.SUM)

    def init(self, b_local):
        self.ops.init_a(b_local)
        self._allreduce("x0")  # x0 = sum_g A_g^H b_g   (src/FISTA.jl:114)
        self.ops.init_b(self.rho, self.theta, self.relTol, self.iterations, self.restart == "gradient")

    def step(self, n: int = 1):
        for _ in range(n):
            self.ops.step_a()
            self._allreduce("res")
            self.ops.step_b()

    def solve(self, b_local):
        self.init(b_local)
        self.step(self.iterations)
        return self.ops.solution()


class HipAdmmOps:
    """Rank-local steps of row-sharded ADMM on the GPU (one regulariser, identity regTrafo): rls_admm_pre / _post for
    the replicated elementwise work, rls_cg_local_apply / _start / _update for the inner cg! whose operator apply is
    the distributed step."""

    def __init__(self, rls, A_local: np.ndarray, device: int, reg=None, proj=()):
        import torch

        self.rls, self.torch = rls, torch
        torch.cuda.set_device(device)
        self.ctx = rls.Context(device, stream=torch.cuda.current_stream().cuda_stream)
        self.A = rls.DeviceMatrix.from_host(A_local, self.ctx)
        self.op = rls.OperatorHandle(self.A)
        n, dt = self.A.N, self.A.dtype
        names = ("x", "xold", "beta", "beta_y", "z", "zold", "u", "cg_u", "cg_r", "cg_c")
        self.t = _torch_vectors(torch, names, n, dt, device)
        # DeviceVector views of the torch storage, for the prox maps and the fused elementwise entry points
        self.v = {k: rls.DeviceVector.borrow(self.t[k].data_ptr(), n, dt, self.ctx, keep=self.t[k]) for k in names}
        self.reg, self.proj = reg, list(proj)
        lib, h = self.ctx.lib, self.ctx.handle
        plan = C.c_void_p()
        check(h, lib.rls_cg_create(self.op.handle, self.t["cg_u"].data_ptr(), self.t["cg_r"].data_ptr(),
                                   self.t["cg_c"].data_ptr(), C.byref(plan)), "rls_cg_create")
        self.cg = plan

    def init_a(self, b_local: np.ndarray):
        b = self.rls.DeviceVector.from_host(b_local, self.ctx)
        self.A.mul_adj_(self.v["beta_y"], b)  # partial A_g^H b_g   (src/ADMM.jl:198)

    def init_b(self):
        for k in ("x", "z", "u"):  # x0 = 0, z = Phi x, u = 0   (:199-206)
            self.v[k].fill_(0)

    def pre(self, rho):
        v, lib, h = self.v, self.ctx.lib, self.ctx.handle
        check(h, lib.rls_admm_pre(h, v["x"].code, v["x"].n, v["beta"].ptr, v["beta_y"].ptr, v["z"].ptr, v["u"].ptr,
                                  v["x"].ptr, v["xold"].ptr, float(rho), 0), "rls_admm_pre")

    def apply_x(self):
        check(self.ctx.handle, self.ctx.lib.rls_cg_local_apply(self.cg, self.v["x"].ptr), "rls_cg_local_apply")

    def cg_start(self, rho, maxiter, reltol):
        check(self.ctx.handle, self.ctx.lib.rls_cg_local_start(self.cg, self.v["x"].ptr, self.v["beta"].ptr, float(rho),
                                                               int(maxiter), float(reltol)), "rls_cg_local_start")

    def apply_u(self):
        check(self.ctx.handle, self.ctx.lib.rls_cg_local_apply(self.cg, None), "rls_cg_local_apply")

    def cg_update(self):
        check(self.ctx.handle, self.ctx.lib.rls_cg_local_update(self.cg, self.v["x"].ptr), "rls_cg_local_update")

    def cg_iterations(self) -> int:
        st = CgStatus()
        check(self.ctx.handle, self.ctx.lib.rls_cg_get_status(self.cg, C.byref(st)), "rls_cg_get_status")
        return int(st.iterations)

    def post(self, prox_lambda):
        """projections, z = prox(x + u), u += x - z and the norms (src/ADMM.jl:246-299); returns the 6-float record"""
        v, lib, h = self.v, self.ctx.lib, self.ctx.handle
        for pr in self.proj:
            pr.prox_(v["x"])
        self.t["z"], self.t["zold"] = self.t["zold"], self.t["z"]
        v["z"], v["zold"] = v["zold"], v["z"]
        v["z"].lincomb_(1.0, v["x"], 1.0, v["u"])
        if prox_lambda is not None and self.reg is not None:
            self.reg.prox_(v["z"], float(prox_lambda))
        out = (C.c_float * 6)()
        check(h, lib.rls_admm_post(h, v["x"].code, v["x"].n, v["x"].ptr, v["xold"].ptr, v["z"].ptr, v["zold"].ptr,
                                   v["u"].ptr, out), "rls_admm_post")
        return [float(o) for o in out]

    def tensor(self, name):
        return self.t[name]

    def solution(self) -> np.ndarray:
        self.ctx.sync()
        return self.t["x"].cpu().numpy()

    def sync(self):
        self.ctx.sync()

    def close(self):
        if self.cg:
            self.ctx.lib.rls_cg_destroy(self.cg)
            self.cg = None


class RowShardedADMM:
    """ADMM (src/ADMM.jl:191-330; one regulariser, identity regTrafo, vary_rho = :none) on a row-partitioned A.
    The x-update's cg! runs a fixed `iterationsCG` half-step pairs (apply_u, all-reduce(c), cg_update); its own
    convergence is a device flag, so the collective count is the same on every rank whatever the data."""

    def __init__(self, ops, dist=None, lam: float = 0.0, rho: float = 0.1, iterations: int = 10, iterationsCG: int = 10,
                 absTol: float = float(np.finfo(np.float32).eps), relTol: float = float(np.finfo(np.float32).eps),
                 tolInner: float = 1e-5):
        self.ops, self.dist = ops, dist
        self.lam, self.rho = np.float32(lam), np.float32(rho)
        self.iterations, self.iterationsCG = int(iterations), int(iterationsCG)
        self.absTol, self.relTol, self.tolInner = np.float32(absTol), np.float32(relTol), np.float32(tolInner)
        self.iteration = 0
        self.cg_iterations: List[int] = []

    def _allreduce(self, name):
        if self.dist is not None and self.dist.get_world_size() > 1:
            self.dist.all_reduce(self.ops.tensor(name), op=self.dist.ReduceOp.SUM)

    def init(self, b_local, M_total: int):
        self.ops.init_a(b_local)
        self._allreduce("beta_y")
        self.ops.init_b()
        f32 = np.float32
        self.rk = self.sk = f32(np.inf)
        self.eps_pri = self.eps_dua = f32(0)
        self.sigma_abs = f32(np.sqrt(f32(M_total))) * self.absTol  # sqrt(length(b)) of the WHOLE b  (:214)
        self.iteration = 0
        self.cg_iterations = []

    def converged(self):
        return bool(self.rk < self.sigma_abs + self.relTol * self.eps_pri and
                    self.sk < self.sigma_abs + self.relTol * self.eps_dua)

    def done(self):
        return self.converged() or self.iteration >= self.iterations

    def iterate(self):
        if self.done():
            return None
        ops, f32 = self.ops, np.float32
        ops.pre(self.rho)                                    # :236-243
        ops.apply_x()
        self._allreduce("cg_c")
        ops.cg_start(self.rho, self.iterationsCG, self.tolInner)
        for _ in range(self.iterationsCG):                   # cg!  :244
            ops.apply_u()
            self._allreduce("cg_c")
            ops.cg_update()
        with np.errstate(divide="ignore"):
            rec = ops.post(None if self.rho == 0 else f32(self.lam) / (f32(2) * self.rho))  # :246-299
        self.cg_iterations.append(ops.cg_iterations())
        self.sk = self.rho * f32(rec[1])
        self.eps_pri = f32(rec[2])
        self.rk = f32(rec[3])
        self.eps_dua = self.rho * f32(rec[4])
        self.iteration += 1
        return self.iteration

    def solve(self, b_local, M_total: int):
        self.init(b_local, M_total)
        while self.iterate() is not None:
            pass
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


def bench_rowsharded(rls, ctx, dist, rank, world, K, W, M=65536, N=8192, agree=None, inject=None):
    """BASELINE config 5 measurement: iterations/s of one 65536 x 8192 ComplexF32 CGNR, row-sharded
    over `world` GPUs (strong scaling of one problem; not the default bench line).
    agree(stage, err): the caller's agreement point (bench.py): this function's setup -- shard generation, upload, plan creation, no
    collective -- ends there, so that a rank whose setup failed does not leave the others alone in the first all-reduce."""
    import torch

    err = None
    try:
        if inject is not None:
            inject("config5")
        A, lo, hi = make_row_shard(M, N, rank, world)
        rng = np.random.default_rng(7)
        x_true = ((rng.standard_normal(N) + 1j * rng.standard_normal(N)) / math.sqrt(2)).astype(np.complex64)
        b_local = (A @ x_true).astype(np.complex64)
        ops = HipLocalOps(rls, A, torch.cuda.current_device())
        seg = 32
        solver = RowShardedCGNR(ops, dist, iterations=seg, relTol=0.0)
    except Exception as e:  # noqa: BLE001
        if agree is None:
            raise
        err = e
    if agree is not None:
        agree("config5: setup", err)

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
    # ---- the dominant kernels of one rank, timed live with hipEvents on the stream they run on (rank 0 reports) ----
    m_loc = hi - lo
    kern = {}
    reps = 10
    for name, fn in (("step_local_a (gemv_n_kernel t = A_g p, then gemv_t_kernel v = A_g^H t)", ops.step_a),):
        fn(); ops.sync()
        ops.ctx.timer_start()
        for _ in range(reps):
            fn()
        us = 1e3 * ops.ctx.timer_stop_ms() / reps
        by = 2 * m_loc * N * s + 2 * (m_loc + N) * s
        kern[name] = {"us_per_call": us, "algorithmic_bytes_per_call": by, "GBps": by / us / 1e3, "frac_hbm": by / us / 1e3 / 8000.0}
    us_ar = None
    if dist is not None and world > 1:
        t = ops.tensor("v")
        for _ in range(3):
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ops.sync(); torch.cuda.synchronize()
        ops.ctx.timer_start()
        for _ in range(reps):
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        us_ar = 1e3 * ops.ctx.timer_stop_ms() / reps
    dom = next(iter(kern))
    return {"metric": "CGNR iterations/sec, row-sharded 65536x8192 CF32 (BASELINE config 5)", "value": K / elapsed,
            "unit": "iterations/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "c64", "data": "synthetic",
            "config": {"workload": f"CGNR {M}x{N} ComplexF32 row-partitioned over {world} GPU(s), one all-reduce of "
                                   f"A^H t ({N * s} B) per iteration", "rows_per_gpu": m_loc,
                       "collective": {"backend": (dist.get_backend() if dist is not None else None),
                                      "world_size_seen_by_the_collective": (dist.get_world_size() if dist is not None else 1),
                                      "all_reduce_us (64 KiB, back to back)": us_ar}},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": min(bytes_iter * K / elapsed / 1e9, 8000.0 * world),
                         "peak": 8000.0 * world, "unit": "GB/s", "frac": min(bytes_iter * K / elapsed / 1e9 / (8000.0 * world), 1.0),
                         "frac_hbm_dominant_kernel": kern[dom]["frac_hbm"], "traffic": None, "per_kernel": kern,
                         "note": "achieved = whole-job algorithmic bytes (A read twice per iteration) / wall time; the shard (512 MiB at 8 "
                                 "ranks) exceeds the 256 MiB Infinity Cache, so both GEMVs stream from HBM"},
            "residual": st["residual"]}


def bench_rowsharded_one_process(rls, rank, world, K, W, M=65536, N=8192, devices=None):
    """BASELINE config 5 from ONE host process (the Julia host's shape): rank 0 of the job drives `world` GPUs through the library's
    own communicator (rls_comm_*: per-rank worker threads, the one-shot direct-write all-reduce over xGMI, then RCCL inside the
    library) while the other ranks of the job wait.  Returns (rank 0) iterations/s per transport and the host time the busiest
    worker spent enqueueing per iteration (rls_comm_debug_busy_seconds); None on the other ranks."""
    if rank != 0:
        return None
    out = {}
    shards, parts = [], []
    rng = np.random.default_rng(7)
    x_true = ((rng.standard_normal(N) + 1j * rng.standard_normal(N)) / math.sqrt(2)).astype(np.complex64)
    for r in range(world):
        A, lo, hi = make_row_shard(M, N, r, world)
        shards.append(A)
        parts.append((A @ x_true).astype(np.complex64))
    seg = 32
    devices = list(range(world)) if devices is None else list(devices)
    shared = len(set(devices)) < len(devices)
    for name, transport in (("direct", COMM_DIRECT), ("rccl", COMM_RCCL)):
        if shared and transport == COMM_RCCL:  # a rehearsal on one GPU: RCCL refuses several ranks on one device
            out[name] = {"skipped": "the ranks share a device (rehearsal): the RCCL transport needs a device per rank"}
            continue
        try:
            s = CommRowShardedCGNR(rls, shards, devices=devices, transport=transport, iterations=seg, relTol=0.0)
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {e}"}
            continue
        try:
            def run(n):
                while n > 0:
                    m = min(n, seg)
                    s.init(parts)
                    s.step(m)
                    n -= m

            run(W)
            s.sync()
            busy0 = (C.c_double * world)()
            s.lib.rls_comm_debug_busy_seconds(s.comm, busy0)
            t0 = time.perf_counter()
            run(K)
            s.sync()
            el = time.perf_counter() - t0
            busy1 = (C.c_double * world)()
            s.lib.rls_comm_debug_busy_seconds(s.comm, busy1)
            st = s.status(0)
            out[name] = {"iterations_per_s": K / el, "us_per_iteration": 1e6 * el / K, "transport_code": int(s.transport), "ranks": world,
                         "peer_probe": s.peer_report(),
                         "host_busy_us_per_iteration_per_rank": [1e6 * (b1 - b0) / K for b0, b1 in zip(busy0, busy1)],
                         "residual": st["residual"]}
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            try:
                s.close()
            except Exception:
                pass
    return out
