"""Host-side mirror of the reference's solver API for device arrays:
createLinearSolver / solve! / init! / iterate / solversolution / solverconvergence, the solver types
CGNR, FISTA, ADMM and the matrix-right-hand-side schedulers.

Julia `f!` is `f_` here.  Every numeric step is a call into librls_mi355x.so; this file owns control
flow only (what src/RegularizedLeastSquares.jl:103-131,288-294 and the ctor / init! / iterate
bodies of src/CGNR.jl, src/FISTA.jl, src/ADMM.jl, src/MultiThreading.jl own in the reference).
The call sequences are exactly the ones the Julia extension issues (julia/ in this repo), so
parity of this harness is parity of the extension.
"""
from __future__ import annotations

import ctypes as C
import inspect
import math
import warnings
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import (PROJ_NONE, PROJ_POSITIVE, PROJ_REAL, REG_L1, REG_L2, REG_L21, REG_NONE, REG_TV, AdmmParams,
                   AdmmStatus, CgnrStatus, CgStatus,
                   FistaStatus, check)
from .arrays import DeviceMatrix, DeviceVector, NormalOperator, OperatorHandle, is_double
from .regularization import (AbstractParameterizedRegularization, AbstractProjectionRegularization, GradientOp,
                             L1Regularization, L2Regularization, findsink, findsinks, is_projection, sink,
                             L21Regularization, MeasurementBasedNormalization, NoNormalization, PositiveRegularization,
                             RealRegularization, SystemMatrixBasedNormalization, TVRegularization, normalize)

_EPS32 = float(np.finfo(np.float32).eps)


def _eps_of(op, tol=None):
    """a tolerance keyword's default: eps(real(eltype(AHA))) (src/CGNR.jl:45, src/FISTA.jl:62, src/ADMM.jl:97-98, ...)"""
    return float(np.finfo(_rt_of(op)).eps) if tol is None else tol


def _rt_of(op):
    """the scalars' type follows the element type (rT = real(eltype) in the reference's solver structs)"""
    return np.float64 if getattr(op, "double", False) else np.float32


# --------------------------------------------------------------------------------------------
# shared plumbing
# --------------------------------------------------------------------------------------------


def _as_list(reg):
    if reg is None:
        return []
    return list(reg) if isinstance(reg, (list, tuple)) else [reg]


def _resolve_operator(A, AHA):
    """Returns (A: DeviceMatrix|None, OperatorHandle).  AHA=None means the constructor default
    `A' * A`, which for this backend's operator type is the lazy normal operator (matrix-free)."""
    if A is not None and not isinstance(A, DeviceMatrix):
        raise TypeError("A must be a DeviceMatrix (the backend is selected by the array type, as in the reference)")
    gram = None
    if AHA is None:
        if A is None:
            raise ValueError("either A or AHA is required")
    elif isinstance(AHA, NormalOperator):
        if A is None:
            A = AHA.A
    elif isinstance(AHA, DeviceMatrix):
        gram = AHA
    else:
        raise TypeError("AHA must be a DeviceMatrix (Gram matrix) or a NormalOperator")
    return A, OperatorHandle(A, gram)



def _cg_is_resident(lib, plan) -> bool:
    """a cg! that runs as ONE resident launch can time out as a no-op (another tenant on the device); its status call
    repeats the solve on the per-iteration pipeline then, so x must not be consumed before that call (include/rls_mi355x.h,
    rls_cg_get_status)"""
    path = C.c_int32(-1)
    return lib.rls_cg_path(plan, C.byref(path)) == 0 and path.value in (4, 5)


class AbstractLinearSolver:
    state = None

    # preserved accessors (src/RegularizedLeastSquares.jl:163-183)
    def solversolution(self):
        return solversolution(self.state)

    def solverconvergence(self):
        return solverconvergence(self.state)


# solver categories (src/RegularizedLeastSquares.jl:135-148)
class AbstractRowActionSolver(AbstractLinearSolver):
    pass


class AbstractPrimalDualSolver(AbstractLinearSolver):
    pass


class AbstractProximalGradientSolver(AbstractLinearSolver):
    pass


class AbstractKrylovSolver(AbstractLinearSolver):
    pass


class AbstractSolverState:
    pass


def solverstate(solver):
    return solver.state


def solversolution(obj):
    st = obj.state if isinstance(obj, AbstractLinearSolver) else obj
    if isinstance(obj, AbstractLinearSolver) and hasattr(obj, "_solution") and isinstance(st, KaczmarzState):
        return obj._solution(st)
    if isinstance(st, BatchedState):
        return st.solutions()
    if isinstance(st, AbstractMatrixSolverState):
        return [solversolution(s) for s in st.states]  # hcat of columns (src/MultiThreading.jl:79)
    return st.x


def solverconvergence(obj):
    st = obj.state if isinstance(obj, AbstractLinearSolver) else obj
    return st.convergence()


# --------------------------------------------------------------------------------------------
# CGNR
# --------------------------------------------------------------------------------------------


class CGNRState(AbstractSolverState):
    """src/CGNR.jl:13-24.  x0 is the normal-equation residual (x₀ in the reference)."""

    def __init__(self, relTol):
        self.x = self.x0 = self.pl = self.vl = None
        self.alphal = self.betal = self.zetal = 0.0
        self.iteration = 0
        self.relTol = float(relTol)
        self.z0 = 0.0
        self._plan = None
        self._done = False

    def _refresh(self, lib):
        st = CgnrStatus()
        check(self.x.ctx.handle, lib.rls_cgnr_get_status(self._plan, C.byref(st)), "rls_cgnr_get_status")
        return self._take(st)

    def _step_status(self, lib, n):
        """advance n iterations and read the status back in ONE call (one host synchronisation): rls_cgnr_step_status"""
        st = CgnrStatus()
        check(self.x.ctx.handle, lib.rls_cgnr_step_status(self._plan, int(n), C.byref(st)), "rls_cgnr_step_status")
        return self._take(st)

    def _take(self, st):
        cplx = self.x.dtype.kind == "c"
        self.alphal = complex(st.alpha_re, st.alpha_im) if cplx else st.alpha_re
        self.betal = complex(st.beta_re, st.beta_im) if cplx else st.beta_re
        self.zetal = st.zeta
        self.iteration = st.iteration
        self.z0 = st.z0
        self._done = bool(st.done)
        self._residual = st.residual
        self.fallbacks = int(st.fallbacks)  # resident launches lost to a co-tenant and re-run on the pipeline
        self._status_valid = True
        return st

    def convergence(self):
        if self._plan:
            self._refresh(self.x.ctx.lib)
        return {"residual": self._residual}  # src/CGNR.jl:136

    def __del__(self):
        try:
            if self._plan and self.x is not None and self.x.ctx.handle:
                self.x.ctx.lib.rls_cgnr_destroy(self._plan)
        except Exception:
            pass
        self._plan = None


class CGNR(AbstractKrylovSolver):
    """src/CGNR.jl:48-89"""

    def __init__(self, A=None, *, AHA=None, reg=None, normalizeReg=None, iterations: int = 10, relTol=None):
        self.A, self._op = _resolve_operator(A, AHA)
        self.AHA = AHA if AHA is not None else self.A.normal_operator()
        regs = normalize(normalizeReg, _as_list(reg), self.A, None)
        i2 = findsink(L2Regularization, regs)  # src/CGNR.jl:69 (sink type: nested / scaled L2 terms count)
        l2 = [] if i2 is None else [regs[i2]]
        self.L2 = l2[0] if l2 else L2Regularization(0.0)
        self.constr = [regs[i] for i in findsinks((RealRegularization, PositiveRegularization), regs)]  # :77-78
        rest = [r for r in regs if r not in l2 and r not in self.constr]
        if rest:
            raise ValueError(f"CGNR does not allow for more additional regularization terms, found {len(rest)}")
        self.normalizeReg = normalizeReg or NoNormalization()
        self.iterations = int(iterations)
        self.state = CGNRState(_eps_of(self._op, relTol))

    def _new_state(self):
        st = self.state
        if isinstance(st, AbstractMatrixSolverState) and not isinstance(st, BatchedState):
            st = st.states[0]
        return CGNRState(st.relTol)

    def init_(self, state: CGNRState, b: DeviceVector, x0=0):
        """init!(solver, state, b; x0 = 0)  src/CGNR.jl:91-130"""
        if not (np.isscalar(x0) and x0 == 0):
            # the reference's x0 != 0 branch reads a field that does not exist (src/CGNR.jl:119)
            raise NotImplementedError("CGNR: x0 != 0 is unsupported (it throws in the reference as well)")
        lib = b.ctx.lib
        if self._op.double:
            return self._init_from_primitives(state, b)
        self._prepare(state, b)
        check(b.ctx.handle, lib.rls_cgnr_init(state._plan, b.ptr, float(self.L2.lam), state.relTol, self.iterations),
              "rls_cgnr_init")
        self._after_init(state)

    # ---- Float64 / ComplexF64: the reference's loop on the L1 protocol (rls_*_d), statement by statement ----------------------------
    def _init_from_primitives(self, state: CGNRState, b: DeviceVector):
        """init!  src/CGNR.jl:107-130 on the primitives"""
        N = self._op.N
        self.L2 = normalize(self.normalizeReg, self.L2, self.A, b, in_solver=True)  # :129
        if state.x is None or state.x.ctx is not b.ctx or state.x.dtype != b.dtype or state.x.n != N:
            state.x, state.x0, state.pl, state.vl = (b.similar(N) for _ in range(4))
        expect = self._op.M if self.A is not None else N
        if b.n != expect:
            raise ValueError(f"DimensionMismatch: b has length {b.n}, expected {expect}")
        state.x.fill_(0)                                   # :108-115 (x0 = 0)
        if self.A is not None:
            self.A.mul_adj_(state.x0, b)                   # initCGNR :132
        else:
            state.x0.copy_from(b)                          # :134
        state.z0 = state.x0.norm()                         # :125
        state.pl.copy_from(state.x0)                       # :126
        state.vl.fill_(0)
        state.alphal = state.betal = state.zetal = 0.0
        state._residual = state.z0
        self._after_init(state)
        state._status_valid = True

    def _iterate_from_primitives(self, state: CGNRState):
        """iterate  src/CGNR.jl:143-178 on the primitives (one dot, two norms, three axpy-like broadcasts, one rmul! per iteration)"""
        with np.errstate(all="ignore"):
            conv = bool(np.float64(state._residual) / np.float64(state.z0) <= state.relTol)   # converged(): :181-183 (0 / 0 = NaN: false)
        if conv or state.iteration >= min(self.iterations, self._op.N):   # done(): :181-185
            if not getattr(state, "_finalised", False):
                for r in self.constr:
                    r.prox_(state.x)
                state._finalised = True
            state._done = True
            return None
        cplx = state.x.dtype.kind == "c"
        lam = float(self.L2.lam)
        self._op.mul_normal_(state.vl, state.pl)           # :151
        zeta = state.x0.norm() ** 2                        # :153
        normvl = state.pl.dot(state.vl)                    # :154
        den = normvl + lam * state.pl.norm() ** 2 if lam > 0 else normvl   # :158-160
        with np.errstate(all="ignore"):
            alpha = np.complex128(zeta) / np.complex128(den) if cplx else np.float64(zeta) / np.float64(den)
        alpha = complex(alpha) if cplx else float(alpha)
        state.x.axpy_(alpha, state.pl)                     # :163
        state.x0.axpy_(-alpha, state.vl)                   # :165
        if lam > 0:
            state.x0.axpy_(-lam * alpha, state.pl)         # :168
        rr = state.x0.dot(state.x0)
        with np.errstate(all="ignore"):
            beta = (np.complex128(rr) / np.complex128(zeta)) if cplx else np.float64(rr) / np.float64(zeta)   # :171
        beta = complex(beta) if cplx else float(beta)
        state.pl.rmul_(beta)                               # :173
        state.pl.axpy_(1.0, state.x0)                      # :174
        state.alphal, state.betal, state.zetal = alpha, beta, zeta
        state.iteration += 1
        state._residual = state.x0.norm()
        return state.x, state

    def _prepare(self, state: CGNRState, b: DeviceVector):
        """everything of init! ahead of the device work: the normalised L2 weight, the state vectors and the plan"""
        N = self._op.N
        lib = b.ctx.lib
        self.L2 = normalize(self.normalizeReg, self.L2, self.A, b, in_solver=True)  # :129
        if state.x is None or state.x.ctx is not b.ctx or state.x.dtype != b.dtype or state.x.n != N:
            state.x, state.x0, state.pl, state.vl = (b.similar(N) for _ in range(4))  # similar(b, ...) :92-95
            if state._plan:
                lib.rls_cgnr_destroy(state._plan)
            plan = C.c_void_p()
            check(b.ctx.handle, lib.rls_cgnr_create(self._op.handle, state.x.ptr, state.x0.ptr, state.pl.ptr,
                                                    state.vl.ptr, C.byref(plan)), "rls_cgnr_create")
            state._plan = plan
            state._keep = (self._op, b.ctx)  # destruction order: plan before operator before context
        expect = self._op.M if self.A is not None else N
        if b.n != expect:
            raise ValueError(f"DimensionMismatch: b has length {b.n}, expected {expect}")

    @staticmethod
    def _after_init(state: CGNRState):
        state.iteration = 0
        state._done = False
        state._finalised = False
        state._status_valid = False

    def iterate(self, state: Optional[CGNRState] = None):
        """iterate(solver, state)  src/CGNR.jl:143-178; returns None when done.  One library call per iteration: the step
        and the status read-back (`done`, the convergence record) travel together (rls_cgnr_step_status)"""
        state = state or self.state
        if self._op.double:
            return self._iterate_from_primitives(state)
        lib = state.x.ctx.lib
        if not getattr(state, "_status_valid", False):
            state._refresh(lib)  # first call after init! (or after work enqueued behind the host's back): is it done already?
        if state._done:
            if not getattr(state, "_finalised", False):
                for r in self.constr:  # constraints applied once, at exit  :145-147
                    r.prox_(state.x)
                state._finalised = True
            return None
        state._step_status(lib, 1)
        return state.x, state

    def _run(self, state: CGNRState):
        """no callbacks: enqueue every remaining iteration (no-ops once done) and finalise"""
        if self._op.double:
            while self.iterate(state) is not None:
                pass
            return
        lib = state.x.ctx.lib
        n = max(min(self.iterations, self._op.N) - state.iteration, 0)
        state._step_status(lib, n)
        while self.iterate(state) is not None:  # normally returns None at once
            pass


def solve_group_(solvers, rhs):
    """K independent solves, each solver with its OWN matrix -- the reference's other multi-solve flavour, one solver per problem
    under `Threads.@threads` (docs/src/literate/howto/multi_threading.jl:8-17).  CGNR solvers with the same L2 weight, relTol and
    iteration count on one context run WITHOUT the host in between:
      * small systems (every plan on the single-workgroup kernel, `rls_cgnr_path` 8): ONE launch -- init! and all iterations of all
        K problems, one workgroup per problem (rls_cgnr_init_step_group);
      * anything else: a QUEUE on the context's stream (rls_cgnr_solve_queue) -- problem k's init! and iterations enqueued behind
        problem k - 1's, one read-back for all statuses.  rhs may be HOST arrays (numpy): they are staged through pinned memory,
        uploaded and the solutions downloaded asynchronously inside the same queue (rls_cgnr_solve_queue_host), and the call
        returns host arrays -- the shape of the reference's task, host arrays in and out.
    Anything else (other solvers, different parameters, several contexts): solve_ one after the other.  Returns the solutions in order."""
    from .arrays import DeviceVector
    solvers, rhs = list(solvers), list(rhs)
    if len(solvers) != len(rhs):
        raise ValueError("solve_group_: one right-hand side per solver")
    host = len(rhs) >= 1 and all(isinstance(b, np.ndarray) for b in rhs)
    ok = len(solvers) >= 1 and all(isinstance(s_, CGNR) and isinstance(s_._op, OperatorHandle) and s_.A is not None and
                                  isinstance(s_.state, CGNRState) and not s_._op.double for s_ in solvers)
    if ok and host:
        ctx = solvers[0].A.ctx
        ok = all(s_.A.ctx is ctx and s_.A.dtype == np.dtype(b.dtype) and b.ndim == 1 and b.shape[0] == s_._op.M
                 for s_, b in zip(solvers, rhs))
        if ok:
            for s_, b in zip(solvers, rhs):   # the plan and its state vectors (created once per solver; `b` only lends its type)
                s_._prepare(s_.state, _ShapeOnly(s_._op.M, np.dtype(b.dtype), ctx))
    elif ok:
        ctx = rhs[0].ctx
        ok = all(isinstance(b, DeviceVector) and b.ctx is ctx and b.dtype == rhs[0].dtype for b in rhs)
        if ok:
            for s_, b in zip(solvers, rhs):
                s_._prepare(s_.state, b)
    if ok:
        first = solvers[0]
        ok = all(s_.L2.lam == first.L2.lam and s_.state.relTol == first.state.relTol and s_.iterations == first.iterations
                 for s_ in solvers)
    if not ok:
        if host:
            return [solve_(s_, DeviceVector.from_host(b, s_.A.ctx), _no_group=True).to_host() for s_, b in zip(solvers, rhs)]
        return [solve_(s_, b, _no_group=True) for s_, b in zip(solvers, rhs)]
    lib, K = ctx.lib, len(solvers)
    small = not host
    path = C.c_int32(-1)
    for s_ in solvers:
        check(ctx.handle, lib.rls_cgnr_path(s_.state._plan, C.byref(path)), "rls_cgnr_path")
        small = small and path.value == 8
    plans = (C.c_void_p * K)(*[s_.state._plan for s_ in solvers])
    sts = (CgnrStatus * K)()
    xs_host = None
    if host:
        bs = [np.ascontiguousarray(b) for b in rhs]
        xs_host = [np.empty(s_._op.N, dtype=b.dtype) for s_, b in zip(solvers, bs)]
        bptr = (C.c_void_p * K)(*[b.ctypes.data for b in bs])
        xptr = (C.c_void_p * K)(*[x.ctypes.data for x in xs_host])
        check(ctx.handle, lib.rls_cgnr_solve_queue_host(plans, bptr, xptr, K, float(first.L2.lam), float(first.state.relTol),
                                                        first.iterations, sts), "rls_cgnr_solve_queue_host")
    elif small:
        bptr = (C.c_void_p * K)(*[b.ptr for b in rhs])
        n = min(first.iterations, max(s_._op.N for s_ in solvers))  # (each plan stops at its own min(iterations, N): src/CGNR.jl:185)
        check(ctx.handle, lib.rls_cgnr_init_step_group(plans, bptr, K, float(first.L2.lam), float(first.state.relTol), first.iterations, n),
              "rls_cgnr_init_step_group")
        check(ctx.handle, lib.rls_cgnr_get_status_group(plans, K, sts), "rls_cgnr_get_status_group")   # ONE read-back for the group
    else:
        bptr = (C.c_void_p * K)(*[b.ptr for b in rhs])
        check(ctx.handle, lib.rls_cgnr_solve_queue(plans, bptr, K, float(first.L2.lam), float(first.state.relTol), first.iterations, sts),
              "rls_cgnr_solve_queue")
    out = []
    for k, (s_, st) in enumerate(zip(solvers, sts)):
        CGNR._after_init(s_.state)
        s_.state._take(st)
        s_.state._status_valid = True
        if host and s_.constr:   # the constraints act on the device vector at exit (src/CGNR.jl:145-147): fetch it behind them
            while s_.iterate(s_.state) is not None:
                pass
            xs_host[k] = s_.state.x.to_host()
        elif host:
            s_.state._finalised = True
        else:
            while s_.iterate(s_.state) is not None:   # (returns None at once where the solve is done: applies the constraints)
                pass
        out.append(xs_host[k] if host else s_.state.x)
    return out


class _ShapeOnly:
    """what CGNR._prepare asks of `b` when the right-hand side lives on the host: its length, element type, context and `similar`"""

    def __init__(self, n, dtype, ctx):
        self.n, self.dtype, self.ctx = int(n), np.dtype(dtype), ctx

    def similar(self, n=None):
        from .arrays import DeviceVector
        return DeviceVector(self.n if n is None else int(n), self.dtype, self.ctx)


# --------------------------------------------------------------------------------------------
# FISTA
# --------------------------------------------------------------------------------------------


def power_iterations(AHA, b0: DeviceVector, rtol=1e-3, maxiter=30) -> float:
    """src/Utils.jl:262-287 on device vectors.  The start vector is an argument: the reference draws
    it from Julia's global RNG, which no other runtime can replay."""
    b = b0.copy()
    bold = b0.similar()
    lam_ = math.inf
    for _ in range(maxiter):
        b.rmul_(1.0 / b.norm())
        b, bold = bold, b
        AHA.mul_(b, bold)
        lam_old = lam_
        lam_ = abs(bold.dot(b))
        if abs(lam_ / lam_old - 1) < rtol:
            return lam_
    return lam_


class FISTAState(AbstractSolverState):
    """src/FISTA.jl:15-27"""

    def __init__(self, rho, theta, relTol):
        self.x = self.x0 = self.xold = self.res = None
        self.rho = float(rho)
        self.theta = self.thetaold = float(theta)
        self.iteration = 0
        self.relTol = float(relTol)
        self.norm_x0 = 1.0
        self.rel_res_norm = math.inf
        self._plan = None
        self._bufs = None
        self._done = False

    def _refresh(self, lib):
        st = FistaStatus()
        h = self._bufs[0].ctx.handle
        check(h, lib.rls_fista_get_status(self._plan, C.byref(st)), "rls_fista_get_status")
        return self._take(st)

    def _step_status(self, lib, n):
        """advance n iterations and read the status back in ONE call (rls_fista_step_status)"""
        st = FistaStatus()
        check(self._bufs[0].ctx.handle, lib.rls_fista_step_status(self._plan, int(n), C.byref(st)), "rls_fista_step_status")
        return self._take(st)

    def _take(self, st):
        self.theta, self.thetaold = st.theta, st.theta_old
        self.iteration = st.iteration
        self.rel_res_norm = st.rel_res_norm
        self.norm_x0 = st.norm_x0
        self._residual = st.residual
        self._done = bool(st.done)
        self.fallbacks = int(st.fallbacks)
        # the reference swaps x / xold by pointer every iteration (src/FISTA.jl:144-146)
        self.x, self.xold = (self._bufs[st.iteration & 1], self._bufs[(st.iteration + 1) & 1])
        self._status_valid = True
        return st

    def convergence(self):
        if self._plan:
            self._refresh(self.x.ctx.lib)
            return {"residual": self._residual}
        return {"residual": self.res.norm()}  # src/FISTA.jl:131

    def __del__(self):
        try:
            if self._plan and self._bufs and self._bufs[0].ctx.handle:
                self._bufs[0].ctx.lib.rls_fista_destroy(self._plan)
        except Exception:
            pass
        self._plan = None


class FISTA(AbstractProximalGradientSolver):
    """src/FISTA.jl:57-92.  `rho` defaults to 0.95 / power_iterations(AHA) as in the reference; pass it
    explicitly for reproducible runs (the reference's default depends on the global RNG)."""

    def __init__(self, A=None, *, AHA=None, reg=None, normalizeReg=None, iterations: int = 50, verbose: bool = False,
                 rho=None, theta=1, relTol=None, restart: str = "none"):
        self.A, self._op = _resolve_operator(A, AHA)
        self.AHA = AHA if AHA is not None else self.A.normal_operator()
        regs = _as_list(reg) or [L1Regularization(0.0)]
        self.proj = [r for r in regs if is_projection(r)]
        rest = [r for r in regs if not is_projection(r)]
        if len(rest) != 1:
            raise ValueError(f"FISTA does not allow for more additional regularization terms, found {len(rest)}")
        self.reg = normalize(normalizeReg, rest, self.A, None)[0]
        self.normalizeReg = normalizeReg or NoNormalization()
        self.verbose = bool(verbose)
        if restart not in ("none", "gradient"):
            raise ValueError("restart must be 'none' or 'gradient'")
        self.restart = restart
        self.iterations = int(iterations)
        if rho is None:
            n = self._op.N
            rng = np.random.default_rng()
            v = rng.standard_normal(n)
            if self._op.dtype.kind == "c":
                v = v + 1j * rng.standard_normal(n)
            start = DeviceVector.from_host(v.astype(self._op.dtype), self._op.ctx)
            rho = 0.95 / power_iterations(_NormalApply(self._op), start)
        self.state = FISTAState(rho, theta, _eps_of(self._op, relTol))

    def _fused_kinds(self):
        """(reg_kind, lambda, slices, proj_kind) when the update is fusable, else None"""
        if self._op.double:
            return None  # Float64 / ComplexF64: the reference's loop on the primitives (rls_*_d); the plans are Float32 / ComplexF32
        r = self.reg
        if type(r) is L1Regularization:
            kind, slices = REG_L1, 1
        elif type(r) is L2Regularization and getattr(r, "lam_vector", None) is None:
            kind, slices = REG_L2, 1
        elif type(r) is L21Regularization:
            kind, slices = REG_L21, r.slices
        elif type(r) is TVRegularization and getattr(self, "_tv_unfused", None) not in (True, (tuple(np.atleast_1d(r.shape)), r.dims)):
            # (_tv_unfused: the geometry the plan refused -- or True, the tests' switch for the primitive-by-primitive sequence)
            kind, slices = REG_TV, 1  # the FGP launch between the two halves of the plan's update (rls_fista_set_reg_tv)
        else:
            return None  # nested / transformed / learned terms: the generic path calls their prox_
        if len(self.proj) > 1:
            return None
        pk = PROJ_NONE
        if self.proj:
            if type(self.proj[0]) is PositiveRegularization:
                pk = PROJ_POSITIVE
            elif type(self.proj[0]) is RealRegularization:
                pk = PROJ_REAL
            else:
                return None
        return kind, float(r.lam), slices, pk

    def _new_state(self):
        s = self.state.states[0] if isinstance(self.state, AbstractMatrixSolverState) else self.state
        return FISTAState(s.rho, s.theta if s.iteration == 0 else 1.0, s.relTol)

    def init_(self, state: FISTAState, b: DeviceVector, x0=0, theta=1):
        """init!(solver, state, b; x0 = 0, theta = 1)   src/FISTA.jl:94-129"""
        N = self._op.N
        lib, h = b.ctx.lib, b.ctx.handle
        if isinstance(self.normalizeReg, MeasurementBasedNormalization):  # :128 normalises with x0 = A^H b
            x0n = b if self.A is None else self.A.mul_adj_(b.similar(N), b)
            self.reg = normalize(self.normalizeReg, self.reg, self.A, x0n, in_solver=True)
        fused = self._fused_kinds()
        fresh = state._bufs is None or state._bufs[0].ctx is not b.ctx or state._bufs[0].dtype != b.dtype
        if fresh:
            state._bufs = [b.similar(N), b.similar(N)]
            state.x0, state.res = b.similar(N), b.similar(N)
            if state._plan:
                lib.rls_fista_destroy(state._plan)
                state._plan = None
            if fused is not None:
                plan = C.c_void_p()
                check(h, lib.rls_fista_create(self._op.handle, state._bufs[0].ptr, state.x0.ptr, state._bufs[1].ptr,
                                              state.res.ptr, C.byref(plan)), "rls_fista_create")
                state._plan = plan
                state._keep = (self._op, b.ctx)
        state.x, state.xold = state._bufs
        if fused is not None and fused[0] == REG_TV:
            from .regularization import _tv_geometry
            shape, d0, cs, cd = _tv_geometry(self.reg.shape, self.reg.dims)
            st_tv = lib.rls_fista_set_reg_tv(state._plan, fused[1], len(shape), cs, len(d0), cd, self.reg.iterationsTV, fused[3])
            if st_tv == -2:  # RLS_E_UNSUPPORTED: the image does not fit the plan's single-workgroup FGP launch -- primitives
                self._tv_unfused = (tuple(np.atleast_1d(self.reg.shape)), self.reg.dims)   # (THIS geometry: another one is tried afresh)
                lib.rls_fista_destroy(state._plan)
                state._plan = None
                fused = None
            else:
                check(h, st_tv, "rls_fista_set_reg_tv")
        if fused is not None:
            kind, lam_, slices, pk = fused
            if kind != REG_TV:
                check(h, lib.rls_fista_set_reg(state._plan, kind, lam_, slices, pk), "rls_fista_set_reg")
            check(h, lib.rls_fista_init(state._plan, b.ptr, state.rho, float(theta), state.relTol, self.iterations,
                                        1 if self.restart == "gradient" else 0), "rls_fista_init")
            if not (np.ndim(x0) == 0 and not isinstance(x0, DeviceVector) and x0 == 0):
                if isinstance(x0, DeviceVector):
                    xs = x0
                elif np.ndim(x0) == 0:   # `state.x .= x0` broadcasts a scalar (src/FISTA.jl:120)
                    xs = DeviceVector.from_host(np.full(N, x0, dtype=b.dtype), b.ctx)
                else:
                    xs = DeviceVector.from_host(np.ascontiguousarray(np.asarray(x0, dtype=b.dtype).reshape(-1)), b.ctx)
                if xs.n != N or xs.dtype != b.dtype:
                    raise ValueError(f"DimensionMismatch: x0 has length {xs.n} ({xs.dtype}), the solution {N} ({b.dtype})")
                check(h, lib.rls_fista_set_start(state._plan, xs.ptr, xs.n), "rls_fista_set_start")
        else:
            # generic path from primitives (TV prox etc.)
            if self.A is None:
                state.x0.copy_from(b)
            else:
                self.A.mul_adj_(state.x0, b)
            state.norm_x0 = state.x0.norm()
            if np.isscalar(x0):
                state.x.fill_(x0)
            else:
                state.x.copy_from(x0 if isinstance(x0, DeviceVector) else DeviceVector.from_host(np.asarray(x0, dtype=b.dtype), b.ctx))
            state.xold.fill_(0)
            state.res.fill_(math.inf)
            state.rel_res_norm = math.inf
        state._status_valid = False
        state.iteration = 0
        state.theta = state.thetaold = float(theta)
        state._done = False

    def iterate(self, state: Optional[FISTAState] = None):
        state = state or self.state
        if state._plan:
            lib = state.x.ctx.lib
            if not getattr(state, "_status_valid", False):
                state._refresh(lib)  # first call after init!: done already?
            if state._done:
                return None
            state._step_status(lib, 1)  # one library call per iteration: step + status (one host synchronisation)
            return state.x, state
        return self._iterate_generic(state)

    def _iterate_generic(self, state: FISTAState):
        """src/FISTA.jl:139-185 from primitives (one host read-back for the residual norm)"""
        if state.rel_res_norm < state.relTol or state.iteration >= self.iterations:
            return None
        f32 = np.float64 if self._op.double else np.float32   # (the scalars' type follows the element type, as rT does in the reference)
        state.x, state.xold = state.xold, state.x
        th, tho = f32(state.theta), f32(state.thetaold)
        state.x.rmul_(float((f32(1) - tho) / th))
        state.x.axpy_(float((tho - f32(1)) / th + f32(1)), state.xold)
        _NormalApply(self._op).mul_(state.res, state.x)
        state.res.axpy_(-1.0, state.x0)
        state.x.axpy_(-state.rho, state.res)
        state.rel_res_norm = state.res.norm() / state.norm_x0
        if self.verbose:
            print(f"Iteration {state.iteration}; rel. residual = {state.rel_res_norm}")
        self.reg.prox_(state.x, float(f32(state.rho) * f32(self.reg.lam)))
        for pr in self.proj:
            pr.prox_(state.x)
        if self.restart == "gradient":
            d = state.x.copy().axpy_(-1.0, state.xold)
            if np.real(state.res.dot(d)) > 0:
                state.theta = 1.0
        state.thetaold = state.theta
        t = f32(state.thetaold)
        state.theta = float((f32(1) + np.sqrt(f32(1) + f32(4) * t * t)) / f32(2))
        state.iteration += 1
        return state.x, state

    def _run(self, state: FISTAState):
        if state._plan:
            lib = state.x.ctx.lib
            state._step_status(lib, max(self.iterations - state.iteration, 0))
        while self.iterate(state) is not None:
            pass


class _NormalApply:
    """mul!(v, AHA, p) through the operator handle"""

    def __init__(self, op: OperatorHandle):
        self.op = op

    def mul_(self, v, p):
        return self.op.mul_normal_(v, p)


# --------------------------------------------------------------------------------------------
# ADMM
# --------------------------------------------------------------------------------------------


class _Identity:
    """opEye (src/ADMM.jl:84)"""

    identity = True

    def __init__(self, n):
        self.n_out = n

    def mul_(self, z, x, alpha=1.0, beta=0.0):
        return z.lincomb_(alpha, x, beta, z) if beta != 0 else z.lincomb_(alpha, x, 0.0, x)

    def mul_adj_(self, x, z, alpha=1.0, beta=0.0):
        return x.lincomb_(alpha, z, beta, x) if beta != 0 else x.lincomb_(alpha, z, 0.0, z)


class ADMMState(AbstractSolverState):
    """src/ADMM.jl:19-46"""

    def __init__(self, nreg, rho, absTol, relTol, tolInner):
        self.x = self.xold = self.beta = self.beta_y = None
        self.z: List[DeviceVector] = []
        self.zold: List[DeviceVector] = []
        self.u: List[DeviceVector] = []
        self.uold: List[DeviceVector] = []
        self.rho = np.array(rho, dtype=np.float32)
        self.iteration = 0
        self.rk = np.full(nreg, np.inf, np.float32)
        self.sk = np.full(nreg, np.inf, np.float32)
        self.eps_pri = np.zeros(nreg, np.float32)
        self.eps_dua = np.zeros(nreg, np.float32)
        self.sigma_abs = np.float32(0)
        self.Delta = np.full(nreg, np.inf, np.float32)
        self.absTol, self.relTol, self.tolInner = np.float32(absTol), np.float32(relTol), np.float32(tolInner)
        self.cg_u = self.cg_r = self.cg_c = None
        self._cg = None
        self._admm = None       # rls_admm plan (whole outer iterations on the device) ...
        self._plan_ok = False   # ... and whether this solve runs through it
        self._zbufs = None
        self.cg_iterations: List[int] = []

    def convergence(self):
        return {"primal": self.rk.copy(), "dual": self.sk.copy()}  # src/ADMM.jl:222

    def __del__(self):
        try:
            if self._admm and self.x is not None and self.x.ctx.handle:
                self.x.ctx.lib.rls_admm_destroy(self._admm)
            if self._cg and self.x is not None and self.x.ctx.handle:
                self.x.ctx.lib.rls_cg_destroy(self._cg)
        except Exception:
            pass
        self._cg = self._admm = None


class DiagonalPreconditioner:
    """a left preconditioner for the inner `cg!` of ADMM / SplitBregman (`precon` keyword, src/ADMM.jl:82,244): Pl \\ r =
    r ./ d for a diagonal d given as a device vector of the solver's element type (Jacobi scaling: d = diag(AHA) + sum rho).
    Any object with `ldiv_(out, r)` on device vectors is accepted in its place (the reference takes anything `ldiv!` accepts)."""

    def __init__(self, d: DeviceVector):
        h = d.to_host()
        self.dinv = DeviceVector.from_host((1.0 / h).astype(h.dtype), d.ctx)

    def ldiv_(self, out: DeviceVector, r: DeviceVector):
        lib, h = r.ctx.lib, r.ctx.handle
        # an N-vector is an N x 1 matrix: diag(dinv) * r through the row-scaling kernel (csrc/setup.hip)
        check(h, lib.rls_scale_rows(h, r.code, r.n, 1, self.dinv.ptr, r.ptr, r.n, out.ptr, out.n), "rls_scale_rows")
        return out


class ADMM(AbstractPrimalDualSolver):
    """src/ADMM.jl:80-162"""

    def __init__(self, A=None, *, AHA=None, precon=None, reg=None, regTrafo=None, normalizeReg=None, rho=1e-1,
                 vary_rho: str = "none", iterations: int = 10, iterationsCG: int = 10, absTol=None, relTol=None,
                 tolInner=1e-5, verbose: bool = False):
        if precon is not None and not hasattr(precon, "ldiv_"):
            raise TypeError("precon: an object with ldiv_(out, r) on device vectors (e.g. DiagonalPreconditioner), or None = Identity()")
        self.precon = precon  # Pl of the inner cg! (src/ADMM.jl:82,244); None = Identity(): the fused cg! kernels
        self.A, self._op = _resolve_operator(A, AHA)
        self.AHA = AHA if AHA is not None else self.A.normal_operator()
        regs = _as_list(reg) or [L1Regularization(0.0)]
        self.proj = [r for r in regs if is_projection(r)]
        self.reg = normalize(normalizeReg, [r for r in regs if not is_projection(r)], self.A, None)
        n = self._op.N
        trafos = _as_list(regTrafo) or [_Identity(n) for _ in self.reg]
        if len(trafos) != len(self.reg):
            raise AssertionError("reg and regTrafo must have the same length")
        self.regTrafo = trafos
        self.rho = [float(rho)] * len(self.reg) if np.isscalar(rho) else [float(r) for r in rho]
        if vary_rho not in ("none", "balance", "PnP"):
            raise ValueError("vary_rho must be 'none', 'balance' or 'PnP'")
        self.vary_rho = vary_rho
        self.verbose = bool(verbose)
        self.iterations = int(iterations)
        self.iterationsCG = int(iterationsCG)
        self.normalizeReg = normalizeReg or NoNormalization()
        self._track_cg = True  # record the inner CG iteration counts (one extra host read-back per outer iteration)
        self.use_device_plan = True  # whole outer iterations through rls_admm_step when the regulariser allows
        self.state = ADMMState(len(self.reg), self.rho, _eps_of(self._op, absTol), _eps_of(self._op, relTol), tolInner)

    def _new_state(self):
        s = self.state.states[0] if isinstance(self.state, AbstractMatrixSolverState) else self.state
        return ADMMState(len(self.reg), self.rho, s.absTol, s.relTol, s.tolInner)

    def _all_identity(self):
        return all(getattr(t, "identity", False) for t in self.regTrafo)

    def init_(self, state: ADMMState, b: DeviceVector, x0=0):
        """src/ADMM.jl:166-220"""
        N = self._op.N
        lib, h = b.ctx.lib, b.ctx.handle
        self.reg = normalize(self.normalizeReg, self.reg, self.A, b, in_solver=True)  # :219
        if state.x is None or state.x.ctx is not b.ctx or state.x.dtype != b.dtype:
            state.x, state.xold, state.beta, state.beta_y = (b.similar(N) for _ in range(4))
            state.z = [b.similar(t.n_out) for t in self.regTrafo]
            state.zold = [b.similar(t.n_out) for t in self.regTrafo]
            state.u = [b.similar(t.n_out) for t in self.regTrafo]
            state.uold = [b.similar(t.n_out) for t in self.regTrafo]
            state.cg_u, state.cg_r, state.cg_c = (b.similar(N) for _ in range(3))  # CGStateVariables :129,177
            if state._admm:
                lib.rls_admm_destroy(state._admm)
                state._admm = None
            if state._cg:
                lib.rls_cg_destroy(state._cg)
                state._cg = None
            if not self._op.double:   # (Float64 / ComplexF64: cg! from the primitives, _cg_generic; the plans are Float32 / ComplexF32)
                plan = C.c_void_p()
                check(h, lib.rls_cg_create(self._op.handle, state.cg_u.ptr, state.cg_r.ptr, state.cg_c.ptr, C.byref(plan)),
                      "rls_cg_create")
                state._cg = plan
            state._keep = (self._op, b.ctx)
            state._zbufs = None
        if len(self.reg) == 1:
            if state._zbufs is None:
                state._zbufs = (state.z[0], state.zold[0])
            state.z[0], state.zold[0] = state._zbufs  # the device plan starts every solve with z in the first buffer
        if np.isscalar(x0):
            state.x.fill_(x0)
        else:
            state.x.copy_from(x0 if isinstance(x0, DeviceVector) else DeviceVector.from_host(np.asarray(x0, dtype=b.dtype), b.ctx))
        if self.A is None:
            state.beta_y.copy_from(b)
        else:
            self.A.mul_adj_(state.beta_y, b)
        for i, t in enumerate(self.regTrafo):
            t.mul_(state.z[i], state.x)
            state.u[i].fill_(0)
        state.rk[:] = np.inf
        state.sk[:] = np.inf
        state.eps_pri[:] = 0
        state.eps_dua[:] = 0
        if self._op.double and state.rk.dtype != np.float64:   # the scalars follow the element type (rT in the reference, src/ADMM.jl:19-46)
            for name in ("rho", "rk", "sk", "eps_pri", "eps_dua", "Delta"):
                setattr(state, name, getattr(state, name).astype(np.float64))
            state.absTol, state.relTol, state.tolInner = (np.float64(v) for v in (state.absTol, state.relTol, state.tolInner))
        rt = np.float64 if self._op.double else np.float32
        state.sigma_abs = rt(np.sqrt(rt(b.n))) * state.absTol
        state.Delta[:] = np.inf
        state.rho[:] = self.rho
        state.iteration = 0
        state.cg_iterations = []
        self._init_plan(state)

    def _plan_params(self, state):
        """rls_admm_params when the whole outer iteration can run on the device, else None"""
        if type(self) not in (ADMM, SplitBregman) or not self.use_device_plan or self.precon is not None or self._op.double or not (
                self._all_identity() and len(self.reg) == 1 and self.vary_rho == "none" and not self.verbose):
            return None
        reg, rho = self.reg[0], np.float32(state.rho[0])
        P = AdmmParams()
        if rho == 0:
            P.reg_kind = REG_NONE
        elif type(reg) is L1Regularization:
            P.reg_kind = REG_L1
        elif type(reg) is L2Regularization and getattr(reg, "lam_vector", None) is None:
            P.reg_kind = REG_L2
        elif type(reg) is TVRegularization and 1 <= len(reg.shape) <= 4:
            from .regularization import _tv_geometry
            shape, d0, _, _ = _tv_geometry(reg.shape, reg.dims)
            if int(np.prod(shape)) != self._op.N or len(d0) > 4:
                return None
            P.reg_kind = REG_TV
            P.tv_ndims, P.tv_ntv, P.tv_iterations = len(shape), len(d0), reg.iterationsTV
            for k, v in enumerate(shape):
                P.tv_shape[k] = v
            for k, v in enumerate(d0):
                P.tv_dims[k] = v
        else:
            return None
        if len(self.proj) > 1 or (self.proj and P.reg_kind == REG_TV):
            return None
        P.proj_kind = PROJ_NONE
        if self.proj:
            if type(self.proj[0]) is PositiveRegularization:
                P.proj_kind = PROJ_POSITIVE
            elif type(self.proj[0]) is RealRegularization:
                P.proj_kind = PROJ_REAL
            else:
                return None
        with np.errstate(divide="ignore"):
            P.prox_lambda = 0.0 if rho == 0 else float(self._prox_threshold(np.float32(reg.lam), rho))
        P.x, P.xold, P.beta, P.beta_y = state.x.ptr, state.xold.ptr, state.beta.ptr, state.beta_y.ptr
        P.z0, P.z1, P.u = state._zbufs[0].ptr, state._zbufs[1].ptr, state.u[0].ptr
        P.rho, P.sigma_abs, P.rel_tol = float(rho), float(state.sigma_abs), float(state.relTol)
        P.iterations, P.iterations_cg, P.tol_inner = self._plan_iterations(), self.iterationsCG, float(state.tolInner)
        return P

    def _prox_threshold(self, lam32, rho32):
        return lam32 / (np.float32(2) * rho32)  # prox!(reg, z, lambda / (2 rho))   src/ADMM.jl:261

    def _plan_iterations(self):
        return self.iterations

    def _init_plan(self, state):
        state._plan_ok = False
        P = self._plan_params(state)
        if P is None:
            return
        lib, h = state.x.ctx.lib, state.x.ctx.handle
        if not state._admm:
            plan = C.c_void_p()
            check(h, lib.rls_admm_create(state._cg, C.byref(plan)), "rls_admm_create")
            state._admm = plan
        st = lib.rls_admm_init(state._admm, C.byref(P))
        if st == -2:  # RLS_E_UNSUPPORTED
            return  # e.g. a TV image too large for one workgroup: per-call path (still device kernels)
        check(h, st, "rls_admm_init")
        state._plan_ok = True

    def _plan_advance(self, state, n_outer):
        """n_outer iterations through the device plan, then ONE read-back of the scalars and the log"""
        lib, h = state.x.ctx.lib, state.x.ctx.handle
        check(h, lib.rls_admm_step(state._admm, int(n_outer)), "rls_admm_step")
        st = AdmmStatus()
        cap = max(self.iterations, 1)
        log = (C.c_float * (8 * cap))()
        check(h, lib.rls_admm_get_status(state._admm, C.byref(st), log, cap), "rls_admm_get_status")
        state.fallbacks = int(st.fallbacks)
        it = int(st.iteration)
        if it > 0:
            state.Delta[0], state.sk[0], state.eps_pri[0] = st.delta, st.sk, st.eps_pri
            state.rk[0], state.eps_dua[0] = st.rk, st.eps_dua
        state.cg_iterations = [int(log[8 * k + 5]) for k in range(it)]
        state.iteration = it
        state.z[0], state.zold[0] = (state._zbufs[1], state._zbufs[0]) if it & 1 else state._zbufs
        return bool(st.done)

    def converged(self, state):
        for i in range(len(self.reg)):
            if state.rk[i] >= state.sigma_abs + state.relTol * state.eps_pri[i]:
                return False
            if state.sk[i] >= state.sigma_abs + state.relTol * state.eps_dua[i]:
                return False
        return True

    def done(self, state):
        return self.converged(state) or state.iteration >= self.iterations

    def _composite_mul(self, state, out, v, tmp_list):
        """compositeAHA * v = AHA v + sum_i rho_i Phi_i^H Phi_i v   (src/ADMM.jl:141-159)"""
        self._op.mul_normal_(out, v)
        for i, t in enumerate(self.regTrafo):
            if getattr(t, "identity", False):
                out.axpy_(float(state.rho[i]), v)
            else:
                t.mul_(tmp_list[i], v)
                t.mul_adj_(out, tmp_list[i], float(state.rho[i]), 1.0)
        return out

    def _cg_precond(self, state):
        """IterativeSolvers.cg!(x, AHA, b; Pl = precon) from primitives: the preconditioned recurrence (oracle `_pcg_inplace`,
        pinned iterate by iterate against SciPy's PCG in tests/test_oracle.py): c = Pl \\ r; rho = <c, r>; u = c + (rho / rho_prev) u;
        c = A u; alpha = rho / <u, c>; the stopping test stays on ||r||"""
        f32 = np.float64 if self._op.double else np.float32
        x, b = state.x, state.beta
        u, r, c = state.cg_u, state.cg_r, state.cg_c
        tmp = state.zold
        u.fill_(0)
        r.copy_from(b)
        self._composite_mul(state, c, x, tmp)
        r.axpy_(-1.0, c)
        residual = f32(r.norm())
        tol = max(f32(state.tolInner) * residual, f32(0))
        rho = complex(1.0)
        it = 0
        while it < self.iterationsCG and residual > tol:
            self.precon.ldiv_(c, r)
            rho_prev, rho = rho, complex(c.dot(r))
            with np.errstate(all="ignore"):   # a breakdown (<c, r> = 0) propagates NaN as the reference's arithmetic does, no exception
                ratio = np.complex128(rho) / np.complex128(rho_prev)
            u.lincomb_(1.0, c, complex(ratio) if r.dtype.kind == "c" else float(ratio.real), u)
            self._composite_mul(state, c, u, tmp)
            with np.errstate(all="ignore"):
                alpha = complex(np.complex128(rho) / np.complex128(complex(u.dot(c))))
            if r.dtype.kind != "c":
                alpha = alpha.real
            x.axpy_(alpha, u)
            r.axpy_(-alpha, c)
            residual = f32(r.norm())
            it += 1
        return it

    def _cg_generic(self, state):
        """IterativeSolvers.cg! from primitives (non-identity regTrafo; Float64 / ComplexF64 element types); see oracle cg_inplace"""
        f32 = np.float64 if self._op.double else np.float32
        x, b = state.x, state.beta
        u, r, c = state.cg_u, state.cg_r, state.cg_c
        tmp = state.zold  # free at this point of the iteration
        u.fill_(0)
        r.copy_from(b)
        self._composite_mul(state, c, x, tmp)
        r.axpy_(-1.0, c)
        residual = f32(r.norm())
        tol = max(f32(state.tolInner) * residual, f32(0))
        prev = f32(1)
        it = 0
        while it < self.iterationsCG and residual > tol:
            beta = residual * residual / (prev * prev)
            u.lincomb_(1.0, r, float(beta), u)
            self._composite_mul(state, c, u, tmp)
            alpha = complex(residual * residual) / complex(u.dot(c))
            x.axpy_(alpha, u)
            r.axpy_(-alpha, c)
            prev = residual
            residual = f32(r.norm())
            it += 1
        return it

    def iterate(self, state: Optional[ADMMState] = None):
        """src/ADMM.jl:230-322"""
        state = state or self.state
        if self.done(state):
            return None
        if state._plan_ok:
            self._plan_advance(state, 1)
            return state.x, state
        f32 = np.float64 if self._op.double else np.float32
        lib, h = state.x.ctx.lib, state.x.ctx.handle
        fused = self._all_identity() and len(self.reg) == 1 and not self._op.double
        # 1. x update                                                                  :236-244
        if fused:
            check(h, lib.rls_admm_pre(h, state.x.code, state.x.n, state.beta.ptr, state.beta_y.ptr, state.z[0].ptr,
                                      state.u[0].ptr, state.x.ptr, state.xold.ptr, float(state.rho[0]), 0), "rls_admm_pre")
        else:
            state.beta.copy_from(state.beta_y)
            for i, t in enumerate(self.regTrafo):
                t.mul_adj_(state.beta, state.z[i], float(state.rho[i]), 1.0)
                t.mul_adj_(state.beta, state.u[i], -float(state.rho[i]), 1.0)
            state.xold.copy_from(state.x)
        if self.precon is not None:   # cg!(...; Pl = precon)   :244
            state.cg_iterations.append(self._cg_precond(state))
        elif self._all_identity() and not self._op.double:
            rho_sum = float(np.sum(state.rho, dtype=np.float32))
            check(h, lib.rls_cg_solve(state._cg, state.x.ptr, state.beta.ptr, rho_sum, self.iterationsCG,
                                      float(state.tolInner)), "rls_cg_solve")
            if not fused or self.verbose or self._track_cg or _cg_is_resident(lib, state._cg):
                st = CgStatus()
                check(h, lib.rls_cg_get_status(state._cg, C.byref(st)), "rls_cg_get_status")
                state.cg_iterations.append(int(st.iterations))
        else:
            state.cg_iterations.append(self._cg_generic(state))
        for pr in self.proj:
            pr.prox_(state.x)
        if fused:
            self._zu_update_fused(state)
            state.iteration += 1
            return state.x, state
        # 2./3. z and u updates + convergence bookkeeping                              :251-309
        for i, t in enumerate(self.regTrafo):
            state.z[i], state.zold[i] = state.zold[i], state.z[i]
            t.mul_(state.z[i], state.x)
            state.z[i].axpy_(1.0, state.u[i])
            if state.rho[i] != 0:
                self.reg[i].prox_(state.z[i], float(f32(self.reg[i].lam) / (f32(2) * state.rho[i])))
            state.uold[i].copy_from(state.u[i])
            t.mul_(state.u[i], state.x, 1.0, 1.0)
            state.u[i].axpy_(-1.0, state.z[i])
            state.xold.lincomb_(1.0, state.x, -1.0, state.xold)
            state.zold[i].lincomb_(1.0, state.z[i], -1.0, state.zold[i])
            state.uold[i].lincomb_(1.0, state.u[i], -1.0, state.uold[i])
            Delta_old = state.Delta[i]
            state.Delta[i] = f32(state.xold.norm()) + f32(state.zold[i].norm()) + f32(state.uold[i].norm())
            t.mul_adj_(state.xold, state.zold[i])
            state.sk[i] = state.rho[i] * f32(state.xold.norm())
            t.mul_(state.zold[i], state.x)
            state.eps_pri[i] = max(f32(state.zold[i].norm()), f32(state.z[i].norm()))
            state.zold[i].axpy_(-1.0, state.z[i])
            state.rk[i] = f32(state.zold[i].norm())
            t.mul_adj_(state.xold, state.u[i])
            state.eps_dua[i] = state.rho[i] * f32(state.xold.norm())
            with np.errstate(divide="ignore", invalid="ignore"):
                if (self.vary_rho == "balance" and state.rk[i] / state.eps_pri[i] > f32(10) * state.sk[i] / state.eps_dua[i]) or (
                        self.vary_rho == "PnP" and state.Delta[i] / Delta_old > f32(0.9)):
                    state.rho[i] *= f32(2)
                    state.u[i].rmul_(0.5)
                elif self.vary_rho == "balance" and state.sk[i] / state.eps_dua[i] > f32(10) * state.rk[i] / state.eps_pri[i]:
                    state.rho[i] /= f32(2)
                    state.u[i].rmul_(2.0)
            if self.verbose:
                print(f"rk[{i}]/eps_pri[{i}] = {state.rk[i] / state.eps_pri[i]}")
                print(f"sk[{i}]/eps_dua[{i}] = {state.sk[i] / state.eps_dua[i]}")
                print(f"new rho[{i}] = {state.rho[i]}")
        state.iteration += 1
        return state.x, state

    def _zu_update_fused(self, state):
        """src/ADMM.jl:251-309 for one regulariser with the identity regTrafo: the z-update, then ONE
        launch for u += x - z and all seven norms of the convergence bookkeeping, ONE host read-back"""
        f32 = np.float32
        lib, h = state.x.ctx.lib, state.x.ctx.handle
        i = 0
        state.z[i], state.zold[i] = state.zold[i], state.z[i]
        state.z[i].lincomb_(1.0, state.x, 1.0, state.u[i])
        if state.rho[i] != 0:
            self.reg[i].prox_(state.z[i], float(f32(self.reg[i].lam) / (f32(2) * state.rho[i])))
        out = (C.c_float * 6)()
        check(h, lib.rls_admm_post(h, state.x.code, state.x.n, state.x.ptr, state.xold.ptr, state.z[i].ptr,
                                   state.zold[i].ptr, state.u[i].ptr, out), "rls_admm_post")
        Delta_old = state.Delta[i]
        state.Delta[i] = f32(out[0])
        state.sk[i] = state.rho[i] * f32(out[1])
        state.eps_pri[i] = f32(out[2])
        state.rk[i] = f32(out[3])
        state.eps_dua[i] = state.rho[i] * f32(out[4])
        with np.errstate(divide="ignore", invalid="ignore"):
            if (self.vary_rho == "balance" and state.rk[i] / state.eps_pri[i] > f32(10) * state.sk[i] / state.eps_dua[i]) or (
                    self.vary_rho == "PnP" and state.Delta[i] / Delta_old > f32(0.9)):
                state.rho[i] *= f32(2)
                state.u[i].rmul_(0.5)
            elif self.vary_rho == "balance" and state.sk[i] / state.eps_dua[i] > f32(10) * state.rk[i] / state.eps_pri[i]:
                state.rho[i] /= f32(2)
                state.u[i].rmul_(2.0)
        if self.verbose:
            with np.errstate(divide="ignore", invalid="ignore"):  # (Inf / NaN print as such, as Julia prints them: src/ADMM.jl:311-315)
                print(f"rk/eps_pri = {state.rk[i] / state.eps_pri[i]}  sk/eps_dua = {state.sk[i] / state.eps_dua[i]}  rho = {state.rho[i]}")

    def _run(self, state):
        if state._plan_ok and not self.done(state):
            self._plan_advance(state, self.iterations - state.iteration)  # `done` is evaluated on the device
        while self.iterate(state) is not None:
            pass


# --------------------------------------------------------------------------------------------
# next tier: OptISTA, POGM, SplitBregman -- re-sequencing of the same device kernels
# --------------------------------------------------------------------------------------------


def _default_rho(op):
    """0.95 / power_iterations(AHA) with a NumPy-seeded start vector (the reference uses Julia's global RNG)"""
    n = op.N
    rng = np.random.default_rng()
    v = rng.standard_normal(n)
    if op.dtype.kind == "c":
        v = v + 1j * rng.standard_normal(n)
    return 0.95 / power_iterations(_NormalApply(op), DeviceVector.from_host(v.astype(op.dtype), op.ctx))


def _split_regs(regs, name):
    regs = _as_list(regs) or [L1Regularization(0.0)]
    proj = [r for r in regs if is_projection(r)]
    rest = [r for r in regs if not is_projection(r)]
    if len(rest) != 1:
        raise ValueError(f"{name} does not allow for more additional regularization terms, found {len(rest)}")
    return rest[0], proj


def _fusable_kinds(reg, proj):
    """(reg_kind, proj_kind) when prox + projection run elementwise inside the fused update kernels, else None"""
    if type(reg) is L1Regularization:
        kind = REG_L1
    elif type(reg) is L2Regularization and getattr(reg, "lam_vector", None) is None:
        kind = REG_L2
    else:
        return None
    if len(proj) > 1:
        return None
    pk = PROJ_NONE
    if proj:
        if type(proj[0]) is PositiveRegularization:
            pk = PROJ_POSITIVE
        elif type(proj[0]) is RealRegularization:
            pk = PROJ_REAL
        else:
            return None
    return kind, pk


class _ProxGradState(AbstractSolverState):
    def __init__(self, rho, theta, relTol, names):
        self.rho = float(rho)
        self.theta = self.thetaold = float(theta)
        self.relTol = float(relTol)
        self.iteration = 0
        self.norm_x0 = 1.0
        self.rel_res_norm = math.inf
        self._names = names
        for n in names:
            setattr(self, n, None)

    def _alloc(self, b, N):
        if self.x is None or self.x.ctx is not b.ctx or self.x.dtype != b.dtype or self.x.n != N:
            for n in self._names:
                setattr(self, n, b.similar(N))

    def convergence(self):
        return {"residual": self.res.norm()}


class OptISTA(AbstractProximalGradientSolver):
    """src/OptISTA.jl:61-110 (ctor), :129-160 (init!), :169-209 (iterate)"""

    def __init__(self, A=None, *, AHA=None, reg=None, normalizeReg=None, iterations: int = 50, verbose: bool = False,
                 rho=None, theta=1, relTol=None):
        self.A, self._op = _resolve_operator(A, AHA)
        self.AHA = AHA if AHA is not None else self.A.normal_operator()
        self.reg, self.proj = _split_regs(reg, "OptISTA")
        self.reg = normalize(normalizeReg, [self.reg], self.A, None)[0]
        self.normalizeReg = normalizeReg or NoNormalization()
        self.verbose = bool(verbose)
        self.iterations = int(iterations)
        self.state = _ProxGradState(_default_rho(self._op) if rho is None else rho, theta, _eps_of(self._op, relTol),
                                    ("x", "x0", "y", "z", "zold", "res"))

    def _new_state(self):
        s = self.state.states[0] if isinstance(self.state, AbstractMatrixSolverState) else self.state
        return _ProxGradState(s.rho, 1.0, s.relTol, s._names)

    def init_(self, st, b: DeviceVector, x0=0, theta=1):
        f32 = _rt_of(self._op)
        st._alloc(b, self._op.N)
        if self.A is None:
            st.x0.copy_from(b)
        else:
            self.A.mul_adj_(st.x0, b)
        st.norm_x0 = st.x0.norm()
        self.reg = normalize(self.normalizeReg, self.reg, self.A, st.x0, in_solver=True)  # src/OptISTA.jl:154
        if np.isscalar(x0):
            st.x.fill_(x0)
        else:
            st.x.copy_from(x0 if isinstance(x0, DeviceVector) else DeviceVector.from_host(np.asarray(x0, dtype=b.dtype), b.ctx))
        for v in (st.y, st.z, st.zold):
            v.copy_from(st.x)
        st.res.fill_(math.inf)
        st.theta = st.thetaold = float(theta)
        tn = f32(theta)
        for _ in range(self.iterations - 1):
            tn = (f32(1) + np.sqrt(f32(1) + f32(4) * tn * tn)) / f32(2)
        st.theta_n = float((f32(1) + np.sqrt(f32(1) + f32(8) * tn * tn)) / f32(2))
        st.rel_res_norm = math.inf
        st.iteration = 0

    def _coefficients(self, st):
        """the index-only scalars of one iteration (src/OptISTA.jl:170-175,196-204), advancing theta"""
        f32 = _rt_of(self._op)
        th, tn, rho = f32(st.theta), f32(st.theta_n), f32(st.rho)
        gamma = f32(2) * th / (tn * tn) * (tn * tn - f32(2) * th * th + th)
        st.thetaold = float(th)
        if st.iteration == self.iterations - 1:
            thn = (f32(1) + np.sqrt(f32(1) + f32(8) * th * th)) / f32(2)
        else:
            thn = (f32(1) + np.sqrt(f32(1) + f32(4) * th * th)) / f32(2)
        st.theta = float(thn)
        alpha, beta = (th - f32(1)) / thn, th / thn
        return rho, gamma, alpha, beta

    def _update_args(self, st, fus, rho, gamma, alpha, beta):
        f32 = _rt_of(self._op)
        return (st.x.ctx.handle, st.x.code, st.x.n, st.res.ptr, st.x0.ptr, st.x.ptr, st.y.ptr, st.z.ptr, st.zold.ptr,
                float(rho * gamma), fus[0], float(rho * gamma * f32(self.reg.lam)), float(f32(-1) / gamma),
                float(f32(1) / gamma), float(-beta), float(f32(1) + alpha + beta), float(-alpha))

    def iterate(self, st=None):
        st = st or self.state
        if st.rel_res_norm < st.relTol or st.iteration >= self.iterations:
            return None
        f32 = _rt_of(self._op)
        rho, gamma, alpha, beta = self._coefficients(st)
        fus = None if self._op.double else _fusable_kinds(self.reg, [])
        if fus is not None:  # one launch for everything after the operator apply (rls_optista_update)
            _NormalApply(self._op).mul_(st.res, st.x)
            ctx = st.x.ctx
            out = (C.c_float * 1)()
            check(ctx.handle, ctx.lib.rls_optista_update(*self._update_args(st, fus, rho, gamma, alpha, beta), out),
                  "rls_optista_update")
            st.rel_res_norm = float(out[0]) / st.norm_x0
            if self.verbose:
                print(f"Iteration {st.iteration}; rel. residual = {st.rel_res_norm}")
            st.iteration += 1
            return st.x, st
        st.zold.copy_from(st.z)
        st.z.copy_from(st.y)
        _NormalApply(self._op).mul_(st.res, st.x)
        st.res.axpy_(-1.0, st.x0)
        st.y.axpy_(-float(rho * gamma), st.res)
        st.rel_res_norm = st.res.norm() / st.norm_x0
        if self.verbose:
            print(f"Iteration {st.iteration}; rel. residual = {st.rel_res_norm}")
        self.reg.prox_(st.y, float(rho * gamma * f32(self.reg.lam)))
        st.z.lincomb_(float(f32(-1) / gamma), st.z, 1.0, st.x)      # z ./= -gamma ; z .+= x ...
        st.z.axpy_(float(f32(1) / gamma), st.y)                     # ... .+ y ./ gamma
        st.x.lincomb_(float(-beta), st.x, float(f32(1) + alpha + beta), st.z)
        st.x.axpy_(float(-alpha), st.zold)
        st.iteration += 1
        return st.x, st

    def _run(self, st):
        """no callbacks: every remaining iteration is enqueued at once (the coefficients depend on the index only);
        `rel_res_norm < relTol` is evaluated on the device, later launches are no-ops, ONE read-back at the end"""
        fus = None if self._op.double else _fusable_kinds(self.reg, [])
        if fus is None or self.verbose or not isinstance(self._op, OperatorHandle) or st.rel_res_norm < st.relTol:
            while self.iterate(st) is not None:
                pass
            return
        ctx = st.x.ctx
        lib, h = ctx.lib, ctx.handle
        if _pgm_plan(self) is not None:
            # whole blocks of iterations as single launches, A in the register files (rls_pgm_step_resident)
            def row(st):
                a = self._update_args(st, fus, *self._coefficients(st))
                return (a[9], a[11]) + a[12:17], (st.theta, st.thetaold)
            _pgm_resident(self, st, 0, row, ("theta", "thetaold"), (st.x, st.y, st.z, st.zold), fus)
            if st.rel_res_norm < st.relTol or st.iteration >= self.iterations:
                return
        rec = _pgm_record(st, ctx)
        thetas = [(st.theta, st.thetaold)]
        for _ in range(st.iteration, self.iterations):
            rho, gamma, alpha, beta = self._coefficients(st)
            st.iteration += 1
            thetas.append((st.theta, st.thetaold))
            check(h, lib.rls_operator_mul_normal_skip(self._op.handle, st.x.ptr, st.res.ptr, rec.ptr + 4), "rls_operator_mul_normal_skip")
            check(h, lib.rls_optista_update_async(*self._update_args(st, fus, rho, gamma, alpha, beta), float(st.norm_x0),
                                                  float(st.relTol), rec.ptr), "rls_optista_update_async")
        done_its, res_norm = _pgm_fetch(rec)
        first = st.iteration - (len(thetas) - 1)
        st.iteration = first + done_its
        st.theta, st.thetaold = thetas[done_its]
        if done_its:
            st.rel_res_norm = res_norm / st.norm_x0


def _pgm_record(st, ctx):
    """4 zeroed device words {iteration, done, ||res||, pad} of the deferred OptISTA / POGM sequences"""
    if getattr(st, "_rec", None) is None or st._rec.ctx is not ctx:
        st._rec = DeviceVector(4, np.float32, ctx)
    st._rec.fill_(0)
    return st._rec


def _pgm_fetch(rec):
    raw = rec.to_host()  # synchronises
    return int(raw[:1].view(np.int32)[0]), float(raw[2])


_PGM_BLOCK = 48   # iterations per resident launch (RLS_PGM_MAX_IT; even, so POGM's buffer roles are the same at every block start)


def _pgm_plan(solver):
    """the resident-launch plan of an OptISTA / POGM solver over a dense operator (None: the operator does not fit, or a launch
    was lost earlier and the plan retired itself)"""
    op = solver._op
    if not isinstance(op, OperatorHandle):
        return None
    cached = getattr(solver, "_pgm", None)
    if cached is None or cached[0] is not op:
        ctx = op.ctx
        out = C.c_void_p()
        rc = ctx.lib.rls_pgm_create(op.handle, C.byref(out))
        if rc == -2:
            plan = None
        else:
            check(ctx.handle, rc, "rls_pgm_create")
            plan = _PgmPlan(ctx, out)
        solver._pgm = cached = (op, plan)
    plan = cached[1]
    return plan if plan is not None and not plan.off else None


class _PgmPlan:
    def __init__(self, ctx, handle):
        self.ctx, self.handle, self.off, self.fallbacks = ctx, handle, False, 0

    def __del__(self):
        try:
            if self.handle:
                self.ctx.lib.rls_pgm_destroy(self.handle)
        except Exception:
            pass
        self.handle = None


def _pgm_resident(solver, st, kind, row, names, vecs, fus):
    """Run the remaining iterations of `st` as resident launches of up to _PGM_BLOCK iterations.  `row(st)` advances the
    index-only scalars of `st` by one iteration and returns (the 7 / 6 coefficient floats of the per-iteration entry point,
    the values of `names` after it).  On return st.iteration, rel_res_norm, the `names` scalars and (POGM) the x / y
    references describe what the device actually did: everything requested, fewer because the stopping test fired, or fewer
    because a launch could not become resident (then the plan retires and the caller continues launch by launch)."""
    plan = _pgm_plan(solver)
    ctx = st.x.ctx
    lib = ctx.lib
    first = st.iteration
    start = tuple(getattr(st, n) for n in names)
    # the table depends on the iteration index and the scalars below only: a repeated solve reuses it
    key = (kind, first, solver.iterations, start, float(st.rho), float(solver.reg.lam), getattr(st, "sigma", None),
           getattr(st, "theta_n", None))
    cached = getattr(solver, "_pgm_table", None)
    if cached is not None and cached[0] == key:
        coefs, hist = cached[1], cached[2]
    else:
        hist, rows = [start], []
        for _ in range(first, solver.iterations):
            r, after = row(st)
            st.iteration += 1
            rows.append(tuple(r) + (0.0,) * (8 - len(r)))
            hist.append(after)
        coefs = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, 8)
        solver._pgm_table = (key, coefs, hist)
    rows = coefs
    rec = _pgm_record(st, ctx)
    v0, v1, v2, o0 = vecs
    done_launch = 0
    for off in range(0, len(rows), _PGM_BLOCK):
        n = min(_PGM_BLOCK, len(rows) - off)
        rc = lib.rls_pgm_step_resident(plan.handle, kind, n, off, coefs[off:].ctypes.data_as(C.POINTER(C.c_float)), v0.ptr, v1.ptr,
                                       v2.ptr, o0.ptr, st.res.ptr, st.x0.ptr, fus[0], fus[1] if kind == 1 else 0,
                                       float(st.norm_x0), float(st.relTol), rec.ptr)
        if rc == -2:
            break
        check(ctx.handle, rc, "rls_pgm_step_resident")
        done_launch += 1
    lost, total = C.c_int32(0), C.c_int32(0)
    if done_launch:
        check(ctx.handle, lib.rls_pgm_lost(plan.handle, C.byref(lost), C.byref(total)), "rls_pgm_lost")
    if lost.value:   # (RLS_E_UNSUPPORTED without a lost launch: resident mode is switched off on the context, nothing to retire)
        plan.off = True
        plan.fallbacks = total.value
    done_its, res_norm = _pgm_fetch(rec)
    st.iteration = first + done_its
    for n, v in zip(names, hist[done_its]):
        setattr(st, n, v)
    if done_its:
        st.rel_res_norm = res_norm / st.norm_x0
    return done_its


class POGM(AbstractProximalGradientSolver):
    """src/POGM.jl:75-110 (ctor), :133-160 (init!), :169-237 (iterate).  gamma starts at 1 and is not reset
    by init! (reference behaviour)."""

    def __init__(self, A=None, *, AHA=None, reg=None, normalizeReg=None, iterations: int = 50, verbose: bool = False,
                 rho=None, theta=1, sigma_fac=1, relTol=None, restart: str = "none"):
        self.A, self._op = _resolve_operator(A, AHA)
        self.AHA = AHA if AHA is not None else self.A.normal_operator()
        self.reg, self.proj = _split_regs(reg, "POGM")
        self.reg = normalize(normalizeReg, [self.reg], self.A, None)[0]
        self.normalizeReg = normalizeReg or NoNormalization()
        if restart not in ("none", "gradient"):
            raise ValueError("restart must be 'none' or 'gradient'")
        self.restart = restart
        self.verbose = bool(verbose)
        self.iterations = int(iterations)
        self.state = _ProxGradState(_default_rho(self._op) if rho is None else rho, theta, _eps_of(self._op, relTol),
                                    ("x", "x0", "xold", "y", "z", "w", "res"))
        self.state.gamma = 1.0
        self.state.sigma = 1.0
        self.state.sigma_fac = float(sigma_fac)

    def _new_state(self):
        s = self.state.states[0] if isinstance(self.state, AbstractMatrixSolverState) else self.state
        n = _ProxGradState(s.rho, 1.0, s.relTol, s._names)
        n.gamma, n.sigma, n.sigma_fac = s.gamma, 1.0, s.sigma_fac
        return n

    def init_(self, st, b: DeviceVector, x0=0, theta=1):
        st._alloc(b, self._op.N)
        if self.A is None:
            st.x0.copy_from(b)
        else:
            self.A.mul_adj_(st.x0, b)
        st.norm_x0 = st.x0.norm()
        self.reg = normalize(self.normalizeReg, self.reg, self.A, st.x0, in_solver=True)  # src/POGM.jl:163
        if np.isscalar(x0):
            st.x.fill_(x0)
        else:
            st.x.copy_from(x0 if isinstance(x0, DeviceVector) else DeviceVector.from_host(np.asarray(x0, dtype=b.dtype), b.ctx))
        for v in (st.xold, st.y, st.z, st.w):
            v.fill_(0)
        st.res.fill_(math.inf)
        st.theta = st.thetaold = float(theta)
        st.sigma = 1.0
        st.rel_res_norm = math.inf
        st.iteration = 0

    def iterate(self, st=None):
        st = st or self.state
        if st.rel_res_norm < st.relTol or st.iteration >= self.iterations:
            return None
        f32 = _rt_of(self._op)
        rho = f32(st.rho)
        fus = None if self._op.double else _fusable_kinds(self.reg, self.proj)
        if fus is None:
            st.xold.copy_from(st.x)
            _NormalApply(self._op).mul_(st.res, st.x)
            st.res.axpy_(-1.0, st.x0)
            st.x.axpy_(-float(rho), st.res)
            st.rel_res_norm = st.res.norm() / st.norm_x0
            if self.verbose:
                print(f"Iteration {st.iteration}; rel. residual = {st.rel_res_norm}")
        tho = f32(st.theta)
        st.thetaold = float(tho)
        if st.iteration == self.iterations - 1 and self.restart != "none":
            th = (f32(1) + np.sqrt(f32(1) + f32(8) * tho * tho)) / f32(2)
        else:
            th = (f32(1) + np.sqrt(f32(1) + f32(4) * tho * tho)) / f32(2)
        st.theta = float(th)
        alpha = (tho - f32(1)) / th
        beta = f32(st.sigma) * tho / th
        gamma_old = f32(st.gamma)
        if self.restart == "gradient":
            gamma = rho * (f32(1) + alpha + beta)
        else:
            gamma = rho * (f32(2) * tho + th - f32(1)) / th
        st.gamma = float(gamma)
        if fus is not None:  # one launch for everything after the operator apply (rls_pogm_update)
            _NormalApply(self._op).mul_(st.res, st.x)
            ctx = st.x.ctx
            out = (C.c_float * 4)()
            restart = self.restart == "gradient"
            check(ctx.handle, ctx.lib.rls_pogm_update(
                ctx.handle, st.x.code, st.x.n, st.res.ptr, st.x0.ptr, st.x.ptr, st.y.ptr, st.xold.ptr, st.z.ptr,
                st.w.ptr, float(rho), float(-alpha), float(f32(1) + alpha + beta), -float(beta + rho * alpha / gamma_old),
                float(rho * alpha / gamma_old), fus[0], float(gamma * f32(self.reg.lam)), fus[1], int(restart),
                float(rho / gamma), out), "rls_pogm_update")
            st.x, st.y = st.y, st.x  # swap x and y (the kernel wrote the new x into the old y buffer)
            st.rel_res_norm = float(out[0]) / st.norm_x0
            if self.verbose:
                print(f"Iteration {st.iteration}; rel. residual = {st.rel_res_norm}")
            if restart:
                crit = (f32(out[1]) - f32(out[2])) / gamma - f32(out[3])
                if crit < 0:
                    if self.verbose:
                        print(f"Gradient restart at iter {st.iteration}")
                    st.sigma = 1.0
                    st.theta = 1.0
                else:
                    st.sigma = float(f32(st.sigma) * f32(st.sigma_fac))
            st.iteration += 1
            return st.x, st
        st.x, st.y = st.y, st.x  # swap x and y
        st.x.lincomb_(float(-alpha), st.x, float(f32(1) + alpha + beta), st.y)
        st.x.axpy_(-float(beta + rho * alpha / gamma_old), st.xold)
        st.x.axpy_(float(rho * alpha / gamma_old), st.z)
        st.z.copy_from(st.x)
        self.reg.prox_(st.x, float(gamma * f32(self.reg.lam)))
        for pr in self.proj:
            pr.prox_(st.x)
        if self.restart == "gradient":
            st.w.axpy_(1.0, st.y)
            st.w.axpy_(float(rho / gamma), st.x)
            st.w.axpy_(-float(rho / gamma), st.z)
            crit = (complex(st.w.dot(st.x)) - complex(st.w.dot(st.z))) / complex(gamma) - complex(st.w.dot(st.res))
            if crit.real < 0:
                if self.verbose:
                    print(f"Gradient restart at iter {st.iteration}")
                st.sigma = 1.0
                st.theta = 1.0
            else:
                st.sigma = float(f32(st.sigma) * f32(st.sigma_fac))
            st.w.lincomb_(float(rho / gamma), st.z, -float(rho / gamma), st.x)
            st.w.axpy_(-1.0, st.y)
        st.iteration += 1
        return st.x, st

    def _run(self, st):
        """restart = :none without callbacks: all remaining iterations enqueued at once (index-only coefficients),
        the stopping test on the device, ONE read-back at the end; otherwise iteration by iteration"""
        fus = None if self._op.double else _fusable_kinds(self.reg, self.proj)
        if (fus is None or self.verbose or not isinstance(self._op, OperatorHandle) or st.rel_res_norm < st.relTol):
            while self.iterate(st) is not None:
                pass
            return
        f32 = _rt_of(self._op)
        ctx = st.x.ctx
        lib, h = ctx.lib, ctx.handle
        if self.restart == "gradient":
            # theta, sigma, gamma live in the device record; every launch derives its coefficients from them and
            # applies the restart rule itself (rls_pogm_update_auto), so nothing is read back until the end
            if getattr(st, "_rec8", None) is None or st._rec8.ctx is not ctx:
                st._rec8 = DeviceVector(8, np.float32, ctx)
            init = np.zeros(8, np.float32)
            init[4:8] = [f32(st.theta), f32(st.thetaold), f32(st.sigma), f32(st.gamma)]
            st._rec8.copy_from_host(init)
            rec, bufs, first = st._rec8, (st.x, st.y), st.iteration
            todo = self.iterations - first  # (the record counts from 0: the last iteration's theta rule, :185, compares with this)
            start = 0
            plan = _pgm_plan(self)
            if plan is not None and todo > 0:
                # whole blocks of iterations as single launches, A in the register files: the kernel forms the coefficients
                # from the record and applies the restart rule itself (rls_pogm_step_resident_restart)
                launched = 0
                for off in range(0, todo, _PGM_BLOCK):
                    n = min(_PGM_BLOCK, todo - off)
                    xb, yb = bufs if off % 2 == 0 else bufs[::-1]
                    rc = lib.rls_pogm_step_resident_restart(plan.handle, n, off, float(f32(st.rho)), float(f32(self.reg.lam)),
                                                            float(f32(st.sigma_fac)), todo, xb.ptr, yb.ptr, st.z.ptr, st.w.ptr,
                                                            st.xold.ptr, st.res.ptr, st.x0.ptr, fus[0], fus[1], float(st.norm_x0),
                                                            float(st.relTol), rec.ptr)
                    if rc == -2:
                        break
                    check(h, rc, "rls_pogm_step_resident_restart")
                    launched += 1
                if launched:
                    lost, total = C.c_int32(0), C.c_int32(0)
                    check(h, lib.rls_pgm_lost(plan.handle, C.byref(lost), C.byref(total)), "rls_pgm_lost")
                    if lost.value:
                        plan.off = True
                        plan.fallbacks = total.value
                    raw = rec.to_host()
                    start = int(raw[:1].view(np.int32)[0])
                    if int(raw[1:2].view(np.int32)[0]):  # the stopping test fired inside a launch
                        todo = start
            for k in range(start, todo):  # what no resident launch did (no plan, or one was lost): launch by launch
                xb, yb = bufs if k % 2 == 0 else bufs[::-1]
                check(h, lib.rls_operator_mul_normal_skip(self._op.handle, xb.ptr, st.res.ptr, rec.ptr + 4), "rls_operator_mul_normal_skip")
                check(h, lib.rls_pogm_update_auto(h, xb.code, xb.n, st.res.ptr, st.x0.ptr, xb.ptr, yb.ptr, st.xold.ptr,
                                                  st.z.ptr, st.w.ptr, float(f32(st.rho)), float(f32(self.reg.lam)),
                                                  float(f32(st.sigma_fac)), self.iterations - first, fus[0], fus[1],
                                                  float(st.norm_x0), float(st.relTol), rec.ptr), "rls_pogm_update_auto")
            raw = rec.to_host()  # synchronises
            done_its = int(raw[:1].view(np.int32)[0])
            st.iteration = first + done_its
            st.x, st.y = bufs if done_its % 2 == 0 else bufs[::-1]
            if done_its:
                st.rel_res_norm = float(raw[2]) / st.norm_x0
                st.theta, st.thetaold, st.sigma, st.gamma = (float(v) for v in raw[4:8])
            return
        rho = f32(st.rho)

        def coefficients(st):
            """(rho, c_y, c_x1, c_xo, c_z, thr) of one iteration, advancing theta / gamma   (src/POGM.jl:183-201, restart == :none)"""
            tho = f32(st.theta)
            st.thetaold = float(tho)
            th = (f32(1) + np.sqrt(f32(1) + f32(4) * tho * tho)) / f32(2)
            st.theta = float(th)
            alpha = (tho - f32(1)) / th
            beta = f32(st.sigma) * tho / th
            gamma_old = f32(st.gamma)
            gamma = rho * (f32(2) * tho + th - f32(1)) / th
            st.gamma = float(gamma)
            return (float(rho), float(-alpha), float(f32(1) + alpha + beta), -float(beta + rho * alpha / gamma_old),
                    float(rho * alpha / gamma_old), float(gamma * f32(self.reg.lam)))

        if _pgm_plan(self) is not None:
            # whole blocks of iterations as single launches, A in the register files (rls_pgm_step_resident)
            def row(st):
                c = coefficients(st)
                return (c[0], c[5]) + c[1:5], (st.theta, st.thetaold, st.gamma)
            bufs = (st.x, st.y)
            done_its = _pgm_resident(self, st, 1, row, ("theta", "thetaold", "gamma"), (st.x, st.y, st.z, st.xold), fus)
            st.x, st.y = bufs if done_its % 2 == 0 else bufs[::-1]
            if st.rel_res_norm < st.relTol or st.iteration >= self.iterations:
                return
        rec = _pgm_record(st, ctx)
        bufs = (st.x, st.y)
        hist = [(st.theta, st.thetaold, st.gamma)]
        first = st.iteration
        for k in range(first, self.iterations):
            c = coefficients(st)
            hist.append((st.theta, st.thetaold, st.gamma))
            xb, yb = bufs if (k - first) % 2 == 0 else bufs[::-1]
            check(h, lib.rls_operator_mul_normal_skip(self._op.handle, xb.ptr, st.res.ptr, rec.ptr + 4), "rls_operator_mul_normal_skip")
            check(h, lib.rls_pogm_update_async(
                h, xb.code, xb.n, st.res.ptr, st.x0.ptr, xb.ptr, yb.ptr, st.xold.ptr, st.z.ptr, c[0], c[1], c[2], c[3], c[4],
                fus[0], c[5], fus[1], float(st.norm_x0), float(st.relTol), rec.ptr), "rls_pogm_update_async")
        done_its, res_norm = _pgm_fetch(rec)
        st.iteration = first + done_its
        st.theta, st.thetaold, st.gamma = hist[done_its]
        st.x, st.y = bufs if done_its % 2 == 0 else bufs[::-1]  # the kernel writes the new x into the old y buffer
        if done_its:
            st.rel_res_norm = res_norm / st.norm_x0


class SplitBregman(ADMM):  # AbstractPrimalDualSolver through ADMM
    """src/SplitBregman.jl:82-140 (ctor), :166-200 (init!), :204-271 (iterate), :273-282 (converged / done).
    Same composite operator and cg! as ADMM; prox threshold lambda / rho; the right-hand side gets its
    Bregman update every `iterationsInner` inner iterations."""

    def __init__(self, A=None, *, AHA=None, precon=None, reg=None, regTrafo=None, normalizeReg=None, rho=1e-1,
                 iterations: int = 10, iterationsInner: int = 10, iterationsCG: int = 10, absTol=None, relTol=None,
                 tolInner=1e-5, verbose: bool = False):
        super().__init__(A, AHA=AHA, precon=precon, reg=reg, regTrafo=regTrafo, normalizeReg=normalizeReg, rho=rho,
                         iterations=iterations, iterationsCG=iterationsCG, absTol=absTol, relTol=relTol,
                         tolInner=tolInner, verbose=verbose)
        self.iterationsInner = int(iterationsInner)

    def init_(self, state, b: DeviceVector, x0=0):
        super().init_(state, b, x0=x0)
        if getattr(state, "ybreg", None) is None or state.ybreg.n != state.x.n or state.ybreg.ctx is not b.ctx:
            state.ybreg = b.similar(self._op.N)
        state.ybreg.copy_from(state.beta_y)
        state.iter_cnt = 1
        state.iteration = 1

    def done(self, state):
        return self.converged(state) or (state.iteration == 1 and state.iter_cnt > self.iterations)

    def _prox_threshold(self, lam32, rho32):
        return lam32 / rho32  # prox!(reg, z, lambda / rho)   src/SplitBregman.jl:235

    def _plan_iterations(self):
        return self.iterationsInner  # one plan run = one block of inner iterations (:204-262)

    def _bregman_update(self, state):
        """src/SplitBregman.jl:264-268: beta_y += ybreg - AHA x ; z = Phi x ; u = 0 ; next block"""
        state.beta_y.axpy_(1.0, state.ybreg)
        self._op.mul_normal_(state.xold, state.x)
        state.beta_y.axpy_(-1.0, state.xold)
        for i, t in enumerate(self.regTrafo):
            t.mul_(state.z[i], state.x)
            state.u[i].fill_(0)
        state.iter_cnt += 1
        state.iteration = 0

    def _plan_block(self, state, n_inner):
        """up to n_inner inner iterations of the current block on the device (`converged` or the block length stop
        it there), one read-back, then the host-side Bregman update if the block ended"""
        lib, h = state.x.ctx.lib, state.x.ctx.handle
        if state.iteration == 1:  # a block starts: re-arm the plan with the current z as its first buffer
            P = self._plan_params(state)
            cur = state.z[0]
            other = state._zbufs[1] if cur is state._zbufs[0] else state._zbufs[0]
            state._zbufs = (cur, other)
            P.z0, P.z1 = cur.ptr, other.ptr
            check(h, lib.rls_admm_init(state._admm, C.byref(P)), "rls_admm_init")
            state._block_done = 0
        check(h, lib.rls_admm_step(state._admm, int(n_inner)), "rls_admm_step")
        st = AdmmStatus()
        cap = max(self.iterationsInner, 1)
        log = (C.c_float * (8 * cap))()
        check(h, lib.rls_admm_get_status(state._admm, C.byref(st), log, cap), "rls_admm_get_status")
        state.fallbacks = int(st.fallbacks)
        it = int(st.iteration)  # inner iterations completed in this block
        state.cg_iterations += [int(log[8 * k + 5]) for k in range(state._block_done, it)]
        state._block_done = it
        state.sk[0], state.eps_pri[0], state.rk[0], state.eps_dua[0] = st.sk, st.eps_pri, st.rk, st.eps_dua
        state.z[0], state.zold[0] = (state._zbufs[1], state._zbufs[0]) if it & 1 else state._zbufs
        state.iteration = it  # == the reference's counter before its end-of-iteration increment
        if self.converged(state) or state.iteration >= self.iterationsInner:
            self._bregman_update(state)
        state.iteration += 1

    def iterate(self, state=None):
        state = state or self.state
        if self.done(state):
            return None
        if state._plan_ok:
            self._plan_block(state, 1)
            return state.x, state
        f32 = _rt_of(self._op)
        lib, h = state.x.ctx.lib, state.x.ctx.handle
        fused = self._all_identity() and len(self.reg) == 1 and not self._op.double
        if fused:  # beta = beta_y + rho (z - u) in one launch (the same elementwise step as ADMM's)
            check(h, lib.rls_admm_pre(h, state.x.code, state.x.n, state.beta.ptr, state.beta_y.ptr, state.z[0].ptr,
                                      state.u[0].ptr, state.x.ptr, state.xold.ptr, float(state.rho[0]), 0), "rls_admm_pre")
        else:
            state.beta.copy_from(state.beta_y)
            for i, t in enumerate(self.regTrafo):
                t.mul_adj_(state.beta, state.z[i], float(state.rho[i]), 1.0)
                t.mul_adj_(state.beta, state.u[i], -float(state.rho[i]), 1.0)
        if self.precon is not None:   # cg!(...; Pl = precon)   src/SplitBregman.jl:218
            self._cg_precond(state)
        elif self._all_identity() and not self._op.double:
            check(h, lib.rls_cg_solve(state._cg, state.x.ptr, state.beta.ptr, float(np.sum(state.rho, dtype=np.float32)),
                                      self.iterationsCG, float(state.tolInner)), "rls_cg_solve")
            if _cg_is_resident(lib, state._cg):
                check(h, lib.rls_cg_get_status(state._cg, C.byref(CgStatus())), "rls_cg_get_status")
        else:
            self._cg_generic(state)
        for pr in self.proj:
            pr.prox_(state.x)
        if fused:
            # z = prox(x + u, lambda / rho); u += x - z and the norms of :243-262 in one launch, one read-back
            # (identity Phi: s = rho ||z - zold||, eps_dua = rho ||u||)
            state.z[0], state.zold[0] = state.zold[0], state.z[0]
            state.z[0].lincomb_(1.0, state.x, 1.0, state.u[0])
            if state.rho[0] != 0:
                self.reg[0].prox_(state.z[0], float(f32(self.reg[0].lam) / state.rho[0]))
            out = (C.c_float * 6)()
            check(h, lib.rls_admm_post(h, state.x.code, state.x.n, state.x.ptr, state.xold.ptr, state.z[0].ptr,
                                       state.zold[0].ptr, state.u[0].ptr, out), "rls_admm_post")
            state.sk[0] = state.rho[0] * f32(out[1])
            state.eps_pri[0] = f32(out[2])
            state.rk[0] = f32(out[3])
            state.eps_dua[0] = state.rho[0] * f32(out[4])
        for i, t in enumerate([] if fused else self.regTrafo):
            state.z[i], state.zold[i] = state.zold[i], state.z[i]
            t.mul_(state.z[i], state.x)
            state.z[i].axpy_(1.0, state.u[i])
            if state.rho[i] != 0:
                self.reg[i].prox_(state.z[i], float(f32(self.reg[i].lam) / state.rho[i]))
            t.mul_(state.u[i], state.x, 1.0, 1.0)
            state.u[i].axpy_(-1.0, state.z[i])
            tx = state.uold[i]  # scratch: Phi x
            t.mul_(tx, state.x)
            state.eps_pri[i] = max(f32(tx.norm()), f32(state.z[i].norm()))
            tx.axpy_(-1.0, state.z[i])
            state.rk[i] = f32(tx.norm())
            tx.lincomb_(1.0, state.z[i], -1.0, state.zold[i])
            t.mul_adj_(state.xold, tx, float(state.rho[i]), 0.0)
            state.sk[i] = f32(state.xold.norm())
            t.mul_adj_(state.xold, state.u[i], float(state.rho[i]), 0.0)
            state.eps_dua[i] = f32(state.xold.norm())
        if self.converged(state) or state.iteration >= self.iterationsInner:
            self._bregman_update(state)
        state.iteration += 1
        return state.x, state

    def _run(self, state):
        while state._plan_ok and not self.done(state):
            self._plan_block(state, self.iterationsInner - (state.iteration - 1))
        while self.iterate(state) is not None:
            pass


# --------------------------------------------------------------------------------------------
# Kaczmarz row-action solver: src/Kaczmarz.jl
# --------------------------------------------------------------------------------------------


def _i32_device(a, ctx) -> DeviceVector:
    """int32 host array -> device buffer (carried in a float32 DeviceVector, bit-preserving)"""
    return DeviceVector.from_host(np.ascontiguousarray(a, dtype=np.int32).view(np.float32), ctx)


class KaczmarzState(AbstractSolverState):
    """src/Kaczmarz.jl:23-32.  nrhs > 1: the columns of a matrix right-hand side advance in ONE launch, one
    workgroup per column (backend scheduler, same per-column results as MultiThreadingState)."""

    def __init__(self):
        self.u = self.x = self.vl = None
        self.eps_w = 0.0
        self.iteration = 0
        self.usedIndices = None
        self.nrhs = 1
        self.matrix = False  # x, u, vl are DeviceMatrix (one column per right-hand side)
        self._rows = self._den = None
        self.A = None

    def solutions(self) -> List[DeviceVector]:
        return [self.x.column(j) for j in range(self.nrhs)]

    def _views(self, M: DeviceMatrix) -> List[DeviceVector]:
        return [M.column_view(j) for j in range(self.nrhs)]

    def convergence(self):
        """(; residual = norm(A * x - u))  src/Kaczmarz.jl:268"""
        xs = self._views(self.x) if self.matrix else [self.x]
        us = self._views(self.u) if self.matrix else [self.u]
        out = []
        for x, u in zip(xs, us):
            t = u.similar(u.n)
            self.A.mul_transpose_(t, x)  # A x through the stored transpose(A)
            t.axpy_(-1.0, u)
            out.append({"residual": t.norm()})
        return out if self.matrix else out[0]


class Kaczmarz(AbstractRowActionSolver):
    """Kaczmarz(A; reg = L2Regularization(0), normalizeReg = NoNormalization(), randomized = false,
    subMatrixFraction = 0.15, shuffleRows = false, seed = 1234, iterations = 10)   src/Kaczmarz.jl:76-159.

    The whole row sweep of one iteration (:283-299) is one kernel launch (rls_kaczmarz_sweep): x stays in
    the registers of one workgroup, the rows of A stream through from a transposed copy of A (the
    row-access layout of :391).  `shuffleRows` / `randomized` draw the row order from a NumPy generator
    seeded with `seed` (the reference seeds Julia's global RNG, so the orders differ, not the method);
    `greedy_randomized` is CPU-only in the reference (test/testKaczmarz.jl:114) and raises here."""

    def __init__(self, A=None, *, reg=None, normalizeReg=None, randomized: bool = False, subMatrixFraction=0.15,
                 shuffleRows: bool = False, seed: int = 1234, iterations: int = 10, greedy_randomized: bool = False,
                 theta=None):
        if not isinstance(A, DeviceMatrix):
            raise TypeError("A must be a DeviceMatrix (the backend is selected by the array type, as in the reference)")
        if greedy_randomized:
            raise NotImplementedError("greedy randomized Kaczmarz is not defined for GPU arrays (test/testKaczmarz.jl:114)")
        self.A_in = A
        self.normalizeReg = normalizeReg or NoNormalization()
        regs = normalize(normalizeReg, _as_list(reg) or [L2Regularization(0.0)], A, None)
        i2 = findsink(L2Regularization, regs)  # src/Kaczmarz.jl:86
        self.L2 = regs[i2] if i2 is not None else L2Regularization(0.0)
        proj = [r for r in regs if is_projection(r)]
        rest = [r for r in regs if r is not self.L2 and r not in proj]
        if len(rest) > 1:
            raise ValueError(f"Kaczmarz does not allow for more than one additional regularization term, found {len(rest)}")
        self.reg = proj + rest
        lam = self.L2.lam_vector if getattr(self.L2, "lam_vector", None) is not None else self.L2.lam
        if np.ndim(lam) == 1 and not isinstance(self.normalizeReg, (NoNormalization, SystemMatrixBasedNormalization)):
            raise ValueError("Tikhonov matrix for Kaczmarz is only valid with no or system matrix based normalization")
        self.randomized, self.shuffleRows, self.seed = bool(randomized), bool(shuffleRows), int(seed)
        self.iterations = int(iterations)
        self.subMatrixSize = int(round(subMatrixFraction * A.M))
        self._setup_rows(lam)
        self.state = KaczmarzState()

    # initkaczmarz (:372-398): transposed operator, denominators, row index
    def _setup_rows(self, lam):
        A, ctx = self.A_in, self.A_in.ctx
        lib, h = ctx.lib, ctx.handle
        dbl = is_double(A.code)   # Float64 / ComplexF64: the same sweep on the double-precision entry points (rls_*_d)
        rt = np.float64 if dbl else np.float32
        self._rt = rt
        transpose = lib.rls_transpose_d if dbl else lib.rls_transpose
        scale_rows = lib.rls_scale_rows_d if dbl else lib.rls_scale_rows
        At = DeviceMatrix(A.N, A.M, A.dtype, ctx)
        check(h, transpose(h, A.code, A.M, A.N, A.ptr, A.lda, At.ptr, At.lda), "rls_transpose")
        self._lam_vec = None
        if np.ndim(lam) == 1:
            # ||Ax - b||² + ||L x||², L = diag(sqrt(lambda)):  A <- A inv(L), lambda <- 1   (:385-395)
            self._lam_vec = np.asarray(lam, dtype=rt)
            w = DeviceVector.from_host((rt(1) / np.sqrt(self._lam_vec)).astype(A.dtype), ctx)
            check(h, scale_rows(h, A.code, At.M, At.N, w.ptr, At.ptr, At.lda, At.ptr, At.lda), "rls_scale_rows")
            Arow = DeviceMatrix(A.M, A.N, A.dtype, ctx)
            check(h, transpose(h, A.code, At.M, At.N, At.ptr, At.lda, Arow.ptr, Arow.lda), "rls_transpose")
            lam = 1.0
        else:
            Arow = A
        self.At = At
        self._lam_used = float(lam)
        s2 = Arow.rownorm2().to_host()
        self._s2 = s2
        self.rowindex = np.nonzero(s2 > 0)[0].astype(np.int64)
        self.denom = (rt(1) / (s2[self.rowindex] + rt(lam))).astype(rt)
        self.rowIndexCycle = np.arange(len(self.rowindex))
        self.probabilities = (s2[self.rowindex] / s2.sum()).astype(np.float64) if self.randomized else None

    def _new_state(self):
        return KaczmarzState()

    def _upload_order(self, st, order):
        ctx = self.A_in.ctx
        st.usedIndices = np.asarray(order, dtype=np.int64)
        st._rows = _i32_device(self.rowindex[st.usedIndices], ctx)
        st._den = DeviceVector.from_host(self.denom[st.usedIndices], ctx)

    def init_(self, st: KaczmarzState, b, x0=0):
        """init!(solver, state, b; x0 = 0)   src/Kaczmarz.jl:178-217"""
        A = self.A_in
        lam_prev = self._lam_used
        if self._lam_vec is None:
            self.L2 = normalize(self.normalizeReg, self.L2, A, b if isinstance(b, DeviceVector) else None, in_solver=True)
            self.reg = normalize(self.normalizeReg, self.reg, A, b if isinstance(b, DeviceVector) else None, in_solver=True)
            if float(self.L2.lam) != lam_prev:  # lambda changed => recompute the denominators (:186-193)
                self._lam_used = float(self.L2.lam)
                self.denom = (self._rt(1) / (self._s2[self.rowindex] + self._rt(self._lam_used))).astype(self._rt)
        self._rng = np.random.default_rng(self.seed) if (self.shuffleRows or self.randomized) else None
        order = self.rowIndexCycle
        if self.shuffleRows and not self.randomized:
            order = self._rng.permutation(len(self.rowindex))
        st.matrix = isinstance(b, DeviceMatrix)
        nrhs = b.N if st.matrix else 1
        st.nrhs = nrhs
        st.A = self.At
        if not st.matrix:
            st.x, st.vl = b.similar(A.N), b.similar(A.M)
            st.u = b.copy()
        else:
            st.x = DeviceMatrix(A.N, nrhs, b.dtype, b.ctx)
            st.vl = DeviceMatrix(A.M, nrhs, b.dtype, b.ctx)
            st.u = DeviceMatrix(A.M, nrhs, b.dtype, b.ctx)
            if b.lda == b.M and st.u.lda == st.u.M:  # contiguous: one copy for all columns
                whole = lambda Mx: DeviceVector(Mx.M * Mx.N, Mx.dtype, Mx.ctx, _buf=Mx._buf, _offset=Mx.ptr - Mx._buf.ptr)
                whole(st.u).copy_from(whole(b))
            else:
                for j, uj in enumerate(st._views(st.u)):
                    uj.copy_from(b.column_view(j))
        if st.matrix and np.isscalar(x0) and st.x.lda == st.x.M and st.vl.lda == st.vl.M:
            for Mx, val in ((st.x, x0), (st.vl, 0)):
                DeviceVector(Mx.M * Mx.N, Mx.dtype, Mx.ctx, _buf=Mx._buf, _offset=Mx.ptr - Mx._buf.ptr).fill_(val)
        else:
            for col in (st._views(st.x) if st.matrix else [st.x]):
                if np.isscalar(x0):
                    col.fill_(x0)
                else:
                    col.copy_from(x0 if isinstance(x0, DeviceVector) else DeviceVector.from_host(np.asarray(x0, dtype=b.dtype), b.ctx))
            for col in (st._views(st.vl) if st.matrix else [st.vl]):
                col.fill_(0)
        if not self.randomized:
            self._upload_order(st, order)
        st.eps_w = 1.0 if self._lam_vec is not None else float(np.sqrt(self._rt(self._lam_used)))
        st.iteration = 0

    def _sweep(self, st, n_sweeps):
        A, ctx = self.A_in, self.A_in.ctx
        ldx = st.x.lda if st.matrix else A.N
        ldu = st.u.lda if st.matrix else A.M
        ldvl = st.vl.lda if st.matrix else A.M
        sweep = ctx.lib.rls_kaczmarz_sweep_d if is_double(A.code) else ctx.lib.rls_kaczmarz_sweep
        check(ctx.handle, sweep(ctx.handle, A.code, A.M, A.N, self.At.ptr, self.At.lda, st.nrhs, st.x.ptr, ldx, st.u.ptr, ldu,
                                st.vl.ptr, ldvl, st._rows.ptr, st._den.ptr, len(st.usedIndices), float(st.eps_w), int(n_sweeps)),
              "rls_kaczmarz_sweep")

    def iterate(self, st: Optional[KaczmarzState] = None):
        st = st or self.state
        if st.iteration >= self.iterations:  # done (:320)
            return None
        if self.randomized:  # sample!(rowIndexCycle, weights(probabilities), usedIndices, replace = false)  :286-288
            p = self.probabilities / self.probabilities.sum()
            self._upload_order(st, self._rng.choice(len(self.rowindex), size=self.subMatrixSize, replace=False, p=p))
        self._sweep(st, 1)
        for r in self.reg:
            for col in (st._views(st.x) if st.matrix else [st.x]):
                r.prox_(col) if is_projection(r) else r.prox_(col, r.lam)
        st.iteration += 1
        return st.x, st

    def _run(self, st):
        if not self.reg and not self.randomized and st.iteration < self.iterations:
            self._sweep(st, self.iterations - st.iteration)  # every remaining sweep in one launch
            st.iteration = self.iterations
            return
        while self.iterate(st) is not None:
            pass

    def _solution(self, st):
        """solversolution(solver::Kaczmarz)  :262-265 (Tikhonov matrix: x .* 1 ./ sqrt.(lambda))"""
        if self._lam_vec is None:
            return st.solutions() if st.matrix else st.x
        w = (self._rt(1) / np.sqrt(self._lam_vec)).astype(st.x.dtype)
        cols = st.solutions() if st.matrix else [st.x]
        out = [DeviceVector.from_host(c.to_host() * w, c.ctx) for c in cols]
        return out if st.matrix else out[0]


# --------------------------------------------------------------------------------------------
# matrix right-hand sides: src/MultiThreading.jl
# --------------------------------------------------------------------------------------------


class AbstractMatrixSolverState(AbstractSolverState):
    def __init__(self, states):
        self.states = list(states)
        self.active = [True] * len(self.states)

    def convergence(self):
        return [s.convergence() for s in self.states]


class SequentialState(AbstractMatrixSolverState):
    """src/MultiThreading.jl:8-12"""


class MultiThreadingState(AbstractMatrixSolverState):
    """src/MultiThreading.jl:19-23.  On one GPU the columns are independent streams of kernels on the
    context's queue; across GPUs, columns are sharded one set per device (multigpu.MultiSolve)."""


class BatchedState(AbstractMatrixSolverState):
    """Backend-specific scheduler for matrix right-hand sides: the K columns advance TOGETHER and share
    one pass over A per iteration (rls_cgnr_*_batched).  Same semantics as MultiThreadingState --
    independent per-column scalars and per-column retirement -- so results are those of column-by-column
    solves.  Solvers / shapes the fused batched plan does not cover fall back to MultiThreadingState."""

    def __init__(self, solver, B: DeviceMatrix):
        self.states = []
        self.active = [True] * B.N
        self.solver = solver
        self.K = B.N
        op = solver._op
        ctx = B.ctx
        N = op.N
        self.X, self.R, self.P, self.V = (DeviceMatrix(N, B.N, B.dtype, ctx) for _ in range(4))
        lib, h = ctx.lib, ctx.handle
        plan = C.c_void_p()
        check(h, lib.rls_cgnr_create_batched(op.handle, B.N, self.X.ptr, self.R.ptr, self.P.ptr, self.V.ptr, N,
                                             C.byref(plan)), "rls_cgnr_create_batched")
        self._plan = plan
        self._keep = (op, ctx)
        self.iteration = 0

    def _step(self, n):
        ctx = self.X.ctx
        check(ctx.handle, ctx.lib.rls_cgnr_step(self._plan, int(n)), "rls_cgnr_step")

    def status(self):
        st = (CgnrStatus * self.K)()
        ctx = self.X.ctx
        check(ctx.handle, ctx.lib.rls_cgnr_get_status_batched(self._plan, st), "rls_cgnr_get_status_batched")
        return list(st)

    def convergence(self):
        return [{"residual": s.residual} for s in self.status()]

    def solutions(self) -> List[DeviceVector]:
        return [self.X.column(j) for j in range(self.K)]

    def __del__(self):
        try:
            if self._plan and self.X.ctx.handle:
                self.X.ctx.lib.rls_cgnr_destroy(self._plan)
        except Exception:
            pass
        self._plan = None


class FistaBatchedState(BatchedState):
    """BatchedState for FISTA: the K extrapolated points share one pass over A per product (rls_fista_*_batched);
    prox, momentum and `done` are per column, exactly as K independent solves."""

    def __init__(self, solver, B: DeviceMatrix):
        self.states = []
        self.active = [True] * B.N
        self.solver = solver
        self.K = B.N
        op, ctx, N = solver._op, B.ctx, solver._op.N
        self.X, self.Xold, self.X0, self.RES = (DeviceMatrix(N, B.N, B.dtype, ctx) for _ in range(4))
        lib, h = ctx.lib, ctx.handle
        plan = C.c_void_p()
        check(h, lib.rls_fista_create_batched(op.handle, B.N, self.X.ptr, self.X0.ptr, self.Xold.ptr, self.RES.ptr, N,
                                              C.byref(plan)), "rls_fista_create_batched")
        self._plan = plan
        self._keep = (op, ctx)
        self.iteration = 0
        self._destroy = lib.rls_fista_destroy

    def _step(self, n):
        ctx = self.X.ctx
        check(ctx.handle, ctx.lib.rls_fista_step(self._plan, int(n)), "rls_fista_step")

    def status(self):
        st = (FistaStatus * self.K)()
        ctx = self.X.ctx
        check(ctx.handle, ctx.lib.rls_fista_get_status_batched(self._plan, st), "rls_fista_get_status_batched")
        return list(st)

    def convergence(self):
        return [{"residual": s.residual} for s in self.status()]

    def solutions(self) -> List[DeviceVector]:
        # state.x of column j is X when its iteration count is even, Xold when odd (src/FISTA.jl:144-146)
        return [(self.Xold if s.iteration & 1 else self.X).column(j) for j, s in enumerate(self.status())]

    def __del__(self):
        try:
            if self._plan and self.X.ctx.handle:
                self.X.ctx.lib.rls_fista_destroy(self._plan)
        except Exception:
            pass
        self._plan = None


class AdmmBatchedState(BatchedState):
    """BatchedState for ADMM (one regulariser, identity regTrafo, vary_rho = :none): the K columns' cg! iterations share
    one pass over A per product (rls_cg_create_batched + rls_admm_step on N x K matrices); prox, z / u updates, the
    residual norms and `done` are per column, exactly as K independent solves (src/MultiThreading.jl:30-79)."""

    def __init__(self, solver, B: DeviceMatrix):
        self.states = []
        self.active = [True] * B.N
        self.solver = solver
        self.K = B.N
        op, ctx, N = solver._op, B.ctx, solver._op.N
        mk = lambda: DeviceMatrix(N, B.N, B.dtype, ctx)
        self.x, self.xold, self.beta, self.beta_y, self.u0 = mk(), mk(), mk(), mk(), mk()
        self._zbufs = (mk(), mk())
        self.cg_u, self.cg_r, self.cg_c = mk(), mk(), mk()
        self.u = [self.u0]
        ref = solver.state.states[0] if isinstance(solver.state, AbstractMatrixSolverState) and solver.state.states else solver.state
        self.absTol, self.relTol, self.tolInner = ref.absTol, ref.relTol, ref.tolInner
        self.rho = np.full(1, solver.rho, np.float32)
        self.sigma_abs = np.float32(np.sqrt(np.float32(B.M))) * self.absTol
        lib, h = ctx.lib, ctx.handle
        cg = C.c_void_p()
        check(h, lib.rls_cg_create_batched(op.handle, B.N, self.cg_u.ptr, self.cg_r.ptr, self.cg_c.ptr, N, C.byref(cg)),
              "rls_cg_create_batched")
        self._cg = cg
        plan = C.c_void_p()
        check(h, lib.rls_admm_create(cg, C.byref(plan)), "rls_admm_create")
        self._plan = plan
        self._keep = (op, ctx)
        self.iteration = 0

    def init(self, B: DeviceMatrix):
        solver, ctx = self.solver, self.x.ctx
        for M_ in (self.x, self.xold, self.u0, self._zbufs[0], self._zbufs[1]):
            M_.fill_(0)  # x0 = 0: z = Phi x = 0, u = 0   (src/ADMM.jl:192-205)
        for j in range(self.K):
            solver.A.mul_adj_(self.beta_y.column_view(j), B.column_view(j))  # beta_y = A' b   (:198)
        P = solver._plan_params(self)
        if P is None:
            raise _lib.RLSError("batched ADMM: this configuration does not run as a device plan")
        check(ctx.handle, ctx.lib.rls_admm_init(self._plan, C.byref(P)), "rls_admm_init")
        self.iteration = 0

    def _step(self, n):
        ctx = self.x.ctx
        check(ctx.handle, ctx.lib.rls_admm_step(self._plan, int(n)), "rls_admm_step")

    def status(self):
        st = (AdmmStatus * self.K)()
        ctx = self.x.ctx
        check(ctx.handle, ctx.lib.rls_admm_get_status_batched(self._plan, st, None, 0), "rls_admm_get_status_batched")
        return list(st)

    def cg_iterations(self):
        """per column: the inner cg! iteration counts of its outer iterations"""
        cap = max(self.solver.iterations, 1)
        st = (AdmmStatus * self.K)()
        log = (C.c_float * (8 * cap * self.K))()
        ctx = self.x.ctx
        check(ctx.handle, ctx.lib.rls_admm_get_status_batched(self._plan, st, log, cap), "rls_admm_get_status_batched")
        return [[int(log[(j * cap + k) * 8 + 5]) for k in range(st[j].iteration)] for j in range(self.K)]

    def convergence(self):
        return [{"primal": s.rk, "dual": s.sk} for s in self.status()]

    def solutions(self) -> List[DeviceVector]:
        return [self.x.column(j) for j in range(self.K)]

    def __del__(self):
        try:
            if self._plan and self.x.ctx.handle:
                self.x.ctx.lib.rls_admm_destroy(self._plan)
                self.x.ctx.lib.rls_cg_destroy(self._cg)
        except Exception:
            pass
        self._plan = None


def _columns(b) -> List[DeviceVector]:
    if isinstance(b, DeviceMatrix):
        return [b.column(j) for j in range(b.N)]
    return list(b)


# --------------------------------------------------------------------------------------------
# driver API
# --------------------------------------------------------------------------------------------


def _check_eltype(solver, b):
    """b and the operator share the element type: with two precisions on the device a Float64 b under a Float32 plan would be
    misread, not converted (the reference converts nothing either: its state vectors are `similar(b)` and `mul!` would throw)"""
    op = getattr(solver, "_op", None) or getattr(solver, "A_in", None)
    want = getattr(op, "dtype", None)
    if want is not None and isinstance(b, (DeviceVector, DeviceMatrix)) and np.dtype(b.dtype) != np.dtype(want):
        raise TypeError(f"element types differ: the operator is {np.dtype(want)}, b is {np.dtype(b.dtype)}")


def init_(solver: AbstractLinearSolver, b, scheduler=SequentialState, **kw):
    """init!(solver, b; kwargs...)   src/RegularizedLeastSquares.jl:190, src/MultiThreading.jl:30-43"""
    _check_eltype(solver, b)
    if isinstance(b, DeviceVector):
        if isinstance(solver.state, AdmmBatchedState):
            ref = solver.state
            solver.state = ADMMState(len(solver.reg), solver.rho, ref.absTol, ref.relTol, ref.tolInner)
        if isinstance(solver.state, FistaBatchedState):
            solver.state = FISTAState(solver.state.rho, 1, solver.state.relTol)
        elif isinstance(solver.state, BatchedState):
            solver.state = CGNRState(solver.state.relTol)
        elif isinstance(solver.state, AbstractMatrixSolverState):
            solver.state = solver.state.states[0]  # :39-43
        solver.init_(solver.state, b, **kw)
        return
    if scheduler is BatchedState and isinstance(solver, Kaczmarz) and isinstance(b, DeviceMatrix):
        solver.state = KaczmarzState()
        solver.init_(solver.state, b, **kw)  # all columns in one launch, one workgroup per column
        return
    if scheduler is BatchedState:
        # the shared-A plans cover only the keyword arguments listed here; anything else (a warm start x0, ...) goes to
        # the per-column path below, which forwards **kw to the solver's own init_ (and raises what it does not support)
        kw_cgnr_ok = all(k == "x0" and np.ndim(v) == 0 and v == 0 for k, v in kw.items())
        kw_fista_ok = all((k == "theta") or (k == "x0" and np.ndim(v) == 0 and not isinstance(v, DeviceVector) and v == 0)
                          for k, v in kw.items())
        if (isinstance(solver, CGNR) and isinstance(b, DeviceMatrix) and b.N > 1 and not solver.constr and kw_cgnr_ok
                and not isinstance(solver.normalizeReg, MeasurementBasedNormalization)):  # per-column lambda: not batched
            try:
                st = BatchedState(solver, b)
                lib, h = b.ctx.lib, b.ctx.handle
                relTol = (solver.state.states[0] if isinstance(solver.state, AbstractMatrixSolverState) and solver.state.states
                          else solver.state).relTol if not isinstance(solver.state, BatchedState) else solver.state.relTol
                st.relTol = relTol
                check(h, lib.rls_cgnr_init_batched(st._plan, b.ptr, b.lda, float(solver.L2.lam), float(relTol),
                                                   solver.iterations), "rls_cgnr_init_batched")
                solver.state = st
                return
            except _lib.RLSError:
                pass  # shape not covered by the one-pass kernel: independent per-column plans instead
        if (type(solver) is FISTA and isinstance(b, DeviceMatrix) and b.N > 1 and solver.A is not None
                and solver._fused_kinds() is not None and solver._fused_kinds()[0] != REG_TV
                and not isinstance(solver.normalizeReg, MeasurementBasedNormalization)
                and kw_fista_ok):
            try:
                st = FistaBatchedState(solver, b)
                lib, h = b.ctx.lib, b.ctx.handle
                ref = solver.state.states[0] if isinstance(solver.state, AbstractMatrixSolverState) and solver.state.states else solver.state
                st.rho, st.relTol = ref.rho, ref.relTol
                kind, lam_, slices, pk = solver._fused_kinds()
                check(h, lib.rls_fista_set_reg(st._plan, kind, lam_, slices, pk), "rls_fista_set_reg")
                check(h, lib.rls_fista_init_batched(st._plan, b.ptr, b.lda, float(st.rho), float(kw.get("theta", 1)),
                                                    float(st.relTol), solver.iterations,
                                                    1 if solver.restart == "gradient" else 0), "rls_fista_init_batched")
                solver.state = st
                return
            except _lib.RLSError:
                pass  # e.g. M or N not a multiple of 16: independent per-column plans instead
        if (type(solver) is ADMM and solver.precon is None and isinstance(b, DeviceMatrix) and b.N > 1 and solver.A is not None and not kw
                and solver.use_device_plan and solver._all_identity() and len(solver.reg) == 1 and solver.vary_rho == "none"
                and not isinstance(solver.normalizeReg, (MeasurementBasedNormalization, SystemMatrixBasedNormalization))):
            try:
                st = solver.state if isinstance(solver.state, AdmmBatchedState) and solver.state.K == b.N else AdmmBatchedState(solver, b)
                st.init(b)
                st.active = [True] * b.N
                solver.state = st
                return
            except _lib.RLSError:
                pass  # shape or regulariser not covered by the batched plan: independent per-column plans instead
        scheduler = MultiThreadingState
    if isinstance(solver.state, AdmmBatchedState):
        ref = solver.state
        solver.state = ADMMState(len(solver.reg), solver.rho, ref.absTol, ref.relTol, ref.tolInner)
    if isinstance(solver.state, FistaBatchedState):
        solver.state = FISTAState(solver.state.rho, 1, solver.state.relTol)
    elif isinstance(solver.state, BatchedState):
        solver.state = CGNRState(solver.state.relTol)
    cols = _columns(b)
    states = [solver._new_state() for _ in cols]  # deep copies of the state  :45-48
    solver.state = scheduler(states)
    for s, col in zip(states, cols):
        solver.init_(s, col, **kw)
    solver.state.active = [True] * len(states)


def iterate(solver: AbstractLinearSolver):
    """iterate(solver)   src/RegularizedLeastSquares.jl:191, src/MultiThreading.jl:52-78"""
    st = solver.state
    if isinstance(st, BatchedState):
        stat = st.status()
        st.active = [not s_.done for s_ in stat]
        if not any(st.active):
            return None
        if st.iteration > solver.iterations + 1:
            raise _lib.RLSError("batched solve: a column did not reach done() within `iterations` steps")
        st._step(1)
        st.iteration += 1
        return st.active, st
    if isinstance(st, AbstractMatrixSolverState):
        idx = [i for i, a in enumerate(st.active) if a]
        if not idx:
            return None
        for i in idx:
            if solver.iterate(st.states[i]) is None:
                st.active[i] = False
        return st.active, st
    return solver.iterate(st)


def solve_(solver: AbstractLinearSolver, b, callbacks=None, **kw):
    """solve!(solver, b; callbacks, kwargs...)   src/RegularizedLeastSquares.jl:103-117.
    Callbacks fire once after init! (iteration 0) and once per completed iteration."""
    if callbacks is None:
        cbs: List[Callable] = []
    elif callable(callbacks):
        cbs = [callbacks]
    else:
        cbs = list(callbacks)
    no_group = kw.pop("_no_group", False)
    _check_eltype(solver, b)
    if (not cbs and not kw and not no_group and isinstance(solver, CGNR) and isinstance(solver.state, CGNRState) and
            isinstance(b, DeviceVector) and isinstance(solver._op, OperatorHandle) and solver.A is not None and not solver._op.double and
            solver._op.M * solver._op.N * b.dtype.itemsize <= 128 * 1024):
        # a system small enough for ONE CU: init! and every iteration as one launch (a group of one; the same bits as init_ + steps)
        return solve_group_([solver], [b])[0]
    init_(solver, b, **kw)
    for cb in cbs:
        cb(solver, 0)
    if not cbs and isinstance(solver.state, BatchedState):
        st = solver.state
        st._step(solver.iterations if isinstance(st, FistaBatchedState) else min(solver.iterations, solver._op.N))
        while iterate(solver) is not None:  # normally returns None at once
            pass
        return solversolution(solver)
    if not cbs:
        st = solver.state
        for s in (st.states if isinstance(st, AbstractMatrixSolverState) else [st]):
            solver._run(s)
        if isinstance(st, AbstractMatrixSolverState):
            st.active = [False] * len(st.states)
        return solversolution(solver)
    it = 0
    while iterate(solver) is not None:
        it += 1
        for cb in cbs:
            cb(solver, it)
    return solversolution(solver)


def _filter_kwargs(T, kwarg_warning, kwargs):
    """filterKwargs  src/RegularizedLeastSquares.jl:267-278"""
    names = set(inspect.signature(T.__init__).parameters) - {"self"}
    kept = {k: v for k, v in kwargs.items() if k in names}
    dropped = [k for k in kwargs if k not in names]
    if dropped and kwarg_warning:
        warnings.warn("The following arguments were passed but filtered out: " + ", ".join(dropped) +
                      ". Please watch closely if this introduces unexpexted behaviour in your code.")
    return kept


def createLinearSolver(solver_type, A=None, *, kwargWarning: bool = True, **kwargs):
    """createLinearSolver(T, A; kwargs...) / createLinearSolver(T; AHA, kwargs...)  :288-294"""
    if not (isinstance(solver_type, type) and issubclass(solver_type, AbstractLinearSolver)):
        raise TypeError("solver must be an AbstractLinearSolver type")
    return solver_type(A, **_filter_kwargs(solver_type, kwargWarning, kwargs))


def isapplicable(solver, *args):
    """isapplicable(solverType, [A, x,] reg)   src/RegularizedLeastSquares.jl:223-258 (quirks included: the fallback for
    categories without a rule is `false`, the primal-dual rule is a TODO that returns `true`)"""
    T = solver if isinstance(solver, type) else type(solver)
    if len(args) == 2:        # (A, x): "TODO" in the reference, always applicable
        return True
    if len(args) == 3:        # (A, x, reg)
        return isapplicable(T, args[0], args[1]) and isapplicable(T, args[2])
    if len(args) != 1:
        raise TypeError("isapplicable(solver, reg) | isapplicable(solver, A, x) | isapplicable(solver, A, x, reg)")
    regs = _as_list(args[0])
    n_param = len(findsinks(AbstractParameterizedRegularization, regs))
    if issubclass(T, AbstractRowActionSolver):
        return n_param <= 2 and len(findsinks(L2Regularization, regs)) == 1
    if issubclass(T, AbstractPrimalDualSolver):
        return True
    if issubclass(T, AbstractProximalGradientSolver):
        return n_param == 1
    return False


def applicableSolverList(*args):
    """applicableSolverList(args...)   :265"""
    return [T for T in linearSolverList() if isapplicable(T, *args)]


def linearSolverList():
    """the solvers of the reference's linearSolverList() that this backend covers (the direct solvers, DAX and
    the primal-dual solver are outside the scope)"""
    return [CGNR, Kaczmarz, FISTA, OptISTA, POGM, ADMM, SplitBregman]


# ---- callbacks (src/Callbacks.jl) -- thin host-side helpers ------------------------------------


class StoreSolutionCallback:
    def __init__(self):
        self.solutions = []

    def __call__(self, solver, _it):
        x = solversolution(solver)
        self.solutions.append([v.to_host() for v in x] if isinstance(x, list) else x.to_host())


class StoreConvergenceCallback:
    def __init__(self):
        self.convMeas = {}

    def __call__(self, solver, _it):
        for k, v in solverconvergence(solver).items():
            self.convMeas.setdefault(k, []).append(v)


class CompareSolutionCallback:
    def __init__(self, ref, cmp=None):
        self.ref = np.asarray(ref)
        self.cmp = cmp or (lambda r, x: float(np.linalg.norm(r - x) / np.linalg.norm(r)))
        self.results = []

    def __call__(self, solver, _it):
        self.results.append(self.cmp(self.ref, solversolution(solver).to_host()))
