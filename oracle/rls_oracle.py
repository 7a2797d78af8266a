"""CPU oracle: NumPy restatement of the RegularizedLeastSquares.jl inner loop.

TEST INFRASTRUCTURE ONLY.  Nothing under ``regularizedleastsquares.jl_amd/`` may
import this file; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and there only as the checker.

What is restated (citations relative to /root/reference, v0.16.12):
  * CGNR            src/CGNR.jl:48-185
  * FISTA           src/FISTA.jl:57-189
  * ADMM            src/ADMM.jl:80-332
  * cg!             IterativeSolvers v0.9 (NOT in the reference tree; restated from
                    the published algorithm, call site src/ADMM.jl:244)
  * prox maps       src/proximalMaps/ProxL1.jl:18-22, ProxL2.jl:18-21, ProxL21.jl:26-35,
                    ProxTV.jl:82-145, ProxPositive.jl:16-20, ProxReal.jl:16-19,
                    src/Utils.jl:114-144 (enfReal!/enfPos!)
  * GradientOp      LinearOperatorCollection v2 (NOT in tree; call sites ProxTV.jl:46,108-109,123)
  * power_iterations src/Utils.jl:262-287
  * solve! cadence  src/RegularizedLeastSquares.jl:103-117
  * matrix solves   src/MultiThreading.jl:30-79

Pinning status: the reference is Julia and Julia is not installed in the build
container, so the reference itself cannot be executed.  The reference holds no
golden vectors for this path; its exact known-answer tests (L2 closed form,
Positive projection, callback cadence, matrix-solve == column solves, CGNR vs the
least-squares solution) are replayed in tests/test_oracle.py.  cg! and GradientOp
come from third-party packages that are absent from the tree: for those two the
oracle is "parity unpinned" (see DESIGN.md section 3).

All arithmetic runs in the dtype of the inputs: pass float32/complex64 to mirror
the reference's Float32 path, float64/complex128 for the high-precision truth.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------


def real_dtype(dt) -> np.dtype:
    return np.empty(0, dtype=dt).real.dtype


def _rt(x):
    """real scalar type constructor for array x"""
    return real_dtype(x.dtype).type


def dotc(x, y):
    """Julia dot(x, y): conjugates the FIRST argument (BLAS dotc)."""
    return np.vdot(x, y)


def nrm2(x):
    return np.linalg.norm(x)


# --------------------------------------------------------------------------------------
# operators (the L1 "array/operator protocol" of SURVEY section 1)
# --------------------------------------------------------------------------------------


class DenseOp:
    """Dense forward operator A (M x N)."""

    def __init__(self, A):
        self.A = np.asarray(A)
        self.dtype = self.A.dtype
        self.shape = self.A.shape
        self._AH = None

    def mul(self, x):
        return self.A @ x

    def mul_adj(self, y):
        # A' * y.  `self.A.conj().T @ y` materialises conj(A) on EVERY call (0.2 s at 4096 x 2048 complex128, against 5 ms for the
        # product): the conjugate transpose is formed once and kept -- the same expression, hence the same bits -- for matrices up
        # to 512 MiB; larger ones use conj(conj(y) A), which needs no copy at all
        if self._AH is None and self.A.nbytes <= (512 << 20):
            self._AH = np.ascontiguousarray(self.A.conj().T)
        if self._AH is not None:
            return self._AH @ y
        return np.conj(np.conj(y) @ self.A)


class NormalOp:
    """Matrix-free normal operator: v = A^H (A p) -- two GEMVs
    (LinearOperatorCollection.normalOperator; docs/src/literate/howto/normal_operator.jl:37-44)."""

    def __init__(self, A: DenseOp):
        self.A = A
        self.dtype = A.dtype
        n = A.shape[1]
        self.shape = (n, n)

    def mul(self, x):
        return self.A.mul_adj(self.A.mul(x))


class GramOp:
    """Gram-mode normal operator: AHA = A'*A formed once (src/CGNR.jl:49), one N x N GEMV per apply."""

    def __init__(self, A=None, AHA=None):
        if AHA is None:
            A = np.asarray(A)
            AHA = A.conj().T @ A
        self.G = np.asarray(AHA)
        self.dtype = self.G.dtype
        self.shape = self.G.shape

    def mul(self, x):
        return self.G @ x


# --------------------------------------------------------------------------------------
# proximal maps
# --------------------------------------------------------------------------------------


def prox_l1(x, lam):
    """src/proximalMaps/ProxL1.jl:18-22.  eps is added to the real part only."""
    T = _rt(x)
    eps = np.finfo(T).eps
    lam = T(lam)
    ax = np.abs(x)
    shrink = np.maximum(ax - lam, T(0))
    x[...] = (shrink * (x + eps)) / (ax + eps)
    return x


def norm_l1(x, lam):
    """src/proximalMaps/ProxL1.jl:29-32"""
    return _rt(x)(lam) * np.sum(np.abs(x))


def prox_l2(x, lam):
    """src/proximalMaps/ProxL2.jl:18-21: the Float64 literals promote the factor to Float64; the
    product is rounded back to the element type on store."""
    factor = 1.0 / (1.0 + 2.0 * float(lam))
    if np.iscomplexobj(x):
        x[...] = (x.astype(np.complex128) * factor).astype(x.dtype)
    else:
        x[...] = (x.astype(np.float64) * factor).astype(x.dtype)
    return x


def prox_l21(x, lam, slices):
    """src/proximalMaps/ProxL21.jl:30-35.  x is sliceLength x slices column-major; a group is one
    ROW (stride sliceLength).  An all-zero group evaluates (0-lam)/0: -Inf -> 0 for lam>0, NaN for
    lam==0 (Julia's max propagates NaN, as does np.maximum)."""
    T = _rt(x)
    lam = T(lam)
    n = x.shape[0]
    slen = n // slices
    # group i = x[i:sliceLength:end]: the stride runs to the END of x, so a ragged tail
    # (n % slices != 0) joins its group; element k uses group k mod slen (mod1 in the reference)
    idx = np.arange(n) % slen
    g = _l21_group_norms(x, slen, idx)
    with np.errstate(divide="ignore", invalid="ignore"):
        fac = np.maximum((g - lam) / g, T(0)).astype(T)
    x[...] = x * fac[idx]
    return x


def _l21_group_norms(x, slen, idx):
    T = _rt(x)
    a2 = (x.real.astype(T) ** 2 + x.imag.astype(T) ** 2) if np.iscomplexobj(x) else x.astype(T) ** 2
    return np.sqrt(np.bincount(idx, weights=a2.astype(np.float64), minlength=slen)).astype(T)


def norm_l21(x, lam, slices):
    """src/proximalMaps/ProxL21.jl:42-46"""
    T = _rt(x)
    slen = x.shape[0] // slices
    g = _l21_group_norms(x, slen, np.arange(x.shape[0]) % slen)
    return T(lam) * np.sum(g)


def enf_real(x):
    """src/Utils.jl:114-125"""
    if np.iscomplexobj(x):
        x.imag[...] = 0
    return x


def enf_pos(x):
    """src/Utils.jl:130-144.  Complex: re<0 => x = im*imag(x) (real part dropped, imaginary kept)."""
    if np.iscomplexobj(x):
        neg = x.real < 0
        x.real[neg] = 0
    else:
        x[x < 0] = 0
    return x


def prox_positive(x):
    """src/proximalMaps/ProxPositive.jl:16-20"""
    enf_real(x)
    enf_pos(x)
    return x


def prox_real(x):
    """src/proximalMaps/ProxReal.jl:16-19"""
    enf_real(x)
    return x


# ---- GradientOp (LinearOperatorCollection v2; restated, not in tree) ------------------


def _grad_block_len(shape, d):
    n = 1
    for k, s in enumerate(shape):
        n *= (s - 1) if k == d else s
    return n


def grad_len(shape, dims):
    return sum(_grad_block_len(shape, d) for d in dims)


def _as_dims(shape, dims):
    """dims are 1-based in the reference API; returns 0-based tuple."""
    if dims is None:
        return tuple(range(len(shape)))
    if isinstance(dims, (int, np.integer)):
        return (int(dims) - 1,)
    return tuple(int(d) - 1 for d in dims)


def grad_apply(x, shape, dims0):
    """g = grad(x): per dim d, g[i] = img[i] - img[i + e_d]; output extent shape[d]-1 along d; blocks
    concatenated in dims order.  Arrays are column-major (Julia reshape)."""
    img = x.reshape(shape, order="F")
    out = []
    for d in dims0:
        lo = [slice(None)] * len(shape)
        hi = [slice(None)] * len(shape)
        lo[d] = slice(0, shape[d] - 1)
        hi[d] = slice(1, shape[d])
        out.append((img[tuple(lo)] - img[tuple(hi)]).reshape(-1, order="F"))
    return np.concatenate(out) if out else np.zeros(0, dtype=x.dtype)


def grad_apply_t(g, shape, dims0):
    """x = grad^T g: +g at i, -g at i + e_d."""
    res = np.zeros(shape, dtype=g.dtype, order="F")
    off = 0
    for d in dims0:
        bshape = list(shape)
        bshape[d] -= 1
        n = int(np.prod(bshape))
        gb = g[off : off + n].reshape(bshape, order="F")
        off += n
        lo = [slice(None)] * len(shape)
        hi = [slice(None)] * len(shape)
        lo[d] = slice(0, shape[d] - 1)
        hi[d] = slice(1, shape[d])
        res[tuple(lo)] += gb
        res[tuple(hi)] -= gb
    return res.reshape(-1, order="F")


def prox_tv_fgp(x, lam, shape, dims=None, iterationsTV=10):
    """Fast gradient projection, src/proximalMaps/ProxTV.jl:89-125 (the only TV algorithm reachable
    through prox!).  Anisotropic per-component clip (ProxTV.jl:135-139); step constant
    1/(8 lam) whatever the number of dims (ProxTV.jl:109)."""
    T = _rt(x)
    lam = T(lam)
    dims0 = _as_dims(shape, dims)
    ng = grad_len(shape, dims0)
    pq = np.zeros(ng, dtype=x.dtype)
    rs = np.zeros(ng, dtype=x.dtype)
    pqOld = np.zeros(ng, dtype=x.dtype)
    t = T(1)
    step = T(1) / (T(8) * lam)
    for _ in range(iterationsTV):
        pqTmp = pqOld
        pqOld = pq
        pq = rs  # aliases rs: pq is updated in place in the buffer that held rs (:104,109)
        xTmp = x.copy()
        xTmp = (xTmp + (-lam) * grad_apply_t(rs, shape, dims0)).astype(x.dtype)
        pq[...] = (step * grad_apply(xTmp, shape, dims0) + pq).astype(x.dtype)
        pq[...] = pq / np.maximum(T(1), np.abs(pq))
        tOld = t
        t = (T(1) + np.sqrt(T(1) + T(4) * tOld * tOld)) / T(2)
        t2 = (tOld - T(1)) / t
        t3 = T(1) + t2
        rs = pqTmp
        rs[...] = t3 * pq - t2 * pqOld
    x[...] = (x + (-lam) * grad_apply_t(pq, shape, dims0)).astype(x.dtype)
    return x


def norm_tv(x, lam, shape, dims=None):
    """src/proximalMaps/ProxTV.jl:152-155"""
    return _rt(x)(lam) * np.sum(np.abs(grad_apply(x, shape, _as_dims(shape, dims))))


# --------------------------------------------------------------------------------------
# regularisation types (names preserved from the reference API)
# --------------------------------------------------------------------------------------


@dataclass
class L1Regularization:
    lam: float

    def prox(self, x, lam=None):
        return prox_l1(x, self.lam if lam is None else lam)

    def norm(self, x, lam=None):
        return norm_l1(x, self.lam if lam is None else lam)


@dataclass
class L2Regularization:
    lam: float

    def prox(self, x, lam=None):
        return prox_l2(x, self.lam if lam is None else lam)

    def norm(self, x, lam=None):
        lam = self.lam if lam is None else lam
        return _rt(x)(lam) * nrm2(x) ** 2


@dataclass
class L21Regularization:
    lam: float
    slices: int = 1

    def prox(self, x, lam=None):
        return prox_l21(x, self.lam if lam is None else lam, self.slices)

    def norm(self, x, lam=None):
        return norm_l21(x, self.lam if lam is None else lam, self.slices)


@dataclass
class TVRegularization:
    lam: float
    shape: Sequence[int] = (0,)
    dims: Optional[Sequence[int]] = None
    iterationsTV: int = 10  # ctor default src/proximalMaps/ProxTV.jl:39

    def prox(self, x, lam=None):
        return prox_tv_fgp(x, self.lam if lam is None else lam, tuple(self.shape), self.dims, self.iterationsTV)

    def norm(self, x, lam=None):
        return norm_tv(x, self.lam if lam is None else lam, tuple(self.shape), self.dims)


class PositiveRegularization:
    lam = None

    def prox(self, x, lam=None):
        return prox_positive(x)


class RealRegularization:
    lam = None

    def prox(self, x, lam=None):
        return prox_real(x)


def _is_projection(r):
    """sinktype(r) <: AbstractProjectionRegularization (findsinks, Regularization.jl:86)"""
    while getattr(r, "reg", None) is not None:  # nested terms: look at the sink
        r = r.reg
    return isinstance(r, (PositiveRegularization, RealRegularization)) or type(r).__name__ == "ProjectionRegularization"


def _normalize_regs(scheme, regs, A, b):
    """normalize(norm, regs, A, b): src/Regularization/NormalizedRegularization.jl:60-84.  `scheme` in {"none",
    "measurement", "systemmatrix"}; A is a DenseOp / ndarray / None.  Terms with a parameterised sink are wrapped in a
    NormalizedRegularization (an existing wrapper is re-wrapped with the new factor, :74), projections pass (:73)."""
    Amat = getattr(A, "A", A)
    factor = normalization_factor(scheme, Amat, b)
    if factor is None:
        return list(regs)
    out = []
    for r in regs:
        if _is_projection(r):
            out.append(r)
        elif isinstance(r, NormalizedRegularization):
            out.append(NormalizedRegularization(r.reg, factor))
        elif reg_lambda(r) is not None:
            out.append(NormalizedRegularization(r, factor))
        else:
            out.append(r)
    return out


def _normalize_in_solver(scheme, regs, A, b):
    """normalize(solver::AbstractLinearSolver, ...) as init! calls it: the system-matrix factor was applied by the
    constructor and is kept (:84); the measurement factor is recomputed from the vector init! hands over."""
    if scheme == "systemmatrix":
        return list(regs)
    return _normalize_regs(scheme, regs, A, b)


# --------------------------------------------------------------------------------------
# CGNR  (src/CGNR.jl)
# --------------------------------------------------------------------------------------


class CGNR:
    """src/CGNR.jl:48-89 (ctor), :107-130 (init!), :143-178 (iterate), :181-185 (done)."""

    def __init__(self, A, AHA=None, reg=None, iterations=10, relTol=None, normal="matrixfree", normalizeReg="none"):
        self.A = A if (A is None or hasattr(A, "mul")) else DenseOp(A)
        self.normalizeReg = normalizeReg
        if AHA is None:
            AHA = NormalOp(self.A) if normal == "matrixfree" else GramOp(self.A.A)
        elif not hasattr(AHA, "mul"):
            AHA = GramOp(AHA=AHA)
        self.AHA = AHA
        self.dtype = np.dtype(AHA.dtype)
        self.T = real_dtype(self.dtype).type
        regs = [] if reg is None else (list(reg) if isinstance(reg, (list, tuple)) else [reg])
        regs = _normalize_regs(normalizeReg, regs, self.A, None)  # src/CGNR.jl:68
        l2 = [r for r in regs if isinstance(sink(r), L2Regularization)]
        if len(l2) > 1:
            raise ValueError("Cannot unambigiously retrieve reg term of type L2Regularization")
        self.L2 = l2[0] if l2 else L2Regularization(0.0)
        self.constr = [r for r in regs if _is_projection(r)]
        rest = [r for r in regs if not isinstance(sink(r), L2Regularization) and not _is_projection(r)]
        if rest:
            raise ValueError(f"CGNR does not allow for more additional regularization terms, found {len(rest)}")
        self.iterations = int(iterations)
        self.relTol = self.T(np.finfo(self.T).eps if relTol is None else relTol)
        n = AHA.shape[1]
        self.N = n
        self.x = np.zeros(n, self.dtype)
        self.r = np.zeros(n, self.dtype)  # x0 in the reference: the normal-equation residual
        self.p = np.zeros(n, self.dtype)
        self.v = np.zeros(n, self.dtype)
        self.alpha = self.dtype.type(0)
        self.beta = self.dtype.type(0)
        self.zeta = self.dtype.type(0)
        self.iteration = 0
        self.z0 = self.T(0)

    def init(self, b):
        b = np.asarray(b, dtype=self.dtype)
        self.p[:] = 0
        self.v[:] = 0
        self.alpha = self.dtype.type(0)
        self.beta = self.dtype.type(0)
        self.zeta = self.dtype.type(0)
        self.iteration = 0
        self.x[:] = 0
        if self.A is None:
            self.r[:] = b  # initCGNR(x0, ::Nothing, b)  src/CGNR.jl:134
        else:
            self.r[:] = self.A.mul_adj(b)  # src/CGNR.jl:132
        self.z0 = self.T(nrm2(self.r))
        self.p[:] = self.r
        self.L2 = _normalize_in_solver(self.normalizeReg, [self.L2], self.A, b)[0]  # src/CGNR.jl:129

    def converged(self):
        with np.errstate(divide="ignore", invalid="ignore"):
            return self.T(nrm2(self.r)) / self.z0 <= self.relTol

    def done(self):
        return bool(self.converged()) or self.iteration >= min(self.iterations, self.N)

    def iterate(self):
        if self.done():
            for c in self.constr:
                c.prox(self.x)
            return None
        T = self.T
        self.v[:] = self.AHA.mul(self.p)
        self.zeta = self.dtype.type(T(nrm2(self.r)) ** 2)
        normvl = self.dtype.type(dotc(self.p, self.v))
        lam = T(self.L2.lam)
        if lam > 0:
            self.alpha = self.dtype.type(self.zeta / (normvl + lam * T(nrm2(self.p)) ** 2))
        else:
            self.alpha = self.dtype.type(self.zeta / normvl)
        self.x += self.p * self.alpha
        self.r += self.v * (-self.alpha)
        if lam > 0:
            self.r += self.p * (-lam) * self.alpha
        self.beta = self.dtype.type(dotc(self.r, self.r) / self.zeta)
        self.p *= self.beta
        self.p += self.r
        self.iteration += 1
        return self.x

    def solution(self):
        return self.x

    def convergence(self):
        return {"residual": self.T(nrm2(self.r))}


# --------------------------------------------------------------------------------------
# FISTA  (src/FISTA.jl)
# --------------------------------------------------------------------------------------


def power_iterations(AHA, b0, rtol=1e-3, maxiter=30):
    """src/Utils.jl:264-287 with an injectable start vector (the reference draws it from Julia's
    global RNG, which cannot be replayed)."""
    b = np.array(b0, dtype=AHA.dtype)
    lam = np.inf
    for _ in range(maxiter):
        b = b / nrm2(b)
        bold = b
        b = AHA.mul(bold)
        lam_old = lam
        lam = abs(dotc(bold, b))
        if abs(lam / lam_old - 1) < rtol:
            return lam
    return lam


class FISTA:
    """src/FISTA.jl:57-92 (ctor), :110-129 (init!), :139-185 (iterate), :187-189 (done)."""

    def __init__(self, A, AHA=None, reg=None, iterations=50, rho=None, theta=1, relTol=None,
                 restart="none", normal="matrixfree", normalizeReg="none"):
        self.A = A if (A is None or hasattr(A, "mul")) else DenseOp(A)
        self.normalizeReg = normalizeReg
        if AHA is None:
            AHA = NormalOp(self.A) if normal == "matrixfree" else GramOp(self.A.A)
        elif not hasattr(AHA, "mul"):
            AHA = GramOp(AHA=AHA)
        self.AHA = AHA
        self.dtype = np.dtype(AHA.dtype)
        self.T = real_dtype(self.dtype).type
        regs = [L1Regularization(0.0)] if reg is None else (list(reg) if isinstance(reg, (list, tuple)) else [reg])
        self.proj = [r for r in regs if _is_projection(r)]
        rest = [r for r in regs if not _is_projection(r)]
        if len(rest) != 1:
            raise ValueError(f"FISTA does not allow for more additional regularization terms, found {len(rest)}")
        self.reg = _normalize_regs(normalizeReg, rest, self.A, None)[0]  # src/FISTA.jl:86
        if rho is None:
            raise ValueError("oracle FISTA needs an explicit rho (the reference default uses the global RNG)")
        self.rho = self.T(rho)
        self.theta0 = self.T(theta)
        self.iterations = int(iterations)
        self.relTol = self.T(np.finfo(self.T).eps if relTol is None else relTol)
        self.restart = restart
        n = AHA.shape[1]
        self.x = np.zeros(n, self.dtype)
        self.x0 = np.zeros(n, self.dtype)
        self.xold = np.zeros(n, self.dtype)
        self.res = np.zeros(n, self.dtype)
        self.theta = self.T(theta)
        self.theta_old = self.T(theta)
        self.iteration = 0
        self.norm_x0 = self.T(1)
        self.rel_res_norm = self.T(np.inf)

    def init(self, b, x0=0, theta=1):
        b = np.asarray(b, dtype=self.dtype)
        if self.A is None:
            self.x0[:] = b
        else:
            self.x0[:] = self.A.mul_adj(b)
        self.iteration = 0
        self.norm_x0 = self.T(nrm2(self.x0))
        self.x[:] = x0
        self.xold[:] = 0
        self.res[:] = np.inf
        self.theta = self.T(theta)
        self.theta_old = self.T(theta)
        self.rel_res_norm = self.T(np.inf)
        # src/FISTA.jl:128: the measurement-based factor is taken of x0 = A^H b, not of b
        self.reg = _normalize_in_solver(self.normalizeReg, [self.reg], self.A, self.x0)[0]

    def done(self):
        return bool(self.rel_res_norm < self.relTol) or self.iteration >= self.iterations

    def iterate(self):
        if self.done():
            return None
        T = self.T
        self.x, self.xold = self.xold, self.x  # pointer swap :144-146
        self.x *= (T(1) - self.theta_old) / self.theta
        self.x += ((self.theta_old - T(1)) / self.theta + T(1)) * self.xold
        self.res[:] = self.AHA.mul(self.x)
        self.res -= self.x0
        self.x -= self.rho * self.res
        self.rel_res_norm = T(nrm2(self.res)) / self.norm_x0
        self.reg.prox(self.x, self.rho * T(self.reg.lam))
        for pr in self.proj:
            pr.prox(self.x)
        if self.restart == "gradient":
            if np.real(dotc(self.res, self.x - self.xold)) > 0:
                self.theta = T(1)
        self.theta_old = self.theta
        self.theta = (T(1) + np.sqrt(T(1) + T(4) * self.theta_old * self.theta_old)) / T(2)
        self.iteration += 1
        return self.x

    def solution(self):
        return self.x

    def convergence(self):
        return {"residual": self.T(nrm2(self.res))}


# --------------------------------------------------------------------------------------
# IterativeSolvers.cg!  (v0.9; restated -- parity unpinned)
# --------------------------------------------------------------------------------------


def cg_inplace(x, Aop: Callable, b, maxiter, reltol, abstol=0.0, Pl=None):
    """Unpreconditioned CG with warm start: u=0; r=b-A x; tol=max(reltol*||r||, abstol); prev=1.
    Returns number of iterations performed.  Pl (a callable r -> Pl \\ r, the `Pl = solver.precon` keyword of the call site
    src/ADMM.jl:244): the preconditioned recurrence of the same package (PCGIterable: c = Pl \\ r; rho = <c, r>;
    u = c + (rho / rho_prev) u; c = A u; alpha = rho / <u, c>), the stopping test still on ||r||."""
    T = _rt(x)
    if Pl is not None:
        return _pcg_inplace(x, Aop, b, maxiter, reltol, abstol, Pl)
    u = np.zeros_like(x)
    r = b.copy()
    c = Aop(x)
    r -= c
    residual = T(nrm2(r))
    tol = max(T(reltol) * residual, T(abstol))
    prev = T(1)
    it = 0
    while it < maxiter and residual > tol:
        beta = residual * residual / (prev * prev)
        u[:] = r + beta * u
        c = Aop(u)
        alpha = x.dtype.type(residual * residual / dotc(u, c))
        x += alpha * u
        r -= alpha * c
        prev = residual
        residual = T(nrm2(r))
        it += 1
    return it


def _pcg_inplace(x, Aop, b, maxiter, reltol, abstol, Pl):
    T = _rt(x)
    u = np.zeros_like(x)
    r = b.copy()
    r -= Aop(x)
    residual = T(nrm2(r))
    tol = max(T(reltol) * residual, T(abstol))
    rho = x.dtype.type(1)
    it = 0
    while it < maxiter and residual > tol:
        c = np.asarray(Pl(r), dtype=x.dtype)
        rho_prev = rho
        rho = x.dtype.type(dotc(c, r))
        beta = rho / rho_prev
        u[:] = c + beta * u
        c = Aop(u)
        alpha = x.dtype.type(rho / dotc(u, c))
        x += alpha * u
        r -= alpha * c
        residual = T(nrm2(r))
        it += 1
    return it


# --------------------------------------------------------------------------------------
# ADMM  (src/ADMM.jl)
# --------------------------------------------------------------------------------------


class IdentityTrafo:
    """opEye (src/ADMM.jl:84)"""

    def __init__(self, n):
        self.n_out = n

    def mul(self, x):
        return x.copy()

    def mul_adj(self, z):
        return z.copy()


class GradientTrafo:
    """regTrafo = GradientOp(...)  (src/ADMM.jl:74)"""

    def __init__(self, shape, dims=None):
        self.shape = tuple(shape)
        self.dims0 = _as_dims(self.shape, dims)
        self.n_out = grad_len(self.shape, self.dims0)

    def mul(self, x):
        return grad_apply(x, self.shape, self.dims0)

    def mul_adj(self, g):
        return grad_apply_t(g, self.shape, self.dims0)


class ADMM:
    """src/ADMM.jl:80-162 (ctor), :191-220 (init!), :230-322 (iterate), :324-332 (converged/done)."""

    def __init__(self, A, AHA=None, reg=None, regTrafo=None, rho=1e-1, vary_rho="none", iterations=10,
                 iterationsCG=10, absTol=None, relTol=None, tolInner=1e-5, normal="matrixfree", normalizeReg="none", precon=None):
        self.A = A if (A is None or hasattr(A, "mul")) else DenseOp(A)
        self.normalizeReg = normalizeReg
        self.precon = precon  # callable r -> Pl \\ r, or None = Identity() (src/ADMM.jl:82)
        if AHA is None:
            AHA = NormalOp(self.A) if normal == "matrixfree" else GramOp(self.A.A)
        elif not hasattr(AHA, "mul"):
            AHA = GramOp(AHA=AHA)
        self.AHA = AHA
        self.dtype = np.dtype(AHA.dtype)
        self.T = real_dtype(self.dtype).type
        T = self.T
        n = AHA.shape[1]
        regs = [L1Regularization(0.0)] if reg is None else (list(reg) if isinstance(reg, (list, tuple)) else [reg])
        self.proj = [r for r in regs if _is_projection(r)]
        self.reg = _normalize_regs(normalizeReg, [r for r in regs if not _is_projection(r)], self.A, None)  # src/ADMM.jl:139
        if regTrafo is None:
            regTrafo = [IdentityTrafo(n) for _ in self.reg]
        elif not isinstance(regTrafo, (list, tuple)):
            regTrafo = [regTrafo]
        self.regTrafo = list(regTrafo)
        assert len(self.reg) == len(self.regTrafo), "reg and regTrafo must have the same length"
        if np.isscalar(rho):
            self.rho0 = np.array([T(rho) for _ in self.reg], dtype=T)
        else:
            self.rho0 = np.asarray(rho, dtype=T)
        self.vary_rho = vary_rho
        self.iterations = int(iterations)
        self.iterationsCG = int(iterationsCG)
        eps = np.finfo(T).eps
        self.absTol = T(eps if absTol is None else absTol)
        self.relTol = T(eps if relTol is None else relTol)
        self.tolInner = T(tolInner)
        self.x = np.zeros(n, self.dtype)
        self.xold = np.zeros(n, self.dtype)
        self.beta = np.zeros(n, self.dtype)
        self.beta_y = np.zeros(n, self.dtype)
        self.z = [np.zeros(t.n_out, self.dtype) for t in self.regTrafo]
        self.zold = [np.zeros(t.n_out, self.dtype) for t in self.regTrafo]
        self.u = [np.zeros(t.n_out, self.dtype) for t in self.regTrafo]
        self.uold = [np.zeros(t.n_out, self.dtype) for t in self.regTrafo]
        k = len(self.reg)
        self.rho = self.rho0.copy()
        self.rk = np.full(k, np.inf, T)
        self.sk = np.full(k, np.inf, T)
        self.eps_pri = np.zeros(k, T)
        self.eps_dua = np.zeros(k, T)
        self.Delta = np.full(k, np.inf, T)
        self.sigma_abs = T(0)
        self.iteration = 0
        self.cg_iters: List[int] = []

    def composite_mul(self, u):
        """compositeAHA = AHA + sum_i rho_i Phi_i^H Phi_i  (src/ADMM.jl:141-159)"""
        out = self.AHA.mul(u)
        for i, t in enumerate(self.regTrafo):
            out = out + self.rho[i] * t.mul_adj(t.mul(u))
        return out.astype(self.dtype)

    def init(self, b, x0=0):
        b = np.asarray(b, dtype=self.dtype)
        T = self.T
        self.x[:] = x0
        if self.A is None:
            self.beta_y[:] = b
        else:
            self.beta_y[:] = self.A.mul_adj(b)
        for i, t in enumerate(self.regTrafo):
            self.z[i][:] = t.mul(self.x)
            self.u[i][:] = 0
        self.rk[:] = np.inf
        self.sk[:] = np.inf
        self.eps_pri[:] = 0
        self.eps_dua[:] = 0
        self.sigma_abs = T(np.sqrt(T(len(b)))) * self.absTol
        self.Delta[:] = np.inf
        self.rho[:] = self.rho0
        self.iteration = 0
        self.cg_iters = []
        self.reg = _normalize_in_solver(self.normalizeReg, self.reg, self.A, b)  # src/ADMM.jl:219, src/SplitBregman.jl:196

    def converged(self):
        for i in range(len(self.reg)):
            if self.rk[i] >= self.sigma_abs + self.relTol * self.eps_pri[i]:
                return False
            if self.sk[i] >= self.sigma_abs + self.relTol * self.eps_dua[i]:
                return False
        return True

    def done(self):
        return self.converged() or self.iteration >= self.iterations

    def iterate(self):
        if self.done():
            return None
        T = self.T
        # 1. x update: (A'A + sum rho Phi'Phi) x = A'b + sum rho Phi'(z - u)       :236-244
        self.beta[:] = self.beta_y
        for i, t in enumerate(self.regTrafo):
            self.beta += self.rho[i] * t.mul_adj(self.z[i])
            self.beta += (-self.rho[i]) * t.mul_adj(self.u[i])
        self.xold[:] = self.x
        self.cg_iters.append(cg_inplace(self.x, self.composite_mul, self.beta, self.iterationsCG, self.tolInner, Pl=self.precon))
        for pr in self.proj:
            pr.prox(self.x)
        for i, t in enumerate(self.regTrafo):
            self.z[i], self.zold[i] = self.zold[i], self.z[i]  # swap :252-254
            self.z[i][:] = t.mul(self.x)
            self.z[i] += self.u[i]
            if self.rho[i] != 0:
                self.reg[i].prox(self.z[i], T(self.reg[i].lam) / (T(2) * self.rho[i]))  # :261
            self.uold[i][:] = self.u[i]
            self.u[i] += t.mul(self.x)
            self.u[i] -= self.z[i]
            # convergence bookkeeping that hijacks xold / zold as scratch :282-299
            self.xold[:] = self.x - self.xold
            self.zold[i][:] = self.z[i] - self.zold[i]
            self.uold[i][:] = self.u[i] - self.uold[i]
            Delta_old = self.Delta[i]
            self.Delta[i] = T(nrm2(self.xold)) + T(nrm2(self.zold[i])) + T(nrm2(self.uold[i]))
            self.xold[:] = t.mul_adj(self.zold[i])
            self.sk[i] = self.rho[i] * T(nrm2(self.xold))
            self.zold[i][:] = t.mul(self.x)
            self.eps_pri[i] = max(T(nrm2(self.zold[i])), T(nrm2(self.z[i])))
            self.zold[i] -= self.z[i]
            self.rk[i] = T(nrm2(self.zold[i]))
            self.xold[:] = t.mul_adj(self.u[i])
            self.eps_dua[i] = self.rho[i] * T(nrm2(self.xold))
            with np.errstate(divide="ignore", invalid="ignore"):
                if (self.vary_rho == "balance" and self.rk[i] / self.eps_pri[i] > T(10) * self.sk[i] / self.eps_dua[i]) or (
                    self.vary_rho == "PnP" and self.Delta[i] / Delta_old > T(0.9)
                ):
                    self.rho[i] *= T(2)
                    self.u[i] /= T(2)
                elif self.vary_rho == "balance" and self.sk[i] / self.eps_dua[i] > T(10) * self.rk[i] / self.eps_pri[i]:
                    self.rho[i] /= T(2)
                    self.u[i] *= T(2)
        self.iteration += 1
        return self.x

    def solution(self):
        return self.x

    def convergence(self):
        return {"primal": self.rk.copy(), "dual": self.sk.copy()}


# --------------------------------------------------------------------------------------
# driver: solve! cadence and matrix right-hand sides
# --------------------------------------------------------------------------------------


def solve(solver, b, callbacks=None, **kw):
    """src/RegularizedLeastSquares.jl:103-117: init!, callbacks(solver, 0), then one callback per
    completed iteration.  A 2-D b runs the SequentialState/MultiThreadingState semantics of
    src/MultiThreading.jl:30-79 (independent per-column states, per-column retirement)."""
    b = np.asarray(b)
    if callbacks is None:
        callbacks = []
    elif callable(callbacks):
        callbacks = [callbacks]
    if b.ndim == 2:
        return _solve_matrix(solver, b, callbacks, **kw)
    solver.init(b, **kw)
    for cb in callbacks:
        cb(solver, 0)
    it = 0
    while solver.iterate() is not None:
        it += 1
        for cb in callbacks:
            cb(solver, it)
    return solver.solution()


def _solve_matrix(solver, B, callbacks, **kw):
    import copy

    states = [copy.deepcopy(solver) for _ in range(B.shape[1])]
    for s, col in zip(states, B.T):
        s.init(np.ascontiguousarray(col), **kw)
    active = [True] * len(states)
    for cb in callbacks:
        cb(solver, 0)
    it = 0
    while any(active):
        for i, s in enumerate(states):
            if active[i] and s.iterate() is None:
                active[i] = False
        it += 1
        # iterate(solver, ::AbstractMatrixSolverState) returns non-nothing whenever a state was
        # active at entry, so the callback also fires for the round that retires the last state
        for cb in callbacks:
            cb(solver, it)
    solver._matrix_states = states
    return np.stack([s.solution() for s in states], axis=1)


# --------------------------------------------------------------------------------------
# synthetic inputs: zero-mean normal entries, planted solution
# --------------------------------------------------------------------------------------


def make_problem(M, N, dtype, seed, n_rhs=None):
    """A ~ randn (complex: (g1 + i g2)/sqrt 2), x_true ~ randn, b = A x_true computed in float64 and
    cast.  Returns (A [Fortran order], x_true, b)."""
    rng = np.random.default_rng(seed)
    dt = np.dtype(dtype)
    cplx = dt.kind == "c"

    def draw(*shape):
        if cplx:
            return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / math.sqrt(2.0)
        return rng.standard_normal(shape)

    A64 = draw(M, N)
    xs = (N,) if n_rhs is None else (N, n_rhs)
    x64 = draw(*xs)
    b64 = A64 @ x64
    return np.asfortranarray(A64.astype(dt)), x64.astype(dt), np.asfortranarray(b64.astype(dt))


# --------------------------------------------------------------------------------------
# next-tier solvers: OptISTA, POGM, SplitBregman -- same op mix as FISTA / ADMM
# --------------------------------------------------------------------------------------


class OptISTA:
    """src/OptISTA.jl:61-110 (ctor), :129-160 (init!), :169-209 (iterate)"""

    def __init__(self, A, AHA=None, reg=None, iterations=50, rho=None, theta=1, relTol=None, normal="matrixfree",
                 normalizeReg="none"):
        self.A = A if (A is None or hasattr(A, "mul")) else DenseOp(A)
        self.normalizeReg = normalizeReg
        if AHA is None:
            AHA = NormalOp(self.A) if normal == "matrixfree" else GramOp(self.A.A)
        elif not hasattr(AHA, "mul"):
            AHA = GramOp(AHA=AHA)
        self.AHA = AHA
        self.dtype = np.dtype(AHA.dtype)
        self.T = real_dtype(self.dtype).type
        regs = [L1Regularization(0.0)] if reg is None else (list(reg) if isinstance(reg, (list, tuple)) else [reg])
        rest = [r for r in regs if not _is_projection(r)]
        if len(rest) != 1:
            raise ValueError(f"OptISTA does not allow for more additional regularization terms, found {len(rest)}")
        self.reg = _normalize_regs(normalizeReg, rest, self.A, None)[0]  # src/OptISTA.jl:99
        if rho is None:
            raise ValueError("oracle OptISTA needs an explicit rho")
        self.rho = self.T(rho)
        self.iterations = int(iterations)
        self.relTol = self.T(np.finfo(self.T).eps if relTol is None else relTol)
        n = AHA.shape[1]
        self.x, self.x0, self.y, self.z, self.zold, self.res = (np.zeros(n, self.dtype) for _ in range(6))
        self.iteration = 0
        self.rel_res_norm = self.T(np.inf)

    def init(self, b, x0=0, theta=1):
        T = self.T
        b = np.asarray(b, dtype=self.dtype)
        self.x0[:] = b if self.A is None else self.A.mul_adj(b)
        self.norm_x0 = T(nrm2(self.x0))
        self.x[:] = x0
        self.y[:] = self.x
        self.z[:] = self.x
        self.zold[:] = self.x
        self.res[:] = np.inf
        self.theta = T(theta)
        self.theta_old = T(theta)
        tn = T(theta)
        for _ in range(self.iterations - 1):
            tn = (T(1) + np.sqrt(T(1) + T(4) * tn * tn)) / T(2)
        self.theta_n = (T(1) + np.sqrt(T(1) + T(8) * tn * tn)) / T(2)
        self.rel_res_norm = T(np.inf)
        self.iteration = 0
        self.reg = _normalize_in_solver(self.normalizeReg, [self.reg], self.A, self.x0)[0]  # src/OptISTA.jl:154

    def done(self):
        return bool(self.rel_res_norm < self.relTol) or self.iteration >= self.iterations

    def iterate(self):
        if self.done():
            return None
        T = self.T
        th, tn = self.theta, self.theta_n
        self.gamma = T(2) * th / (tn * tn) * (tn * tn - T(2) * th * th + th)
        self.theta_old = th
        if self.iteration == self.iterations - 1:
            self.theta = (T(1) + np.sqrt(T(1) + T(8) * th * th)) / T(2)
        else:
            self.theta = (T(1) + np.sqrt(T(1) + T(4) * th * th)) / T(2)
        alpha = (self.theta_old - T(1)) / self.theta
        beta = self.theta_old / self.theta
        self.zold[:] = self.z
        self.z[:] = self.y
        self.res[:] = self.AHA.mul(self.x)
        self.res -= self.x0
        self.y -= (self.rho * self.gamma) * self.res
        self.rel_res_norm = T(nrm2(self.res)) / self.norm_x0
        self.reg.prox(self.y, self.rho * self.gamma * T(self.reg.lam))
        self.z /= -self.gamma
        self.z += self.x + self.y / self.gamma
        self.x *= -beta
        self.x += (T(1) + alpha + beta) * self.z
        self.x -= alpha * self.zold
        self.iteration += 1
        return self.x

    def solution(self):
        return self.x


class POGM:
    """src/POGM.jl:75-110 (ctor), :133-160 (init!), :169-237 (iterate).  gamma is NOT reset by init!
    (reference behaviour): it starts at 1 in a fresh solver and carries over between solves."""

    def __init__(self, A, AHA=None, reg=None, iterations=50, rho=None, theta=1, sigma_fac=1, relTol=None,
                 restart="none", normal="matrixfree", normalizeReg="none"):
        self.A = A if (A is None or hasattr(A, "mul")) else DenseOp(A)
        self.normalizeReg = normalizeReg
        if AHA is None:
            AHA = NormalOp(self.A) if normal == "matrixfree" else GramOp(self.A.A)
        elif not hasattr(AHA, "mul"):
            AHA = GramOp(AHA=AHA)
        self.AHA = AHA
        self.dtype = np.dtype(AHA.dtype)
        self.T = real_dtype(self.dtype).type
        regs = [L1Regularization(0.0)] if reg is None else (list(reg) if isinstance(reg, (list, tuple)) else [reg])
        self.proj = [r for r in regs if _is_projection(r)]
        rest = [r for r in regs if not _is_projection(r)]
        if len(rest) != 1:
            raise ValueError(f"POGM does not allow for more additional regularization terms, found {len(rest)}")
        self.reg = _normalize_regs(normalizeReg, rest, self.A, None)[0]  # src/POGM.jl:108
        if rho is None:
            raise ValueError("oracle POGM needs an explicit rho")
        T = self.T
        self.rho = T(rho)
        self.sigma_fac = T(sigma_fac)
        self.iterations = int(iterations)
        self.relTol = T(np.finfo(T).eps if relTol is None else relTol)
        self.restart = restart
        n = AHA.shape[1]
        self.x, self.x0, self.xold, self.y, self.z, self.w, self.res = (np.zeros(n, self.dtype) for _ in range(7))
        self.gamma = T(1)
        self.gamma_old = T(1)
        self.sigma = T(1)
        self.iteration = 0
        self.rel_res_norm = T(np.inf)

    def init(self, b, x0=0, theta=1):
        T = self.T
        b = np.asarray(b, dtype=self.dtype)
        self.x0[:] = b if self.A is None else self.A.mul_adj(b)
        self.norm_x0 = T(nrm2(self.x0))
        self.x[:] = x0
        self.xold[:] = 0
        self.y[:] = 0
        self.z[:] = 0
        if self.restart != "none":
            self.w[:] = 0
        self.res[:] = np.inf
        self.theta = T(theta)
        self.theta_old = T(theta)
        self.sigma = T(1)
        self.rel_res_norm = T(np.inf)
        self.iteration = 0
        self.reg = _normalize_in_solver(self.normalizeReg, [self.reg], self.A, self.x0)[0]  # src/POGM.jl:163

    def done(self):
        return bool(self.rel_res_norm < self.relTol) or self.iteration >= self.iterations

    def iterate(self):
        if self.done():
            return None
        T = self.T
        self.xold[:] = self.x
        self.res[:] = self.AHA.mul(self.x)
        self.res -= self.x0
        self.x -= self.rho * self.res
        self.rel_res_norm = T(nrm2(self.res)) / self.norm_x0
        self.theta_old = self.theta
        if self.iteration == self.iterations - 1 and self.restart != "none":
            self.theta = (T(1) + np.sqrt(T(1) + T(8) * self.theta_old ** 2)) / T(2)
        else:
            self.theta = (T(1) + np.sqrt(T(1) + T(4) * self.theta_old ** 2)) / T(2)
        alpha = (self.theta_old - T(1)) / self.theta
        beta = self.sigma * self.theta_old / self.theta
        self.gamma_old = self.gamma
        if self.restart == "gradient":
            self.gamma = self.rho * (T(1) + alpha + beta)
        else:
            self.gamma = self.rho * (T(2) * self.theta_old + self.theta - T(1)) / self.theta
        self.x, self.y = self.y, self.x  # swap
        self.x *= -alpha
        self.x += (T(1) + alpha + beta) * self.y
        self.x -= (beta + self.rho * alpha / self.gamma_old) * self.xold
        self.x += (self.rho * alpha / self.gamma_old) * self.z
        self.z[:] = self.x
        self.reg.prox(self.x, self.gamma * T(self.reg.lam))
        for pr in self.proj:
            pr.prox(self.x)
        if self.restart == "gradient":
            self.w += self.y + (self.rho / self.gamma) * (self.x - self.z)
            if np.real((dotc(self.w, self.x) - dotc(self.w, self.z)) / self.gamma - dotc(self.w, self.res)) < 0:
                self.sigma = T(1)
                self.theta = T(1)
            else:
                self.sigma = self.sigma * self.sigma_fac
            self.w[:] = (self.rho / self.gamma) * (self.z - self.x) - self.y
        self.iteration += 1
        return self.x

    def solution(self):
        return self.x


class SplitBregman(ADMM):
    """src/SplitBregman.jl:82-140 (ctor), :166-200 (init!), :204-271 (iterate), :273-282 (converged/done).
    Shares ADMM's composite operator and cg!; prox threshold lambda/rho (no factor 2), Bregman update of
    the right-hand side every iterationsInner inner iterations."""

    def __init__(self, A, AHA=None, reg=None, regTrafo=None, rho=1e-1, iterations=10, iterationsInner=10,
                 iterationsCG=10, absTol=None, relTol=None, tolInner=1e-5, normal="matrixfree", normalizeReg="none", precon=None):
        super().__init__(A, AHA=AHA, reg=reg, regTrafo=regTrafo, rho=rho, iterations=iterations,
                         iterationsCG=iterationsCG, absTol=absTol, relTol=relTol, tolInner=tolInner, normal=normal,
                         normalizeReg=normalizeReg, precon=precon)
        self.iterationsInner = int(iterationsInner)
        self.ybreg = np.zeros_like(self.x)

    def init(self, b, x0=0):
        super().init(b, x0=x0)
        self.ybreg[:] = self.beta_y
        self.iter_cnt = 1
        self.iteration = 1

    def done(self):
        return self.converged() or (self.iteration == 1 and self.iter_cnt > self.iterations)

    def iterate(self):
        if self.done():
            return None
        T = self.T
        self.beta[:] = self.beta_y
        for i, t in enumerate(self.regTrafo):
            self.beta += self.rho[i] * t.mul_adj(self.z[i])
            self.beta += (-self.rho[i]) * t.mul_adj(self.u[i])
        self.cg_iters.append(cg_inplace(self.x, self.composite_mul, self.beta, self.iterationsCG, self.tolInner, Pl=self.precon))
        for pr in self.proj:
            pr.prox(self.x)
        for i, t in enumerate(self.regTrafo):
            self.z[i], self.zold[i] = self.zold[i], self.z[i]
            self.z[i][:] = t.mul(self.x)
            self.z[i] += self.u[i]
            if self.rho[i] != 0:
                self.reg[i].prox(self.z[i], T(self.reg[i].lam) / self.rho[i])
            self.u[i] += t.mul(self.x)
            self.u[i] -= self.z[i]
            self.rk[i] = T(nrm2(t.mul(self.x) - self.z[i]))
            self.sk[i] = T(nrm2(self.rho[i] * t.mul_adj(self.z[i] - self.zold[i])))
            self.eps_pri[i] = max(T(nrm2(t.mul(self.x))), T(nrm2(self.z[i])))
            self.eps_dua[i] = T(nrm2(self.rho[i] * t.mul_adj(self.u[i])))
        if self.converged() or self.iteration >= self.iterationsInner:
            self.beta_y += self.ybreg
            self.beta_y -= self.AHA.mul(self.x)
            for i, t in enumerate(self.regTrafo):
                self.z[i][:] = t.mul(self.x)
                self.u[i][:] = 0
            self.iter_cnt += 1
            self.iteration = 0
        self.iteration += 1
        return self.x


# --------------------------------------------------------------------------------------------
# regularisation normalisation (src/Regularization/NormalizedRegularization.jl:40-58) and the
# row-weighted operator ProdOp(WeightingOp(w), A) (docs/src/literate/howto/normal_operator.jl:41-44)
# --------------------------------------------------------------------------------------------


def normalization_factor(scheme: str, A=None, b=None):
    """scheme in {"none", "measurement", "systemmatrix"}; returns None for "none" (:59)"""
    if scheme == "none":
        return None
    if scheme == "measurement":  # norm(b, 1) / length(b), 1 without b (:40-43)
        return 1.0 if b is None else float(np.sum(np.abs(b)) / b.size)
    if scheme == "systemmatrix":  # energy[m] = sqrt(rownorm²(A, m)); norm(energy)^2 / N (:47-58)
        if A is None:
            raise ValueError("SystemMatrixBasedNormalization requires supplying A to the constructor of the solver")
        energy = np.sqrt(np.sum(np.abs(A) ** 2, axis=1))
        return float(np.linalg.norm(energy) ** 2 / A.shape[1])
    raise ValueError(scheme)


def weighted_operator(w, A):
    """(W A) as a dense matrix; its normal operator is A^H W^H W A"""
    return np.asarray(w)[:, None] * A


# --------------------------------------------------------------------------------------
# Kaczmarz  (src/Kaczmarz.jl)
# --------------------------------------------------------------------------------------


def init_kaczmarz(A, lam):
    """initkaczmarz (src/Kaczmarz.jl:372-383): denom[i] = 1 / (rownorm²(A, row) + lambda) for the rows with
    non-zero norm, rowindex = those rows (0-based here)"""
    T = real_dtype(A.dtype).type
    s2 = np.sum(np.abs(A) ** 2, axis=1).astype(T)
    rowindex = np.nonzero(s2 > 0)[0]
    denom = (T(1) / (s2[rowindex] + T(lam))).astype(T)
    return denom, rowindex


def row_probabilities(A, rowindex):
    """rowProbabilities (src/Kaczmarz.jl:327-335)"""
    s2 = np.sum(np.abs(A) ** 2, axis=1)
    return s2[rowindex] / np.sum(s2)


class Kaczmarz:
    """src/Kaczmarz.jl:76-159 (ctor), :178-217 (init!), :283-299 (iterate), :303-308 (iterate_row_index),
    :320 (done).  Deterministic row orders only are pinned: `shuffleRows` / `randomized` take the row order from
    a NumPy generator (the reference uses Julia's global RNG), passed in as `order_fn(iteration) -> positions`.
    The greedy-randomized variant is CPU-only in the reference (test/testKaczmarz.jl:114) and not restated.
    L2 lambda may be a vector (Tikhonov matrix, :385-398): A <- A * diag(1/sqrt(lambda)), lambda = 1,
    solution scaled by 1/sqrt(lambda) (:262-264)."""

    def __init__(self, A, reg=None, iterations=10, order_fn=None):
        A = np.asarray(A)
        self.dtype = A.dtype
        self.T = real_dtype(self.dtype).type
        regs = [] if reg is None else (list(reg) if isinstance(reg, (list, tuple)) else [reg])
        l2 = [r for r in regs if isinstance(r, L2Regularization)]
        self.L2 = l2[0] if l2 else L2Regularization(0.0)
        proj = [r for r in regs if _is_projection(r)]
        rest = [r for r in regs if not isinstance(r, L2Regularization) and not _is_projection(r)]
        if len(rest) > 1:
            raise ValueError(f"Kaczmarz does not allow for more than one additional regularization term, found {len(rest)}")
        self.reg = proj + rest
        lam = self.L2.lam
        self.lam_vec = None
        if np.ndim(lam) == 1:
            self.lam_vec = np.asarray(lam, dtype=self.T)
            A = (A * (self.T(1) / np.sqrt(self.lam_vec))[None, :]).astype(self.dtype)
            lam = self.T(1)
        self.A = A
        self.lam = self.T(lam)
        self.denom, self.rowindex = init_kaczmarz(A, self.lam)
        self.iterations = int(iterations)
        self.order_fn = order_fn
        M, N = A.shape
        self.u = np.zeros(M, self.dtype)
        self.x = np.zeros(N, self.dtype)
        self.vl = np.zeros(M, self.dtype)
        self.eps_w = self.T(0)
        self.iteration = 0

    def init(self, b, x0=0):
        self.x[:] = x0
        self.vl[:] = 0
        self.u[:] = b
        self.eps_w = self.T(1) if self.lam_vec is not None else self.T(np.sqrt(self.lam))
        self.iteration = 0

    def done(self):
        return self.iteration >= self.iterations

    def iterate(self):
        if self.done():
            return None
        order = range(len(self.rowindex)) if self.order_fn is None else self.order_fn(self.iteration)
        for i in order:
            row = self.rowindex[i]
            a = self.A[row]
            tau = np.sum(a * self.x)  # dot_with_matrix_row: dotu, no conjugation (src/Utils.jl:55-88)
            alpha = self.denom[i] * (self.u[row] - tau - self.eps_w * self.vl[row])
            self.x += alpha * np.conj(a)  # kaczmarz_update! (src/Kaczmarz.jl:435-439)
            self.vl[row] += alpha * self.eps_w
        for r in self.reg:
            r.prox(self.x)
        self.iteration += 1
        return self.x

    def solution(self):
        if self.lam_vec is not None:
            return self.x * (self.T(1) / np.sqrt(self.lam_vec))
        return self.x

    def convergence(self):
        return {"residual": self.T(nrm2(self.A @ self.x - self.u))}


# --------------------------------------------------------------------------------------
# singular-value thresholding prox maps: src/proximalMaps/ProxNuclear.jl, ProxLLR.jl
# --------------------------------------------------------------------------------------


def _svt(X, lam):
    """U, S, V = svd(X); prox!(L1Regularization, S, lam); U * Diagonal(S) * V'  (ProxNuclear.jl:27-29)"""
    U, S, Vh = np.linalg.svd(X, full_matrices=False)
    S = prox_l1(S.astype(X.real.dtype), lam)
    return (U * S[None, :]) @ Vh


def prox_nuclear(x, lam, svtShape):
    X = np.asarray(x).reshape(svtShape, order="F")
    x[:] = _svt(X, lam).reshape(-1, order="F")
    return x


def prox_llr(x, lam, shape, blockSize, shift=None):
    """proxLLRNonOverlapping! (ProxLLR.jl:43-88) with an explicit block-grid shift (the reference draws
    rand(CartesianIndices(blockSize)) when randshift = true; all zeros = randshift false)"""
    shape, blockSize = tuple(shape), tuple(blockSize)
    nd = len(shape)
    K = x.size // int(np.prod(shape))
    X = np.asarray(x).reshape(shape + (K,), order="F")
    shift = (0,) * nd if shift is None else tuple(int(s_) for s_ in shift)
    xs = np.roll(X, shift, axis=tuple(range(nd)))  # circshift(x, shift): xs[i] = x[i - shift]
    for start in np.ndindex(*[-(-s_ // b) for s_, b in zip(shape, blockSize)]):
        sl = tuple(slice(st * b, min((st + 1) * b, s_)) for st, b, s_ in zip(start, blockSize, shape))
        blk = xs[sl]                                   # (b1', b2', ..., K), possibly cut at the edge
        m = int(np.prod(blk.shape[:-1]))
        M = np.zeros((int(np.prod(blockSize)), K), dtype=X.dtype)
        M[:m] = blk.reshape(m, K, order="F")
        if np.any(M != 0):
            Y = _svt(M, lam)
            xs[sl] = Y[:m].reshape(blk.shape, order="F")
    X[...] = np.roll(xs, tuple(-s_ for s_ in shift), axis=tuple(range(nd)))
    x[:] = X.reshape(-1, order="F")
    return x


def prox_llr_overlapping(x, lam, shape, blockSize):
    """proxLLROverlapping! (ProxLLR.jl:165-203): average of the distinct-block prox over every shift of the grid"""
    shape, blockSize = tuple(shape), tuple(blockSize)
    nd = len(shape)
    K = x.size // int(np.prod(shape))
    X = np.asarray(x).reshape(shape + (K,), order="F")
    pad = [(-s_) % b for s_, b in zip(shape, blockSize)]
    pshape = tuple(s_ + p for s_, p in zip(shape, pad))
    xp = np.zeros(pshape + (K,), dtype=X.dtype)
    core = tuple(slice(0, s_) for s_ in shape)
    xp[core] = X
    acc = np.zeros_like(X)
    n = 0
    for idx in np.ndindex(*blockSize):
        sh = tuple(i + 1 for i in idx)
        w = xp.reshape(-1, order="F").copy()
        prox_llr(w, lam, pshape, blockSize, sh)
        acc += w.reshape(pshape + (K,), order="F")[core]
        n += 1
    x[:] = (acc / n).reshape(-1, order="F")
    return x


class NuclearRegularization:
    def __init__(self, lam, svtShape):
        self.lam, self.svtShape = lam, tuple(svtShape)

    def prox(self, x, lam=None):
        return prox_nuclear(x, self.lam if lam is None else lam, self.svtShape)


class LLRRegularization:
    def __init__(self, lam, shape, blockSize, shift=None, fullyOverlapping=False):
        self.lam, self.shape, self.blockSize, self.shift, self.full = lam, tuple(shape), tuple(blockSize), shift, fullyOverlapping

    def prox(self, x, lam=None):
        lam = self.lam if lam is None else lam
        if self.full:
            return prox_llr_overlapping(x, lam, self.shape, self.blockSize)
        return prox_llr(x, lam, self.shape, self.blockSize, self.shift)


# --------------------------------------------------------------------------------------
# nested regularisation terms, projections by function, plug-and-play prior and its input transforms
#   src/Regularization/NestedRegularization.jl, ScaledRegularization.jl, MaskedRegularization.jl,
#   TransformedRegularization.jl, PlugAndPlayRegularization.jl, src/proximalMaps/ProxProj.jl, src/Transforms.jl
# (test infrastructure, like everything in this file)
# --------------------------------------------------------------------------------------


def innerreg(reg):
    """Regularization.jl:5 / NestedRegularization.jl:9"""
    return getattr(reg, "reg", None)


def collect_regs(reg):
    """iterate(reg) (Regularization.jl:6): the chain outermost -> innermost"""
    out = []
    while reg is not None:
        out.append(reg)
        reg = innerreg(reg)
    return out


def sink(reg):
    """NestedRegularization.jl:15"""
    return collect_regs(reg)[-1]


def reg_lambda(reg):
    """lambda(reg): Regularization.jl:29, NestedRegularization.jl:23, ScaledRegularization.jl:23"""
    if isinstance(reg, _Scaled):
        return reg_lambda(reg.reg) * reg.scalefactor()
    if innerreg(reg) is not None:
        return reg_lambda(reg.reg)
    return getattr(reg, "lam", None)


class _Nested:
    """prox!/norm forward to the inner term (NestedRegularization.jl:27-28); without lambda the nested lambda is
    used for parameterised sinks (:25-26)"""

    def __init__(self, reg):
        self.reg = reg

    @property
    def lam(self):
        return reg_lambda(self)

    def _args(self, lam):
        if lam is None and not _is_projection(self):
            lam = reg_lambda(self)
        return () if lam is None else (lam,)

    def prox(self, x, lam=None):
        return self._prox(x, *self._args(lam))

    def norm(self, x, lam=None):
        return self._norm(x, *self._args(lam))

    def _prox(self, x, *args):
        return self.reg.prox(x, *args)

    def _norm(self, x, *args):
        return self.reg.norm(x, *args)


class MaskedRegularization(_Nested):
    """MaskedRegularization.jl:19-37: prox!/norm on view(x, findall(mask))"""

    def __init__(self, reg, mask):
        super().__init__(reg)
        self.mask = np.asarray(mask, dtype=bool)

    def _prox(self, x, *args):
        z = x[self.mask]
        self.reg.prox(z, *args)
        x[self.mask] = z
        return x

    def _norm(self, x, *args):
        return self.reg.norm(x[self.mask], *args)


class TransformedRegularization(_Nested):
    """TransformedRegularization.jl:19-37: z = trafo * x ; prox!(reg, z) ; x = adjoint(trafo) * z.
    `trafo`: a matrix (ndarray) or an object with mul / mul_adj."""

    def __init__(self, reg, trafo):
        super().__init__(reg)
        self.trafo = DenseOp(trafo) if isinstance(trafo, np.ndarray) else trafo

    def _prox(self, x, *args):
        z = self.trafo.mul(x)
        self.reg.prox(z, *args)
        x[:] = self.trafo.mul_adj(z)
        return x

    def _norm(self, x, *args):
        return self.reg.norm(self.trafo.mul(x), *args)


class _Scaled(_Nested):
    def scalefactor(self):
        raise NotImplementedError


class FixedScaledRegularization(_Scaled):
    """ScaledRegularization.jl:27-35"""

    def __init__(self, reg, factor):
        super().__init__(reg)
        self.factor = factor

    def scalefactor(self):
        return self.factor


class NormalizedRegularization(FixedScaledRegularization):
    """NormalizedRegularization.jl:30-37: lambda(reg) = lambda(inner) * factor (ScaledRegularization.jl:23)"""


class FixedParameterRegularization(_Scaled):
    """ScaledRegularization.jl:43-52: discards the lambda passed in, uses the inner one"""

    def scalefactor(self):
        return 1.0

    def _prox(self, x, *discard):
        return self.reg.prox(x, reg_lambda(self.reg))

    def _norm(self, x, *discard):
        return self.reg.norm(x, reg_lambda(self.reg))


class AutoScaledRegularization(_Scaled):
    """ScaledRegularization.jl:55-77: the factor is maximum(abs.(x)) of the first vector it sees"""

    def __init__(self, reg):
        super().__init__(reg)
        self.factor = None

    def scalefactor(self):
        return 1.0 if self.factor is None else self.factor

    def _first(self, x, lam):
        if self.factor is None:
            self.factor = _rt(x)(np.max(np.abs(x)))
            return lam * self.factor
        return lam

    def _prox(self, x, lam):
        return self.reg.prox(x, self._first(x, lam))

    def _norm(self, x, lam):
        return self.reg.norm(x, self._first(x, lam))


class ProjectionRegularization:
    """src/proximalMaps/ProxProj.jl:3-20"""

    def __init__(self, projFunc=lambda x: x):
        self.projFunc = projFunc

    def prox(self, x, lam=None):
        x[:] = self.projFunc(x)
        return x

    def norm(self, x, lam=None):
        y = x.copy()
        self.prox(y)
        return np.inf if np.any(y != x) else 0.0


class MinMaxTransform:
    """src/Transforms.jl:4-16"""

    def __init__(self, x):
        self.min, self.max = np.min(x), np.max(x)

    def transform(self, x):
        return (x - self.min) / (self.max - self.min)

    def inverse_transform(self, x):
        return x * (self.max - self.min) + self.min


class IdentityTransform:
    """src/Transforms.jl:20-31"""

    def __init__(self, x=None):
        pass

    def transform(self, x):
        return x

    def inverse_transform(self, x):
        return x


class ZTransform:
    """src/Transforms.jl:34-46 (std is Julia's corrected sample standard deviation)"""

    def __init__(self, x):
        self.mean, self.std = np.mean(x), np.std(x, ddof=1)

    def transform(self, x):
        return (x - self.mean) / self.std

    def inverse_transform(self, x):
        return x * self.std + self.mean


class ClampedScalingTransform:
    """src/Transforms.jl:49-68"""

    def __init__(self, x, v_min, v_max):
        self.v_min, self.v_max = v_min, v_max
        self.mask = (x < v_min) | (x >= v_max)
        self.x = x

    def transform(self, x):
        return (np.clip(x, self.v_min, self.v_max) - self.v_min) / (self.v_max - self.v_min)

    def inverse_transform(self, x):
        out = x * (self.v_max - self.v_min) + self.v_min
        out[self.mask] = self.x[self.mask]
        return out


class PlugAndPlayRegularization:
    """src/Regularization/PlugAndPlayRegularization.jl:14-54.  PlugAndPlayRegularization(model, shape; ...) is the
    reduced constructor (:22) with lambda = 1."""

    def __init__(self, lam=1.0, model=None, shape=None, input_transform=MinMaxTransform, ignoreIm=False, **_kw):
        if callable(lam) and not isinstance(lam, (int, float)):
            lam, model, shape = 1.0, lam, model
        self.lam, self.model, self.shape = lam, model, list(shape)
        self.input_transform, self.ignoreIm = input_transform, bool(ignoreIm)
        self.warnings = []

    def prox(self, x, lam=None):
        lam = self.lam if lam is None else lam
        if np.iscomplexobj(x):
            re = self.prox(np.ascontiguousarray(x.real), lam)
            im = np.ascontiguousarray(x.imag) if self.ignoreIm else self.prox(np.ascontiguousarray(x.imag), lam)
            x[:] = re + 1j * im
            return x
        if lam != self.lam and (lam < 0.0 or lam > 1.0):
            temp = min(max(lam, 0.0), 1.0)
            self.warnings.append(f"{type(self).__name__} was given λ with value {lam}. Valid range is [0, 1]. λ changed to temp")
            lam = temp
        out = x.copy().reshape(self.shape, order="F")
        tf = self.input_transform(out)
        out = tf.transform(out)
        out = out - _rt(x)(lam) * (out - self.model(out))
        out = tf.inverse_transform(out)
        x[:] = out.reshape(-1, order="F")
        return x


PnPRegularization = PlugAndPlayRegularization
