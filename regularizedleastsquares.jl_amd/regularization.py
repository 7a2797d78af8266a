"""Regularisation types and prox!/norm dispatch, mirroring the reference's type tree
(src/Regularization/Regularization.jl:1-57, src/proximalMaps/*.jl) for device vectors.

Julia's `prox!(reg, x, lambda)` is `prox_(reg, x, lam)` here (trailing underscore = in place).
Both call forms of the reference are kept:
    prox_(reg_instance, x)            -> uses lambda(reg)         Regularization.jl:17
    prox_(RegType, x, lam, **kw)      -> constructs RegType(lam; kw...) first   Regularization.jl:39,55
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

from ._lib import check
from .arrays import DeviceVector


class AbstractRegularization:
    pass


class AbstractParameterizedRegularization(AbstractRegularization):
    lam: float


class AbstractProjectionRegularization(AbstractRegularization):
    lam = None


class L1Regularization(AbstractParameterizedRegularization):
    """src/proximalMaps/ProxL1.jl"""

    def __init__(self, lam, **_kw):
        self.lam = float(lam)

    def prox_(self, x: DeviceVector, lam: float):
        check(x.ctx.handle, x.ctx.lib.rls_prox_l1(x.ctx.handle, x.code, x.n, x.ptr, float(lam)), "rls_prox_l1")
        return x

    def norm(self, x: DeviceVector, lam: float) -> float:
        return float(lam) * x.norm1()


class L2Regularization(AbstractParameterizedRegularization):
    """src/proximalMaps/ProxL2.jl.  lambda may be a vector (Tikhonov matrix, used by Kaczmarz: src/Kaczmarz.jl:385-398);
    it is then kept in `lam_vector` and `lam` is NaN for the scalar consumers."""

    def __init__(self, lam, **_kw):
        import numpy as _np
        if _np.ndim(lam) == 1:
            self.lam_vector = _np.asarray(lam, dtype=_np.float32)
            self.lam = float("nan")
        else:
            self.lam_vector = None
            self.lam = float(lam)

    def prox_(self, x: DeviceVector, lam: float):
        check(x.ctx.handle, x.ctx.lib.rls_prox_l2(x.ctx.handle, x.code, x.n, x.ptr, float(lam)), "rls_prox_l2")
        return x

    def norm(self, x: DeviceVector, lam: float) -> float:
        return float(lam) * x.norm() ** 2


class L21Regularization(AbstractParameterizedRegularization):
    """src/proximalMaps/ProxL21.jl"""

    def __init__(self, lam, slices: int = 1, **_kw):
        self.lam = float(lam)
        self.slices = int(slices)

    def prox_(self, x: DeviceVector, lam: float):
        check(x.ctx.handle, x.ctx.lib.rls_prox_l21(x.ctx.handle, x.code, x.n, self.slices, x.ptr, float(lam)), "rls_prox_l21")
        return x

    def norm(self, x: DeviceVector, lam: float) -> float:
        r = (C.c_float * 2)()
        check(x.ctx.handle, x.ctx.lib.rls_norm_l21(x.ctx.handle, x.code, x.n, self.slices, x.ptr, float(lam), r), "rls_norm_l21")
        return float(r[0])


def _tv_geometry(shape: Sequence[int], dims) -> Tuple:
    shape = tuple(int(s) for s in shape)
    if dims is None:
        d0 = tuple(range(len(shape)))
    elif isinstance(dims, int):
        d0 = (dims - 1,)  # dims are 1-based in the reference API
    else:
        d0 = tuple(int(d) - 1 for d in dims)
    cs = (C.c_int64 * len(shape))(*shape)
    cd = (C.c_int32 * max(len(d0), 1))(*d0)
    return shape, d0, cs, cd


class TVRegularization(AbstractParameterizedRegularization):
    """src/proximalMaps/ProxTV.jl: FGP (the only algorithm reachable through prox!).
    Default iterationsTV = 10 as the constructor has it (:39)."""

    def __init__(self, lam, shape=(0,), dims=None, iterationsTV: int = 10, **_kw):
        self.lam = float(lam)
        self.shape = tuple(int(s) for s in shape)
        self.dims = dims
        self.iterationsTV = int(iterationsTV)
        self._ws = None  # TVParams scratch: per regulariser instance, as in the reference (:83-85)

    def prox_(self, x: DeviceVector, lam: float):
        shape, d0, cs, cd = _tv_geometry(self.shape, self.dims)
        n = 1
        for s in shape:
            n *= s
        if n != x.n:
            raise ValueError(f"TVRegularization: prod(shape)={n} does not match length(x)={x.n}")
        lib, h = x.ctx.lib, x.ctx.handle
        need = lib.rls_prox_tv_workspace_bytes(x.code, len(shape), cs, len(d0), cd)
        if self._ws is None or self._ws.n * self._ws.dtype.itemsize < need or self._ws.ctx is not x.ctx:
            self._ws = DeviceVector((need + x.dtype.itemsize - 1) // x.dtype.itemsize, x.dtype, x.ctx)
        check(h, lib.rls_prox_tv_fgp(h, x.code, len(shape), cs, len(d0), cd, x.ptr, float(lam), self.iterationsTV,
                                     self._ws.ptr, self._ws.n * self._ws.dtype.itemsize), "rls_prox_tv_fgp")
        return x

    def norm(self, x: DeviceVector, lam: float) -> float:
        g = GradientOp(self.shape, self.dims).mul(x)
        return float(lam) * g.norm1()


class NuclearRegularization(AbstractParameterizedRegularization):
    """src/proximalMaps/ProxNuclear.jl: singular-value soft-thresholding of reshape(x, svtShape)"""

    def __init__(self, lam, svtShape=(), **_kw):
        self.lam = float(lam)
        self.svtShape = tuple(int(s) for s in svtShape)

    def prox_(self, x: DeviceVector, lam: float):
        if len(self.svtShape) != 2 or self.svtShape[0] * self.svtShape[1] != x.n:
            raise ValueError(f"NuclearRegularization: svtShape {self.svtShape} does not match length(x)={x.n}")
        check(x.ctx.handle, x.ctx.lib.rls_prox_nuclear(x.ctx.handle, x.code, self.svtShape[0], self.svtShape[1], x.ptr,
                                                        float(lam)), "rls_prox_nuclear")
        return x

    def norm(self, x: DeviceVector, lam: float) -> float:
        """lambda * norm(S, 1): singular values on the host (setup / diagnostics only)"""
        import numpy as _np
        s = _np.linalg.svd(x.to_host().reshape(self.svtShape, order="F"), compute_uv=False)
        return float(lam) * float(s.sum())


class LLRRegularization(AbstractParameterizedRegularization):
    """src/proximalMaps/ProxLLR.jl: locally low rank regularisation, distinct blocks.  `randshift` draws the block-grid
    shift from a NumPy generator (the reference uses Julia's global RNG); fully overlapping blocks average the
    prox over all shifts of the block grid (:165-203)."""

    def __init__(self, lam, shape=(), blockSize=None, randshift: bool = True, fullyOverlapping: bool = False, L: int = 1,
                 seed=None, **_kw):
        import numpy as _np
        self.lam = float(lam)
        self.shape = tuple(int(s) for s in shape)
        self.blockSize = tuple(int(b) for b in (blockSize if blockSize is not None else (2,) * len(self.shape)))
        if len(self.blockSize) != len(self.shape) or not 1 <= len(self.shape) <= 3:
            raise ValueError("LLRRegularization: shape and blockSize must have the same length (1..3)")
        self.randshift, self.fullyOverlapping, self.L = bool(randshift), bool(fullyOverlapping), int(L)
        self._rng = _np.random.default_rng(seed)

    def _call(self, x: DeviceVector, lam: float, shift, shape=None):
        shape = shape or self.shape
        nd = len(shape)
        cs = (C.c_int64 * nd)(*shape)
        cb = (C.c_int64 * nd)(*self.blockSize)
        csh = (C.c_int64 * nd)(*shift)
        check(x.ctx.handle, x.ctx.lib.rls_prox_llr(x.ctx.handle, x.code, nd, cs, cb, csh, x.n, x.ptr, float(lam)), "rls_prox_llr")

    def prox_(self, x: DeviceVector, lam: float):
        import numpy as _np
        ns = 1
        for s_ in self.shape:
            ns *= s_
        if x.n % ns:
            raise ValueError(f"LLRRegularization: length(x)={x.n} is not a multiple of prod(shape)={ns}")
        if not self.fullyOverlapping:
            shift = [int(self._rng.integers(1, b + 1)) for b in self.blockSize] if self.randshift else [0] * len(self.shape)
            self._call(x, lam, shift)
            return x
        # proxLLROverlapping!: zero-pad to a multiple of the block size, prox for every shift of the block grid, average
        K = x.n // ns
        pad = [(-s_) % b for s_, b in zip(self.shape, self.blockSize)]
        pshape = tuple(s_ + p for s_, p in zip(self.shape, pad))
        xh = x.to_host().reshape(self.shape + (K,), order="F")
        xp = _np.zeros(pshape + (K,), dtype=xh.dtype)
        xp[tuple(slice(0, s_) for s_ in self.shape)] = xh
        acc = _np.zeros_like(xh)
        work = DeviceVector(xp.size, x.dtype, x.ctx)
        n_shift = 0
        for idx in _np.ndindex(*self.blockSize):
            shift = [i + 1 for i in idx]  # CartesianIndices(blockSize) is 1-based
            work.copy_from_host(xp.reshape(-1, order="F"))
            saved, self.randshift = self.randshift, False
            try:
                # the inner call of the reference runs with the regulariser's own randshift; with a shifted input
                # the shift composes -- here the explicit shift IS the circshift of :193
                self._call(work, lam, shift, shape=pshape)
            finally:
                self.randshift = saved
            acc += work.to_host().reshape(pshape + (K,), order="F")[tuple(slice(0, s_) for s_ in self.shape)]
            n_shift += 1
        x.copy_from_host((acc / n_shift).reshape(-1, order="F").astype(x.dtype))
        return x


class PositiveRegularization(AbstractProjectionRegularization):
    """src/proximalMaps/ProxPositive.jl"""

    def __init__(self, **_kw):
        pass

    def prox_(self, x: DeviceVector, lam=None):
        check(x.ctx.handle, x.ctx.lib.rls_prox_positive(x.ctx.handle, x.code, x.n, x.ptr), "rls_prox_positive")
        return x


class RealRegularization(AbstractProjectionRegularization):
    """src/proximalMaps/ProxReal.jl"""

    def __init__(self, **_kw):
        pass

    def prox_(self, x: DeviceVector, lam=None):
        check(x.ctx.handle, x.ctx.lib.rls_prox_real(x.ctx.handle, x.code, x.n, x.ptr), "rls_prox_real")
        return x


class GradientOp:
    """LinearOperatorCollection.GradientOp on device vectors (call sites ProxTV.jl:46,108-109,123;
    as an ADMM regTrafo: src/ADMM.jl:74)."""

    def __init__(self, shape, dims=None):
        self.shape, self.d0, self._cs, self._cd = _tv_geometry(shape, dims)
        self.n_in = 1
        for s in self.shape:
            self.n_in *= s
        from . import _lib
        self.n_out = int(_lib.load().rls_tv_grad_len(len(self.shape), self._cs, len(self.d0), self._cd))

    def size(self, i):
        return (self.n_out, self.n_in)[i - 1]

    def mul_(self, g: DeviceVector, x: DeviceVector, alpha=1.0, beta=0.0):
        lib, h = x.ctx.lib, x.ctx.handle
        check(h, lib.rls_tv_grad(h, x.code, len(self.shape), self._cs, len(self.d0), self._cd, x.ptr, g.ptr,
                                 float(alpha), float(beta)), "rls_tv_grad")
        return g

    def mul_adj_(self, x: DeviceVector, g: DeviceVector, alpha=1.0, beta=0.0):
        lib, h = x.ctx.lib, x.ctx.handle
        check(h, lib.rls_tv_grad_t(h, x.code, len(self.shape), self._cs, len(self.d0), self._cd, g.ptr, x.ptr,
                                   float(alpha), float(beta)), "rls_tv_grad_t")
        return x

    def mul(self, x: DeviceVector) -> DeviceVector:
        return self.mul_(DeviceVector(self.n_out, x.dtype, x.ctx), x)


# ---- normalisation plumbing (src/Regularization/NormalizedRegularization.jl:40-84) ------------


class AbstractRegularizationNormalization:
    pass


class NoNormalization(AbstractRegularizationNormalization):
    """lambda is used as given (:3-8)"""


class MeasurementBasedNormalization(AbstractRegularizationNormalization):
    """lambda is scaled by norm(b, 1) / length(b) (:9-14, 40-43)"""


class SystemMatrixBasedNormalization(AbstractRegularizationNormalization):
    """lambda is scaled by the mean energy of the system-matrix rows, sum_m rownorm²(A, m) / N (:15-20, 47-58)"""


def NormalizedRegularization(reg, factor):
    """NormalizedRegularization(reg, factor) (:29-38): lambda(reg) * factor.  Represented as a shallow copy of
    `reg` whose `lam` is the scaled value (the unscaled one is kept, so a later normalize() *updates* the
    factor as :73 does); type checks of the solvers keep seeing the inner regulariser."""
    import copy
    out = copy.copy(reg)
    base = getattr(reg, "_base_lam", reg.lam)
    out._base_lam = base
    out._factor = float(factor)
    out.lam = float(base) * float(factor)
    if getattr(reg, "lam_vector", None) is not None:  # Tikhonov matrix: the vector is what gets scaled
        base_vec = getattr(reg, "_base_lam_vector", reg.lam_vector)
        out._base_lam_vector = base_vec
        out.lam_vector = base_vec * type(base_vec[0])(factor)
    return out


def innerreg(reg):
    if hasattr(reg, "_base_lam"):
        import copy
        out = copy.copy(reg)
        out.lam = reg._base_lam
        del out._base_lam, out._factor
        return out
    return reg


def scalefactor(reg):
    return getattr(reg, "_factor", 1.0)


def normalization_factor(norm_scheme, A=None, b=None):
    """normalize(scheme, A, b) -> factor or None (:40-59)"""
    if norm_scheme is None or isinstance(norm_scheme, NoNormalization):
        return None
    if isinstance(norm_scheme, MeasurementBasedNormalization):
        if b is None:
            return 1.0
        return b.norm1() / b.n
    if isinstance(norm_scheme, SystemMatrixBasedNormalization):
        if A is None:
            raise ValueError("SystemMatrixBasedNormalization requires supplying A to the constructor of the solver")
        # energy[m] = sqrt(rownorm²(A, m)); trace = norm(energy)^2 / N   -- the sum of the squared row norms
        return A.rownorm2().norm1() / A.N
    raise TypeError(f"unknown normalization scheme {type(norm_scheme).__name__}")


def normalize(norm_scheme, regs, A=None, b=None, in_solver: bool = False):
    """normalize(scheme, regs, A, b) (:60-68) and, with in_solver=True, normalize(solver, scheme, regs, A, b)
    as init! calls it (:82-84: the system-matrix factor was already applied by the constructor)."""
    if in_solver and isinstance(norm_scheme, SystemMatrixBasedNormalization):
        return regs
    factor = normalization_factor(norm_scheme, A, b)
    single = not isinstance(regs, (list, tuple))
    out = []
    for r in ([regs] if single else regs):
        if factor is None or isinstance(r, AbstractProjectionRegularization) or not hasattr(r, "lam"):
            out.append(r)
        else:
            out.append(NormalizedRegularization(r, factor))
    return out[0] if single else out


# ---- generic entry points --------------------------------------------------------------------


def lam(reg):
    """lambda(reg)  (Regularization.jl:29; projections have none, :42)"""
    return getattr(reg, "lam", None)


def prox_(reg, x: DeviceVector, lam_: Optional[float] = None, **kw):
    if isinstance(reg, type):
        if issubclass(reg, AbstractProjectionRegularization):
            return reg(**kw).prox_(x)
        if lam_ is None:
            raise TypeError("prox_(RegType, x, lam): lam is required")
        return reg(lam_, **kw).prox_(x, lam_)
    if isinstance(reg, AbstractProjectionRegularization):
        return reg.prox_(x)
    return reg.prox_(x, reg.lam if lam_ is None else lam_)


def norm(reg, x: DeviceVector, lam_: Optional[float] = None, **kw):
    if isinstance(reg, type):
        return reg(lam_, **kw).norm(x, lam_)
    return reg.norm(x, reg.lam if lam_ is None else lam_)
