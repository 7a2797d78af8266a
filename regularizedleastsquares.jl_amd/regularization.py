"""Regularisation types and prox!/norm dispatch, mirroring the reference's type tree
(src/Regularization/Regularization.jl:1-57, src/proximalMaps/*.jl) for device vectors.

Julia's `prox!(reg, x, lambda)` is `prox_(reg, x, lam)` here (trailing underscore = in place).
Both call forms of the reference are kept:
    prox_(reg_instance, x)            -> uses lambda(reg)         Regularization.jl:17
    prox_(RegType, x, lam, **kw)      -> constructs RegType(lam; kw...) first   Regularization.jl:39,55
"""
from __future__ import annotations

import ctypes as C

import numpy as np
from typing import Optional, Sequence, Tuple

from ._lib import check
from .arrays import DeviceVector


def _dbl(x) -> bool:
    """Float64 / ComplexF64 vector: the rls_*_d entry points (double scalars)"""
    return x.dtype in (np.dtype(np.float64), np.dtype(np.complex128))


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
        if _dbl(x):
            check(x.ctx.handle, x.ctx.lib.rls_prox_l1_d(x.ctx.handle, x.code, x.n, x.ptr, float(lam)), "rls_prox_l1_d")
        else:
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
            given = _np.asarray(lam)   # (a Float64 vector keeps its precision for a Float64 operator; the solver casts to its own real type)
            self.lam_vector = given.astype(_np.float64 if given.dtype == _np.float64 else _np.float32)
            self.lam = float("nan")
        else:
            self.lam_vector = None
            self.lam = float(lam)

    def prox_(self, x: DeviceVector, lam: float):
        if _dbl(x):
            check(x.ctx.handle, x.ctx.lib.rls_prox_l2_d(x.ctx.handle, x.code, x.n, x.ptr, float(lam)), "rls_prox_l2_d")
        else:
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
        if _dbl(x):
            check(x.ctx.handle, x.ctx.lib.rls_prox_l21_d(x.ctx.handle, x.code, x.n, self.slices, x.ptr, float(lam)), "rls_prox_l21_d")
        else:
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
        if _dbl(x):   # Float64 / ComplexF64: the generic FGP sequence with double scalars (its workspace comes from the context's pool)
            check(h, lib.rls_prox_tv_fgp_d(h, x.code, len(shape), cs, len(d0), cd, x.ptr, float(lam), self.iterationsTV), "rls_prox_tv_fgp_d")
            return x
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
        if _dbl(x):
            check(x.ctx.handle, x.ctx.lib.rls_prox_positive_d(x.ctx.handle, x.code, x.n, x.ptr), "rls_prox_positive_d")
        else:
            check(x.ctx.handle, x.ctx.lib.rls_prox_positive(x.ctx.handle, x.code, x.n, x.ptr), "rls_prox_positive")
        return x


class RealRegularization(AbstractProjectionRegularization):
    """src/proximalMaps/ProxReal.jl"""

    def __init__(self, **_kw):
        pass

    def prox_(self, x: DeviceVector, lam=None):
        if _dbl(x):
            check(x.ctx.handle, x.ctx.lib.rls_prox_real_d(x.ctx.handle, x.code, x.n, x.ptr), "rls_prox_real_d")
        else:
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


# ---- nested regularisation terms (src/Regularization/NestedRegularization.jl, ScaledRegularization.jl,
# MaskedRegularization.jl, TransformedRegularization.jl), ProjectionRegularization (src/proximalMaps/ProxProj.jl)


def collect(reg):
    """iterate(reg) (Regularization.jl:6): the chain of terms, outermost first"""
    out = []
    while reg is not None:
        out.append(reg)
        reg = reg.reg if isinstance(reg, AbstractNestedRegularization) else None
    return out


def sink(reg):
    """the innermost term (NestedRegularization.jl:15; a plain term is its own sink, Regularization.jl:8)"""
    return collect(reg)[-1]


def sinktype(reg):
    return type(sink(reg))


def is_projection(reg) -> bool:
    """sinktype(reg) <: AbstractProjectionRegularization -- how the solvers sort their `reg` argument
    (findsinks, Regularization.jl:86; call sites src/FISTA.jl:80, src/ADMM.jl:102, src/Kaczmarz.jl:98)"""
    return isinstance(sink(reg), AbstractProjectionRegularization)


def findsinks(T, regs):
    """indices of the terms whose sink is a T (Regularization.jl:86)"""
    return [i for i, r in enumerate(regs) if isinstance(sink(r), T)]


def findsink(T, regs):
    """index of THE term whose sink is a T, None if there is none, an error if ambiguous (Regularization.jl:75-84)"""
    idx = findsinks(T, regs)
    if not idx:
        return None
    if len(idx) > 1:
        raise ValueError(f"Cannot unambigiously retrieve reg term of type {T.__name__}, found {len(idx)} instances")
    return idx[0]


def findfirst(T, reg):
    """first term of the chain that is a T (Regularization.jl:69-73)"""
    for r in collect(reg):
        if isinstance(r, T):
            return r
    return None


class AbstractNestedRegularization(AbstractRegularization):
    """prox!/norm/lambda forward to the inner term (NestedRegularization.jl:23-28)"""

    def __init__(self, reg):
        self.reg = reg

    @property
    def lam(self):
        return lam(self.reg)

    def prox_(self, x: DeviceVector, *args):
        return self.reg.prox_(x, *args)

    def norm(self, x: DeviceVector, *args):
        return self.reg.norm(x, *args)


class MaskedRegularization(AbstractNestedRegularization):
    """MaskedRegularization.jl:19-37: prox!/norm only see the elements of x whose mask entry is true
    (rls_gather -> inner prox -> rls_scatter; the index list lives on the device)"""

    def __init__(self, reg, mask):
        import numpy as _np
        super().__init__(reg)
        self.mask = _np.asarray(mask, dtype=bool)
        self._idx_h = _np.flatnonzero(self.mask).astype(_np.int32)
        self._idx = None

    def _gather(self, x: DeviceVector):
        import numpy as _np
        if x.n != self.mask.size:
            raise ValueError(f"MaskedRegularization: mask has {self.mask.size} entries, x has {x.n}")
        if self._idx is None or self._idx.ctx is not x.ctx:
            self._idx = DeviceVector.from_host(self._idx_h.view(_np.float32), x.ctx)  # int32 payload in a 4-byte vector
        z = DeviceVector(self._idx_h.size, x.dtype, x.ctx)
        check(x.ctx.handle, x.ctx.lib.rls_gather(x.ctx.handle, x.code, z.n, self._idx.ptr, x.ptr, z.ptr), "rls_gather")
        return z

    def prox_(self, x: DeviceVector, *args):
        z = self._gather(x)
        self.reg.prox_(z, *args)
        check(x.ctx.handle, x.ctx.lib.rls_scatter(x.ctx.handle, x.code, z.n, self._idx.ptr, z.ptr, x.ptr), "rls_scatter")
        return x

    def norm(self, x: DeviceVector, *args):
        return self.reg.norm(self._gather(x), *args)


class TransformedRegularization(AbstractNestedRegularization):
    """TransformedRegularization.jl:19-37: z = trafo * x ; prox!(reg, z) ; x = adjoint(trafo) * z.  `trafo` is a
    DeviceMatrix, a GradientOp or any object with mul_(z, x) / mul_adj_(x, z) and size(1)."""

    def __init__(self, reg, trafo):
        super().__init__(reg)
        self.trafo = trafo

    def _forward(self, x: DeviceVector):
        t = self.trafo
        n_out = t.size(1) if hasattr(t, "size") else t.n_out
        return t.mul_(DeviceVector(n_out, x.dtype, x.ctx), x)

    def prox_(self, x: DeviceVector, *args):
        z = self._forward(x)
        self.reg.prox_(z, *args)
        self.trafo.mul_adj_(x, z)
        return x

    def norm(self, x: DeviceVector, *args):
        return self.reg.norm(self._forward(x), *args)


class AbstractScaledRegularization(AbstractNestedRegularization):
    """lambda(reg) = lambda(innerreg(reg)) .* scalefactor(reg)   (ScaledRegularization.jl:9-23)"""

    def scalefactor(self):
        raise NotImplementedError(f"Scaled regularization term {type(self).__name__} must implement scalefactor")

    @property
    def lam(self):
        return lam(self.reg) * self.scalefactor()


class FixedScaledRegularization(AbstractScaledRegularization):
    """ScaledRegularization.jl:27-35"""

    def __init__(self, reg, factor):
        super().__init__(reg)
        self.factor = float(factor)

    def scalefactor(self):
        return self.factor


class _NormalizedNested(FixedScaledRegularization):
    """NormalizedRegularization around a nested term (NormalizedRegularization.jl:29-38,75-79)"""


class FixedParameterRegularization(AbstractScaledRegularization):
    """ScaledRegularization.jl:43-52: discards any lambda passed to it and uses the inner term's"""

    def scalefactor(self):
        return 1.0

    def prox_(self, x: DeviceVector, *_discard):
        return self.reg.prox_(x, lam(self.reg))

    def norm(self, x: DeviceVector, *_discard):
        return self.reg.norm(x, lam(self.reg))


class AutoScaledRegularization(AbstractScaledRegularization):
    """ScaledRegularization.jl:55-77: the first prox!/norm fixes the factor to maximum(abs.(x)) (rls_stats)"""

    def __init__(self, reg):
        super().__init__(reg)
        self.factor = None

    def scalefactor(self):
        return 1.0 if self.factor is None else self.factor

    def _first(self, x: DeviceVector, lam_):
        if self.factor is None:
            import numpy as _np
            self.factor = float(_np.float32(x.stats()[4]))
            return lam_ * self.factor
        return lam_

    def prox_(self, x: DeviceVector, lam_):
        return self.reg.prox_(x, self._first(x, lam_))

    def norm(self, x: DeviceVector, lam_):
        return self.reg.norm(x, self._first(x, lam_))


class ProjectionRegularization(AbstractProjectionRegularization):
    """src/proximalMaps/ProxProj.jl:3-20.  projFunc maps a DeviceVector to a DeviceVector (a new one or x itself)."""

    def __init__(self, projFunc=None, **_kw):
        self.projFunc = projFunc if projFunc is not None else (lambda x: x)

    def prox_(self, x: DeviceVector, lam=None):
        y = self.projFunc(x)
        if y is not x:
            x.copy_from(y)
        return x

    def norm(self, x: DeviceVector, lam=None) -> float:
        y = x.copy()
        self.prox_(y)
        y.axpy_(-1.0, x)
        return float("inf") if y.norm() != 0 else 0.0


# ---- plug-and-play prior and its input transforms (src/Regularization/PlugAndPlayRegularization.jl, src/Transforms.jl)


class MinMaxTransform:
    """src/Transforms.jl:4-16 on a Float32 device vector, in place"""

    def __init__(self, x: DeviceVector):
        st = x.stats()
        self.min, self.max = float(st[0]), float(st[1])

    def transform(self, x: DeviceVector):
        return x.shift_scale_(self.min, self.max - self.min, inverse=False)

    def inverse_transform(self, x: DeviceVector):
        return x.shift_scale_(self.min, self.max - self.min, inverse=True)


class IdentityTransform:
    """src/Transforms.jl:20-31"""

    def __init__(self, x=None):
        pass

    def transform(self, x):
        return x

    def inverse_transform(self, x):
        return x


class ZTransform:
    """src/Transforms.jl:34-46; std = the corrected sample standard deviation (Statistics.std)"""

    def __init__(self, x: DeviceVector):
        import math
        st, n = x.stats(), x.n
        self.mean = st[2] / n
        self.std = math.sqrt(max(st[3] - n * self.mean * self.mean, 0.0) / (n - 1)) if n > 1 else float("nan")

    def transform(self, x: DeviceVector):
        return x.shift_scale_(self.mean, self.std, inverse=False)

    def inverse_transform(self, x: DeviceVector):
        return x.shift_scale_(self.mean, self.std, inverse=True)


class ClampedScalingTransform:
    """src/Transforms.jl:49-68"""

    def __init__(self, x: DeviceVector, v_min, v_max):
        self.v_min, self.v_max = float(v_min), float(v_max)
        self.x = x.copy()

    def transform(self, x: DeviceVector):
        check(x.ctx.handle, x.ctx.lib.rls_clamp(x.ctx.handle, x.n, x.ptr, self.v_min, self.v_max), "rls_clamp")
        return x.shift_scale_(self.v_min, self.v_max - self.v_min, inverse=False)

    def inverse_transform(self, x: DeviceVector):
        x.shift_scale_(self.v_min, self.v_max - self.v_min, inverse=True)
        check(x.ctx.handle, x.ctx.lib.rls_restore_outside(x.ctx.handle, x.n, x.ptr, self.x.ptr, self.v_min, self.v_max),
              "rls_restore_outside")
        return x


class PlugAndPlayRegularization(AbstractParameterizedRegularization):
    """src/Regularization/PlugAndPlayRegularization.jl:14-54.  `model` maps a real Float32 DeviceVector (attribute
    `.shape` = the image shape) to a DeviceVector of the same length -- the learned prior is the user's code; the
    input transform, the blend x - lambda (x - model(x)) and the real/imaginary split run on the device.
    PlugAndPlayRegularization(model, shape; ...) is the reduced constructor (:22) with lambda = 1."""

    def __init__(self, lam=1.0, *args, model=None, shape=None, input_transform=MinMaxTransform, ignoreIm: bool = False,
                 **_kw):
        if callable(lam):  # (model, shape; kwargs...)
            model, shape, lam = lam, (args[0] if args else shape), 1.0
        if model is None or shape is None:
            raise TypeError("PlugAndPlayRegularization needs model and shape")
        self.lam = float(lam)
        self.model = model
        self.shape = [int(s) for s in shape]
        self.input_transform = input_transform
        self.ignoreIm = bool(ignoreIm)

    def prox_(self, x: DeviceVector, lam_):
        import numpy as _np
        import warnings
        lib, h = x.ctx.lib, x.ctx.handle
        if x.dtype.kind == "c":  # :24-31
            re, im = DeviceVector(x.n, _np.float32, x.ctx), DeviceVector(x.n, _np.float32, x.ctx)
            check(h, lib.rls_complex_split(h, x.n, x.ptr, re.ptr, im.ptr), "rls_complex_split")
            self.prox_(re, lam_)
            if not self.ignoreIm:
                self.prox_(im, lam_)
            check(h, lib.rls_complex_merge(h, x.n, re.ptr, im.ptr, x.ptr), "rls_complex_merge")
            return x
        lam_ = float(lam_)
        if lam_ != self.lam and (lam_ < 0.0 or lam_ > 1.0):  # :34-38
            temp = min(max(lam_, 0.0), 1.0)
            warnings.warn(f"{type(self).__name__} was given λ with value {lam_}. Valid range is [0, 1]. λ changed to temp")
            lam_ = temp
        out = x.copy()
        out.shape = tuple(self.shape)
        tf = self.input_transform(out)
        out = tf.transform(out)
        out.shape = tuple(self.shape)
        m = self.model(out)
        diff = out.copy().axpy_(-1.0, m)  # out - model(out)
        out.axpy_(-lam_, diff)            # out - lambda * (out - model(out))   :43
        out = tf.inverse_transform(out)
        x.copy_from(out)
        return x

    def norm(self, x, lam_=None):
        raise NotImplementedError("the plug-and-play prior is defined by its proximal map only (no norm), as in the reference")


PnPRegularization = PlugAndPlayRegularization


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
    if isinstance(reg, AbstractNestedRegularization):
        inner = reg.reg if isinstance(reg, _NormalizedNested) else reg  # :73 -- update, do not stack
        return _NormalizedNested(inner, factor)
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
    if isinstance(reg, AbstractNestedRegularization):
        return reg.reg
    if hasattr(reg, "_base_lam"):
        import copy
        out = copy.copy(reg)
        out.lam = reg._base_lam
        del out._base_lam, out._factor
        return out
    return reg


def scalefactor(reg):
    if isinstance(reg, AbstractScaledRegularization):
        return reg.scalefactor()
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
        if factor is None or not isinstance(sink(r), AbstractParameterizedRegularization):  # :70-79
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
    if is_projection(reg):
        return reg.prox_(x)
    return reg.prox_(x, reg.lam if lam_ is None else lam_)


def norm(reg, x: DeviceVector, lam_: Optional[float] = None, **kw):
    if isinstance(reg, type):
        if issubclass(reg, AbstractProjectionRegularization):
            return reg(**kw).norm(x)
        return reg(lam_, **kw).norm(x, lam_)
    if is_projection(reg):
        return reg.norm(x)
    return reg.norm(x, reg.lam if lam_ is None else lam_)
