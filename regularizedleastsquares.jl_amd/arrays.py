"""Device array / operator types: the backend's answer to the reference's implicit array protocol
(layer L1; section 8b "methods the unchanged solvers call").

  Context       rls_ctx: one per GPU (device + stream + reduction workspace)
  DeviceVector  the vector type of b and of every solver state vector (`similar(b, n)`)
  DeviceMatrix  the operator type of A: column-major dense matrix; `A.H @ A`-style normal operators
                are lazy (`NormalOperator`) so the unchanged constructors land in matrix-free mode
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

from . import _lib
from ._lib import C32, C64, F32, F64, OP_C, OP_N, OP_T, RLSError, check

_DT = {np.dtype(np.float32): F32, np.dtype(np.complex64): C32, np.dtype(np.float64): F64, np.dtype(np.complex128): C64}
_NP = {F32: np.dtype(np.float32), C32: np.dtype(np.complex64), F64: np.dtype(np.float64), C64: np.dtype(np.complex128)}


def dtype_code(dt) -> int:
    dt = np.dtype(dt)
    if dt not in _DT:
        raise TypeError(f"the MI355X backend computes in Float32 / ComplexF32 (tuned path) and Float64 / ComplexF64 (L1 protocol); got {dt}")
    return _DT[dt]


def is_double(code: int) -> bool:
    """Float64 / ComplexF64: the rls_*_d entry points (the L1 protocol with double scalars); no fused plans, no resident kernels"""
    return code in (F64, C64)


class Context:
    """rls_ctx wrapper.  `stream=None` creates a private non-blocking stream; pass a hipStream_t
    handle (e.g. torch.cuda.current_stream().cuda_stream) to share one."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = _lib.load()
        h = C.c_void_p()
        if stream is None:
            st = self.lib.rls_ctx_create(device, C.byref(h))
        else:
            st = self.lib.rls_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(h))
        if st != 0 or not h:
            raise RLSError(f"rls_ctx_create(device={device}) failed with status {st}: no usable MI355X device? "
                           "(there is no CPU fallback)")
        self.handle = h
        self.device = device

    def sync(self):
        check(self.handle, self.lib.rls_ctx_sync(self.handle), "rls_ctx_sync")

    def tune(self, **kw):
        for k, v in kw.items():
            check(self.handle, self.lib.rls_tune_set(self.handle, k.encode(), int(v)), f"rls_tune_set({k})")

    @property
    def stream(self) -> int:
        return self.lib.rls_ctx_stream(self.handle) or 0

    def timer_start(self):
        check(self.handle, self.lib.rls_timer_start(self.handle), "rls_timer_start")

    def timer_stop_ms(self) -> float:
        ms = C.c_float()
        check(self.handle, self.lib.rls_timer_stop_ms(self.handle, C.byref(ms)), "rls_timer_stop_ms")
        return float(ms.value)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.rls_ctx_destroy(self.handle)
            self.handle = None


_default_ctx: Dict[int, Context] = {}


def default_context(device: int = 0) -> Context:
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class _DeviceBuffer:
    """owning device allocation (the Julia side would attach a finalizer calling rls_free)"""

    _GUARD = 2 << 20  # RLS_GUARD_ALLOC=1 (test runs): own 2 MiB-granular allocation, data at its END

    def __init__(self, ctx: Context, nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        if os.environ.get("RLS_GUARD_ALLOC") == "1":
            # out-of-bounds reads past the end of a vector / matrix then leave the allocation and fault instead of
            # silently reading a neighbour (how the slab_load overrun for N < 128 was found)
            used = (max(self.nbytes, 1) + 15) // 16 * 16
            total = (used + self._GUARD - 1) // self._GUARD * self._GUARD
            check(ctx.handle, ctx.lib.rls_malloc(ctx.handle, total, C.byref(p)), "rls_malloc")
            self._base = p.value
            self.ptr = p.value + total - used
        else:
            check(ctx.handle, ctx.lib.rls_malloc(ctx.handle, max(self.nbytes, 1), C.byref(p)), "rls_malloc")
            self._base = self.ptr = p.value

    def __del__(self):
        try:
            if self._base and self.ctx.handle:
                self.ctx.lib.rls_free(self.ctx.handle, C.c_void_p(self._base))
        except Exception:
            pass
        self.ptr = self._base = None


class _BorrowedBuffer:
    def __init__(self, ptr: int, keep=None):
        self.ptr, self._keep = int(ptr), keep


class DeviceVector:
    """Length-n Float32 / ComplexF32 vector in HBM."""

    def __init__(self, n: int, dtype, ctx: Optional[Context] = None, _buf=None, _offset=0):
        self.ctx = ctx or default_context()
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        self.code = dtype_code(self.dtype)
        self._buf = _buf or _DeviceBuffer(self.ctx, self.n * self.dtype.itemsize)
        self.ptr = self._buf.ptr + _offset

    # --- construction -----------------------------------------------------------------------
    @classmethod
    def from_host(cls, a, ctx: Optional[Context] = None) -> "DeviceVector":
        a = np.ascontiguousarray(a)
        if a.ndim != 1:
            raise ValueError("DeviceVector.from_host expects a 1-D array")
        v = cls(a.shape[0], a.dtype, ctx)
        v.copy_from_host(a)
        return v

    @classmethod
    def borrow(cls, ptr: int, n: int, dtype, ctx: Optional[Context] = None, keep=None) -> "DeviceVector":
        """view of device memory owned by someone else (e.g. a torch tensor that torch.distributed all-reduces in
        place); `keep` is held so the owner outlives the view"""
        return cls(n, dtype, ctx, _buf=_BorrowedBuffer(ptr, keep))

    def similar(self, n: Optional[int] = None) -> "DeviceVector":
        return DeviceVector(self.n if n is None else n, self.dtype, self.ctx)

    def copy(self) -> "DeviceVector":
        out = self.similar()
        out.copy_from(self)
        return out

    # --- transfers --------------------------------------------------------------------------
    def copy_from_host(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.size != self.n:
            raise ValueError(f"size mismatch: {a.size} vs {self.n}")
        lib, h = self.ctx.lib, self.ctx.handle
        check(h, lib.rls_memcpy_h2d(h, self.ptr, a.ctypes.data, a.nbytes), "rls_memcpy_h2d")

    def to_host(self) -> np.ndarray:
        out = np.empty(self.n, dtype=self.dtype)
        lib, h = self.ctx.lib, self.ctx.handle
        check(h, lib.rls_memcpy_d2h(h, out.ctypes.data, self.ptr, out.nbytes), "rls_memcpy_d2h")
        return out

    def copy_from(self, other: "DeviceVector"):
        if other.n != self.n or other.dtype != self.dtype:
            raise ValueError("copy_from: shape/dtype mismatch")
        lib, h = self.ctx.lib, self.ctx.handle
        check(h, lib.rls_memcpy_d2d(h, self.ptr, other.ptr, self.n * self.dtype.itemsize), "rls_memcpy_d2d")

    # --- BLAS-1 (names follow LinearAlgebra) ----------------------------------------------------
    # (Float64 / ComplexF64 vectors take the rls_*_d entry points: double scalars in, double results out)
    def fill_(self, value):
        value = complex(value)
        lib, h = self.ctx.lib, self.ctx.handle
        if is_double(self.code):
            check(h, lib.rls_fill_d(h, self.code, self.n, self.ptr, value.real, value.imag), "rls_fill_d")
        else:
            check(h, lib.rls_fill(h, self.code, self.n, self.ptr, value.real, value.imag), "rls_fill")
        return self

    def _reduce(self, name, *ptrs):
        lib, h = self.ctx.lib, self.ctx.handle
        if is_double(self.code):
            r = (C.c_double * 2)()
            check(h, getattr(lib, name + "_d")(h, self.code, self.n, *ptrs, r), name + "_d")
        else:
            r = (C.c_float * 2)()
            check(h, getattr(lib, name)(h, self.code, self.n, *ptrs, r), name)
        return r

    def _same(self, *others):
        """operands of one launch share the element type (two precisions coexist since round 6: a Float32 vector handed to a
        Float64 kernel would be read past its end)"""
        for o in others:
            if o.dtype != self.dtype:
                raise TypeError(f"element types differ: {self.dtype} and {o.dtype} (no mixed-precision broadcasts; convert on the host)")

    def norm(self) -> float:
        return float(self._reduce("rls_nrm2", self.ptr)[0])

    def norm1(self) -> float:
        return float(self._reduce("rls_asum", self.ptr)[0])

    def dot(self, other: "DeviceVector"):
        """dot(self, other) = conj(self) . other"""
        self._same(other)
        r = self._reduce("rls_dotc", self.ptr, other.ptr)
        return complex(r[0], r[1]) if self.code in (C32, C64) else float(r[0])

    def rmul_(self, a):
        a = complex(a)
        lib, h = self.ctx.lib, self.ctx.handle
        if is_double(self.code):
            check(h, lib.rls_scal_d(h, self.code, self.n, a.real, a.imag, self.ptr), "rls_scal_d")
        else:
            check(h, lib.rls_scal(h, self.code, self.n, a.real, a.imag, self.ptr), "rls_scal")
        return self

    def axpy_(self, a, x: "DeviceVector"):
        """self .+= a .* x"""
        self._same(x)
        a = complex(a)
        lib, h = self.ctx.lib, self.ctx.handle
        if is_double(self.code):
            check(h, lib.rls_axpy_d(h, self.code, self.n, a.real, a.imag, x.ptr, self.ptr), "rls_axpy_d")
        else:
            check(h, lib.rls_axpy(h, self.code, self.n, a.real, a.imag, x.ptr, self.ptr), "rls_axpy")
        return self

    def axpby_(self, a, x: "DeviceVector", b):
        """self = a x + b self"""
        return self.lincomb_(a, x, b, self)

    def lincomb_(self, a, x: "DeviceVector", b, y: "DeviceVector"):
        """self = a x + b y"""
        self._same(x, y)
        a, b = complex(a), complex(b)
        lib, h = self.ctx.lib, self.ctx.handle
        if is_double(self.code):
            check(h, lib.rls_lincomb_d(h, self.code, self.n, a.real, a.imag, x.ptr, b.real, b.imag, y.ptr, self.ptr), "rls_lincomb_d")
        else:
            check(h, lib.rls_lincomb(h, self.code, self.n, a.real, a.imag, x.ptr, b.real, b.imag, y.ptr, self.ptr),
                  "rls_lincomb")
        return self

    def stats(self):
        """(min, max, sum, sum of squares) of the real parts and max |x|, one device pass (rls_stats)"""
        out = (C.c_double * 5)()
        check(self.ctx.handle, self.ctx.lib.rls_stats(self.ctx.handle, self.code, self.n, self.ptr, out), "rls_stats")
        return [float(v) for v in out]

    def shift_scale_(self, shift, scale, inverse: bool = False):
        """Float32 only: x = (x - shift) / scale, or with inverse: x = x * scale + shift (src/Transforms.jl)"""
        if self.dtype != np.float32:
            raise TypeError("shift_scale_ acts on Float32 vectors")
        check(self.ctx.handle, self.ctx.lib.rls_shift_scale(self.ctx.handle, self.n, self.ptr, float(shift), float(scale),
                                                            1 if inverse else 0), "rls_shift_scale")
        return self

    def __len__(self):
        return self.n


class DeviceMatrix:
    """Dense M x N operator in HBM, column-major with leading dimension lda (Julia `Matrix`)."""

    def __init__(self, M: int, N: int, dtype, ctx: Optional[Context] = None, lda: Optional[int] = None):
        self.ctx = ctx or default_context()
        self.M, self.N = int(M), int(N)
        self.lda = int(lda if lda is not None else max(self.M, 1))
        self.dtype = np.dtype(dtype)
        self.code = dtype_code(self.dtype)
        self._buf = _DeviceBuffer(self.ctx, self.lda * self.N * self.dtype.itemsize)
        self.ptr = self._buf.ptr
        self._op = None

    @classmethod
    def from_host(cls, A, ctx: Optional[Context] = None) -> "DeviceMatrix":
        A = np.asarray(A)
        if A.ndim != 2:
            raise ValueError("DeviceMatrix.from_host expects a 2-D array")
        Af = np.asfortranarray(A)
        m = cls(A.shape[0], A.shape[1], Af.dtype, ctx)
        lib, h = m.ctx.lib, m.ctx.handle
        check(h, lib.rls_memcpy_h2d(h, m.ptr, Af.ctypes.data, Af.nbytes), "rls_memcpy_h2d")
        return m

    def to_host(self) -> np.ndarray:
        out = np.empty((self.lda, self.N), dtype=self.dtype, order="F")
        lib, h = self.ctx.lib, self.ctx.handle
        check(h, lib.rls_memcpy_d2h(h, out.ctypes.data, self.ptr, out.nbytes), "rls_memcpy_d2h")
        return out[: self.M, :]

    @property
    def shape(self):
        return (self.M, self.N)

    def size(self, i: int) -> int:  # 1-based like Julia's size(A, i)
        return (self.M, self.N)[i - 1]

    def fill_(self, value):
        """A .= value (every stored element, padding rows of lda included)"""
        value = complex(value)
        lib, h = self.ctx.lib, self.ctx.handle
        if is_double(self.code):
            check(h, lib.rls_fill_d(h, self.code, self.lda * self.N, self.ptr, value.real, value.imag), "rls_fill_d")
        else:
            check(h, lib.rls_fill(h, dtype_code(self.dtype), self.lda * self.N, self.ptr, value.real, value.imag), "rls_fill")
        return self

    def column(self, j: int) -> DeviceVector:
        """b[:, j] (0-based j) as a fresh vector: the reference copies columns (src/MultiThreading.jl:35)"""
        v = DeviceVector(self.M, self.dtype, self.ctx)
        lib, h = self.ctx.lib, self.ctx.handle
        off = j * self.lda * self.dtype.itemsize
        check(h, lib.rls_memcpy_d2d(h, v.ptr, self.ptr + off, self.M * self.dtype.itemsize), "rls_memcpy_d2d")
        return v

    def column_view(self, j: int) -> DeviceVector:
        """view(b, :, j): shares the matrix's memory (in-place operations on one column)"""
        off = (self.ptr - self._buf.ptr) + j * self.lda * self.dtype.itemsize
        return DeviceVector(self.M, self.dtype, self.ctx, _buf=self._buf, _offset=off)

    # --- mul! -------------------------------------------------------------------------------
    def gemv_(self, op: int, x: DeviceVector, y: DeviceVector, alpha=1.0, beta=0.0):
        """y = alpha * op(A) * x + beta * y     (5-arg mul!)"""
        alpha, beta = complex(alpha), complex(beta)
        nx, ny = (self.N, self.M) if op == OP_N else (self.M, self.N)
        if x.n != nx or y.n != ny:
            raise ValueError(f"gemv: dimension mismatch: A is {self.M}x{self.N}, x {x.n}, y {y.n}, op {op}")
        if x.dtype != self.dtype or y.dtype != self.dtype:
            raise TypeError(f"gemv: element types differ: A {self.dtype}, x {x.dtype}, y {y.dtype}")
        lib, h = self.ctx.lib, self.ctx.handle
        if is_double(self.code):
            check(h, lib.rls_gemv_d(h, self.code, op, self.M, self.N, alpha.real, alpha.imag, self.ptr, self.lda, x.ptr,
                                    beta.real, beta.imag, y.ptr), "rls_gemv_d")
        else:
            check(h, lib.rls_gemv(h, self.code, op, self.M, self.N, alpha.real, alpha.imag, self.ptr, self.lda, x.ptr,
                                  beta.real, beta.imag, y.ptr), "rls_gemv")
        return y

    def mul_(self, y: DeviceVector, x: DeviceVector, alpha=1.0, beta=0.0):
        return self.gemv_(OP_N, x, y, alpha, beta)

    def mul_adj_(self, x: DeviceVector, y: DeviceVector, alpha=1.0, beta=0.0):
        return self.gemv_(OP_C, y, x, alpha, beta)

    def mul_transpose_(self, x: DeviceVector, y: DeviceVector, alpha=1.0, beta=0.0):
        return self.gemv_(OP_T, y, x, alpha, beta)

    def __matmul__(self, x: DeviceVector) -> DeviceVector:  # allocating A * x  (src/ADMM.jl:203)
        return self.mul_(DeviceVector(self.M, self.dtype, self.ctx), x)

    def gram(self) -> "DeviceMatrix":
        """explicit AHA = A' * A on device (src/CGNR.jl:49); pass as AHA= for Gram mode"""
        G = DeviceMatrix(self.N, self.N, self.dtype, self.ctx)
        lib, h = self.ctx.lib, self.ctx.handle
        check(h, lib.rls_gram(h, self.code, self.M, self.N, self.ptr, self.lda, G.ptr, G.lda), "rls_gram")
        return G

    def rownorm2(self) -> "DeviceVector":
        """rownorm²(A, m) for every row m (src/Utils.jl:20-23; GPU ext NormalizedRegularization.jl:1-5): real vector"""
        lib, h = self.ctx.lib, self.ctx.handle
        if is_double(self.code):
            out = DeviceVector(self.M, np.float64, self.ctx)
            check(h, lib.rls_rownorm2_d(h, self.code, self.M, self.N, self.ptr, self.lda, out.ptr), "rls_rownorm2_d")
            return out
        out = DeviceVector(self.M, np.float32, self.ctx)
        check(h, lib.rls_rownorm2(h, self.code, self.M, self.N, self.ptr, self.lda, out.ptr), "rls_rownorm2")
        return out

    def scale_rows(self, w: "DeviceVector") -> "DeviceMatrix":
        """diag(w) * A as a new dense matrix (ProdOp(WeightingOp(w), A) materialised)"""
        if w.n != self.M or w.dtype != self.dtype:
            raise ValueError("scale_rows: weights must have length size(A, 1) and the element type of A")
        B = DeviceMatrix(self.M, self.N, self.dtype, self.ctx)
        lib, h = self.ctx.lib, self.ctx.handle
        fn = lib.rls_scale_rows_d if is_double(self.code) else lib.rls_scale_rows
        check(h, fn(h, self.code, self.M, self.N, w.ptr, self.ptr, self.lda, B.ptr, B.lda), "rls_scale_rows")
        return B

    def normal_operator(self) -> "NormalOperator":
        """A' * A as the unchanged constructors evaluate it: lazy, matrix-free (two GEMVs per apply)"""
        return NormalOperator(self)


class WeightingOp:
    """LinearOperatorCollection.WeightingOp(weights): diag(weights) (docs/src/literate/howto/normal_operator.jl:41)"""

    def __init__(self, weights: "DeviceVector"):
        self.weights = weights


def ProdOp(W: WeightingOp, A: DeviceMatrix) -> DeviceMatrix:
    """ProdOp(WeightingOp(w), A) (docs/src/literate/howto/normal_operator.jl:42, src/Utils.jl:23,102).  On a GPU with
    288 GB the weighted operator is materialised once as diag(w) A: A, its adjoint and the normal operator
    A^H W^H W A (`normalOperator`) then run on the same one-pass / matrix-core kernels as any dense matrix,
    instead of carrying a diagonal scale through every kernel."""
    if not isinstance(W, WeightingOp):
        raise TypeError("ProdOp: only ProdOp(WeightingOp(w), A) is supported")
    w = W.weights
    if w.dtype != A.dtype:
        w = DeviceVector.from_host(w.to_host().astype(A.dtype), A.ctx)
    WA = A.scale_rows(w)
    WA.weights, WA.B = W.weights, A
    return WA


def normalOperator(A: DeviceMatrix) -> "NormalOperator":
    """LinearOperatorCollection.normalOperator(A) (docs/src/literate/howto/normal_operator.jl:43)"""
    return A.normal_operator()


class OperatorHandle:
    """rls_operator: forward matrix and/or Gram matrix bound to a context."""

    def __init__(self, A: Optional[DeviceMatrix], gram: Optional[DeviceMatrix] = None):
        src = A if A is not None else gram
        if src is None:
            raise ValueError("OperatorHandle needs A or a Gram matrix")
        self.ctx = src.ctx
        self.A, self.gram = A, gram
        self.dtype, self.code = src.dtype, src.code
        self.M = A.M if A is not None else 0
        self.N = A.N if A is not None else gram.N
        lib, h = self.ctx.lib, self.ctx.handle
        if gram is not None and (gram.M != self.N or gram.N != self.N):
            raise ValueError("Gram matrix must be N x N")
        self.double = is_double(self.code)
        self._t = None
        if self.double:
            # Float64 / ComplexF64: no rls_operator (the fused plans are Float32 / ComplexF32): v = AHA p is the explicit Gram GEMV or
            # the two GEMVs of the matrix-free normal operator, through rls_gemv_d
            self.handle = None
            return
        o = C.c_void_p()
        check(h, lib.rls_operator_create(h, self.code, self.M, self.N, A.ptr if A is not None else None,
                                         A.lda if A is not None else 0, C.byref(o)), "rls_operator_create")
        self.handle = o
        if gram is not None:
            check(h, lib.rls_operator_set_gram(o, gram.ptr, gram.lda), "rls_operator_set_gram")

    def mul_normal_(self, v: DeviceVector, p: DeviceVector):
        if self.double:
            if self.gram is not None:
                return self.gram.mul_(v, p)
            if self._t is None:
                self._t = DeviceVector(self.M, self.dtype, self.ctx)
            self.A.mul_(self._t, p)
            return self.A.mul_adj_(v, self._t)
        check(self.ctx.handle, self.ctx.lib.rls_operator_mul_normal(self.handle, p.ptr, v.ptr), "rls_operator_mul_normal")
        return v

    def __del__(self):
        try:
            if self.handle and self.ctx.handle:
                self.ctx.lib.rls_operator_destroy(self.handle)
        except Exception:
            pass
        self.handle = None


class NormalOperator:
    """Lazy A^H A (LinearOperatorCollection.normalOperator): size N x N, eltype of A."""

    def __init__(self, A: DeviceMatrix):
        self.A = A
        self.dtype = A.dtype
        self.ctx = A.ctx
        self.shape = (A.N, A.N)
        self._handle = None

    def size(self, i: int) -> int:
        return self.A.N

    def handle(self) -> OperatorHandle:
        if self._handle is None:
            self._handle = OperatorHandle(self.A)
        return self._handle

    def mul_(self, v: DeviceVector, p: DeviceVector):
        return self.handle().mul_normal_(v, p)
