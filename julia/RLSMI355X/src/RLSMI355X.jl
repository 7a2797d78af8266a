# RLSMI355X.jl -- thin `ccall` binding of librls_mi355x.so (C ABI: include/rls_mi355x.h) and the
# device array / operator types that select the MI355X backend by dispatch, the way the reference's
# GPU extensions do (ext/RegularizedLeastSquaresGPUArraysExt/*.jl, loading mechanism Project.toml:20-27).
#
# This file cannot be executed in the build container (no Julia); it is kept in lock-step with the
# Python harness (regularizedleastsquares.jl_amd/*.py) and the plain-C driver tests/abi_smoke.c, which issue
# exactly the same ABI call sequences and are what the parity tests run.
module RLSMI355X

using LinearAlgebra, Libdl

const librls = Ref{String}(get(ENV, "RLS_MI355X_LIB", "librls_mi355x.so"))

const RLS_F32 = Int32(0); const RLS_C32 = Int32(1)
const RLS_F64 = Int32(2); const RLS_C64 = Int32(3)   # the rls_*_d entry points only: the L1 protocol with double scalars
"element types of the tuned path (fused plans, resident kernels, matrix cores) / of the double-precision L1 protocol"
const RLSSingle = Union{Float32, ComplexF32}
const RLSDouble = Union{Float64, ComplexF64}
const RLS_OP_N = Int32(0); const RLS_OP_T = Int32(1); const RLS_OP_C = Int32(2)

struct RLSError <: Exception
  code::Int32
  msg::String
end

mutable struct Context
  handle::Ptr{Cvoid}
  device::Int32
  owner::Any  # a borrowed context keeps its communicator reachable: arrays allocated on it hold the context, hence the Comm
  function Context(device::Integer = 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    st = ccall((:rls_ctx_create, librls[]), Int32, (Int32, Ref{Ptr{Cvoid}}), device, h)
    st == 0 || throw(RLSError(st, "rls_ctx_create failed: no usable MI355X (there is no CPU fallback)"))
    ctx = new(h[], Int32(device), nothing)
    finalizer(c -> ccall((:rls_ctx_destroy, librls[]), Int32, (Ptr{Cvoid},), c.handle), ctx)
  end
  # a context owned by a communicator (rls_comm_ctx): wrapped, never destroyed from here
  Context(handle::Ptr{Cvoid}, device::Integer, ::Val{:borrowed}) = new(handle, Int32(device), nothing)
end

function check(ctx::Context, st::Int32, what)
  st == 0 && return nothing
  msg = unsafe_string(ccall((:rls_last_error_string, librls[]), Cstring, (Ptr{Cvoid},), ctx.handle))
  throw(RLSError(st, "$what: $msg"))
end

const default_ctx = Ref{Union{Nothing,Context}}(nothing)
context() = something(default_ctx[], (default_ctx[] = Context(0)))

dtypecode(::Type{Float32}) = RLS_F32
dtypecode(::Type{ComplexF32}) = RLS_C32
dtypecode(::Type{Float64}) = RLS_F64
dtypecode(::Type{ComplexF64}) = RLS_C64
dtypecode(T) = throw(ArgumentError("the MI355X backend computes in Float32 / ComplexF32 (tuned path) and Float64 / ComplexF64 (L1 protocol); got $T"))

# ---- device vector: the array type of b and of every solver state vector ------------------------
mutable struct RLSVector{T} <: AbstractVector{T}
  ptr::Ptr{Cvoid}
  n::Int
  ctx::Context
  function RLSVector{T}(::UndefInitializer, n::Integer; ctx = context()) where {T}
    dtypecode(T)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, ccall((:rls_malloc, librls[]), Int32, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), ctx.handle, max(n, 1) * sizeof(T), p), "rls_malloc")
    v = new{T}(p[], n, ctx)
    finalizer(x -> ccall((:rls_free, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), x.ctx.handle, x.ptr), v)
  end
  # a NON-owning vector over memory that something else owns and keeps alive (a column of a device matrix)
  RLSVector{T}(ptr::Ptr{Cvoid}, n::Integer, ctx::Context, ::Val{:view}) where {T} = new{T}(ptr, n, ctx)
end
Base.size(v::RLSVector) = (v.n,)
Base.similar(v::RLSVector{T}, ::Type{S}, dims::Dims{1}) where {T,S} = RLSVector{S}(undef, dims[1]; ctx = v.ctx)
Base.similar(v::RLSVector{T}, dims::Dims{1}) where {T} = RLSVector{T}(undef, dims[1]; ctx = v.ctx)
Base.getindex(::RLSVector, ::Int) = error("scalar indexing of a device vector is disabled; use Array(v)")

function RLSVector(a::Vector{T}; ctx = context()) where {T}
  v = RLSVector{T}(undef, length(a); ctx)
  check(ctx, ccall((:rls_memcpy_h2d, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{T}, Csize_t), ctx.handle, v.ptr, a, sizeof(a)), "rls_memcpy_h2d")
  v
end
function Base.Array(v::RLSVector{T}) where {T}
  a = Vector{T}(undef, v.n)
  check(v.ctx, ccall((:rls_memcpy_d2h, librls[]), Int32, (Ptr{Cvoid}, Ptr{T}, Ptr{Cvoid}, Csize_t), v.ctx.handle, a, v.ptr, sizeof(a)), "rls_memcpy_d2h")
  a
end
function Base.copyto!(dst::RLSVector{T}, src::RLSVector{T}) where {T}
  check(dst.ctx, ccall((:rls_memcpy_d2d, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), dst.ctx.handle, dst.ptr, src.ptr, dst.n * sizeof(T)), "rls_memcpy_d2d")
  dst
end
function Base.fill!(v::RLSVector{T}, c) where {T}
  z = ComplexF32(c)
  check(v.ctx, ccall((:rls_fill, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Float32, Float32), v.ctx.handle, dtypecode(T), v.n, v.ptr, real(z), imag(z)), "rls_fill")
  v
end

# BLAS-1 of the hot path: norm, dot (conjugating), rmul!, axpy-style broadcasts
function LinearAlgebra.norm(v::RLSVector{T}) where {T}
  r = Ref{NTuple{2,Float32}}((0f0, 0f0))
  check(v.ctx, ccall((:rls_nrm2, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}), v.ctx.handle, dtypecode(T), v.n, v.ptr, r), "rls_nrm2")
  r[][1]
end
"norm(v, p) for p = 2 and p = 1 (`norm(b, 1) / length(b)`: MeasurementBasedNormalization, src/Regularization/NormalizedRegularization.jl:40-42;
complex modulus as in ProxL1.jl:29-32)"
function LinearAlgebra.norm(v::RLSVector{T}, p::Real) where {T}
  p == 2 && return norm(v)
  p == 1 || throw(ArgumentError("norm(::RLSVector, p): p = 1 or 2"))
  r = Ref{Float32}(0f0)
  check(v.ctx, ccall((:rls_asum, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ref{Float32}), v.ctx.handle, dtypecode(T), v.n, v.ptr, r), "rls_asum")
  r[]
end
function LinearAlgebra.dot(x::RLSVector{T}, y::RLSVector{T}) where {T}
  r = Ref{NTuple{2,Float32}}((0f0, 0f0))
  check(x.ctx, ccall((:rls_dotc, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), x.ctx.handle, dtypecode(T), x.n, x.ptr, y.ptr, r), "rls_dotc")
  T <: Complex ? T(r[][1], r[][2]) : T(r[][1])
end
function LinearAlgebra.rmul!(v::RLSVector{T}, a::Number) where {T}
  z = ComplexF32(a)
  check(v.ctx, ccall((:rls_scal, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Float32, Float32, Ptr{Cvoid}), v.ctx.handle, dtypecode(T), v.n, real(z), imag(z), v.ptr), "rls_scal")
  v
end
"y .+= a .* x  (the fused broadcasts of src/CGNR.jl:163-174, src/FISTA.jl:147-154)"
function axpy!(a::Number, x::RLSVector{T}, y::RLSVector{T}) where {T}
  z = ComplexF32(a)
  check(y.ctx, ccall((:rls_axpy, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Float32, Float32, Ptr{Cvoid}, Ptr{Cvoid}), y.ctx.handle, dtypecode(T), y.n, real(z), imag(z), x.ptr, y.ptr), "rls_axpy")
  y
end

"z = a x + b y (z may alias x or y): the two-vector form every fused broadcast of the solver loops reduces to"
function lincomb!(z::RLSVector{T}, a::Number, x::RLSVector{T}, b::Number, y::RLSVector{T}) where {T}
  za, zb = ComplexF32(a), ComplexF32(b)
  check(z.ctx, ccall((:rls_lincomb, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Float32, Float32, Ptr{Cvoid}, Float32, Float32, Ptr{Cvoid}, Ptr{Cvoid}),
                     z.ctx.handle, dtypecode(T), z.n, real(za), imag(za), x.ptr, real(zb), imag(zb), y.ptr, z.ptr), "rls_lincomb")
  z
end
"y = a x + b y"
function axpby!(a::Number, x::RLSVector{T}, b::Number, y::RLSVector{T}) where {T}
  za, zb = ComplexF32(a), ComplexF32(b)
  check(y.ctx, ccall((:rls_axpby, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Float32, Float32, Ptr{Cvoid}, Float32, Float32, Ptr{Cvoid}),
                     y.ctx.handle, dtypecode(T), y.n, real(za), imag(za), x.ptr, real(zb), imag(zb), y.ptr), "rls_axpby")
  y
end
Base.zero(v::RLSVector{T}) where {T} = fill!(similar(v), zero(T))          # CGStateVariables(zero(x), ...) src/ADMM.jl:129
Base.copy(v::RLSVector{T}) where {T} = copyto!(similar(v), v)
# prepareMatrixStates deep-copies the solver state once per column of B (src/MultiThreading.jl:43-46): a field-wise copy
# of the struct would alias the device memory of every column, so a deep copy is a new device vector with the same content
function Base.deepcopy_internal(v::RLSVector{T}, dict::IdDict) where {T}
  haskey(dict, v) && return dict[v]
  w = copyto!(similar(v), v)
  dict[v] = w
  w
end
Base.dotview(v::RLSVector, ::Colon) = v                                     # state.res[:] .= Inf     src/FISTA.jl:123

# ---- broadcasting -----------------------------------------------------------------------------------
# The unchanged solver loops update their state with fused broadcasts -- `v .= w`, `v .= 0`, `v .*= c`, `v .+= c .* w`,
# `v .-= w`, `v .-= rho .* w`, `x .- xold`, `xold .= x .- xold` (src/CGNR.jl:163-174, src/FISTA.jl:147-154,172,
# src/ADMM.jl:236,243,259-267,282-284; IterativeSolvers.cg! uses the same shapes).  Every one of them is a LINEAR
# COMBINATION of device vectors with scalar coefficients, so a lazy Broadcasted tree is read off as a list of
# (coefficient, vector) terms and evaluated with rls_scal / rls_axpy / rls_axpby / rls_lincomb: no scalar indexing,
# one or two launches per statement.  Anything that is not such a combination raises (use Array(v)).
struct RLSStyle <: Broadcast.AbstractArrayStyle{1} end
RLSStyle(::Val{N}) where {N} = RLSStyle()
Base.BroadcastStyle(::Type{<:RLSVector}) = RLSStyle()

isscalar(x) = x isa Number || (x isa Base.RefValue && x[] isa Number)
scalar(x) = x isa Base.RefValue ? x[] : x
unsupported(bc) = error("broadcast over an RLSVector that is not a linear combination of device vectors: $(bc.f); use Array(v)")

terms(v::RLSVector, s) = Any[(ComplexF32(s), v)]
terms(x, s) = error("broadcast over an RLSVector with a $(typeof(x)) operand; device vectors and scalars only")
function terms(bc::Broadcast.Broadcasted, s)
  f, a = bc.f, bc.args
  if f === identity && length(a) == 1
    return terms(a[1], s)
  elseif f === +
    return reduce(vcat, [terms(x, s) for x in a])
  elseif f === - && length(a) == 1
    return terms(a[1], -s)
  elseif f === - && length(a) == 2
    return vcat(terms(a[1], s), terms(a[2], -s))
  elseif f === * && length(a) == 2 && isscalar(a[1])
    return terms(a[2], s * scalar(a[1]))
  elseif f === * && length(a) == 2 && isscalar(a[2])
    return terms(a[1], s * scalar(a[2]))
  elseif f === / && length(a) == 2 && isscalar(a[2])
    return terms(a[1], s / scalar(a[2]))
  end
  unsupported(bc)
end

firstvector(v::RLSVector) = v
firstvector(x) = nothing
function firstvector(bc::Broadcast.Broadcasted)
  for x in bc.args
    v = firstvector(x)
    v === nothing || return v
  end
  nothing
end
function Base.similar(bc::Broadcast.Broadcasted{RLSStyle}, ::Type{T}) where {T}
  v = firstvector(bc)
  RLSVector{T}(undef, length(axes(bc)[1]); ctx = v.ctx)
end

"dest = sum of the terms; terms that refer to dest itself are its own coefficient"
function evaluate!(dest::RLSVector{T}, ts) where {T}
  cd = ComplexF32(0)
  others = Any[]
  for (c, v) in ts
    length(v) == length(dest) || throw(DimensionMismatch("broadcast over device vectors of lengths $(length(v)) and $(length(dest))"))
    if v === dest
      cd += c
    else
      k = findfirst(t -> t[2] === v, others)
      k === nothing ? push!(others, (c, v)) : (others[k] = (others[k][1] + c, v))
    end
  end
  if isempty(others)
    cd == 1 || (cd == 0 ? fill!(dest, 0) : rmul!(dest, cd))
    return dest
  end
  if length(others) >= 2 && cd == 0                # z = a x + b y
    lincomb!(dest, others[1][1], others[1][2], others[2][1], others[2][2])
    rest = others[3:end]
  else                                             # y = a x + cd y
    cd == 1 ? axpy!(others[1][1], others[1][2], dest) : axpby!(others[1][1], others[1][2], cd, dest)
    rest = others[2:end]
  end
  for (c, v) in rest
    axpy!(c, v, dest)
  end
  dest
end

Base.copyto!(dest::RLSVector, bc::Broadcast.Broadcasted{RLSStyle}) = evaluate!(dest, terms(bc, 1f0))
# `v .= 0`, `v .= x0` with a scalar x0, `res[:] .= Inf`
Base.copyto!(dest::RLSVector, bc::Broadcast.Broadcasted{<:Broadcast.AbstractArrayStyle{0}}) = fill!(dest, bc[])

# ---- dense operator: the type of A --------------------------------------------------------------
mutable struct RLSMatrix{T} <: AbstractMatrix{T}
  ptr::Ptr{Cvoid}
  M::Int
  N::Int
  lda::Int
  ctx::Context
  op::Ptr{Cvoid}      # rls_operator handle (C_NULL for plain data matrices: right-hand sides, solutions)
end
Base.size(A::RLSMatrix) = (A.M, A.N)
Base.getindex(::RLSMatrix, ::Int, ::Int) = error("scalar indexing of a device matrix is disabled; use Array(A)")

"an M x N device matrix without an operator handle: the type of a matrix right-hand side B and of a matrix of solutions"
function RLSMatrix{T}(::UndefInitializer, M::Integer, N::Integer; ctx = context()) where {T}
  dtypecode(T)
  p = Ref{Ptr{Cvoid}}(C_NULL)
  check(ctx, ccall((:rls_malloc, librls[]), Int32, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), ctx.handle, max(M * N, 1) * sizeof(T), p), "rls_malloc")
  A = RLSMatrix{T}(p[], M, N, M, ctx, C_NULL)
  finalizer(x -> ccall((:rls_free, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), x.ctx.handle, x.ptr), A)
end
"upload a host matrix as plain data (no operator): `solve!(solver, rhs(B))` for a matrix of right-hand sides"
function rhs(b::Matrix{T}; ctx = context()) where {T}
  B = RLSMatrix{T}(undef, size(b, 1), size(b, 2); ctx)
  check(ctx, ccall((:rls_memcpy_h2d, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{T}, Csize_t), ctx.handle, B.ptr, b, sizeof(b)), "rls_memcpy_h2d")
  B
end
function Base.Array(A::RLSMatrix{T}) where {T}
  A.lda == A.M || error("Array(::RLSMatrix) needs a contiguous matrix")
  a = Matrix{T}(undef, A.M, A.N)
  check(A.ctx, ccall((:rls_memcpy_d2h, librls[]), Int32, (Ptr{Cvoid}, Ptr{T}, Ptr{Cvoid}, Csize_t), A.ctx.handle, a, A.ptr, sizeof(a)), "rls_memcpy_d2h")
  a
end
"device pointer of column j (1-based)"
colptr(A::RLSMatrix{T}, j::Integer) where {T} = A.ptr + (j - 1) * A.lda * sizeof(T)
# b[:, i] (src/MultiThreading.jl:35): indexing copies in Julia, so this is a new device vector holding column i
function Base.getindex(B::RLSMatrix{T}, ::Colon, j::Integer) where {T}
  1 <= j <= B.N || throw(BoundsError(B, (:, j)))
  v = RLSVector{T}(undef, B.M; ctx = B.ctx)
  check(B.ctx, ccall((:rls_memcpy_d2d, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), B.ctx.handle, v.ptr, colptr(B, j), B.M * sizeof(T)), "rls_memcpy_d2d")
  v
end
# hcat of the columns' solutions (src/MultiThreading.jl:79: mapreduce(solversolution, hcat, states) folds pairwise)
ncols(v::RLSVector) = 1
ncols(A::RLSMatrix) = A.N
nrows(v::RLSVector) = v.n
nrows(A::RLSMatrix) = A.M
function Base.hcat(xs::Union{RLSVector{T},RLSMatrix{T}}...) where {T}
  M = nrows(xs[1])
  all(x -> nrows(x) == M, xs) || throw(DimensionMismatch("hcat of device arrays with different numbers of rows"))
  ctx = xs[1].ctx
  out = RLSMatrix{T}(undef, M, sum(ncols, xs); ctx)
  j = 1
  for x in xs
    if x isa RLSMatrix && x.lda != x.M
      for c in 1:x.N   # padded leading dimension: column by column
        check(ctx, ccall((:rls_memcpy_d2d, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), ctx.handle, colptr(out, j + c - 1), colptr(x, c), M * sizeof(T)), "rls_memcpy_d2d")
      end
    else
      check(ctx, ccall((:rls_memcpy_d2d, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), ctx.handle, colptr(out, j), x.ptr, M * ncols(x) * sizeof(T)), "rls_memcpy_d2d")
    end
    j += ncols(x)
  end
  out
end
function RLSMatrix(a::Matrix{T}; ctx = context()) where {T}
  p = Ref{Ptr{Cvoid}}(C_NULL)
  check(ctx, ccall((:rls_malloc, librls[]), Int32, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), ctx.handle, sizeof(a), p), "rls_malloc")
  check(ctx, ccall((:rls_memcpy_h2d, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{T}, Csize_t), ctx.handle, p[], a, sizeof(a)), "rls_memcpy_h2d")
  o = Ref{Ptr{Cvoid}}(C_NULL)
  if T <: RLSSingle   # (Float64 / ComplexF64 operators carry no rls_operator: their products go through rls_gemv_d)
    check(ctx, ccall((:rls_operator_create, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Int64, Ptr{Cvoid}, Int64, Ref{Ptr{Cvoid}}),
                     ctx.handle, dtypecode(T), size(a, 1), size(a, 2), p[], size(a, 1), o), "rls_operator_create")
  else
    dtypecode(T)
  end
  A = RLSMatrix{T}(p[], size(a, 1), size(a, 2), size(a, 1), ctx, o[])
  finalizer(A) do x
    x.op == C_NULL || ccall((:rls_operator_destroy, librls[]), Int32, (Ptr{Cvoid},), x.op)
    ccall((:rls_free, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), x.ctx.handle, x.ptr)
  end
end
"the operator constructor on a context other than the default one (row shards: one context per GPU)"
RLSMatrix(a::Matrix, ctx::Context) = RLSMatrix(a; ctx)

"5-arg mul!: y = alpha * op(A) * x + beta * y"
function gemv!(op::Int32, A::RLSMatrix{T}, x::RLSVector{T}, y::RLSVector{T}, alpha::Number, beta::Number) where {T}
  a, b = ComplexF32(alpha), ComplexF32(beta)
  check(A.ctx, ccall((:rls_gemv, librls[]), Int32,
                     (Ptr{Cvoid}, Int32, Int32, Int64, Int64, Float32, Float32, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Float32, Float32, Ptr{Cvoid}),
                     A.ctx.handle, dtypecode(T), op, A.M, A.N, real(a), imag(a), A.ptr, A.lda, x.ptr, real(b), imag(b), y.ptr), "rls_gemv")
  y
end
LinearAlgebra.mul!(y::RLSVector{T}, A::RLSMatrix{T}, x::RLSVector{T}, alpha::Number, beta::Number) where {T} = gemv!(RLS_OP_N, A, x, y, alpha, beta)
LinearAlgebra.mul!(y::RLSVector{T}, A::RLSMatrix{T}, x::RLSVector{T}) where {T} = gemv!(RLS_OP_N, A, x, y, 1, 0)
LinearAlgebra.mul!(x::RLSVector{T}, At::Adjoint{T,RLSMatrix{T}}, y::RLSVector{T}, alpha::Number, beta::Number) where {T} = gemv!(RLS_OP_C, parent(At), y, x, alpha, beta)
LinearAlgebra.mul!(x::RLSVector{T}, At::Adjoint{T,RLSMatrix{T}}, y::RLSVector{T}) where {T} = gemv!(RLS_OP_C, parent(At), y, x, 1, 0)
LinearAlgebra.mul!(x::RLSVector{T}, At::Transpose{T,RLSMatrix{T}}, y::RLSVector{T}) where {T} = gemv!(RLS_OP_T, parent(At), y, x, 1, 0)

"A' * A is LAZY for this operator type, so the unchanged constructors (src/CGNR.jl:49) land in the
matrix-free mode whose apply is the one-pass kernel (rls_operator_mul_normal)."
struct RLSNormalOp{T} <: AbstractMatrix{T}
  A::RLSMatrix{T}
end
Base.size(N::RLSNormalOp) = (N.A.N, N.A.N)
Base.:*(At::Adjoint{T,RLSMatrix{T}}, A::RLSMatrix{T}) where {T} = (parent(At) === A ? RLSNormalOp{T}(A) : error("only A' * A is supported"))
function LinearAlgebra.mul!(v::RLSVector{T}, N::RLSNormalOp{T}, p::RLSVector{T}) where {T}
  check(N.A.ctx, ccall((:rls_operator_mul_normal, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), N.A.op, p.ptr, v.ptr), "rls_operator_mul_normal")
  v
end

# ---- Float64 / ComplexF64: the same protocol through the rls_*_d entry points (double scalars in, double results out) ------------
# The reference's own suites run every solver in Float32 AND Float64 (test/testSolvers.jl:242) and the prox tests in ComplexF64
# (test/testProxMaps.jl:47,78,106).  The tuned path is Float32 / ComplexF32 (SURVEY 8a); on double-precision arrays the UNCHANGED
# loops of src/CGNR.jl, src/FISTA.jl and src/ADMM.jl run on these methods (the extension's fused init! / iterate are restricted
# to RLSSingle element types).
function Base.fill!(v::RLSVector{T}, c) where {T<:RLSDouble}
  z = ComplexF64(c)
  check(v.ctx, ccall((:rls_fill_d, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Float64, Float64), v.ctx.handle, dtypecode(T), v.n, v.ptr, real(z), imag(z)), "rls_fill_d")
  v
end
function LinearAlgebra.norm(v::RLSVector{T}) where {T<:RLSDouble}
  r = Ref{NTuple{2,Float64}}((0.0, 0.0))
  check(v.ctx, ccall((:rls_nrm2_d, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}), v.ctx.handle, dtypecode(T), v.n, v.ptr, r), "rls_nrm2_d")
  r[][1]
end
function LinearAlgebra.norm(v::RLSVector{T}, p::Real) where {T<:RLSDouble}
  p == 2 && return norm(v)
  p == 1 || throw(ArgumentError("norm(::RLSVector, p): p = 1 or 2"))
  r = Ref{NTuple{2,Float64}}((0.0, 0.0))
  check(v.ctx, ccall((:rls_asum_d, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}), v.ctx.handle, dtypecode(T), v.n, v.ptr, r), "rls_asum_d")
  r[][1]
end
function LinearAlgebra.dot(x::RLSVector{T}, y::RLSVector{T}) where {T<:RLSDouble}
  r = Ref{NTuple{2,Float64}}((0.0, 0.0))
  check(x.ctx, ccall((:rls_dotc_d, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), x.ctx.handle, dtypecode(T), x.n, x.ptr, y.ptr, r), "rls_dotc_d")
  T <: Complex ? T(r[][1], r[][2]) : T(r[][1])
end
function LinearAlgebra.rmul!(v::RLSVector{T}, a::Number) where {T<:RLSDouble}
  z = ComplexF64(a)
  check(v.ctx, ccall((:rls_scal_d, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Float64, Float64, Ptr{Cvoid}), v.ctx.handle, dtypecode(T), v.n, real(z), imag(z), v.ptr), "rls_scal_d")
  v
end
function axpy!(a::Number, x::RLSVector{T}, y::RLSVector{T}) where {T<:RLSDouble}
  z = ComplexF64(a)
  check(y.ctx, ccall((:rls_axpy_d, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}), y.ctx.handle, dtypecode(T), y.n, real(z), imag(z), x.ptr, y.ptr), "rls_axpy_d")
  y
end
function lincomb!(z::RLSVector{T}, a::Number, x::RLSVector{T}, b::Number, y::RLSVector{T}) where {T<:RLSDouble}
  za, zb = ComplexF64(a), ComplexF64(b)
  check(z.ctx, ccall((:rls_lincomb_d, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Float64, Float64, Ptr{Cvoid}, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}),
                     z.ctx.handle, dtypecode(T), z.n, real(za), imag(za), x.ptr, real(zb), imag(zb), y.ptr, z.ptr), "rls_lincomb_d")
  z
end
function gemv!(op::Int32, A::RLSMatrix{T}, x::RLSVector{T}, y::RLSVector{T}, alpha::Number, beta::Number) where {T<:RLSDouble}
  a, b = ComplexF64(alpha), ComplexF64(beta)
  check(A.ctx, ccall((:rls_gemv_d, librls[]), Int32,
                     (Ptr{Cvoid}, Int32, Int32, Int64, Int64, Float64, Float64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Float64, Float64, Ptr{Cvoid}),
                     A.ctx.handle, dtypecode(T), op, A.M, A.N, real(a), imag(a), A.ptr, A.lda, x.ptr, real(b), imag(b), y.ptr), "rls_gemv_d")
  y
end
"v = A' * (A * p) as two products (no one-pass kernel in double precision)"
function LinearAlgebra.mul!(v::RLSVector{T}, N::RLSNormalOp{T}, p::RLSVector{T}) where {T<:RLSDouble}
  t = RLSVector{T}(undef, N.A.M; ctx = N.A.ctx)
  gemv!(RLS_OP_N, N.A, p, t, 1, 0)
  gemv!(RLS_OP_C, N.A, t, v, 1, 0)
end

"""
    gram(A::RLSMatrix) -> RLSMatrix

The explicit Gram matrix `A' * A` on the device (Hermitian rank-M update on the matrix cores, `rls_gram`).  Passing it as
`CGNR(A; AHA = gram(A))` (likewise FISTA, ADMM) selects Gram mode: the per-iteration operator apply is one N x N GEMV, and
where AHA fits the register files the whole `step` call runs as one resident launch (DESIGN.md 4.3).  `A' * A` itself stays
lazy (matrix-free), so the unchanged constructor default keeps the reference's matrix-free arithmetic.
"""
function gram(A::RLSMatrix{T}) where {T}
  T <: RLSSingle || throw(ArgumentError("gram: the matrix-core Gram GEMM is Float32 / ComplexF32; got $T (form A' * A on the host and upload it)"))
  N = A.N
  p = Ref{Ptr{Cvoid}}(C_NULL)
  check(A.ctx, ccall((:rls_malloc, librls[]), Int32, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), A.ctx.handle, N * N * sizeof(T), p), "rls_malloc")
  check(A.ctx, ccall((:rls_gram, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64),
                     A.ctx.handle, dtypecode(T), A.M, A.N, A.ptr, A.lda, p[], N), "rls_gram")
  o = Ref{Ptr{Cvoid}}(C_NULL)   # the operator of the pair (A, AHA): rls_operator_set_gram switches its plans to Gram mode
  check(A.ctx, ccall((:rls_operator_create, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Int64, Ptr{Cvoid}, Int64, Ref{Ptr{Cvoid}}),
                     A.ctx.handle, dtypecode(T), A.M, A.N, A.ptr, A.lda, o), "rls_operator_create")
  check(A.ctx, ccall((:rls_operator_set_gram, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), o[], p[], N), "rls_operator_set_gram")
  G = RLSMatrix{T}(p[], N, N, N, A.ctx, o[])
  finalizer(G) do x
    ccall((:rls_operator_destroy, librls[]), Int32, (Ptr{Cvoid},), x.op)
    ccall((:rls_free, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), x.ctx.handle, x.ptr)
  end
end

"""
    PgmPlan(A::RLSMatrix) -> PgmPlan or nothing

Launch scratch of the resident OptISTA / POGM (restart = :none) iteration blocks (`rls_pgm_create`); `nothing` when the
operator does not fit the resident form (the caller then stays on `rls_optista_update_async` / `rls_pogm_update_async`).
"""
mutable struct PgmPlan
  handle::Ptr{Cvoid}
  ctx::Context
  function PgmPlan(handle::Ptr{Cvoid}, ctx::Context)
    p = new(handle, ctx)
    finalizer(x -> ccall((:rls_pgm_destroy, librls[]), Int32, (Ptr{Cvoid},), x.handle), p)
  end
end
function PgmPlan(A::RLSMatrix)
  h = Ref{Ptr{Cvoid}}(C_NULL)
  st = ccall((:rls_pgm_create, librls[]), Int32, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), A.op, h)
  st == Int32(-2) && return nothing   # RLS_E_UNSUPPORTED: not an error
  check(A.ctx, st, "rls_pgm_create")
  PgmPlan(h[], A.ctx)
end

"""
    pgm_step_resident!(plan, kind, coefs, first_iteration, v0, v1, v2, o0, res, x0, reg_kind, proj_kind, norm_x0, rel_tol, state) -> Bool

Up to 48 iterations (`size(coefs, 2)`; `coefs` is 8 x n Float32, one column per iteration: the float arguments of the
per-iteration `*_update_async` entry points) as ONE launch.  kind 0 = OptISTA (x, y, z, zold), 1 = POGM (xbuf, ybuf, z, xold).
Returns false when the plan has retired (a launch was lost earlier): continue launch by launch.
"""
function pgm_step_resident!(plan::PgmPlan, kind::Integer, coefs::Matrix{Float32}, first_iteration::Integer, v0::RLSVector{T},
                            v1::RLSVector{T}, v2::RLSVector{T}, o0::RLSVector{T}, res::RLSVector{T}, x0::RLSVector{T},
                            reg_kind::Integer, proj_kind::Integer, norm_x0::Real, rel_tol::Real, state::RLSVector{Float32}) where {T}
  size(coefs, 1) == 8 || error("coefs must be 8 x n")
  st = ccall((:rls_pgm_step_resident, librls[]), Int32,
             (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Float32}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32,
              Float32, Float32, Ptr{Cvoid}),
             plan.handle, Int32(kind), Int32(size(coefs, 2)), Int32(first_iteration), coefs, v0.ptr, v1.ptr, v2.ptr, o0.ptr, res.ptr, x0.ptr,
             Int32(reg_kind), Int32(proj_kind), Float32(norm_x0), Float32(rel_tol), state.ptr)
  st == Int32(-2) && return false
  check(plan.ctx, st, "rls_pgm_step_resident")
  true
end

"""
    pogm_step_resident_restart!(plan, n_steps, first_iteration, rho, lambda, sigma_fac, iterations, xbuf, ybuf, z, w, xold, res, x0,
                                reg_kind, proj_kind, norm_x0, rel_tol, state) -> Bool

POGM with `restart = :gradient` as ONE launch of `n_steps` iterations: theta, sigma, gamma live in the 8-word device record
`state` and the kernel forms every iteration's coefficients from them (src/POGM.jl:183-232).  `iterations` = the solve's
iteration count minus the count the record started from.  Returns false when the plan has retired.
"""
function pogm_step_resident_restart!(plan::PgmPlan, n_steps::Integer, first_iteration::Integer, rho::Real, lambda::Real, sigma_fac::Real,
                                     iterations::Integer, xbuf::RLSVector{T}, ybuf::RLSVector{T}, z::RLSVector{T}, w::RLSVector{T},
                                     xold::RLSVector{T}, res::RLSVector{T}, x0::RLSVector{T}, reg_kind::Integer, proj_kind::Integer,
                                     norm_x0::Real, rel_tol::Real, state::RLSVector{Float32}) where {T}
  st = ccall((:rls_pogm_step_resident_restart, librls[]), Int32,
             (Ptr{Cvoid}, Int32, Int32, Float32, Float32, Float32, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
              Ptr{Cvoid}, Int32, Int32, Float32, Float32, Ptr{Cvoid}),
             plan.handle, Int32(n_steps), Int32(first_iteration), Float32(rho), Float32(lambda), Float32(sigma_fac), Int32(iterations),
             xbuf.ptr, ybuf.ptr, z.ptr, w.ptr, xold.ptr, res.ptr, x0.ptr, Int32(reg_kind), Int32(proj_kind), Float32(norm_x0),
             Float32(rel_tol), state.ptr)
  st == Int32(-2) && return false
  check(plan.ctx, st, "rls_pogm_step_resident_restart")
  true
end

"launches of the sequence that gave up (they changed nothing); synchronises"
function pgm_lost(plan::PgmPlan)
  lost = Ref{Int32}(0); total = Ref{Int32}(0)
  check(plan.ctx, ccall((:rls_pgm_lost, librls[]), Int32, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}), plan.handle, lost, total), "rls_pgm_lost")
  Int(lost[])
end

"whole solve in one enqueue; methods are added by the RegularizedLeastSquares extension"
function solve_fused! end
"K solvers with their own small matrices as ONE launch (defined by the RegularizedLeastSquares extension)"
function solve_group! end
"`scheduler = RLSMI355X.BatchedState` for `solve!(solver, B::RLSMatrix)`: the columns share every pass over A (matrix cores); defined by the extension"
function BatchedState end

"a NON-owning device vector over column j of a device matrix (the matrix must outlive it)"
column_view(A::RLSMatrix{T}, j::Integer) where {T} = RLSVector{T}(colptr(A, j), A.M, A.ctx, Val(:view))

# ---- single-process multi-GPU: the library's communicator (include/rls_mi355x.h, rls_comm_*) ------------------------
# One Julia process drives every GPU of the node (BASELINE config 5: one tall A row-partitioned, one all-reduce of the
# length-N partial product per operator apply).  The fan-out over the ranks happens INSIDE the library (one host worker
# thread per rank, the Threads.@threads of src/MultiThreading.jl:60-78), so these calls are made from one task.
const RLS_COMM_AUTO = Int32(0); const RLS_COMM_RCCL = Int32(1); const RLS_COMM_DIRECT = Int32(2)

mutable struct Comm
  handle::Ptr{Cvoid}
  ctxs::Vector{Context}
  function Comm(devices::AbstractVector{<:Integer}; transport::Integer = RLS_COMM_AUTO, threads::Bool = true)
    devs = collect(Int32, devices)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    st = ccall((:rls_comm_create, librls[]), Int32, (Int32, Ptr{Int32}, Ptr{Ptr{Cvoid}}, Int32, Ref{Ptr{Cvoid}}), length(devs), devs, C_NULL, transport, h)
    st == 0 || throw(RLSError(st, "rls_comm_create failed (RCCL needs one distinct device per rank; the direct transport needs peer access)"))
    ctxs = Context[]
    for r in 0:length(devs)-1
      c = Ref{Ptr{Cvoid}}(C_NULL)
      ccall((:rls_comm_ctx, librls[]), Int32, (Ptr{Cvoid}, Int32, Ref{Ptr{Cvoid}}), h[], r, c) == 0 || throw(RLSError(Int32(-1), "rls_comm_ctx"))
      push!(ctxs, Context(c[], devs[r+1], Val(:borrowed)))
    end
    ccall((:rls_comm_set_threads, librls[]), Int32, (Ptr{Cvoid}, Int32), h[], threads ? 1 : 0)
    comm = new(h[], ctxs)
    # Arrays and plans allocated on a rank's context reference that Context, and through `owner` this Comm: the communicator
    # (which owns the contexts) stays reachable as long as anything allocated on it is.  Julia still runs the finalizers of
    # objects that die TOGETHER in no particular order (always so at exit): for that case rls_free never dereferences a handle
    # that is not a live context (csrc/api.hip) and plans check rls_ctx_alive before they free through their context.
    for c in ctxs
      c.owner = comm
    end
    finalizer(c -> ccall((:rls_comm_destroy, librls[]), Int32, (Ptr{Cvoid},), c.handle), comm)
  end
end
Base.length(c::Comm) = Int(ccall((:rls_comm_size, librls[]), Int32, (Ptr{Cvoid},), c.handle))
transport(c::Comm) = ccall((:rls_comm_transport, librls[]), Int32, (Ptr{Cvoid},), c.handle)
"(matrix, requested): the hipDeviceCanAccessPeer matrix `Comm(...)` probed (1 / 0 / -1 = query failed; row r, column t: rank r's device
can store into rank t's) and the transport that was asked for; `transport(c)` is the one in use"
function peer_access(c::Comm)
  n = length(c)
  m = zeros(Int32, n * n)
  req = Ref{Int32}(-1)
  ccall((:rls_comm_peer_access, librls[]), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ref{Int32}), c.handle, m, req) == 0 || throw(RLSError(Int32(-1), "rls_comm_peer_access"))
  (permutedims(reshape(m, n, n)), req[])
end
synchronize(c::Comm) = check(c.ctxs[1], ccall((:rls_comm_sync, librls[]), Int32, (Ptr{Cvoid},), c.handle), "rls_comm_sync")

"in place: every rank's vector becomes the sum over the ranks (asynchronous, on the ranks' streams)"
function allreduce_sum!(c::Comm, vs::Vector{RLSVector{T}}) where {T}
  length(vs) == length(c) || throw(DimensionMismatch("one vector per rank"))
  ptrs = Ptr{Cvoid}[v.ptr for v in vs]
  check(c.ctxs[1], ccall((:rls_allreduce_sum, librls[]), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Int64, Int32), c.handle, ptrs, length(vs[1]), dtypecode(T)), "rls_allreduce_sum")
  vs
end

"row blocks [lo, hi] (1-based, multiples of 4 rows) of an M-row matrix for `n` ranks"
function shard_rows(M::Integer, n::Integer, r::Integer)
  lo = (M * (r - 1) ÷ n) ÷ 4 * 4
  hi = r == n ? M : (M * r ÷ n) ÷ 4 * 4
  (lo + 1, hi)
end

"upload the row shards of a host matrix, one operator per rank's context"
shard_operator(c::Comm, a::Matrix{T}) where {T} =
  [RLSMatrix(a[first(shard_rows(size(a, 1), length(c), r)):last(shard_rows(size(a, 1), length(c), r)), :], c.ctxs[r]) for r in 1:length(c)]

end # module
