# RLSMI355X.jl -- thin `ccall` binding of librls_mi355x.so (C ABI: include/rls_mi355x.h) and the
# device array / operator types that select the MI355X backend by dispatch, the way the reference's
# GPU extensions do (ext/RegularizedLeastSquaresGPUArraysExt/*.jl, loading mechanism Project.toml:20-27).
#
# This file cannot be executed in the build container (no Julia); it is kept in lock-step with the
# Python harness (regularizedleastsquares.jl_amd/*.py), which issues exactly the same ABI call
# sequences and is what the parity tests run.
module RLSMI355X

using LinearAlgebra, Libdl

const librls = Ref{String}(get(ENV, "RLS_MI355X_LIB", "librls_mi355x.so"))

const RLS_F32 = Int32(0); const RLS_C32 = Int32(1)
const RLS_OP_N = Int32(0); const RLS_OP_T = Int32(1); const RLS_OP_C = Int32(2)

struct RLSError <: Exception
  code::Int32
  msg::String
end

mutable struct Context
  handle::Ptr{Cvoid}
  device::Int32
  function Context(device::Integer = 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    st = ccall((:rls_ctx_create, librls[]), Int32, (Int32, Ref{Ptr{Cvoid}}), device, h)
    st == 0 || throw(RLSError(st, "rls_ctx_create failed: no usable MI355X (there is no CPU fallback)"))
    ctx = new(h[], Int32(device))
    finalizer(c -> ccall((:rls_ctx_destroy, librls[]), Int32, (Ptr{Cvoid},), c.handle), ctx)
  end
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
dtypecode(T) = throw(ArgumentError("the MI355X backend computes in Float32 / ComplexF32; got $T"))

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

# ---- dense operator: the type of A --------------------------------------------------------------
mutable struct RLSMatrix{T} <: AbstractMatrix{T}
  ptr::Ptr{Cvoid}
  M::Int
  N::Int
  lda::Int
  ctx::Context
  op::Ptr{Cvoid}      # rls_operator handle
end
Base.size(A::RLSMatrix) = (A.M, A.N)
function RLSMatrix(a::Matrix{T}; ctx = context()) where {T}
  p = Ref{Ptr{Cvoid}}(C_NULL)
  check(ctx, ccall((:rls_malloc, librls[]), Int32, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), ctx.handle, sizeof(a), p), "rls_malloc")
  check(ctx, ccall((:rls_memcpy_h2d, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{T}, Csize_t), ctx.handle, p[], a, sizeof(a)), "rls_memcpy_h2d")
  o = Ref{Ptr{Cvoid}}(C_NULL)
  check(ctx, ccall((:rls_operator_create, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Int64, Ptr{Cvoid}, Int64, Ref{Ptr{Cvoid}}),
                   ctx.handle, dtypecode(T), size(a, 1), size(a, 2), p[], size(a, 1), o), "rls_operator_create")
  A = RLSMatrix{T}(p[], size(a, 1), size(a, 2), size(a, 1), ctx, o[])
  finalizer(A) do x
    ccall((:rls_operator_destroy, librls[]), Int32, (Ptr{Cvoid},), x.op)
    ccall((:rls_free, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), x.ctx.handle, x.ptr)
  end
end

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

end # module
