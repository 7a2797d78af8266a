# Extension loaded when both RLSMI355X and RegularizedLeastSquares are present (weakdeps mechanism,
# the same the reference uses for its GPUArrays / CUDA extensions: Project.toml:20-27).
# It overloads, for RLSVector / RLSMatrix, exactly the methods SURVEY.md 8(b) lists:
#   * the internal helpers the GPUArrays ext overloads (prox pieces, enfReal!/enfPos!)
#   * the fused fast paths init!/iterate for CGNR and FISTA (sanctioned: docs/src/solvers.md:85-98,
#     precedent ext/RegularizedLeastSquaresGPUArraysExt/Kaczmarz.jl:1)
# Everything else (createLinearSolver, solve!, callbacks, Regularization types, ADMM's outer loop,
# MultiThreading schedulers) runs UNCHANGED from the reference on top of these methods.
module RLSMI355XRegularizedLeastSquaresExt

using RLSMI355X, RegularizedLeastSquares, LinearAlgebra
using RLSMI355X: RLSVector, RLSMatrix, RLSNormalOp, librls, check, dtypecode
import RegularizedLeastSquares: prox!, proxL21!, proxTV!, enfReal!, enfPos!, tv_restrictMagnitude!, tv_linearcomb!,
                                init!, iterate, CGNR, CGNRState, FISTA, FISTAState, L1Regularization, L2Regularization,
                                TVParams, λ, Kaczmarz, KaczmarzState, normalize, SystemMatrixBasedNormalization, done

const V{T} = Union{RLSVector{T}, RLSVector{Complex{T}}}

# ---- proximal maps ------------------------------------------------------------------------------
# src/proximalMaps/ProxL1.jl:18-22
function prox!(::L1Regularization, x::V{T}, lam::T) where {T<:Real}
  check(x.ctx, ccall((:rls_prox_l1, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Float32), x.ctx.handle, dtypecode(eltype(x)), length(x), x.ptr, lam), "rls_prox_l1"); x
end
# src/proximalMaps/ProxL2.jl:18-21
function prox!(::L2Regularization, x::V{T}, lam::T) where {T<:Real}
  check(x.ctx, ccall((:rls_prox_l2, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Float32), x.ctx.handle, dtypecode(eltype(x)), length(x), x.ptr, lam), "rls_prox_l2"); x
end
# src/proximalMaps/ProxL21.jl:30-35  (ext/RegularizedLeastSquaresGPUArraysExt/ProxL21.jl:1)
function proxL21!(x::V{T}, lam::T, slices::Int64) where {T<:Real}
  check(x.ctx, ccall((:rls_prox_l21, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Int64, Ptr{Cvoid}, Float32), x.ctx.handle, dtypecode(eltype(x)), length(x), slices, x.ptr, lam), "rls_prox_l21"); x
end
# src/Utils.jl:114-144  (ext/.../Utils.jl:4-34)
function enfReal!(x::RLSVector{T}) where {T<:Complex}
  check(x.ctx, ccall((:rls_prox_real, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}), x.ctx.handle, dtypecode(T), length(x), x.ptr), "rls_prox_real"); nothing
end
enfReal!(::RLSVector{T}) where {T<:Real} = nothing
function enfPos!(x::RLSVector{T}) where {T}
  # enfReal! has already run (prox!(::PositiveRegularization) calls both, ProxPositive.jl:16-20)
  check(x.ctx, ccall((:rls_prox_positive, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}), x.ctx.handle, dtypecode(T), length(x), x.ptr), "rls_prox_positive"); nothing
end
# whole FGP loop: proxTV!(x, lambda, p::TVParams; iterationsTV)   src/proximalMaps/ProxTV.jl:89-125
function proxTV!(reg, x::V{T}, lam::T, shape, dims; iterationsTV = 10, kwargs...) where {T<:Real}
  sh = collect(Int64, shape); d0 = collect(Int32, dims) .- Int32(1)
  dt = dtypecode(eltype(x))
  need = ccall((:rls_prox_tv_workspace_bytes, librls[]), Csize_t, (Int32, Int32, Ptr{Int64}, Int32, Ptr{Int32}), dt, length(sh), sh, length(d0), d0)
  ws = RLSVector{eltype(x)}(undef, cld(need, sizeof(eltype(x))); ctx = x.ctx)  # TVParams scratch
  check(x.ctx, ccall((:rls_prox_tv_fgp, librls[]), Int32,
                     (Ptr{Cvoid}, Int32, Int32, Ptr{Int64}, Int32, Ptr{Int32}, Ptr{Cvoid}, Float32, Int32, Ptr{Cvoid}, Csize_t),
                     x.ctx.handle, dt, length(sh), sh, length(d0), d0, x.ptr, lam, iterationsTV, ws.ptr, need), "rls_prox_tv_fgp")
  x
end
# the two helpers the GPUArrays ext overloads one by one (ext/.../ProxTV.jl:1-17)
function tv_restrictMagnitude!(x::RLSVector{T}) where {T}
  check(x.ctx, ccall((:rls_tv_restrict, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}), x.ctx.handle, dtypecode(T), length(x), x.ptr), "rls_tv_restrict")
end
function tv_linearcomb!(rs::RLSVector{T}, t3, pq::RLSVector{T}, t2, pqOld::RLSVector{T}) where {T}
  check(rs.ctx, ccall((:rls_tv_lincomb, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Float32, Ptr{Cvoid}, Float32, Ptr{Cvoid}),
                      rs.ctx.handle, dtypecode(T), length(rs), rs.ptr, t3, pq.ptr, t2, pqOld.ptr), "rls_tv_lincomb")
end

# ---- fused CGNR: init! + iterate on device state --------------------------------------------------
const cgnr_plans = IdDict{Any,Ptr{Cvoid}}()   # state => rls_cgnr plan (destroyed with the state)

struct CgnrStatus
  iteration::Int32; done::Int32; alpha_re::Float32; alpha_im::Float32; beta_re::Float32; beta_im::Float32
  zeta::Float32; residual::Float32; z0::Float32
end

function plan_for(solver::CGNR, state::CGNRState{T,Tc,<:RLSVector}) where {T,Tc}
  get!(cgnr_plans, state) do
    A = solver.A::RLSMatrix
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(A.ctx, ccall((:rls_cgnr_create, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                       A.op, state.x.ptr, state.x₀.ptr, state.pl.ptr, state.vl.ptr, p), "rls_cgnr_create")
    finalizer(_ -> ccall((:rls_cgnr_destroy, librls[]), Int32, (Ptr{Cvoid},), p[]), state)
    p[]
  end
end

# src/CGNR.jl:107-130
function init!(solver::CGNR, state::CGNRState{T,Tc,vecTc}, b::vecTc; x0 = 0) where {T,Tc,vecTc<:RLSVector{Tc}}
  all(x0 .== 0) || error("CGNR: x0 != 0 is unsupported (the reference's branch throws as well, src/CGNR.jl:119)")
  plan = plan_for(solver, state)
  check(b.ctx, ccall((:rls_cgnr_init, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Int32),
                     plan, b.ptr, Float32(λ(solver.L2)), state.relTol, solver.iterations), "rls_cgnr_init")
  state.iteration = 0
end

# src/CGNR.jl:143-178
function iterate(solver::CGNR, state::CGNRState{T,Tc,<:RLSVector}) where {T,Tc}
  plan = plan_for(solver, state)
  st = Ref{CgnrStatus}()
  check(state.x.ctx, ccall((:rls_cgnr_get_status, librls[]), Int32, (Ptr{Cvoid}, Ref{CgnrStatus}), plan, st), "rls_cgnr_get_status")
  state.iteration = st[].iteration; state.z0 = st[].z0
  state.αl = Tc <: Complex ? Tc(st[].alpha_re, st[].alpha_im) : Tc(st[].alpha_re)
  state.βl = Tc(st[].beta_re); state.ζl = Tc(st[].zeta)
  if st[].done != 0
    for r in solver.constr
      prox!(r, state.x)
    end
    return nothing
  end
  check(state.x.ctx, ccall((:rls_cgnr_step, librls[]), Int32, (Ptr{Cvoid}, Int32), plan, 1), "rls_cgnr_step")
  state.iteration += 1
  return state.x, state
end

# ---- setup path: SystemMatrixBasedNormalization (ext/RegularizedLeastSquaresGPUArraysExt/NormalizedRegularization.jl:1-5)
function normalize(::SystemMatrixBasedNormalization, A::RLSMatrix{T}, b) where {T}
  M, N = size(A)
  e = RLSVector{real(T)}(undef, M; ctx = A.ctx)
  check(A.ctx, ccall((:rls_rownorm2, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
                     A.ctx.handle, dtypecode(T), M, N, A.ptr, M, e.ptr), "rls_rownorm2")
  return sum(Array(e)) / N   # norm(sqrt.(rownorm²))^2 / N
end

# ---- fused Kaczmarz sweep: iterate(::Kaczmarz, ::KaczmarzState) = the loop of iterate_row_index (src/Kaczmarz.jl:283-308)
# The solver is constructed on an RLSMatrix; the transposed copy and the device copies of rowindex / denom live
# beside the state (rebuilt when init! recomputes the denominators, src/Kaczmarz.jl:186-193).
const kaczmarz_aux = IdDict{Any,Any}()

function kaczmarz_aux_for(solver::Kaczmarz, state::KaczmarzState{T,<:RLSVector}) where {T}
  get!(kaczmarz_aux, state) do
    A = solver.A::RLSMatrix{T}
    M, N = size(A)
    At = RLSVector{T}(undef, M * N; ctx = A.ctx)     # transpose(A), N x M column-major
    check(A.ctx, ccall((:rls_transpose, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64),
                       A.ctx.handle, dtypecode(T), M, N, A.ptr, M, At.ptr, N), "rls_transpose")
    (At = At, rows = Ref{Any}(nothing), den = Ref{Any}(nothing), key = Ref{Any}(nothing))
  end
end

function iterate(solver::Kaczmarz, state::KaczmarzState{T,<:RLSVector}) where {T}
  done(solver, state) && return nothing
  aux = kaczmarz_aux_for(solver, state)
  if solver.randomized   # the sampling stays on the host, as in the reference (src/Kaczmarz.jl:286-288)
    RegularizedLeastSquares.StatsBase.sample!(RegularizedLeastSquares.Random.GLOBAL_RNG, solver.rowIndexCycle,
      RegularizedLeastSquares.StatsBase.weights(solver.probabilities), state.usedIndices, replace = false)
  end
  key = (objectid(solver.denom), copy(state.usedIndices))
  if aux.key[] != key   # upload the processing order: 0-based rows and their denominators
    aux.rows[] = RLSVector(reinterpret(Float32, Int32.(solver.rowindex[state.usedIndices] .- 1)); ctx = state.x.ctx)
    aux.den[] = RLSVector(Float32.(solver.denom[state.usedIndices]); ctx = state.x.ctx)
    aux.key[] = key
  end
  A = solver.A::RLSMatrix{T}
  M, N = size(A)
  check(A.ctx, ccall((:rls_kaczmarz_sweep, librls[]), Int32,
                     (Ptr{Cvoid}, Int32, Int64, Int64, Ptr{Cvoid}, Int64, Int32, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64,
                      Ptr{Cvoid}, Ptr{Cvoid}, Int32, Float32, Int32),
                     A.ctx.handle, dtypecode(T), M, N, aux.At.ptr, N, 1, state.x.ptr, N, state.u.ptr, M, state.vl.ptr, M,
                     aux.rows[].ptr, aux.den[].ptr, length(state.usedIndices), Float32(real(state.ɛw)), 1), "rls_kaczmarz_sweep")
  for r in solver.reg
    prox!(r, state.x)
  end
  state.iteration += 1
  return state.x, state
end

end # module
