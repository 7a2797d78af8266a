# Extension loaded when both RLSMI355X and RegularizedLeastSquares are present (weakdeps mechanism,
# the same the reference uses for its GPUArrays / CUDA extensions: Project.toml:20-27).
# It overloads, for RLSVector / RLSMatrix, exactly the methods SURVEY.md 8(b) lists:
#   * the internal helpers the GPUArrays ext overloads (prox pieces, enfReal!/enfPos!)
#   * the fused fast paths init!/iterate for CGNR, FISTA, ADMM and Kaczmarz (sanctioned: docs/src/solvers.md:85-98,
#     precedent ext/RegularizedLeastSquaresGPUArraysExt/Kaczmarz.jl:1); configurations a fused plan does not
#     cover (several regularisers, a non-identity regTrafo, vary_rho, L21 / LLR / nuclear terms in ADMM, ...) fall
#     through to the reference's own generic methods, which run on RLSVector through its BLAS-1 methods and its
#     broadcast lowering (RLSMI355X.jl: RLSStyle)
#   * solve! without callbacks on a device right-hand side = the whole solve in one enqueue (solve_fused!)
# Everything else (createLinearSolver, solve! with callbacks, Regularization types, MultiThreading schedulers) runs
# UNCHANGED from the reference on top of these methods.
module RLSMI355XRegularizedLeastSquaresExt

using RLSMI355X, RegularizedLeastSquares, LinearAlgebra
using RLSMI355X: RLSVector, RLSMatrix, RLSNormalOp, librls, check, dtypecode, Comm, colptr, RLSSingle, RLSDouble
import RegularizedLeastSquares: prox!, proxL21!, proxTV!, enfReal!, enfPos!, tv_restrictMagnitude!, tv_linearcomb!,
                                init!, iterate, CGNR, CGNRState, FISTA, FISTAState, ADMM, ADMMState, L1Regularization,
                                L2Regularization, L21Regularization, TVRegularization, PositiveRegularization,
                                RealRegularization, NoNormalization, TVParams, λ, Kaczmarz, KaczmarzState, normalize,
                                SystemMatrixBasedNormalization, done, AbstractSolverState, AbstractLinearSolver,
                                AbstractMatrixSolverState, solversolution, solverconvergence

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
# ---- the same maps on Float64 / ComplexF64 vectors (test/testProxMaps.jl:47,78,106 run them in ComplexF64): double scalars ----------
function prox!(::L1Regularization, x::V{Float64}, lam::Float64)
  check(x.ctx, ccall((:rls_prox_l1_d, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Float64), x.ctx.handle, dtypecode(eltype(x)), length(x), x.ptr, lam), "rls_prox_l1_d"); x
end
function prox!(::L2Regularization, x::V{Float64}, lam::Float64)
  check(x.ctx, ccall((:rls_prox_l2_d, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Float64), x.ctx.handle, dtypecode(eltype(x)), length(x), x.ptr, lam), "rls_prox_l2_d"); x
end
function proxL21!(x::V{Float64}, lam::Float64, slices::Int64)
  check(x.ctx, ccall((:rls_prox_l21_d, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Int64, Ptr{Cvoid}, Float64), x.ctx.handle, dtypecode(eltype(x)), length(x), slices, x.ptr, lam), "rls_prox_l21_d"); x
end
function enfReal!(x::RLSVector{ComplexF64})
  check(x.ctx, ccall((:rls_prox_real_d, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}), x.ctx.handle, dtypecode(ComplexF64), length(x), x.ptr), "rls_prox_real_d"); nothing
end
function enfPos!(x::RLSVector{T}) where {T<:RLSDouble}
  check(x.ctx, ccall((:rls_prox_positive_d, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}), x.ctx.handle, dtypecode(T), length(x), x.ptr), "rls_prox_positive_d"); nothing
end
function proxTV!(reg, x::V{Float64}, lam::Float64, shape, dims; iterationsTV = 10, kwargs...)
  sh = collect(Int64, shape); d0 = collect(Int32, dims) .- Int32(1)
  check(x.ctx, ccall((:rls_prox_tv_fgp_d, librls[]), Int32,
                     (Ptr{Cvoid}, Int32, Int32, Ptr{Int64}, Int32, Ptr{Int32}, Ptr{Cvoid}, Float64, Int32),
                     x.ctx.handle, dtypecode(eltype(x)), length(sh), sh, length(d0), d0, x.ptr, lam, iterationsTV), "rls_prox_tv_fgp_d")
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

struct CgnrStatus   # rls_cgnr_status (include/rls_mi355x.h), field for field
  iteration::Int32; done::Int32; alpha_re::Float32; alpha_im::Float32; beta_re::Float32; beta_im::Float32
  zeta::Float32; residual::Float32; z0::Float32; fallbacks::Int32
end

function plan_for(solver::CGNR, state::CGNRState{T,Tc,<:RLSVector}) where {T,Tc<:RLSSingle}
  get!(cgnr_plans, state) do
    A = solver.A::RLSMatrix
    op = something(operator_of(A, solver.AHA), A.op)   # matrix-free (A' * A lazy) or Gram mode (AHA = gram(A))
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(A.ctx, ccall((:rls_cgnr_create, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                       op, state.x.ptr, state.x₀.ptr, state.pl.ptr, state.vl.ptr, p), "rls_cgnr_create")
    finalizer(_ -> ccall((:rls_cgnr_destroy, librls[]), Int32, (Ptr{Cvoid},), p[]), state)
    p[]
  end
end

# src/CGNR.jl:107-130
function init!(solver::CGNR, state::CGNRState{T,Tc,vecTc}, b::vecTc; x0 = 0) where {T,Tc<:RLSSingle,vecTc<:RLSVector{Tc}}
  all(x0 .== 0) || error("CGNR: x0 != 0 is unsupported (the reference's branch throws as well, src/CGNR.jl:119)")
  plan = plan_for(solver, state)
  # normalization of the regularization parameter (:129) comes FIRST here: the device init takes lambda as an argument
  solver.L2 = normalize(solver, solver.normalizeReg, solver.L2, solver.A, b)
  check(b.ctx, ccall((:rls_cgnr_init, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Int32),
                     plan, b.ptr, Float32(λ(solver.L2)), state.relTol, solver.iterations), "rls_cgnr_init")
  state.iteration = 0
  delete!(cgnr_done, state)    # the device's `done` of this solve is not known yet
  nothing
end

"the scalars of the device plan into the host-side state (after a status read-back)"
function cgnr_take!(state::CGNRState{T,Tc}, st::CgnrStatus) where {T,Tc}
  state.iteration = st.iteration; state.z0 = st.z0
  state.αl = Tc <: Complex ? Tc(st.alpha_re, st.alpha_im) : Tc(st.alpha_re)
  state.βl = Tc(st.beta_re); state.ζl = Tc(st.zeta)
  cgnr_done[state] = st.done != 0
  st
end
const cgnr_done = IdDict{Any,Bool}()   # state => `done` as the device last reported it

# src/CGNR.jl:143-178.  One library call per iteration: the step and the status read-back (`done`, the convergence record
# the callbacks read) travel together (rls_cgnr_step_status: one host synchronisation).
function iterate(solver::CGNR, state::CGNRState{T,Tc,<:RLSVector}) where {T,Tc<:RLSSingle}
  plan = plan_for(solver, state)
  st = Ref{CgnrStatus}()
  if !haskey(cgnr_done, state)   # first call after init!: is the solve done before it starts (iterations == 0, r == 0)?
    check(state.x.ctx, ccall((:rls_cgnr_get_status, librls[]), Int32, (Ptr{Cvoid}, Ref{CgnrStatus}), plan, st), "rls_cgnr_get_status")
    cgnr_take!(state, st[])
  end
  if cgnr_done[state]
    for r in solver.constr
      prox!(r, state.x)
    end
    return nothing
  end
  check(state.x.ctx, ccall((:rls_cgnr_step_status, librls[]), Int32, (Ptr{Cvoid}, Int32, Ref{CgnrStatus}), plan, 1, st), "rls_cgnr_step_status")
  cgnr_take!(state, st[])
  return state.x, state
end

# ---- fused FISTA: init! + iterate on device state (src/FISTA.jl:110-129, :139-185) ---------------------------
const RLS_REG_NONE = Int32(0); const RLS_REG_L1 = Int32(1); const RLS_REG_L2 = Int32(2); const RLS_REG_L21 = Int32(3); const RLS_REG_TV = Int32(4)
const RLS_PROJ_NONE = Int32(0); const RLS_PROJ_REAL = Int32(1); const RLS_PROJ_POSITIVE = Int32(2)

"(reg_kind, slices) of the elementwise updates the fused kernels apply, or nothing (nested / other terms)"
fused_reg(r::L1Regularization) = (RLS_REG_L1, 1)
fused_reg(r::L2Regularization) = (RLS_REG_L2, 1)
fused_reg(r::L21Regularization) = (RLS_REG_L21, r.slices)
fused_reg(r::TVRegularization) = length(r.shape) <= 4 ? (RLS_REG_TV, 1) : nothing   # rls_fista_set_reg_tv (the FGP launch inside the plan)
fused_reg(r) = nothing
"rls_fista_set_reg / rls_fista_set_reg_tv for one plan (src/FISTA.jl:164-168: prox, then the projection).  `false`: the library refuses
the regulariser for this plan (a TV image that does not fit the single-workgroup FGP launch) -- the caller takes the reference's own path"
function fista_set_reg!(ctx, plan, reg, proj)
  kind, slices = fused_reg(reg)
  if kind == RLS_REG_TV
    sh = collect(Int64, reg.shape)
    d0 = collect(Int32, reg.dims isa Integer ? (reg.dims,) : reg.dims) .- Int32(1)
    st = ccall((:rls_fista_set_reg_tv, librls[]), Int32, (Ptr{Cvoid}, Float32, Int32, Ptr{Int64}, Int32, Ptr{Int32}, Int32, Int32),
               plan, Float32(λ(reg)), length(sh), sh, length(d0), d0, reg.iterationsTV, fused_proj(proj))
    st == -2 && return false                              # RLS_E_UNSUPPORTED
    check(ctx, st, "rls_fista_set_reg_tv")
  else
    check(ctx, ccall((:rls_fista_set_reg, librls[]), Int32, (Ptr{Cvoid}, Int32, Float32, Int64, Int32),
                     plan, kind, Float32(λ(reg)), slices, fused_proj(proj)), "rls_fista_set_reg")
  end
  true
end
function fused_proj(projs)
  isempty(projs) && return RLS_PROJ_NONE
  length(projs) == 1 || return nothing
  projs[1] isa PositiveRegularization && return RLS_PROJ_POSITIVE
  projs[1] isa RealRegularization && return RLS_PROJ_REAL
  nothing
end
"the operator handle behind A / AHA when the solver sits on this backend's types (matrix-free or explicit Gram)"
function operator_of(A, AHA)
  A isa RLSMatrix && AHA isa RLSNormalOp && AHA.A === A && return A.op
  # explicit Gram matrix from RLSMI355X.gram(A): its operator handle carries A and AHA (rls_operator_set_gram)
  A isa RLSMatrix && AHA isa RLSMatrix && size(AHA) == (A.N, A.N) && return AHA.op
  nothing
end

const fista_plans = IdDict{Any,Ptr{Cvoid}}()   # state => rls_fista plan

struct FistaStatus   # rls_fista_status, field for field
  iteration::Int32; done::Int32; theta::Float32; theta_old::Float32; rel_res_norm::Float32; residual::Float32; norm_x0::Float32
  fallbacks::Int32
end

function fista_plan_for(solver::FISTA, state::FISTAState{rT,<:RLSVector}) where {rT<:Float32}
  haskey(fista_plans, state) && return fista_plans[state]
  op = operator_of(solver.A, solver.AHA)
  (op === nothing || fused_reg(solver.reg) === nothing || fused_proj(solver.proj) === nothing) && return C_NULL
  p = Ref{Ptr{Cvoid}}(C_NULL)
  check(state.x.ctx, ccall((:rls_fista_create, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                           op, state.x.ptr, state.x₀.ptr, state.xᵒˡᵈ.ptr, state.res.ptr, p), "rls_fista_create")
  finalizer(_ -> ccall((:rls_fista_destroy, librls[]), Int32, (Ptr{Cvoid},), p[]), state)
  fista_plans[state] = p[]
end

function fista_status(state, plan)
  st = Ref{FistaStatus}()
  check(state.x.ctx, ccall((:rls_fista_get_status, librls[]), Int32, (Ptr{Cvoid}, Ref{FistaStatus}), plan, st), "rls_fista_get_status")
  st[]
end

function init!(solver::FISTA, state::FISTAState{rT,vecT}, b::vecT; x0 = 0, theta = 1) where {rT<:Float32,vecT<:RLSVector}
  plan = fista_plan_for(solver, state)
  # (configurations the fused kernels do not cover run the reference's own method.  `invoke` needs a signature that
  #  only the generic method matches: V ranges over every vector type, so this method -- V <: RLSVector -- is not it)
  plan == C_NULL && return invoke(init!, Tuple{FISTA,FISTAState{rT,V},V} where {V<:Union{AbstractVector{rT},AbstractVector{Complex{rT}}}},
                                  solver, state, b; x0, theta)
  ctx = state.x.ctx
  if !(solver.normalizeReg isa NoNormalization)      # the normalisation factor needs x0 = A'b before lambda is fixed (:128)
    mul!(state.x₀, adjoint(solver.A), b)
    solver.reg = normalize(solver, solver.normalizeReg, solver.reg, solver.A, state.x₀)
  end
  if !fista_set_reg!(ctx, plan, solver.reg, solver.proj)   # (a TV image too large for the plan's FGP launch)
    fista_plans[state] = C_NULL
    return invoke(init!, Tuple{FISTA,FISTAState{rT,V},V} where {V<:Union{AbstractVector{rT},AbstractVector{Complex{rT}}}},
                  solver, state, b; x0, theta)
  end
  check(ctx, ccall((:rls_fista_init, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Float32, Int32, Int32),
                   plan, b.ptr, state.ρ, Float32(theta), state.relTol, solver.iterations, solver.restart == :gradient), "rls_fista_init")
  if !all(x0 .== 0)                                   # warm start: state.x .= x0 (:120), scalar or vector
    xs = x0 isa RLSVector ? x0 : fill!(similar(state.x), x0)
    length(xs) == length(state.x) || throw(DimensionMismatch("x0 has length $(length(xs)), the solution $(length(state.x))"))
    check(ctx, ccall((:rls_fista_set_start, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), plan, xs.ptr, length(xs)), "rls_fista_set_start")
  end
  # the plan starts every solve with state.x in the vector it was created with; after a solve that ended on an odd
  # iteration count the host-side x / xold are swapped, so bring the roles back in line with the device (rls_fista_solution)
  st = fista_refresh!(state, plan)
  state.iteration = 0; state.norm_x₀ = st.norm_x0; state.theta = theta; state.thetaᵒˡᵈ = theta; state.rel_res_norm = rT(Inf)
  nothing
end

"host-side scalars and the x / xold roles after the device has advanced (`st`: a status already read, else read now)"
function fista_refresh!(state, plan, st = fista_status(state, plan))
  state.iteration = st.iteration; state.theta = st.theta; state.thetaᵒˡᵈ = st.theta_old
  st.iteration > 0 && (state.rel_res_norm = st.rel_res_norm)
  sol = Ref{Ptr{Cvoid}}(C_NULL)                        # the plan swaps x / xold by pointer, as :144-146 does
  check(state.x.ctx, ccall((:rls_fista_solution, librls[]), Int32, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), plan, sol), "rls_fista_solution")
  if state.x.ptr != sol[]
    state.x, state.xᵒˡᵈ = state.xᵒˡᵈ, state.x
  end
  st
end

function iterate(solver::FISTA, state::FISTAState{rT,<:RLSVector}) where {rT<:Float32}
  plan = get(fista_plans, state, C_NULL)
  plan == C_NULL && return invoke(iterate, Tuple{FISTA,FISTAState}, solver, state)
  done(solver, state) && return nothing
  st = Ref{FistaStatus}()   # step + status in one call: one host synchronisation per iteration
  check(state.x.ctx, ccall((:rls_fista_step_status, librls[]), Int32, (Ptr{Cvoid}, Int32, Ref{FistaStatus}), plan, 1, st), "rls_fista_step_status")
  fista_refresh!(state, plan, st[])
  solver.verbose && println("Iteration $(state.iteration); rel. residual = $(state.rel_res_norm)")
  return state.x, state
end

# ---- fused ADMM: whole outer iterations as a device plan (src/ADMM.jl:191-220, :230-322) -----------------------
struct AdmmParams
  x::Ptr{Cvoid}; xold::Ptr{Cvoid}; beta::Ptr{Cvoid}; beta_y::Ptr{Cvoid}; z0::Ptr{Cvoid}; z1::Ptr{Cvoid}; u::Ptr{Cvoid}
  rho::Float32; sigma_abs::Float32; rel_tol::Float32
  iterations::Int32; iterations_cg::Int32
  tol_inner::Float32
  reg_kind::Int32
  prox_lambda::Float32
  proj_kind::Int32
  tv_ndims::Int32; tv_ntv::Int32; tv_iterations::Int32
  tv_dims::NTuple{4,Int32}
  tv_shape::NTuple{4,Int64}
end
struct AdmmStatus   # rls_admm_status, field for field
  iteration::Int32; done::Int32; rk::Float32; sk::Float32; eps_pri::Float32; eps_dua::Float32; delta::Float32; cg_iterations::Int32
  fallbacks::Int32
end

const admm_plans = IdDict{Any,Any}()   # state => (cg plan, admm plan, z buffers)

admm_reg(r::L1Regularization) = RLS_REG_L1
admm_reg(r::L2Regularization) = RLS_REG_L2
admm_reg(r::TVRegularization) = length(r.shape) <= 4 ? RLS_REG_TV : nothing
admm_reg(r) = nothing
"regTrafo == identity (the constructor default opEye, src/ADMM.jl:84), probed through the public `*` only: Phi * (A'b) == A'b"
function is_identity_trafo(op, probe::RLSVector)
  size(op) == (length(probe), length(probe)) || return false
  t = op * probe
  t isa RLSVector && norm(t .- probe) == 0
end

function admm_fusable(solver::ADMM, state)
  # (a preconditioner other than IterativeSolvers.Identity() -- `precon`, src/ADMM.jl:82,244 -- is not part of the fused cg!: the
  #  reference's own iterate then runs IterativeSolvers' preconditioned loop on RLSVector through the BLAS-1 methods)
  length(solver.reg) == 1 && solver.vary_ρ == :none && nameof(typeof(solver.precon)) === :Identity &&
    operator_of(solver.A, solver.AHA) !== nothing &&
    admm_reg(solver.reg[1]) !== nothing && fused_proj(solver.proj) !== nothing &&
    !(solver.reg[1] isa TVRegularization && !isempty(solver.proj)) &&
    is_identity_trafo(solver.regTrafo[1], state.β_y)      # beta_y = A'b is set by the reference's init! just before
end

function admm_plan_for(solver::ADMM, state::ADMMState)
  haskey(admm_plans, state) && return admm_plans[state]
  admm_fusable(solver, state) || return nothing
  ctx = state.x.ctx
  cg = Ref{Ptr{Cvoid}}(C_NULL); pl = Ref{Ptr{Cvoid}}(C_NULL)
  sv = state.cgStateVars                               # CGStateVariables(u, r, c)   src/ADMM.jl:129
  check(ctx, ccall((:rls_cg_create, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                   operator_of(solver.A, solver.AHA), sv.u.ptr, sv.r.ptr, sv.c.ptr, cg), "rls_cg_create")
  check(ctx, ccall((:rls_admm_create, librls[]), Int32, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), cg[], pl), "rls_admm_create")
  finalizer(state) do _
    ccall((:rls_admm_destroy, librls[]), Int32, (Ptr{Cvoid},), pl[])
    ccall((:rls_cg_destroy, librls[]), Int32, (Ptr{Cvoid},), cg[])
  end
  admm_plans[state] = (cg = cg[], plan = pl[], zbuf = (state.z[1], state.zᵒˡᵈ[1]))
end

function init!(solver::ADMM, state::ADMMState{rT,rvecT,vecT}, b::vecT; x0 = 0) where {rT<:Float32,rvecT,vecT<:RLSVector}
  # the reference's own init! (:191-220), selected by a signature this method does not match (see the FISTA init! above)
  invoke(init!, Tuple{ADMM,ADMMState{rT,rvecT,V},V} where {V<:Union{AbstractVector{rT},AbstractVector{Complex{rT}}}}, solver, state, b; x0)
  P = admm_plan_for(solver, state)
  P === nothing && return nothing
  # The reference's init! has just written z = Phi x0 into state.z[1] -- whichever of the two buffers that is after the
  # previous solve (a solve that stopped on an odd outer-iteration count leaves the roles swapped).  The device plan
  # starts at parity 0 = z0, so z0 must be the buffer init! wrote: rebuild the pair from the CURRENT roles every time.
  P = merge(P, (zbuf = (state.z[1], state.zᵒˡᵈ[1]),))
  admm_plans[state] = P
  reg = solver.reg[1]
  ρ = Float32(state.ρ[1])
  tv = reg isa TVRegularization
  nd = tv ? length(reg.shape) : 0
  dims = tv ? collect(Int32, reg.dims isa Integer ? (reg.dims,) : reg.dims) .- Int32(1) : Int32[]
  pad4(v, T) = ntuple(i -> i <= length(v) ? T(v[i]) : zero(T), 4)
  prm = AdmmParams(state.x.ptr, state.xᵒˡᵈ.ptr, state.β.ptr, state.β_y.ptr, P.zbuf[1].ptr, P.zbuf[2].ptr, state.u[1].ptr,
                   ρ, state.σᵃᵇˢ, state.relTol, solver.iterations, solver.iterationsCG, state.tolInner,
                   ρ == 0 ? RLS_REG_NONE : admm_reg(reg), ρ == 0 ? 0f0 : Float32(λ(reg)) / (2f0 * ρ), fused_proj(solver.proj),
                   nd, length(dims), tv ? reg.iterationsTV : 0, pad4(dims, Int32), pad4(tv ? collect(reg.shape) : Int[], Int64))
  st = ccall((:rls_admm_init, librls[]), Int32, (Ptr{Cvoid}, Ref{AdmmParams}), P.plan, prm)
  if st == -2                                           # RLS_E_UNSUPPORTED: this regulariser runs through the generic iterate
    delete!(admm_plans, state); admm_plans[state] = nothing
  else
    check(state.x.ctx, st, "rls_admm_init")
  end
  nothing
end

function iterate(solver::ADMM, state::ADMMState{rT,rvecT,<:RLSVector}) where {rT<:Float32,rvecT}
  P = get(admm_plans, state, nothing)
  P === nothing && return invoke(iterate, Tuple{ADMM,ADMMState}, solver, state)
  done(solver, state) && return nothing
  st = Ref{AdmmStatus}()   # step + status in one call
  check(state.x.ctx, ccall((:rls_admm_step_status, librls[]), Int32, (Ptr{Cvoid}, Int32, Ref{AdmmStatus}, Ptr{Cvoid}, Int32), P.plan, 1, st, C_NULL, 0), "rls_admm_step_status")
  admm_refresh!(state, P, st)
  return state.x, state
end

function admm_refresh!(state, P, st = nothing)
  if st === nothing
    st = Ref{AdmmStatus}()
    check(state.x.ctx, ccall((:rls_admm_get_status, librls[]), Int32, (Ptr{Cvoid}, Ref{AdmmStatus}, Ptr{Cvoid}, Int32), P.plan, st, C_NULL, 0), "rls_admm_get_status")
  end
  state.iteration = st[].iteration
  if st[].iteration > 0
    state.rᵏ[1] = st[].rk; state.sᵏ[1] = st[].sk; state.ɛᵖʳⁱ[1] = st[].eps_pri; state.ɛᵈᵘᵃ[1] = st[].eps_dua; state.Δ[1] = st[].delta
  end
  cur = isodd(state.iteration) ? 2 : 1                  # z alternates between the two buffers, as the swap of :252-254
  state.z[1], state.zᵒˡᵈ[1] = P.zbuf[cur], P.zbuf[3 - cur]
  st[]
end

# ---- whole solves without a read-back per iteration ----------------------------------------------------------
"""
    solve_fused!(solver, b)

`init!` followed by ALL iterations enqueued in one call (`rls_*_step(plan, iterations)`: the device stops at the
iteration where `done` holds, exactly as the iterate-by-iterate loop of `solve!` does) and one status read-back.
`solve!(solver, b::RLSVector)` without callbacks lands here; with callbacks it stays the reference's loop.
"""
function RLSMI355X.solve_fused!(solver::Union{CGNR,FISTA,ADMM}, b::RLSVector{<:RLSSingle})
  init!(solver, b)
  state = solver.state
  if solver isa CGNR
    st = Ref{CgnrStatus}()
    check(b.ctx, ccall((:rls_cgnr_step_status, librls[]), Int32, (Ptr{Cvoid}, Int32, Ref{CgnrStatus}), plan_for(solver, state), solver.iterations, st), "rls_cgnr_step_status")
    cgnr_take!(state, st[])
    iterate(solver, state)   # done: applies `constr`, returns nothing
  elseif solver isa FISTA && get(fista_plans, state, C_NULL) != C_NULL
    check(b.ctx, ccall((:rls_fista_step, librls[]), Int32, (Ptr{Cvoid}, Int32), fista_plans[state], solver.iterations), "rls_fista_step")
    fista_refresh!(state, fista_plans[state])
  elseif solver isa ADMM && get(admm_plans, state, nothing) !== nothing
    check(b.ctx, ccall((:rls_admm_step, librls[]), Int32, (Ptr{Cvoid}, Int32), admm_plans[state].plan, solver.iterations), "rls_admm_step")
    admm_refresh!(state, admm_plans[state])
  else
    while iterate(solver, state) !== nothing end
  end
  return solver.state.x
end


# ---- K solvers, each with its own matrix, small enough for one CU each: ONE launch ---------------------------------------------
"""
    solve_group!(solvers::Vector{<:CGNR}, bs::Vector{<:RLSVector})

The reference's other multi-solve flavour (docs/src/literate/howto/multi_threading.jl:8-17: one solver and one A per problem under
`Threads.@threads`) for problems that each fit one CU's registers: `init!` and every iteration of all K problems as ONE launch, one
workgroup per problem (`rls_cgnr_init_step_group`).  Larger problems (any shape, any kernel path) run as a queue on the context's
stream with one read-back at the end (`rls_cgnr_solve_queue`).  The solvers must share the L2 weight, relTol and iteration count;
otherwise they are solved one after the other.
"""
function RLSMI355X.solve_group!(solvers::Vector{<:CGNR}, bs::Vector{<:RLSVector})
  length(solvers) == length(bs) || error("one right-hand side per solver")
  states = [s.state for s in solvers]
  plans = Ptr{Cvoid}[]
  for (s, st, b) in zip(solvers, states, bs)
    push!(plans, plan_for(s, st))         # (the plan is created on first use; init! itself runs inside the group launch)
    s.L2 = normalize(s, s.normalizeReg, s.L2, s.A, b)   # src/CGNR.jl:129 (a measurement-based factor differs per right-hand side)
    st.iteration = 0
    delete!(cgnr_done, st)
  end
  first_ = solvers[1]
  lam = Float32(λ(first_.L2)); tol = Float32(states[1].relTol)
  if !all(s -> Float32(λ(s.L2)) == lam && s.iterations == first_.iterations, solvers) || !all(st -> Float32(st.relTol) == tol, states)
    return [RLSMI355X.solve_fused!(s, b) for (s, b) in zip(solvers, bs)]   # one launch needs one lambda / relTol / iteration count
  end
  rc = ccall((:rls_cgnr_init_step_group, librls[]), Int32, (Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Int32, Float32, Float32, Int32, Int32),
             plans, Ptr{Cvoid}[b.ptr for b in bs], Int32(length(plans)), lam, tol, Int32(first_.iterations), Int32(first_.iterations))
  if rc == Int32(-2)
    # not all on the single-workgroup path: the queue -- problem k's init! and iterations enqueued behind problem k - 1's on the
    # context's stream, ONE read-back for all statuses (rls_cgnr_solve_queue; any shape, any kernel path)
    sts = Vector{CgnrStatus}(undef, length(plans))
    rc = ccall((:rls_cgnr_solve_queue, librls[]), Int32, (Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Int32, Float32, Float32, Int32, Ptr{CgnrStatus}),
               plans, Ptr{Cvoid}[b.ptr for b in bs], Int32(length(plans)), lam, tol, Int32(first_.iterations), sts)
    check(bs[1].ctx, rc, "rls_cgnr_solve_queue")
    for (s, st, stt) in zip(solvers, states, sts)
      cgnr_take!(st, stt)
      iterate(s, st)                      # done: applies `constr`, returns nothing
    end
    return [st.x for st in states]
  end
  check(bs[1].ctx, rc, "rls_cgnr_init_step_group")
  for (s, st) in zip(solvers, states)
    stt = Ref{CgnrStatus}()
    check(bs[1].ctx, ccall((:rls_cgnr_get_status, librls[]), Int32, (Ptr{Cvoid}, Ref{CgnrStatus}), plan_for(s, st), stt), "rls_cgnr_get_status")
    cgnr_take!(st, stt[])
    iterate(s, st)                        # done: applies `constr`, returns nothing
  end
  return [st.x for st in states]
end

# ---- matrix right-hand sides: the shared-A scheduler (BASELINE configs[3]; src/MultiThreading.jl:30-79) -------------------
# `solve!` on a device right-hand side WITHOUT callbacks: nothing observes the iterates, so the whole solve is enqueued at once
# (the reference's loop, src/RegularizedLeastSquares.jl:103-117, would synchronise with the host once per iteration for a
# `done` nobody else reads: 44 us against 11.7 us per CGNR iteration at 4096 x 2048 ComplexF32).  With callbacks, or with
# init! keywords (x0, ...), the reference's loop runs as written.
function RegularizedLeastSquares.solve!(solver::Union{CGNR,FISTA,ADMM}, b::RLSVector{<:RLSSingle}; callbacks = nothing, kwargs...)
  if callbacks === nothing && isempty(kwargs)
    RLSMI355X.solve_fused!(solver, b)
    return solversolution(solver)
  end
  cbs = callbacks === nothing ? Any[] : (callbacks isa Vector ? callbacks : Any[callbacks])
  init!(solver, b; kwargs...)
  foreach(cb -> cb(solver, 0), cbs)
  for (iteration, _) = enumerate(solver)
    foreach(cb -> cb(solver, iteration), cbs)
  end
  return solversolution(solver)
end

# `solve!(solver, B; scheduler = RLSMI355X.BatchedState)` with B an RLSMatrix (RLSMI355X.rhs(b)): the K columns advance
# TOGETHER through one plan (rls_cgnr_create_batched: both products of an iteration as skinny GEMMs on the matrix cores,
# A streamed once per product for all columns), each column with its own scalars and its own `done` -- the reference's
# per-column `active` retirement (src/MultiThreading.jl:60-78).  SequentialState / MultiThreadingState keep working on a
# device matrix through b[:, i], deepcopy(state) and hcat (RLSMI355X.jl): one plan per column.
mutable struct CgnrBatchedState{S, ST <: AbstractSolverState{S}} <: AbstractMatrixSolverState{S}
  states::Vector{ST}        # the single-column state this was built from (a later vector solve goes back to it)
  active::Vector{Bool}
  X::RLSMatrix; R::RLSMatrix; P::RLSMatrix; V::RLSMatrix    # N x K
  plan::Ptr{Cvoid}
  iteration::Int
end
"`scheduler = RLSMI355X.BatchedState`: the marker the matrix init! below looks for"
RLSMI355X.BatchedState(states::Vector) = error("BatchedState is selected with solve!(solver, B::RLSMatrix; scheduler = RLSMI355X.BatchedState)")

function init!(solver::CGNR, state::AbstractSolverState, B::RLSMatrix{Tc}; scheduler = RegularizedLeastSquares.SequentialState, x0 = 0, kwargs...) where {Tc}
  if scheduler !== RLSMI355X.BatchedState || solver.normalizeReg isa MeasurementBasedNormalization
    # (a measurement-based factor is a per-column lambda, src/CGNR.jl:129: the batched plan has one lambda for all columns)
    scheduler === RLSMI355X.BatchedState && (scheduler = RegularizedLeastSquares.SequentialState)
    # the reference's own matrix init! (src/MultiThreading.jl:30-38), selected by a signature this method does not match
    return invoke(init!, Tuple{AbstractLinearSolver,AbstractSolverState,AbstractMatrix}, solver, state, B; scheduler, x0, kwargs...)
  end
  all(x0 .== 0) || error("CGNR: x0 != 0 is unsupported (src/CGNR.jl:119)")
  single = state isa AbstractMatrixSolverState ? first(state.states) : state
  A = solver.A::RLSMatrix
  op = something(operator_of(A, solver.AHA), A.op)
  K, N, ctx = size(B, 2), A.N, A.ctx
  bs = state isa CgnrBatchedState && size(state.X) == (N, K) ? state : nothing
  if bs === nothing
    X, R, P, V = (RLSMatrix{Tc}(undef, N, K; ctx) for _ in 1:4)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, ccall((:rls_cgnr_create_batched, librls[]), Int32,
                     (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ref{Ptr{Cvoid}}),
                     op, K, X.ptr, R.ptr, P.ptr, V.ptr, N, p), "rls_cgnr_create_batched")
    bs = CgnrBatchedState{typeof(single).parameters[1], typeof(single)}([single], fill(true, K), X, R, P, V, p[], 0)
    finalizer(s -> ccall((:rls_cgnr_destroy, librls[]), Int32, (Ptr{Cvoid},), s.plan), bs)
  end
  check(ctx, ccall((:rls_cgnr_init_batched, librls[]), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float32, Float32, Int32),
                   bs.plan, B.ptr, B.lda, Float32(λ(solver.L2)), single.relTol, solver.iterations), "rls_cgnr_init_batched")
  bs.active .= true
  bs.iteration = 0
  solver.state = bs
  nothing
end

function iterate(solver::CGNR, state::CgnrBatchedState, activeIdx)
  K = length(state.active)
  st = Vector{CgnrStatus}(undef, K)
  ctx = state.X.ctx
  if state.iteration == 0   # columns that are done before they start (iterations == 0, a zero right-hand side)
    check(ctx, ccall((:rls_cgnr_get_status_batched, librls[]), Int32, (Ptr{Cvoid}, Ptr{CgnrStatus}), state.plan, st), "rls_cgnr_get_status_batched")
    for i in activeIdx
      st[i].done != 0 && (state.active[i] = false)
    end
    any(state.active) || return finish_batched!(solver, state)
  end
  check(ctx, ccall((:rls_cgnr_step, librls[]), Int32, (Ptr{Cvoid}, Int32), state.plan, 1), "rls_cgnr_step")
  check(ctx, ccall((:rls_cgnr_get_status_batched, librls[]), Int32, (Ptr{Cvoid}, Ptr{CgnrStatus}), state.plan, st), "rls_cgnr_get_status_batched")
  state.iteration += 1
  for i in activeIdx   # a column retires at the iteration where ITS `done` holds; the device skips it from then on
    st[i].done != 0 && (state.active[i] = false)
  end
  any(state.active) || finish_batched!(solver, state)
  return state.active, state
end
"constraints applied once, at exit, column by column (src/CGNR.jl:145-147)"
function finish_batched!(solver::CGNR, state::CgnrBatchedState)
  for r in solver.constr, j in 1:size(state.X, 2)
    # prox! on a column in place: a non-owning vector over the column's memory (kept alive by `state`)
    prox!(r, RLSMI355X.column_view(state.X, j))
  end
  nothing
end
solversolution(state::CgnrBatchedState) = state.X
iterate(solver::CGNR, state::CgnrBatchedState) = (idx = findall(state.active); isempty(idx) ? nothing : iterate(solver, state, idx))

# ---- a single oversized A, row-partitioned over the GPUs of one node (BASELINE configs[4]) ---------------------------------
"""
    RowSharded(comm, solvers)

`solvers[r]`: a CGNR / FISTA / ADMM built on rank r's row shard of A (`RLSMI355X.shard_operator(comm, a)`), all with the same
parameters.  `solve!(rs, b_parts)` runs the solver with the one distributed step of each operator apply -- the all-reduce of
the length-N partial product -- inside the library (rls_*_rowsharded: one host worker thread per rank); state vectors and
scalars are replicated, so any rank's solution is the solution.
"""
struct RowSharded{S}
  comm::Comm
  solvers::Vector{S}
end

function RegularizedLeastSquares.solve!(rs::RowSharded{<:CGNR}, b_parts::Vector{<:RLSVector})
  n = length(rs.comm)
  length(b_parts) == n == length(rs.solvers) || throw(DimensionMismatch("one solver and one slice of b per rank"))
  plans = Ptr{Cvoid}[plan_for(s, s.state) for s in rs.solvers]
  s1 = rs.solvers[1]
  ctx = rs.comm.ctxs[1]
  check(ctx, ccall((:rls_cgnr_init_rowsharded, librls[]), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Float32, Float32, Int32),
                   rs.comm.handle, plans, Ptr{Cvoid}[b.ptr for b in b_parts], Float32(λ(s1.L2)), s1.state.relTol, s1.iterations), "rls_cgnr_init_rowsharded")
  check(ctx, ccall((:rls_cgnr_step_rowsharded, librls[]), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Int32), rs.comm.handle, plans, s1.iterations), "rls_cgnr_step_rowsharded")
  for s in rs.solvers
    delete!(cgnr_done, s.state)
    while iterate(s, s.state) !== nothing end   # status read-back (done at once), constraints at exit
  end
  return s1.state.x
end

function RegularizedLeastSquares.solve!(rs::RowSharded{<:FISTA}, b_parts::Vector{<:RLSVector}; theta = 1)
  n = length(rs.comm)
  length(b_parts) == n == length(rs.solvers) || throw(DimensionMismatch("one solver and one slice of b per rank"))
  s1 = rs.solvers[1]
  plans = Ptr{Cvoid}[]
  for s in rs.solvers
    p = fista_plan_for(s, s.state)
    p == C_NULL && error("row-sharded FISTA: L1 / L2 / L21 / TV regularisation with at most one projection")
    fista_set_reg!(s.state.x.ctx, p, s.reg, s.proj) ||
      error("row-sharded FISTA + TV: the image does not fit the single-workgroup FGP kernel")
    push!(plans, p)
  end
  ctx = rs.comm.ctxs[1]
  check(ctx, ccall((:rls_fista_init_rowsharded, librls[]), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Float32, Float32, Float32, Int32, Int32),
                   rs.comm.handle, plans, Ptr{Cvoid}[b.ptr for b in b_parts], s1.state.ρ, Float32(theta), s1.state.relTol, s1.iterations,
                   s1.restart == :gradient), "rls_fista_init_rowsharded")
  check(ctx, ccall((:rls_fista_step_rowsharded, librls[]), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Int32), rs.comm.handle, plans, s1.iterations), "rls_fista_step_rowsharded")
  for (s, p) in zip(rs.solvers, plans)
    fista_refresh!(s.state, p)
  end
  return s1.state.x
end

"ADMM: every rank's solver has been through the reference's init! with ITS slice of b first (x, z, u, sigma_abs from the length of the whole b)"
function RegularizedLeastSquares.solve!(rs::RowSharded{<:ADMM}, b_parts::Vector{<:RLSVector}; M_total::Integer)
  n = length(rs.comm)
  length(b_parts) == n == length(rs.solvers) || throw(DimensionMismatch("one solver and one slice of b per rank"))
  for (s, b) in zip(rs.solvers, b_parts)
    s.state.σᵃᵇˢ = sqrt(eltype(s.state.ρ)(M_total)) * s.state.absTol   # sqrt(length(b)) of the WHOLE b   (src/ADMM.jl:214)
    init!(s, s.state, b)                                             # x = 0, z = Phi x, u = 0, the device plan's parameters
    get(admm_plans, s.state, nothing) === nothing && error("row-sharded ADMM: one L1 / L2 / TV term, identity regTrafo, vary_rho = :none")
  end
  plans = Ptr{Cvoid}[admm_plans[s.state].plan for s in rs.solvers]
  ctx = rs.comm.ctxs[1]
  check(ctx, ccall((:rls_admm_init_rowsharded, librls[]), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}), rs.comm.handle, plans,
                   Ptr{Cvoid}[b.ptr for b in b_parts]), "rls_admm_init_rowsharded")
  s1 = rs.solvers[1]
  check(ctx, ccall((:rls_admm_step_rowsharded, librls[]), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Int32), rs.comm.handle, plans, s1.iterations), "rls_admm_step_rowsharded")
  for s in rs.solvers
    admm_refresh!(s.state, admm_plans[s.state])
  end
  return s1.state.x
end

# ---- setup path: SystemMatrixBasedNormalization (ext/RegularizedLeastSquaresGPUArraysExt/NormalizedRegularization.jl:1-5)
function normalize(::SystemMatrixBasedNormalization, A::RLSMatrix{T}, b) where {T}
  M, N = size(A)
  e = RLSVector{real(T)}(undef, M; ctx = A.ctx)
  if T <: RLSDouble
    check(A.ctx, ccall((:rls_rownorm2_d, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
                       A.ctx.handle, dtypecode(T), M, N, A.ptr, M, e.ptr), "rls_rownorm2_d")
  else
    check(A.ctx, ccall((:rls_rownorm2, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
                       A.ctx.handle, dtypecode(T), M, N, A.ptr, M, e.ptr), "rls_rownorm2")
  end
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
    if T <: RLSDouble   # Float64 / ComplexF64: the same sweep on the double-precision entry points
      check(A.ctx, ccall((:rls_transpose_d, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64),
                         A.ctx.handle, dtypecode(T), M, N, A.ptr, M, At.ptr, N), "rls_transpose_d")
    else
      check(A.ctx, ccall((:rls_transpose, librls[]), Int32, (Ptr{Cvoid}, Int32, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64),
                         A.ctx.handle, dtypecode(T), M, N, A.ptr, M, At.ptr, N), "rls_transpose")
    end
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
    aux.rows[] = RLSVector(collect(reinterpret(Float32, Int32.(solver.rowindex[state.usedIndices] .- 1))); ctx = state.x.ctx)  # a Vector: 0-based Int32 row numbers, bit-cast
    aux.den[] = RLSVector(real(T).(solver.denom[state.usedIndices]); ctx = state.x.ctx)
    aux.key[] = key
  end
  A = solver.A::RLSMatrix{T}
  M, N = size(A)
  if T <: RLSDouble
    check(A.ctx, ccall((:rls_kaczmarz_sweep_d, librls[]), Int32,
                       (Ptr{Cvoid}, Int32, Int64, Int64, Ptr{Cvoid}, Int64, Int32, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64,
                        Ptr{Cvoid}, Ptr{Cvoid}, Int32, Float64, Int32),
                       A.ctx.handle, dtypecode(T), M, N, aux.At.ptr, N, 1, state.x.ptr, N, state.u.ptr, M, state.vl.ptr, M,
                       aux.rows[].ptr, aux.den[].ptr, length(state.usedIndices), Float64(real(state.ɛw)), 1), "rls_kaczmarz_sweep_d")
  else
    check(A.ctx, ccall((:rls_kaczmarz_sweep, librls[]), Int32,
                       (Ptr{Cvoid}, Int32, Int64, Int64, Ptr{Cvoid}, Int64, Int32, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64,
                        Ptr{Cvoid}, Ptr{Cvoid}, Int32, Float32, Int32),
                       A.ctx.handle, dtypecode(T), M, N, aux.At.ptr, N, 1, state.x.ptr, N, state.u.ptr, M, state.vl.ptr, M,
                       aux.rows[].ptr, aux.den[].ptr, length(state.usedIndices), Float32(real(state.ɛw)), 1), "rls_kaczmarz_sweep")
  end
  for r in solver.reg
    prox!(r, state.x)
  end
  state.iteration += 1
  return state.x, state
end

end # module
