# ConjugateGradientOptimAMD.jl — Julia host binding of the MI355X-native engine.
#
# Thin `ccall` layer over include/cgo.h (lib/libcgo_hip.so).  It re-creates the
# reference module's surface for the hot path — same names, same field names,
# same assertion behaviour — so that `examples/min.jl` runs unchanged except for
# the objective, which is a device descriptor instead of a Julia closure:
#
#     import ConjugateGradientOptimAMD as CGO
#     fdf! = CGO.Booth()                       # was: boothfdf! (examples/helpers/test_funcs.jl:3-12)
#     ls   = CGO.setupStrongWolfeBisection(1e-5, 0.8; a_max_growth_factor = 2.0,
#                                          max_iters = 1000, zoom_max_iters = 100)
#     cfg  = CGO.setupCGConfig(1e-5, CGO.HagerZhang(), CGO.EnableTrace(); max_iters = 1000)
#     ret  = CGO.minimizeobjective(fdf!, [0.43; 1.23], cfg, ls)
#     ret.minimizer, ret.objective, ret.status, ret.trace.objective_evals
#
# NOTE: there is no `julia` binary in the build image, so this file is written
# against the C ABI but has not been executed there; the Python binding
# (../api.py) exercises the identical symbols in the test-suite.
module ConjugateGradientOptimAMD

export TraceContainer, EnableTrace, DisableTrace, Results, LineSearchContainer, solvesystem, minimizeobjective  # ConjugateGradientOptim.jl:23-29

const libcgo = get(ENV, "CGO_LIB", joinpath(@__DIR__, "..", "lib", "libcgo_hip.so"))

# ---- src/types.jl:1-11 -------------------------------------------------------
abstract type LineSearchConfig end
abstract type βConfig end
abstract type CGβConfig <: βConfig end
abstract type QNβConfig <: βConfig end
abstract type TraceTrait end
struct EnableTrace <: TraceTrait end
struct DisableTrace <: TraceTrait end

# ---- βConfig subtypes (src/cg_flavours.jl) ------------------------------------
struct HagerZhang <: CGβConfig end
struct YuanWangSheng{T} <: CGβConfig
    μ::T
end
struct SallehAlhawarat <: CGβConfig end
struct LiuStorrey <: CGβConfig end
struct PolakRibiere <: CGβConfig end      # new
struct HestenesStiefel <: CGβConfig end   # new
struct DaiYuan <: CGβConfig end           # new
struct LBFGS <: QNβConfig                 # new
    m::Int
end
# qn_flavours.jl:53-66.  The reference's update solves s = B\y first (:81), so Bs = y, v = 0 and B_new = B up to
# rounding (:83-87): B stays the identity and u = B\(−g) is steepest descent for every θ.  The engine runs
# u = −g exactly; the N×N matrix is never built (include/cgo.h, CGO_BETA_BROYDEN_FAMILY).
struct BroydenFamily{T} <: QNβConfig
    θ::T
end
function setupBroydenFamily(θ::T, N::Int)::BroydenFamily{T} where {T<:AbstractFloat}
    @assert zero(T) <= θ
    return BroydenFamily(θ)
end


# C mirrors (include/cgo.h)
struct CBetaConfig
    kind::Int32
    lbfgs_m::Int32
    mu::Float64
end
struct CCGConfig
    eps::Float64
    beta::CBetaConfig
    max_iters::Int64
    verbose::Int32
    trace_enabled::Int32
end
struct CLSConfig
    kind::Int32
    cond_kind::Int32
    c1::Float64
    c2::Float64
    a_max_growth_factor::Float64
    delta1::Float64
    max_step_size::Float64
    max_iters::Int64
    zoom_max_iters::Int64
    feasibility_max_iters::Int64
    discount_factor::Float64
end
mutable struct CResults
    objective::Float64
    minimizer::Ptr{Float64}
    gradient::Ptr{Float64}
    iters_ran::Int64
    status::Int32
    _pad::Int32
    trace_objective::Ptr{Float64}
    trace_grad_norm::Ptr{Float64}
    trace_step_size::Ptr{Float64}
    trace_objective_evals::Ptr{Int64}
    total_fdf_evals::Int64
    total_launches::Int64
end

cbeta(::HagerZhang) = CBetaConfig(0, 0, 0.0)
cbeta(b::YuanWangSheng) = CBetaConfig(1, 0, Float64(b.μ))
cbeta(::SallehAlhawarat) = CBetaConfig(2, 0, 0.0)
cbeta(::LiuStorrey) = CBetaConfig(3, 0, 0.0)
cbeta(::PolakRibiere) = CBetaConfig(4, 0, 0.0)
cbeta(::HestenesStiefel) = CBetaConfig(5, 0, 0.0)
cbeta(::DaiYuan) = CBetaConfig(6, 0, 0.0)
cbeta(b::LBFGS) = CBetaConfig(7, Int32(b.m), 0.0)
cbeta(b::BroydenFamily) = CBetaConfig(8, 0, Float64(b.θ))

lasterror() = unsafe_string(ccall((:cgo_last_error, libcgo), Cstring, ()))
function check(rc::Cint)
    rc == 0 && return nothing
    msg = lasterror()
    # config @asserts of the reference surface as AssertionError (types.jl:187, nocedal.jl:22-26, wolfe.jl:233,278)
    startswith(msg, "AssertionError") && throw(AssertionError(msg[17:end]))
    error("cgo error $rc: $msg")
end

# ---- CGConfig (src/types.jl:156-203) -------------------------------------------
struct CGConfig{T,BT,ET}
    ϵ::T
    β_config::BT
    max_iters::Int
    verbose::Bool
    trace_status::ET
end
ccfg(c::CGConfig) = CCGConfig(Float64(c.ϵ), cbeta(c.β_config), c.max_iters, c.verbose, c.trace_status isa EnableTrace)

function setupCGConfig(ϵ::T, β_config::BT, trace_status::ET; max_iters = 1000, verbose = false) where {T<:AbstractFloat,BT<:βConfig,ET<:TraceTrait}
    cfg = CGConfig(ϵ, β_config, max_iters, verbose, trace_status)
    check(ccall((:cgo_check_cg_config, libcgo), Cint, (Ref{CCGConfig},), ccfg(cfg)))   # @assert 0 < ϵ < 1
    return cfg
end

# ---- line searches (src/linesearch/nocedal.jl:3-30, wolfe.jl:6-11,213-217,259-262) -----
struct StrongWolfeBisection{T} <: LineSearchConfig
    c1::T
    c2::T
    a_max_growth_factor::T
    max_iters::Int
    zoom_max_iters::Int
end
function setupStrongWolfeBisection(c1::T, c2::T; a_max_growth_factor::T = convert(T, 2), max_iters::Int = 1000, zoom_max_iters::Int = 100) where {T}
    ls = StrongWolfeBisection(c1, c2, a_max_growth_factor, max_iters, zoom_max_iters)
    check(ccall((:cgo_check_ls_config, libcgo), Cint, (Ref{CLSConfig},), cls(ls)))
    return ls
end
struct Wolfe{T}
    c1::T
    c2::T
end
struct YuanWeiLuWolfe{T}
    c1::T
    c2::T
    δ1::T
end
struct WolfeBisection{T,CT} <: LineSearchConfig
    condition::CT
    max_iters::Int
    max_step_size::T
    feasibility_max_iters::Int
end
struct Armijo{T}            # geometric.jl:159-162
    c1::T
end
struct Backtracking{T,CT} <: LineSearchConfig   # geometric.jl:15-20
    condition::CT
    discount_factor::T
    max_iters::Int
    feasibility_max_iters::Int
end
cls(l::StrongWolfeBisection) = CLSConfig(0, 0, l.c1, l.c2, l.a_max_growth_factor, 0.0, 0.0, l.max_iters, l.zoom_max_iters, 0, 0.0)
cls(l::WolfeBisection{T,Wolfe{T}}) where {T} = CLSConfig(1, 0, l.condition.c1, l.condition.c2, 2.0, 0.0, l.max_step_size, l.max_iters, 0, l.feasibility_max_iters, 0.0)
cls(l::WolfeBisection{T,YuanWeiLuWolfe{T}}) where {T} = CLSConfig(1, 1, l.condition.c1, l.condition.c2, 2.0, l.condition.δ1, l.max_step_size, l.max_iters, 0, l.feasibility_max_iters, 0.0)
cls(l::Backtracking{T,Armijo{T}}) where {T} = CLSConfig(2, 2, l.condition.c1, 0.0, 2.0, 0.0, 0.0, l.max_iters, 0, l.feasibility_max_iters, l.discount_factor)

# ---- Results / TraceContainer (src/types.jl:17-23,107-114) -------------------------------
struct TraceContainer{T,ET}
    objective::Vector{T}
    grad_norm::Vector{T}
    step_size::Vector{T}
    objective_evals::Vector{Int}
    status::ET
end
mutable struct Results{T,TrT}
    objective::T
    minimizer::Vector{T}
    gradient::Vector{T}
    iters_ran::Int
    status::Symbol
    trace::TrT
end

# ---- device context and objective descriptors ------------------------------------------------
mutable struct Context
    h::Ptr{Cvoid}
    function Context(device::Integer = 0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:cgo_ctx_create, libcgo), Cint, (Int32, Ref{Ptr{Cvoid}}), device, r))
        c = new(r[])
        finalizer(x -> ccall((:cgo_ctx_destroy, libcgo), Cint, (Ptr{Cvoid},), x.h), c)
        return c
    end
end
# cgo_solver_policy (include/cgo.h): HOW a solve runs — launches, kernels, hand-off protocols — never WHAT it computes.  Field for
# field the C struct (120 bytes); `SolverPolicy()` = library policy everywhere.  setdefaultpolicy!(ctx, p) makes every solver the
# context creates from then on — minimizeobjective, minimizeobjectivererun and solvesystem included — run with it.
Base.@kwdef struct SolverPolicy
    size::Int32 = Int32(120)
    points::Int32 = 0                  # trial steps per fused launch: 0 library policy | 1 | 3 | 5 | 7
    resident::Int32 = -1               # resident solver: -1 library policy | 0 | 1
    controller_depth::Int32 = -1       # on-device line-search controller: -1 | 0 host-driven | k rounds in flight
    controller_graph::Int32 = -1
    controller_fused::Int32 = -1
    stored_gradient::Int32 = 0         # 1: the stored-gradient k_fused family
    fused_tail::Int32 = -1             # context-wide
    strict_tail::Int32 = -1            # context-wide: formally fenced hand-offs
    placement_search::Int32 = -1       # opt-in buffer placement search at pure-HBM sizes
    placement_stages::Int32 = 0
    placement_max_bytes::Int64 = 0     # cap on the search's transient device memory
    lbfgs_form::Int32 = 0              # 1 one ring pass | 2 one pass + own state-update launch | 3 Gram, two passes | 4 chained two-loop
    lbfgs_fuse_grad::Int32 = -1
    lbfgs_fuse_trial::Int32 = -1
    lse_fixed_reference::Int32 = -1
    resident_points::Int32 = 0
    resident_chunk::Int32 = 0
    hbm_stream_bytes::Float64 = 0.0
    reserved::NTuple{8,Int32} = ntuple(_ -> Int32(0), 8)
end
@assert sizeof(SolverPolicy) == 120
function setdefaultpolicy!(ctx::Context, p::Union{Nothing,SolverPolicy})
    if p === nothing
        check(ccall((:cgo_ctx_set_default_policy, libcgo), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.h, C_NULL))
    else
        check(ccall((:cgo_ctx_set_default_policy, libcgo), Cint, (Ptr{Cvoid}, Ref{SolverPolicy}), ctx.h, p))
    end
    return ctx
end

const default_ctx = Ref{Union{Nothing,Context}}(nothing)
defaultcontext() = (default_ctx[] === nothing && (default_ctx[] = Context(0)); default_ctx[])

# Multi-GPU hosts (one Julia process per GPU; INTEGRATION.md §4): the shared-memory mailbox of one node.  Rank 0 attaches with
# create = true, the host synchronises (e.g. MPI.Barrier), the others attach, the host synchronises again; connectdevices! is
# COLLECTIVE and lets the GPUs exchange the blocks of controller-armed launches among themselves (hipIpc / xGMI).
function setcommshm!(ctx::Context, rank::Integer, world::Integer, name::String, create::Bool)
    check(ccall((:cgo_ctx_set_comm_shm, libcgo), Cint, (Ptr{Cvoid}, Int32, Int32, Cstring, Int32), ctx.h, rank, world, name, create ? 1 : 0))
    return ctx
end
function connectdevices!(ctx::Context)::Bool
    ok = Ref{Int32}(0)
    check(ccall((:cgo_ctx_comm_connect_devices, libcgo), Cint, (Ptr{Cvoid}, Ref{Int32}), ctx.h, ok))
    return ok[] == 1
end

mutable struct DeviceObjective
    h::Ptr{Cvoid}
    ctx::Context
    n::Int
    function DeviceObjective(h::Ptr{Cvoid}, ctx::Context, n::Int)
        o = new(h, ctx, n)
        finalizer(x -> ccall((:cgo_objective_destroy, libcgo), Cint, (Ptr{Cvoid},), x.h), o)
        return o
    end
    function DeviceObjective(kind::Integer, n::Integer, ctx::Context = defaultcontext())
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:cgo_objective_create, libcgo), Cint, (Ptr{Cvoid}, Int32, Int64, Int64, Int64, Ref{Ptr{Cvoid}}),
                    ctx.h, kind, n, 0, n, r))
        o = new(r[], ctx, n)
        finalizer(x -> ccall((:cgo_objective_destroy, libcgo), Cint, (Ptr{Cvoid},), x.h), o)
        return o
    end
end
"f(x) = ½ Σ D_i x_i²"
function QuadDiag(D::Vector{Float64}, ctx::Context = defaultcontext())
    o = DeviceObjective(0, length(D), ctx)
    check(ccall((:cgo_objective_set_param_host, libcgo), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), o.h, 0, D))
    return o
end
RosenbrockPaired(n::Integer, ctx::Context = defaultcontext()) = DeviceObjective(1, n, ctx)
"chained Rosenbrock (examples/helpers/test_funcs.jl:50-57) as a device stencil objective; n even"
RosenbrockChained(n::Integer, ctx::Context = defaultcontext()) = DeviceObjective(6, n, ctx)
"""
    ElementwiseObjective(n, source; param = nothing)

A user-supplied element-wise f/∇f: `source` is the HIP C++ body setting `fi` and `gi` from `x`, `p`,
`s0` (e.g. `"gi = p*x; fi = 0.5*(gi*x);"`), compiled at run time into the fused kernels — the device
counterpart of handing `minimizeobjective` your own `fdf!` closure (src/engine/optim.jl:25).
"""
function ElementwiseObjective(n::Integer, source::String; param::Union{Nothing,Vector{Float64}} = nothing, ctx::Context = defaultcontext(),
                              cheap::Bool = false)   # cheap: ≲ 10 flops per element → seven speculative trial steps per launch
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:cgo_objective_create_from_source, libcgo), Cint,
                (Ptr{Cvoid}, Cstring, Int32, Int64, Int64, Int64, Ref{Ptr{Cvoid}}),
                ctx.h, source, param === nothing ? 0 : 1, n, 0, n, r))
    o = DeviceObjective(r[], ctx, Int(n))
    param === nothing || check(ccall((:cgo_objective_set_param_host, libcgo), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), o.h, 0, param))
    cheap && check(ccall((:cgo_objective_set_cost_class, libcgo), Cint, (Ptr{Cvoid}, Int32), o.h, 1))
    return o
end
Booth(ctx::Context = defaultcontext()) = DeviceObjective(2, 2, ctx)
"f = fdf!(g, x) on host vectors — the reference's callback contract (src/cg_utils.jl:19)"
function (o::DeviceObjective)(g::Vector{Float64}, x::Vector{Float64})
    f = Ref{Float64}(0.0)
    check(ccall((:cgo_objective_eval_host, libcgo), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}), o.h, x, g, f))
    return f[]
end

statussymbol(s::Integer) = Symbol(unsafe_string(ccall((:cgo_status_name, libcgo), Cstring, (Int32,), s)))

function unpack(r::CResults, x, g, to, tg, ts, te, trace_status::ET) where {ET}
    k = trace_status isa EnableTrace ? Int(r.iters_ran) : 0
    tr = TraceContainer(to[1:k], tg[1:k], ts[1:k], Vector{Int}(te[1:k]), trace_status)
    return Results(r.objective, x, g, Int(r.iters_ran), statussymbol(r.status), tr)
end

# ---- src/engine/optim.jl:6-171 ----------------------------------------------------------------
function minimizeobjective(fdf!::DeviceObjective, x_initial::Vector{T}, config::CGConfig{T,BT,ET},
                           linesearch_config::LineSearchConfig) where {T<:AbstractFloat,BT<:βConfig,ET}
    T === Float64 || throw(MethodError(minimizeobjective, (fdf!, x_initial, config, linesearch_config)))  # reference is Float64-only (optim.jl:47)
    n = length(x_initial)
    n == fdf!.n || throw(DimensionMismatch("objective is $(fdf!.n)-dimensional, x_initial has length $n"))
    cap = max(config.max_iters, 1)
    x, g = Vector{Float64}(undef, n), Vector{Float64}(undef, n)
    to, tg, ts, te = zeros(cap), zeros(cap), zeros(cap), zeros(Int64, cap)
    r = CResults(0.0, pointer(x), pointer(g), 0, 0, 0, pointer(to), pointer(tg), pointer(ts), pointer(te), 0, 0)
    GC.@preserve x g to tg ts te begin
        check(ccall((:cgo_minimize, libcgo), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ref{CCGConfig}, Ref{CLSConfig}, Ref{CResults}),
                    fdf!.ctx.h, fdf!.h, x_initial, ccfg(config), cls(linesearch_config), r))
    end
    return unpack(r, x, g, to, tg, ts, te, config.trace_status)
end
# ---- the reference's own call form: an arbitrary Julia closure `f = fdf!(g, x)` (optim.jl:25, cg_utils.jl:19) ------
# `minimizeobjective(boothfdf!, x0, config, ls)` (examples/min.jl:41) works unchanged: the closure becomes a C callback
# (cgo_fdf_fn) through @cfunction; the solve is the GPU engine's (state, direction updates, dots, line-search state
# machine on the device), only x + a·u goes out to the host and ∇f comes back around each call of the closure.
function _fdf_trampoline(user::Ptr{Cvoid}, g::Ptr{Float64}, x::Ptr{Float64}, n::Int64)::Float64
    f! = unsafe_pointer_to_objref(user)[]
    return Float64(f!(unsafe_wrap(Array, g, n), unsafe_wrap(Array, x, n)))
end
"""
    HostObjective(fdf!, n)

Wraps a closure with the reference's contract `f = fdf!(g, x)` (writes ∇f into `g`, returns `f`) as an objective of the
GPU engine (`cgo_objective_create_callback`).  Built automatically by `minimizeobjective(fdf!, x0, …)` for any callable.
"""
function HostObjective(fdf!, n::Integer, ctx::Context = defaultcontext())
    box = Ref{Any}(fdf!)
    cb = @cfunction(_fdf_trampoline, Float64, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64))
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:cgo_objective_create_callback, libcgo), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Any, Int64, Int64, Int64, Ref{Ptr{Cvoid}}), ctx.h, cb, box, n, 0, n, r))
    o = DeviceObjective(r[], ctx, Int(n))
    HOST_ROOTS[o] = box          # keep the closure alive as long as the objective
    return o
end
const HOST_ROOTS = IdDict{Any,Any}()
function minimizeobjective(fdf!, x_initial::Vector{T}, config::CGConfig{T,BT,ET},
                           linesearch_config::LineSearchConfig) where {T<:AbstractFloat,BT<:βConfig,ET}
    o = HostObjective(fdf!, length(x_initial))
    try
        return minimizeobjective(o, x_initial, config, linesearch_config)
    finally
        delete!(HOST_ROOTS, o)
    end
end

# ---- src/engine/optim.jl:173-208 -----------------------------------------------------------------
# (Stage by stage through minimizeobjective, because every stage may carry another βConfig / LineSearchConfig TYPE; each stage's
#  minimizer crosses PCIe twice this way.  A host that wants the restart vector to stay on the GPU calls cgo_minimize_rerun
#  directly — one ccall, arrays of CCGConfig / CLSConfig, NULL minimizer buffers for the stages it does not read: include/cgo.h.
#  On a sharded context every rank runs the same chain on its own shard; the statuses, hence the stage count, are identical.)
function minimizeobjectivererun(fdf!, x_initial::Vector{T}, config::CGConfig{T,BT,ET},
                                linesearch_config::LineSearchConfig, rerun_config_tuples...) where {T<:AbstractFloat,BT<:βConfig,ET}
    rets = [minimizeobjective(fdf!, x_initial, config, linesearch_config)]
    for k in eachindex(rerun_config_tuples)
        rets[end].status != :success || return rets
        rerun_config, backup_linesearch_config = rerun_config_tuples[k]
        push!(rets, minimizeobjective(fdf!, rets[end].minimizer, rerun_config, backup_linesearch_config))
    end
    return rets
end

# ---- src/types.jl:84-100: exported by the reference; here only the host-side operand of evalϕdϕ! —
#      the engine's own work state (x, u) lives in HBM and the trial vectors never exist at all
struct LineSearchContainer{T}
    xp::Vector{T}
    df_xp::Vector{T}
    x::Vector{T}
    u::Vector{T}
end
LineSearchContainer(::Type{T}, N::Int) where {T<:AbstractFloat} =
    LineSearchContainer(Vector{T}(undef, N), Vector{T}(undef, N), Vector{T}(undef, N), Vector{T}(undef, N))

# ---- src/engine/solve_system.jl:6-27, 64-237 -------------------------------------------------------
struct LinesearchSolveSys{T}
    ρ::T # 0 < ρ < 1
    σ::T # σ > 0
    s::T # s > 0
    max_iters::Int
end
struct CLSSConfig
    rho::Float64; sigma::Float64; s::Float64; max_iters::Int64
end
function setupLinesearchSolveSys(s::T; σ = convert(T, 0.5), ρ = convert(T, 0.95), max_iters = round(Int, log(ρ, 1e-6))) where {T}
    @assert zero(T) < ρ < one(T)
    @assert ρ > zero(T)
    @assert s > zero(T)
    return LinesearchSolveSys(ρ, σ, s, max_iters)
end
# Restated bug for bug (include/cgo.h, cgo_solver_create_sys).  Where the reference throws
# UndefVarError (solve_system.jl:55) this returns its intended record, status :linesearch_failed.
function solvesystem(fdf!::DeviceObjective, x_initial::Vector{T}, config::CGConfig{T,BT,ET},
                     linesearch_config::LinesearchSolveSys{T}) where {T<:AbstractFloat,BT<:CGβConfig,ET}
    T === Float64 || throw(MethodError(solvesystem, (fdf!, x_initial, config, linesearch_config)))
    n = length(x_initial)
    n == fdf!.n || throw(DimensionMismatch("objective is $(fdf!.n)-dimensional, x_initial has length $n"))
    cap = max(config.max_iters, 1)
    x, g = Vector{Float64}(undef, n), Vector{Float64}(undef, n)
    to, tg, ts, te = zeros(cap), zeros(cap), zeros(cap), zeros(Int64, cap)
    r = CResults(0.0, pointer(x), pointer(g), 0, 0, 0, pointer(to), pointer(tg), pointer(ts), pointer(te), 0, 0)
    l = linesearch_config
    GC.@preserve x g to tg ts te begin
        check(ccall((:cgo_solvesystem, libcgo), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ref{CCGConfig}, Ref{CLSSConfig}, Ref{CResults}),
                    fdf!.ctx.h, fdf!.h, x_initial, ccfg(config), CLSSConfig(l.ρ, l.σ, l.s, l.max_iters), r))
    end
    return unpack(r, x, g, to, tg, ts, te, config.trace_status)
end

# ---- src/engine/primal_barrier.jl: a host-side caller of minimizeobjectivererun -----------------------
struct PrimalBarrierResults{T,TrT}            # :1-7
    centering_results::Vector{Vector{Results{T,TrT}}}
    status::Symbol
    iters_ran::Int
    t_final::T
    total_objective_evals::Int
end
struct PrimalBarrierConfig{T}                 # :130-136
    barrier_tol::T
    barrier_growth_factor::T
    max_iters::Int
    t_initial::T
    inf_f0_lb::T
end
setupPrimalBarrierConfig(barrier_tol::T, barrier_growth_factor::T, max_iters::Int; t_initial = convert(T, NaN)) where {T} =
    PrimalBarrierConfig(barrier_tol, barrier_growth_factor, max_iters, t_initial, zero(T))   # :138-154

# The box constraints of examples/constrained.jl:18-48 (`boxhdh!` + CvxInequalityConstraint, :38-60) as a
# device-side descriptor: ψ = −Σ log(−h_i) and ∇ψ (:70-94) become element-wise terms of the fused kernels.
struct BoxConstraints
    lb::Float64
    ub::Float64
end
hexlit(v::Float64) = isinf(v) ? (v > 0 ? "(1.0/0.0)" : "(-1.0/0.0)") : string(v)   # shortest round-trip decimal
function barrier_objective_source(base::String, box::BoxConstraints; barrier::Bool = true)
    name = strip(base) in ("ObjQuadDiag", "ObjRosenPaired", "ObjBooth") ? strip(base) : "BaseObjective"
    pre = name == "BaseObjective" ? base : ""
    barrier || return pre * """
struct UserObjective { using B = $name; static constexpr bool kParam = B::kParam; static constexpr bool kPairOnly = B::kPairOnly;
  __device__ static inline void eval1(double x, double p, double s0, double &f, double &g) { B::eval1(x, p, s0, f, g); }
  __device__ static inline void eval2(d2 xx, d2 pp, double s0, double &f, d2 &gg) { B::eval2(xx, pp, s0, f, gg); } };
"""
    return pre * """
struct UserObjective { using B = $name; static constexpr bool kParam = B::kParam; static constexpr bool kPairOnly = B::kPairOnly;
  __device__ static inline void bar(double x, double &psi, double &dpsi) {
    const double hu = x - ($(hexlit(box.ub))), hl = ($(hexlit(box.lb))) - x;
    const double cu = hu > 0.0 ? 0.0 : hu, cl = hl > 0.0 ? 0.0 : hl;        // clamp!(fi_evals, -Inf, 0)  :81
    psi = -(log(-cu) + log(-cl));                                            // :82
    double d = 0.0; d -= 1.0 / cu; d -= -1.0 / cl; dpsi = d; }               // :85-89
  __device__ static inline void eval1(double x, double p, double s0, double &f, double &g) {
    double f0 = 0.0, g0 = 0.0, psi, dpsi; B::eval1(x, p, 0.0, f0, g0); bar(x, psi, dpsi);
    f += s0 * f0 + psi; g = s0 * g0 + dpsi; }                                // evalbarrier!  :111-128
  __device__ static inline void eval2(d2 xx, d2 pp, double s0, double &f, d2 &gg) {
    double f0 = 0.0, psi0, psi1, d0, d1; d2 g0; B::eval2(xx, pp, 0.0, f0, g0); bar(xx.x, psi0, d0); bar(xx.y, psi1, d1);
    f += s0 * f0 + (psi0 + psi1); gg.x = s0 * g0.x + d0; gg.y = s0 * g0.y + d1; } };
"""
end
function setscalar!(o::DeviceObjective, v::Float64)
    check(ccall((:cgo_objective_set_scalar, libcgo), Cint, (Ptr{Cvoid}, Int32, Float64), o.h, 0, v))
end
# primalbarriermethod!(constraints, f0df0!, hdh!, x_initial, centering_config, linesearch_config, barrier_config,
# rerun_config_tuples...) (:156-255) with (constraints, hdh!) → BoxConstraints and f0df0! → device source.
# As in the reference, `x` is never updated (:172,:214-220): every centering step restarts from x_initial.
function primalbarriermethod!(constraints::BoxConstraints, f0df0::String, x_initial::Vector{Float64},
                              centering_config::CGConfig{Float64,BT,ET}, linesearch_config::LineSearchConfig,
                              barrier_config::PrimalBarrierConfig{Float64}, rerun_config_tuples...;
                              param::Union{Nothing,Vector{Float64}} = nothing) where {BT,ET}
    bc = barrier_config
    D = length(x_initial)
    rets = Vector{Vector{Results{Float64,TraceContainer{Float64,ET}}}}()
    assemble(status, it, t) = PrimalBarrierResults(rets[1:it], status, it, t,
        isempty(rets) ? 0 : sum(sum(sum(r.trace.objective_evals; init = 0) for r in rr; init = 0) for rr in rets[1:it]; init = 0))
    (any(x_initial .- constraints.ub .>= 0) || any(constraints.lb .- x_initial .>= 0)) && return assemble(:infeasible_start, 0, bc.t_initial)
    obj = ElementwiseObjective(D, barrier_objective_source(f0df0, constraints); param = param)
    t = bc.t_initial
    if !isfinite(t) || t < 0                                                   # verifyt0  :259-276
        base = ElementwiseObjective(D, barrier_objective_source(f0df0, constraints; barrier = false); param = param)
        t = (base(Vector{Float64}(undef, D), x_initial) - bc.inf_f0_lb) * bc.barrier_growth_factor
    end
    for i = 1:bc.max_iters
        setscalar!(obj, t)
        push!(rets, minimizeobjectivererun(obj, x_initial, centering_config, linesearch_config, rerun_config_tuples...))
        rets[end][end].status != :success && return assemble(:centering_step_issue, i, t)
        2 * D / t < bc.barrier_tol && return assemble(:success, i, t)
        t = bc.barrier_growth_factor * t
    end
    return assemble(:max_iters_reached, bc.max_iters, t)
end

# ---- the line-search conditions as scalar functions (wolfe.jl:219-294, geometric.jl:164-186) ----------
function evalwolfeconditions(condition::Union{Wolfe{Float64},YuanWeiLuWolfe{Float64}}, ϕ_a::Float64, dϕ_a::Float64, a::Float64,
                             u::Vector{Float64}, ϕ_0::Float64, dϕ_0::Float64)::Tuple{Bool,Bool}
    v1, v2 = Ref{Int32}(0), Ref{Int32}(0)
    ls = cls(WolfeBisection(condition, 1, 1.0, 1))
    check(ccall((:cgo_evalwolfeconditions, libcgo), Cint,
                (Ref{CLSConfig}, Float64, Float64, Float64, Float64, Float64, Float64, Ref{Int32}, Ref{Int32}),
                ls, ϕ_a, dϕ_a, a, sum(abs2, u), ϕ_0, dϕ_0, v1, v2))
    return v1[] != 0, v2[] != 0
end
function evalbacktrackcondition(condition::Armijo{Float64}, ϕ_a::Float64, a::Float64, ϕ_0::Float64, dϕ_0::Float64)::Bool
    v = Ref{Int32}(0)
    ls = cls(Backtracking(condition, 0.5, 1, 1))
    check(ccall((:cgo_evalbacktrackcondition, libcgo), Cint, (Ref{CLSConfig}, Float64, Float64, Float64, Float64, Ref{Int32}),
                ls, ϕ_a, a, ϕ_0, dϕ_0, v))
    return v[] != 0
end

# ---- kernel-level generics (src/cg_flavours.jl:2-15, 46-170; src/cg_utils.jl:4-23) ---------------------
function updatedir!(u::Vector{Float64}, df_x::Vector{Float64}, β::Float64; ctx::Context = defaultcontext())
    @assert length(u) == length(df_x)
    out = zeros(2)
    check(ccall((:cgo_kernel_dir, libcgo), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Int64, Ptr{Float64}),
                ctx.h, u, df_x, β, length(u), out))
    return nothing
end
function getβ(β_config::CGβConfig, g_next::Vector{Float64}, g::Vector{Float64}, u::Vector{Float64}; ctx::Context = defaultcontext())
    β = Ref{Float64}(0.0)
    check(ccall((:cgo_getbeta, libcgo), Cint, (Ptr{Cvoid}, Ref{CBetaConfig}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ref{Float64}),
                ctx.h, cbeta(β_config), g_next, g, u, length(u), β))
    return β[]
end
function evalϕdϕ!(xp::Vector{Float64}, df_xp::Vector{Float64}, fdf!::DeviceObjective, a::Float64, x::Vector{Float64}, u::Vector{Float64})
    out = zeros(2)
    check(ccall((:cgo_kernel_trial, libcgo), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}),
                fdf!.h, x, u, a, df_xp, out))
    for i in eachindex(x)
        xp[i] = x[i] + a * u[i]
    end
    return out[1], out[2]
end

end # module
