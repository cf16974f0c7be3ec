# BazingaHIP.jl — the reference-side binding a Bazinga.jl maintainer would add.
#
# NOT EXECUTED — PARITY UNPINNED: there is no `julia` in the build container or on the GPU box, so this file has
# never run.  It is written against include/bazinga_hip.h and kept in sync with the ctypes binding
# bazinga.jl_amd/_lib.py, whose struct layouts ARE tested against the header (tests/test_abi.py) and which
# exercises every entry point used here (tests/, -m gpu).
#
# Usage — nothing else in user code changes:
#     using Bazinga, BazingaHIP
#     out = Bazinga.alps(f, g, c, D, x0, y0; subsolver = BazingaHIP.PANOCplus)      # seam alps.jl:24,64-66
# or, with device-resident outer loop:
#     out = BazingaHIP.alps(f, g, c, D, x0, y0)
module BazingaHIP

using Bazinga
import ProximalOperators

const lib = get(ENV, "BAZINGA_HIP_LIB", "libbazinga_hip.so")

# ---- mirrors of the C structs (include/bazinga_hip.h) ----------------------------------------
struct CtxOpts
    device::Int32; rank::Int32; nranks::Int32; flags::Int32; comm_id::Ptr{Cvoid}      # flags: BZ_CTX_RUNTIME_TUNING = 1, BZ_CTX_SHARED_DEVICE = 2
end
Base.@kwdef mutable struct ProblemDesc
    dtype::Int32 = 0; f_kind::Int32 = 0; g_kind::Int32 = 0; c_kind::Int32 = 0; D_kind::Int32 = 0
    slack::Int32 = 0
    n::Int64 = 0; ny::Int64 = 0
    f_q::Ptr{Cvoid} = C_NULL; f_b::Ptr{Cvoid} = C_NULL; f_grid_nx::Int64 = 0; f_grid_ny::Int64 = 0
    f_A::Ptr{Cvoid} = C_NULL; f_rows::Int64 = 0
    g_lambda::Float64 = 0; g_p::Float64 = 0; g_u::Ptr{Cvoid} = C_NULL; g_lo::Float64 = 0; g_hi::Float64 = 0
    g_lo_vec::Ptr{Cvoid} = C_NULL; g_hi_vec::Ptr{Cvoid} = C_NULL
    c_A::Ptr{Cvoid} = C_NULL; c_b::Ptr{Cvoid} = C_NULL
    D_lo::Float64 = 0; D_hi::Float64 = 0; D_lo_vec::Ptr{Cvoid} = C_NULL; D_hi_vec::Ptr{Cvoid} = C_NULL
    # generic oracles (all four kinds BZ_*_CALLBACK): host callbacks, see lower_generic!
    cb_user::Ptr{Cvoid} = C_NULL; cb_f_gradient::Ptr{Cvoid} = C_NULL; cb_g_prox::Ptr{Cvoid} = C_NULL
    cb_c_eval::Ptr{Cvoid} = C_NULL; cb_c_jtprod::Ptr{Cvoid} = C_NULL; cb_D_proj::Ptr{Cvoid} = C_NULL
end
Base.@kwdef mutable struct PanocOpts
    tol::Float64 = 1e-8; maxit::Int64 = 1000; freq::Int32 = 10; verbose::Int32 = 0
    minimum_gamma::Float64 = 1e-7; alpha::Float64 = 0.95; beta::Float64 = 0.5
    max_backtracks::Int32 = 20; lbfgs_memory::Int32 = 5; fuse::Int32 = 1; persist::Int32 = 1
    lbfgs_compact::Int32 = 2; affine_refresh::Int32 = 16
    directions::Int32 = 0; reserved::Int32 = 0; broyden_theta_bar::Float64 = 0.2      # BZ_DIR_LBFGS / _ANDERSON (1) / _BROYDEN (2)
    gamma::Float64 = 0; Lf::Float64 = 0; adaptive::Int32 = -1; reserved2::Int32 = 0  # 0 = nothing; adaptive -1 = (gamma === nothing)
end
Base.@kwdef mutable struct PanocStats
    iters::Int64 = 0; f_z::Float64 = 0; g_z::Float64 = 0; al_z::Float64 = 0; gamma::Float64 = 0
    tau::Float64 = 0; stop_norm::Float64 = 0; n_grad::Int64 = 0; n_prox::Int64 = 0
    n_backtracks::Int64 = 0; n_gamma_halvings::Int64 = 0; n_fused_iters::Int64 = 0
    n_lbfgs_skips::Int64 = 0; elapsed_s::Float64 = 0; status::Int32 = 0; persist_fallbacks::Int32 = 0
    n_affine_images::Int64 = 0; n_gated_launches::Int64 = 0; n_gate_aborts::Int64 = 0; n_gate_fallbacks::Int64 = 0
    n_dense_onepass::Int64 = 0; n_dense_fallbacks::Int64 = 0
end

Base.@kwdef mutable struct AlpsOpts
    tol_prim::Float64 = 1e-6; tol_dual::Float64 = 1e-6; inner_tol::Float64 = cbrt(1e-6)
    maxit::Int64 = 100; theta_penalty::Float64 = 0.8; kappa_penalty::Float64 = 0.5; kappa_tol::Float64 = 0.1
    subsolver_maxit::Int64 = 1_000_000_000; verbose::Int32 = 0; warm_start::Int32 = 0
end
Base.@kwdef mutable struct AlpsStats
    tot_it::Int64 = 0; tot_inner_it::Int64 = 0; elapsed_s::Float64 = 0; status::Int32 = 0; reserved::Int32 = 0
    inner_tol::Float64 = 0; norm_res_prim::Float64 = 0; objective::Float64 = 0
end

function check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:bz_last_error, lib), Cstring, ()))
    error(msg)                       # same ErrorException the reference throws (auglagfun.jl:33-34)
end
# library calls on a problem whose oracles are callbacks: an exception parked by a callback (BZ_ERR_CALLBACK) is rethrown
# as it was raised — the caller sees what the reference's own `gradient!` / `prox!` / `eval!` would have thrown
function check(rc::Cint, p)
    box = p.keep isa Tuple ? p.keep[end] : nothing
    if box !== nothing && box[].err !== nothing
        e = box[].err
        box[].err = nothing
        throw(e)
    end
    check(rc)
end

const _ctx = Ref{Ptr{Cvoid}}(C_NULL)
function context()
    if _ctx[] == C_NULL
        o = Ref(CtxOpts(0, 0, 1, 0, C_NULL))
        check(ccall((:bz_ctx_create, lib), Cint, (Ref{CtxOpts}, Ref{Ptr{Cvoid}}), o, _ctx))
    end
    _ctx[]
end

# ---- structured oracle types this build adds (SURVEY §8(b)) ------------------------------------
struct DiagQuadratic{T} <: Bazinga.ProximableFunction
    q::Vector{T}; b::Vector{T}            # f(x) = sum x_i (0.5 q_i x_i - b_i)
end

"f(x) = 0.5 x'A_h x - b'x, A_h the 5-point Laplacian (4,-1,-1,-1,-1) on an nx-by-ny grid, row-major, Dirichlet-0"
struct Stencil5ptQuadratic{T} <: Bazinga.ProximableFunction
    nx::Int; ny::Int; b::Vector{T}
end
"c(x) = A x - b with a dense A (demo/basispursuit.jl:38-49); `At` keeps A row-major for the device"
struct DenseAffine{T} <: Bazinga.SmoothFunction
    A::Matrix{T}; b::Vector{T}; At::Matrix{T}
    DenseAffine(A::Matrix{T}, b::Vector{T}) where {T} = new{T}(A, b, permutedims(A))
end
Bazinga.eval!(cx, c::DenseAffine, x) = (cx .= c.A * x .- c.b; nothing)
Bazinga.jtprod!(jtv, c::DenseAffine, x, v) = (jtv .= c.A' * v; nothing)

"`LBFGS(M; compact = nothing)`: how the operator is evaluated (bz_panoc_opts.lbfgs_compact) — `false` the two-loop recursion in the reference's order, `true` the compact representation, `nothing` (default) compact where the one-pass kernel applies"
struct LBFGS
    memory::Int; compact::Union{Nothing,Bool}
    LBFGS(memory = 5; compact = nothing) = new(memory, compact)
end

# ---- lowering: pattern-match the oracle structs -> bz_problem_desc ----------------------------
dtype_code(::Type{Float64}) = Int32(0)
dtype_code(::Type{Float32}) = Int32(1)

# Every lower_*! returns the temporaries whose memory the descriptor points into (or `nothing`): the Problem
# constructor roots them with GC.@preserve until bz_problem_create has copied the data.
lower_f!(d, f::Bazinga.Zero) = (d.f_kind = 0; nothing)
lower_f!(d, f::ProximalOperators.Zero) = (d.f_kind = 0; nothing)
lower_f!(d, f::DiagQuadratic) = (d.f_kind = 1; d.f_q = pointer(f.q); d.f_b = pointer(f.b); nothing)
lower_f!(d, f::Stencil5ptQuadratic) = (d.f_kind = 2; d.f_grid_nx = f.nx; d.f_grid_ny = f.ny; d.f_b = pointer(f.b); nothing)
# dense f: Julia matrices are column-major, the library wants row-major -> pass the transpose's memory
function lower_f!(d, f::ProximalOperators.LeastSquares)
    At = permutedims(f.A)
    d.f_kind = 3; d.f_A = pointer(At); d.f_rows = size(f.A, 1); d.f_b = pointer(f.b)
    return At                                                          # kept alive by the caller
end
lower_f!(d, f::ProximalOperators.Quadratic) = (d.f_kind = 4; d.f_A = pointer(f.Q); d.f_rows = size(f.Q, 1);
                                               d.f_b = pointer(f.q); nothing)      # Q symmetric: layout-agnostic
lower_f!(d, f) = :generic

lower_g!(d, g::Bazinga.Zero) = (d.g_kind = 0)
lower_g!(d, g::ProximalOperators.Zero) = (d.g_kind = 0)
lower_g!(d, g::ProximalOperators.IndFree) = (d.g_kind = 0)
lower_g!(d, g::ProximalOperators.NormL1{<:Real}) = (d.g_kind = 1; d.g_lambda = g.lambda)
lower_g!(d, g::Bazinga.NormL1Nonneg) = (d.g_kind = 2; d.g_lambda = g.lambda)
lower_g!(d, g::Bazinga.NormL1Box) = (d.g_kind = 3; d.g_lambda = g.lambda; d.g_u = pointer(g.u))
lower_g!(d, g::ProximalOperators.IndBox{<:Real,<:Real}) = (d.g_kind = 4; d.g_lo = g.lb; d.g_hi = g.ub)
lower_g!(d, g::Bazinga.NormL0Box) = (d.g_kind = 5; d.g_lambda = g.lambda; d.g_u = pointer(g.u))
lower_g!(d, g::Bazinga.NormLpPowerNonneg) = (d.g_kind = 6; d.g_lambda = g.alpha; d.g_p = g.p)
lower_g!(d, g::Bazinga.NormLpPowerBox) = (d.g_kind = 7; d.g_lambda = g.alpha; d.g_p = g.p; d.g_u = pointer(g.u))
lower_g!(d, g) = :generic

# c: any SmoothFunction whose eval!/jtprod! are the identity (e.g. test/definitions/identityFunction.jl)
abstract type IdentityLike <: Bazinga.SmoothFunction end
lower_c!(d, c::IdentityLike) = (d.c_kind = 0)
lower_c!(d, c::DenseAffine) = (d.c_kind = 1; d.c_A = pointer(c.At); d.c_b = pointer(c.b))
lower_c!(d, c) = :generic

lower_D!(d, D::Bazinga.ZeroSet) = (d.D_kind = 0)
lower_D!(d, D::Bazinga.FreeSet) = (d.D_kind = 1)
lower_D!(d, D::Bazinga.IndicatorSet{<:ProximalOperators.IndBox{<:Real,<:Real}}) =
    (d.D_kind = 2; d.D_lo = D.f.lb; d.D_hi = D.f.ub)
"""
    PairwiseSet(kind)   kind in (:vc, :cc, :eitheror, :xor)

The package's 2-element projections (`project_onto_VC_set!` etc.) applied to every adjacent pair
`(cx[2j-1], cx[2j])`, the way `demo/mpvca.jl:103-107` and `demo/eitheror.jl:121-131` define their sets.
"""
struct PairwiseSet <: Bazinga.ClosedSet
    kind::Symbol
end
function Bazinga.proj!(z, D::PairwiseSet, x)
    p! = D.kind === :vc ? Bazinga.project_onto_VC_set! : D.kind === :cc ? Bazinga.project_onto_CC_set! :
         D.kind === :eitheror ? Bazinga.project_onto_EITHEROR_set! : Bazinga.project_onto_XOR_set!
    for j in 1:2:length(x)
        p!(@view(z[j:j+1]), x[j:j+1])
    end
    return nothing
end
lower_D!(d, D::PairwiseSet) = (d.D_kind = Dict(:vc => 3, :cc => 4, :eitheror => 5, :xor => 6)[D.kind])
lower_D!(d, D) = :generic

# ---- generic oracles: host callbacks (BZ_*_CALLBACK) --------------------------------------------
# Whatever the structured types above do not cover — the closures of demo/rosenbrock.jl:39-80, any user type with
# the package's protocol (src/Bazinga.jl:11-16) — is handed to the library as @cfunction callbacks.  The host
# evaluates f / grad f, prox_g, c, J'v and proj_D; the L-BFGS and line-search vector work stays on the device.
# When one of the four is generic, all four travel as callbacks (the structured types have the protocol anyway).
mutable struct GenericOracles{F,G,C,DD,T}
    f::F; g::G; c::C; D::DD
    err::Any                     # an exception raised inside a callback, parked until the library call has returned
end
# A callback cannot unwind through the library's C frames: it parks the exception and asks the library to end the call
# in progress (bz_callback_abort -> BZ_ERR_CALLBACK); `check_generic` rethrows it on the Julia side.
function _cb_guard(body, o, default)
    o.err === nothing || (ccall((:bz_callback_abort, lib), Cvoid, ()); return default)
    try
        return body()
    catch e
        o.err = e
        ccall((:bz_callback_abort, lib), Cvoid, ())
        return default
    end
end
function _cb_f(u::Ptr{Cvoid}, x::Ptr{T}, dfx::Ptr{T}, n::Int64)::Float64 where {T}
    o = unsafe_pointer_to_objref(u)
    _cb_guard(o, NaN) do
        Float64(Bazinga.gradient!(unsafe_wrap(Array, dfx, n), o.f, unsafe_wrap(Array, x, n)))
    end
end
function _cb_g(u::Ptr{Cvoid}, x::Ptr{T}, gamma::Float64, z::Ptr{T}, n::Int64)::Float64 where {T}
    o = unsafe_pointer_to_objref(u)
    _cb_guard(o, NaN) do
        Float64(Bazinga.prox!(unsafe_wrap(Array, z, n), o.g, unsafe_wrap(Array, x, n), T(gamma)))
    end
end
function _cb_ceval(u::Ptr{Cvoid}, x::Ptr{T}, cx::Ptr{T}, n::Int64, ny::Int64)::Cvoid where {T}
    o = unsafe_pointer_to_objref(u)
    _cb_guard(o, nothing) do
        Bazinga.eval!(unsafe_wrap(Array, cx, ny), o.c, unsafe_wrap(Array, x, n)); nothing
    end
end
function _cb_cjt(u::Ptr{Cvoid}, x::Ptr{T}, v::Ptr{T}, jtv::Ptr{T}, n::Int64, ny::Int64)::Cvoid where {T}
    o = unsafe_pointer_to_objref(u)
    _cb_guard(o, nothing) do
        Bazinga.jtprod!(unsafe_wrap(Array, jtv, n), o.c, unsafe_wrap(Array, x, n), unsafe_wrap(Array, v, ny)); nothing
    end
end
function _cb_D(u::Ptr{Cvoid}, v::Ptr{T}, s::Ptr{T}, ny::Int64)::Cvoid where {T}
    o = unsafe_pointer_to_objref(u)
    _cb_guard(o, nothing) do
        Bazinga.proj!(unsafe_wrap(Array, s, ny), o.D, unsafe_wrap(Array, v, ny)); nothing
    end
end
function lower_generic!(d, f, g, c, D, ::Type{T}) where {T}
    box = Ref(GenericOracles{typeof(f),typeof(g),typeof(c),typeof(D),T}(f, g, c, D, nothing))     # rooted by the Problem
    d.f_kind = 5; d.g_kind = 8; d.c_kind = 2; d.D_kind = 7
    d.cb_user = pointer_from_objref(box[])
    d.cb_f_gradient = @cfunction(_cb_f, Float64, (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Int64))
    d.cb_g_prox = @cfunction(_cb_g, Float64, (Ptr{Cvoid}, Ptr{T}, Float64, Ptr{T}, Int64))
    d.cb_c_eval = @cfunction(_cb_ceval, Cvoid, (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Int64, Int64))
    d.cb_c_jtprod = @cfunction(_cb_cjt, Cvoid, (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Ptr{T}, Int64, Int64))
    d.cb_D_proj = @cfunction(_cb_D, Cvoid, (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Int64))
    return box
end

mutable struct Problem
    h::Ptr{Cvoid}
    keep::Any
    function Problem(f, g, c, D, n, ny, ::Type{T}) where {T}
        d = ProblemDesc(dtype = dtype_code(T), n = n, ny = ny)
        tmp = (lower_f!(d, f), lower_g!(d, g), lower_c!(d, c), lower_D!(d, D))      # temporaries the descriptor points into
        keep = any(t -> t === :generic, tmp) ? lower_generic!(d, f, g, c, D, T) : nothing
        h = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve f g c D tmp keep check(ccall((:bz_problem_create, lib), Cint,
            (Ptr{Cvoid}, Ref{ProblemDesc}, Ref{Ptr{Cvoid}}), context(), Ref(d), h))
        p = new(h[], (f, g, c, D, keep))      # generic oracles: the callback box lives as long as the problem
        finalizer(p -> ccall((:bz_problem_destroy, lib), Cvoid, (Ptr{Cvoid},), p.h), p)
    end
end

# ---- the subsolver seam: drop-in for ProximalAlgorithms.PANOCplus -----------------------------
struct PANOCplusHIP
    opts::PanocOpts
end
"""`PANOCplus(; directions=LBFGS(5), maxit, tol, verbose, freq, minimum_gamma, ...)` — same keywords
as ProximalAlgorithms.PANOCplus as the reference configures it (demo/rosenbrock.jl:109-115)."""
function PANOCplus(; directions = nothing, maxit = 1000, tol = 1e-8, verbose = false, freq = 10,
                   minimum_gamma = 1e-7, alpha = 0.95, beta = 0.5, max_backtracks = 20,
                   Lf = nothing, gamma = Lf === nothing ? nothing : alpha / Lf, adaptive = gamma === nothing, kwargs...)
    M = directions === nothing ? 5 : directions.memory
    compact = !(directions isa LBFGS) || directions.compact === nothing ? 2 : Int(directions.compact)
    PANOCplusHIP(PanocOpts(tol = tol, maxit = min(maxit, typemax(Int64)), freq = min(freq, typemax(Int32)),
                           verbose = verbose, minimum_gamma = minimum_gamma, alpha = alpha, beta = beta,
                           max_backtracks = max_backtracks, lbfgs_memory = M, lbfgs_compact = compact,
                           affine_refresh = get(kwargs, :affine_refresh, 16),
                           gamma = gamma === nothing ? 0.0 : gamma, Lf = Lf === nothing ? 0.0 : Lf, adaptive = Int32(adaptive)))
end

# one device problem per live AugLagFun; the entry (and, through the finalizer, the device buffers) goes with the functor
const _problems = WeakKeyDict{Any,Problem}()

"`solver(f = alFun, g = gFun, x0 = x) -> (sol, it)`   (alps.jl:66)"
function (s::PANOCplusHIP)(; f::Bazinga.AugLagFun, g::Bazinga.NonsmoothCostFun, x0::AbstractVector{T}) where {T}
    p = get!(() -> Problem(f.f, g.g, f.c, f.D, length(x0), length(f.y), T), _problems, f)
    mu = convert(Vector{T}, f.mu); y = convert(Vector{T}, f.y)
    check(ccall((:bz_problem_set_multipliers, lib), Cint, (Ptr{Cvoid}, Ptr{T}, Ptr{T}), p.h, mu, y), p)
    x = similar(x0); st = Ref(PanocStats())
    check(ccall((:bz_panoc_solve, lib), Cint, (Ptr{Cvoid}, Ref{PanocOpts}, Ptr{T}, Ptr{T}, Ref{PanocStats}),
                p.h, Ref(s.opts), x0, x, st), p)
    f.fx = T(st[].f_z)          # side channels alps reads back (alps.jl:68)
    g.gz = T(st[].g_z)
    g.gamma = st[].gamma
    return x, Int(st[].iters)
end

# ---- the whole outer loop with device-resident vectors (bz_alps_solve) --------------------------
const _status = (:first_order, :max_iter, :exception, :unknown)          # alps.jl:105-113

"""`alps(f, g, c, D, x0, y0; kw...)`: same keywords, defaults and 10-tuple as `Bazinga.alps` (alps.jl:14-25,115);
only scalars cross PCIe between subproblems.  `warm_start = true` (not a keyword of the reference; default `false` = alps.jl:64)
starts every subproblem after the first at the step size the previous one ended with."""
function alps(f, g, c, D, x0::AbstractVector{T}, y0::AbstractVector{T}; tol::Real = T(1e-6), tol_prim::Real = tol,
              tol_dual::Real = tol, inner_tol::Real = cbrt(tol_dual), maxit::Integer = 100,
              theta_penalty::Real = 0.8, kappa_penalty::Real = 0.5, kappa_tol::Real = 0.1, verbose::Bool = false,
              subsolver = PANOCplus, subsolver_maxit::Integer = 1_000_000_000, warm_start::Bool = false) where {T}
    p = Problem(f, g, c, D, length(x0), length(y0), T)
    ao = AlpsOpts(tol_prim = tol_prim, tol_dual = tol_dual, inner_tol = inner_tol, maxit = maxit,
                  theta_penalty = theta_penalty, kappa_penalty = kappa_penalty, kappa_tol = kappa_tol,
                  subsolver_maxit = subsolver_maxit, verbose = verbose, warm_start = Int32(warm_start))
    po = subsolver(tol = inner_tol, verbose = verbose).opts
    x = similar(x0); y = similar(y0); s = similar(y0); mu = similar(y0); st = Ref(AlpsStats())
    check(ccall((:bz_alps_solve, lib), Cint,
                (Ptr{Cvoid}, Ref{AlpsOpts}, Ref{PanocOpts}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{AlpsStats}),
                p.h, Ref(ao), Ref(po), x0, y0, x, y, s, mu, st), p)
    r = st[]
    return x, y, Int(r.tot_it), Int(r.tot_inner_it), r.elapsed_s, _status[r.status + 1], T(r.inner_tol),
           (r.tot_it == 0 ? nothing : T(r.norm_res_prim)), s, mu      # alps.jl:34,115: `nothing` before the first outer iteration
end

end # module
