# ClearSkyHIP.jl -- Julia-side glue for the MI355X line-by-line core (libclearsky_hip.so, include/clearsky_hip.h).
#
# SOURCE ONLY: no Julia toolchain exists in the build or GPU environments of this project, so this file has never
# been executed.  It shows exactly what a ClearSky.jl maintainer would add; the same C ABI is exercised end to end by
# the Python/ctypes host mirror (clearsky.jl_amd/core.py) and by tests/test_gpu_parity.py.
#
# What it adds to ClearSky.jl (reference paths in brackets):
#   * `HIPDiscretized <: ClearSky.AbstractNumericalCore`  [src/core/shared.jl:36,55-62] and a method of
#     `ClearSky.monochromaticfluxes!(M⁺, M⁻, τ, core::HIPDiscretized, P, g, T, μ, 𝒻S, 𝒻a, absorbers...; θₛ)`
#     [src/fluxes.jl:238-249] -- so `radiate!`, `fluxes`, `netfluxes`, `monochromaticfluxes`, `heating!` work unchanged
#     with `core=HIPDiscretized()`.
#   * `DirectGas <: ClearSky.AbstractGas`: a gas evaluated line-by-line at every (T,P) node instead of through baked
#     opacity tables ("Mode D"); equivalent to the function absorber (ν,T,P) -> C*voigt(ν, sl, T, P, C*P)
#     [src/absorption/absorbers.jl:16,24; src/absorption/line_shapes.jl:399-405].
#   * `hipvoigt!`, `hiplorentz!`, `hipdoppler!`, `hipPHCO2!`: drop-in `shape!` arguments of `Gas(sl, fC, ν, Ω, shape!, Δνcut)`
#     [src/absorption/gases.jl:225-231, invoked at :126], and `hipbake`, which evaluates all nT*nP states in ONE launch.
module ClearSkyHIP

using ClearSky
using ClearSky: AbstractNumericalCore, AbstractGas, SpectralLines, GrayGas, MOLPARAM, formprofiles,
                lobattoevaluations, lobattonodes, checkstreams, checkazimuth, ∫F!

const LIB = get(ENV, "CLEARSKY_HIP_LIB", joinpath(@__DIR__, "..", "clearsky.jl_amd", "csrc", "libclearsky_hip.so"))
const CHEB_LD = 16
const CS_MAX_GAS = 16        # gas slots per context (include/clearsky_hip.h)
const SHAPES = Dict(:voigt=>0, :lorentz=>1, :doppler=>2, :PHCO2=>3)

lasterror() = unsafe_string(ccall((:cs_last_error, LIB), Cstring, ()))
check(rc::Cint) = rc == 0 ? nothing : error("clearsky_hip ($rc): $(lasterror())")

#-------------------------------------------------------------------------------
# context: one per Julia thread (a context is not re-entrant and bake calls shape! from @threads, gases.jl:115)

mutable struct Context
    handle::Ptr{Cvoid}
    device::Int
    slots::IdDict{Any,Cint}            # SpectralLines objects (slot!) or the arguments of slotfrompar!
    order::Vector{Any}                 # keys of `slots`, oldest first (eviction order once all CS_MAX_GAS slots are taken)
    function Context(device::Integer=0)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:cs_create, LIB), Cint, (Cint, Ref{Ptr{Cvoid}}), device, h))
        c = new(h[], device, IdDict{Any,Cint}(), Any[])
        finalizer(x -> ccall((:cs_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.handle), c)
        return c
    end
end

# a free gas slot, or the oldest table's (same policy as the Python mirror's Context._take_slot; the library refuses slots >= CS_MAX_GAS)
function takeslot!(ctx::Context, key)::Cint
    if length(ctx.order) >= CS_MAX_GAS
        old = popfirst!(ctx.order)
        slot = ctx.slots[old]
        delete!(ctx.slots, old)
    else
        slot = Cint(length(ctx.order))
    end
    push!(ctx.order, key)
    ctx.slots[key] = slot
    return slot
end

# one context per (Julia thread, device): a context is not re-entrant, and `ngpu` devices take `ngpu` contexts
const CONTEXTS = Dict{Tuple{Int,Int},Context}()
const CTXLOCK = ReentrantLock()
context(device::Integer=parse(Int, get(ENV, "CLEARSKY_HIP_DEVICE", "0"))) = lock(CTXLOCK) do
    get!(() -> Context(device), CONTEXTS, (Threads.threadid(), Int(device)))
end

# upload a SpectralLines table (hitran/par.jl:224-251) + the MOLPARAM rows of its molecule, once per context
function slot!(ctx::Context, sl::SpectralLines)::Cint
    haskey(ctx.slots, sl) && return ctx.slots[sl]
    slot = takeslot!(ctx, sl)
    mp = MOLPARAM[sl.M]
    niso = length(mp.I)
    ncheb = Int32[mp.hascheb[i] ? mp.ncheb[i] : 0 for i in 1:niso]
    cheb = zeros(Float64, CHEB_LD, niso)            # column-major: [niso][CHEB_LD] in C
    for i in 1:niso, k in 1:length(mp.cheb[i])
        cheb[k,i] = mp.cheb[i][k]
    end
    check(ccall((:cs_gas_upload, LIB), Cint,
        (Ptr{Cvoid}, Cint, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Int16}, Cint, Ptr{Int32}, Ptr{Float64}),
        ctx.handle, slot, sl.N, sl.ν, sl.S, sl.γa, sl.γs, sl.Epp, sl.na, sl.μ, sl.I, niso, ncheb, cheb))
    return slot
end

# a .par file straight into a gas slot (f3): readpar's filters + SpectralLines' constructor on the native side
# [hitran/par.jl:91-193, 224-286]; returns the slot and the number of lines kept.  `M` is the molecule the file holds.
function slotfrompar!(ctx::Context, filename::String, M::Integer; νmin::Real=0, νmax::Real=Inf, Scut::Real=0, I=[], maxlines::Integer=-1)
    mp = MOLPARAM[M]
    niso = length(mp.I)
    ncheb = Int32[mp.hascheb[i] ? mp.ncheb[i] : 0 for i in 1:niso]
    cheb = zeros(Float64, CHEB_LD, niso)
    for i in 1:niso, k in 1:length(mp.cheb[i])
        cheb[k,i] = mp.cheb[i][k]
    end
    keep = Cint[x isa Char ? ClearSky.ISOINDEX[x] : x for x in I]
    slot = takeslot!(ctx, (filename, M, νmin, νmax, Scut, Tuple(keep), maxlines))
    L = Ref{Int64}(0)
    check(ccall((:cs_gas_upload_par, LIB), Cint,
        (Ptr{Cvoid}, Cint, Cstring, Cdouble, Cdouble, Cdouble, Ptr{Cint}, Cint, Int64, Cint, Ptr{Float64}, Cint, Ptr{Int32},
         Ptr{Float64}, Ref{Int64}),
        ctx.handle, slot, filename, νmin, min(νmax, 1e300), Scut, keep, length(keep), maxlines, M, mp.μ, niso, ncheb, cheb, L))
    return slot, L[]
end

# far-line sums on the matrix cores: 1 where the grid is long enough (default), 2 always, 0 never
matrixcores!(ctx::Context, on::Integer=1) = check(ccall((:cs_set_matrix_cores, LIB), Cint, (Ptr{Cvoid}, Cint), ctx.handle, on))

#-------------------------------------------------------------------------------
# B1: shape! operators [line_shapes.jl:412-424, :313-324, :200-211, :527-540]

function hipshape!(shape::Symbol, σ::AbstractVector{Float64}, ν::AbstractVector, sl::SpectralLines, T, P, Pₚ, Δνcut)
    ctx = context()
    νv = collect(Float64, ν)
    out = σ isa Vector{Float64} ? σ : Vector{Float64}(undef, length(σ))     # views of σ[:,i,j] are unit-stride: write straight in
    Tv, Pv, Pp = Float64[T], Float64[P], Float64[Pₚ]
    GC.@preserve νv out Tv Pv Pp begin
        dst = σ isa Vector{Float64} ? pointer(out) : (stride(σ,1) == 1 ? pointer(σ) : pointer(out))
        check(ccall((:cs_shape_batch, LIB), Cint,
            (Ptr{Cvoid}, Cint, Cint, Float64, Int64, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Ptr{Float64}, Int64),
            ctx.handle, slot!(ctx, sl), SHAPES[shape], Float64(Δνcut), length(νv), νv, 1, Tv, Pv, Pp, dst, length(νv)))
        if dst == pointer(out) && !(σ isa Vector{Float64})
            copyto!(σ, out)
        end
    end
    nothing
end

hipvoigt!(σ, ν, sl, T, P, Pₚ, Δνcut=25.0)   = hipshape!(:voigt,   σ, ν, sl, T, P, Pₚ, Δνcut)
hiplorentz!(σ, ν, sl, T, P, Pₚ, Δνcut=25.0) = hipshape!(:lorentz, σ, ν, sl, T, P, Pₚ, Δνcut)
hipdoppler!(σ, ν, sl, T, P, Pₚ, Δνcut=25.0) = hipshape!(:doppler, σ, ν, sl, T, P, Pₚ, Δνcut)
hipPHCO2!(σ, ν, sl, T, P, Pₚ, Δνcut=500.0)  = hipshape!(:PHCO2,   σ, ν, sl, T, P, Pₚ, Δνcut)

# all nT*nP states of bake [gases.jl:109-130] in one launch: σ[nν, nT, nP] is exactly [state][ν] with ld = nν
function hipbake!(σ::Array{Float64,3}, sl::SpectralLines, fC, ν::Vector{Float64}, Ω; shape::Symbol=:voigt, Δνcut=25.0)
    ctx = context()
    nT, nP = Ω.nT, Ω.nP
    T  = Float64[Ω.T[i] for i in 1:nT, j in 1:nP][:]
    P  = Float64[Ω.P[j] for i in 1:nT, j in 1:nP][:]
    Pp = Float64[fC(Ω.T[i], Ω.P[j])*Ω.P[j] for i in 1:nT, j in 1:nP][:]
    GC.@preserve σ ν T P Pp check(ccall((:cs_shape_batch, LIB), Cint,
        (Ptr{Cvoid}, Cint, Cint, Float64, Int64, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Int64),
        ctx.handle, slot!(ctx, sl), SHAPES[shape], Float64(Δνcut), length(ν), ν, nT*nP, T, P, Pp, σ, length(ν)))
    σ
end

#-------------------------------------------------------------------------------
# B2: a gas evaluated directly at the nodes

struct DirectGas{F} <: AbstractGas
    name::String
    formula::String
    μ::Float64
    ν::Vector{Float64}
    sl::SpectralLines
    fC::F
    shape::Symbol
    Δνcut::Float64
end

function DirectGas(sl::SpectralLines, fC, ν::AbstractVector{<:Real}; shape::Symbol=:voigt, Δνcut::Real=(shape == :PHCO2 ? 500 : 25))
    ClearSky.checkν(collect(Float64, ν))
    f = fC isa Real ? ((T,P)->float(fC)) : fC
    DirectGas(sl.name, sl.formula, sum(sl.A .* sl.μ)/sum(sl.A), collect(Float64, ν), sl, f, shape, Float64(Δνcut))
end

# scalar access keeps the reference semantics (σchain, absorbers.jl:84-92), e.g. for the Radau core
(g::DirectGas)(i::Int, T, P) = (C = g.fC(T,P); C*ClearSky.voigt(g.ν[i], g.sl, T, P, C*P, g.Δνcut))

#-------------------------------------------------------------------------------
# B3: numerical core dispatch [shared.jl:36,55-62; fluxes.jl:238-279]

struct HIPDiscretized <: AbstractNumericalCore
    nstream::Int64
    nlobatto::Int64
    ngpu::Int64     # devices 0 .. ngpu-1 of this node share the wavenumber grid (cs_fluxes_discretized_multi); 1: one device
end
HIPDiscretized(; nstream::Int=5, nlobatto::Int=2, ngpu::Int=1) = HIPDiscretized(nstream, nlobatto, ngpu)

function ClearSky.monochromaticfluxes!(M⁺::AbstractMatrix, M⁻::AbstractMatrix, τ::AbstractMatrix, core::HIPDiscretized,
                                       P::AbstractVector{<:Real}, g::Real, T, μ, 𝒻S, 𝒻a, absorbers...; θₛ::Real=0.841)::Nothing
    𝒜, ν, nν = ClearSky.unifyabsorbers(absorbers)
    𝒻T, 𝒻μ = formprofiles(P, T, μ)
    nstream, nlobatto = core.nstream, core.nlobatto
    @assert issorted(P) "pressure coordinates must be in ascending order (sorted)"
    # closures are evaluated here, exactly where fluxes.jl:253-267 / discretized.jl:11-30,46-58 evaluate them
    Tn, μn = lobattoevaluations(P, 𝒻T, 𝒻μ, nlobatto)
    Tlev = Float64[𝒻T(p) for p in P]
    checkstreams(nstream); checkazimuth(θₛ)
    np = length(P); nl = np - 1; K = nl*(nlobatto - 1) + 1
    𝓍, _ = lobattonodes(nlobatto)
    Pk = Vector{Float64}(undef, K); Tk = similar(Pk)
    Pk[1] = P[1]; Tk[1] = Tn[1,1]
    for i in 1:nl, n in 2:nlobatto
        k = (i-1)*(nlobatto-1) + n
        Pk[k] = n == nlobatto ? P[i+1] : P[i] + (P[i+1]-P[i])*𝓍[n]
        Tk[k] = Tn[n,i]
    end
    direct = filter(x -> x isa DirectGas, collect(𝒜.gas))
    gray   = filter(x -> x isa GrayGas, collect(𝒜.gas))
    length(direct) + length(gray) == length(𝒜.gas) || error("HIPDiscretized needs DirectGas / GrayGas members (baked Gas objects: see DESIGN.md, row f1)")
    # one context per device; every context holds the same tables in the same slots (same upload order on each)
    ctxs = [context(d) for d in 0:core.ngpu-1]
    ctx = ctxs[1]
    ngas = length(direct)
    for c in ctxs[2:end], x in direct
        slot!(c, x.sl)
    end
    slots  = Cint[slot!(ctx, x.sl) for x in direct]
    shapes = Cint[SHAPES[x.shape] for x in direct]
    cuts   = Float64[x.Δνcut for x in direct]
    conc   = Float64[direct[gi].fC(Tk[k], Pk[k]) for gi in 1:ngas, k in 1:K]      # [ngas, K] column-major
    σgray  = isempty(gray) ? 0.0 : sum(x.σ for x in gray)
    # functions σ(ν,T,P) and CIA objects are evaluated on the host, [nν, K]
    extra  = (isempty(𝒜.fun) && isempty(𝒜.cia)) ? nothing :
             Float64[ClearSky.σchain(𝒜.cia, ν[j], Tk[k], Pk[k]) + ClearSky.σchain(𝒜.fun, ν[j], Tk[k], Pk[k]) for j in 1:nν, k in 1:K]
    Stoa = Float64[𝒻S(x) for x in ν]; alb = Float64[𝒻a(x) for x in ν]
    F⁺ = Vector{Float64}(undef, np); F⁻ = similar(F⁺)
    dense(A) = (A isa Matrix{Float64}) ? A : Matrix{Float64}(undef, size(A))
    Mu, Md, Ta = dense(M⁺), dense(M⁻), dense(τ)
    handles = Ptr{Cvoid}[c.handle for c in ctxs]
    GC.@preserve ν P Tn μn Tlev slots shapes cuts conc extra Stoa alb Mu Md Ta F⁺ F⁻ handles begin
        if core.ngpu > 1      # ν cut into ngpu cost-balanced ranges, one per device; band fluxes added on the host in device order
            check(ccall((:cs_fluxes_discretized_multi, LIB), Cint,
                (Ptr{Ptr{Cvoid}}, Cint, Int64, Ptr{Float64}, Cint, Ptr{Float64}, Float64, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Float64, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                handles, core.ngpu, nν, ν, np, collect(Float64, P), Float64(g), nlobatto, Tn, μn, Tlev, ngas, slots, shapes, cuts, conc,
                σgray, extra === nothing ? C_NULL : pointer(extra), Stoa, alb, Float64(θₛ), nstream, Ta, Mu, Md, F⁺, F⁻))
        else
        check(ccall((:cs_fluxes_discretized, LIB), Cint,
            (Ptr{Cvoid}, Int64, Ptr{Float64}, Cint, Ptr{Float64}, Float64, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Float64, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
            ctx.handle, nν, ν, np, collect(Float64, P), Float64(g), nlobatto, Tn, μn, Tlev, ngas, slots, shapes, cuts, conc,
            σgray, extra === nothing ? C_NULL : pointer(extra), Stoa, alb, Float64(θₛ), nstream, Ta, Mu, Md, F⁺, F⁻))
        end
    end
    Mu === M⁺ || copyto!(M⁺, Mu); Md === M⁻ || copyto!(M⁻, Md); Ta === τ || copyto!(τ, Ta)
    nothing
end

export HIPDiscretized, DirectGas, hipvoigt!, hiplorentz!, hipdoppler!, hipPHCO2!, hipbake!

end # module

# ---------------------------------------------------------------------------------------------------------------------
# Addendum (same status: source only).  Baked gases and CIA through the resident-column entry points:
#
#   cs_bake(ctx, gas_slot, table_slot, shape, Δνcut, nν, ν, nT, Ω.T, nP, Ω.P, conc[nT,nP], lnσ_out_or_NULL)
#       replaces bake + OpacityTable (gases.jl:97-145, 75-82); the ln σ tables stay in HBM.  A `HIPGas <: AbstractGas`
#       wrapping (table_slot, Ω, fC, ν) then plays the role of `Gas` (gases.jl:205-249):
#         (g::HIPGas)(i, T, P) = g.fC(T,P) * rawσ via cs_table_eval(ctx, slot, T, P, i-1, 1, out)      # gases.jl:278
#   cs_cia_begin / cs_cia_band upload a CIATables object (collision_induced_absorption.jl:145-235): one call per Φ
#       (BilinearInterpolator grid: ν = Φ.G.x, T = Φ.G.y, ln k = Φ.G.Z) and per ϕ (single-temperature range, nt = 1).
#   monochromaticfluxes!(…, core::HIPDiscretized, …) with such members uses, instead of cs_fluxes_discretized:
#       cs_column_setup(…) ; cs_column_set_tables(ctx, ntab, slots, conc_tab[ntab,K]) ;
#       cs_column_set_cia(ctx, ncia, slots, flags, P1[ncia,K], P2[ncia,K]) ; cs_column_run(ctx, C_NULL) ;
#       cs_column_fetch(ctx, nν, np, τ, M⁺, M⁻, F⁺, F⁻)      # (nν, np: what the caller's arrays were allocated for; any of τ, M⁺, M⁻ may be C_NULL)
#   with conc_tab[t,k] = fC_t(T_k,P_k) and P1/P2 = P_k*concentration(g₁/g₂, T_k, P_k) (cia…jl:378-382), all evaluated on
#   the Julia side at the node states (T_k, P_k) built in the method above.
#   AcceleratedAbsorber / update! / Σ(A, i, T, P) (absorbers.jl:114-207), what RCM holds (radiative_convective.jl:95):
#       knots = a resident column over U's members with nlobatto = 2 on the knot pressures (node k = knot k = (T_k, P_k));
#       cs_accel_store(ctx, slot) evaluates ln Σ(U, i, T_k, P_k) for every ν and knot and keeps it in HBM; call it again after
#       cs_column_update_state(new T) = update!(A, T);  cs_accel_eval(ctx, slot, P, i-1, 1, out) = Σ(A, i, ·, P);
#       a column over A: cs_column_setup(ngas = 0, …) ; cs_column_set_accel(ctx, slot) ; cs_column_run(ctx, C_NULL) ;
#       cs_column_fetch(ctx, nν, np, τ, M⁺, M⁻, F⁺, F⁻)  (and cs_column_sigma_fetch(ctx, nν, K, σ) for the node cross-sections).
#       jacobian! (radiative_convective.jl:154-171): cs_column_batch(ctx, np+1, T_nodes, μ_nodes, T_levels, conc, conc_tab,
#       cia_P1, cia_P2, F⁺[np, B], F⁻[np, B]) evaluates all perturbed profiles side by side.
#   Scalar Σ(U, i, T, P) (absorbers.jl:95): cs_shape_points = the scalar-ν line-shape methods (inclusive cut-off) for DirectGas
#       members, cs_table_eval for baked ones; CIA and function members stay Julia calls.
#   This file has never been executed (no Julia toolchain on either box): tests/test_gpu_boundary.py drives the same symbol
#   with the same memory layout through ctypes.
#   cs_set_precision(ctx, 1, 1e6) selects the fp32 far-wing variant (BASELINE configs[4]).
#   The gases of a column that share shape and cut-off run as ONE merged line table (cs_set_merge, on by default): nothing changes
#   on the Julia side, conc stays [ngas, K] over the gases as named.
#   cs_set_interp(ctx, 0) switches the far-wing interpolation off (every (nu, line) pair evaluated, as surf! does); it is on
#   by default and exact to rounding (DESIGN.md section 3, K2c).
