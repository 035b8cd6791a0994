# ClearSkyHIP.jl -- Julia-side glue for the MI355X line-by-line core (libclearsky_hip.so, include/clearsky_hip.h).
#
# SOURCE ONLY: no Julia toolchain exists in the build or GPU environments of this project, so this file has never
# been executed.  It shows exactly what a ClearSky.jl maintainer would add; the same C ABI is exercised end to end by
# the Python/ctypes host mirror (clearsky.jl_amd/core.py) and by tests/test_gpu_boundary.py, and every `ccall` below is
# checked statically -- symbol, arity, C type of each argument, number of values passed -- against the header by
# tests/test_julia_binding.py.
#
# What it adds to ClearSky.jl (reference paths in brackets):
#   * `HIPDiscretized <: ClearSky.AbstractNumericalCore`  [src/core/shared.jl:36,55-62] and a method of
#     `ClearSky.monochromaticfluxes!(M⁺, M⁻, τ, core::HIPDiscretized, P, g, T, μ, 𝒻S, 𝒻a, absorbers...; θₛ)`
#     [src/fluxes.jl:238-249] -- so `radiate!`, `radiate`, `fluxes`, `netfluxes`, `monochromaticfluxes` and `heating!`
#     [src/radiative_convective.jl:109-113] work unchanged with `core=HIPDiscretized()`, for every member an
#     `AbstractAbsorber` can hold (B2, SURVEY.md 8b):
#       - `DirectGas`   line-by-line at every (T,P) node                         -> cs_gas_upload + the line kernels
#       - `HIPGas`      baked on the device, tables resident in HBM               -> cs_bake
#       - `Gas`         the reference's own baked object [gases.jl:205-249]       -> cs_table_upload of its knot values
#       - `GrayGas`     [gases.jl:342-360]                                        -> sigma_gray
#       - `SemiGrayGas` [gases.jl:366-386]                                        -> a per-ν vector in sigma_extra
#       - `CIATables`   paired with any of the gases above by formula [cia...jl:431-465] -> cs_cia_begin / cs_cia_band
#       - functions σ(ν,T,P) [absorbers.jl:24,71]                                 -> evaluated here, sigma_extra
#       - `AcceleratedAbsorber` [absorbers.jl:114-203], what `RCM` holds and `heating!` passes -> cs_accel_upload, or
#         kept on the device by the `update!` method below                        -> cs_accel_store
#   * `DirectGas <: ClearSky.AbstractGas`: equivalent to the function absorber (ν,T,P) -> C*voigt(ν, sl, T, P, C*P)
#     [src/absorption/absorbers.jl:16,24; src/absorption/line_shapes.jl:399-405].
#   * `hipvoigt!`, `hiplorentz!`, `hipdoppler!`, `hipPHCO2!`: drop-in `shape!` arguments of `Gas(sl, fC, ν, Ω, shape!, Δνcut)`
#     [src/absorption/gases.jl:225-231, invoked at :126], and `hipbake!`, which evaluates all nT*nP states in ONE launch.
module ClearSkyHIP

using ClearSky
using ClearSky: AbstractNumericalCore, AbstractGas, AbstractAbsorber, SpectralLines, Gas, GrayGas, SemiGrayGas, CIATables, CIA,
                UnifiedAbsorber, AcceleratedAbsorber, AtmosphericDomain, MOLPARAM, formprofiles, lobattoevaluations,
                lobattonodes, checkstreams, checkazimuth, checkpressures, getwavenumbers, cia, TMIN, TMAX
import ClearSky: monochromaticfluxes!, radiate!, update!, concentration, rawσ, reconcentrate

const LIB = get(ENV, "CLEARSKY_HIP_LIB", joinpath(@__DIR__, "..", "clearsky.jl_amd", "csrc", "libclearsky_hip.so"))
const CHEB_LD = 16
const CS_MAX_GAS = 16        # gas slots per context (include/clearsky_hip.h)
const CS_MAX_TABLE = 16      # opacity-table slots
const CS_MAX_CIA = 8         # CIA slots
const CS_MAX_ACCEL = 4       # accelerated-absorber slots
const SHAPES = Dict(:voigt=>0, :lorentz=>1, :doppler=>2, :PHCO2=>3)

lasterror() = unsafe_string(ccall((:cs_last_error, LIB), Cstring, ()))
check(rc::Cint) = rc == 0 ? nothing : error("clearsky_hip ($rc): $(lasterror())")

#-------------------------------------------------------------------------------
# context: one per (Julia thread, device) -- a context is not re-entrant and bake calls shape! from @threads, gases.jl:115

mutable struct AccelEntry
    slot::Cint
    T::Vector{Float64}       # the temperatures the knots in HBM were evaluated at (A.T at that moment)
end

mutable struct Context
    handle::Ptr{Cvoid}
    device::Int
    slots::IdDict{Any,Cint}            # SpectralLines objects (slot!) or the arguments of slotfrompar!
    order::Vector{Any}                 # keys of `slots`, oldest first (eviction order once all CS_MAX_GAS slots are taken)
    tables::IdDict{Any,Cint}           # table keys (HIPGas.key, or the Π vector of a reference Gas) -> opacity-table slot
    torder::Vector{Any}
    cias::IdDict{Any,Cint}             # CIATables -> CIA slot
    corder::Vector{Any}
    accels::IdDict{Any,AccelEntry}     # AcceleratedAbsorber -> slot + the temperatures its knots in HBM belong to
    aorder::Vector{Any}
    function Context(device::Integer=0)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:cs_create, LIB), Cint, (Cint, Ref{Ptr{Cvoid}}), device, h))
        c = new(h[], device, IdDict{Any,Cint}(), Any[], IdDict{Any,Cint}(), Any[], IdDict{Any,Cint}(), Any[],
                IdDict{Any,AccelEntry}(), Any[])
        finalizer(x -> ccall((:cs_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.handle), c)
        return c
    end
end

# A free slot of a resource with `cap` slots, or the slot of the oldest key that the current call does not use (`keep`): a table
# is never evicted to make room for another table of the same call (both would name one slot and the first would silently be
# computed with the second's data).  All slots taken by the current call: an error.
function takeslot!(slots::IdDict, order::Vector{Any}, cap::Integer, key, keep)
    if length(order) >= cap
        idx = findfirst(k -> !any(x -> x === k, keep), order)
        idx === nothing && error("all $cap slots hold objects of the current call")
        old = order[idx]
        deleteat!(order, idx)
        slot = slots[old]
        delete!(slots, old)
    else
        used = Set(values(slots))
        slot = Cint(first(s for s in 0:cap-1 if !(Cint(s) in used)))
    end
    push!(order, key)
    slots[key] = slot
    return slot
end

# one context per (Julia thread, device): a context is not re-entrant, and `ngpu` devices take `ngpu` contexts
const CONTEXTS = Dict{Tuple{Int,Int},Context}()
const CTXLOCK = ReentrantLock()
context(device::Integer=parse(Int, get(ENV, "CLEARSKY_HIP_DEVICE", "0"))) = lock(CTXLOCK) do
    get!(() -> Context(device), CONTEXTS, (Threads.threadid(), Int(device)))
end

# MOLPARAM rows of a molecule as the C side takes them: ncheb[niso] (0 = no fit) and cheb[niso][CHEB_LD]
function molparamrows(M::Integer)
    mp = MOLPARAM[M]
    niso = length(mp.I)
    ncheb = Int32[mp.hascheb[i] ? mp.ncheb[i] : 0 for i in 1:niso]
    cheb = zeros(Float64, CHEB_LD, niso)            # column-major: [niso][CHEB_LD] in C
    for i in 1:niso, k in 1:length(mp.cheb[i])
        cheb[k,i] = mp.cheb[i][k]
    end
    return mp, niso, ncheb, cheb
end

function uploadlines!(ctx::Context, slot::Cint, sl::SpectralLines)
    _, niso, ncheb, cheb = molparamrows(sl.M)
    check(ccall((:cs_gas_upload, LIB), Cint,
        (Ptr{Cvoid}, Cint, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Int16}, Cint, Ptr{Int32}, Ptr{Float64}),
        ctx.handle, slot, sl.N, sl.ν, sl.S, sl.γa, sl.γs, sl.Epp, sl.na, sl.μ, sl.I, niso, ncheb, cheb))
    return slot
end

# upload a SpectralLines table (hitran/par.jl:224-251) + the MOLPARAM rows of its molecule, once per context.
# keep: the other line tables of the call this one belongs to
function slot!(ctx::Context, sl::SpectralLines; keep=())::Cint
    haskey(ctx.slots, sl) && return ctx.slots[sl]
    return uploadlines!(ctx, takeslot!(ctx.slots, ctx.order, CS_MAX_GAS, sl, keep), sl)
end

# the same table in a GIVEN slot (contexts of a multi-GPU call must number the column's tables alike: the C side takes ONE slot list)
function forceslot!(ctx::Context, sl::SpectralLines, slot::Cint)::Cint
    get(ctx.slots, sl, Cint(-1)) == slot && return slot
    if haskey(ctx.slots, sl)                       # held elsewhere on this context: drop that copy
        filter!(k -> k !== sl, ctx.order)
        delete!(ctx.slots, sl)
    end
    for k in collect(keys(ctx.slots))              # whatever sits in the wanted slot makes room
        if ctx.slots[k] == slot
            filter!(x -> x !== k, ctx.order)
            delete!(ctx.slots, k)
        end
    end
    push!(ctx.order, sl)
    ctx.slots[sl] = slot
    return uploadlines!(ctx, slot, sl)
end

# a .par file straight into a gas slot (f3): readpar's filters + SpectralLines' constructor on the native side
# [hitran/par.jl:91-193, 224-286]; returns the slot and the number of lines kept.  `M` is the molecule the file holds.
function slotfrompar!(ctx::Context, filename::String, M::Integer; νmin::Real=0, νmax::Real=Inf, Scut::Real=0, I=[], maxlines::Integer=-1)
    mp, niso, ncheb, cheb = molparamrows(M)
    keep = Cint[x isa Char ? ClearSky.ISOINDEX[x] : x for x in I]
    slot = takeslot!(ctx.slots, ctx.order, CS_MAX_GAS, (filename, M, νmin, νmax, Scut, Tuple(keep), maxlines), ())
    L = Ref{Int64}(0)
    check(ccall((:cs_gas_upload_par, LIB), Cint,
        (Ptr{Cvoid}, Cint, Cstring, Cdouble, Cdouble, Cdouble, Ptr{Cint}, Cint, Int64, Cint, Ptr{Float64}, Cint, Ptr{Int32},
         Ptr{Float64}, Ref{Int64}),
        ctx.handle, slot, filename, νmin, min(νmax, 1e300), Scut, keep, length(keep), maxlines, M, mp.μ, niso, ncheb, cheb, L))
    return slot, L[]
end

# fp32 far wings (BASELINE configs[4]): mode 1 with the x² threshold far_s ≥ 1e6; mode 0 = fp64 everywhere
precision!(ctx::Context, mode::Integer=0, far_s::Real=1e6) =
    check(ccall((:cs_set_precision, LIB), Cint, (Ptr{Cvoid}, Cint, Cdouble), ctx.handle, mode, far_s))
# far wings by spectral interpolation (on by default, exact to rounding); 0: every (ν, line) pair evaluated, as surf! does
interp!(ctx::Context, on::Integer=1) = check(ccall((:cs_set_interp, LIB), Cint, (Ptr{Cvoid}, Cint), ctx.handle, on))

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

# scalar-ν semantics of voigt(ν, sl, T, P, Pₚ, Δνcut) etc. [line_shapes.jl:399-405,290-296,177-183,514-520] mapped over ν:
# every line with |ν - νl| ≤ Δνcut counts (includedlines(::Real), :12-16) -- what a function absorber evaluates
function hipshapepoints(shape::Symbol, ν::Vector{Float64}, sl::SpectralLines, T::Vector{Float64}, P::Vector{Float64},
                        Pₚ::Vector{Float64}, Δνcut::Real)
    ctx = context()
    σ = Matrix{Float64}(undef, length(ν), length(T))
    GC.@preserve ν T P Pₚ σ check(ccall((:cs_shape_points, LIB), Cint,
        (Ptr{Cvoid}, Cint, Cint, Float64, Int64, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Int64),
        ctx.handle, slot!(ctx, sl), SHAPES[shape], Float64(Δνcut), length(ν), ν, length(T), T, P, Pₚ, σ, length(ν)))
    σ
end

#-------------------------------------------------------------------------------
# B2, members: a gas evaluated directly at the nodes

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

concentration(g::DirectGas, T, P) = g.fC(T,P)     # gases.jl:270

# scalar access keeps the reference semantics (σchain, absorbers.jl:84-92), e.g. for the Radau core
function (g::DirectGas)(i::Int, T, P)
    C = g.fC(T,P)
    f = g.shape == :voigt ? ClearSky.voigt : g.shape == :lorentz ? ClearSky.lorentz : g.shape == :doppler ? ClearSky.doppler : ClearSky.PHCO2
    C*f(g.ν[i], g.sl, T, P, C*P, g.Δνcut)
end

#-------------------------------------------------------------------------------
# B2, members: a gas baked on the device [bake gases.jl:97-145, OpacityTable :68-85, Gas :205-281]

mutable struct TableKey end     # identity of one set of tables (shared by reconcentrate'd copies, as g.Π is in the reference)

struct HIPGas{F} <: AbstractGas
    name::String
    formula::String
    μ::Float64
    ν::Vector{Float64}
    Ω::AtmosphericDomain
    fC::F                 # the concentration the gas is USED with (gases.jl:270,278)
    sl::SpectralLines
    fCbake::Function      # the concentration the tables were baked with (self-broadening, gases.jl:122-126); reconcentrate keeps it
    shape::Symbol
    Δνcut::Float64
    key::TableKey
end

function HIPGas(sl::SpectralLines, fC, ν::AbstractVector{<:Real}, Ω::AtmosphericDomain; shape::Symbol=:voigt, Δνcut::Real=(shape == :PHCO2 ? 500 : 25))
    @assert length(ν) > 0
    ν = collect(Float64, ν)
    ClearSky.checkν(ν)
    f = fC isa Real ? ((T,P)->float(fC)) : fC
    for P ∈ Ω.P, T ∈ Ω.T      # gases.jl:122-124
        C = f(T,P)
        @assert 0 <= C <= 1 "gas molar concentrations must be in [0,1], not $C (encountered @ $T K, $P Pa)"
    end
    HIPGas(sl.name, sl.formula, sum(sl.A .* sl.μ)/sum(sl.A), ν, Ω, f, sl, f, shape, Float64(Δνcut), TableKey())
end
HIPGas(filename::String, fC, ν, Ω; shape::Symbol=:voigt, Δνcut::Real=(shape == :PHCO2 ? 500 : 25), kwargs...) =
    HIPGas(SpectralLines(filename; kwargs...), fC, ν, Ω; shape=shape, Δνcut=Δνcut)

concentration(g::HIPGas, T, P) = g.fC(T,P)

# the table slot of a baked gas on this context, baked (HIPGas) or uploaded (reference Gas) on first use.
# keep: the table keys of the other baked members of the current call
function tableslot!(ctx::Context, g::HIPGas; keep=())::Cint
    haskey(ctx.tables, g.key) && return ctx.tables[g.key]
    slot = takeslot!(ctx.tables, ctx.torder, CS_MAX_TABLE, g.key, keep)
    Ω = g.Ω
    conc = Float64[g.fCbake(Ω.T[i], Ω.P[j]) for i in 1:Ω.nT, j in 1:Ω.nP]      # [nT, nP] column-major
    GC.@preserve conc check(ccall((:cs_bake, LIB), Cint,
        (Ptr{Cvoid}, Cint, Cint, Cint, Float64, Int64, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        ctx.handle, slot!(ctx, g.sl), slot, SHAPES[g.shape], g.Δνcut, length(g.ν), g.ν, Ω.nT, Ω.T, Ω.nP, Ω.P, conc, C_NULL))
    return slot
end

# The reference's own baked Gas: its tables live in Julia memory as one OpacityTable per wavenumber (gases.jl:68-85, field Π).  A
# Bichebyshev interpolant passes through its knots, so the knot values ln σ(ν_i, Ω.T[a], Ω.P[b]) -- Π[i].Φ(T, ln P) at the grid
# points -- are the whole table.  One-off, nν*nT*nP interpolator evaluations, parallel over ν like the reference's own loops.
function tableslot!(ctx::Context, g::Gas; keep=())::Cint
    haskey(ctx.tables, g.Π) && return ctx.tables[g.Π]
    slot = takeslot!(ctx.tables, ctx.torder, CS_MAX_TABLE, g.Π, keep)
    Ω = g.Ω
    nν = length(g.ν)
    lnσ = Array{Float64,3}(undef, nν, Ω.nT, Ω.nP)
    lnP = log.(Ω.P)
    Threads.@threads for i in 1:nν
        Φ = g.Π[i].Φ
        for b in 1:Ω.nP, a in 1:Ω.nT
            lnσ[i,a,b] = Φ(Ω.T[a], lnP[b])
        end
    end
    GC.@preserve lnσ check(ccall((:cs_table_upload, LIB), Cint,
        (Ptr{Cvoid}, Cint, Int64, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}),
        ctx.handle, slot, nν, g.ν, Ω.nT, Ω.T, Ω.nP, Ω.P, lnσ))
    return slot
end
tablekey(g::HIPGas) = g.key
tablekey(g::Gas) = g.Π

# rawσ(g, i, T, P), rawσ(g, T, P) [gases.jl:256-263] and the functors [:278-281]
function rawσ(g::HIPGas, T, P)
    ctx = context()
    out = Vector{Float64}(undef, length(g.ν))
    check(ccall((:cs_table_eval, LIB), Cint, (Ptr{Cvoid}, Cint, Float64, Float64, Int64, Int64, Ptr{Float64}),
        ctx.handle, tableslot!(ctx, g), Float64(T), Float64(P), 0, length(g.ν), out))
    out
end
function rawσ(g::HIPGas, i::Int, T, P)
    ctx = context()
    out = Vector{Float64}(undef, 1)
    check(ccall((:cs_table_eval, LIB), Cint, (Ptr{Cvoid}, Cint, Float64, Float64, Int64, Int64, Ptr{Float64}),
        ctx.handle, tableslot!(ctx, g), Float64(T), Float64(P), i - 1, 1, out))
    out[1]
end
(g::HIPGas)(i::Int, T, P) = concentration(g, T, P)*rawσ(g, i, T, P)
(g::HIPGas)(T, P) = concentration(g, T, P)*rawσ(g, T, P)

# opacityerror [gases.jl:152-175] for the table of wavenumber index i of a device-baked gas: Π = that wavenumber's table, Ω = g.Ω,
# sl = g.sl, ν = g.ν[i], C = the concentration the tables were baked with.  Same N × N grid and the same four results; the N² exact
# values are ONE cs_shape_points call (the scalar-ν shape at N² states), the interpolated ones cs_table_eval.
function ClearSky.opacityerror(g::HIPGas, i::Int, N::Int=50)
    T = collect(LinRange(g.Ω.Tmin, g.Ω.Tmax, N))
    P = 10 .^ collect(LinRange(log10(g.Ω.Pmin), log10(g.Ω.Pmax), N))
    Ts = Float64[T[a] for a in 1:N, b in 1:N][:]; Ps = Float64[P[b] for a in 1:N, b in 1:N][:]
    Pp = Float64[g.fCbake(Ts[k], Ps[k])*Ps[k] for k in 1:N*N]
    σex = reshape(hipshapepoints(g.shape, Float64[g.ν[i]], g.sl, Ts, Ps, Pp, g.Δνcut), N, N)
    σop = Float64[rawσ(g, i, T[a], P[b]) for a in 1:N, b in 1:N]
    aerr = σop .- σex
    return T, P, aerr, aerr./σex
end

# reconcentrate [gases.jl:292-320]: same tables (same key), new concentration function
function reconcentrate(g::HIPGas, fC)
    f = fC isa Real ? ((T,P)->float(fC)) : fC
    for P ∈ g.Ω.P, T ∈ g.Ω.T
        C = f(T,P)
        @assert 0 <= C <= 1.0 "gas molar concentrations must be in [0,1], not $C, which was encountered at T=$T P=$P"
    end
    HIPGas(g.name, g.formula, g.μ, g.ν, g.Ω, f, g.sl, g.fCbake, g.shape, g.Δνcut, g.key)
end

#-------------------------------------------------------------------------------
# B2, members: CIA pairs.  The reference's CIA(x, gases) [cia...jl:431-465] dispatches on ::Gas and UnifiedAbsorber hands it only
# the `Gas` members (absorbers.jl:67-69), so CIATables beside DirectGas / HIPGas members are paired here, by formula, with the same
# rules (findgas, :443-448) and the same arithmetic (the scalar path :378-382 -> :318-323 -> :251-276, :295-303).

const HIPLineGas = Union{DirectGas, HIPGas}
const PairGas = Union{DirectGas, HIPGas, Gas}     # everything with a formula and a concentration

struct HIPCIA{T,U}
    name::String
    formulae::Tuple{String,String}
    x::CIATables
    g₁::T
    g₂::U
end

function findpairgas(f::String, cianame::String, gases::Tuple)
    idx = findall(g -> g.formula == f, gases)
    @assert length(idx) > 0 "pairing failed for $cianame CIA, gas $f is missing"
    @assert length(idx) == 1 "pairing failed for $cianame CIA, duplicate $f gases found"
    return gases[idx[1]]
end

function HIPCIA(x::CIATables, gases::Tuple)
    isempty(gases) && error("no Gas objects provided, cannot create CIA object")
    f₁, f₂ = x.formulae
    HIPCIA(x.name, x.formulae, x, findpairgas(f₁, x.name, gases), findpairgas(f₂, x.name, gases))
end

(χ::HIPCIA)(ν, T, P) = cia(ν, χ.x, T, P, P*concentration(χ.g₁, T, P), P*concentration(χ.g₂, T, P))

# UnifiedAbsorber(absorbers) [absorbers.jl:50-77] for line-ups that hold DirectGas / HIPGas members: the same checks and the same
# struct (its type parameters are open), with the CIA pairs formed over ALL real gases.  Line-ups without such members fall through
# to the reference's method untouched.
const HIPInput = Union{AbstractGas, CIATables, Function}
function ClearSky.UnifiedAbsorber(absorbers::Tuple{Vararg{HIPInput}})
    any(a -> a isa HIPLineGas, absorbers) || return invoke(UnifiedAbsorber, Tuple{Tuple}, absorbers)
    @assert length(absorbers) > 0 "no absorbers... nothing to group"
    @assert length(absorbers) == length(unique(absorbers)) "duplicate absorbers"
    gas = Tuple(a for a in absorbers if a isa AbstractGas)
    realgas = Tuple(g for g in gas if g isa PairGas)                              # "real gases, ignoring Gray", absorbers.jl:67
    ciax = Tuple(HIPCIA(x, realgas) for x in absorbers if x isa CIATables)
    fun = Tuple(a for a in absorbers if !(a isa AbstractGas) && !(a isa CIATables))
    ν = getwavenumbers(gas...)
    UnifiedAbsorber(gas, ciax, fun, ν, length(ν))
end

# a CIATables object into a CIA slot [cia...jl:145-235]: one band per Φ (BilinearInterpolator of ln k on ν × T) and per ϕ
# (LinearInterpolator in ν at one temperature, used only with `singles`).  BasicInterpolators internals read here (v0.6/0.7 field
# names, the ones the reference itself reads at cia...jl:255-270 plus the sample arrays): Φ.G.x, Φ.G.y, Φ.Z [nν, nT]; ϕ.r.x, ϕ.y.
function ciaslot!(ctx::Context, x::CIATables; keep=())::Cint
    haskey(ctx.cias, x) && return ctx.cias[x]
    slot = takeslot!(ctx.cias, ctx.corder, CS_MAX_CIA, x, keep)
    nband = length(x.Φ) + length(x.ϕ)
    check(ccall((:cs_cia_begin, LIB), Cint, (Ptr{Cvoid}, Cint, Cint), ctx.handle, slot, nband))
    b = 0
    for Φ in x.Φ
        νb = collect(Float64, Φ.G.x); Tb = collect(Float64, Φ.G.y); lnk = Matrix{Float64}(Φ.Z)      # [nν, nT]: ν fastest
        check(ccall((:cs_cia_band, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}),
            ctx.handle, slot, b, length(νb), νb, length(Tb), Tb, lnk))
        b += 1
    end
    for (j, ϕ) in enumerate(x.ϕ)
        νb = collect(Float64, ϕ.r.x); Tb = Float64[x.T[j]]; lnk = collect(Float64, ϕ.y)
        check(ccall((:cs_cia_band, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}),
            ctx.handle, slot, b, length(νb), νb, 1, Tb, lnk))
        b += 1
    end
    return slot
end

#-------------------------------------------------------------------------------
# B2: the members of a UnifiedAbsorber at a set of node states, as the arrays the C side takes

struct Members
    slots::Vector{Cint}; shapes::Vector{Cint}; cuts::Vector{Float64}; conc::Matrix{Float64}        # line-by-line gases, [ngas, K]
    tslots::Vector{Cint}; conctab::Matrix{Float64}                                                 # baked gases, [ntab, K]
    cslots::Vector{Cint}; cflags::Vector{Cint}; P₁::Matrix{Float64}; P₂::Matrix{Float64}           # CIA pairs, [ncia, K]
    σgray::Float64
    extra::Union{Nothing,Matrix{Float64}}                                                          # functions σ(ν,T,P): [nν, K]
end

function members(ctx::Context, U::UnifiedAbsorber, Tk::Vector{Float64}, Pk::Vector{Float64})::Members
    K = length(Tk)
    ν = U.ν
    direct = [g for g in U.gas if g isa DirectGas]
    baked  = [g for g in U.gas if g isa Union{HIPGas,Gas}]
    gray   = [g for g in U.gas if g isa GrayGas]
    semi   = [g for g in U.gas if g isa SemiGrayGas]      # (ν[i] ≤ νcut) ? σ : 0 [gases.jl:366-386]: a per-ν vector at every node state
    length(direct) + length(baked) + length(gray) + length(semi) == length(U.gas) ||
        error("HIPDiscretized takes DirectGas, HIPGas, Gas, GrayGas and SemiGrayGas members (got $(map(typeof, U.gas)))")
    tables = [g.sl for g in direct]
    slots  = Cint[slot!(ctx, g.sl; keep=tables) for g in direct]
    shapes = Cint[SHAPES[g.shape] for g in direct]
    cuts   = Float64[g.Δνcut for g in direct]
    conc   = Float64[direct[gi].fC(Tk[k], Pk[k]) for gi in 1:length(direct), k in 1:K]             # [ngas, K] column-major
    tkeys  = [tablekey(g) for g in baked]
    tslots = Cint[tableslot!(ctx, g; keep=tkeys) for g in baked]
    conctab = Float64[concentration(baked[t], Tk[k], Pk[k]) for t in 1:length(baked), k in 1:K]    # gases.jl:270,278
    xs     = [χ.x for χ in U.cia]
    cslots = Cint[ciaslot!(ctx, χ.x; keep=xs) for χ in U.cia]
    cflags = Cint[(χ.x.extrapolate ? 1 : 0) | (χ.x.singles ? 2 : 0) for χ in U.cia]
    P₁ = Float64[Pk[k]*concentration(U.cia[c].g₁, Tk[k], Pk[k]) for c in 1:length(U.cia), k in 1:K]   # cia...jl:378-382
    P₂ = Float64[Pk[k]*concentration(U.cia[c].g₂, Tk[k], Pk[k]) for c in 1:length(U.cia), k in 1:K]
    σgray = isempty(gray) ? 0.0 : Float64(sum(g.σ for g in gray))
    extra = (isempty(U.fun) && isempty(semi)) ? nothing :
        Float64[ClearSky.σchain(U.fun, ν[j], Tk[k], Pk[k]) + sum(Float64[g(j) for g in semi]) for j in 1:length(ν), k in 1:K]
    Members(slots, shapes, cuts, conc, tslots, conctab, cslots, cflags, P₁, P₂, σgray, extra)
end

# node states of a column: k = (i-1)(nlobatto-1) + n  [discretized.jl:150,162,169]
function nodestates(P::AbstractVector, Tn::Matrix, nlobatto::Int)
    np = length(P); nl = np - 1; K = nl*(nlobatto - 1) + 1
    𝓍, _ = lobattonodes(nlobatto)
    Pk = Vector{Float64}(undef, K); Tk = similar(Pk)
    Pk[1] = P[1]; Tk[1] = Tn[1,1]
    for i in 1:nl, n in 2:nlobatto
        k = (i-1)*(nlobatto-1) + n
        Pk[k] = n == nlobatto ? P[i+1] : P[i] + (P[i+1]-P[i])*𝓍[n]
        Tk[k] = Tn[n,i]
    end
    return Pk, Tk
end

# checkpressures [absorbers.jl:101,209,237-246; called at fluxes.jl:265]: the reference's check covers its `Gas` members; HIPGas
# members carry the same kind of domain
function hipcheckpressures(𝒜, Pₛ, Pₜ)
    checkpressures(𝒜, Pₛ, Pₜ)
    U = 𝒜 isa AcceleratedAbsorber ? 𝒜.U : 𝒜
    for g in U.gas
        if g isa HIPGas
            for P ∈ (Pₛ, Pₜ)
                @assert P >= g.Ω.Pmin "Pressure $P Pa too low, domain minimum is $(g.Ω.Pmin)"
                @assert P <= g.Ω.Pmax "Pressure $P Pa too low, domain minimum is $(g.Ω.Pmax)"
            end
        end
    end
end

#-------------------------------------------------------------------------------
# B2: AcceleratedAbsorber [absorbers.jl:114-203] -- what RCM holds (radiative_convective.jl:18,95) and heating! hands to radiate! (:113)

# BasicInterpolators' LinearInterpolator keeps its samples in the field `y` (what `ϕ[idx] = v`, absorbers.jl:195, writes)
knotvalues(A::AcceleratedAbsorber) = Float64[A.ϕ[i].y[k] for i in 1:A.nν, k in 1:length(A.P)]       # [nν, nk]: ν fastest

# the slot holding A's knots in HBM, brought up to date if A was updated on the host since (update! records the temperatures in A.T,
# absorbers.jl:198: unchanged temperatures on an unchanged A.U = unchanged knots)
function accelslot!(ctx::Context, A::AcceleratedAbsorber)::Cint
    e = get(ctx.accels, A, nothing)
    e !== nothing && e.T == A.T && return e.slot
    if e === nothing
        slot = takeslot!(ctx.accels, ctx.aorder, CS_MAX_ACCEL, A, ())
        e = AccelEntry(slot, Float64[])
        ctx.accels[A] = e
    end
    lnσ = knotvalues(A)
    GC.@preserve lnσ check(ccall((:cs_accel_upload, LIB), Cint,
        (Ptr{Cvoid}, Cint, Int64, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}),
        ctx.handle, e.slot, A.nν, A.ν, length(A.P), collect(Float64, A.P), lnσ))
    e.T = collect(Float64, A.T)
    return e.slot
end
# (takeslot! stores a Cint under the key; the registry of accelerated absorbers stores entries)
function takeslot!(slots::IdDict{Any,AccelEntry}, order::Vector{Any}, cap::Integer, key, keep)
    if length(order) >= cap
        old = popfirst!(order)
        slot = slots[old].slot
        delete!(slots, old)
    else
        used = Set(e.slot for e in values(slots))
        slot = Cint(first(s for s in 0:cap-1 if !(Cint(s) in used)))
    end
    push!(order, key)
    return slot
end

# update!(A, T) [absorbers.jl:173-200] on the device for absorbers with DirectGas / HIPGas members: Σ(A.U, i, T_k, P_k) for every ν
# and knot in ONE pass of the kernels -- a resident column whose node k is knot k (nlobatto = 2, levels = knots; g, μ, 𝒻S, 𝒻a play no
# role in the cross-sections) + cs_accel_store -- instead of nν × nk scalar Σ calls.  The knots stay in HBM for the flux calls that
# follow, and are copied back into A.ϕ so that the reference's own Σ(A, i, ·, P) and A(P) keep working.  Absorbers made only of
# reference members keep the reference's method.
const HIPCapable = UnifiedAbsorber{<:Tuple{Vararg{Union{DirectGas,HIPGas,Gas,GrayGas,SemiGrayGas}}}}
function update!(A::AcceleratedAbsorber{V,Q}, T::AbstractVector)::Nothing where {V,Q<:HIPCapable}
    any(g -> g isa HIPLineGas, A.U.gas) || return invoke(update!, Tuple{AcceleratedAbsorber,AbstractVector}, A, T)
    @assert length(T) == length(A.P)
    ctx = context()
    nk = length(A.P); nν = A.nν
    Pk = collect(Float64, A.P); Tk = collect(Float64, T)
    m = members(ctx, A.U, Tk, Pk)
    Tn = Float64[n == 1 ? Tk[i] : Tk[i+1] for n in 1:2, i in 1:nk-1]          # [nlobatto = 2, np - 1]
    μn = ones(Float64, 2, nk-1)
    e = get(ctx.accels, A, nothing)
    if e === nothing
        e = AccelEntry(takeslot!(ctx.accels, ctx.aorder, CS_MAX_ACCEL, A, ()), Float64[])
        ctx.accels[A] = e
    end
    lnσ = Matrix{Float64}(undef, nν, nk)
    GC.@preserve Pk Tk Tn μn m lnσ begin
        check(ccall((:cs_column_setup, LIB), Cint,
            (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Float64, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Float64, Cint, Cint, Cint),
            ctx.handle, nν, A.ν, C_NULL, nk, Pk, 1.0, 2, Tn, μn, Tk, length(m.slots), m.slots, m.shapes, m.cuts, m.conc,
            m.σgray, m.extra === nothing ? C_NULL : pointer(m.extra), C_NULL, C_NULL, 0.0, 1, 0, 0))
        isempty(m.tslots) || check(ccall((:cs_column_set_tables, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cint}, Ptr{Float64}),
            ctx.handle, length(m.tslots), m.tslots, m.conctab))
        isempty(m.cslots) || check(ccall((:cs_column_set_cia, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}),
            ctx.handle, length(m.cslots), m.cslots, m.cflags, m.P₁, m.P₂))
        check(ccall((:cs_accel_store, LIB), Cint, (Ptr{Cvoid}, Cint), ctx.handle, e.slot))      # max(ln Σ, ln floatmin), absorbers.jl:185-196
        check(ccall((:cs_accel_fetch, LIB), Cint, (Ptr{Cvoid}, Cint, Int64, Cint, Ptr{Float64}), ctx.handle, e.slot, nν, nk, lnσ))
    end
    for i in 1:nν, k in 1:nk
        A.ϕ[i].y[k] = lnσ[i,k]
    end
    A.T .= Tk                         # "remember the temperature", absorbers.jl:198
    e.T = copy(Tk)
    return nothing
end

#-------------------------------------------------------------------------------
# B3: numerical core dispatch [shared.jl:36,55-62; fluxes.jl:238-279]

struct HIPDiscretized <: AbstractNumericalCore
    nstream::Int64
    nlobatto::Int64
    ngpu::Int64         # devices 0 .. ngpu-1 of this node share the wavenumber grid (cs_fluxes_discretized_multi); 1: one device
    fluxpack::Symbol    # what radiate! brings back: :full = τ, M⁺, M⁻ and the band fluxes, as the reference fills its FluxPack
                        # [fluxes.jl:357-383]; :bands = F⁺, F⁻, Fnet only -- all heating! reads [radiative_convective.jl:109-144] --
                        # τ, M⁺, M⁻ stay in HBM (C_NULL across the ABI: no 146 MB copy per call at 1e5 × 60, no host ∫F!)
end
function HIPDiscretized(; nstream::Int=5, nlobatto::Int=2, ngpu::Int=1, fluxpack::Symbol=:full)
    fluxpack in (:full, :bands) || error("fluxpack must be :full or :bands, not :$fluxpack")
    HIPDiscretized(nstream, nlobatto, ngpu, fluxpack)
end

# One whole-column evaluation through the C ABI.  F⁺, F⁻ [np] always come back -- ∫F! [shared.jl:125-137] runs on the device, in
# the same trapezoid order; M⁺, M⁻ [np, nν] and τ [np-1, nν] are filled when given and skipped (C_NULL) when `nothing`.
function hipcolumn!(F⁺::Vector{Float64}, F⁻::Vector{Float64}, M⁺::Union{Nothing,AbstractMatrix}, M⁻::Union{Nothing,AbstractMatrix},
                    τ::Union{Nothing,AbstractMatrix}, core::HIPDiscretized, P::AbstractVector{<:Real}, g::Real, T, μ, 𝒻S, 𝒻a,
                    absorbers...; θₛ::Real=0.841)::Nothing
    𝒜, ν, nν = ClearSky.unifyabsorbers(absorbers)      # a UnifiedAbsorber (method above for HIP members) or an AcceleratedAbsorber
    𝒻T, 𝒻μ = formprofiles(P, T, μ)
    nstream, nlobatto = core.nstream, core.nlobatto
    @assert issorted(P) "pressure coordinates must be in ascending order (sorted)"
    # closures are evaluated here, exactly where fluxes.jl:253-267 / discretized.jl:11-30,46-58 evaluate them
    Tn, μn = lobattoevaluations(P, 𝒻T, 𝒻μ, nlobatto)
    Tn = Matrix{Float64}(Tn); μn = Matrix{Float64}(μn)
    Tlev = Float64[𝒻T(p) for p in P]
    hipcheckpressures(𝒜, P[end], P[1])                  # fluxes.jl:265
    checkstreams(nstream); checkazimuth(θₛ)
    np = length(P)
    @assert length(F⁺) == length(F⁻) == np
    Pv = collect(Float64, P)
    Pk, Tk = nodestates(Pv, Tn, nlobatto)
    ctxs = [context(d) for d in 0:core.ngpu-1]          # one context per device
    ctx = ctxs[1]
    accel = Cint(-1)
    if 𝒜 isa AcceleratedAbsorber                       # the slot stands for all absorbers (unifyabsorbers(::Tuple{AcceleratedAbsorber}), absorbers.jl:216)
        core.ngpu == 1 || error("an AcceleratedAbsorber lives on one context: use ngpu = 1")
        accel = accelslot!(ctx, 𝒜)
        m = Members(Cint[], Cint[], Float64[], zeros(0, length(Pk)), Cint[], zeros(0, length(Pk)), Cint[], Cint[], zeros(0, length(Pk)),
                    zeros(0, length(Pk)), 0.0, nothing)
    else
        m = members(ctx, 𝒜, Tk, Pk)
        if core.ngpu > 1
            (isempty(m.tslots) && isempty(m.cslots)) ||
                error("baked gases and CIA pairs live on one context: ngpu > 1 takes line-by-line, gray and function absorbers")
            direct = [x for x in 𝒜.gas if x isa DirectGas]
            for c in ctxs[2:end], (gi, x) in enumerate(direct)      # every context numbers the column's tables as the first one does
                forceslot!(c, x.sl, m.slots[gi])
            end
        end
    end
    Stoa = Float64[𝒻S(x) for x in ν]; alb = Float64[𝒻a(x) for x in ν]
    # caller's arrays are written in place when they are dense Float64 matrices (FluxPack's are); anything else goes through a copy
    dense(A) = A === nothing ? nothing : ((A isa Matrix{Float64}) ? A : Matrix{Float64}(undef, size(A)))
    Mu, Md, Ta = dense(M⁺), dense(M⁻), dense(τ)
    handles = Ptr{Cvoid}[c.handle for c in ctxs]
    ngas, ntab, ncia = length(m.slots), length(m.tslots), length(m.cslots)
    extra = m.extra
    GC.@preserve ν Pv Tn μn Tlev m extra Stoa alb Mu Md Ta F⁺ F⁻ handles begin
        pτ = Ta === nothing ? C_NULL : pointer(Ta)      # C_NULL: the output stays in HBM (include/clearsky_hip.h: "nullable")
        p⁺ = Mu === nothing ? C_NULL : pointer(Mu)
        p⁻ = Md === nothing ? C_NULL : pointer(Md)
        if core.ngpu > 1      # ν cut into ngpu cost-balanced ranges, one per device; band fluxes added on the host in device order
            check(ccall((:cs_fluxes_discretized_multi, LIB), Cint,
                (Ptr{Ptr{Cvoid}}, Cint, Int64, Ptr{Float64}, Cint, Ptr{Float64}, Float64, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Float64, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                handles, core.ngpu, nν, ν, np, Pv, Float64(g), nlobatto, Tn, μn, Tlev, ngas, m.slots, m.shapes, m.cuts, m.conc,
                m.σgray, extra === nothing ? C_NULL : pointer(extra), Stoa, alb, Float64(θₛ), nstream, pτ, p⁺, p⁻, F⁺, F⁻))
        elseif ntab == 0 && ncia == 0 && accel < 0
            check(ccall((:cs_fluxes_discretized, LIB), Cint,
                (Ptr{Cvoid}, Int64, Ptr{Float64}, Cint, Ptr{Float64}, Float64, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Float64, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                ctx.handle, nν, ν, np, Pv, Float64(g), nlobatto, Tn, μn, Tlev, ngas, m.slots, m.shapes, m.cuts, m.conc,
                m.σgray, extra === nothing ? C_NULL : pointer(extra), Stoa, alb, Float64(θₛ), nstream, pτ, p⁺, p⁻, F⁺, F⁻))
        else                  # baked gases, CIA pairs or an accelerated absorber among the members
            check(ccall((:cs_fluxes_discretized_members, LIB), Cint,
                (Ptr{Cvoid}, Int64, Ptr{Float64}, Cint, Ptr{Float64}, Float64, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Float64}, Ptr{Float64},
                 Cint, Ptr{Cint}, Ptr{Float64},
                 Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Float64}, Ptr{Float64},
                 Cint, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Float64, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                ctx.handle, nν, ν, np, Pv, Float64(g), nlobatto, Tn, μn, Tlev, ngas, m.slots, m.shapes, m.cuts, m.conc,
                ntab, m.tslots, m.conctab,
                ncia, m.cslots, m.cflags, m.P₁, m.P₂,
                accel, m.σgray, extra === nothing ? C_NULL : pointer(extra), Stoa, alb,
                Float64(θₛ), nstream, pτ, p⁺, p⁻, F⁺, F⁻))
        end
    end
    (Mu === nothing || Mu === M⁺) || copyto!(M⁺, Mu)
    (Md === nothing || Md === M⁻) || copyto!(M⁻, Md)
    (Ta === nothing || Ta === τ) || copyto!(τ, Ta)
    nothing
end

# the reference's in-place entry point [fluxes.jl:238-249]: M⁺, M⁻, τ are the outputs, so all three cross PCIe; the band fluxes the
# device formed on the way have no place in this signature (the callers that want them dispatch to the radiate! method below)
function monochromaticfluxes!(M⁺::AbstractMatrix, M⁻::AbstractMatrix, τ::AbstractMatrix, core::HIPDiscretized,
                              P::AbstractVector{<:Real}, g::Real, T, μ, 𝒻S, 𝒻a, absorbers...; θₛ::Real=0.841)::Nothing
    np = length(P)
    hipcolumn!(Vector{Float64}(undef, np), Vector{Float64}(undef, np), M⁺, M⁻, τ, core, P, g, T, μ, 𝒻S, 𝒻a, absorbers...; θₛ=θₛ)
end

# radiate!(F, core, P, g, T, μ, 𝒻S, 𝒻a, absorbers...) [fluxes.jl:357-383] takes `core` positionally, so this method is what
# `heating!` [radiative_convective.jl:113], `step!`, `jacobian!` and `radiate` reach with an RCM built on `core=HIPDiscretized(...)`.
# Same checks as the reference's body; F.F⁺, F.F⁻ come from the device's ∫F! instead of the serial strided host trapz over
# 2·np rows of nν [shared.jl:125-137, util.jl:26-33], and with `fluxpack=:bands` τ, M⁺, M⁻ are neither copied nor touched
# (they keep whatever the FluxPack held: zeros from its constructor).
function radiate!(F::ClearSky.FluxPack, core::HIPDiscretized, P::AbstractVector{<:Real}, g::Real, T, μ, 𝒻S, 𝒻a,
                           absorbers...; θₛ::Real=0.841)::Nothing
    𝒜, ν, nν = ClearSky.unifyabsorbers(absorbers)
    np = length(P)
    @assert size(F) == (np, nν) "size of FluxPack does not match number of pressure or wavenumber coordinates"
    F⁺ = F.F⁺ isa Vector{Float64} ? F.F⁺ : Vector{Float64}(undef, np)
    F⁻ = F.F⁻ isa Vector{Float64} ? F.F⁻ : Vector{Float64}(undef, np)
    if core.fluxpack == :bands
        hipcolumn!(F⁺, F⁻, nothing, nothing, nothing, core, P, g, T, μ, 𝒻S, 𝒻a, 𝒜; θₛ=θₛ)
    else
        hipcolumn!(F⁺, F⁻, F.M⁺, F.M⁻, F.τ, core, P, g, T, μ, 𝒻S, 𝒻a, 𝒜; θₛ=θₛ)
    end
    F⁺ === F.F⁺ || copyto!(F.F⁺, F⁺)
    F⁻ === F.F⁻ || copyto!(F.F⁻, F⁻)
    @. F.Fnet = F.F⁺ - F.F⁻
    return nothing
end

# fluxes / netfluxes [fluxes.jl:311-352] take `core` as a KEYWORD, which Julia does not dispatch on, so the reference's bodies
# always allocate M⁺, M⁻, τ [np, nν] and integrate on the host.  These two return the same F⁺, F⁻ [W/m²] from the device's ∫F!
# without any of the three leaving HBM.
function hipfluxes(P::AbstractVector{<:Real}, g::Real, T, μ, 𝒻S, 𝒻a, absorbers...; core::HIPDiscretized=HIPDiscretized(),
                   θₛ::Real=0.841)
    np = length(P)
    F⁺ = Vector{Float64}(undef, np); F⁻ = Vector{Float64}(undef, np)
    hipcolumn!(F⁺, F⁻, nothing, nothing, nothing, core, P, g, T, μ, 𝒻S, 𝒻a, absorbers...; θₛ=θₛ)
    return F⁺, F⁻
end
function hipnetfluxes(P::AbstractVector{<:Real}, g::Real, T, μ, 𝒻S, 𝒻a, absorbers...; kwargs...)
    F⁺, F⁻ = hipfluxes(P, g, T, μ, 𝒻S, 𝒻a, absorbers...; kwargs...)
    return F⁺ .- F⁻
end

# jacobian! [radiative_convective.jl:154-171] as ONE device batch: the band fluxes of B temperature profiles on the column of the
# last monochromaticfluxes! call of this thread's context (same grid, levels and members).  Ts[b] is whatever `T` was in that call
# (vector at the levels or a function of P).  Returns F⁺, F⁻ as [np, B].
function batchfluxes(core::HIPDiscretized, P::AbstractVector{<:Real}, Ts::AbstractVector, μ, absorbers...)
    𝒜, ν, nν = ClearSky.unifyabsorbers(absorbers)
    ctx = context(0)
    np = length(P); nl = np - 1; B = length(Ts); nlob = core.nlobatto
    Pv = collect(Float64, P)
    K = nl*(nlob - 1) + 1
    Tn_all = Matrix{Float64}(undef, nlob*nl, B); μn_all = similar(Tn_all); Tlev_all = Matrix{Float64}(undef, np, B)
    accel = 𝒜 isa AcceleratedAbsorber
    ng = accel ? 0 : count(x -> x isa DirectGas, 𝒜.gas)
    nt = accel ? 0 : count(x -> x isa Union{HIPGas,Gas}, 𝒜.gas)
    nc = accel ? 0 : length(𝒜.cia)
    conc_all = zeros(Float64, max(ng, 1)*K, B); ctab_all = zeros(Float64, max(nt, 1)*K, B)
    P₁_all = zeros(Float64, max(nc, 1)*K, B); P₂_all = zeros(Float64, max(nc, 1)*K, B)
    for b in 1:B
        𝒻T, 𝒻μ = formprofiles(Pv, Ts[b], μ)
        Tn, μn = lobattoevaluations(Pv, 𝒻T, 𝒻μ, nlob)
        Tn_all[:,b] = vec(Tn); μn_all[:,b] = vec(μn)
        Tlev_all[:,b] = Float64[𝒻T(p) for p in Pv]
        if !accel
            Pk, Tk = nodestates(Pv, Matrix{Float64}(Tn), nlob)
            m = members(ctx, 𝒜, Tk, Pk)
            m.extra === nothing || error("function absorbers cannot be batched")
            ng > 0 && (conc_all[1:ng*K,b] = vec(m.conc))
            nt > 0 && (ctab_all[1:nt*K,b] = vec(m.conctab))
            nc > 0 && (P₁_all[1:nc*K,b] = vec(m.P₁); P₂_all[1:nc*K,b] = vec(m.P₂))
        end
    end
    F⁺ = Matrix{Float64}(undef, np, B); F⁻ = similar(F⁺)
    GC.@preserve Tn_all μn_all Tlev_all conc_all ctab_all P₁_all P₂_all F⁺ F⁻ check(ccall((:cs_column_batch, LIB), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}),
        ctx.handle, B, Tn_all, μn_all, Tlev_all, conc_all, nt > 0 ? pointer(ctab_all) : C_NULL, nc > 0 ? pointer(P₁_all) : C_NULL,
        nc > 0 ? pointer(P₂_all) : C_NULL, F⁺, F⁻))
    return F⁺, F⁻
end

export HIPDiscretized, DirectGas, HIPGas, HIPCIA, hipvoigt!, hiplorentz!, hipdoppler!, hipPHCO2!, hipbake!, hipshapepoints, batchfluxes,
       hipfluxes, hipnetfluxes

end # module
