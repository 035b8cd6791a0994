/*
 * clearsky_hip.h -- C ABI of the MI355X (gfx950) line-by-line radiative-transfer core.
 *
 * Drop-in boundary for ClearSky.jl's Discretized hot path.  The reference has no FFI; each
 * entry point below names the Julia interface it replaces (paths relative to the reference root) and is what
 * a `ccall` from the Julia glue in julia/ClearSkyHIP.jl binds.  Plain C linkage, plain pointers and sizes.
 *
 * Conventions
 *   - return 0 on success, a negative CS_E* code on failure; cs_last_error() returns a thread-local message.
 *   - "host" pointers are borrowed for the duration of the call (Julia: GC.@preserve); device memory is owned
 *     by the opaque context.  Host entry points are synchronous.
 *   - a context is not re-entrant: use one per host thread (the reference calls shape! from @threads,
 *     gases.jl:115).
 *   - all reals are IEEE fp64, as in the reference.
 *   - matrices the reference stores column-major [level, nu] (core/shared.jl:93-101) are exchanged in exactly
 *     that layout unless a call says "nu-fastest".
 */
#ifndef CLEARSKY_HIP_H
#define CLEARSKY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cs_ctx cs_ctx;

enum {
    CS_OK = 0,
    CS_EINVAL = -1,    /* bad argument (the reference's @assert / error paths) */
    CS_ETEMP = -2,     /* temperature outside [TMIN,TMAX] = [25,1000] K, line_shapes.jl:29 */
    CS_ENOCHEB = -3,   /* isotopologue without a Qref/Q Chebyshev fit, line_shapes.jl:115-120 */
    CS_EORDER = -4,    /* nu not strictly ascending (line_shapes.jl:59) or P not ascending (fluxes.jl:257) */
    CS_EHIP = -5,      /* HIP runtime error */
    CS_ESTATE = -6     /* call sequence error (e.g. run before setup) */
};

/* line shapes: voigt! line_shapes.jl:412, lorentz! :313, doppler! :200, PHCO2! :527 */
enum { CS_SHAPE_VOIGT = 0, CS_SHAPE_LORENTZ = 1, CS_SHAPE_DOPPLER = 2, CS_SHAPE_PHCO2 = 3 };

#define CS_MAX_GAS 16
#define CS_MAX_TABLE 16
#define CS_MAX_CIA 8
#define CS_MAX_ACCEL 4
#define CS_CHEB_LD 16 /* leading dimension of the Chebyshev coefficient table */
#define CS_MAX_STREAM 16
#define CS_MAX_LOBATTO 16

int cs_version(void);
/* 16 hex digits naming the sources this binary was compiled from (sha256 over csrc/ and this header, set by the build recipe;
 * "unknown" for a build that did not pass it): measurements quote it, so a profile can be matched to the library it was taken on */
const char *cs_build_id(void);
const char *cs_last_error(void);

/* Create a context on HIP device `device` (one per host thread). */
int cs_create(int device, cs_ctx **out);
void cs_destroy(cs_ctx *ctx);

/*
 * Upload one gas's line table into `slot` (0 <= slot < CS_MAX_GAS).
 * Replaces: SpectralLines (hitran/par.jl:224-251; lines sorted by nu, :267-283) plus the MOLPARAM rows of its
 * molecule (hitran/molparam.jl; struct hitran/par.jl:18-45).
 *   nu,S,gamma_a,gamma_s,Epp,na : [L] as in SpectralLines;  mu_iso : [L] molar mass of each line's isotopologue
 *   iso : [L] 1-based local isotopologue number (par.jl:263);  ncheb : [niso] (0 = no fit, hascheb=false)
 *   cheb : [niso][CS_CHEB_LD] Chebyshev coefficients of Q/Qref (line_shapes.jl:27-48)
 */
int cs_gas_upload(cs_ctx *ctx, int slot, int64_t L, const double *nu, const double *S, const double *gamma_a,
                  const double *gamma_s, const double *Epp, const double *na, const double *mu_iso,
                  const int16_t *iso, int niso, const int32_t *ncheb, const double *cheb);
int cs_gas_clear(cs_ctx *ctx, int slot);

/*
 * Arithmetic of the Voigt far wings (BASELINE configs[4], "fp32 mixed-precision variant with tolerance sweep"):
 *   mode 0 (default): fp64 everywhere;
 *   mode 1: (nu, line) pairs with x^2 >= far_s (far_s >= 1e6) that the VECTOR unit evaluates go to fp32 (nu - nul is still formed in
 *           fp64, partial sums of 4 terms, fp64 accumulation); everything nearer, and every piece the matrix cores sum
 *           (cs_set_matrix_cores: v_mfma_f64 beats the fp32 vector bodies), stays fp64.  far_s is the knob of the tolerance sweep
 *           (larger: fewer pairs in fp32).  Cross-sections then agree with mode 0 to ~1e-7.
 * Applies to every later cs_shape_batch / cs_bake / cs_column_run of the context.
 */
int cs_set_precision(cs_ctx *ctx, int mode, double far_s);

/*
 * Far wings by spectral interpolation (on by default).  The reference evaluates every (nu, line) pair of surf!
 * (line_shapes.jl:56-96); with `on`, lines further than max(100 Doppler widths, 0.3 x the half-width of a 256-point
 * interval) from the interval -- and inside the cut-off of all its points -- are summed at 64 Chebyshev nodes of the interval
 * and interpolated, which reproduces the pointwise sum to ~1e-15.  on = 0: every pair is evaluated.
 * Applies to every later cs_shape_batch / cs_bake / cs_column_setup of the context.
 */
int cs_set_interp(cs_ctx *ctx, int on);
/* Tuning of the interpolation plan (defaults: -1, 128, 2048): `first_level` forces the first (largest) interval level every
 * gas uses -- 0 = all levels of the grid, n = skip the n largest, >= the number of levels = none (every pair evaluated
 * directly), -1 = chosen per gas from its line density; [size_min, size_max] restricts the interval sizes considered.
 * Results do not depend on the plan beyond rounding (tests/test_gpu_interp.py); it only moves work between kernels. */
int cs_set_interp_plan(cs_ctx *ctx, int first_level, int size_min, int size_max);
/* Far lines on the matrix cores: where the 4-term series in 1/dnu^2 holds for every state of a group of 16 (|dnu| >= 133.6
 * sqrt(gamma^2 + 4.33 alpha^2)), the node sums of the interpolated far wings and the window ends of the
 * per-point sum -- cut-off edges included, as a mask (K2e) -- are matrix products on v_mfma_f64_16x16x4.  on = 1 (default):
 * where the grid has enough (interval | tile, state group) blocks to fill the chip; 2: always; 0: everything on the vector
 * unit; | 4 keeps the tile-wide near-zone pass where the default hands the core of a window to 16-point sub-tiles (k_voigt_sub).
 * Same results to rounding (tests/test_gpu_interp.py). */
int cs_set_matrix_cores(cs_ctx *ctx, int on);

/* One launch set per column (on by default): the Voigt (Lorentz) gases of a column that share a cut-off are merged into one sorted
 * line table when the column is set up -- sigma_total = sum_g C_g sigma_g (absorbers.jl:84-95), and a per-(state, line) record
 * carries its gas's concentration and partial pressure, so the kernels run once per column instead of once per gas, on windows as
 * dense as all the column's lines together.  on = 0: one launch set per gas.  Same results to rounding (the order of the sum over
 * lines changes).  Applies to every later cs_column_setup / cs_fluxes_discretized of the context. */
int cs_set_merge(cs_ctx *ctx, int on);

/* Tuning switches for A/B measurements (results do not depend on them beyond rounding; defaults from profiles/r03_notes.md):
 *   key 0: the interpolated far wings are carried to the grid inside k_voigt_edge_mx where the column has one launch group (1), or
 *          by their own launch (0, default: one launch more, the same time at C3, less with five interval levels);
 *   key 1: the matrix-core kernels also on grids too short to fill the chip with one (interval | tile, state group) per wave, through
 *          their four-waves-per-item variants (1, default; 0: such grids stay on the vector unit);
 *   key 2: the node sums of a launch group run on a side stream beside its per-point kernels -- 2 (default) always, 1 on short
 *          grids only (fewer than 16384 (tile, state) waves: a nu-shard), 0 never;
 *   key 3: distance of an interval's interpolated set from the interval, per cent of its half-width (0 = the default 30; 15..100);
 *   key 4: cs_column_run replays the step as one hipGraph (captured on the second run after anything changed launch geometry or
 *          kernel arguments; cs_column_update_state does not) instead of enqueuing its kernels one by one -- 0 (default) off, 1 on;
 *   key 5: on short grids (up to 400 tiles) the flux sweeps run one wave per (sweep, stream) of a tile (k_rt_streams) instead of one
 *          per sweep -- 1 (default), 0 off;
 *   key 6: how many interval sizes, largest first, have their matrix-core node sums shared by the four waves of a block (0 = the
 *          default, see profiles/r03_notes.md);
 *   key 7: the sub-tile cores (k_voigt_sub) and the near-line kernels run on a second side stream -- the former beside k_voigt_far,
 *          the latter beside k_voigt_edge_mx and what follows it -- adding into a plane of their own that k_rt reads together with
 *          sigma: 1 (default) on grids of 8192 .. 300000 (tile, state) waves, where it was measured to pay; 2 always; 0: on the
 *          main stream, into sigma;
 *   key 8: a far line joins a state group's matrix-core node piece when at least this many of the group's 16 states are beyond their
 *          own series radius (the others' coefficients are masked, the vector unit sums them) -- default 7; 16 = the group's widest
 *          line decides (round 2), 1 = its narrowest;
 *   key 9: PHCO2: the pairs within 3 cm^-1 of a line (chi = 1 there: the plain Voigt profile, every near-line pair among them) go
 *          through the Voigt kernels with a 3 cm^-1 cut-off after k_phco2 (0, default), or through k_phco2's own core loop (1);
 *   key 10: PHCO2 interpolation levels: bit 0 = 64 nodes on every interval (default: 16 or 32 where a region's lines are many
 *          half-widths from the intervals of a size), bit 1 = the 64-point tiles themselves as the smallest interval size;
 *   key 11: the far pieces of an interval's matrix-core node sums (the lines beyond its parent's set: 3.8 .. 12 half-widths away on
 *          the bench grid) are summed on 32 or 16 nodes and carried to the interval's 64 (0, default), or on all 64 (1);
 *   key 12: the node sums of a level are added into the next smaller level's (a 64 x 64 matrix per interval) and only the smallest
 *          interval size is carried to the grid (k_cheb_cascade) -- 0 (default) with four or more levels in use, 1 always, 2 never;
 *   key 13: the vector-unit node kernel with four waves per (interval, state), a quarter of every window each -- 0 (default) on
 *          grids of fewer than 16384 (interval, state) waves (a nu-shard), 1 always, 2 never;
 *   key 14: k_voigt_edge_mx cuts a cut-off edge where the next 16-column sub-tile of the tile comes into the lines' reach and
 *          multiplies only the sub-tiles a part can reach (0, default), or all four for every line (1);
 *   key 15: the flux kernel finishes the cross-sections on chip (k_flux: the interpolated wings as a matrix product, the CIA pairs and
 *          the near-line plane are added per 64-point tile in LDS instead of by read-modify-write passes over the [K][nnu] plane, and
 *          the last block adds the band-flux partials; fluxes.jl:270-277 does depth and flux of a wavenumber in one loop body) -- 0
 *          (default) where it pays: grids of up to 1024 tiles (a nu-shard, a small column) and of 4096 tiles or more, 1 never, 2 always;
 *          | 4: the block partials are always added by k_freduce's own launch, | 8: the long-grid form with four waves per SIMD (A/B);
 *          | 16: the matrix-core piece tables with one thread per (interval | tile, state group) (k_mxzones) instead of sixteen lanes (A/B);
 *          | 64: short grids with one wave per (sweep, stream) walking all layers (k_flux_streams) instead of the scan over layer chunks
 *          (k_flux_scan) (A/B);
 *          | 256: on short grids the interval levels are NOT folded into the smallest one on the node-sum side stream (A/B);
 *          | 512: the level cascade (key 12) as ONE launch for all levels (k_cheb_cascade_tree) instead of one per level (A/B: no faster);
 *          | 1024: the scan form also on grids of 1024 .. 4096 tiles (A/B: slower at 1563 tiles);
 *          | 2048: k_flux_scan forms the transmissivities of a layer chunk again in its second sweep and second pass instead of
 *          keeping them in registers (A/B; same results);
 *          | 32 (on the first context of a cs_fluxes_discretized_multi call): the partition is re-cut from measured times also when
 *          contexts share a device (tests);
 *   key 16: issue priority (s_setprio 3) for the waves of the near-line stream's kernels (k_voigt_sub, k_voigt_near), whose chains of
 *          dependent gathers otherwise lose their issue slots to the streaming kernels on the other two streams -- 0 (default) on grids
 *          of 512 tiles or more, 1 never, 2 always (bench column 2.00 -> 1.95 ms; an eighth of it 0.378 -> 0.387, hence the threshold).
 *          | 4: the two tiers of the near-line pairs in two launches (k_voigt_near<0>, <1>) also where a wave takes one tile, instead of
 *          one (k_voigt_near_both) (A/B; same results).
 * Applies to every later cs_column_setup / cs_column_run of the context. */
int cs_set_tuning(cs_ctx *ctx, int key, int value);

/*
 * B1: batched in-place line shape.  For every state k:  sigma[k*ld_state + i] = shape(nu[i]; T[k], P[k], Pp[k]).
 * Replaces: shape!(sigma, nu, sl, T, P, Pp, dnu_cut) -- voigt!/lorentz!/doppler!/PHCO2!, line_shapes.jl:412-424,
 * :313-324, :200-211, :527-540 -- as invoked by bake, gases.jl:126 (K = nT*nP states in one launch instead of
 * nT*nP calls).  sigma is overwritten (line_shapes.jl:85).  Semantics of includedlines(::Vector) (:18-22) and
 * surf! (:53-87) are reproduced, including the strict end-point pre-filter and the inclusive cut-off.
 * All pointers are host pointers.
 */
int cs_shape_batch(cs_ctx *ctx, int slot, int shape, double dnu_cut, int64_t nnu, const double *nu, int K,
                   const double *T, const double *P, const double *Pp, double *sigma, int64_t ld_state);

/*
 * Bake a gas into a resident opacity table ("Mode T": the reference's default Gas objects).
 * Replaces: bake(sl, fC, shape!, dnu_cut, nu, Omega) gases.jl:97-145 (all nT*nP line sums in one launch, the
 * zero-row scrub :132-142) and OpacityTable(T, P, sigma) gases.jl:75-82 (ln sigma, or ln(floatmin) for empty rows).
 *   T[nT], P[nP] : Omega.T, Omega.P (Chebyshev extrema in T and ln P, gases.jl:57-58)
 *   conc         : [nT, nP] column-major, conc[i + nT*j] = fC(T_i, P_j) (partial pressure = conc*P, gases.jl:126)
 *   lnsigma_out  : NULL or host [nnu, nT, nP] column-major (nu fastest) -- the tables themselves, for inspection
 * The table stays in HBM as slot `table_slot` and is evaluated by cs_column_set_tables / cs_column_run through the
 * 2-D Chebyshev interpolant (BichebyshevInterpolator, gases.jl:80,85) in barycentric form.
 */
/* The same kernels with the semantics of the scalar-wavenumber methods  voigt(nu, sl, T, P, Pp, dnu_cut)  etc. (line_shapes.jl:
 * 399-405, 290-296, 177-183, 514-520) mapped over the n wavenumbers: every line with |nu - nul| <= dnu_cut counts
 * (includedlines(::Real), line_shapes.jl:12-16; no strict end-point pre-filter).  This is what a function absorber
 * (nu,T,P) -> C*voigt(nu, sl, T, P, C*P) and the scalar Sigma(U, i, T, P) (absorbers.jl:84-95) evaluate.  nu ascending. */
int cs_shape_points(cs_ctx *ctx, int slot, int shape, double dnu_cut, int64_t nnu, const double *nu, int K,
                    const double *T, const double *P, const double *Pp, double *sigma, int64_t ld_state);

int cs_bake(cs_ctx *ctx, int gas_slot, int table_slot, int shape, double dnu_cut, int64_t nnu, const double *nu, int nT,
            const double *T, int nP, const double *P, const double *conc, double *lnsigma_out);
int cs_table_clear(cs_ctx *ctx, int table_slot);
/* A table the HOST baked -- the reference's own Gas object (gases.jl:205-249), whatever shape! filled it: lnsigma[nnu, nT, nP]
 * column-major (nu fastest) = the knot values ln sigma of its OpacityTables (gases.jl:75-82; finite: ln(floatmin) for empty rows) on
 * T[nT] x P[nP] = Omega.T, Omega.P.  The slot then behaves exactly as one filled by cs_bake. */
int cs_table_upload(cs_ctx *ctx, int table_slot, int64_t nnu, const double *nu, int nT, const double *T, int nP, const double *P,
                    const double *lnsigma);
/* sigma(nu[i0..i0+n), T, P) of a baked table WITHOUT the concentration factor: rawsigma(g, T, P) gases.jl:256-263 */
int cs_table_eval(cs_ctx *ctx, int table_slot, double T, double P, int64_t i0, int64_t n, double *sigma_out);

/*
 * Collision-induced absorption tables.  Replaces: CIATables (collision_induced_absorption.jl:145-235) and its functor
 * (:251-276).  A CIA object is a set of bands; band b holds ln k [cm^5/molecule^2] on nu_b[nb] x T_b[nt] (nu fastest),
 * evaluated bilinearly inside the grid (BilinearInterpolator of ln k, :207); nt == 1 marks a single-temperature range
 * (LinearInterpolator in nu, :188), used only when `singles` is set.  k <= 0 must already be replaced as the reference
 * does (floatmin for grids :205, 0 -> ln 0 = -inf for singles :187).
 */
int cs_cia_begin(cs_ctx *ctx, int cia_slot, int nband);
int cs_cia_band(cs_ctx *ctx, int cia_slot, int band, int nb, const double *nu_b, int nt, const double *T_b, const double *lnk);
int cs_cia_clear(cs_ctx *ctx, int cia_slot);

/*
 * B3: whole-column monochromatic fluxes + band integrals with the Discretized core, line-by-line at every
 * Lobatto node ("Mode D").
 * Replaces: monochromaticfluxes!(M+, M-, tau, core::Discretized, P, g, T, mu, fS, fa, absorbers...; theta_s)
 * fluxes.jl:238-279, followed by intF! core/shared.jl:125-137 (as radiate! fluxes.jl:357-383 does).
 * Closures cannot cross the ABI; the caller pre-evaluates them where the reference does:
 *   T_nodes, mu_nodes : [nlobatto, np-1] column-major = lobattoevaluations(P, fT, fmu, nlobatto) discretized.jl:11-30
 *   T_levels          : [np] = fT(P[i])  (planckevaluations, discretized.jl:46-58)
 *   conc              : [ngas, K] column-major, conc[g + ngas*k] = fC_g(T_k, P_k) at node k (gases.jl:270,278),
 *                       K = (np-1)*(nlobatto-1) + 1, node k = i*(nlobatto-1) + n for Lobatto point n of layer i
 *   sigma_gray        : constant cross-section added everywhere (GrayGas, gases.jl:342-360); 0 for none
 *   sigma_extra       : NULL or [nnu, K] nu-fastest, host-evaluated sigma(nu,T,P) functions / CIA (absorbers.jl:84-92)
 *   S_toa, albedo     : NULL (= 0) or [nnu] = fS(nu[j]), fa(nu[j])  (discretized.jl:299,309)
 * Outputs (any of tau/Mup/Mdn may be NULL):
 *   tau : [np-1, nnu];  Mup, Mdn : [np, nnu]  column-major, level fastest;  Fup, Fdn : [np]
 * P must be ascending (index 0 = top of atmosphere), nu strictly ascending.
 */
int cs_fluxes_discretized(cs_ctx *ctx, int64_t nnu, const double *nu, int np, const double *P, double g,
                          int nlobatto, const double *T_nodes, const double *mu_nodes, const double *T_levels,
                          int ngas, const int *gas_slots, const int *shapes, const double *dnu_cuts,
                          const double *conc, double sigma_gray, const double *sigma_extra, const double *S_toa,
                          const double *albedo, double theta_s, int nstream, double *tau, double *Mup,
                          double *Mdn, double *Fup, double *Fdn);

/*
 * B3 on several GPUs of one node, from one process: radiate! (fluxes.jl:357-383) with `ngpu` devices behind it.  Every wavenumber is
 * independent through the whole path; the only coupling is the trapezoid over nu (intF!, shared.jl:125-137).  The grid is cut into
 * nctx contiguous ranges of equal estimated device time (cs_balanced_ranges); context i -- created by the caller on the device of
 * its choice, holding the SAME gas tables in the same slots -- evaluates range i with slices of the global trapezoid weights (no
 * halo), all contexts side by side (one host thread each); tau, M+, M- land in the caller's arrays range by range, and the band
 * fluxes are added on the host in context order -- 2*np doubles per GPU, bitwise repeatable (SURVEY.md 8e's deterministic
 * alternative to an all-reduce; the RCCL form for one process per GPU is bench.py's).  Arguments as cs_fluxes_discretized.
 * Repeated calls on an unchanged grid re-use each context's resident shard.  nctx = 1 is cs_fluxes_discretized.
 */
int cs_fluxes_discretized_multi(cs_ctx *const *ctxs, int nctx, int64_t nnu, const double *nu, int np, const double *P, double g, int nlobatto,
                                const double *T_nodes, const double *mu_nodes, const double *T_levels, int ngas, const int *gas_slots,
                                const int *shapes, const double *dnu_cuts, const double *conc, double sigma_gray, const double *sigma_extra,
                                const double *S_toa, const double *albedo, double theta_s, int nstream, double *tau, double *Mup,
                                double *Mdn, double *Fup, double *Fdn);
/*
 * B3 for every member an AbstractAbsorber can hold (B2): monochromaticfluxes!(M+, M-, tau, core::Discretized, ...,
 * absorbers...) fluxes.jl:238-279 + intF! shared.jl:125-137 with the sigma-chain of absorbers.jl:84-95 on the device -- line-by-line
 * gases (gas_slots, as cs_fluxes_discretized), baked Gas objects (gases.jl:205-281: table_slots of cs_bake / cs_table_upload with
 * conc_tab[ntab, K] = fC_t(T_k, P_k)), CIA pairs (collision_induced_absorption.jl:431-465: cia_slots of cs_cia_begin / cs_cia_band,
 * cia_flags bit 0 = extrapolate, bit 1 = singles, cia_P1 / cia_P2 [ncia, K] = P_k * concentration(g1 / g2, T_k, P_k), :378-382), the
 * gray term and host-evaluated functions (sigma_gray, sigma_extra) -- or, instead of all of these, an AcceleratedAbsorber
 * (absorbers.jl:114-203, what heating! hands radiate!, radiative_convective.jl:112-113): accel_slot >= 0 with ngas = ntab = ncia = 0;
 * accel_slot = -1 otherwise.  Everything else as cs_fluxes_discretized, which is this call without tables, CIA pairs and accelerated
 * absorber.  Repeated calls on an unchanged grid, level set and member line-up keep the column resident and refresh only the node
 * states, spectra and per-node member inputs (the RCM loop).
 */
int cs_fluxes_discretized_members(cs_ctx *ctx, int64_t nnu, const double *nu, int np, const double *P, double g, int nlobatto,
                                  const double *T_nodes, const double *mu_nodes, const double *T_levels, int ngas,
                                  const int *gas_slots, const int *shapes, const double *dnu_cuts, const double *conc,
                                  int ntab, const int *table_slots, const double *conc_tab,
                                  int ncia, const int *cia_slots, const int *cia_flags, const double *cia_P1, const double *cia_P2,
                                  int accel_slot, double sigma_gray, const double *sigma_extra, const double *S_toa, const double *albedo,
                                  double theta_s, int nstream, double *tau, double *Mup, double *Mdn, double *Fup, double *Fdn);

/* The partition of the above (host only, no GPU needed): nparts contiguous ranges [ranges[2r], ranges[2r+1]) of the grid with equal
 * estimated device time -- a fixed cost per wavenumber plus, per gas, its local line density times a factor growing with nu
 * (near-line pairs scale with the Doppler width); edges on multiples of 64 points where the grid is long enough; every range
 * non-empty.  line_nu[g][0 .. nlines[g]) = the sorted line positions of gas g. */
int cs_balanced_ranges(int64_t nnu, const double *nu, int ngas, const int64_t *nlines, const double *const *line_nu, int nparts, int64_t *ranges);
/* The same partition re-cut from MEASURED times (host only): prev_ranges = an earlier partition of this grid into nparts ranges,
 * prev_time[r] = what range r took (any unit), fixed_time = the share of a step that does not move with the edges (launch chain; 0 if
 * unknown: the correction then falls short and a second pass finishes it).  The cost model's density is rescaled range by range to
 * reproduce the measured times, then cut into equal parts again -- the column's own behaviour instead of constants fitted to another
 * column (BASELINE configs[2]).  cs_fluxes_discretized_multi does this once, inside its first call on a grid, when its contexts sit on
 * different devices; bench.py does it across ranks before its timed region. */
int cs_rebalance_ranges(int64_t nnu, const double *nu, int ngas, const int64_t *nlines, const double *const *line_nu, int nparts,
                        const int64_t *prev_ranges, const double *prev_time, double fixed_time, int64_t *ranges);

/*
 * Device-resident form of B3 for callers that keep the column in HBM (benchmarks, torch/RCCL plumbing, RCM loops):
 *   cs_column_setup  uploads the inputs of cs_fluxes_discretized once and allocates all workspaces;
 *                    `wts` is NULL (trapezoid weights of `nu`, util.jl:26-33) or [nnu] weights of a nu-shard
 *                    taken from the global grid (multi-GPU: SURVEY.md 8e);
 *   cs_column_run    enqueues the kernels on `stream` (a hipStream_t, NULL = the context's stream); asynchronous;
 *                    results stay in HBM;
 *   cs_column_flux_ptr  device address of [2*np] doubles (Fup then Fdn) for an in-place RCCL all-reduce;
 *   cs_column_fetch  synchronises and copies results to host (NULL = skip), layouts as in cs_fluxes_discretized; nnu and np
 *                    are the sizes the caller's buffers were allocated for -- CS_ESTATE if the resident column differs
 *                    (a context holds ONE resident column; a later cs_column_setup replaces it);
 *   cs_column_sigma_fetch  copies the total cross-section at the nodes, [nnu, K] nu-fastest (test hook), same size check;
 *   cs_column_counts line-shape evaluations of one run (sum over nu, node, gas of lines inside the cut-off).
 */
int cs_column_setup(cs_ctx *ctx, int64_t nnu, const double *nu, const double *wts, int np, const double *P,
                    double g, int nlobatto, const double *T_nodes, const double *mu_nodes, const double *T_levels,
                    int ngas, const int *gas_slots, const int *shapes, const double *dnu_cuts, const double *conc,
                    double sigma_gray, const double *sigma_extra, const double *S_toa, const double *albedo,
                    double theta_s, int nstream, int want_tau, int want_M);
/* add baked gases to the resident column: the Gas functor fC(T,P)*exp(Phi_i(T, ln P)) (gases.jl:278) at every node.
 * conc_tab: [ntab, K] column-major = fC_t(T_k, P_k).  Node states outside a table's (T,P) domain are an error
 * (checkpressures absorbers.jl:237-246; the interpolator's own bounds check for T). */
int cs_column_set_tables(cs_ctx *ctx, int ntab, const int *table_slots, const double *conc_tab);
/* add CIA pairs to the resident column: sigma += cia(nu, tables, T, P, P1, P2) (collision_induced_absorption.jl:295-303,
 * :318-323, the CIA functor :465).  P1, P2: [ncia, K] column-major partial pressures of the two gases at the nodes
 * (= P*concentration(g, T, P), :378-382); flags[c] bit 0 = extrapolate, bit 1 = singles (:163). */
int cs_column_set_cia(cs_ctx *ctx, int ncia, const int *cia_slots, const int *flags, const double *P1, const double *P2);
int cs_column_run(cs_ctx *ctx, void *stream);
/* only the cross-section stage of cs_column_run: sigma[K][nnu] = Sigma(absorbers, i, T_k, P_k) for every wavenumber and node
 * (absorbers.jl:95), left in HBM for cs_column_sigma_fetch / cs_accel_store.  Asynchronous. */
int cs_column_sigma_run(cs_ctx *ctx, void *stream);
int cs_column_sync(cs_ctx *ctx);
/* run `reps` evaluations with HIP events between the kernel classes on `stream`; ms[10] = average milliseconds per
 * evaluation spent in {k_gas_setup (+ k_mxzones), k_cheb_nodes, k_cheb_apply (+ k_table_eval, k_cia), k_voigt_far (or
 * k_linesum), k_voigt_near, k_rt, k_freduce, k_cheb_nodes_mx, k_voigt_edge_mx, k_voigt_sub}, summed over gases */
int cs_column_profile(cs_ctx *ctx, void *stream, int reps, double *ms);
int cs_column_flux_ptr(cs_ctx *ctx, double **dF);
/* asynchronously copy the [2*np] band fluxes (Fup then Fdn) into caller-owned DEVICE memory on `stream` */
int cs_column_flux_to(cs_ctx *ctx, double *dst_device, void *stream);
/* from now on the resident column WRITES its [2*np] band fluxes into caller-owned device memory (a buffer a collective reduces in place,
 * a tensor of the host framework) instead of its own -- no copy per step; NULL: back to its own.  Holds until the next cs_column_setup.
 * cs_column_flux_ptr / cs_column_fetch / cs_column_flux_to read from wherever the fluxes are. */
int cs_column_set_flux_dst(cs_ctx *ctx, double *dst_device);
int cs_column_fetch(cs_ctx *ctx, int64_t nnu, int np, double *tau, double *Mup, double *Mdn, double *Fup, double *Fdn);
int cs_column_sigma_fetch(cs_ctx *ctx, int64_t nnu, int K, double *sigma);
int cs_column_counts(cs_ctx *ctx, int64_t *pair_evals, int64_t *lines_in_range);
/* measurement hook, out[8]: out[0] = launch groups of the resident column (merged gases count once), out[1] = kernel launches of
 * the last cs_column_run, out[2] = lines of all groups, out[3] = cs_set_merge at setup time, out[4] = gases in the largest group,
 * out[5] = the flux kernel of the last run: 0 = k_rt / k_rt_streams on finished cross-sections, 1 = k_flux_streams, 2 = k_flux_chunk,
 * 3 = k_flux_scan */
int cs_column_info(cs_ctx *ctx, int64_t *out);
/* line-shape evaluations the last cs_column_run actually issued for its Voigt gases (measurement hook): out[0] = per-point
 * evaluations of k_voigt_far/k_voigt_near (64 lanes x lines per wave), out[1] = node evaluations of k_cheb_nodes,
 * out[2] = interpolation levels in use, out[3] = intervals over all levels; out[4..9] = the per-point evaluations by loop body
 * (2-term, 2-term + cut-off predicate, 3-term, 3-term + predicate, 4-term + predicate, near-zone pass), out[10..12] = the node
 * evaluations by body (2-, 3-, 4-term) on the vector unit -- what bench.py weights with the VALU instruction count of each
 * body -- and out[13], out[14] = the node and the per-point evaluations summed on the matrix cores (cs_set_matrix_cores;
 * out[0] and out[4..9] do not include the latter), out[15] = those of out[13] + out[14] that take three series terms instead of
 * four, out[16] = (lane, line) evaluations of k_voigt_sub (the window core on 16-point sub-tiles; 37 instructions each like the
 * near-zone pass; not in out[0]), out[17] = (tile, state) pairs whose window core is k_voigt_sub's, summed over the gases, out[18] =
 * those of out[14] that take eight series terms (the sub-tile cores), out[19] = the part of out[15] that belongs to out[13];
 * out[20], out[21] = (nu, line, state) pairs of k_voigt_near<0> (100 <= x^2+y^2 < 1e3) and <1> (< 100), counted on the host from the
 * records of the last Voigt launch group; out[22], out[23] = flops of k_voigt_edge_mx: useful (2 x series terms for every (point,
 * line, state) with the point inside the cut-off and outside the core radius, real states only) and issued (2048 per matrix
 * instruction: masked columns, padded states and the fill of the last 4-line step included); out[24], out[25] = the same for
 * k_cheb_nodes_mx; out[26] = flops of the node-sum -> grid contraction (k_cheb_apply_mfma, or inside k_voigt_edge_mx / k_flux_*); with
 * cs_set_tuning key 15 | 128, out[27..30] = nanoseconds block 0 of k_flux_scan spent on cross-sections, optical depths + Planck values,
 * first pass over its layer chunk, hand-over of the incoming intensities, and out[31] = from its first instruction to the last block's
 * store of the band fluxes (100 MHz wall clock; 0 otherwise).  `out` holds 32 values.
 * cs_column_counts is the reference's count. */
int cs_column_work(cs_ctx *ctx, int64_t *out);
/* interval sizes (descending, <= 5, each 128..2048 points) cs_set_interp(on) would use for this grid and cut-off; returns
 * their number (0: the grid is too coarse for the cut-off -- every pair is evaluated directly) */
int cs_interp_plan(int64_t nnu, const double *nu, double dnu_cut, int *interval_sizes);
/* the same for PHCO2! (line_shapes.jl:467-540) with the default settings: the "virtual levels" its far wings are interpolated on --
 * interval size (8192 .. 128 points), Chebyshev nodes per interval (64, 32 or 16: by how many half-widths the region's lines stay
 * away) and the chi-regions summed with that node count (bit 0: 3-30 cm^-1, bit 1: 30-120, bit 2: 120-cut-off).  Host only.  Returns
 * their number (at most 16; 0: every pair per point); the arrays hold `cap` entries. */
int cs_phco2_plan(int64_t nnu, const double *nu, double dnu_cut, int cap, int *interval_sizes, int *nodes, int *regions);
/* update only the temperature-dependent inputs of a resident column (RCM stepping, radiative_convective.jl:109-144) */
int cs_column_update_state(cs_ctx *ctx, const double *T_nodes, const double *mu_nodes, const double *T_levels,
                           const double *conc, const double *conc_tab);

/*
 * Native HITRAN .par ingestion (host only, no GPU needed).  Replaces the parsing loop of readpar (hitran/par.jl:127-152;
 * 160-column layout :131-149) with a memory-mapped, multi-threaded fixed-width parser.  Filters (nu range, intensity cut,
 * isotopologue selection, strongest-N) and the final stable sort stay with the caller, as in par.jl:153-191.
 *   cs_par_count : number of records in the file
 *   cs_par_parse : fills the caller's arrays of length n (= cs_par_count): M (molecule number), I (isotopologue character),
 *                  nu, S, A, gamma_a, gamma_s, Epp, na, delta_a
 */
int cs_par_count(const char *filename, int64_t *n);
int cs_par_parse(const char *filename, int64_t n, int16_t *M, char *I, double *nu, double *S, double *A, double *gamma_a,
                 double *gamma_s, double *Epp, double *na, double *delta_a);
/*
 * f3 to the letter -- a HITRAN .par file straight into a gas slot, no host-language arrays in between: readpar
 * (hitran/par.jl:91-193: parse, nu range / intensity cut / isotopologue filter :153-175, the `maxlines` strongest :177-186, stable
 * sort by wavenumber :188-191) + SpectralLines (:224-286: one molecule only, isotopologue numbers through ISOINDEX :6-13, molar
 * masses from MOLPARAM) + cs_gas_upload.
 *   iso_keep[n_iso_keep] : isotopologue numbers to keep (0 entries: all);  maxlines <= 0: no limit
 *   M_expected, mu_table[niso], ncheb[niso], cheb[niso][16] : the molecule and its MOLPARAM rows (src/hitran/molparam.jl)
 *   *L_out : lines kept.  Errors: CS_EINVAL with the reference's messages (filtered to nothing, several molecules).
 * cs_gas_fetch copies a slot's line arrays back (the host mirror of a table loaded this way; any pointer may be NULL).
 */
int cs_gas_upload_par(cs_ctx *ctx, int slot, const char *filename, double numin, double numax, double Scut, const int *iso_keep,
                      int n_iso_keep, int64_t maxlines, int M_expected, const double *mu_table, int niso, const int32_t *ncheb,
                      const double *cheb, int64_t *L_out);
int cs_gas_fetch(cs_ctx *ctx, int slot, int64_t L, double *nu, double *S, double *gamma_a, double *gamma_s, double *Epp, double *na,
                 double *mu_iso, int16_t *iso);

/*
 * B thermal states of the resident column in one go: the np+1 perturbed profiles of jacobian! or the successive profiles of
 * an RCM step loop (radiative_convective.jl:109-171).  Line sums for all B*K node states go through K1/K2 as one batch
 * (chunked to bound the workspace), baked tables and CIA pairs of the column are evaluated at all B*K states, then one k_rt
 * launch solves the B columns side by side.  (Host-evaluated sigma_extra terms cannot be batched: CS_EINVAL.)
 *   T_nodes, mu_nodes : [B][nlobatto*(np-1)]   T_levels : [B][np]   conc : [B][ngas*K]  (each column laid out as in
 *   cs_column_update_state);  conc_tab : [B][ntab*K] (as cs_column_set_tables; NULL without tables);  cia_P1, cia_P2 :
 *   [B][ncia*K] (as cs_column_set_cia; NULL without CIA pairs);  outputs Fup, Fdn : [B][np] on the host.
 * The resident column's own state is left untouched.
 */
int cs_column_batch(cs_ctx *ctx, int B, const double *T_nodes, const double *mu_nodes, const double *T_levels,
                    const double *conc, const double *conc_tab, const double *cia_P1, const double *cia_P2, double *Fup, double *Fdn);

/*
 * AcceleratedAbsorber (absorbers.jl:114-203): per wavenumber, ln Sigma on a set of pressure knots, interpolated linearly in
 * ln P (LinearInterpolator, NoBoundaries) -- what RCM holds and hands to radiate! on every step (radiative_convective.jl:95,113).
 *   cs_accel_store   : AcceleratedAbsorber(T, P, U) and update!(A, T) (:134-200).  The knots are the node states of the RESIDENT
 *                      column (set it up over U's members with nlobatto = 2, so that node k = knot k = (T_k, P_k)); evaluates
 *                      Sigma(U, i, T_k, P_k) for every wavenumber and knot in one pass of the line kernels and keeps
 *                      max(ln Sigma, ln floatmin) in HBM as slot `accel_slot`.  Call again after cs_column_update_state = update!.
 *   cs_accel_eval    : Sigma(A, i, T, P) = exp(phi_i(ln P)) for wavenumbers [i0, i0+n)  (:203; A(i, P), A(P) :205-207)
 *   cs_column_set_accel : the resident column (set up with ngas = 0) takes its cross-sections from the slot -- a column over an
 *                      AcceleratedAbsorber (unifyabsorbers(::Tuple{AcceleratedAbsorber}) :216).  In cs_column_batch such a column's
 *                      cross-sections are evaluated once and shared by the B thermal states, as in jacobian! (the reference never
 *                      calls update! between the perturbed radiate! calls, radiative_convective.jl:154-171).
 */
int cs_accel_store(cs_ctx *ctx, int accel_slot);
int cs_accel_clear(cs_ctx *ctx, int accel_slot);
/*   cs_accel_upload  : an AcceleratedAbsorber the HOST holds (the reference's struct as RCM keeps it, radiative_convective.jl:18,95):
 *                      lnsigma[nnu, nk] column-major (nu fastest) = the knot values of its interpolators phi_i (absorbers.jl:116,195)
 *                      on P_knots[nk] = A.P (ascending) -- whatever its members were.  The slot then behaves as one filled by
 *                      cs_accel_store.
 *   cs_accel_fetch   : the knot values of a slot back to the host, same layout (so that a device-side update! can keep the
 *                      reference's host struct current); nnu, nk = what the caller's buffer was allocated for. */
int cs_accel_upload(cs_ctx *ctx, int accel_slot, int64_t nnu, const double *nu, int nk, const double *P_knots, const double *lnsigma);
int cs_accel_fetch(cs_ctx *ctx, int accel_slot, int64_t nnu, int nk, double *lnsigma);
int cs_accel_eval(cs_ctx *ctx, int accel_slot, double P, int64_t i0, int64_t n, double *sigma_out);
int cs_column_set_accel(cs_ctx *ctx, int accel_slot);

/* Scalar helpers exported for tests of the host logic (same formulas the kernels use). */
int cs_streamnodes(int n, double *m, double *W);    /* core/shared.jl:4-21 */
int cs_lobattonodes(int n, double *x, double *w);   /* core/discretized.jl:2-9 */
/* Re w(x+iy) evaluated on the device for n points (host pointers): the kernels' Faddeeva, test hook for
 * Faddeyeva985.faddeyeva (call site line_shapes.jl:375). */
int cs_faddeeva_batch(cs_ctx *ctx, int64_t n, const double *x, const double *y, double *out);
/* The device functions of the flux kernel, point by point (host pointers; test hook): which = 0: its own exp (Cody-Waite reduction +
 * degree-13 polynomial, replaces libm's in every transmission and Planck value) at x; 1: planck(nu = x, T = y) radiation.jl:48-54;
 * 2: layerplanck (discretized.jl:85-87) for B1 = x, B2 = y, tau = z > 0 with t = exp(-tau).  y, z may be NULL where unused. */
int cs_devfn_batch(cs_ctx *ctx, int which, int64_t n, const double *x, const double *y, const double *z, double *out);

#ifdef __cplusplus
}
#endif
#endif
