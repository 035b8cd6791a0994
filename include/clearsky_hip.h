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
 *
 * This header is the PRODUCT surface.  The same library also exports the laboratory bench -- A/B tuning switches, per-class kernel
 * timers, launch / evaluation / flop counters, interpolation-plan queries, device-function test hooks -- declared in
 * clearsky_hip_dev.h; nothing a ClearSky.jl caller needs lives there.
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
int cs_column_flux_ptr(cs_ctx *ctx, double **dF);
/* asynchronously copy the [2*np] band fluxes (Fup then Fdn) into caller-owned DEVICE memory on `stream` */
int cs_column_flux_to(cs_ctx *ctx, double *dst_device, void *stream);
/* from now on the resident column WRITES its [2*np] band fluxes into caller-owned device memory (a buffer a collective reduces in place,
 * a tensor of the host framework) instead of its own -- no copy per step; NULL: back to its own.  Holds until the next cs_column_setup.
 * cs_column_flux_ptr / cs_column_fetch / cs_column_flux_to read from wherever the fluxes are. */
int cs_column_set_flux_dst(cs_ctx *ctx, double *dst_device);
int cs_column_fetch(cs_ctx *ctx, int64_t nnu, int np, double *tau, double *Mup, double *Mdn, double *Fup, double *Fdn);
int cs_column_sigma_fetch(cs_ctx *ctx, int64_t nnu, int K, double *sigma);
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

/* Quadrature nodes of the two rules of the path, as the host side of a binding needs them (same formulas the kernels use). */
int cs_streamnodes(int n, double *m, double *W);    /* core/shared.jl:4-21 */
int cs_lobattonodes(int n, double *x, double *w);   /* core/discretized.jl:2-9 */

#ifdef __cplusplus
}
#endif
#endif
