/*
 * clearsky_hip_dev.h -- the laboratory bench of libclearsky_hip.so: measurement hooks, A/B switches and test hooks.
 *
 * Same library, second header (round 5: product and lab separated).  Nothing here is needed to USE the path: bench.py, tools/ and the
 * parity tests call these; julia/ClearSkyHIP.jl and INTEGRATION.md's bindings do not.  Results never depend on any switch below beyond
 * rounding (tests/test_gpu_interp.py, tests/test_gpu_merge.py, tests/test_gpu_fuzz.py compare them with the defaults and the oracle).
 */
#ifndef CLEARSKY_HIP_DEV_H
#define CLEARSKY_HIP_DEV_H

#include "clearsky_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

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
 *          interval size is carried to the grid (k_cheb_cascade) -- 0 (default) with four or more levels in use and, on the node-sum side
 *          stream (key 2), from two levels on (it is off the critical path there), 1 always, 2 never;
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
 *          | 256: on short grids the interval levels are NOT folded into the smallest one on the node-sum side stream (A/B);
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
 *   key 19: the near-line plane of a step is cleared by a memset in front of its first kernel (1) instead of written by that kernel,
 *          k_voigt_sub (0, default: one launch and 8 B per (state, point) of writes less) (A/B; same results);
 *   key 21: the piece tables of the matrix-core kernels as blocks of k_gas_setup's launch, computing the zones they need themselves
 *          (k_gas_setup_mx) -- 0 (default) on grids below 1024 tiles, 1 never (k_mxzones16 as its own launch), 2 always (A/B; same tables);
 *   key 22: waves per 64-point tile of k_voigt_far: 1, 2 or 4 (0, default: by the number of (tile, state) waves) (A/B).
 *   key 23: the window-end lines of k_voigt_edge_mx inside the cut-off of every point of a tile at the points, masked, like the others (1)
 *          instead of on 16 Chebyshev nodes of the tile (0, default, where the grid allows: cut-off far beyond tile + smallest interval) (A/B).
 *   key 17: where the four waves of a block share every item of k_cheb_nodes_mx (short grids), the far pieces of an interval on all 64
 *          nodes (1) instead of on its level's 16 or 32, a quarter of the lines per wave (0, default) (A/B).
 *   (keys 18, 20 are unused.)
 * Applies to every later cs_column_setup / cs_column_run of the context. */
int cs_set_tuning(cs_ctx *ctx, int key, int value);

/* run `reps` evaluations with HIP events between the kernel classes on `stream`; ms[10] = average milliseconds per
 * evaluation spent in {k_gas_setup (+ k_mxzones), k_cheb_nodes, k_cheb_apply (+ k_table_eval, k_cia), k_voigt_far (or
 * k_linesum), k_voigt_near, k_rt, k_freduce, k_cheb_nodes_mx, k_voigt_edge_mx, k_voigt_sub}, summed over gases */
int cs_column_profile(cs_ctx *ctx, void *stream, int reps, double *ms);

int cs_column_counts(cs_ctx *ctx, int64_t *pair_evals, int64_t *lines_in_range);

/* measurement hook, out[8]: out[0] = launch groups of the resident column (merged gases count once), out[1] = kernel launches of
 * the last cs_column_run, out[2] = lines of all groups, out[3] = cs_set_merge at setup time, out[4] = gases in the largest group,
 * out[5] = the flux kernel of the last run: 0 = k_rt / k_rt_streams on finished cross-sections, 2 = k_flux_chunk, 3 = k_flux_scan;
 * out[6] = near-line launches of the last run, all groups (k_voigt_near_both: 1 per group; k_voigt_near<0> + <1>: 2); out[7] = what
 * summed the per-point far lines of its last group: 0 = k_voigt_far, 1 = k_linesum<shape> (Doppler; PHCO2 without its fast path), 2 = k_phco2 */
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
 * store of the band fluxes (100 MHz wall clock; 0 otherwise); out[32], out[33] = bytes of per-(state, line) records k_voigt_edge_mx and
 * k_cheb_nodes_mx REQUEST per launch (lines of every piece x 16 states x 32 B: neighbouring tiles / intervals ask for the same record again --
 * the unique ones are K x lines in range x 32 B).  `out` holds 40 values.
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
