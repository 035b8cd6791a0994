"""GPU parity of the interpolated far wings (k_cheb_nodes + k_cheb_apply).

The interpolation replaces per-point evaluations of surf! (line_shapes.jl:56-96) by sums at Chebyshev nodes; these tests
pin it (a) against the oracle, which evaluates every pair, and (b) against the same device path with interpolation off,
on grids chosen so that 1..5 interval levels are active, intervals are ragged, the cut-off is narrow or wide.
Tolerance: 1e-11 vs the oracle (the repo-wide cross-section tolerance), 5e-14 on/off (interpolation error proper;
1.3e-14 observed with five levels stacked).
"""
import math

import numpy as np
import pytest

from conftest import relerr, source_rounding_bound

pytestmark = pytest.mark.gpu

STATES = [(220.0, 50.0, 0.02), (296.0, 101325.0, 40.53), (260.0, 3e3, 30.0), (190.0, 2.0, 0.0)]


@pytest.fixture(scope="module")
def ctx_on(cs):
    c = cs.Context(0)
    c.set_matrix_cores(2)   # (the matrix-core kernels also on these short grids; the vector-only path: test_matrix_core_node_sums_on_off)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_off(cs):
    c = cs.Context(0)
    c.set_interp(False)
    yield c
    c.close()


def test_plan():
    import clearsky_jl_amd as cs
    assert cs.interp_plan(np.linspace(1, 2500, 100000), 25.0) == [512, 256, 128]      # the bench grid: 3 levels
    assert cs.interp_plan(np.linspace(1, 2500, 5003), 25.0) == []                      # coarse grid: all direct
    assert cs.interp_plan(np.linspace(600, 700, 100000), 25.0) == [2048, 1024, 512, 256, 128]
    assert cs.interp_plan(np.linspace(600, 700, 100), 25.0) == []                      # fewer than 128 points


GRIDS = {
    # name: (nu, cut, expected number of levels)
    "fine-5-levels": (np.linspace(640.0, 700.0, 60001), 25.0, 5),
    "bench-spacing": (np.linspace(600.0, 760.0, 6401), 25.0, 3),
    "ragged-1-point-tail": (np.linspace(660.0, 680.0, 4 * 2048 + 1), 25.0, 5),
    "ragged-short": (np.linspace(666.0, 669.0, 1153), 25.0, 5),
    "narrow-cut": (np.linspace(2300.0, 2380.0, 40000), 3.0, 3),
    "wide-cut": (np.linspace(500.0, 900.0, 8000), 120.0, 4),
    "nonuniform": (np.sort(np.concatenate([600.0 + 160.0 * np.random.default_rng(5).random(9000) ** 2, [600.0, 760.0]])), 25.0, None),
}


@pytest.mark.parametrize("name", list(GRIDS))
def test_shape_batch_interp_vs_oracle_and_direct(cs, O, lines, ctx_on, ctx_off, name):
    nu, cut, nlev = GRIDS[name]
    if nlev is not None:
        assert len(cs.interp_plan(nu, cut)) == nlev
    sl = lines("CO2")
    T, P, Pp = map(list, zip(*STATES))
    on = cs.shape_batch(sl, "voigt", nu, T, P, Pp, cut, ctx_on)
    off = cs.shape_batch(sl, "voigt", nu, T, P, Pp, cut, ctx_off)
    assert np.array_equal(on == 0, off == 0)
    assert relerr(on, off, floor=1e-250) < 5e-14
    for k in (0, 1, 3):
        so = O.shape_bang("voigt", nu, sl, T[k], P[k], Pp[k], cut)
        assert np.array_equal(on[k] == 0, so == 0)
        assert relerr(on[k], so, floor=1e-250) < 1e-11


def test_dense_synthetic_lines(cs, O, ctx_on, ctx_off):
    """Line density of the bench workload (40 lines per cm^-1), strong/weak lines side by side."""
    sl = cs.SpectralLines.synthetic(1, 20000, seed=9, numin=400.0, numax=900.0)
    nu = np.linspace(600.0, 700.0, 20000)
    T, P, Pp = [250.0, 288.0], [1e4, 1e5], [100.0, 1e3]
    on = cs.shape_batch(sl, "voigt", nu, T, P, Pp, 25.0, ctx_on)
    off = cs.shape_batch(sl, "voigt", nu, T, P, Pp, 25.0, ctx_off)
    assert relerr(on, off, floor=1e-250) < 5e-14
    idx = np.sort(np.random.default_rng(2).choice(nu.size, 400, replace=False))
    for k in range(2):
        assert relerr(on[k][idx], O.shape_bang("voigt", nu[idx], sl, T[k], P[k], Pp[k], 25.0), floor=1e-250) < 1e-11


def test_column_interp_on_off(cs, lines):
    """Whole column (two gases, accumulation across gases, batch path): on/off agree to rounding; the work counters show
    that interpolation really ran."""
    nu = np.linspace(580.0, 780.0, 20000)
    P = cs.pressuregrid(10.0, 1e5, 21)
    T = cs.AtmosphericProfile(P, np.linspace(200.0, 290.0, 21))
    res = {}
    for on in (True, False):
        c = cs.Context(0)
        c.set_interp(on)
        gases = [cs.DirectGas(lines("CO2"), 400e-6, nu), cs.DirectGas(lines("H2O"), 5e-3, nu)]
        col = cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, *gases, core=cs.Discretized(4, 3), want_tau=True, want_M=True, ctx=c)
        col.run()
        F = cs.FluxPack(col.np, col.nnu)
        F.Fup[:], F.Fdn[:] = col.fetch(F.tau, F.Mup, F.Mdn)
        w = col.work()
        B = col.run_batch([T, cs.AtmosphericProfile(P, np.linspace(205.0, 288.0, 21))])
        res[on] = (F, w, B)
        c.close()
    Fon, won, Bon = res[True]
    Foff, woff, Boff = res[False]
    assert won["levels"] >= 3 and won["node_evals"] > 0 and woff["levels"] == 0 and woff["node_evals"] == 0
    assert won["direct_evals"] + won["node_evals"] < 0.6 * woff["direct_evals"]
    assert relerr(Fon.tau, Foff.tau) < 5e-14
    sm = Foff.Mup.max()
    # tau agrees to 5e-14 relative (above).  (i) A relative change e of one layer's tau changes its transmission exp(-tau m) by
    # e * (tau m) exp(-tau m) <= e / e_euler, and the linear-in-tau source term by as much again; an intensity crosses nl = 20
    # layers: 2 * nl / e_euler * 5e-14 * max M.  (ii) The source term (1 - t)(B1 - B2)/tau (discretized.jl:85-87) divides a
    # difference of order tau by tau: when the two runs' tau differ in the last bit, t = exp(-tau m) can round the other way
    # (2^-53), which moves (1 - t)/tau by 2^-53/tau -- 1e-10 just above the 1e-6 floor -- times |B1 - B2| per layer and stream.
    # Both are rounding of the reference's own formula, not of the line sums; (ii) was left out of this bound until the merged
    # line table changed which last bits differ (observed then: 7e-13 against 1.1e-13 before).
    bound = 2 * col.nl / math.e * 5e-14 * sm + source_rounding_bound(cs, nu, [T(p) for p in P], Foff.tau)
    assert np.max(np.abs(Fon.Mup - Foff.Mup)) < bound and np.max(np.abs(Fon.Mdn - Foff.Mdn)) < bound
    assert relerr(Fon.Fup, Foff.Fup) < 1e-13
    for a, b in zip(Bon, Boff):
        assert relerr(np.asarray(a), np.asarray(b), floor=1e-9) < 1e-13
    assert relerr(np.asarray(Bon[0])[0], Fon.Fup) < 1e-13      # batch member 0 = the resident state


@pytest.mark.parametrize("seed", list(range(1, 13)))
def test_random_columns_on_off(cs, seed):
    """Seeded random columns off the bench's beaten track -- geometric and uniform grids that end inside an interval, cut-offs of
    8 .. 60 cm^-1 (two to five interval sizes, the cascade from four up), 5 .. 40 layers (ragged state groups), dense and thin
    synthetic tables merged into one launch group, two or three Lobatto nodes per layer: the cross-sections of the full machinery
    (interpolated wings, low-order far pieces, matrix-core pieces with phased cut-off edges, sub-tile cores, side streams) against
    every pair evaluated per point on the vector unit."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(9000, 40000))
    lo = float(rng.uniform(200.0, 1500.0))
    hi = lo + float(rng.uniform(60.0, 400.0))
    nu = np.geomspace(lo, hi, n) if seed % 2 else np.linspace(lo, hi, n)
    cut = float(rng.choice([8.0, 15.0, 25.0, 40.0, 60.0]))
    nlay = int(rng.integers(5, 41))
    P = cs.pressuregrid(float(rng.uniform(0.5, 50.0)), float(rng.uniform(2e4, 2e5)), nlay + 1)
    T = cs.AtmosphericProfile(P, np.linspace(float(rng.uniform(180.0, 230.0)), float(rng.uniform(260.0, 320.0)), nlay + 1))
    dens = cs.SpectralLines.synthetic(2, int(rng.integers(4000, 30000)), 50 + seed, numin=lo - 80.0, numax=hi + 80.0)
    thin = cs.SpectralLines.synthetic(1, int(rng.integers(50, 800)), 80 + seed, numin=lo - 80.0, numax=hi + 80.0)
    conc = float(rng.uniform(1e-4, 1e-2))
    res = []
    for on in (True, False):
        c = cs.Context(0)
        c.set_interp(on)
        if not on:
            c.set_matrix_cores(0)
        g1 = cs.DirectGas(dens, conc, nu, dnu_cut=cut)
        g2 = cs.DirectGas(thin, 3e-3, nu, dnu_cut=cut)
        col = cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, g1, g2, core=cs.Discretized(4, 2 + seed % 2), ctx=c)
        col.run()
        res.append((col.sigma_nodes(), col.fetch(), col.work()))
        c.close()
    assert (res[0][2]["node_evals"] > 0 or res[0][2]["levels"] <= 1) and res[1][2]["levels"] == 0   # (one usable size and a thin table: the plan may skip it)
    assert np.array_equal(res[0][0] == 0, res[1][0] == 0)
    assert relerr(res[0][0], res[1][0], floor=1e-300) < 2e-13
    assert relerr(res[0][1][0], res[1][1][0]) < 1e-12


def test_bake_with_interp(cs, O, lines, ctx_on, ctx_off):
    nu = np.linspace(640.0, 700.0, 6000)
    Om = cs.AtmosphericDomain((180.0, 320.0), 4, (10.0, 1e5), 5)
    gon = cs.Gas(lines("CO2"), 400e-6, nu, Om, ctx=ctx_on, keep_host_tables=True)
    goff = cs.Gas(lines("CO2"), 400e-6, nu, Om, ctx=ctx_off, keep_host_tables=True)
    a, b = np.asarray(gon.lnsigma), np.asarray(goff.lnsigma)
    assert np.max(np.abs(a - b)) < 5e-14       # ln sigma: absolute = relative in sigma


def test_more_than_65535_intervals(cs, lines, ctx_on, ctx_off):
    """5e6 wavenumbers, five levels: 75 000 intervals in the node kernel's (1-D) grid; on vs off."""
    nu = np.linspace(640.0, 700.0, 5_000_000)
    assert len(cs.interp_plan(nu, 25.0)) == 5
    sl = lines("CO2")
    on = cs.shape_batch(sl, "voigt", nu, [250.0], [2e4], [8.0], 25.0, ctx_on)
    off = cs.shape_batch(sl, "voigt", nu, [250.0], [2e4], [8.0], 25.0, ctx_off)
    assert relerr(on, off, floor=1e-250) < 5e-14


def test_first_level_choice(cs, lines):
    """A gas may skip the largest interval sizes (choose_l0, by line density).  Forcing every possible first level -- including
    "none" -- gives the same cross-sections; the automatic choice drops levels for the sparse H2O fixture on a fine grid."""
    nu = np.linspace(1500.0, 1560.0, 30001)          # 0.002 cm^-1: five levels on the grid
    assert len(cs.interp_plan(nu, 25.0)) == 5
    P = cs.pressuregrid(10.0, 1e5, 9)
    T = np.linspace(210.0, 290.0, 9)
    ref = None
    for l0 in (None, 0, 1, 3, 4, 5):
        ctx = cs.Context(0)
        ctx.set_interp_plan(first_level=-1 if l0 is None else l0)
        col = cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, cs.DirectGas(lines("H2O"), 5e-3, nu), cs.DirectGas(lines("CH4"), 2e-6, nu),
                        core=cs.Discretized(3, 2), ctx=ctx)
        col.run()
        sig, w = col.sigma_nodes(), col.work()
        ctx.close()
        if l0 == 5:
            assert w["node_evals"] == 0                # no level left: every pair evaluated directly
        if l0 == 0:
            full = w["node_evals"]
        if l0 is None:
            auto = w["node_evals"]
        if ref is None:
            ref = sig
        else:
            assert relerr(sig, ref, floor=1e-250) < 5e-14, l0
    assert auto > full > 0          # the sparse tables skip the largest intervals: more node evaluations, fewer matrix passes


@pytest.mark.parametrize("gas", ["CO2", "H2O"])
def test_lorentz_fast_path(cs, O, lines, ctx_on, ctx_off, gas):
    """lorentz! (line_shapes.jl:273,313-324) on the far-wing machinery with its own exact body: interpolated far wings (fine grid,
    several interval levels) vs the oracle, which evaluates every pair, and vs the device with interpolation off."""
    sl = lines(gas)
    nu = np.linspace(640.0, 700.0, 24001) if gas == "CO2" else np.linspace(1500.0, 1560.0, 24001)
    assert len(cs.interp_plan(nu, 25.0)) >= 4
    T, P, Pp = map(list, zip(*STATES))
    on = cs.shape_batch(sl, "lorentz", nu, T, P, Pp, 25.0, ctx_on)
    off = cs.shape_batch(sl, "lorentz", nu, T, P, Pp, 25.0, ctx_off)
    assert relerr(on, off, floor=1e-250) < 5e-14
    for k in range(len(T)):
        so = O.shape_bang("lorentz", nu, sl, T[k], P[k], Pp[k], 25.0)
        assert np.array_equal(on[k] == 0, so == 0)
        assert relerr(on[k], so, floor=1e-250) < 5e-12
    # a narrow and a wide cut-off, ragged tail
    for cut, n in ((3.0, 7777), (120.0, 9001)):
        nu2 = np.linspace(2300.0, 2330.0, n)
        a = cs.shape_batch(lines("CO2"), "lorentz", nu2, T[:2], P[:2], Pp[:2], cut, ctx_on)
        for k in range(2):
            assert relerr(a[k], O.shape_bang("lorentz", nu2, lines("CO2"), T[k], P[k], Pp[k], cut), floor=1e-250) < 5e-12


def test_lorentz_and_doppler_columns_vs_oracle(cs, O, lines):
    """Whole columns with the other shapes on a fine grid (the Doppler sum only walks the lines within the profile's non-zero
    reach, sqrt(750) Doppler widths: beyond, exp(-x^2) is an exact zero in fp64)."""
    ctx = cs.Context(0)
    nu = np.linspace(660.0, 680.0, 8001)
    P = cs.pressuregrid(1.0, 1e5, 13)
    T = np.linspace(205.0, 290.0, 13)
    for shape, tol in (("lorentz", 1e-11), ("doppler", 1e-9)):
        g1 = cs.DirectGas(lines("CO2"), 400e-6, nu, shape=shape)
        g2 = cs.DirectGas(lines("H2O"), 3e-3, nu, shape=shape)
        col = cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, g1, g2, core=cs.Discretized(5, 2), ctx=ctx)
        col.run()
        tau = np.zeros((col.nl, col.nnu), order="F")
        F = col.fetch(tau)
        r = O.fluxes_discretized(nu, P, 9.8, 2, col.Tn, col.mun, col.Tlev, [g1.sl, g2.sl], [shape] * 2, [25.0] * 2, col.conc, want_sigma=True)
        s = col.sigma_nodes()
        assert np.array_equal(s == 0, r["sigma"] == 0)
        assert relerr(s, r["sigma"], floor=1e-300) < tol
        assert relerr(tau, r["tau"]) < tol and abs(F[0][0] - r["Fup"][0]) < 1e-10 * r["Fup"][0]
    ctx.close()


def test_phco2_fast_path(cs, O, lines, ctx_on):
    """PHCO2! (line_shapes.jl:467-540) at its default 500 cm^-1 cut-off on fine grids: far lines run region by region with the
    factorised chi (k_phco2), the rest through the generic per-lane body -- against the oracle, which evaluates chi and the full
    profile for every pair.  Grids cross every region boundary (3, 30, 120 cm^-1 from line clusters) and the cut-off edge."""
    sl = lines("CO2")
    T, P, Pp = map(list, zip(*STATES))
    for lo, hi, n in ((600.0, 760.0, 6401), (2200.0, 2420.0, 7001), (30.0, 95.0, 3003)):
        nu = np.linspace(lo, hi, n)
        a = cs.shape_batch(sl, "PHCO2", nu, T, P, Pp, 500.0, ctx_on)
        for k in range(len(T)):
            so = O.shape_bang("PHCO2", nu, sl, T[k], P[k], Pp[k], 500.0)
            assert np.array_equal(a[k] == 0, so == 0)
            assert relerr(a[k], so, floor=1e-250) < 2e-11
    # narrower cut-offs: 200 cm^-1 still takes the fast path, 100 cm^-1 falls back to the generic kernel; same answers
    nu = np.linspace(640.0, 700.0, 4001)
    for cut in (200.0, 100.0):
        a = cs.shape_batch(sl, "PHCO2", nu, T[:2], P[:2], Pp[:2], cut, ctx_on)
        for k in range(2):
            assert relerr(a[k], O.shape_bang("PHCO2", nu, sl, T[k], P[k], Pp[k], cut), floor=1e-250) < 2e-11
    # a whole column (scalar-nu semantics, accumulation onto a second gas)
    Pl = cs.pressuregrid(10.0, 1e5, 9)
    Tl = np.linspace(210.0, 290.0, 9)
    g1 = cs.DirectGas(sl, 0.5, nu, shape="PHCO2")
    g2 = cs.DirectGas(lines("H2O"), 1e-3, nu)
    col = cs.Column(Pl, 9.8, Tl, 0.04, 0.0, 0.0, g2, g1, core=cs.Discretized(5, 2), ctx=ctx_on)
    col.run()
    r = O.fluxes_discretized(nu, Pl, 9.8, 2, col.Tn, col.mun, col.Tlev, [g2.sl, g1.sl], ["voigt", "PHCO2"], [25.0, 500.0], col.conc,
                             want_sigma=True)
    assert relerr(col.sigma_nodes(), r["sigma"], floor=1e-300) < 2e-11


def test_phco2_interpolated_wings(cs, O, lines, ctx_on, ctx_off):
    """PHCO2 far wings through the Chebyshev machinery (k_phco2_nodes): inside one chi-region and on one side of a line the term is
    analytic in nu, so the lines that are region-uniform for a whole interval of 128 .. 2048 points are summed at its 64 nodes
    (own ranges minus the parent's; 16 or 32 nodes where the lines are many half-widths away) and carried to the grid; k_phco2
    keeps the tile's sets minus the smallest interval's and the boundary sets (two-way chi select); the pairs within 3 cm^-1, where
    chi = 1, go through the Voigt kernels with that cut-off.  Same cross-sections as with every pair summed
    per point (5e-13: different summation orders) and as the oracle (2e-11), on a dense table and the sparse fixture, on grids that
    end inside an interval, for interval size ranges that leave one or five levels, and in a column."""
    dense = cs.SpectralLines.synthetic(2, 8000, 91, numin=200.0, numax=1400.0)
    T, P, Pp = map(list, zip(*STATES))
    for sl, lo, hi, n in ((dense, 640.0, 800.0, 6401), (dense, 655.0, 700.0, 3119), (lines("CO2"), 2200.0, 2420.0, 7001)):
        nu = np.linspace(lo, hi, n)
        a = cs.shape_batch(sl, "PHCO2", nu, T, P, Pp, 500.0, ctx_on)
        b = cs.shape_batch(sl, "PHCO2", nu, T, P, Pp, 500.0, ctx_off)
        assert np.array_equal(a == 0, b == 0)
        assert relerr(a, b, floor=1e-250) < 5e-13
        for k in (1, 3):
            so = O.shape_bang("PHCO2", nu, sl, T[k], P[k], Pp[k], 500.0)
            assert relerr(a[k], so, floor=1e-250) < 2e-11
    nu = np.linspace(640.0, 800.0, 6401)
    ref = cs.shape_batch(dense, "PHCO2", nu, T[:2], P[:2], Pp[:2], 500.0, ctx_off)
    for smin, smax, tune in ((128, 128, {}), (2048, 2048, {}), (256, 1024, {}), (128, 2048, {10: 1}), (128, 2048, {10: 2}), (128, 2048, {10: 3}),
                             (128, 2048, {9: 1})):
        c = cs.Context(0)
        c.set_interp_plan(size_min=smin, size_max=smax)
        for key, val in tune.items():   # 64 nodes everywhere / tiles as 64-point intervals / k_phco2's own core loop
            c.set_tuning(key, val)
        a = cs.shape_batch(dense, "PHCO2", nu, T[:2], P[:2], Pp[:2], 500.0, c)
        c.close()
        assert relerr(a, ref, floor=1e-250) < 5e-13
    # a column: two PHCO2 gases on the same grid (the levels are built once per column setup) beside a Voigt gas
    Pl = cs.pressuregrid(10.0, 1e5, 9)
    Tl = np.linspace(210.0, 290.0, 9)
    res = []
    for ctx in (ctx_on, ctx_off):
        g1 = cs.DirectGas(dense, 0.3, nu, shape="PHCO2")
        g2 = cs.DirectGas(lines("CO2"), 0.2, nu, shape="PHCO2")
        g3 = cs.DirectGas(lines("H2O"), 1e-3, nu)
        col = cs.Column(Pl, 9.8, Tl, 0.04, 0.0, 0.0, g3, g1, g2, core=cs.Discretized(5, 2), ctx=ctx)
        col.run()
        col.run()
        res.append((col.sigma_nodes(), col.fetch()))
    assert relerr(res[0][0], res[1][0], floor=1e-300) < 5e-13
    assert abs(res[0][1][0][0] - res[1][1][0][0]) < 1e-12 * res[1][1][0][0]


def test_phco2_ragged_grids(cs, O, lines, ctx_on, ctx_off):
    """PHCO2 interval machinery on grids that are not the bench's: geometric spacing (interval widths differ along the grid, so the
    node counts follow the widest interval of a size and a region may be carried at one size and not the next), a grid that starts
    below the cut-off (no lower bound on the Doppler width: four-term body everywhere), a grid shorter than one interval (every pair per
    point) and one with tiles wider than a region (the generic kernel) -- each against the per-point path and the oracle."""
    dense = cs.SpectralLines.synthetic(2, 6000, 17, numin=1.0, numax=1500.0)
    T, P, Pp = map(list, zip(*STATES[:3]))
    grids = (np.geomspace(400.0, 900.0, 9001), np.linspace(5.0, 260.0, 10201), np.linspace(700.0, 702.0, 90),
             np.linspace(100.0, 1300.0, 2400))
    for nu in grids:
        a = cs.shape_batch(dense, "PHCO2", nu, T, P, Pp, 500.0, ctx_on)
        b = cs.shape_batch(dense, "PHCO2", nu, T, P, Pp, 500.0, ctx_off)
        assert np.array_equal(a == 0, b == 0)
        assert relerr(a, b, floor=1e-250) < 5e-13
        so = O.shape_bang("PHCO2", nu, dense, T[1], P[1], Pp[1], 500.0)
        assert relerr(a[1], so, floor=1e-250) < 2e-11


@pytest.mark.parametrize("seed", list(range(1, 9)))
def test_phco2_random_on_off(cs, ctx_on, ctx_off, seed):
    """Seeded random PHCO2 evaluations -- uniform and geometric grids of 3e3 .. 3e4 points anywhere between 20 and 3000 cm^-1,
    cut-offs of 150 .. 600 cm^-1 (which interval sizes exist, which regions they carry and with how many nodes all follow from these),
    tables of 2e2 .. 2e4 lines, cold / warm, thin / thick states: interval machinery against every pair per point."""
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.integers(3000, 30000))
    lo = float(rng.uniform(20.0, 2500.0))
    hi = lo + float(rng.uniform(30.0, 500.0))
    nu = np.geomspace(lo, hi, n) if seed % 2 else np.linspace(lo, hi, n)
    cut = float(rng.choice([150.0, 250.0, 500.0, 600.0]))
    sl = cs.SpectralLines.synthetic(2, int(rng.integers(200, 20000)), 300 + seed, numin=max(lo - cut, 1.0), numax=hi + cut)
    K = int(rng.integers(1, 7))
    T = rng.uniform(160.0, 340.0, K).tolist()
    P = (10.0 ** rng.uniform(0.0, 5.2, K)).tolist()
    Pp = [float(p * rng.uniform(0.0, 1.0)) for p in P]
    a = cs.shape_batch(sl, "PHCO2", nu, T, P, Pp, cut, ctx_on)
    b = cs.shape_batch(sl, "PHCO2", nu, T, P, Pp, cut, ctx_off)
    assert np.array_equal(a == 0, b == 0)
    assert relerr(a, b, floor=1e-250) < 5e-13


def test_matrix_core_node_sums_on_off(cs, O, lines):
    """K2d, K2e, K2f: far lines inside the validity range of the 4-term series in 1/dnu^2 are summed as matrix products on
    v_mfma_f64_16x16x4 -- at the interpolation nodes (k_cheb_nodes_mx) and, for the window ends of the per-point sum with the
    cut-off as a mask, at the points themselves (k_voigt_edge_mx); the rest stays on the vector unit.  Same cross-sections as with every node sum on the
    vector unit (5e-14) and as the oracle (1e-11), for state groups that mix low and high pressures, a ragged last group (K = 41),
    a dense synthetic table and the sparse fixtures, the column, shape-batch and batch paths."""
    nu = np.linspace(580.0, 780.0, 20000)
    P = cs.pressuregrid(1.0, 1e5, 21)
    T = cs.AtmosphericProfile(P, np.linspace(200.0, 295.0, 21))
    dense = cs.SpectralLines.synthetic(2, 20000, 77, numin=500.0, numax=860.0)
    res = {}
    for on in (True, False, "tile-wide"):
        c = cs.Context(0)
        c.set_matrix_cores({True: 2, False: 0, "tile-wide": 2 | 4}[on])   # (2: also on this grid, too short for the default to choose
                                                                            #  the matrix path; | 4: without the sub-tile cores, K2f)
        gases = [cs.DirectGas(dense, 400e-6, nu), cs.DirectGas(lines("H2O"), 5e-3, nu)]
        col = cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, *gases, core=cs.Discretized(4, 3), want_tau=True, want_M=False, ctx=c)
        assert col.K == 41
        col.run()
        tau = np.zeros((col.nl, col.nnu), order="F")
        F = col.fetch(tau)
        B = col.run_batch([T, cs.AtmosphericProfile(P, np.linspace(205.0, 288.0, 21))])
        sb = cs.shape_batch(dense, "voigt", nu, [220.0, 296.0, 250.0], [50.0, 101325.0, 3e3], [0.02, 40.53, 1.2], 25.0, c)
        wk = col.work()
        assert (wk["node_evals_matrix"] > 0 and wk["direct_evals_matrix"] > 0) == bool(on)   # (both matrix-core kernels really ran)
        assert (wk["sub_evals"] > 0) == (on is True)                                              # (... and the sub-tile kernel)
        res[on] = (col.sigma_nodes(), tau, F, B, sb, col)
        if on is True:
            r = O.fluxes_discretized(nu, P, 9.8, 3, col.Tn, col.mun, col.Tlev, [dense, lines("H2O")], ["voigt"] * 2, [25.0] * 2, col.conc,
                                     nstream=4, want_sigma=True)
            assert relerr(res[on][0], r["sigma"], floor=1e-300) < 1e-11 and relerr(tau, r["tau"]) < 1e-11
        c.close()
    for a, b in ((res[True], res["tile-wide"]), (res["tile-wide"], res[False])):
        assert relerr(a[0], b[0], floor=1e-300) < 5e-14 and relerr(a[1], b[1]) < 5e-14
        assert not np.array_equal(a[0], b[0])
    a, b = res[True], res[False]
    assert relerr(a[0], b[0], floor=1e-300) < 5e-14 and relerr(a[1], b[1]) < 5e-14
    assert relerr(a[2][0], b[2][0]) < 1e-13
    assert relerr(np.asarray(a[3][0]), np.asarray(b[3][0])) < 1e-13
    assert relerr(a[4], b[4], floor=1e-250) < 5e-14
    assert not np.array_equal(a[0], b[0])            # (the two paths really differ in rounding: the matrix-core one ran)
