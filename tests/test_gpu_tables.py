"""GPU tests of "Mode T": baked Gas objects (bake -> OpacityTable -> 2-D Chebyshev interpolation, gases.jl:68-145,205-281)
against a numpy restatement built on the oracle's shape!.  Tolerance 1e-10 on ln(sigma) tables and interpolated sigma
(the interpolating polynomial is unique; only rounding differs from BasicInterpolators -- "parity unpinned" for that package)."""
import math

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(cs):
    c = cs.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def baked(cs, O, lines, ctx):
    nu = np.linspace(500.0, 900.0, 1500)
    Om = cs.AtmosphericDomain((180.0, 320.0), 8, (1.0, 1.1e5), 12)
    g = cs.Gas(lines("CO2"), 400e-6, nu, Om, ctx=ctx, keep_host_tables=True)
    ref = O.bake(lines("CO2"), np.full((8, 12), 400e-6), nu, Om.T, Om.P)
    return nu, Om, g, ref


def test_domain_grid(cs):
    Om = cs.AtmosphericDomain((25, 550), 12, (1, 1e6), 24)
    assert Om.T[0] == 25 and Om.T[-1] == 550 and len(Om.T) == 12 and np.all(np.diff(Om.T) > 0)
    assert Om.P[0] == pytest.approx(1.0) and Om.P[-1] == pytest.approx(1e6) and len(Om.P) == 24
    assert np.allclose(Om.T, (np.cos(np.pi * np.arange(11, -1, -1) / 11) + 1) * (550 - 25) / 2 + 25)
    with pytest.raises(AssertionError):
        cs.AtmosphericDomain((10, 300), 8, (1, 1e5), 8)        # below TMIN (gases.jl:51)
    with pytest.raises(AssertionError):
        cs.AtmosphericDomain((300, 200), 8, (1, 1e5), 8)


def test_bake_matches_oracle(baked):
    nu, Om, g, ref = baked
    assert g.lnsigma.shape == ref.shape == (len(nu), 8, 12)
    assert np.max(np.abs(g.lnsigma - ref)) < 1e-10            # ln(sigma): absolute = relative on sigma


def test_interpolated_cross_sections(cs, O, baked):
    nu, Om, g, ref = baked
    for T, P in [(250.0, 1e4), (Om.T[3], Om.P[5]), (180.0, 1.0), (320.0, 1.1e5), (301.3, 77.0)]:
        a = g.rawsigma(T, P)
        b = O.table_sigma(ref, Om.T, Om.P, T, P)
        assert relerr(a, b) < 1e-10
        assert g(10, T, P) == pytest.approx(400e-6 * b[10], rel=1e-10)       # Gas functor: fC * rawsigma (gases.jl:278)
    # on a grid node the interpolant returns the baked value itself
    assert relerr(g.rawsigma(Om.T[2], Om.P[7]), np.exp(ref[:, 2, 7])) < 1e-12
    # the reference's accuracy claim for such tables: ~1 % (gases.jl:7) -- check against direct line-by-line
    direct = cs.voigt(nu, g.sl, 250.0, 1e4, 400e-6 * 1e4, ctx=g.ctx)
    m = direct > 1e-4 * direct.max()
    assert relerr(g.rawsigma(250.0, 1e4)[m], direct[m]) < 0.05
    with pytest.raises(cs.ClearSkyHIPError):
        g.rawsigma(330.0, 1e4)                                                 # outside the temperature domain


def test_zero_row_scrub(cs, O, lines, ctx):
    """Wavenumbers that see no line at all become ln(floatmin) tables (gases.jl:76-79)."""
    nu = np.linspace(14080.0, 14200.0, 300)           # CO2 fixture ends at 14044 cm^-1: the upper part sees nothing
    Om = cs.AtmosphericDomain((200.0, 300.0), 4, (10.0, 1e5), 5)
    g = cs.Gas(lines("CO2"), 1e-3, nu, Om, ctx=ctx, keep_host_tables=True)
    ref = O.bake(lines("CO2"), np.full((4, 5), 1e-3), nu, Om.T, Om.P)
    assert np.max(np.abs(g.lnsigma - ref)) < 1e-10
    assert np.all(g.lnsigma[-1] == math.log(np.finfo(float).tiny))


@pytest.mark.parametrize("nlob", [2, 3])
def test_column_with_baked_and_direct_gases(cs, O, lines, baked, ctx, nlob):
    """radiate! with a baked CO2 Gas + a direct H2O gas vs the oracle fed with the numpy-interpolated cross-sections."""
    import workloads as W
    nu, Om, g, ref = baked
    h2o = cs.DirectGas(lines("H2O"), W.fC_h2o, nu)
    P = cs.pressuregrid(2.0, 1e5, 11)
    T = np.clip(W.earth_temperature(P), 185.0, 315.0)
    col = cs.Column(P, 9.8, T, 0.029, 0.0, 0.1, g, h2o, core=cs.Discretized(5, nlob), ctx=ctx)
    col.run()
    F = cs.FluxPack(len(P), len(nu))
    F.Fup[:], F.Fdn[:] = col.fetch(F.tau, F.Mup, F.Mdn)
    extra = np.array([400e-6 * O.table_sigma(ref, Om.T, Om.P, col.Tk[k], col.Pk[k]) for k in range(col.K)])
    r = O.fluxes_discretized(nu, P, 9.8, nlob, col.Tn, col.mun, col.Tlev, [h2o.sl], ["voigt"], [25.0], col.conc,
                             sigma_extra=extra, albedo=col.albedo, want_sigma=True)
    assert relerr(col.sigma_nodes(), r["sigma"], floor=1e-300) < 1e-10
    assert relerr(F.tau, r["tau"]) < 1e-10
    sm = r["Mup"].max()
    assert np.max(np.abs(F.Mup - r["Mup"])) < 1e-10 * sm and np.max(np.abs(F.Mdn - r["Mdn"])) < 1e-10 * sm
    assert np.max(np.abs(F.Fup - r["Fup"])) < 1e-10 * r["Fup"].max()
    # new temperatures on the resident column == a fresh column (tables are re-weighted, not re-baked)
    T2 = np.clip(T + 3.0, 185.0, 315.0)
    col.update(T2, 0.029)
    col.run()
    a = col.fetch()
    b = cs.fluxes(P, 9.8, T2, 0.029, 0.0, 0.1, g, h2o, core=cs.Discretized(5, nlob), ctx=ctx)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_table_domain_errors(cs, baked, ctx):
    nu, Om, g, ref = baked
    P = cs.pressuregrid(1.0, 1e5, 6)
    with pytest.raises(cs.ClearSkyHIPError):
        cs.fluxes(P, 9.8, 350.0, 0.029, 0.0, 0.0, g, ctx=ctx)                 # T outside the baked domain
    with pytest.raises(AssertionError):
        cs.fluxes(cs.pressuregrid(0.1, 1e5, 6), 9.8, 250.0, 0.029, 0.0, 0.0, g, ctx=ctx)    # checkpressures
    g2 = g.reconcentrate(800e-6)
    Tp = np.linspace(200.0, 300.0, len(P))                                     # (an isothermal column would hide the change)
    a = cs.fluxes(P, 9.8, Tp, 0.029, 0.0, 0.0, g2, ctx=ctx)
    b = cs.fluxes(P, 9.8, Tp, 0.029, 0.0, 0.0, g, ctx=ctx)
    assert a[0][0] < b[0][0]                                                   # more CO2, less OLR; same tables
    with pytest.raises(AssertionError):
        g.reconcentrate(1.5)


def test_refilled_slot_of_a_resident_column_is_refused(cs, lines, ctx):
    """a table slot baked or uploaded again (another (T, P) grid), or an accelerated-absorber slot given other knots, while a
    resident column still holds weights / knot cells formed for the old contents: cs_column_run must refuse (CS_ESTATE) instead of
    reading W [nT * nP][K] sized for the old grid; cs_column_set_tables / cs_column_set_accel make the column current again."""
    import ctypes as C
    nu = np.linspace(660.0, 680.0, 193)
    Om = cs.AtmosphericDomain((180.0, 320.0), 5, (10.0, 1e5), 6)
    g = cs.Gas(lines("CO2"), 400e-6, nu, Om, ctx=ctx, keep_host_tables=True)
    P = cs.pressuregrid(20.0, 9e4, 7)
    T = np.linspace(200.0, 290.0, 7)
    col = cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, g, core=cs.Discretized(5, 2), ctx=ctx)
    col.run()
    F0 = col.fetch()
    # the reference's own table object handed over into the SAME slot with a larger grid (cs_table_upload, B2)
    Om2 = cs.AtmosphericDomain((180.0, 320.0), 7, (10.0, 1e5), 9)
    g2 = cs.Gas(lines("CO2"), 400e-6, nu, Om2, ctx=ctx, keep_host_tables=True)
    dp = lambda a: np.ascontiguousarray(a, dtype=float).ctypes.data_as(C.POINTER(C.c_double))
    lns = np.ascontiguousarray(np.transpose(g2.lnsigma, (2, 1, 0)))       # [nP][nT][nnu]: nu fastest, then T, then P
    cs.check(cs.lib().cs_table_upload(ctx.handle, g.slot, len(nu), dp(nu), Om2.nT, dp(Om2.T), Om2.nP, dp(Om2.P), dp(lns)))
    with pytest.raises(cs.ClearSkyHIPError) as ei:
        col.run()
    assert ei.value.code == -6 and "table slot" in str(ei.value)
    slots = np.array([g.slot], dtype=np.int32)
    conc = np.full(col.K, 400e-6)
    cs.check(cs.lib().cs_column_set_tables(ctx.handle, 1, slots.ctypes.data_as(C.POINTER(C.c_int)), dp(conc)))
    col.run()
    F1 = col.fetch()
    col2 = cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, g2, core=cs.Discretized(5, 2), ctx=ctx)     # the finer table in its own slot: same numbers
    col2.run()
    F2 = col2.fetch()
    assert np.array_equal(F1[0], F2[0]) and np.array_equal(F1[1], F2[1])
    assert relerr(F1[0], F0[0]) < 0.05 and not np.array_equal(F1[0], F0[0])


def test_other_knots_under_a_resident_accelerated_column_are_refused(cs, lines):
    import ctypes as C
    c = cs.Context(0)
    try:
        nu = np.linspace(660.0, 680.0, 193)
        gas = cs.DirectGas(lines("CO2"), 400e-6, nu)
        Pk = cs.pressuregrid(20.0, 9e4, 9)
        Tk = np.linspace(200.0, 290.0, 9)
        A = cs.AcceleratedAbsorber(Tk, Pk, gas, ctx=c)
        P = cs.pressuregrid(30.0, 8e4, 6)
        T = np.linspace(205.0, 285.0, 6)
        col = cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, A, core=cs.Discretized(5, 2), ctx=c)
        col.run()
        F0 = col.fetch()
        dp = lambda a: np.ascontiguousarray(a, dtype=float).ctypes.data_as(C.POINTER(C.c_double))
        L = np.zeros((len(Pk), len(nu)))
        cs.check(cs.lib().cs_accel_fetch(c.handle, A.slot, len(nu), len(Pk), dp(L)))
        # the same knots with new values (what update! leaves): still current
        cs.check(cs.lib().cs_accel_upload(c.handle, A.slot, len(nu), dp(nu), len(Pk), dp(Pk), dp(L)))
        col.run()
        assert np.array_equal(col.fetch()[0], F0[0])
        # the same number of knots at OTHER pressures: the column's knot cells are stale
        Pk2 = Pk * 1.1
        cs.check(cs.lib().cs_accel_upload(c.handle, A.slot, len(nu), dp(nu), len(Pk2), dp(Pk2), dp(L)))
        with pytest.raises(cs.ClearSkyHIPError) as ei:
            col.run()
        assert ei.value.code == -6 and "knots" in str(ei.value)
        cs.check(cs.lib().cs_column_set_accel(c.handle, A.slot))
        col.run()
        assert not np.array_equal(col.fetch()[0], F0[0])
    finally:
        c.close()
