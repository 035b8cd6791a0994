"""GPU tests of the AbstractAbsorber operator surface and of batched columns with every member type
(row f4), against the oracle:

  * scalar access  Sigma(U, i, T, P) / U(i, T, P) / U(T, P)  (absorbers.jl:84-99) -- the sigma-chain over line-by-line gases,
    baked gases, gray gases, CIA objects and functions, with the scalar-wavenumber line-shape semantics (line_shapes.jl:12-16);
  * AcceleratedAbsorber + update! (absorbers.jl:114-207): knots, ln P interpolation, floatmin clamp, a column and a batch over it
    (what RCM does: radiative_convective.jl:95,113,154-171);
  * cs_column_batch with baked Gas objects and CIA pairs as members.
Tolerances: 1e-11 vs the oracle's line sums, 1e-10 where the numpy opacity-table restatement is involved.
"""
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


def _sigma_ref(cs, O, nu, members, T, P, cia_data=None):
    """Sigma(U, :, T, P) from the oracle: members = list of (sl, fC) line-by-line gases evaluated with the scalar-nu semantics."""
    out = np.zeros(len(nu))
    for sl, fC in members:
        Cv = fC(T, P) if callable(fC) else fC
        out += Cv * O.shape_bang("voigt", nu, sl, T, P, Cv * P, strict_ends=False)
    return out


def test_scalar_sigma_chain(cs, O, lines, ctx):
    import workloads as W
    nu = np.linspace(600.0, 760.0, 801)
    g1 = cs.DirectGas(lines("CO2"), 400e-6, nu)
    g2 = cs.DirectGas(lines("H2O"), W.fC_h2o, nu)
    gray = cs.GrayGas(3e-27, nu)
    fun = lambda v, T, P: 1e-27 * (P / 1e5) * (v / 700.0)
    U = cs.UnifiedAbsorber(g1, g2, gray, fun)
    for T, P in ((250.0, 2e4), (296.0, 101325.0), (210.0, 30.0)):
        ref = _sigma_ref(cs, O, nu, [(g1.sl, 400e-6), (g2.sl, W.fC_h2o)], T, P) + 3e-27 + fun(nu, T, P)
        for i in (0, 137, 800):
            assert cs.Sigma(U, i, T, P) == pytest.approx(ref[i], rel=1e-11)          # Σ(U, i, T, P), absorbers.jl:95
            assert U(i, T, P) == cs.Sigma(U, i, T, P)                                  # functor form :97
        assert relerr(U(T, P), ref) < 1e-11                                            # all wavenumbers :99
        def gcall(ctx=ctx): return g1(T, P, ctx=ctx)
        assert relerr(gcall(), 400e-6 * O.shape_bang("voigt", nu, g1.sl, T, P, 400e-6 * P, strict_ends=False)) < 1e-11
    assert U.update_(np.zeros(3)) is None                                              # blank update!, absorbers.jl:80
    # scalar-nu semantics: a line exactly at the cut-off distance counts (cutline is a strict >, line_shapes.jl:10), while the
    # vector method's end-point pre-filter drops it (:21)
    sl = lines("CO2")
    j = int(np.argmin(np.abs(sl.nu - 700.0)))
    v = sl.nu[j] - 25.0
    if abs((sl.nu[j] - v) - 25.0) == 0.0:
        s_scalar = cs.voigt(float(v), sl, 250.0, 1e4, 4.0, ctx=ctx)
        s_vector = cs.voigt(np.array([v]), sl, 250.0, 1e4, 4.0, ctx=ctx)[0]
        assert s_scalar == pytest.approx(float(O.shape_bang("voigt", [v], sl, 250.0, 1e4, 4.0, strict_ends=False)[0]), rel=1e-11)
        assert s_vector == pytest.approx(float(O.shape_bang("voigt", [v], sl, 250.0, 1e4, 4.0, strict_ends=True)[0]), rel=1e-11)
        assert s_scalar > s_vector


def _accel_ref(lnsig_knots, lnP_knots, P):
    """exp(phi(ln P)) with phi = LinearInterpolator(lnP, y, NoBoundaries()): (x - xa)*(yb - ya)/(xb - xa) + ya"""
    x = math.log(P)
    i = min(max(int(np.searchsorted(lnP_knots, x, side="right")) - 1, 0), len(lnP_knots) - 2)
    ya, yb = lnsig_knots[i], lnsig_knots[i + 1]
    return np.exp((x - lnP_knots[i]) * (yb - ya) / (lnP_knots[i + 1] - lnP_knots[i]) + ya)


def test_accelerated_absorber_vs_oracle(cs, O, lines):
    import workloads as W
    ctx = cs.Context(0)
    nu = np.linspace(560.0, 780.0, 5001)                        # fine enough for two interpolation levels
    nu = np.concatenate([nu, np.linspace(14100.0, 14150.0, 40)])     # + a stretch no CO2/H2O fixture line reaches... (sigma = 0)
    Pe = cs.pressuregrid(5.0, 1e5, 12)
    Te = W.earth_temperature(Pe)
    g1 = cs.DirectGas(lines("CO2"), 400e-6, nu)
    members = [(g1.sl, 400e-6)]
    A = cs.AcceleratedAbsorber(Te[::-1], Pe[::-1], g1, ctx=ctx)          # unsorted input is sorted (absorbers.jl:140-142)
    assert np.array_equal(A.P, Pe) and np.array_equal(A.T, Te) and A.nnu == len(nu)

    def knots(T):
        s = np.array([_sigma_ref(cs, O, nu, members, T[k], Pe[k]) for k in range(len(Pe))])
        with np.errstate(divide="ignore"):
            return np.maximum(np.log(s), math.log(np.finfo(float).tiny))          # absorbers.jl:185-196
    L, lnP = knots(Te), np.log(Pe)
    tiny = np.finfo(float).tiny
    for P in (Pe[0], Pe[5], Pe[-1], 37.0, 4321.0, 9.9e4, 2.0, 2e5):                # knots, between, and beyond both ends (NoBoundaries)
        a, b = A(P), _accel_ref(L, lnP, P)
        m = b > 1e3 * tiny
        assert relerr(a[m], b[m]) < 2e-11
        assert np.all(a[~m] <= 1e4 * tiny)                                         # the sigma = 0 stretch sits at floatmin
        assert A(17, P) == a[17] and cs.Sigma(A, 17, 123.0, P) == a[17]            # Σ(A, i, ·, P) ignores T (absorbers.jl:203)
    assert relerr(A(Pe[3])[-40:], np.full(40, tiny)) < 1e-12      # exp(ln floatmin)
    # update!(A, T) and update!(A, T, idx)
    T2 = Te + np.linspace(-4.0, 6.0, len(Pe))
    assert cs.update_(A, T2) is None and np.array_equal(A.T, T2)
    L2 = knots(T2)
    assert relerr(A(4321.0)[:5001], _accel_ref(L2, lnP, 4321.0)[:5001]) < 2e-11
    A.update_(251.5, 4)
    T3 = T2.copy(); T3[4] = 251.5
    assert relerr(A(Pe[4])[:5001], np.exp(knots(T3)[4])[:5001]) < 2e-11
    # a column over the accelerated absorber on finer radiative levels (RCM: radmul = 2, radiative_convective.jl:66-78)
    Pr = np.sort(np.concatenate([Pe, 0.5 * (Pe[:-1] + Pe[1:])]))
    Tr = cs.AtmosphericProfile(Pe, T3)
    L3 = knots(T3)
    for nlob in (2, 3):
        core = cs.Discretized(5, nlob)
        F = cs.radiate(Pr, 9.8, Tr, 0.029, 0.0, 0.2, A, core=core)
        col = cs.Column(Pr, 9.8, Tr, 0.029, 0.0, 0.2, A, core=core)
        extra = np.array([_accel_ref(L3, lnP, p) for p in col.Pk])
        r = O.fluxes_discretized(nu, Pr, 9.8, nlob, col.Tn, col.mun, col.Tlev, [], [], [], np.zeros((0, col.K)), sigma_extra=extra,
                                 albedo=col.albedo)
        assert relerr(F.tau, r["tau"]) < 2e-11
        sm = r["Mup"].max()
        assert np.max(np.abs(F.Mup - r["Mup"])) < 1e-11 * sm and np.max(np.abs(F.Fup - r["Fup"])) < 1e-11 * r["Fup"].max()
    # jacobian!: the np+1 perturbed profiles as ONE batch; the cross-sections stay those of the last update! (the reference does
    # not update 𝒜 between the perturbed radiate! calls, radiative_convective.jl:154-171)
    col = cs.Column(Pr, 9.8, Tr, 0.029, 0.0, 0.0, A, core=cs.Discretized(5, 2), want_tau=False, want_M=False)
    Tlev = np.array([Tr(p) for p in Pr])
    Ts = [Tlev] + [Tlev + 1.0 * (np.arange(len(Pr)) == i) for i in (0, 7, len(Pr) - 1)]
    Bu, Bd = col.run_batch(Ts)
    extra = np.array([_accel_ref(L3, lnP, p) for p in col.Pk])
    for b, Tb in enumerate(Ts):
        fT = cs.formprofile(Pr, Tb)
        Tn, mun = cs.lobattoevaluations(Pr, fT, cs.formprofile(Pr, 0.029), 2)
        r = O.fluxes_discretized(nu, Pr, 9.8, 2, Tn, mun, np.array([fT(p) for p in Pr]), [], [], [], np.zeros((0, col.K)), sigma_extra=extra)
        assert np.max(np.abs(Bu[b] - r["Fup"])) < 1e-11 * r["Fup"].max() and np.max(np.abs(Bd[b] - r["Fdn"])) < 1e-11 * r["Fup"].max()
    assert np.max(np.abs(Bu[1] - Bu[0])) > 0
    # an AcceleratedAbsorber stands for all absorbers of a column
    with pytest.raises(TypeError):                              # (the reference throws its "absorbers must only be ..." string)
        cs.Column(Pr, 9.8, Tr, 0.029, 0.0, 0.0, A, g1)
    assert cs.temperaturelimits(A) == (0.0, float("inf"))
    # AcceleratedAbsorber(T, P, A) hands A itself back (absorbers.jl:161-164): one object, one owner of the device slot -- dropping
    # the alias must leave A usable, and dropping an absorber whose column is resident must not leave that column half-alive
    import gc
    before = A(4321.0).copy()
    B = cs.AcceleratedAbsorber(T3, Pe, A)
    assert B is A
    del B
    gc.collect()
    assert np.array_equal(A(4321.0), before)
    F1 = cs.radiate(Pr, 9.8, Tr, 0.029, 0.0, 0.2, A, core=cs.Discretized(5, 2))
    with pytest.raises(AssertionError):
        cs.AcceleratedAbsorber(T3, Pe * 1.01, A)                # "cannot change ... pressure coordinates after construction"
    A2 = cs.AcceleratedAbsorber(Te, Pe, g1, ctx=ctx)            # a second absorber on the same context, then gone again
    col2 = cs.Column(Pr, 9.8, Tr, 0.029, 0.0, 0.2, A2, core=cs.Discretized(5, 2))
    col2.run()
    del A2, col2
    gc.collect()
    F2 = cs.radiate(Pr, 9.8, Tr, 0.029, 0.0, 0.2, A, core=cs.Discretized(5, 2))
    assert np.array_equal(F1.Fup, F2.Fup)
    ctx.close()


def test_batch_with_baked_gas_and_cia_vs_oracle(cs, O, lines):
    """cs_column_batch with every member type the reference's RCM can hold: a baked CO2 Gas, a line-by-line CH4 gas, the CO2-CO2
    and CO2-CH4 CIA pairs and a gray term; B = 4 thermal states against the oracle (tables: numpy restatement, 1e-10)."""
    import workloads as W
    ctx = cs.Context(0)
    nu = np.linspace(1200.0, 1420.0, 1400)
    Om = cs.AtmosphericDomain((170.0, 330.0), 8, (1.0, 1.2e5), 12)
    co2 = cs.Gas(lines("CO2"), 0.9, nu, Om, ctx=ctx, keep_host_tables=True)
    ch4 = cs.DirectGas(lines("CH4"), lambda T, P: 0.05 * (1.0 + 1e-3 * (T - 250.0)), nu)
    x1, x2 = cs.CIATables(W.fixture("CO2-CO2_2018.cia")), cs.CIATables(W.fixture("CO2-CH4_2018.cia"), extrapolate=True)
    P = cs.pressuregrid(3.0, 1e5, 11)
    T0 = np.clip(W.earth_temperature(P), 180.0, 320.0)
    col = cs.Column(P, 9.8, T0, 0.044, 0.0, 0.0, co2, ch4, x1, x2, cs.GrayGas(1e-28, nu), core=cs.Discretized(5, 3), want_tau=False,
                    want_M=False, ctx=ctx)
    ref_tab = O.bake(lines("CO2"), np.full((8, 12), 0.9), nu, Om.T, Om.P)
    d1, d2 = cs.readcia(W.fixture("CO2-CO2_2018.cia")), cs.readcia(W.fixture("CO2-CH4_2018.cia"))

    def oracle(T):
        fT = cs.formprofile(P, T)
        Tn, mun = cs.lobattoevaluations(P, fT, cs.formprofile(P, 0.044), 3)
        Tk, Pk = cs.nodevalues(Tn, 3), cs.nodepressures(P, 3)
        cch4 = np.array([[ch4.fC(Tk[k], Pk[k]) for k in range(len(Pk))]])
        extra = np.zeros((len(Pk), len(nu)))
        for k in range(len(Pk)):
            extra[k] = 0.9 * O.table_sigma(ref_tab, Om.T, Om.P, Tk[k], Pk[k])
            extra[k] += O.cia_sigma(d1, nu, Tk[k], Pk[k], 0.9 * Pk[k], 0.9 * Pk[k])
            extra[k] += O.cia_sigma(d2, nu, Tk[k], Pk[k], 0.9 * Pk[k], cch4[0, k] * Pk[k], extrapolate=True)
        return O.fluxes_discretized(nu, P, 9.8, 3, Tn, mun, np.array([fT(p) for p in P]), [ch4.sl], ["voigt"], [25.0], cch4,
                                    sigma_gray=1e-28, sigma_extra=extra)
    Ts = [T0, T0 + 2.0, np.clip(T0 - np.linspace(0.0, 8.0, len(P)), 180.0, 320.0), T0 + 1.5 * (np.arange(len(P)) == 5)]
    Bu, Bd = col.run_batch(Ts)
    for b, Tb in enumerate(Ts):
        r = oracle(Tb)
        assert np.max(np.abs(Bu[b] - r["Fup"])) < 1e-10 * r["Fup"].max() and np.max(np.abs(Bd[b] - r["Fdn"])) < 1e-10 * r["Fup"].max()
    # the resident column itself is untouched by the batch
    col.run()
    a = col.fetch()
    assert np.max(np.abs(a[0] - Bu[0])) < 1e-13 * a[0].max()
    with pytest.raises(cs.ClearSkyHIPError):
        col.run_batch([np.full(len(P), 400.0)])                # outside the baked temperature domain
    # domain helpers (absorbers.jl:237-270)
    U = col.U
    assert cs.pressurelimits(U.gas) == (1.0, 1.2e5) and cs.temperaturelimits(U) == (170.0, 330.0)
    cs.checkpressures(U, 1e5, 3.0)
    with pytest.raises(AssertionError):
        cs.checkpressures(U, 2e5, 3.0)
    with pytest.raises(AssertionError):
        cs.checkpressures(U, 3.0, 1e5)
    ctx.close()
