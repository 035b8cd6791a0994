"""GPU parity tests: the HIP path, called through the C ABI (ctypes host mirror), against the CPU oracle on the same
inputs and against the committed goldens.

Tolerances (fp64 everywhere; north star: 1e-6 relative):
  * cross-sections / optical depths vs the oracle: 1e-11 (same algorithm; differences = FMA contraction, device libm,
    v_rcp+Newton instead of IEEE division)
  * vs the wofz-based goldens: 1e-10
  * monochromatic fluxes: absolute, 1e-11 of the column maximum (values near the top of the atmosphere are pure
    rounding of the linear-in-tau source function)
"""
import math
import os

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu

STATES = [(220.0, 50.0, 0.02), (296.0, 101325.0, 40.53), (260.0, 3e3, 30.0)]


@pytest.fixture(scope="module")
def ctx(cs):
    c = cs.Context(0)
    yield c
    c.close()


def test_faddeeva_device(cs, O, golden, ctx):
    g = golden("faddeeva")
    w = cs.faddeeva(g["x"], g["y"], ctx)
    m = g["w"] > 1e-290
    assert relerr(w[m], g["w"][m]) < 2e-13
    rng = np.random.default_rng(3)
    n = 500000
    x = np.concatenate([rng.uniform(0, 12, n), 10 ** rng.uniform(0, 7.5, n), -rng.uniform(0, 30, 1000)])
    y = np.concatenate([10 ** rng.uniform(-10, 1.2, n), 10 ** rng.uniform(-6, 4, n), rng.uniform(1e-4, 3, 1000)])
    assert relerr(cs.faddeeva(x, y, ctx), O.faddeeva(x, y)) < 1e-12


@pytest.mark.parametrize("gas", ["CO2", "H2O", "CH4"])
@pytest.mark.parametrize("shape", ["voigt", "lorentz", "doppler", "PHCO2"])
def test_shape_batch_vs_oracle(cs, O, lines, ctx, gas, shape):
    sl = lines(gas)
    nu = np.linspace(1.0, 2500.0, 5003)     # not a multiple of the 256-point tile
    cut = 500.0 if shape == "PHCO2" else 25.0
    T, P, Pp = map(list, zip(*STATES))
    sg = cs.shape_batch(sl, shape, nu, T, P, Pp, cut, ctx)
    tol = 5e-12 if shape != "doppler" else 2e-11   # exp(-x^2) amplifies 1-ulp differences in x^2
    for k in range(len(T)):
        so = O.shape_bang(shape, nu, sl, T[k], P[k], Pp[k], cut)
        assert np.array_equal(sg[k] == 0, so == 0)
        assert relerr(sg[k], so, floor=1e-250) < tol


def test_shapes_vs_golden(cs, golden, lines, ctx):
    g = golden("lineshapes")
    T, P, Pp = map(list, zip(*g["states"]))
    for key, shape, gas, nukey, cut in (("voigt_co2", "voigt", "CO2", "nu_co2", 25.0), ("voigt_h2o", "voigt", "H2O", "nu_h2o", 25.0),
                                        ("lorentz_co2", "lorentz", "CO2", "nu_co2", 25.0), ("doppler_co2", "doppler", "CO2", "nu_co2", 25.0),
                                        ("phco2_co2", "PHCO2", "CO2", "nu_co2", 500.0)):
        s = cs.shape_batch(lines(gas), shape, g[nukey], T, P, Pp, cut, ctx)
        assert relerr(s, g[key], floor=1e-12 * g[key].max()) < 1e-10, key
    s = cs.voigt(g["nu_edge"], lines("CO2"), *g["states"][1], 25.0, ctx=ctx)
    assert relerr(s, g["voigt_edge"]) < 1e-10        # strict end-point pre-filter + inclusive cut-off


def test_inplace_and_scalar_methods(cs, O, lines, ctx):
    sl = lines("CO2")
    nu = np.linspace(660.0, 675.0, 300)
    sig = np.full(300, 7.0)
    assert cs.voigt_(sig, nu, sl, 250.0, 1e4, 4.0, ctx=ctx) is None        # voigt! returns nothing, overwrites sigma
    assert relerr(sig, O.shape_bang("voigt", nu, sl, 250.0, 1e4, 4.0)) < 5e-12
    assert cs.voigt(667.0, sl, 250.0, 1e4, 4.0, ctx=ctx) == pytest.approx(1.5591e-21, rel=1e-4)    # SURVEY.md 4 anchor
    assert cs.voigt(667.5, sl, 296.0, 101325.0, 40.53, ctx=ctx) == pytest.approx(1.1332e-19, rel=1e-4)
    for f_, f, name in ((cs.lorentz_, cs.lorentz, "lorentz"), (cs.doppler_, cs.doppler, "doppler"), (cs.PHCO2_, cs.PHCO2, "PHCO2")):
        f_(sig, nu, sl, 250.0, 1e4, 4.0, ctx=ctx)
        assert np.array_equal(sig, f(nu, sl, 250.0, 1e4, 4.0, ctx=ctx))


def test_edge_cases(cs, O, lines, ctx):
    sl = lines("CO2")
    # a window with no line at all -> exactly zero (sigma overwritten, not accumulated)
    gap = np.linspace(14100.0, 14200.0, 700)
    assert np.all(cs.voigt(gap, sl, 250.0, 1e4, 4.0, ctx=ctx) == 0.0)
    # one wavenumber, exactly on a line centre (x = 0, small y: the hardest Faddeeva region)
    j = int(np.argmin(np.abs(sl.nu - 667.4)))
    v = np.array([sl.nu[j]])
    assert relerr(cs.voigt(v, sl, 200.0, 1.0, 0.0, ctx=ctx), O.shape_bang("voigt", v, sl, 200.0, 1.0, 0.0)) < 1e-12
    # ragged: 257 points = one full tile + 1
    nu = np.linspace(2300.0, 2400.0, 257)
    assert relerr(cs.voigt(nu, sl, 300.0, 5e4, 20.0, ctx=ctx), O.shape_bang("voigt", nu, sl, 300.0, 5e4, 20.0)) < 5e-12
    # zero pressure: pure Doppler core through the Voigt path (y = 0)
    a, b = cs.voigt(nu, sl, 300.0, 0.0, 0.0, ctx=ctx), O.shape_bang("voigt", nu, sl, 300.0, 0.0, 0.0)
    assert relerr(a, b, floor=1e-300) < 1e-11
    # temperature range ends are legal, outside is an error (line_shapes.jl:29)
    for T in (25.0, 1000.0):
        assert relerr(cs.voigt(nu, sl, T, 1e4, 4.0, ctx=ctx), O.shape_bang("voigt", nu, sl, T, 1e4, 4.0), floor=1e-300) < 5e-11
    with pytest.raises(cs.ClearSkyHIPError) as e:
        cs.voigt(nu, sl, 24.9, 1e4, 4.0, ctx=ctx)
    assert e.value.code == -2
    with pytest.raises(cs.ClearSkyHIPError) as e:
        cs.voigt(nu[::-1].copy(), sl, 250.0, 1e4, 4.0, ctx=ctx)
    assert e.value.code == -4
    # isotopologue without a Qref/Q fit -> error, not a silent (Tref/T)^1.5 fallback (line_shapes.jl:115-120)
    par = dict(M=np.array([34], np.int16), I=np.array([1], np.int16), nu=np.array([68.7]), S=np.array([1e-22]),
               gamma_a=np.array([0.05]), gamma_s=np.array([0.05]), Epp=np.array([0.0]), na=np.array([0.7]))
    with pytest.raises(cs.ClearSkyHIPError) as e:
        cs.voigt(np.linspace(60, 80, 10), cs.SpectralLines(par), 250.0, 1e4, 0.0, ctx=ctx)
    assert e.value.code == -3


def _column_vs(cs, r, F, scale_M=None, tol=1e-11):
    assert relerr(F.tau, r["tau"]) < tol
    sm = np.max(r["Mup"])
    assert np.max(np.abs(F.Mup - r["Mup"])) < tol * sm and np.max(np.abs(F.Mdn - r["Mdn"])) < tol * sm
    assert np.max(np.abs(F.Fup - r["Fup"])) < tol * np.max(r["Fup"])
    assert np.max(np.abs(F.Fdn - r["Fdn"])) < tol * max(np.max(r["Fdn"]), np.max(r["Fup"]))
    assert np.array_equal(F.Fnet, F.Fup - F.Fdn)


def test_column_gray_vs_golden(cs, golden, ctx):
    """Config 1 plumbing: GrayGas through the whole flux path (20 layers, analytic kappa)."""
    g = golden("column_gray")
    gas = cs.GrayGas(float(g["sigma"]), g["nu"])
    F = cs.radiate(g["P"], float(g["g"]), g["T"], float(g["mu"]), 0.0, 0.0, gas, core=cs.Discretized(5, 2), ctx=ctx)
    _column_vs(cs, dict(tau=g["tau"], Mup=g["Mup"], Mdn=g["Mdn"], Fup=g["Fup"], Fdn=g["Fdn"]), F, tol=1e-11)


@pytest.mark.parametrize("name", ["column_co2", "column_co2_lob4"])
def test_column_co2_vs_golden(cs, golden, lines, ctx, name):
    g = golden(name)
    gas = cs.DirectGas(lines("CO2"), float(g["conc"]), g["nu"])
    core = cs.Discretized(int(g["nstream"]), int(g["nlobatto"]))
    F = cs.radiate(g["P"], float(g["g"]), g["T"], float(g["mu"]), float(g["fS"]), float(g["fa"]), gas, core=core, ctx=ctx)
    _column_vs(cs, dict(tau=g["tau"], Mup=g["Mup"], Mdn=g["Mdn"], Fup=g["Fup"], Fdn=g["Fdn"]), F, tol=1e-10)


@pytest.mark.parametrize("nlob,ns", [(2, 5), (3, 4), (5, 8)])
def test_column_two_gases_vs_oracle(cs, O, lines, ctx, nlob, ns):
    """H2O + CO2 on the reference's fixtures with a function absorber, stellar beam and albedo: every output of
    monochromaticfluxes!/radiate! (tau, M+, M-, F+, F-) against the oracle."""
    import workloads as W
    nu = np.linspace(1.0, 2500.0, 3001)
    P = cs.pressuregrid(1.0, 1e5, 13)
    T = W.earth_temperature(P)
    g1 = cs.DirectGas(lines("H2O"), W.fC_h2o, nu)
    g2 = cs.DirectGas(lines("CO2"), 400e-6, nu)
    cont = lambda v, T_, P_: 1e-27 * (P_ / 1e5) * np.ones_like(v)           # sigma(nu,T,P) function, absorbers.jl:24
    fS = lambda v: 1e-3 * math.exp(-((v - 2000.0) / 300.0) ** 2)
    col = cs.Column(P, 9.8, T, 0.029, fS, 0.25, g1, g2, cont, core=cs.Discretized(ns, nlob), ctx=ctx)
    col.run()
    F = cs.FluxPack(len(P), len(nu))
    F.Fup[:], F.Fdn[:] = col.fetch(F.tau, F.Mup, F.Mdn)
    F.Fnet[:] = F.Fup - F.Fdn
    r = O.fluxes_discretized(nu, P, 9.8, nlob, col.Tn, col.mun, col.Tlev, [g1.sl, g2.sl], ["voigt"] * 2, [25.0] * 2,
                             col.conc, sigma_extra=col.sigma_extra, S_toa=col.S_toa, albedo=col.albedo, nstream=ns,
                             want_sigma=True)
    assert relerr(col.sigma_nodes(), r["sigma"], floor=1e-300) < 1e-11
    _column_vs(cs, r, F)
    # the in-place API gives the same arrays (monochromaticfluxes!, fluxes.jl:238-279)
    Mup, Mdn, tau = (np.zeros(s, order="F") for s in ((len(P), len(nu)), (len(P), len(nu)), (len(P) - 1, len(nu))))
    assert cs.monochromaticfluxes_(Mup, Mdn, tau, cs.Discretized(ns, nlob), P, 9.8, T, 0.029, fS, 0.25, g1, g2, cont, ctx=ctx) is None
    assert np.array_equal(Mup, F.Mup) and np.array_equal(Mdn, F.Mdn) and np.array_equal(tau, F.tau)
    Fu, Fd = cs.fluxes(P, 9.8, T, 0.029, fS, 0.25, g1, g2, cont, core=cs.Discretized(ns, nlob), ctx=ctx)
    assert np.array_equal(Fu, F.Fup) and np.array_equal(Fd, F.Fdn)


def test_errors_of_the_flux_path(cs, lines, ctx):
    nu = np.linspace(600.0, 700.0, 64)
    gas = cs.DirectGas(lines("CO2"), 400e-6, nu)
    P = cs.pressuregrid(1.0, 1e5, 5)
    with pytest.raises(AssertionError):                                  # fluxes.jl:257
        cs.fluxes(P[::-1].copy(), 9.8, 250.0, 0.029, 0.0, 0.0, gas, ctx=ctx)
    with pytest.raises(AssertionError):                                  # fluxes.jl:5
        cs.fluxes(P, 9.8, 250.0, 0.029, 0.0, 0.0, gas, theta_s=1.6, ctx=ctx)
    with pytest.raises(cs.ClearSkyHIPError):                             # T outside [25,1000]
        cs.fluxes(P, 9.8, 1200.0, 0.029, 0.0, 0.0, gas, ctx=ctx)
    with pytest.raises(AssertionError):                                  # FluxPack size check, fluxes.jl:375
        cs.radiate_(cs.FluxPack(4, 64), cs.Discretized(), P, 9.8, 250.0, 0.029, 0.0, 0.0, gas, ctx=ctx)


def test_opticaldepth_and_transmittance(cs, O, lines, ctx):
    nu = np.linspace(640.0, 700.0, 500)
    gas = cs.DirectGas(lines("CO2"), 400e-6, nu)
    P = cs.pressuregrid(10.0, 1e5, 9)
    tau = cs.opticaldepth(P, 9.8, 260.0, 0.029, 0.3, gas, nlobatto=4, ctx=ctx)
    xs, ws = cs.lobattonodes(4)
    Pk = cs.nodepressures(P, 4)
    ref = np.zeros(len(nu))
    sig = np.array([400e-6 * O.shape_bang("voigt", nu, lines("CO2"), 260.0, p, 400e-6 * p, strict_ends=False) for p in Pk])
    beta = 1e-4 * 6.02214076e23 / 9.8 * sig / 0.029
    for i in range(len(P) - 1):
        ref += sum((P[i + 1] - P[i]) * ws[n] * beta[i * 3 + n] for n in range(4)) / math.cos(0.3)
    assert relerr(tau, ref) < 1e-11
    assert np.array_equal(cs.transmittance(P, 9.8, 260.0, 0.029, 0.3, gas, nlobatto=4, ctx=ctx), np.exp(-tau))


def test_c2_full_parity_vs_oracle(cs, O, ctx):
    """BASELINE configs[1] at full size: CO2 fixture, 1e4 wavenumbers x 40 layers -- every output against the oracle."""
    import workloads as W
    cfg = W.config("C2")
    F = cs.radiate(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], cfg["fS"], cfg["fa"], *cfg["absorbers"], core=cfg["core"], ctx=ctx)
    col = cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], cfg["fS"], cfg["fa"], *cfg["absorbers"], core=cfg["core"],
                    want_tau=False, want_M=False, ctx=ctx)
    r = O.fluxes_discretized(cfg["nu"], cfg["P"], cfg["g"], 2, col.Tn, col.mun, col.Tlev, [g.sl for g in col.gases], ["voigt"],
                             [25.0], col.conc)
    _column_vs(cs, r, F)
    assert abs(F.Fup[0] - r["Fup"][0]) < 1e-9          # OLR error [W/m^2]


# ---- full BASELINE sizes: size-independent properties + sparse direct parity ---------------------------------------

@pytest.fixture(scope="module")
def c3(cs, ctx):
    import workloads as W
    cfg = W.config("C3")
    col = cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], cfg["fS"], cfg["fa"], *cfg["absorbers"], core=cfg["core"],
                    theta_s=cfg["theta_s"], ctx=ctx)
    col.run()
    F = cs.FluxPack(col.np, col.nnu)
    F.Fup[:], F.Fdn[:] = col.fetch(F.tau, F.Mup, F.Mdn)
    return cfg, col, F


def test_c3_sparse_parity_vs_oracle(cs, O, c3):
    """1e5 x 60 x ~1e5 lines: 96 random wavenumber columns recomputed by the oracle (it handles any sorted nu subset)."""
    cfg, col, F = c3
    rng = np.random.default_rng(11)
    idx = np.sort(rng.choice(col.nnu, 96, replace=False))
    r = O.fluxes_discretized(cfg["nu"][idx], cfg["P"], cfg["g"], 2, col.Tn, col.mun, col.Tlev, [g.sl for g in col.gases],
                             ["voigt"] * 2, [25.0] * 2, col.conc)
    assert relerr(F.tau[:, idx], r["tau"]) < 1e-11
    sm = r["Mup"].max()
    assert np.max(np.abs(F.Mup[:, idx] - r["Mup"])) < 1e-11 * sm and np.max(np.abs(F.Mdn[:, idx] - r["Mdn"])) < 1e-11 * sm


def test_c3_full_grid_every_element_vs_oracle(cs, O, c3):
    """BASELINE configs[2] at FULL size, every element of what radiate! returns (fluxes.jl:357-383) and of the node cross-sections:
    the oracle evaluates the whole 1e5-point column (a few seconds on the GPU box's host cores).  An error confined to one interval
    or tile of the full grid cannot hide in a sample of columns or in the band integral.  Tolerance 1e-11 (sigma, tau relative; M+,
    M- over the column maximum; F over max F+) -- observed ~1e-13."""
    cfg, col, F = c3
    r = O.fluxes_discretized(cfg["nu"], cfg["P"], cfg["g"], 2, col.Tn, col.mun, col.Tlev, [g.sl for g in col.gases],
                             ["voigt"] * 2, [25.0] * 2, col.conc, want_sigma=True)
    col.run()
    sig = col.sigma_nodes()
    pos = r["sigma"] > 0
    assert np.all(sig[~pos] == 0.0)
    assert float(np.max(np.abs(sig[pos] - r["sigma"][pos]) / r["sigma"][pos])) < 1e-11
    assert relerr(F.tau, r["tau"]) < 1e-11
    sm = max(r["Mup"].max(), r["Mdn"].max())
    assert np.max(np.abs(F.Mup - r["Mup"])) < 1e-11 * sm and np.max(np.abs(F.Mdn - r["Mdn"])) < 1e-11 * sm
    fm = r["Fup"].max()
    assert np.max(np.abs(F.Fup - r["Fup"])) < 1e-11 * fm and np.max(np.abs(F.Fdn - r["Fdn"])) < 1e-11 * fm
    assert abs(F.Fup[0] - r["Fup"][0]) < 1e-9          # OLR error [W/m^2]


def test_c3_band_integral_and_boundaries(cs, c3):
    cfg, col, F = c3
    w = cs.trapz_weights(cfg["nu"])
    assert relerr(F.Fup, F.Mup @ w) < 1e-12 and relerr(F.Fdn, F.Mdn @ w, floor=1e-9) < 1e-12   # intF! on device
    assert np.all(F.Mdn[0] == 0.0)                                              # no stellar beam: nothing enters at the top
    Bs = cs.planck(cfg["nu"], col.Tlev[-1])
    assert relerr(F.Mup[-1], math.pi * Bs) < 1e-13                              # black surface, zero albedo
    assert np.all(F.tau >= 1e-6) and np.all(np.isfinite(F.tau))                 # floor of dDepth!
    assert 50.0 < F.Fup[0] < 400.0                                              # an Earth-like OLR [W/m^2]


def test_c3_deterministic_and_shards_add_up(cs, c3, ctx):
    cfg, col, F = c3
    col.run()
    F2 = col.fetch()
    assert np.array_equal(F2[0], F.Fup) and np.array_equal(F2[1], F.Fdn)        # fixed-order reductions: bitwise repeatable
    import workloads as W
    tot = np.zeros(2 * col.np)
    for r in W.balanced_ranges(cfg["nu"], cfg["absorbers"], 4):
        sh = cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], cfg["fS"], cfg["fa"], *cfg["absorbers"], core=cfg["core"],
                       theta_s=cfg["theta_s"], want_tau=False, want_M=False, nu_range=r, ctx=ctx)
        sh.run()
        tot += np.concatenate(sh.fetch())
    assert relerr(tot[: col.np], F.Fup) < 1e-13 and relerr(tot[col.np:], F.Fdn, floor=1e-9) < 1e-13


def test_update_state_matches_fresh_setup(cs, lines, ctx):
    """RCM inner loop: new temperatures on a resident column == building the column anew."""
    nu = np.linspace(550.0, 800.0, 2000)
    P = cs.pressuregrid(1.0, 1e5, 21)
    gas = cs.DirectGas(lines("CO2"), 400e-6, nu)
    T1 = np.linspace(210.0, 290.0, 21)
    T2 = T1 + np.linspace(-5.0, 3.0, 21)
    col = cs.Column(P, 9.8, T1, 0.029, 0.0, 0.0, gas, ctx=ctx)
    col.run()
    col.update(T2, 0.029)
    col.run()
    a = col.fetch()
    b = cs.fluxes(P, 9.8, T2, 0.029, 0.0, 0.0, gas, ctx=ctx)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_c5_reduced_vs_oracle(cs, O):
    """BASELINE configs[4] physics at reduced size (4 gases incl. synthetic O3 + both CIA pairs, 100 layers, 3000 wavenumbers):
    fp64 outputs vs the oracle fed with the numpy CIA restatement."""
    import workloads as W
    ctx = cs.Context(0)
    cfg = W.config("C5", nnu=3000)
    col = cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], 0.0, 0.0, *cfg["absorbers"], core=cfg["core"], ctx=ctx)
    col.run()
    F = cs.FluxPack(col.np, col.nnu)
    F.Fup[:], F.Fdn[:] = col.fetch(F.tau, F.Mup, F.Mdn)
    F.Fnet[:] = F.Fup - F.Fdn
    assert len(col.gases) == 4 and len(col.U.cia) == 2 and col.K == 101
    d = [cs.readcia(W.fixture(f)) for f in ("CO2-CO2_2018.cia", "CO2-CH4_2018.cia")]
    extra = np.zeros((col.K, col.nnu))
    for k in range(col.K):
        for ci, x in enumerate(col.U.cia):
            extra[k] += O.cia_sigma(d[ci], cfg["nu"], col.Tk[k], col.Pk[k], col.cia_P1[ci, k], col.cia_P2[ci, k])
    r = O.fluxes_discretized(cfg["nu"], cfg["P"], cfg["g"], 2, col.Tn, col.mun, col.Tlev, [g.sl for g in col.gases], ["voigt"] * 4,
                             [25.0] * 4, col.conc, sigma_extra=extra)
    _column_vs(cs, r, F)
    ctx.close()


def test_mixed_precision_variant(cs, O, lines):
    """BASELINE configs[4]: fp32 far wings.  Cross-sections within 1e-6 of the fp64 path and of the oracle (north-star
    tolerance), OLR within 1e-5 W/m^2; widening the fp64 region (far_s) tightens the agreement; fp64 mode is untouched."""
    import workloads as W
    ctx = cs.Context(0)
    cfg = W.config("C2", nnu=4000, nl=20)
    def run():
        col = cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], 0.0, 0.0, *cfg["absorbers"], core=cfg["core"], ctx=ctx)
        col.run()
        F = col.fetch()
        return col, col.sigma_nodes(), F
    col, s64, F64 = run()
    ctx.set_precision("mixed", 1e6)
    _, s32, F32 = run()
    ctx.set_precision("mixed", 1e8)
    _, s32b, F32b = run()
    ctx.set_precision("fp64")
    _, s64b, F64b = run()
    assert np.array_equal(s64, s64b) and np.array_equal(F64[0], F64b[0])
    m = s64 > 0
    e1, e2 = np.max(np.abs(s32[m] / s64[m] - 1)), np.max(np.abs(s32b[m] / s64[m] - 1))
    assert 0 < e1 < 1e-6 and e2 <= e1
    assert abs(F32[0][0] - F64[0][0]) < 1e-5 and abs(F32b[0][0] - F64[0][0]) <= abs(F32[0][0] - F64[0][0]) + 1e-9
    r = O.fluxes_discretized(cfg["nu"], cfg["P"], cfg["g"], 2, col.Tn, col.mun, col.Tlev, [g.sl for g in col.gases], ["voigt"],
                             [25.0], col.conc, want_sigma=True)
    assert np.max(np.abs(s32[m] / r["sigma"][m] - 1)) < 1e-6
    with pytest.raises(cs.ClearSkyHIPError):
        ctx.set_precision("mixed", 1e3)
    ctx.close()


def test_batched_columns_match_sequential(cs, lines):
    """cs_column_batch (the np+1 perturbed profiles of jacobian!, radiative_convective.jl:154-171): one device batch == the same
    profiles evaluated one after the other on the resident column (summation grouping may differ: 1e-13)."""
    import workloads as W
    ctx = cs.Context(0)
    nu = np.linspace(400.0, 1100.0, 3000)
    P = cs.pressuregrid(5.0, 1e5, 16)
    T0 = W.earth_temperature(P)
    g1 = cs.DirectGas(lines("CO2"), 400e-6, nu)
    g2 = cs.DirectGas(lines("H2O"), W.fC_h2o, nu)
    col = cs.Column(P, 9.8, T0, 0.029, 0.0, 0.0, g1, g2, cs.GrayGas(1e-27, nu), core=cs.Discretized(5, 3), want_tau=False, want_M=False,
                    ctx=ctx)
    Ts = [T0] + [T0 + 0.5 * (np.arange(len(P)) == i) for i in range(len(P))]      # jacobian!: one level perturbed at a time
    Fu, Fd = col.run_batch(Ts, 0.029)
    assert Fu.shape == (len(P) + 1, len(P))
    for b in (0, 1, 7, len(P)):
        col.update(Ts[b], 0.029)
        col.run()
        a = col.fetch()
        assert np.max(np.abs(Fu[b] - a[0])) < 1e-13 * a[0].max() and np.max(np.abs(Fd[b] - a[1])) < 1e-13 * a[0].max()
    J = (Fu[1:] - Fd[1:] - (Fu[0] - Fd[0])) / 0.5                                   # dFnet/dT: finite and not all zero
    assert np.all(np.isfinite(J)) and np.abs(J).max() > 0
    with pytest.raises(cs.ClearSkyHIPError):
        col.run_batch([np.full(len(P), 2000.0)])                                    # T outside [25, 1000]
    ctx.close()


def test_dense_table_large_grid_sparse_parity(cs, O):
    """4e5 lines x 2e5 wavenumbers x 31 levels (2e11 line evaluations, windows of ~16 000 lines, the near-line queue spills into
    its fallback): 48 random columns against the oracle, and bitwise repeatability."""
    ctx = cs.Context(0)
    sl = cs.SpectralLines.synthetic(2, 400_000, 4242, numin=200.0, numax=1450.0)
    nu = np.linspace(300.0, 1300.0, 200_000)
    P = cs.pressuregrid(1.0, 1e5, 31)
    T = np.linspace(210.0, 295.0, 31)
    gas = cs.DirectGas(sl, 3e-4, nu)
    col = cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, gas, want_M=False, ctx=ctx)
    col.run()
    tau = np.zeros((col.nl, col.nnu), order="F")
    F1 = col.fetch(tau)
    col.run()
    F2 = col.fetch()
    assert np.array_equal(F1[0], F2[0]) and np.all(np.isfinite(tau))
    idx = np.sort(np.random.default_rng(5).choice(col.nnu, 48, replace=False))
    r = O.fluxes_discretized(nu[idx], P, 9.8, 2, col.Tn, col.mun, col.Tlev, [sl], ["voigt"], [25.0], col.conc)
    assert relerr(tau[:, idx], r["tau"]) < 1e-11
    ctx.close()
