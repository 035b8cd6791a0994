"""CPU tests: the oracle (oracle/cs_oracle.c) against the committed golden vectors and analytic identities.
Tolerances: the goldens use scipy.special.wofz (~1e-13 rel) and numpy's reassociated sums, so 1e-10 on cross-sections
and 1e-9 on fluxes separates "same formulas" from "different formulas" by many orders of magnitude."""
import math

import numpy as np
import pytest
from scipy.integrate import quad

from conftest import relerr


def test_faddeeva_known_answers(O, golden):
    g = golden("faddeeva")
    w = O.faddeeva(g["x"], g["y"])
    m = g["w"] > 1e-290
    assert relerr(w[m], g["w"][m]) < 2e-13          # 40-digit mpmath values
    assert np.all(np.abs(w[~m]) < 1e-280)


def test_faddeeva_vs_wofz_dense(O):
    from scipy.special import wofz
    rng = np.random.default_rng(1)
    n = 100000
    x = np.concatenate([rng.uniform(0, 12, n), 10 ** rng.uniform(0, 7.5, n), -rng.uniform(0, 50, 100)])
    y = np.concatenate([10 ** rng.uniform(-10, 1.2, n), 10 ** rng.uniform(-6, 4, n), rng.uniform(1e-3, 2, 100)])
    assert relerr(O.faddeeva(x, y), wofz(x + 1j * y).real) < 5e-13   # wofz itself is ~1e-13


def test_chebyQrefQ_anchor(O, lines):
    sl = lines("CO2")
    a = sl.cheb[0, : sl.ncheb[0]]
    assert O.chebyQrefQ(296.0, a) == pytest.approx(0.9987408464004868, rel=1e-14)   # SURVEY.md 4 anchors
    assert O.chebyQrefQ(250.0, a) == pytest.approx(1.2273351054954134, rel=1e-14)
    with pytest.raises(AssertionError):
        O.chebyQrefQ(24.0, a)
    with pytest.raises(AssertionError):
        O.chebyQrefQ(1000.5, a)


def test_planck_anchors_and_stefan_boltzmann(O):
    assert O.planck([667.0], 288.0)[0] == pytest.approx(0.13090535521240354, rel=1e-14)
    assert O.planck([1000.0], 250.0)[0] == pytest.approx(0.03783489465301437, rel=1e-14)
    nu = np.linspace(1e-3, 6000.0, 200001)
    B = O.planck(nu, 288.0)
    assert O.trapz(nu, math.pi * B) == pytest.approx(5.67037442e-8 * 288.0 ** 4, rel=2e-5)   # reference mixes CODATA years


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 16])
def test_streamnodes(O, golden, n):
    q = golden("quadrature")
    m, W = O.streamnodes(n)
    assert relerr(m, q[f"m{n}"]) < 1e-13 and relerr(W, q[f"W{n}"]) < 1e-13
    if n >= 5:
        assert abs(W.sum() - math.pi) < 2e-7     # sum(W) -> pi


@pytest.mark.parametrize("n", [2, 3, 4, 5])
def test_lobattonodes(O, golden, n):
    q = golden("quadrature")
    x, w = O.lobattonodes(n)
    assert np.max(np.abs(x - q[f"lx{n}"])) < 1e-15 and np.max(np.abs(w - q[f"lw{n}"])) < 1e-15


@pytest.mark.parametrize("gas,key,nukey", [("CO2", "voigt_co2", "nu_co2"), ("H2O", "voigt_h2o", "nu_h2o"),
                                           ("CO2", "lorentz_co2", "nu_co2"), ("CO2", "doppler_co2", "nu_co2"),
                                           ("CO2", "phco2_co2", "nu_co2")])
def test_lineshapes_vs_golden(O, golden, lines, gas, key, nukey):
    g = golden("lineshapes")
    shape = {"voigt": "voigt", "lorentz": "lorentz", "doppler": "doppler", "phco2": "PHCO2"}[key.split("_")[0]]
    cut = 500.0 if shape == "PHCO2" else 25.0
    for k, (T, P, Pp) in enumerate(g["states"]):
        s = O.shape_bang(shape, g[nukey], lines(gas), T, P, Pp, cut)
        ref = g[key][k]
        assert relerr(s, ref, floor=1e-12 * ref.max()) < 1e-10


def test_cutoff_edge_semantics(O, golden, lines):
    """cutline is strict (lines at exactly |dnu| = cut are summed) but includedlines(::Vector) drops lines sitting exactly
    at min(nu)-cut / max(nu)+cut (line_shapes.jl:10,21)."""
    g = golden("lineshapes")
    s = O.shape_bang("voigt", g["nu_edge"], lines("CO2"), *g["states"][1], 25.0)
    assert relerr(s, g["voigt_edge"]) < 1e-10
    s2 = O.shape_bang("voigt", g["nu_edge"], lines("CO2"), *g["states"][1], 25.0, strict_ends=False)
    assert s2[0] > s[0] or s2[2] > s[2] or np.allclose(s2, s, rtol=1e-6)


def test_voigt_normalised(O, cs):
    """int fvoigt dnu = 1 for one isolated line (S chosen so that S(T)=S at 296 K)."""
    par = dict(M=np.array([2], np.int16), I=np.array([1], np.int16), nu=np.array([1000.0]), S=np.array([1.0]),
               gamma_a=np.array([0.07]), gamma_s=np.array([0.09]), Epp=np.array([100.0]), na=np.array([0.7]))
    sl = cs.SpectralLines(par)
    nu = np.linspace(1000.0 - 24.0, 1000.0 + 24.0, 480001)
    s = O.shape_bang("voigt", nu, sl, 296.0, 2000.0, 0.0, 25.0)
    scale = O.chebyQrefQ(296.0, sl.cheb[0, : sl.ncheb[0]])
    # Lorentz wings outside +-24 cm^-1 carry gamma/(pi*24)*2 of the area
    gam = 0.07 * 2000.0 / 101325.0
    assert O.trapz(nu, s) / scale == pytest.approx(1.0 - 2 * gam / (math.pi * 24.0), rel=2e-6)


def _run_column(O, cs, lines, g, gases, extra=None):
    P, T = g["P"], g["T"]
    nlob, ns = int(g["nlobatto"]), int(g["nstream"])
    fT = cs.AtmosphericProfile(P, T)
    Tn, mun = cs.lobattoevaluations(P, fT, lambda *a: float(g["mu"]), nlob)
    Tlev = np.array([fT(p) for p in P])
    K = (len(P) - 1) * (nlob - 1) + 1
    conc = np.full((len(gases), K), float(g["conc"])) if gases else np.zeros((0, K))
    nnu = len(g["nu"])
    fS = float(g["fS"]) if "fS" in g.files else 0.0
    fa = float(g["fa"]) if "fa" in g.files else 0.0
    return O.fluxes_discretized(g["nu"], P, float(g["g"]), nlob, Tn, mun, Tlev, [lines(x) for x in gases],
                                ["voigt"] * len(gases), [25.0] * len(gases), conc,
                                sigma_gray=float(g["sigma"]) if not gases else 0.0, S_toa=np.full(nnu, fS),
                                albedo=np.full(nnu, fa), nstream=ns, want_sigma=True)


def test_column_gray_vs_golden(O, cs, golden, lines):
    g = golden("column_gray")
    r = _run_column(O, cs, lines, g, [])
    assert relerr(r["tau"], g["tau"]) < 1e-12
    scale = g["Mup"].max()
    assert np.max(np.abs(r["Mup"] - g["Mup"])) < 1e-12 * scale and np.max(np.abs(r["Mdn"] - g["Mdn"])) < 1e-12 * scale
    # F- near the top is ~1e-31 W/m^2 (pure rounding of the linear-in-tau source), so compare on the scale of the column
    assert relerr(r["Fup"], g["Fup"]) < 1e-11 and relerr(r["Fdn"], g["Fdn"], floor=1e-6 * g["Fdn"].max()) < 1e-11


@pytest.mark.parametrize("name", ["column_co2", "column_co2_lob4"])
def test_column_co2_vs_golden(O, cs, golden, lines, name):
    g = golden(name)
    r = _run_column(O, cs, lines, g, ["CO2"])
    assert relerr(r["sigma"], g["sigma"], floor=1e-12 * g["sigma"].max()) < 1e-10
    assert relerr(r["tau"], g["tau"]) < 1e-10
    scale = g["Mup"].max()
    assert np.max(np.abs(r["Mup"] - g["Mup"])) < 1e-10 * scale and np.max(np.abs(r["Mdn"] - g["Mdn"])) < 1e-10 * scale
    assert relerr(r["Fup"], g["Fup"]) < 1e-10 and relerr(r["Fdn"], g["Fdn"], floor=1e-6 * g["Fdn"].max()) < 1e-10


def test_gray_olr_analytic(O, cs):
    """Config 1 (test/test_gray.jl:13-24,54-72): gray gas on a dry adiabat, one vertical stream of weight pi; OLR within
    1 % of Pierrehumbert eq. 4.32.  The reference integrates to 1e-6 Pa with an adaptive ODE; the fixed grid here uses
    enough layers for the optically thickest case (achieved errors are recorded in DESIGN.md)."""
    Rg, g_, mu, cp, Ps, Ts = 8.31446262, 10.0, 0.01, 1e3, 1e5, 300.0
    nu = np.concatenate([cs.logrange(1e-6, 1e5, 4000, 4), [1e6]])
    P = cs.pressuregrid(1e-3, Ps, 401)
    T = Ts * (P / Ps) ** (Rg / (mu * cp))
    gam = Rg / (mu * cp)
    for sigma in 10.0 ** np.linspace(-29, -23, 7):
        tau_inf = cs.dtaudP(sigma, g_, mu) * Ps
        f = lambda t: math.exp(-t) * t ** (4 * gam)
        integral = quad(f, 0, tau_inf, epsabs=0, epsrel=1e-10, limit=500)[0]
        exact = 5.67037442e-8 * Ts ** 4 * (math.exp(-tau_inf) + tau_inf ** (-4 * gam) * integral)
        beta = np.full(len(P), 1e-4 * 6.02214076e23 / g_ * sigma / mu)
        tau = O.depth_bang(P, beta, 2)
        olr = np.zeros(len(nu))
        m, W = np.array([1.0]), np.array([math.pi])
        with np.errstate(over="ignore"):
            Ball = np.array([cs.planck(nu, t) for t in T])     # (np, nnu), radiation.jl:48-54
        for j, v in enumerate(nu):
            B = np.ascontiguousarray(Ball[:, j])
            Mup, Mdn = np.zeros(len(P)), np.zeros(len(P))
            O.lib().cso_monoflux_bang(O._p(Mup), O._p(Mdn), O._p(tau), len(P), O._p(B), 0.0, 0.0, 0.841, 1, O._p(m), O._p(W))
            olr[j] = Mup[0]
        num = O.trapz(nu, olr)
        assert abs(num / exact - 1) < 0.01, (sigma, num, exact)


def test_alg985_backend_is_labelled_and_self_consistent(O):
    """The oracle's second Faddeeva back-end restates ACM TOMS Algorithm 985 (what the reference's Faddeyeva985 dependency
    implements; source absent, unverifiable).  It exists to quantify the expected gap to the Julia reference (tools/alg985_gap.py),
    never for parity: the default back-end is the exact one, and the restatement reproduces the accuracy class the paper states."""
    from scipy.special import wofz
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(0, 8, 50000), 10 ** rng.uniform(-3, 4, 50000)])
    y = np.concatenate([10 ** rng.uniform(-8, 1, 50000), 10 ** rng.uniform(-6, 3, 50000)])
    ex = wofz(x + 1j * y).real
    assert O.lib().cso_get_faddeeva_backend() == 0
    assert np.max(np.abs(O.faddeeva(x, y) / ex - 1)) < 2e-13
    with O.faddeeva_backend("alg985"):
        e = np.max(np.abs(O.faddeeva(x, y) / ex - 1))
    assert 1e-5 < e < 6e-5                                   # paper: "< 4e-5"; this restatement: 4.9e-5
    assert O.lib().cso_get_faddeeva_backend() == 0           # restored
