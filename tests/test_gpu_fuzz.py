"""Randomised GPU-vs-oracle parity on awkward inputs: non-uniform / tiny / gappy wavenumber grids, duplicated and very dense
lines (the near-line queue overflows into its fallback), light molecules and hot gas (near zone = whole window), tiny and
huge cut-offs, zero and very high pressure.  Same tolerances as test_gpu_parity.py."""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


def _table(cs, rng, M, L, lo, hi, dense=None, dup=False):
    nu = np.sort(rng.uniform(lo, hi, L))
    if dense is not None:                      # a clump of lines inside ~1 cm^-1
        c0 = rng.uniform(lo, hi)
        nu[: dense] = c0 + rng.uniform(0, 1.0, dense)
        nu = np.sort(nu)
    if dup:
        nu[1::7] = nu[0::7][: len(nu[1::7])]
        nu = np.sort(nu)
    niso = len(cs.MOLPARAM[M].I)
    iso = rng.integers(1, min(niso, 3) + 1, L).astype(np.int16)
    par = dict(M=np.full(L, M, np.int16), I=iso, nu=nu, S=10.0 ** rng.uniform(-27, -19, L), gamma_a=rng.uniform(0.01, 0.12, L),
               gamma_s=rng.uniform(0.02, 0.5, L), Epp=rng.uniform(0, 4000, L), na=rng.uniform(0.3, 0.9, L))
    return cs.SpectralLines(par)


def _grid(rng, kind, lo, hi, n):
    if kind == "uniform":
        return np.linspace(lo, hi, n)
    if kind == "log":
        return np.unique(np.exp(np.linspace(np.log(max(lo, 1e-3)), np.log(hi), n)))
    if kind == "random":
        return np.unique(np.sort(rng.uniform(lo, hi, n)))
    # clustered with gaps
    a = np.concatenate([rng.normal(c, 0.05 * (hi - lo) / 10, n // 4) for c in rng.uniform(lo, hi, 4)])
    return np.unique(np.clip(np.sort(a), lo, hi))


CASES = [
    # (molecule, L, line range, grid kind, grid range, nnu, cut, dense, dup, states)
    (2, 3000, (500, 900), "uniform", (550, 850), 1000, 25.0, None, False, [(250, 1e4, 4.0), (180, 30.0, 0.0)]),
    (1, 2000, (1, 300), "log", (0.5, 280), 700, 25.0, None, True, [(296, 101325.0, 1e3), (300, 0.0, 0.0)]),
    (2, 4000, (600, 700), "random", (590, 710), 513, 25.0, 1500, False, [(220, 50.0, 0.01), (260, 5e3, 50.0)]),
    (45, 500, (300, 4000), "clustered", (200, 4200), 900, 25.0, None, False, [(1000, 1e5, 1e5), (25, 1e3, 10.0)]),   # H2: light, hot
    (2, 1500, (2000, 2400), "uniform", (2100, 2300), 65, 0.05, None, False, [(250, 1e4, 4.0)]),                      # tiny cut-off
    (2, 1500, (100, 3000), "uniform", (1000, 2000), 129, 600.0, None, False, [(250, 3e5, 3e5)]),                     # huge cut-off, 3 bar
    (6, 800, (1200, 1400), "uniform", (1299.9, 1300.1), 64, 25.0, 300, True, [(200, 1.0, 0.0), (320, 2e4, 1.0)]),  # 2e-4 cm^-1 spacing
    (2, 50, (660, 670), "uniform", (667.0, 667.0001), 2, 25.0, None, False, [(250, 1e3, 1.0)]),
    (2, 50, (660, 670), "uniform", (667.3, 667.3), 1, 25.0, None, False, [(250, 1e3, 1.0)]),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_shape_batch_fuzz(cs, O, case):
    M, L, (llo, lhi), kind, (glo, ghi), n, cut, dense, dup, states = CASES[case]
    rng = np.random.default_rng(100 + case)
    sl = _table(cs, rng, M, L, llo, lhi, dense, dup)
    nu = _grid(rng, kind, glo, ghi, n) if n > 1 else np.array([glo])
    T, P, Pp = map(list, zip(*states))
    ctx = cs.Context(0)
    for shape in ("voigt", "lorentz", "doppler"):
        sg = cs.shape_batch(sl, shape, nu, T, P, Pp, cut, ctx)
        for k in range(len(T)):
            so = O.shape_bang(shape, nu, sl, T[k], P[k], Pp[k], cut)
            assert np.array_equal(sg[k] == 0, so == 0), (shape, k)
            tol = 1e-11 if shape != "doppler" else 1e-9      # exp(-x^2) at x^2 ~ 700 amplifies an ulp of x^2 by 700
            assert relerr(sg[k], so, floor=1e-280) < tol, (shape, k, relerr(sg[k], so, floor=1e-280))
    ctx.close()


@pytest.mark.parametrize("seed", range(4))
def test_column_fuzz(cs, O, seed):
    rng = np.random.default_rng(500 + seed)
    n = int(rng.integers(100, 1500))
    nu = _grid(rng, ["uniform", "log", "random", "clustered"][seed], 50.0, 2500.0, n)
    g1 = cs.DirectGas(_table(cs, rng, 2, 2500, 1, 2600, dense=400 if seed == 2 else None), float(rng.uniform(1e-5, 0.3)), nu)
    g2 = cs.DirectGas(_table(cs, rng, 1, 1500, 1, 2600), lambda T, P: min(0.05, 1e-3 * (P / 1e5) ** 2 * (T / 250) ** 4), nu)
    npl = int(rng.integers(3, 30))
    P = cs.pressuregrid(float(10 ** rng.uniform(-2, 1)), float(10 ** rng.uniform(4, 5.5)), npl)
    T = np.sort(rng.uniform(150, 400, npl))
    nlob, ns = int(rng.integers(2, 6)), int(rng.integers(1, 9))
    ctx = cs.Context(0)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        col = cs.Column(P, float(rng.uniform(3, 25)), T, float(rng.uniform(0.002, 0.05)), float(rng.uniform(0, 5)), float(rng.uniform(0, 1)),
                        g1, g2, core=cs.Discretized(ns, nlob), theta_s=float(rng.uniform(0, 1.4)), ctx=ctx)
    col.run()
    F = cs.FluxPack(npl, len(nu))
    F.Fup[:], F.Fdn[:] = col.fetch(F.tau, F.Mup, F.Mdn)
    r = O.fluxes_discretized(nu, P, col.g, nlob, col.Tn, col.mun, col.Tlev, [g1.sl, g2.sl], ["voigt"] * 2, [25.0] * 2, col.conc,
                             S_toa=col.S_toa, albedo=col.albedo, theta_s=col.theta_s, nstream=ns, want_sigma=True)
    assert relerr(col.sigma_nodes(), r["sigma"], floor=1e-280) < 1e-11
    assert relerr(F.tau, r["tau"]) < 1e-11
    sm = max(r["Mup"].max(), r["Mdn"].max())
    assert np.max(np.abs(F.Mup - r["Mup"])) < 1e-11 * sm and np.max(np.abs(F.Mdn - r["Mdn"])) < 1e-11 * sm
    fm = max(np.abs(r["Fup"]).max(), np.abs(r["Fdn"]).max())
    assert np.max(np.abs(F.Fup - r["Fup"])) < 1e-11 * fm and np.max(np.abs(F.Fdn - r["Fdn"])) < 1e-11 * fm
    ctx.close()


# ---- interpolated far wings (K2c): the same device path with and without interpolation on random fine grids --------------

@pytest.mark.parametrize("seed", range(16))
def test_interp_fuzz(cs, O, seed):
    """Random grid kind / size / spacing / cut-off / molecule / states; interpolation on vs off (every pair evaluated) at
    2e-13, and vs the oracle at 1e-11 on a subset of the points."""
    rng = np.random.default_rng(9000 + seed)
    cut = float(rng.choice([1.0, 5.0, 25.0, 25.0, 60.0]))
    nlev_target = int(rng.integers(1, 6))
    n = int(rng.integers(130, 9000))
    # spacing so that the interval of 128 * 2^(nlev_target-1) points is the largest one that fits 2.3 W <= 1.5 cut
    dnu = 1.5 * cut / 2.3 / (128 * 2 ** (nlev_target - 1)) * float(rng.uniform(0.55, 0.95))
    c0 = float(rng.uniform(300, 2500))
    span = dnu * (n - 1)
    kind = ["uniform", "random", "log", "jitter"][seed % 4]
    if kind == "uniform":
        nu = c0 + dnu * np.arange(n)
    elif kind == "random":
        nu = np.unique(c0 + np.sort(rng.uniform(0, span, n)))
    elif kind == "log":
        nu = np.unique(c0 * np.exp(np.linspace(0, np.log1p(span / c0), n)))
    else:
        nu = c0 + dnu * (np.arange(n) + rng.uniform(-0.4, 0.4, n))
    M = int(rng.choice([1, 2, 2, 6, 45]))
    L = int(rng.integers(200, 6000))
    sl = _table(cs, rng, M, L, c0 - 2 * cut - 5, c0 + span + 2 * cut + 5, dense=int(L // 4) if seed % 5 == 0 else None, dup=seed % 7 == 0)
    K = int(rng.integers(1, 40))
    T = rng.uniform(25, 1000, K) if M != 45 else rng.uniform(100, 1000, K)
    P = 10 ** rng.uniform(-1, 5.5, K)
    P[rng.random(K) < 0.1] = 0.0
    Pp = P * rng.uniform(0, 1, K)
    on, off, onv = cs.Context(0), cs.Context(0), cs.Context(0)
    off.set_interp(False)
    on.set_matrix_cores(2)     # far lines on the matrix cores wherever the series holds (K2d, K2e), whatever the grid length
    onv.set_matrix_cores(0)    # ... and everything on the vector unit
    a = cs.shape_batch(sl, "voigt", nu, list(T), list(P), list(Pp), cut, on)
    av = cs.shape_batch(sl, "voigt", nu, list(T), list(P), list(Pp), cut, onv)
    b = cs.shape_batch(sl, "voigt", nu, list(T), list(P), list(Pp), cut, off)
    assert np.array_equal(a == 0, b == 0) and np.array_equal(av == 0, b == 0)
    assert relerr(a, b, floor=1e-280) < 2e-13, (seed, kind, cs.interp_plan(nu, cut), relerr(a, b, floor=1e-280))
    assert relerr(av, b, floor=1e-280) < 2e-13, (seed, kind, cs.interp_plan(nu, cut), relerr(av, b, floor=1e-280))
    idx = np.sort(rng.choice(len(nu), min(len(nu), 300), replace=False))
    for k in rng.choice(K, min(K, 3), replace=False):
        so = O.shape_bang("voigt", nu, sl, T[k], P[k], Pp[k], cut)[idx] if len(nu) <= 3000 else None
        if so is not None:
            assert relerr(a[k][idx], so, floor=1e-280) < 1e-11
    on.close(); off.close(); onv.close()


@pytest.mark.parametrize("seed", range(6))
def test_interp_fuzz_long(cs, seed):
    """The same on grids LONG enough for the forms a short grid never reaches (every grid above is below 141 tiles): one
    (tile, group) per wave in k_voigt_edge_mx with its phases and the 16-node path of its window ends, one (interval, group) per
    wave in k_cheb_nodes_mx with its far pieces on 16 / 32 nodes -- or on 64 where the node positions' rounding says so (seeds
    with a small cut-off or a large nu; round 5: a latent 7e-13 at cut-off 1 cm^-1).  Interpolation on vs off at 2e-13."""
    rng = np.random.default_rng(9500 + seed)
    cut = [25.0, 5.0, 1.0, 25.0, 8.0, 60.0][seed]
    nlev_target = int(rng.integers(1, 4))
    n = int(rng.integers(92_000, 120_000))                          # >= 1438 tiles, >= 719 intervals: with three state groups both kernels
                                                                     # take one item per wave (cs_api.hip: mx_big)
    dnu = 1.5 * cut / 2.3 / (128 * 2 ** (nlev_target - 1)) * float(rng.uniform(0.55, 0.95))
    c0 = float(rng.uniform(300, 2500))
    span = dnu * (n - 1)
    kind = ["uniform", "jitter", "random", "log", "jitter", "uniform"][seed]
    if kind == "uniform":
        nu = c0 + dnu * np.arange(n)
    elif kind == "random":
        nu = np.unique(c0 + np.sort(rng.uniform(0, span, n)))
    elif kind == "log":
        nu = np.unique(c0 * np.exp(np.linspace(0, np.log1p(span / c0), n)))
    else:
        nu = c0 + dnu * (np.arange(n) + rng.uniform(-0.4, 0.4, n))
    M = int(rng.choice([1, 2, 6]))
    L = int(min(60_000, max(4_000, 12.0 * (span + 4 * cut))))       # ~12 lines per cm^-1: every level stays in use
    sl = _table(cs, rng, M, L, c0 - 2 * cut - 5, c0 + span + 2 * cut + 5)
    K = int(rng.integers(33, 48))                                    # three state groups
    T = rng.uniform(150, 350, K)
    P = np.sort(10 ** rng.uniform(0, 5.3, K))
    Pp = P * rng.uniform(0, 1, K)
    on, off = cs.Context(0), cs.Context(0)
    off.set_interp(False)
    on.set_matrix_cores(2)     # (whatever the table's density)
    a = cs.shape_batch(sl, "voigt", nu, list(T), list(P), list(Pp), cut, on)
    b = cs.shape_batch(sl, "voigt", nu, list(T), list(P), list(Pp), cut, off)
    plan = cs.interp_plan(nu, cut)
    assert len(plan) >= 1, plan
    assert np.array_equal(a == 0, b == 0)
    assert relerr(a, b, floor=1e-280) < 2e-13, (seed, kind, cut, len(nu), plan, relerr(a, b, floor=1e-280))
    on.close(); off.close()


def test_interp_column_mixed_gases(cs, O):
    """Two Voigt gases with different cut-offs (levels follow the narrower one), a Lorentz gas in between, gray term, stellar
    beam: on vs off and vs the oracle."""
    rng = np.random.default_rng(77)
    nu = np.linspace(900.0, 1100.0, 12001)
    g1 = cs.DirectGas(_table(cs, rng, 2, 3000, 800, 1200), 3e-4, nu, dnu_cut=25.0)
    g2 = cs.DirectGas(_table(cs, rng, 1, 2000, 800, 1200), 2e-3, nu, dnu_cut=8.0)
    g3 = cs.DirectGas(_table(cs, rng, 6, 500, 800, 1200), 1e-5, nu, shape="lorentz", dnu_cut=25.0)
    P = cs.pressuregrid(5.0, 1e5, 12)
    T = np.linspace(210.0, 295.0, 12)
    out = {}
    for flag in (True, False):
        ctx = cs.Context(0)
        ctx.set_interp(flag)
        ctx.set_matrix_cores(2)
        col = cs.Column(P, 9.8, T, 0.029, 1.0, 0.1, g1, g3, g2, cs.GrayGas(1e-27, nu), core=cs.Discretized(4, 3), theta_s=0.5, ctx=ctx)
        col.run()
        F = cs.FluxPack(len(P), len(nu))
        F.Fup[:], F.Fdn[:] = col.fetch(F.tau, F.Mup, F.Mdn)
        out[flag] = (F, col.sigma_nodes(), col.work(), col)
    Fon, son, won, col = out[True]
    Foff, soff, woff, _ = out[False]
    assert won["levels"] >= 2 and woff["levels"] == 0
    assert relerr(son, soff) < 2e-13 and relerr(Fon.tau, Foff.tau) < 2e-13
    idx = np.sort(rng.choice(len(nu), 200, replace=False))
    r = O.fluxes_discretized(nu[idx], P, col.g, 3, col.Tn, col.mun, col.Tlev, [g1.sl, g3.sl, g2.sl], ["voigt", "lorentz", "voigt"],
                             [25.0, 25.0, 8.0], col.conc, S_toa=col.S_toa[idx], albedo=col.albedo[idx], theta_s=0.5, nstream=4,
                             sigma_gray=1e-27)
    assert relerr(Fon.tau[:, idx], r["tau"]) < 1e-11


@pytest.mark.parametrize("kind,cut", [("jitter", 5.0), ("log", 25.0)])
def test_interp_column_long_nonuniform(cs, kind, cut):
    """A whole column on a long NON-UNIFORM grid (the bench grids are uniform): three streams, level cascade on the node-sum
    stream, tile nodes of the window ends, k_cheb_apply_mfma + k_rt -- interpolation on vs off: sigma at the nodes, tau, fluxes."""
    rng = np.random.default_rng(4242)
    n = 100_000
    if kind == "jitter":
        dnu = 0.004
        nu = 1800.0 + dnu * (np.arange(n) + rng.uniform(-0.4, 0.4, n))
    else:
        nu = np.unique(400.0 * np.exp(np.linspace(0.0, np.log(6.0), n)))       # 400 .. 2400 cm^-1, spacing 0.007 .. 0.043
    lo, hi = nu[0] - 2 * cut - 5, nu[-1] + 2 * cut + 5
    g1 = cs.DirectGas(_table(cs, rng, 2, int(15 * (hi - lo)), lo, hi), 4e-4, nu, dnu_cut=cut)
    g2 = cs.DirectGas(_table(cs, rng, 1, int(8 * (hi - lo)), lo, hi), 3e-3, nu, dnu_cut=cut)
    P = cs.pressuregrid(2.0, 1e5, 34)
    T = np.linspace(200.0, 290.0, 34)
    out = {}
    for flag in (True, False):
        ctx = cs.Context(0)
        ctx.set_interp(flag)
        col = cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, g1, g2, core=cs.Discretized(5, 2), ctx=ctx)
        col.run()
        F = cs.FluxPack(len(P), len(nu))
        F.Fup[:], F.Fdn[:] = col.fetch(F.tau, F.Mup, F.Mdn)
        out[flag] = (F, col.sigma_nodes(), col.work())
        ctx.close()
    Fon, son, won = out[True]
    Foff, soff, woff = out[False]
    assert won["levels"] >= 2 and woff["levels"] == 0
    assert won["direct_evals_matrix"] > 0 and won["node_evals_matrix"] > 0, won       # (the matrix-core forms are what ran)
    assert relerr(son, soff) < 2e-13 and relerr(Fon.tau, Foff.tau) < 2e-13
    assert relerr(Fon.Fup, Foff.Fup) < 2e-13 and np.max(np.abs(Fon.Mup - Foff.Mup)) < 1e-12 * np.max(Foff.Mup)   # (M: exp(-tau m) of a tau at 2e-13)
