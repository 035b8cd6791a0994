"""BASELINE.json configs[4] AS WRITTEN: H2O+CO2+CH4+O3 with the CIA continuum, 5e5 wavenumbers x 100 layers, fp64 and the
fp32 mixed-precision variant with its tolerance sweep -- against the CPU oracle.

  * on a 24 001-point window of the full grid (same 0.005 cm^-1 spacing, so the interpolated far wings and the fp32 body run at
    the nodes exactly as at full size) EVERY output of the oracle is compared: sigma, tau <= 1e-11 (fp64) / <= 1e-6 relative
    (mixed, far_s = 1e6 and 1e8: the north-star tolerance), band fluxes and the window's "OLR" [W/m^2];
  * at the full 5e5 x 100 size: 48 random wavenumber columns recomputed by the oracle (fp64 and mixed), the band-integral
    identity, boundary identities, tau floor, bitwise repeatability.
"""
import math

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


def _oracle(cs, O, cfg, col, nu):
    """The oracle on wavenumbers `nu` with the column's node states: line gases + CIA through the numpy CIA restatement."""
    import workloads as W
    d = [cs.readcia(W.fixture(f)) for f in ("CO2-CO2_2018.cia", "CO2-CH4_2018.cia")]
    extra = np.zeros((col.K, len(nu)))
    for k in range(col.K):
        for ci in range(len(col.U.cia)):
            extra[k] += O.cia_sigma(d[ci], nu, col.Tk[k], col.Pk[k], col.cia_P1[ci, k], col.cia_P2[ci, k])
    return O.fluxes_discretized(nu, cfg["P"], cfg["g"], 2, col.Tn, col.mun, col.Tlev, [g.sl for g in col.gases], ["voigt"] * 4,
                                [25.0] * 4, col.conc, sigma_extra=extra, theta_s=cfg["theta_s"], nstream=5, want_sigma=True)


def test_c5_window_fp64_and_mixed_sweep_vs_oracle(cs, O):
    import workloads as W
    cfg = W.config("C5", nnu=24001, nu_span=(600.0, 720.0))          # the full grid's spacing (2499/499999 = 0.005 cm^-1)
    ctx = cs.Context(0)

    def run():
        col = cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], 0.0, 0.0, *cfg["absorbers"], core=cfg["core"], theta_s=cfg["theta_s"],
                        ctx=ctx)
        col.run()
        tau = np.zeros((col.nl, col.nnu), order="F")
        F = col.fetch(tau)
        return col, col.sigma_nodes(), tau, F, col.work()

    col, s64, t64, F64, w = run()
    assert len(col.gases) == 4 and len(col.U.cia) == 2 and col.K == 101 and w["levels"] >= 3 and w["node_evals"] > 0
    r = _oracle(cs, O, cfg, col, cfg["nu"])
    m = r["sigma"] > 0
    assert np.array_equal(s64 > 0, m)
    assert np.max(np.abs(s64[m] / r["sigma"][m] - 1)) < 1e-11 and relerr(t64, r["tau"]) < 1e-11
    assert np.max(np.abs(F64[0] - r["Fup"])) < 1e-11 * r["Fup"].max() and np.max(np.abs(F64[1] - r["Fdn"])) < 1e-11 * r["Fup"].max()
    errs = {}
    for far_s in (1e6, 1e8):                                          # the tolerance sweep of configs[4]
        ctx.set_precision("mixed", far_s)
        _, s32, t32, F32, _ = run()
        es, et = np.max(np.abs(s32[m] / r["sigma"][m] - 1)), relerr(t32, r["tau"])
        eo = abs(F32[0][0] - r["Fup"][0])
        errs[far_s] = (es, et, eo)
        assert 0 < es < 1e-6 and et < 1e-6                            # north star: 1e-6 relative
        assert eo < 1e-6 * r["Fup"][0]                                # band flux at the top [W/m^2], same relative bar
        assert np.max(np.abs(F32[0] - r["Fup"])) < 1e-6 * r["Fup"].max()
    assert errs[1e8][0] <= errs[1e6][0]                               # a wider fp64 region can only tighten the agreement
    ctx.set_precision("fp64")
    _, s64b, _, F64b, _ = run()
    assert np.array_equal(s64b, s64) and np.array_equal(F64b[0], F64[0])   # fp64 mode is untouched by the switch
    print("C5 window: mixed-precision errors (max rel sigma, max rel tau, |dF_toa| W/m^2):", errs)
    ctx.close()


@pytest.fixture(scope="module")
def c5(cs):
    import workloads as W
    cfg = W.config("C5")
    ctx = cs.Context(0)
    col = cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], cfg["fS"], cfg["fa"], *cfg["absorbers"], core=cfg["core"],
                    theta_s=cfg["theta_s"], ctx=ctx)
    col.run()
    F = cs.FluxPack(col.np, col.nnu)
    F.Fup[:], F.Fdn[:] = col.fetch(F.tau, F.Mup, F.Mdn)
    yield cfg, ctx, col, F
    ctx.close()


def test_c5_full_size_sparse_parity_and_properties(cs, O, c5):
    cfg, ctx, col, F = c5
    assert col.nnu == 500_000 and col.nl == 100 and len(col.gases) == 4 and len(col.U.cia) == 2
    idx = np.sort(np.random.default_rng(23).choice(col.nnu, 48, replace=False))
    r = _oracle(cs, O, cfg, col, cfg["nu"][idx])
    assert relerr(F.tau[:, idx], r["tau"]) < 1e-11
    sm = r["Mup"].max()
    assert np.max(np.abs(F.Mup[:, idx] - r["Mup"])) < 1e-11 * sm and np.max(np.abs(F.Mdn[:, idx] - r["Mdn"])) < 1e-11 * sm
    # size-independent properties (as for C3)
    w = cs.trapz_weights(cfg["nu"])
    assert relerr(F.Fup, F.Mup @ w) < 1e-12 and relerr(F.Fdn, F.Mdn @ w, floor=1e-9) < 1e-12     # intF! on device
    assert np.all(F.Mdn[0] == 0.0)
    assert relerr(F.Mup[-1], math.pi * cs.planck(cfg["nu"], col.Tlev[-1])) < 1e-13
    assert np.all(F.tau >= 1e-6) and np.all(np.isfinite(F.tau)) and np.all(np.isfinite(F.Mup))
    assert 50.0 < F.Fup[0] < 400.0
    col.run()
    F2 = col.fetch()
    assert np.array_equal(F2[0], F.Fup) and np.array_equal(F2[1], F.Fdn)                         # bitwise repeatable


def test_c5_full_size_mixed_precision(cs, O, c5):
    """configs[4] to the letter: the fp32 mixed-precision variant on the 4-gas + CIA workload at 5e5 x 100."""
    cfg, ctx, col, F = c5
    idx = np.sort(np.random.default_rng(29).choice(col.nnu, 48, replace=False))
    r = _oracle(cs, O, cfg, col, cfg["nu"][idx])
    res = {}
    try:
        for far_s in (1e6, 1e8):
            ctx.set_precision("mixed", far_s)
            col.run()
            tau = np.zeros((col.nl, col.nnu), order="F")
            Fm = col.fetch(tau)
            et = relerr(tau[:, idx], r["tau"])
            assert 0 < et < 1e-6                                                                 # vs the oracle
            assert relerr(tau, F.tau) < 1e-6                                                     # vs the fp64 device path, every point
            dolr = abs(Fm[0][0] - F.Fup[0])
            assert dolr < 1e-5                                                                   # OLR [W/m^2]
            res[far_s] = (et, dolr)
        assert res[1e8][1] <= res[1e6][1] + 1e-9
    finally:
        ctx.set_precision("fp64")
    print("C5 full size, mixed precision: (max rel tau on 48 columns, |dOLR| W/m^2):", res)
