"""k_flux: the flux kernel that finishes the cross-sections on chip (interpolated wings as a matrix product, CIA pairs, near-line plane)
and adds the block partials itself -- fluxes.jl:270-277 does depth and flux of a wavenumber in one loop body -- against the separate
kernels (k_cheb_apply_mfma, k_cia, k_fold, k_rt / k_rt_streams, k_freduce: cs_set_tuning key 15 = 1) and against the oracle.

k_flux_streams and k_flux_chunk run the separate kernels' operations in the same order, so optical depths and monochromatic fluxes of
line-by-line columns must come out BITWISE equal; band fluxes differ by the order in which block partials are added (1e-15), CIA terms
by the order of the bilinear interpolation (1e-14 of the CIA term).  k_flux_scan (the default on short grids) walks the layers in chunks:
optical depths bitwise, intensities to a few units in the last place (the intensity entering a chunk is formed as A I + B instead of
layer by layer; asserted at 1e-13 of the column maximum).  Tolerances vs the oracle as everywhere: 1e-11.
"""
import numpy as np
import pytest

import workloads as W
from conftest import relerr, source_rounding_bound

pytestmark = pytest.mark.gpu


def _run(cs, ctx, P, T, absorbers, core, fS=0.0, fa=0.0, **kw):
    col = cs.Column(P, 9.8, T, 0.029, fS, fa, *absorbers, core=core, ctx=ctx, **kw)
    col.run()
    tau = np.zeros((col.nl, col.nnu), order="F")
    Mu = np.zeros((col.np, col.nnu), order="F")
    Md = np.zeros((col.np, col.nnu), order="F")
    Fup, Fdn = col.fetch(tau, Mu, Md)
    return col, dict(tau=tau, Mup=Mu, Mdn=Md, Fup=Fup, Fdn=Fdn, launches=col.info()["launches"])


def _ctx(cs, key15):
    c = cs.Context(0)
    c.set_tuning(15, key15)
    return c


@pytest.mark.parametrize("nnu,nlob,ns,fS,fa", [(6000, 2, 5, 0.0, 0.0), (6001, 3, 4, 0.3, 0.2), (2500, 4, 8, 0.0, 0.15), (20000, 2, 5, 0.0, 0.0), (4100, 2, 2, 0.7, 0.0)])
def test_short_grid_forms_vs_separate_kernels(cs, O, lines, nnu, nlob, ns, fS, fa, key15=0):
    """short grids -- k_flux_scan (sweeps as a scan over layer chunks): line-by-line H2O + CO2, with and without stellar beam / albedo,
    ragged last tile, 13 layers over 4 waves.  (The first short-grid form, k_flux_streams -- one wave per stream and sweep, bitwise the
    separate kernels -- lost its A/B in round 4 and was removed in round 5.)"""
    nu = np.linspace(580.0, 780.0, nnu)
    P = cs.pressuregrid(5.0, 1e5, 14)
    T = W.earth_temperature(P)
    gases = (cs.DirectGas(lines("H2O"), W.fC_h2o, nu), cs.DirectGas(lines("CO2"), 400e-6, nu))
    core = cs.Discretized(ns, nlob)
    a_ctx, b_ctx = _ctx(cs, key15), _ctx(cs, 1)
    col, a = _run(cs, a_ctx, P, T, gases, core, fS, fa)
    _, b = _run(cs, b_ctx, P, T, gases, core, fS, fa)
    assert a["launches"] < b["launches"]          # wings, plane fold, sweeps and the band sum in one launch
    assert col.info()["flux_form"] == (1 if key15 else 3)
    sm_ = max(b["Mup"].max(), b["Mdn"].max())
    if key15:
        assert np.array_equal(a["tau"], b["tau"])
    else:      # (the interval levels are folded into the smallest one before the carry: the same polynomials, other roundings)
        assert relerr(a["tau"], b["tau"]) < 5e-13
    if key15:
        assert np.array_equal(a["Mup"], b["Mup"]) and np.array_equal(a["Mdn"], b["Mdn"])
    else:
        amp = source_rounding_bound(cs, nu, col.Tlev, b["tau"])     # (optical depths that differ in the last bit, through (1 - t) / tau: see conftest)
        assert np.max(np.abs(a["Mup"] - b["Mup"])) < 1e-13 * sm_ + amp and np.max(np.abs(a["Mdn"] - b["Mdn"])) < 1e-13 * sm_ + amp
    fm = np.max(b["Fup"])
    assert np.max(np.abs(a["Fup"] - b["Fup"])) < 1e-13 * fm and np.max(np.abs(a["Fdn"] - b["Fdn"])) < 1e-13 * fm
    r = O.fluxes_discretized(nu, P, 9.8, nlob, col.Tn, col.mun, col.Tlev, [g.sl for g in col.gases], ["voigt"] * 2, [25.0] * 2, col.conc,
                             S_toa=col.S_toa, albedo=col.albedo, nstream=ns)
    assert relerr(a["tau"], r["tau"]) < 1e-11
    sm = max(r["Mup"].max(), r["Mdn"].max())
    assert np.max(np.abs(a["Mup"] - r["Mup"])) < 1e-11 * sm and np.max(np.abs(a["Mdn"] - r["Mdn"])) < 1e-11 * sm
    assert np.max(np.abs(a["Fup"] - r["Fup"])) < 1e-11 * r["Fup"].max()
    # repeatable bit for bit (the last block adds the partials in a fixed order), also after a state update and back
    col.update(T + 2.0)
    col.run()
    col.update(T)
    col.run()
    F2 = col.fetch()
    assert np.array_equal(F2[0], a["Fup"]) and np.array_equal(F2[1], a["Fdn"])
    # the node cross-sections are still there for whoever asks (evaluated again, all the way into HBM)
    sig = col.sigma_nodes()
    bcol = cs.Column(P, 9.8, T, 0.029, fS, fa, *gases, core=core, ctx=b_ctx)
    bcol.run()
    assert relerr(sig, bcol.sigma_nodes(), floor=1e-300) < 5e-13
    a_ctx.close(); b_ctx.close()


@pytest.mark.parametrize("nnu,nlob,fa", [(40000, 2, 0.0), (30011, 3, 0.25)])
def test_chunk_form_bitwise_vs_separate_kernels(cs, lines, nnu, nlob, fa):
    """long-grid form (k_flux_chunk), forced on a mid-size grid: one wave per tile, cross-sections finished 16 states at a time"""
    nu = np.linspace(560.0, 800.0, nnu)
    P = cs.pressuregrid(5.0, 1e5, 22)                 # K = 22 or 43: more than one 16-state chunk, a partial last chunk
    T = W.earth_temperature(P)
    gases = (cs.DirectGas(lines("H2O"), W.fC_h2o, nu), cs.DirectGas(lines("CO2"), 400e-6, nu))
    core = cs.Discretized(5, nlob)
    a_ctx, b_ctx = _ctx(cs, 2), _ctx(cs, 1)
    _, a = _run(cs, a_ctx, P, T, gases, core, 0.0, fa)
    _, b = _run(cs, b_ctx, P, T, gases, core, 0.0, fa)
    assert a["launches"] < b["launches"]
    assert np.array_equal(a["tau"], b["tau"]) and np.array_equal(a["Mup"], b["Mup"]) and np.array_equal(a["Mdn"], b["Mdn"])
    fm = np.max(b["Fup"])
    assert np.max(np.abs(a["Fup"] - b["Fup"])) < 1e-14 * fm and np.max(np.abs(a["Fdn"] - b["Fdn"])) < 1e-14 * fm
    # band fluxes only (no tau / M outputs): the optical depths go through scratch
    col = cs.Column(P, 9.8, T, 0.029, 0.0, fa, *gases, core=core, ctx=a_ctx, want_tau=False, want_M=False)
    col.run()
    F = col.fetch()
    assert np.array_equal(F[0], a["Fup"]) and np.array_equal(F[1], a["Fdn"])
    a_ctx.close(); b_ctx.close()


@pytest.mark.parametrize("key15,nnu", [(0, 9000), (2, 36000)])
def test_fused_cia_and_gray_vs_separate_and_oracle(cs, O, lines, key15, nnu):
    """CO2 + CH4 line-by-line with both CIA pairs of the fixtures (CO2-CO2: several bands; CO2-CH4), a gray term and a function absorber:
    CIA terms inside the flux kernel (temperature half of the interpolation per band sample, the rest per point) vs k_cia, vs the oracle"""
    nu = np.linspace(1.0, 2200.0, nnu)
    P = cs.pressuregrid(50.0, 1e5, 19)
    T = W.earth_temperature(P)
    members = (cs.DirectGas(lines("CO2"), 0.9, nu), cs.DirectGas(lines("CH4"), 0.02, nu), cs.CIATables(W.fixture("CO2-CO2_2018.cia")),
               cs.CIATables(W.fixture("CO2-CH4_2018.cia")), cs.GrayGas(2e-28, nu), lambda v, T_, P_: 1e-28 * (P_ / 1e5) * np.asarray(v) / 1000.0)
    core = cs.Discretized(5, 2)
    a_ctx, b_ctx = _ctx(cs, key15), _ctx(cs, 1)
    col, a = _run(cs, a_ctx, P, T, members, core)
    _, b = _run(cs, b_ctx, P, T, members, core)
    assert a["launches"] < b["launches"]
    assert relerr(a["tau"], b["tau"]) < 1e-12
    sm = max(b["Mup"].max(), b["Mdn"].max())
    assert np.max(np.abs(a["Mup"] - b["Mup"])) < 1e-12 * sm and np.max(np.abs(a["Mdn"] - b["Mdn"])) < 1e-12 * sm
    d = [cs.readcia(W.fixture(f)) for f in ("CO2-CO2_2018.cia", "CO2-CH4_2018.cia")]
    extra = np.zeros((col.K, col.nnu))
    for k in range(col.K):
        for ci, x in enumerate(col.U.cia):
            extra[k] += O.cia_sigma(d[ci], nu, col.Tk[k], col.Pk[k], col.cia_P1[ci, k], col.cia_P2[ci, k])
        extra[k] += col.sigma_extra[k]
    r = O.fluxes_discretized(nu, P, 9.8, 2, col.Tn, col.mun, col.Tlev, [g.sl for g in col.gases], ["voigt"] * 2, [25.0] * 2, col.conc,
                             sigma_gray=col.sigma_gray, sigma_extra=extra)
    assert relerr(a["tau"], r["tau"]) < 1e-10
    assert np.max(np.abs(a["Fup"] - r["Fup"])) < 1e-10 * r["Fup"].max()
    a_ctx.close(); b_ctx.close()


def test_long_grid_default_is_chunk_form(cs, lines):
    """>= 4096 tiles: the chunked form is the default; against the separate kernels (key 15 = 1) on a 300 000-point grid"""
    nu = np.linspace(600.0, 780.0, 300000)
    P = cs.pressuregrid(100.0, 1e5, 7)
    T = W.earth_temperature(P)
    gases = (cs.DirectGas(lines("CO2"), 400e-6, nu),)
    a_ctx, b_ctx = _ctx(cs, 0), _ctx(cs, 1)
    _, a = _run(cs, a_ctx, P, T, gases, cs.Discretized(5, 2))
    _, b = _run(cs, b_ctx, P, T, gases, cs.Discretized(5, 2))
    assert a["launches"] < b["launches"]
    assert np.array_equal(a["tau"], b["tau"]) and np.array_equal(a["Mup"], b["Mup"]) and np.array_equal(a["Mdn"], b["Mdn"])
    assert np.max(np.abs(a["Fup"] - b["Fup"])) < 1e-14 * np.max(b["Fup"])
    a_ctx.close(); b_ctx.close()


@pytest.mark.parametrize("nnu,np_", [(12500, 61), (9001, 22)])
def test_piece_tables_sixteen_lanes_equal_one_thread(cs, lines, nnu, np_):
    """k_mxzones16 (sixteen lanes per (interval | tile, state group): reductions and shuffled ranks) builds the same matrix-core piece
    tables as the one-thread-per-item kernel (cs_set_tuning key 15 | 16): every output of the column bitwise equal, K not a multiple of 16"""
    nu = np.linspace(500.0, 900.0, nnu)
    P = cs.pressuregrid(1.0, 1e5, np_)
    T = W.earth_temperature(P)
    gases = (cs.DirectGas(W.lines("synthetic", "H2O"), W.fC_h2o, nu), cs.DirectGas(W.lines("synthetic", "CO2"), 400e-6, nu))
    a_ctx, b_ctx = _ctx(cs, 0), _ctx(cs, 16)
    for c_ in (a_ctx, b_ctx):
        c_.set_matrix_cores(2)
    _, a = _run(cs, a_ctx, P, T, gases, cs.Discretized(5, 2))
    _, b = _run(cs, b_ctx, P, T, gases, cs.Discretized(5, 2))
    for k in ("tau", "Mup", "Mdn", "Fup", "Fdn"):
        assert np.array_equal(a[k], b[k]), k
    a_ctx.close(); b_ctx.close()


@pytest.mark.parametrize("nlay,ns,fS,fa", [(60, 5, 0.4, 0.0), (40, 4, 0.0, 0.2), (23, 6, 0.0, 0.0), (75, 5, 0.0, 0.0)])
def test_scan_transmissivities_in_registers_equal_recomputed(cs, lines, nlay, ns, fS, fa):
    """k_flux_scan<NS, 5> keeps the chunk's exp(-tau m_k), 1/tau and the beam's attenuation in registers between the sweeps and passes;
    k_flux_scan<NS, 0> (key 15 | 2048; and chunks of more than five layers: 75 layers over 12 waves) forms them again -- same
    operations on the same operands: every output bitwise equal.  60 layers = 12 waves x 5, 40 = 8 x 5, 23 = 8 x 3 with a short tail"""
    nu = np.linspace(600.0, 700.0, 5000)
    P = cs.pressuregrid(5.0, 1e5, nlay + 1)
    T = W.earth_temperature(P)
    gases = (cs.DirectGas(lines("CO2"), 400e-6, nu),)
    core = cs.Discretized(ns, 2)
    col, a = _run(cs, _ctx(cs, 0), P, T, gases, core, fS, fa)
    _, b = _run(cs, _ctx(cs, 2048), P, T, gases, core, fS, fa)
    assert col.info()["flux_form"] == 3
    for k in ("tau", "Mup", "Mdn", "Fup", "Fdn"):
        assert np.array_equal(a[k], b[k]), k


def test_wave_priority_changes_no_result(cs, lines):
    """cs_set_tuning key 16: s_setprio for the near-line stream's kernels is scheduling only"""
    nu = np.linspace(600.0, 760.0, 40000)
    P = cs.pressuregrid(5.0, 1e5, 21)
    T = W.earth_temperature(P)
    gases = (cs.DirectGas(lines("H2O"), W.fC_h2o, nu), cs.DirectGas(lines("CO2"), 400e-6, nu))
    core = cs.Discretized(5, 2)
    res = []
    for v in (1, 2, 0):
        ctx = cs.Context(0)
        ctx.set_tuning(16, v)
        res.append(_run(cs, ctx, P, T, gases, core)[1])
    for r in res[1:]:
        for k in ("tau", "Mup", "Mdn", "Fup", "Fdn"):
            assert np.array_equal(r[k], res[0][k]), k


def test_scan_form_up_to_1024_tiles(cs, lines):
    """the scan form is the default up to 1024 tiles (a half of the bench column): 50 000 points = 782 tiles, k_freduce adds the block
    partials there (more than 512 blocks); against the separate kernels at the scan form's tolerances"""
    nu = np.linspace(560.0, 800.0, 50000)
    P = cs.pressuregrid(5.0, 1e5, 31)
    T = W.earth_temperature(P)
    gases = (cs.DirectGas(lines("H2O"), W.fC_h2o, nu), cs.DirectGas(lines("CO2"), 400e-6, nu))
    core = cs.Discretized(5, 2)
    col, a = _run(cs, _ctx(cs, 0), P, T, gases, core, 0.2, 0.1)
    _, b = _run(cs, _ctx(cs, 1), P, T, gases, core, 0.2, 0.1)
    assert col.info()["flux_form"] == 3 and a["launches"] < b["launches"]
    assert relerr(a["tau"], b["tau"]) < 5e-13
    sm_ = max(b["Mup"].max(), b["Mdn"].max())
    amp = source_rounding_bound(cs, nu, col.Tlev, b["tau"])
    assert np.max(np.abs(a["Mup"] - b["Mup"])) < 1e-13 * sm_ + amp and np.max(np.abs(a["Mdn"] - b["Mdn"])) < 1e-13 * sm_ + amp
    assert np.max(np.abs(a["Fup"] - b["Fup"])) < 1e-12 * np.max(b["Fup"])


def test_near_tiers_in_one_launch_equal_two(cs, lines):
    """k_voigt_near_both (default where a wave takes one tile) against k_voigt_near<0> + <1> (cs_set_tuning key 16 | 4): (sigma + tier 0)
    + tier 1 either way -- every output bitwise equal, one launch fewer"""
    nu = np.linspace(600.0, 760.0, 20011)
    P = cs.pressuregrid(1.0, 1e5, 25)
    T = W.earth_temperature(P)
    gases = (cs.DirectGas(lines("H2O"), W.fC_h2o, nu), cs.DirectGas(lines("CO2"), 400e-6, nu))
    core = cs.Discretized(5, 2)
    res = []
    for v in (0, 4):
        ctx = cs.Context(0)
        ctx.set_tuning(16, v)
        res.append(_run(cs, ctx, P, T, gases, core)[1])
    assert res[0]["launches"] == res[1]["launches"] - 1
    for k in ("tau", "Mup", "Mdn", "Fup", "Fdn"):
        assert np.array_equal(res[0][k], res[1][k]), k


def test_in_kernel_band_sum_vs_freduce(cs, lines):
    """The last-block band sum of k_flux_scan (agent-scope relaxed atomics + a ticket, on gfx950 only: DESIGN.md section 4) against the same
    kernel with its block partials added by k_freduce's own launch (cs_set_tuning key 15 | 4): everything per wavenumber bitwise equal, the
    band fluxes equal to rounding of the different association (two stages of 16 and <= 32 terms against one fixed-order sum), and each
    form repeatable bit for bit over many runs (a partial arriving late would show as a changed last digit)."""
    nu = np.linspace(580.0, 780.0, 12000)         # 188 tiles: several groups of 16 blocks, a partial last group
    P = cs.pressuregrid(5.0, 1e5, 21)
    T = W.earth_temperature(P)
    gases = (cs.DirectGas(lines("H2O"), W.fC_h2o, nu), cs.DirectGas(lines("CO2"), 400e-6, nu))
    core = cs.Discretized(5, 2)
    res = {}
    for key in (0, 4):
        ctx = _ctx(cs, key)
        col, r = _run(cs, ctx, P, T, gases, core, 0.3, 0.1)
        assert col.info()["flux_form"] == 3
        for _ in range(20):
            col.run()
            F = col.fetch()
            assert np.array_equal(F[0], r["Fup"]) and np.array_equal(F[1], r["Fdn"])
        res[key] = r
        ctx.close()
    assert res[0]["launches"] == res[4]["launches"] - 1
    for k in ("tau", "Mup", "Mdn"):
        assert np.array_equal(res[0][k], res[4][k]), k
    fm = np.max(res[4]["Fup"])
    assert np.max(np.abs(res[0]["Fup"] - res[4]["Fup"])) < 1e-13 * fm and np.max(np.abs(res[0]["Fdn"] - res[4]["Fdn"])) < 1e-13 * fm
