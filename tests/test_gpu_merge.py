"""One launch set per column: the Voigt (Lorentz) gases of a column that share a cut-off are merged into one sorted line table
(cs_set_merge, on by default).  sigma_total = sum_g C_g sigma_g (absorbers.jl:84-95) must come out the same -- to rounding: only
the order of the sum over lines changes -- as with one launch set per gas, and as the oracle's per-gas sums.

Tolerances: merged vs per-gas on the device 5e-13 relative on sigma and tau (observed ~1e-14), fluxes 1e-12 of the column maximum;
vs the oracle 1e-11 like every other column test.
"""
import numpy as np
import pytest

import workloads as W
from conftest import relerr, source_rounding_bound

pytestmark = pytest.mark.gpu


DEFAULT_TUNE = {0: 0, 1: 1, 2: 2, 5: 1, 7: 1}    # the library's defaults (cs_api.hip: cs_ctx::tune)


def _column(cs, ctx, absorbers, P, T, nu_range=None, core=None, **kw):
    return cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, *absorbers, core=core or cs.Discretized(5, 2), ctx=ctx, nu_range=nu_range, **kw)


def _results(col):
    col.run()
    tau = np.zeros((col.nl, col.nnu), order="F")
    Mu = np.zeros((col.np, col.nnu), order="F")
    Md = np.zeros((col.np, col.nnu), order="F")
    Fup, Fdn = col.fetch(tau, Mu, Md)
    return dict(sigma=col.sigma_nodes(), tau=tau, Mup=Mu, Mdn=Md, Fup=Fup, Fdn=Fdn, nu=col.nu, Tlev=col.Tlev)


def _oracle(O, col, cs):
    return O.fluxes_discretized(col.nu, col.P, col.g, col.core.nlobatto, col.Tn, col.mun, col.Tlev, [g.sl for g in col.gases],
                                [g.shape for g in col.gases], [g.dnu_cut for g in col.gases], col.conc, sigma_gray=col.sigma_gray,
                                theta_s=col.theta_s, nstream=col.core.nstream, want_sigma=True)


def _close(a, b, tol_rel, tol_flux):
    import clearsky_jl_amd as cs
    assert relerr(a["sigma"], b["sigma"], floor=1e-300) < tol_rel
    assert relerr(a["tau"], b["tau"]) < tol_rel
    sm = np.max(b["Mup"])
    amp = source_rounding_bound(cs, b["nu"], b["Tlev"], b["tau"])     # (last-bit differences of tau through (1 - t)/tau, see conftest)
    for k in ("Mup", "Mdn"):
        assert np.max(np.abs(a[k] - b[k])) < tol_flux * sm + amp, k
    for k in ("Fup", "Fdn"):
        assert np.max(np.abs(a[k] - b[k])) < tol_flux * np.max(b["Fup"]), k


@pytest.mark.parametrize("matrix", [1, 2, 0])
def test_merged_vs_per_gas_vs_oracle(cs, O, lines, matrix):
    """H2O + CO2 + CH4 fixtures on a 6000-point window: one group of three gases against three launch sets and against the oracle"""
    nu = np.linspace(1200.0, 1350.0, 6000)
    P = cs.pressuregrid(5.0, 1e5, 13)
    T = W.earth_temperature(P)
    gases = [cs.DirectGas(lines("H2O"), W.fC_h2o, nu), cs.DirectGas(lines("CO2"), 400e-6, nu), cs.DirectGas(lines("CH4"), 1.8e-6, nu)]
    res = {}
    for merge in (True, False):
        ctx = cs.Context(0)
        ctx.set_merge(merge)
        ctx.set_matrix_cores(matrix)
        col = _column(cs, ctx, gases, P, T)
        res[merge] = _results(col)
        info = col.info()
        assert info["groups"] == (1 if merge else 3) and info["merge"] == int(merge)
        assert info["lines"] == sum(len(g.sl.nu) for g in gases)
        if merge:
            assert info["max_members"] == 3
            ref = _oracle(O, col, cs)
        launches = info["launches"]
        res[("launches", merge)] = launches
        ctx.close()
    assert res[("launches", True)] < res[("launches", False)]
    _close(res[True], res[False], 5e-13, 1e-12)
    r = res[True]
    assert relerr(r["sigma"], ref["sigma"], floor=1e-300) < 1e-11
    assert relerr(r["tau"], ref["tau"]) < 1e-11
    sm = np.max(ref["Mup"])
    assert np.max(np.abs(r["Mup"] - ref["Mup"])) < 1e-11 * sm and np.max(np.abs(r["Mdn"] - ref["Mdn"])) < 1e-11 * sm
    assert np.max(np.abs(r["Fup"] - ref["Fup"])) < 1e-11 * np.max(ref["Fup"])


def test_what_merges_and_what_does_not(cs, O, lines):
    """same shape and cut-off merge; another shape, another cut-off or the same table twice stay apart -- same results either way"""
    nu = np.linspace(600.0, 760.0, 3000)
    P = cs.pressuregrid(10.0, 1e5, 9)
    T = W.earth_temperature(P)
    co2, h2o = lines("CO2"), lines("H2O")
    cases = [
        ([cs.DirectGas(co2, 400e-6, nu), cs.DirectGas(h2o, W.fC_h2o, nu)], 1),
        ([cs.DirectGas(co2, 400e-6, nu), cs.DirectGas(h2o, W.fC_h2o, nu, shape="lorentz")], 2),
        ([cs.DirectGas(co2, 400e-6, nu), cs.DirectGas(h2o, W.fC_h2o, nu, dnu_cut=10.0)], 2),
        ([cs.DirectGas(co2, 300e-6, nu), cs.DirectGas(co2, 100e-6, nu)], 2),                      # one slot named twice
        ([cs.DirectGas(co2, 400e-6, nu, shape="lorentz"), cs.DirectGas(h2o, W.fC_h2o, nu, shape="lorentz")], 1),
        ([cs.DirectGas(co2, 400e-6, nu, shape="doppler"), cs.DirectGas(h2o, W.fC_h2o, nu, shape="doppler")], 2),
        ([cs.DirectGas(co2, 400e-6, nu), cs.DirectGas(h2o, W.fC_h2o, nu), cs.GrayGas(1e-27, nu)], 1),
    ]
    ctx = cs.Context(0)
    for absorbers, ngroups in cases:
        col = _column(cs, ctx, absorbers, P, T)
        got = _results(col)
        assert col.info()["groups"] == ngroups
        ref = _oracle(O, col, cs)
        tol = 2e-9 if absorbers[0].shape == "doppler" else 1e-11
        assert relerr(got["tau"], ref["tau"]) < tol
        assert np.max(np.abs(got["Fup"] - ref["Fup"])) < tol * np.max(ref["Fup"])
    ctx.close()


def test_merged_update_and_reupload(cs, O, lines):
    """update! on a merged column re-evaluates the members' concentrations; re-uploading a member's table invalidates the column"""
    nu = np.linspace(1500.0, 1600.0, 2500)
    P = cs.pressuregrid(10.0, 1e5, 11)
    T = W.earth_temperature(P)
    ctx = cs.Context(0)
    gases = [cs.DirectGas(lines("H2O"), W.fC_h2o, nu), cs.DirectGas(lines("CO2"), 400e-6, nu)]
    col = _column(cs, ctx, gases, P, T)
    _results(col)
    col.update(T + 7.0)          # H2O concentration follows the temperature (psat)
    got = _results(col)
    ref = _oracle(O, col, cs)
    assert relerr(got["tau"], ref["tau"]) < 1e-11
    assert np.max(np.abs(got["Fup"] - ref["Fup"])) < 1e-11 * np.max(ref["Fup"])
    # batch of perturbed profiles through the merged group
    Ts = [T + d for d in (-3.0, 0.0, 4.0)]
    Fup, Fdn = col.run_batch(Ts)
    for b, Tb in enumerate(Ts):
        c2 = _column(cs, cs.Context(0), gases, P, Tb, _setup=False)
        r2 = _oracle(O, c2, cs)
        assert np.max(np.abs(Fup[b] - r2["Fup"])) < 1e-11 * np.max(r2["Fup"])
        assert np.max(np.abs(Fdn[b] - r2["Fdn"])) < 1e-11 * np.max(r2["Fup"])
    # a new table in a member's slot: the resident column must refuse to run on its stale windows
    slot = ctx.slot_of(gases[1].sl)
    sl = gases[1].sl
    import ctypes as C
    from clearsky_jl_amd._lib import lib, dptr, as_f64
    arrs = [as_f64(a[: len(sl.nu) // 2]) for a in (sl.nu, sl.S, sl.gamma_a, sl.gamma_s, sl.Epp, sl.na, sl.mu)]
    iso = np.ascontiguousarray(sl.I[: len(sl.nu) // 2], dtype=np.int16)
    ncheb = np.ascontiguousarray(sl.ncheb, dtype=np.int32)
    cheb = as_f64(sl.cheb)
    rc = lib().cs_gas_upload(ctx.handle, slot, len(arrs[0]), *[dptr(a) for a in arrs], iso.ctypes.data_as(C.POINTER(C.c_int16)), len(ncheb),
                             ncheb.ctypes.data_as(C.POINTER(C.c_int32)), dptr(cheb))
    assert rc == 0
    rc = lib().cs_column_run(ctx.handle, None)
    assert rc == -6 and b"re-uploaded" in lib().cs_last_error()
    ctx.close()


def test_merged_shards_add_up(cs, O, lines):
    """nu-shards of a merged column (global trapezoid weights) add up to the whole column"""
    nu = np.linspace(550.0, 800.0, 5000)
    P = cs.pressuregrid(10.0, 1e5, 9)
    T = W.earth_temperature(P)
    gases = [cs.DirectGas(lines("H2O"), W.fC_h2o, nu), cs.DirectGas(lines("CO2"), 400e-6, nu)]
    ctx = cs.Context(0)
    whole = _results(_column(cs, ctx, gases, P, T))
    Fup = np.zeros_like(whole["Fup"])
    for r in W.balanced_ranges(nu, gases, 3):
        col = _column(cs, ctx, gases, P, T, nu_range=r)
        part = _results(col)
        Fup += part["Fup"]
        assert relerr(part["tau"], whole["tau"][:, r[0]:r[1]]) < 5e-13
    assert np.max(np.abs(Fup - whole["Fup"])) < 1e-12 * np.max(whole["Fup"])
    ctx.close()


@pytest.mark.parametrize("tune", [{0: 1}, {1: 0}, {2: 0}, {2: 1}, {0: 1, 2: 0}, {3: 20}, {4: 1}, {5: 0}, {6: 2}, {7: 0}, {7: 2}, {2: 0, 7: 2}, {8: 16}, {11: 1}, {12: 1}, {12: 2}, {13: 1}, {13: 2}, {14: 1}])
def test_tuning_switches_same_results(cs, O, lines, tune):
    """cs_set_tuning: interpolated wings applied inside k_voigt_edge_mx (0), matrix-core kernels on short grids through their
    four-waves-per-item variants (1), node sums on a side stream (2), interpolation margin (3), the step as one hipGraph (4) -- none
    of them may change a result beyond rounding.  Every case is compared with the library's defaults."""
    nu = np.linspace(580.0, 780.0, 20000)
    P = cs.pressuregrid(10.0, 1e5, 21)
    T = W.earth_temperature(P)
    # (the seeded synthetic tables of the bench workload: dense enough -- 40 lines per cm^-1 together -- for every matrix-core piece)
    gases = [cs.DirectGas(W.lines("synthetic", "H2O"), W.fC_h2o, nu), cs.DirectGas(W.lines("synthetic", "CO2"), 400e-6, nu)]
    res = []
    for t in ({}, tune):
        for mc in ((2,) if 1 not in tune else (1,)):     # key 1 only matters where the grid-length rule would say "vector unit"
            ctx = cs.Context(0)
            ctx.set_matrix_cores(mc)
            for k, v in t.items():
                ctx.set_tuning(k, v)
            col = _column(cs, ctx, gases, P, T)
            if 4 in t:       # graph replay: eager run, capturing run, two replays -- the last one after update! with a new profile and back
                for _ in range(3):
                    col.run()
                col.update(T + 5.0)
                col.run()
                col.update(T)
            res.append(_results(col))
            res[-1]["work"] = col.work()
            res[-1]["launches"] = col.info()["launches"]
            ctx.close()
    _close(res[1], res[0], 5e-13, 1e-12)
    if 0 in tune and DEFAULT_TUNE[0] != tune[0]:
        assert res[0]["launches"] - res[1]["launches"] >= 1     # the wings' own launch (and, with four or more levels, the cascade before it), or not
    if 1 in tune and DEFAULT_TUNE[1] != tune[1]:
        on, off = (res[1], res[0]) if tune[1] else (res[0], res[1])
        assert on["work"]["direct_evals_matrix"] > 0 and off["work"]["direct_evals_matrix"] == 0
        assert on["work"]["node_evals_matrix"] > 0 and off["work"]["node_evals_matrix"] == 0
    ref = _oracle(O, _column(cs, cs.Context(0), gases, P, T, _setup=False), cs)
    assert relerr(res[1]["tau"], ref["tau"]) < 1e-11
    assert np.max(np.abs(res[1]["Fup"] - ref["Fup"])) < 1e-11 * np.max(ref["Fup"])
