"""GPU tests of the entry points the reference-side binding needs to carry every AbstractAbsorber member:

  * cs_fluxes_discretized_members -- ONE host-pointer call for a column with baked tables, CIA pairs or an accelerated absorber among
    its members (what julia/ClearSkyHIP.jl's monochromaticfluxes! calls), against the resident-column calls it wraps (bitwise) and,
    through them, the oracle (tests/test_gpu_absorbers.py, test_gpu_tables.py);
  * cs_table_upload  -- a host-held opacity table (the reference's own baked Gas) against the same table baked on the device;
  * cs_accel_upload / cs_accel_fetch -- a host-held AcceleratedAbsorber against the device-evaluated one;
  * the merged-table cache cannot free a table a column still runs on; hipGraph replay keeps the near-line plane's bookkeeping.
"""
import ctypes as C

import numpy as np
import pytest

import workloads as W
from conftest import relerr

pytestmark = pytest.mark.gpu


def _fetch_all(col):
    tau = np.zeros((col.nl, col.nnu), order="F")
    Mu = np.zeros((col.np, col.nnu), order="F")
    Md = np.zeros((col.np, col.nnu), order="F")
    Fup, Fdn = col.fetch(tau, Mu, Md)
    return tau, Mu, Md, Fup, Fdn


def test_members_entry_point_equals_resident_calls(cs, lines):
    """baked Gas + DirectGas + CIA pair + gray term + a function absorber through cs_fluxes_discretized_members: bitwise what
    cs_column_setup / _set_tables / _set_cia / _run / _fetch give, first call (setup) and repeated call (resident column, new state)"""
    ctx = cs.Context(0)
    nu = np.linspace(600.0, 800.0, 4001)
    P = cs.pressuregrid(50.0, 1e5, 15)
    T = W.earth_temperature(P)
    Om = cs.AtmosphericDomain((150, 330), 6, (10, 1.1e5), 8)
    baked = cs.Gas(lines("CO2"), 0.9, nu, Om, ctx=ctx)
    direct = cs.DirectGas(lines("H2O"), W.fC_h2o, nu)
    ch4 = cs.DirectGas(lines("CH4"), 1e-3, nu)
    x = cs.CIATables(W.fixture("CO2-CH4_2018.cia"))
    gray = cs.GrayGas(1e-27, nu)
    fun = lambda v, T_, P_: 2e-28 * (P_ / 1e5) * np.asarray(v) / 700.0
    members = (baked, direct, ch4, x, gray, fun)
    for T_now in (T, T + 3.0):
        F = cs.radiate(P, 9.8, T_now, 0.040, 0.0, 0.2, *members, core=cs.Discretized(5, 3), ctx=ctx)      # -> _members entry point
        col = cs.Column(P, 9.8, T_now, 0.040, 0.0, 0.2, *members, core=cs.Discretized(5, 3), ctx=ctx)   # resident calls
        col.run()
        tau, Mu, Md, Fup, Fdn = _fetch_all(col)
        assert np.array_equal(F.tau, tau) and np.array_equal(F.Mup, Mu) and np.array_equal(F.Mdn, Md)
        assert np.array_equal(F.Fup, Fup) and np.array_equal(F.Fdn, Fdn)
        assert np.all(np.isfinite(Fup)) and Fup[0] > 0
    # repeated call on the SAME line-up without an intervening Column: the resident path of the entry point (no setup) -- same answer
    F1 = cs.radiate(P, 9.8, T, 0.040, 0.0, 0.2, *members, core=cs.Discretized(5, 3), ctx=ctx)
    F2 = cs.radiate(P, 9.8, T + 3.0, 0.040, 0.0, 0.2, *members, core=cs.Discretized(5, 3), ctx=ctx)
    F3 = cs.radiate(P, 9.8, T, 0.040, 0.0, 0.2, *members, core=cs.Discretized(5, 3), ctx=ctx)
    assert np.array_equal(F1.Fup, F3.Fup) and np.array_equal(F1.tau, F3.tau)
    assert not np.array_equal(F1.Fup, F2.Fup)
    ctx.close()


def test_members_entry_point_argument_checks(cs, lines):
    ctx = cs.Context(0)
    L = cs.lib()
    nu = np.linspace(600.0, 700.0, 512)
    P = cs.pressuregrid(50.0, 1e5, 5)
    T = W.earth_temperature(P)
    col = cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, cs.GrayGas(1e-26, nu), ctx=ctx, _setup=False)
    dp = cs.dptr
    Fu, Fd = np.zeros(col.np), np.zeros(col.np)
    one = np.zeros(1, dtype=np.int32)
    ip = one.ctypes.data_as(C.POINTER(C.c_int))
    args = (ctx.handle, col.nnu, dp(col.nu), col.np, dp(col.P), col.g, 2, dp(col.Tn.ravel(order="F").copy()), dp(col.mun.ravel(order="F").copy()),
            dp(col.Tlev), 0, None, None, None, None)
    tail = (1e-26, None, None, None, 0.841, 5, None, None, None, dp(Fu), dp(Fd))
    # an accelerated absorber stands for all absorbers: not beside tables (absorbers.jl:216)
    assert L.cs_fluxes_discretized_members(*args, 1, ip, dp(np.zeros(col.K)), 0, None, None, None, None, 0, *tail) == -1
    # empty table slot, empty accelerated-absorber slot
    assert L.cs_fluxes_discretized_members(*args, 1, ip, dp(np.zeros(col.K)), 0, None, None, None, None, -1, *tail) == -1
    assert b"empty" in L.cs_last_error()
    assert L.cs_fluxes_discretized_members(*args, 0, None, None, 0, None, None, None, None, 2, *tail) == -1
    # and with nothing beyond the gray term it is cs_fluxes_discretized
    assert L.cs_fluxes_discretized_members(*args, 0, None, None, 0, None, None, None, None, -1, *tail) == 0
    Fu2, Fd2 = np.zeros(col.np), np.zeros(col.np)
    assert L.cs_fluxes_discretized(*args, 1e-26, None, None, None, 0.841, 5, None, None, None, dp(Fu2), dp(Fd2)) == 0
    assert np.array_equal(Fu, Fu2) and np.array_equal(Fd, Fd2)
    ctx.close()


def test_table_upload_equals_device_bake(cs, lines):
    """cs_bake hands the tables back (lnsigma_out); uploaded into another slot they must evaluate bitwise alike -- point by point
    (cs_table_eval = rawsigma) and inside a column -- which is how a Gas object baked by the reference itself travels"""
    ctx = cs.Context(0)
    nu = np.linspace(640.0, 700.0, 1500)
    Om = cs.AtmosphericDomain((180, 320), 5, (100, 1.05e5), 7)
    g = cs.Gas(lines("CO2"), 400e-6, nu, Om, ctx=ctx, keep_host_tables=True)
    lns = np.ascontiguousarray(g.lnsigma.ravel(order="F"))          # [nnu, nT, nP] column-major
    slot2 = 9
    cs.check(cs.lib().cs_table_upload(ctx.handle, slot2, len(nu), cs.dptr(nu), Om.nT, cs.dptr(cs.as_f64(Om.T)), Om.nP, cs.dptr(cs.as_f64(Om.P)),
                                      cs.dptr(lns)))
    a, b = np.zeros(len(nu)), np.zeros(len(nu))
    for T, Pv in ((250.0, 2e4), (300.0, 9e4), (181.0, 150.0)):
        cs.check(cs.lib().cs_table_eval(ctx.handle, g.slot, T, Pv, 0, len(nu), cs.dptr(a)))
        cs.check(cs.lib().cs_table_eval(ctx.handle, slot2, T, Pv, 0, len(nu), cs.dptr(b)))
        assert np.array_equal(a, b) and np.all(a > 0)
    # refusals: non-finite knots, unsorted temperatures
    bad = lns.copy(); bad[5] = -np.inf
    assert cs.lib().cs_table_upload(ctx.handle, slot2 + 1, len(nu), cs.dptr(nu), Om.nT, cs.dptr(cs.as_f64(Om.T)), Om.nP, cs.dptr(cs.as_f64(Om.P)), cs.dptr(bad)) == -1
    Tbad = cs.as_f64(Om.T)[::-1].copy()
    assert cs.lib().cs_table_upload(ctx.handle, slot2 + 1, len(nu), cs.dptr(nu), Om.nT, cs.dptr(Tbad), Om.nP, cs.dptr(cs.as_f64(Om.P)), cs.dptr(lns)) == -4
    ctx.close()


def test_accel_upload_fetch_roundtrip(cs, lines):
    """knots evaluated on the device (cs_accel_store), fetched (cs_accel_fetch), uploaded into another slot (cs_accel_upload): both slots
    give the same cross-sections at any pressure and the same column -- how an AcceleratedAbsorber held by the host (RCM's field) travels"""
    ctx = cs.Context(0)
    nu = np.linspace(600.0, 760.0, 3001)
    Pe = cs.pressuregrid(20.0, 1e5, 9)
    Te = W.earth_temperature(Pe)
    A = cs.AcceleratedAbsorber(Te, Pe, cs.DirectGas(lines("CO2"), 400e-6, nu), cs.DirectGas(lines("H2O"), W.fC_h2o, nu), ctx=ctx)
    nk = len(Pe)
    kn = np.zeros((nk, len(nu)))                                     # [nnu, nk] column-major = [nk][nnu]
    cs.check(cs.lib().cs_accel_fetch(ctx.handle, A.slot, len(nu), nk, cs.dptr(kn)))
    assert np.all(np.isfinite(kn)) and np.all(kn >= np.log(np.finfo(float).tiny))
    assert cs.lib().cs_accel_fetch(ctx.handle, A.slot, len(nu), nk + 1, cs.dptr(np.zeros((nk + 1, len(nu))))) == -6      # sized for other knots
    other = (A.slot + 1) % 4
    cs.check(cs.lib().cs_accel_upload(ctx.handle, other, len(nu), cs.dptr(nu), nk, cs.dptr(cs.as_f64(A.P)), cs.dptr(kn)))
    a, b = np.zeros(len(nu)), np.zeros(len(nu))
    for Pv in (25.0, 3.3e3, 9.9e4, 2e5):
        cs.check(cs.lib().cs_accel_eval(ctx.handle, A.slot, Pv, 0, len(nu), cs.dptr(a)))
        cs.check(cs.lib().cs_accel_eval(ctx.handle, other, Pv, 0, len(nu), cs.dptr(b)))
        assert np.array_equal(a, b)
    Pr = cs.pressuregrid(20.0, 1e5, 17)
    Tr = W.earth_temperature(Pr)
    F1 = cs.radiate(Pr, 9.8, Tr, 0.029, 0.0, 0.0, A, ctx=ctx)          # cs_fluxes_discretized_members(accel_slot = A.slot)
    slot_was = A.slot
    A.slot = other
    F2 = cs.radiate(Pr, 9.8, Tr, 0.029, 0.0, 0.0, A, ctx=ctx)
    A.slot = slot_was
    assert np.array_equal(F1.Fup, F2.Fup) and np.array_equal(F1.tau, F2.tau)
    cs.check(cs.lib().cs_accel_clear(ctx.handle, other))
    ctx.close()


def test_merged_cache_never_frees_a_table_in_use(cs, O, lines):
    """the context keeps a bounded number of merged tables; a column shares ownership of the ones its launch groups run on.
    Fill the cache with other line-ups, then set up ONE column with two merged groups (two cut-offs x two gases) whose first group is
    the oldest cache entry, run it, and compare with per-gas launch sets"""
    nu = np.linspace(1250.0, 1330.0, 3000)
    P = cs.pressuregrid(5.0, 1e5, 9)
    T = W.earth_temperature(P)
    ctx = cs.Context(0)
    h2o, co2, ch4 = lines("H2O"), lines("CO2"), lines("CH4")
    def g(sl, c, cut): return cs.DirectGas(sl, c, nu, dnu_cut=cut)
    lineups = [(g(h2o, 1e-3, 25.0), g(co2, 4e-4, 25.0)),                    # the oldest entry: group 0 of the column below
               (g(h2o, 1e-3, 10.0), g(ch4, 2e-6, 10.0)), (g(co2, 4e-4, 12.0), g(ch4, 2e-6, 12.0)),
               (g(h2o, 1e-3, 14.0), g(co2, 4e-4, 14.0), g(ch4, 2e-6, 14.0)), (g(h2o, 1e-3, 16.0), g(ch4, 2e-6, 16.0)),
               (g(co2, 4e-4, 18.0), g(h2o, 1e-3, 18.0)), (g(ch4, 2e-6, 20.0), g(co2, 4e-4, 20.0)),
               (g(ch4, 2e-6, 22.0), g(h2o, 1e-3, 22.0)), (g(co2, 4e-4, 23.0), g(ch4, 2e-6, 23.0))]
    for lu in lineups:
        cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, *lu, ctx=ctx).run()
    two = (g(h2o, 1e-3, 25.0), g(co2, 4e-4, 25.0), g(h2o, 5e-4, 9.0), g(ch4, 2e-6, 9.0))
    col = cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, *two, ctx=ctx)
    col.run()
    assert col.info()["groups"] == 2
    col.update(T + 1.0)      # (state tables are rebuilt from the groups' tables: a freed one would be read here)
    col.update(T)
    col.run()
    tau_m, Mu_m, Md_m, Fup_m, Fdn_m = _fetch_all(col)
    ctx2 = cs.Context(0)
    ctx2.set_merge(False)
    col2 = cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, *two, ctx=ctx2)
    col2.run()
    tau_s, _, _, Fup_s, Fdn_s = _fetch_all(col2)
    assert relerr(tau_m, tau_s) < 5e-13
    assert np.max(np.abs(Fup_m - Fup_s)) < 1e-12 * np.max(Fup_s)
    ctx.close(); ctx2.close()


def test_graph_replay_keeps_near_plane_bookkeeping(cs, lines):
    """with the step replayed as a hipGraph (cs_set_tuning key 4) and the near-line kernels on their side stream (key 7 = 2),
    cs_column_sigma_fetch after a REPLAY must fold the near-line plane in again, like after an eager run"""
    nu = np.linspace(640.0, 700.0, 6000)
    P = cs.pressuregrid(5.0, 1e5, 9)
    T = W.earth_temperature(P)
    gases = (cs.DirectGas(lines("CO2"), 400e-6, nu), cs.DirectGas(lines("H2O"), W.fC_h2o, nu))
    ref_ctx = cs.Context(0)
    ref_ctx.set_tuning(7, 0)
    ref = cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, *gases, ctx=ref_ctx)
    ref.run()
    sig_ref = ref.sigma_nodes()
    ctx = cs.Context(0)
    ctx.set_tuning(4, 1)
    ctx.set_tuning(7, 2)
    col = cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, *gases, ctx=ctx)
    sigs = []
    for _ in range(4):        # eager, capture, replay, replay -- a sigma fetch (which folds the plane in) after each
        col.run()
        sigs.append(col.sigma_nodes())
        assert col.info()["launches"] > 0
    for s_ in sigs:
        assert relerr(s_, sig_ref, floor=1e-300) < 5e-13
    assert np.array_equal(sigs[2], sigs[3])
    ref_ctx.close(); ctx.close()
