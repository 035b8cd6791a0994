"""GPU tests of the drop-in boundary itself and of the rows the round-1 suite only checked against the device:

  * cs_fluxes_discretized -- the ONE symbol the Julia method monochromaticfluxes!(..., core::HIPDiscretized, ...) binds
    (julia/ClearSkyHIP.jl:142-188) -- called through raw ctypes with the arrays laid out as that ccall passes them, against
    the committed goldens and the oracle, including the NULL-output combinations, sigma_extra and the resident-column re-use
    of repeated calls (radiate! once per RCM step, radiative_convective.jl:109-113);
  * Column.update / Column.run_batch (RCM stepping, jacobian!: radiative_convective.jl:109-171) against the ORACLE, with a
    non-constant molar-mass profile;
  * two columns on one context: results of a replaced column are refused, not silently swapped.
"""
import ctypes as C
import math

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu

_dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def ctx(cs):
    c = cs.Context(0)
    yield c
    c.close()


def _julia_call(cs, ctx, nu, P, g, nlob, Tn, mun, Tlev, sls, shapes, cuts, conc, sigma_gray, extra, Stoa, alb, theta_s, nstream,
                want_tau=True, want_M=True):
    """Marshals exactly like julia/ClearSkyHIP.jl:164-185: Matrix{Float64} arguments are column-major, Cint vectors for
    slots/shapes, C_NULL for `extra === nothing`; tau/M+/M- are dense column-major matrices [level, nu]."""
    L = cs.lib()
    f64 = lambda a: np.asfortranarray(np.asarray(a, dtype=np.float64))          # Julia array memory
    ptr = lambda a: None if a is None else a.ctypes.data_as(_dp)
    nnu, npl = len(nu), len(P)
    nu_, P_, Tn_, mun_, Tlev_ = f64(nu), f64(P), f64(Tn), f64(mun), f64(Tlev)
    slots = np.array([ctx.slot_of(sl) for sl in sls], dtype=np.int32)
    shp = np.array([cs.SHAPES[s] for s in shapes], dtype=np.int32)
    cuts_, conc_ = f64(cuts), f64(conc)                                          # conc: [ngas, K] column-major
    extra_ = None if extra is None else f64(np.asarray(extra).T)                 # Julia extra[j, k]: [nnu, K] column-major
    Stoa_, alb_ = f64(Stoa), f64(alb)
    Ta = np.zeros((npl - 1, nnu), order="F") if want_tau else None
    Mu = np.zeros((npl, nnu), order="F") if want_M else None
    Md = np.zeros((npl, nnu), order="F") if want_M else None
    Fu, Fd = np.zeros(npl), np.zeros(npl)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int)) if len(a) else None
    rc = L.cs_fluxes_discretized(ctx.handle, nnu, ptr(nu_), npl, ptr(P_), float(g), int(nlob), ptr(Tn_), ptr(mun_), ptr(Tlev_),
                                 len(sls), ip(slots), ip(shp), ptr(cuts_) if len(sls) else None, ptr(conc_) if len(sls) else None,
                                 float(sigma_gray), ptr(extra_), ptr(Stoa_), ptr(alb_), float(theta_s), int(nstream),
                                 ptr(Ta), ptr(Mu), ptr(Md), ptr(Fu), ptr(Fd))
    ctx._resident = None
    cs.check(rc)
    return dict(tau=Ta, Mup=Mu, Mdn=Md, Fup=Fu, Fdn=Fd)


def _close(a, r, tol):
    assert relerr(a["tau"], r["tau"]) < tol
    sm = np.max(r["Mup"])
    assert np.max(np.abs(a["Mup"] - r["Mup"])) < tol * sm and np.max(np.abs(a["Mdn"] - r["Mdn"])) < tol * sm
    assert np.max(np.abs(a["Fup"] - r["Fup"])) < tol * np.max(r["Fup"])
    assert np.max(np.abs(a["Fdn"] - r["Fdn"])) < tol * max(np.max(r["Fdn"]), np.max(r["Fup"]))


@pytest.mark.parametrize("name", ["column_co2", "column_co2_lob4"])
def test_b3_symbol_vs_golden(cs, golden, lines, ctx, name):
    """cs_fluxes_discretized, marshalled like the Julia ccall, against the independent numpy/scipy goldens."""
    g = golden(name)
    nlob, ns = int(g["nlobatto"]), int(g["nstream"])
    P, nu = g["P"], g["nu"]
    fT = cs.formprofile(P, g["T"])
    fmu = cs.formprofile(P, float(g["mu"]))
    Tn, mun = cs.lobattoevaluations(P, fT, fmu, nlob)                            # fluxes.jl:262
    Tlev = np.array([fT(p) for p in P])
    K = (len(P) - 1) * (nlob - 1) + 1
    conc = np.full((1, K), float(g["conc"]))
    args = (nu, P, float(g["g"]), nlob, Tn, mun, Tlev, [lines("CO2")], ["voigt"], [25.0], conc, 0.0, None,
            np.full(len(nu), float(g["fS"])), np.full(len(nu), float(g["fa"])), 0.841, ns)
    ref = dict(tau=g["tau"], Mup=g["Mup"], Mdn=g["Mdn"], Fup=g["Fup"], Fdn=g["Fdn"])
    a = _julia_call(cs, ctx, *args)
    _close(a, ref, 1e-10)
    # NULL tau, NULL M+/M- (want flags are derived from the pointers): band fluxes must not change
    b = _julia_call(cs, ctx, *args, want_tau=False, want_M=True)
    c = _julia_call(cs, ctx, *args, want_tau=True, want_M=False)
    d = _julia_call(cs, ctx, *args, want_tau=False, want_M=False)
    for x in (b, c, d):
        assert np.array_equal(x["Fup"], a["Fup"]) and np.array_equal(x["Fdn"], a["Fdn"])
    assert np.array_equal(b["Mup"], a["Mup"]) and np.array_equal(b["Mdn"], a["Mdn"]) and np.array_equal(c["tau"], a["tau"])


def test_b3_symbol_extra_gray_and_reuse_vs_oracle(cs, O, lines, ctx):
    """Two gases + gray term + host-evaluated sigma_extra + stellar beam + albedo through the raw symbol, then the call sequence
    of an RCM loop: same grid, new temperatures (the resident column is re-used), then another grid (full setup again)."""
    import workloads as W
    nu = np.linspace(500.0, 900.0, 4001)
    P = cs.pressuregrid(2.0, 1e5, 17)
    nlob, ns = 3, 4
    sls = [lines("H2O"), lines("CO2")]
    Pk = cs.nodepressures(P, nlob)
    K = len(Pk)
    Stoa = 2e-3 * np.exp(-((nu - 800.0) / 150.0) ** 2)
    alb = np.full(len(nu), 0.3)
    # a molar-mass function mu(T,P), not a constant (a mu VECTOR is unusable in the reference too: formprofile turns it into a
    # one-argument AtmosphericProfile, fluxes.jl:13, which lobattoevaluations calls with two, discretized.jl:25)
    mu_prof = lambda T_, P_: 0.0285 + 0.0007 * (P_ / 1e5) + 2e-6 * (T_ - 250.0)

    def inputs(T):
        fT, fmu = cs.formprofile(P, T), cs.formprofile(P, mu_prof)
        Tn, mun = cs.lobattoevaluations(P, fT, fmu, nlob)
        Tk = cs.nodevalues(Tn, nlob)
        conc = np.array([[W.fC_h2o(Tk[k], Pk[k]) for k in range(K)], [400e-6] * K])
        extra = 2e-27 * (Pk[:, None] / 1e5) * (Tk[:, None] / 250.0) * np.ones((K, len(nu)))   # sigma(nu,T,P), [K][nnu]
        return Tn, mun, np.array([fT(p) for p in P]), conc, extra

    def oracle(nu_, Tn, mun, Tlev, conc, extra, S, a):
        return O.fluxes_discretized(nu_, P, 9.8, nlob, Tn, mun, Tlev, sls, ["voigt"] * 2, [25.0] * 2, conc, sigma_gray=3e-28,
                                    sigma_extra=extra, S_toa=S, albedo=a, theta_s=0.6, nstream=ns)

    T1 = W.earth_temperature(P)
    for T in (T1, T1 + np.linspace(-6.0, 4.0, len(P)), T1 - 3.0):               # 2nd and 3rd call hit the re-use path
        Tn, mun, Tlev, conc, extra = inputs(T)
        a = _julia_call(cs, ctx, nu, P, 9.8, nlob, Tn, mun, Tlev, sls, ["voigt"] * 2, [25.0] * 2, conc, 3e-28, extra, Stoa, alb,
                        0.6, ns)
        _close(a, oracle(nu, Tn, mun, Tlev, conc, extra, Stoa, alb), 1e-11)
    # another grid on the same context: everything is rebuilt
    nu2 = np.linspace(640.0, 700.0, 1500)
    Tn, mun, Tlev, conc, extra = inputs(T1)
    S2, a2 = np.interp(nu2, nu, Stoa), np.full(len(nu2), 0.3)
    b = _julia_call(cs, ctx, nu2, P, 9.8, nlob, Tn, mun, Tlev, sls, ["voigt"] * 2, [25.0] * 2, conc, 3e-28, extra[:, :len(nu2)], S2, a2,
                    0.6, ns)
    _close(b, oracle(nu2, Tn, mun, Tlev, conc, extra[:, :len(nu2)], S2, a2), 1e-11)
    # a re-uploaded line table in the same slot invalidates the resident column (no stale windows)
    sl_small = lines("CH4")
    slot = ctx.slot_of(sls[1])
    ctx._slots.pop(id(sls[1]))
    ctx._slots[id(sl_small)] = (slot, sl_small)
    arrs = [cs.as_f64(x) for x in (sl_small.nu, sl_small.S, sl_small.gamma_a, sl_small.gamma_s, sl_small.Epp, sl_small.na, sl_small.mu)]
    iso = np.ascontiguousarray(sl_small.I, dtype=np.int16)
    ncheb = np.ascontiguousarray(sl_small.ncheb, dtype=np.int32)
    cs.check(cs.lib().cs_gas_upload(ctx.handle, slot, len(arrs[0]), *[cs.dptr(x) for x in arrs], iso.ctypes.data_as(C.POINTER(C.c_int16)),
                                    len(ncheb), ncheb.ctypes.data_as(C.POINTER(C.c_int32)), cs.dptr(cs.as_f64(sl_small.cheb))))
    sls2 = [sls[0], sl_small]
    c = _julia_call(cs, ctx, nu2, P, 9.8, nlob, Tn, mun, Tlev, sls2, ["voigt"] * 2, [25.0] * 2, conc, 3e-28, None, S2, a2, 0.6, ns)
    r = O.fluxes_discretized(nu2, P, 9.8, nlob, Tn, mun, Tlev, sls2, ["voigt"] * 2, [25.0] * 2, conc, sigma_gray=3e-28, S_toa=S2,
                             albedo=a2, theta_s=0.6, nstream=ns)
    _close(c, r, 1e-11)


def test_two_columns_on_one_context(cs, O, lines):
    """A context holds ONE resident column: reading a replaced column's results is an error on both sides of the
    ABI, never another column's data or a write past the caller's buffers."""
    ctx = cs.Context(0)
    nu_a, nu_b = np.linspace(600.0, 700.0, 700), np.linspace(600.0, 760.0, 2100)
    P = cs.pressuregrid(10.0, 1e5, 9)
    a = cs.Column(P, 9.8, 250.0, 0.029, 0.0, 0.0, cs.DirectGas(lines("CO2"), 400e-6, nu_a), ctx=ctx)
    b = cs.Column(P, 9.8, 260.0, 0.029, 0.0, 0.0, cs.DirectGas(lines("CO2"), 400e-6, nu_b), ctx=ctx)
    a.run()                       # a is set up again (b replaced it), evaluated ...
    Fa = a.fetch()
    b.run()                       # ... and replaced by b
    for f in (a.fetch, a.sigma_nodes, a.flux_ptr, a.counts, a.work, lambda: a.flux_to(0)):
        with pytest.raises(cs.ClearSkyHIPError) as e:
            f()
        assert e.value.code == -6
    # the C side refuses a size mismatch by itself (a caller that skipped the host-side check)
    tau = np.zeros((a.nl, a.nnu), order="F")
    Fu, Fd = np.zeros(a.np), np.zeros(a.np)
    rc = cs.lib().cs_column_fetch(ctx.handle, a.nnu, a.np, tau.ctypes.data_as(_dp), None, None, cs.dptr(Fu), cs.dptr(Fd))
    assert rc == -6 and b"resident column" in cs.lib().cs_last_error()
    sg = np.zeros((a.K, a.nnu))
    assert cs.lib().cs_column_sigma_fetch(ctx.handle, a.nnu, a.K, cs.dptr(sg)) == -6
    Fb = b.fetch()
    a.run()
    assert np.array_equal(a.fetch()[0], Fa[0])                                   # running again makes it resident again
    rb = O.fluxes_discretized(nu_b, P, 9.8, 2, b.Tn, b.mun, b.Tlev, [lines("CO2")], ["voigt"], [25.0], b.conc)
    assert relerr(Fb[0], rb["Fup"]) < 1e-11
    ctx.close()


def test_update_and_batch_vs_oracle(cs, O, lines):
    """RCM stepping and jacobian! (radiative_convective.jl:109-171) against the oracle: update(T) keeps the column's own
    molar-mass PROFILE (re-evaluated at the new nodes, discretized.jl:19-27), run_batch evaluates the np+1 perturbed profiles."""
    import workloads as W
    ctx = cs.Context(0)
    nu = np.linspace(400.0, 1100.0, 3000)
    P = cs.pressuregrid(5.0, 1e5, 16)
    T0 = W.earth_temperature(P)
    mu = lambda T, Pp: 0.028 + 0.001 * (Pp / 1e5) + 1e-6 * (T - 250.0)           # mu(T,P): callable form, fluxes.jl:15
    g1 = cs.DirectGas(lines("CO2"), 400e-6, nu)
    g2 = cs.DirectGas(lines("H2O"), W.fC_h2o, nu)
    core = cs.Discretized(5, 3)
    col = cs.Column(P, 9.8, T0, mu, 0.0, 0.0, g1, g2, cs.GrayGas(1e-27, nu), core=core, want_tau=True, want_M=False, ctx=ctx)

    def oracle(T):
        fT = cs.formprofile(P, T)
        Tn, mun = cs.lobattoevaluations(P, fT, mu, 3)
        Tk, Pk = cs.nodevalues(Tn, 3), cs.nodepressures(P, 3)
        conc = np.array([[400e-6] * len(Pk), [W.fC_h2o(Tk[k], Pk[k]) for k in range(len(Pk))]])
        return O.fluxes_discretized(nu, P, 9.8, 3, Tn, mun, np.array([fT(p) for p in P]), [g1.sl, g2.sl], ["voigt"] * 2, [25.0] * 2,
                                    conc, sigma_gray=1e-27)
    T2 = T0 + np.linspace(-5.0, 3.0, len(P))
    col.run()
    col.update(T2)                                                                # mu=None: keep the column's mu(T,P)
    col.run()
    tau = np.zeros((col.nl, col.nnu), order="F")
    Fu, Fd = col.fetch(tau)
    r = oracle(T2)
    assert relerr(tau, r["tau"]) < 1e-11
    assert np.max(np.abs(Fu - r["Fup"])) < 1e-11 * r["Fup"].max() and np.max(np.abs(Fd - r["Fdn"])) < 1e-11 * r["Fup"].max()
    Ts = [T0] + [T0 + 0.5 * (np.arange(len(P)) == i) for i in range(len(P))]      # jacobian!: one level perturbed at a time
    Bu, Bd = col.run_batch(Ts)                                                    # mus=None: same mu(T,P)
    assert Bu.shape == (len(P) + 1, len(P))
    for b in (0, 1, 8, len(P)):
        r = oracle(Ts[b])
        assert np.max(np.abs(Bu[b] - r["Fup"])) < 1e-11 * r["Fup"].max() and np.max(np.abs(Bd[b] - r["Fdn"])) < 1e-11 * r["Fup"].max()
    # a new mu replaces the stored one
    col.update(T2, 0.029)
    col.run()
    Fc = col.fetch()
    fT = cs.formprofile(P, T2)
    Tn, mun = cs.lobattoevaluations(P, fT, cs.formprofile(P, 0.029), 3)
    Tk, Pk = cs.nodevalues(Tn, 3), cs.nodepressures(P, 3)
    conc = np.array([[400e-6] * len(Pk), [W.fC_h2o(Tk[k], Pk[k]) for k in range(len(Pk))]])
    r = O.fluxes_discretized(nu, P, 9.8, 3, Tn, mun, np.array([fT(p) for p in P]), [g1.sl, g2.sl], ["voigt"] * 2, [25.0] * 2, conc,
                             sigma_gray=1e-27)
    assert np.max(np.abs(Fc[0] - r["Fup"])) < 1e-11 * r["Fup"].max()
    ctx.close()


def test_par_file_straight_into_a_gas_slot(cs, O, ctx):
    """f3 to the letter (hitran/par.jl:91-286): cs_gas_upload_par parses, filters (range, intensity cut, isotopologues), keeps the
    strongest N, sorts, looks the molar masses up and uploads on the native side.  The slot's arrays are bit-equal to the Python
    mirror's readpar + SpectralLines for several filter sets (also with isotopologue CHARACTERS, as the reference accepts), the
    reference's error cases raise, and a column built on the natively loaded table matches the oracle."""
    import workloads as W
    f = W.fixture("CO2.par")
    for kw in (dict(), dict(numin=600.0, numax=720.0), dict(Scut=1e-24, I=(1, 2)), dict(I=("1", "3")), dict(maxlines=300),
               dict(numin=550.0, numax=800.0, Scut=1e-26, I=(1, 2, 3), maxlines=500)):
        a = ctx.load_par(f, 2, **kw)
        b = cs.SpectralLines(f, **kw)
        assert a.N == b.N and a.M == b.M
        for name in ("nu", "S", "gamma_a", "gamma_s", "Epp", "na", "mu", "A"):
            assert np.array_equal(getattr(a, name), getattr(b, name)), (kw, name)
        assert np.array_equal(a.I, b.I)
        mu_dev = np.zeros(a.N)
        cs.check(cs.lib().cs_gas_fetch(ctx.handle, ctx.slot_of(a), a.N, None, None, None, None, None, None,
                                       mu_dev.ctypes.data_as(_dp), None))
        assert np.array_equal(mu_dev, b.mu)                                    # (the MOLPARAM lookup happened on the native side)
    with pytest.raises(cs.ClearSkyHIPError, match="filtered to nothing"):
        ctx.load_par(f, 2, numin=1e5)
    with pytest.raises(cs.ClearSkyHIPError, match="only one molecule"):
        ctx.load_par(f, 1)                                                      # (the file holds CO2, not H2O)
    with pytest.raises(AssertionError):
        ctx.load_par(f + ".txt", 2)
    sl = ctx.load_par(f, 2, numin=550.0, numax=800.0)
    nslots = len(ctx._slots)
    nu = np.linspace(600.0, 750.0, 3001)
    P = cs.pressuregrid(10.0, 1e5, 13)
    T = cs.AtmosphericProfile(P, np.linspace(210.0, 290.0, 13))
    col = cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, cs.DirectGas(sl, 400e-6, nu), core=cs.Discretized(4, 2), want_tau=True, ctx=ctx)
    assert len(ctx._slots) == nslots                                           # bound to its slot: no second upload
    col.run()
    tau = np.zeros((col.nl, col.nnu), order="F")
    Fup, Fdn = col.fetch(tau)
    ref_sl = cs.SpectralLines(f, numin=550.0, numax=800.0)
    r = O.fluxes_discretized(nu, P, 9.8, 2, col.Tn, col.mun, col.Tlev, [ref_sl], ["voigt"], [25.0], col.conc, nstream=4)
    assert relerr(tau, r["tau"]) < 1e-11 and abs(Fup[0] - r["Fup"][0]) < 1e-11 * r["Fup"][0]
