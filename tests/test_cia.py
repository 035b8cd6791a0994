"""CIA: reader + tables on the CPU, and the k_cia kernel inside a column on the GPU (collision_induced_absorption.jl)."""
import os

import numpy as np
import pytest

from conftest import HITRAN, relerr


def test_readcia_and_tables(cs):
    d = cs.readcia(os.path.join(HITRAN, "CO2-CO2_2018.cia"))
    assert len(d) == 20 and d[0]["symbol"] == "CO2-CO2" and d[0]["npts"] == len(d[0]["nu"]) == 750
    assert d[0]["nu"][0] == 1.0 and d[0]["k"][0] == 3.439e-46 and d[0]["T"] == 200.0
    x = cs.CIATables(d)
    assert x.formulae == ("CO2", "CO2") and len(x.grids) == 3 and len(x.single) == 2
    y = cs.CIATables(os.path.join(HITRAN, "CO2-CH4_2018.cia"))
    assert y.formulae == ("CO2", "CH4") and len(y.grids) == 1 and len(y.single) == 0
    with pytest.raises(AssertionError):
        cs.readcia("x.par")


def test_host_functor_vs_numpy_restatement(cs, O):
    d = cs.readcia(os.path.join(HITRAN, "CO2-CO2_2018.cia"))
    nu = np.array([1.0, 100.5, 749.9, 750.0, 900.0, 1200.25, 2600.0, 3000.0])
    for extrap, singles, T in ((False, False, 288.0), (True, False, 150.0), (False, True, 250.0), (True, True, 900.0)):
        x = cs.CIATables(d, extrapolate=extrap, singles=singles)
        a = np.array([cs.cia(v, x, T, 1e5, 4e4, 4e4) for v in nu])
        b = O.cia_sigma(d, nu, T, 1e5, 4e4, 4e4, extrap, singles)
        assert relerr(a, b, floor=1e-300) < 1e-12
    x = cs.CIATables(d)
    assert x(100.0, 288.0) > 0 and x(900.0, 288.0) == 0.0 and x(100.0, 150.0) == 0.0     # outside nu / outside T without extrapolation
    assert cs.cia(2e-44, 300.0, 1e5, 4e4, 3e4) == pytest.approx((2e-44 * 7.21879268e38) * (4e4 / 101325 * 273.15 / 300) *
                                                                (3e4 / 101325 * 273.15 / 300) / (1e-6 * 1e5 / (1.38064852e-23 * 300)), rel=1e-14)


def test_pairing_rules(cs, lines):
    nu = np.linspace(1, 2000, 50)
    co2 = cs.DirectGas(lines("CO2"), 0.9, nu)
    ch4 = cs.DirectGas(lines("CH4"), 1e-3, nu)
    x = cs.CIATables(os.path.join(HITRAN, "CO2-CH4_2018.cia"))
    U = cs.UnifiedAbsorber(co2, ch4, x)
    assert len(U.cia) == 1 and U.cia[0].g1 is co2 and U.cia[0].g2 is ch4
    with pytest.raises(AssertionError):
        cs.UnifiedAbsorber(co2, x)                        # CH4 gas missing
    with pytest.raises(ValueError):
        cs.UnifiedAbsorber(cs.GrayGas(1e-26, nu), x)      # gray gases are ignored for pairing -> no gas at all


@pytest.mark.gpu
@pytest.mark.parametrize("extrap,singles", [(False, False), (True, False), (True, True)])
def test_column_with_cia_vs_oracle(cs, O, lines, extrap, singles):
    """A thick CO2/CH4 column with both CIA pairs: sigma at the nodes, tau and fluxes vs the oracle fed the numpy CIA."""
    ctx = cs.Context(0)
    nu = np.linspace(1.0, 2900.0, 2400)
    co2 = cs.DirectGas(lines("CO2"), 0.95, nu)
    ch4 = cs.DirectGas(lines("CH4"), lambda T, P: 0.01 * (P / 2e5) ** 0.1, nu)
    d1 = cs.readcia(os.path.join(HITRAN, "CO2-CO2_2018.cia"))
    d2 = cs.readcia(os.path.join(HITRAN, "CO2-CH4_2018.cia"))
    x1 = cs.CIATables(d1, extrapolate=extrap, singles=singles)
    x2 = cs.CIATables(d2, extrapolate=extrap, singles=singles)
    P = cs.pressuregrid(10.0, 2e5, 10)
    T = np.linspace(170.0, 320.0, 10)
    col = cs.Column(P, 3.7, T, 0.044, 0.0, 0.0, co2, ch4, x1, x2, core=cs.Discretized(5, 3), ctx=ctx)
    col.run()
    F = cs.FluxPack(len(P), len(nu))
    F.Fup[:], F.Fdn[:] = col.fetch(F.tau, F.Mup, F.Mdn)
    extra = np.zeros((col.K, len(nu)))
    for k in range(col.K):
        Tk, Pk = col.Tk[k], col.Pk[k]
        c1, c2 = co2.fC(Tk, Pk), ch4.fC(Tk, Pk)
        extra[k] = (O.cia_sigma(d1, nu, Tk, Pk, Pk * c1, Pk * c1, extrap, singles)
                    + O.cia_sigma(d2, nu, Tk, Pk, Pk * c1, Pk * c2, extrap, singles))
    assert np.nanmax(extra) > 0
    if singles:
        # the reference's single-temperature ranges contain k <= 0 samples -> ln 0 = -Inf knots -> NaN between two of them
        # (collision_induced_absorption.jl:187-188).  Same NaN pattern, same numbers elsewhere.
        sg = col.sigma_nodes()
        lines_only = O.fluxes_discretized(nu, P, 3.7, 3, col.Tn, col.mun, col.Tlev, [co2.sl, ch4.sl], ["voigt"] * 2, [25.0] * 2,
                                          col.conc, want_sigma=True)["sigma"]
        ref = lines_only + extra
        assert np.isnan(ref).sum() > 0 and np.array_equal(np.isnan(sg), np.isnan(ref))
        ok = ~np.isnan(ref)
        assert relerr(sg[ok], ref[ok], floor=1e-300) < 1e-11
        ctx.close()
        return
    r = O.fluxes_discretized(nu, P, 3.7, 3, col.Tn, col.mun, col.Tlev, [co2.sl, ch4.sl], ["voigt"] * 2, [25.0] * 2, col.conc,
                             sigma_extra=extra, want_sigma=True)
    assert relerr(col.sigma_nodes(), r["sigma"], floor=1e-300) < 1e-11
    assert relerr(F.tau, r["tau"]) < 1e-11
    sm = r["Mup"].max()
    assert np.max(np.abs(F.Mup - r["Mup"])) < 1e-11 * sm and np.max(np.abs(F.Mdn - r["Mdn"])) < 1e-11 * sm
    assert np.max(np.abs(F.Fup - r["Fup"])) < 1e-11 * r["Fup"].max()
    # the continuum matters in this column (otherwise the test would not see the kernel)
    r0 = O.fluxes_discretized(nu, P, 3.7, 3, col.Tn, col.mun, col.Tlev, [co2.sl, ch4.sl], ["voigt"] * 2, [25.0] * 2, col.conc)
    assert abs(r0["Fup"][0] - r["Fup"][0]) > 1e-3
    ctx.close()
