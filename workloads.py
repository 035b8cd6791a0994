"""Synthetic inputs of the BASELINE.json configurations (SURVEY.md 8d "Concrete synthetic inputs").

Bench / test support, not part of the product package: it reads the HITRAN fixture files under tests/golden/hitran and is
imported by bench.py, tests/ and tools/ only.
"""
import os

import numpy as np

from clearsky_jl_amd import constants as K
from clearsky_jl_amd import CIATables, DirectGas, Discretized, GrayGas, SpectralLines, ozonelayer, pressuregrid, psatH2O

_HITRAN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "hitran")


def fixture(name: str) -> str:
    return os.path.join(_HITRAN, name)


def earth_temperature(P):
    """T_k = max(288 (P/1e5)^(R/(0.029*1040)), 200) K at the levels"""
    return np.maximum(288.0 * (np.asarray(P) / 1e5) ** (K.R / (0.029 * 1040.0)), 200.0)


def fC_h2o(T, P):
    return min(0.8 * psatH2O(T) / P, 0.04)


_cache = {}


def lines(kind: str, which: str):
    """which in {"H2O","CO2","CH4"}; kind "fixture" (reference test/HITRAN files) or "synthetic" (5e4 seeded lines)."""
    key = (kind, which)
    if key not in _cache:
        if kind == "fixture":
            _cache[key] = SpectralLines(fixture(which + ".par"))
        else:
            M = {"H2O": 1, "CO2": 2, "O3": 3, "CH4": 6}[which]
            _cache[key] = SpectralLines.synthetic(M, 50000, 20260101 + M)
    return _cache[key]


def config(name: str, nnu=None, nl=None, lines_kind=None, nu_span=(1.0, 2500.0), shape="voigt"):
    """Returns dict(P, g, T, mu, fS, fa, absorbers, core, theta_s, nu) for
    "C2" CO2 fixture, 1e4 nu x 40 layers;  "C3" H2O+CO2, 1e5 nu x 60 layers (synthetic ~1e5-line table by default);
    "C5" H2O+CO2+CH4+O3 + CIA, 5e5 nu x 100 layers.  `nu_span` cuts a window out of the 1..2500 cm^-1 grid (parity tests at
    the full grid's spacing on a grid small enough for the CPU checker)."""
    if name == "C2":
        nnu, nl, lines_kind = nnu or 10_000, nl or 40, lines_kind or "fixture"
        gases = ["CO2"]
    elif name == "C3":
        nnu, nl, lines_kind = nnu or 100_000, nl or 60, lines_kind or "synthetic"
        gases = ["H2O", "CO2"]
    elif name == "C5":
        # H2O+CO2+CH4+O3 with CIA continuum, 5e5 wavenumbers x 100 layers: reference fixtures for H2O/CO2/CH4, a seeded
        # synthetic O3 table (the container holds no O3 file) and the two CIA files of test/HITRAN
        nnu, nl, lines_kind = nnu or 500_000, nl or 100, lines_kind or "fixture"
        gases = ["H2O", "CO2", "CH4", "O3"]
    else:
        raise ValueError(name)
    nu = np.linspace(float(nu_span[0]), float(nu_span[1]), nnu)
    P = pressuregrid(1.0, 1e5, nl + 1)
    T = earth_temperature(P)
    absorbers = []
    for gname in gases:
        fC = {"H2O": fC_h2o, "CO2": 400e-6, "CH4": 1.8e-6, "O3": lambda T, P_: ozonelayer(P_)}[gname]
        kind = "synthetic" if gname == "O3" else lines_kind
        absorbers.append(DirectGas(lines(kind, gname), fC, nu, shape=shape))   # (PHCO2: its default 500 cm^-1 cut-off)
    if name == "C5":
        absorbers.append(CIATables(fixture("CO2-CO2_2018.cia")))
        absorbers.append(CIATables(fixture("CO2-CH4_2018.cia")))
    return dict(name=name, nu=nu, P=P, g=9.8, T=T, mu=0.029, fS=0.0, fa=0.0, absorbers=absorbers,
                core=Discretized(5, 2), theta_s=0.841, lines_kind=lines_kind, nl=nl)


def balanced_ranges(nu, absorbers, nparts: int):
    """Contiguous nu ranges of equal estimated device time: the multi-GPU partition of SURVEY.md 8e, as the product cuts it
    (cs_balanced_ranges in the C ABI -- what cs_fluxes_discretized_multi uses; cost model and edge rules documented there)."""
    from clearsky_jl_amd import balanced_ranges as _br
    return _br(np.asarray(nu, float), [a.sl.nu for a in absorbers if isinstance(a, DirectGas)], nparts)
