#!/usr/bin/env python3
"""Outgoing longwave radiation of an Earth-like column, line by line, on one MI355X -- the calls a ClearSky.jl user would make
(reference names; `!` -> trailing underscore).  Needs a GPU:  python examples/olr_earth.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clearsky_jl_amd as cs
import workloads as W

H = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hitran")
nu = np.linspace(1.0, 2500.0, 100_000)                       # wavenumber grid [cm^-1]
P = cs.pressuregrid(1.0, 1e5, 61)                            # 60 layers, top of atmosphere first
T = W.earth_temperature(P)                                   # adiabat with a 200 K stratosphere
co2 = cs.DirectGas(cs.SpectralLines(os.path.join(H, "CO2.par")), 400e-6, nu)
h2o = cs.DirectGas(cs.SpectralLines(os.path.join(H, "H2O.par")), W.fC_h2o, nu)
F = cs.radiate(P, 9.8, T, 0.029, 0.0, 0.0, co2, h2o, core=cs.Discretized(nstream=5, nlobatto=2))
print(f"OLR = {F.Fup[0]:.3f} W/m^2, surface downwelling = {F.Fdn[-1]:.3f} W/m^2")

# the reference's default objects: cross-sections baked on a (T, ln P) Chebyshev grid, then interpolated
Om = cs.AtmosphericDomain((150.0, 350.0), 12, (0.9, 1.1e5), 24)
co2b = cs.Gas(co2.sl, 400e-6, nu, Om)
h2ob = cs.Gas(h2o.sl, W.fC_h2o, nu, Om)
Fb = cs.radiate(P, 9.8, T, 0.029, 0.0, 0.0, co2b, h2ob)
print(f"OLR with baked opacity tables = {Fb.Fup[0]:.3f} W/m^2 (table interpolation error, gases.jl:7)")
