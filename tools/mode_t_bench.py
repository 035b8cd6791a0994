#!/usr/bin/env python3
"""Mode T timing on the GPU box: bake two gases on the docs' (T, P) grid (12 x 24 states, gases.jl:97-145) over the C3 wavenumber
grid, then time whole-column flux evaluations with the baked Gas objects (the reference's default mode).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clearsky_jl_amd as cs  # noqa: E402
import workloads as W  # noqa: E402

cfg = W.config("C3")
ctx = cs.Context(0)
Om = cs.AtmosphericDomain((180.0, 320.0), 12, (0.9, 1.1e5), 24)
t0 = time.perf_counter()
gases = [cs.Gas(g.sl, g.fC, cfg["nu"], Om, ctx=ctx) for g in cfg["absorbers"]]
t_bake = time.perf_counter() - t0
col = cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], 0.0, 0.0, *gases, core=cfg["core"], want_tau=False, want_M=False, ctx=ctx)
for _ in range(3):
    col.run()
col.sync()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    col.run()
col.sync()
ms = (time.perf_counter() - t0) / n * 1e3
prof = col.profile(reps=5)
F = col.fetch()
print(json.dumps(dict(bake_s_two_gases=t_bake, states=Om.nT * Om.nP, ms_per_step=ms, kernel_ms=prof, olr=float(F[0][0]))))
