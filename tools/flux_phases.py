#!/usr/bin/env python3
"""Phase times of k_flux_scan's block 0 (cs_set_tuning key 15 | 128): tools/flux_phases.py [C2 | shard]"""
import ctypes as C
import sys
sys.path.insert(0, ".")
import numpy as np
import clearsky_jl_amd as cs
import workloads as W
which = sys.argv[1] if len(sys.argv) > 1 else "shard"
cfg = W.config("C2" if which == "C2" else "C3")
ctx = cs.Context(0)
ctx.set_tuning(15, 128)
rng = None if which == "C2" else W.balanced_ranges(cfg["nu"], cfg["absorbers"], 8)[3]
col = cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], cfg["fS"], cfg["fa"], *cfg["absorbers"], core=cfg["core"], theta_s=cfg["theta_s"], nu_range=rng, ctx=ctx)
for _ in range(5):
    col.run()
col.sync()
out = (C.c_int64 * 32)()
cs.check(cs.lib().cs_column_work(ctx.handle, out))
import os
print(which, "flux form", col.info()["flux_form"], "block 0 [us]: sigma %.1f  depths+planck %.1f  pass1 %.1f  handover %.1f | block 0 start -> band fluxes stored %.1f" % tuple(out[27 + q] / 1e3 for q in range(5)))
prof = col.profile(reps=5)
print("   rt class (HIP events) %.1f us, reduce %.1f, apply %.1f" % (prof["rt"] * 1e3, prof["reduce"] * 1e3, prof["apply"] * 1e3))
