#!/usr/bin/env python3
"""The drop-in entry point with the whole FluxPack copied back (tau, M+, M-: 146 MB at C3), through n contexts on ONE device:
cs_fluxes_discretized_multi cuts the grid into n ranges, one host thread per context -- the device-to-host copies of one range run
beside the kernels of the others.  Prints ms per call (repeat calls on an unchanged grid) for n = 1, 2, 4, 8:
tools/multi_overlap.py <output.json> [config]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import clearsky_jl_amd as cs
import workloads as W
from clearsky_jl_amd.core import _fluxes_discretized

cfg = W.config(sys.argv[2] if len(sys.argv) > 2 else "C3")
out = {}
for n in (1, 2, 4, 8):
    mc = cs.MultiContext([0] * n) if n > 1 else cs.Context(0)
    d = cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], cfg["fS"], cfg["fa"], *cfg["absorbers"], core=cfg["core"], theta_s=cfg["theta_s"], ctx=mc, _setup=False)
    tau = np.zeros((d.nl, d.nnu), order="F"); Mu = np.zeros((d.np, d.nnu), order="F"); Md = np.zeros((d.np, d.nnu), order="F")
    res = {}
    for name, bufs in (("band_fluxes", (None, None, None)), ("with_tau_M", (tau, Mu, Md))):
        _fluxes_discretized(d, *bufs)            # setup
        _fluxes_discretized(d, *bufs)
        t0 = time.perf_counter()
        for _ in range(5):
            F = _fluxes_discretized(d, *bufs)
        res[name] = (time.perf_counter() - t0) / 5 * 1e3
    res["olr"] = float(F[0][0])
    out[n] = res
    print(n, res, flush=True)
    mc.close()
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/multi_overlap.json", "w"), indent=1)
