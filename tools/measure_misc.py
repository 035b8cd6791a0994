"""Numbers quoted in DESIGN.md: PCIe-inclusive rate of the host-pointer entry point, gray-gas achieved errors."""
import ctypes as C, math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clearsky_jl_amd as cs
import workloads as W
from clearsky_jl_amd._lib import lib, dptr, check

cfg = W.config("C3")
ctx = cs.Context(0)
col = cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], 0.0, 0.0, *cfg["absorbers"], core=cfg["core"], ctx=ctx)
nnu, npl = col.nnu, col.np
tau = np.zeros((npl - 1, nnu), order="F"); Mup = np.zeros((npl, nnu), order="F"); Mdn = np.zeros((npl, nnu), order="F")
Fup = np.zeros(npl); Fdn = np.zeros(npl)
ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
Tn = col.Tn.ravel(order="F").copy(); mun = col.mun.ravel(order="F").copy(); conc = col.conc.ravel(order="F").copy()
def call(full):
    t = time.perf_counter()
    check(lib().cs_fluxes_discretized(ctx.handle, nnu, dptr(col.nu), npl, dptr(col.P), col.g, 2, dptr(Tn), dptr(mun), dptr(col.Tlev),
          2, ip(col.slots), ip(col.shapes), dptr(col.cuts), dptr(conc), 0.0, None, None, None, 0.841, 5,
          fp(tau) if full else None, fp(Mup) if full else None, fp(Mdn) if full else None, dptr(Fup), dptr(Fdn)))
    return time.perf_counter() - t
for full in (True, False):
    call(full)
    ts = [call(full) for _ in range(5)]
    print("cs_fluxes_discretized host-pointer call, %s: %.1f ms -> %.3e spectral-points/s (OLR %.6f)" % (
        "tau+M+,M- returned (146 MB D2H)" if full else "band fluxes only", 1e3 * min(ts), nnu * (npl - 1) / min(ts), Fup[0]))
