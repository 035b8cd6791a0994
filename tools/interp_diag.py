"""Where does the interpolated far-wing sum differ from the per-point sum?  (GPU box)  usage: interp_diag.py [grid]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clearsky_jl_amd as cs

H = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hitran")
sl = cs.SpectralLines(os.path.join(H, "CO2.par"))
nu = np.linspace(640.0, 700.0, 60001)
STATES = [(220.0, 50.0, 0.02), (296.0, 101325.0, 40.53), (260.0, 3e3, 30.0), (190.0, 2.0, 0.0)]
T, P, Pp = map(list, zip(*STATES))
print("plan", cs.interp_plan(nu, 25.0))
con, coff = cs.Context(0), cs.Context(0)
coff.set_interp(False)
on = cs.shape_batch(sl, "voigt", nu, T, P, Pp, 25.0, con)
off = cs.shape_batch(sl, "voigt", nu, T, P, Pp, 25.0, coff)
from oracle import oracle as O
for k in range(4):
    e = np.abs(on[k] / off[k] - 1)
    i = int(e.argmax())
    j = np.searchsorted(sl.nu, nu[i])
    near = sl.nu[max(j - 2, 0): j + 2] - nu[i]
    print(f"state {k}: max rel {e.max():.3e} at i={i} (i%128={i%128}, i%2048={i%2048}) nu={nu[i]:.4f} sigma={off[k][i]:.3e} nearest lines {near}")
    print("   99.9th pct %.2e, median %.2e" % (np.quantile(e, 0.999), np.median(e)))
    idx = np.arange(max(i - 3, 0), min(i + 4, nu.size))
    so = O.shape_bang("voigt", nu[idx], sl, T[k], P[k], Pp[k], 25.0)
    print("   on/oracle-1:", (on[k][idx] / so - 1), "\n   off/oracle-1:", (off[k][idx] / so - 1))
