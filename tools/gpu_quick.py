"""First-contact GPU check: Faddeeva batch, shape batch and a small column vs the oracle; prints max rel diffs + timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clearsky_jl_amd as cs
from oracle import oracle as O

rng = np.random.default_rng(0)
n = 400000
x = np.concatenate([rng.uniform(0, 12, n), 10 ** rng.uniform(0, 7, n)])
y = np.concatenate([10 ** rng.uniform(-10, 1.2, n), 10 ** rng.uniform(-6, 3, n)])
g = cs.faddeeva(x, y); o = O.faddeeva(x, y)
e = np.abs(g / o - 1); i = e.argmax()
print("faddeeva gpu vs oracle: max rel %.3e at (%g,%g)" % (e.max(), x[i], y[i]), flush=True)

H = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hitran")
for name in ("CO2", "H2O"):
    sl = cs.SpectralLines(os.path.join(H, name + ".par"))
    nu = np.linspace(1.0, 2500.0, 4001)
    for shape in ("voigt", "lorentz", "doppler", "PHCO2"):
        Ts = [220.0, 296.0, 260.0]; Ps = [50.0, 101325.0, 3e3]; Pps = [p * 400e-6 for p in Ps]
        cut = 500.0 if shape == "PHCO2" else 25.0
        t0 = time.time(); sg = cs.shape_batch(sl, shape, nu, Ts, Ps, Pps, cut); t1 = time.time()
        worst = 0
        for k in range(3):
            so = O.shape_bang(shape, nu, sl, Ts[k], Ps[k], Pps[k], cut)
            m = so > 0
            worst = max(worst, np.max(np.abs(sg[k][m] / so[m] - 1)))
            assert np.all(sg[k][~m] == 0)
        print(f"{name} {shape}: max rel diff {worst:.3e}  gpu {t1-t0:.3f}s", flush=True)

import __graft_entry__ as ge
ge.smoke()
