"""dev: one seed of tests/test_gpu_fuzz.py::test_interp_fuzz under tuning variants (which path carries an error)"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import clearsky_jl_amd as cs
from conftest import relerr
from test_gpu_fuzz import _table

def one(seed, tunes):
    rng = np.random.default_rng(9000 + seed)
    cut = float(rng.choice([1.0, 5.0, 25.0, 25.0, 60.0]))
    nlev_target = int(rng.integers(1, 6))
    n = int(rng.integers(130, 9000))
    dnu = 1.5 * cut / 2.3 / (128 * 2 ** (nlev_target - 1)) * float(rng.uniform(0.55, 0.95))
    c0 = float(rng.uniform(300, 2500))
    span = dnu * (n - 1)
    kind = ["uniform", "random", "log", "jitter"][seed % 4]
    if kind == "uniform":
        nu = c0 + dnu * np.arange(n)
    elif kind == "random":
        nu = np.unique(c0 + np.sort(rng.uniform(0, span, n)))
    elif kind == "log":
        nu = np.unique(c0 * np.exp(np.linspace(0, np.log1p(span / c0), n)))
    else:
        nu = c0 + dnu * (np.arange(n) + rng.uniform(-0.4, 0.4, n))
    M = int(rng.choice([1, 2, 2, 6, 45]))
    L = int(rng.integers(200, 6000))
    sl = _table(cs, rng, M, L, c0 - 2 * cut - 5, c0 + span + 2 * cut + 5, dense=int(L // 4) if seed % 5 == 0 else None, dup=seed % 7 == 0)
    K = int(rng.integers(1, 40))
    T = rng.uniform(25, 1000, K) if M != 45 else rng.uniform(100, 1000, K)
    P = 10 ** rng.uniform(-1, 5.5, K)
    P[rng.random(K) < 0.1] = 0.0
    Pp = P * rng.uniform(0, 1, K)
    off = cs.Context(0); off.set_interp(False)
    b = cs.shape_batch(sl, "voigt", nu, list(T), list(P), list(Pp), cut, off)
    out = []
    for tn in tunes:
        on = cs.Context(0); on.set_matrix_cores(2)
        for k, v in tn: on.set_tuning(k, v)
        a = cs.shape_batch(sl, "voigt", nu, list(T), list(P), list(Pp), cut, on)
        e = np.abs(a - b) / np.maximum(np.abs(b), 1e-280)
        kk, ii = np.unravel_index(np.argmax(e), e.shape)
        out.append((tn, float(e.max()), int(kk), int(ii), float(P[kk]), float(T[kk])))
        on.close()
    off.close()
    print(seed, kind, cut, n, M, L, K, cs.interp_plan(nu, cut))
    for o in out: print("   ", o)

tunes = [[], [(17, 1)], [(20, 1)], [(17, 1), (20, 1)], [(11, 1)], [(23, 1)]]
for seed in (int(s) for s in sys.argv[1:]):
    one(seed, tunes)
