"""Numerical basis of the interpolated far wings; CPU only.

1. Pure interpolation error (40-digit arithmetic) of a line wing 1/((x-x0)^2+eps) through 64 Chebyshev extrema when the line
   centre sits c half-widths beyond the interval: c = 0.3 -> 6e-19 pointwise, c = 0.2 -> 4e-15.  kChebMargin = 0.3.
2. In fp64 the nodes cen + h cos(pi m/63) are rounded by ~ulp(nu)/h of the interval; with the closed-form barycentric weights
   (-1)^m {1/2,1,...,1/2} the interpolant is then off by 1e-13 (h = 1.6 at nu = 690) to 1e-12 (h = 0.064).  Weights computed
   from the rounded nodes, w_m = 1/prod(x_m - x_j), bring it back to 1e-15 -- k_cheb_setup does that.
3. Relative noise eps in the node values comes out as ~4 eps pointwise (no amplification by the dynamic range of the wing).
"""
import numpy as np

n = 64


def pure_error(c, pts=64):
    from mpmath import mp, mpf, cos, pi
    mp.dps = 40
    xs = [cos(pi * m / (n - 1)) for m in range(n)]
    w = [(-1) ** m * (mpf(1) / 2 if m in (0, n - 1) else 1) for m in range(n)]
    x0 = mpf(1) + c
    f = lambda x: 1 / ((x - x0) ** 2 + mpf("1e-8"))
    F = [f(x) for x in xs]
    worst = 0
    for t in np.linspace(-0.999, 0.999, pts):
        t = mpf(float(t))
        num = sum(w[m] / (t - xs[m]) * F[m] for m in range(n))
        den = sum(w[m] / (t - xs[m]) for m in range(n))
        worst = max(worst, abs(num / den - f(t)) / f(t))
    return float(worst)


def fp64_error(c, cen, h, exact_weights, noise=0.0, npts=256, seed=0):
    rng = np.random.default_rng(seed)
    xs = cen + h * np.cos(np.pi * np.arange(n) / (n - 1))
    if exact_weights:
        D = 2.0 * (xs[:, None] - xs[None, :]) / h
        np.fill_diagonal(D, 1.0)
        w = 1 / np.prod(D, axis=1)
    else:
        w = (-1.0) ** np.arange(n)
        w[0] *= 0.5
        w[-1] *= 0.5
    x0 = cen + h * (1 + c)
    f = lambda x: 1 / ((x - x0) ** 2 + 1e-10)
    t = np.linspace(cen - h, cen + h, npts)
    d = t[:, None] - xs[None, :]
    hit = d == 0
    d[hit] = 1
    C = (w / d) / np.sum(np.where(hit, 0, w / d), axis=1, keepdims=True)
    p = C @ (f(xs) * (1 + noise * rng.standard_normal(n)))
    e = np.abs(p / f(t) - 1)
    e[hit.any(axis=1)] = 0
    return e.max()


if __name__ == "__main__":
    print("1. pure interpolation error vs margin c")
    for c in (0.2, 0.25, 0.3, 0.4, 0.5):
        print(f"   c = {c}: {pure_error(c):.2e}")
    print("2. fp64, nu = 690, closed-form vs computed weights")
    for h in (1.6, 0.064):
        print(f"   h = {h}: closed form {fp64_error(0.3, 690.0, h, False):.2e}, from rounded nodes {fp64_error(0.3, 690.0, h, True):.2e}")
    print("3. node noise -> pointwise error (c = 0.3, exact weights)")
    for eps in (0.0, 3e-15, 1e-14):
        print(f"   eps = {eps:g}: {fp64_error(0.3, 690.0, 1.6, True, eps):.2e}")
