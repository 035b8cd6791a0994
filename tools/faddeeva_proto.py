"""Prototype of the tiered Re w(x+iy) algorithm used by oracle/ and the HIP kernels.

Validated here against scipy.special.wofz and mpmath (50 digits).  Scratch/validation tool, not product.
"""
import numpy as np, mpmath as mp
from scipy.special import wofz

SQPI = np.sqrt(np.pi)

def tierA(x, y, nterm=4):
    """real asymptotic series: K*sqrt(pi) = sum_k a_k rho^-(2k+1) sin((2k+1)theta)"""
    s = x*x + y*y
    inv = 1.0/s
    t = y*y*inv
    p1 = 1.5 - 2.0*t                       # 0.5*(3-4t)
    p2 = 3.75 + t*(-15.0 + 12.0*t)         # 0.75*(5-20t+16t^2)
    p3 = 1.875*(7.0 + t*(-56.0 + t*(112.0 - 64.0*t)))
    if nterm == 3:
        inner = p1 + inv*p2
    else:
        inner = p1 + inv*(p2 + inv*p3)
    return y*inv*(1.0 + inv*inner)/SQPI

def cf(x, y, J):
    """J-term Laplace continued fraction, complex arithmetic, bottom-up"""
    z = x + 1j*y
    t = z.copy() if isinstance(z, np.ndarray) else z
    for k in range(J-1, 0, -1):
        t = z - (0.5*k)/t
    return (1j/(SQPI*t)).real

def trap(x, y, h=0.5, N=13):
    """Poisson-corrected trapezoid rule on the (possibly half-shifted) grid"""
    x = np.asarray(x, float); y = np.asarray(y, float)
    u = x/h
    fr = u - np.floor(u)            # in [0,1)
    shift = (np.abs(fr - 0.5) > 0.25)   # x near an integer node -> use half-shifted grid
    y2 = y*y
    acc0 = np.zeros_like(x); acc1 = np.zeros_like(x)
    for k in range(-N, N+1):
        tn = k*h
        acc0 += np.exp(-tn*tn)/((x - tn)**2 + y2)
    for k in range(-N, N):
        tn = (k+0.5)*h
        acc1 += np.exp(-tn*tn)/((x - tn)**2 + y2)
    acc = np.where(shift, acc1, acc0)
    res = h*y/np.pi*acc
    z = x + 1j*y
    sgn = np.where(shift, 1.0, -1.0)
    with np.errstate(over='ignore', invalid='ignore'):
        E = np.exp(-2j*np.pi*z/h)
        corr = 2.0*np.exp(-z*z)/(1.0 + sgn*E)
    use = y < np.pi/h
    res = res + np.where(use, corr.real, 0.0)
    return res

def exact(x, y):
    mp.mp.dps = 40
    z = mp.mpc(x, y)
    return float((mp.exp(-z*z)*mp.erfc(-1j*z)).real)

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    # tier A
    for smin, smax, nt in [(1e4, 1e5, 4), (1e5, 1e7, 4), (1e5, 1e7, 3), (1e7, 1e16, 3)]:
        n = 200000
        rho = np.sqrt(10**rng.uniform(np.log10(smin), np.log10(smax), n))
        th = rng.uniform(0, np.pi/2, n)
        # also include tiny y
        y = np.where(rng.random(n) < 0.5, rho*np.sin(th), 10**rng.uniform(-8, 0, n))
        x = np.sqrt(np.maximum(rho**2 - y**2, 0))
        ref = wofz(x + 1j*y).real
        a = tierA(x, y, nt)
        print("tierA n=%d s in [%g,%g]: max rel err vs wofz %.3e" % (nt, smin, smax, np.max(np.abs(a/ref - 1))))
    # CF tier
    for J in (6, 8, 10, 12, 16):
        for smin, smax in [(36, 64), (64, 100), (100, 200), (200, 1000), (1000, 1e4)]:
            n = 100000
            rho = np.sqrt(10**rng.uniform(np.log10(smin), np.log10(smax), n))
            th = rng.uniform(0, np.pi/2, n)
            y = np.where(rng.random(n) < 0.5, rho*np.sin(th), 10**rng.uniform(-8, 0, n))
            x = np.sqrt(np.maximum(rho**2 - y**2, 0))
            ref = wofz(x + 1j*y).real
            a = cf(x, y, J)
            print("CF J=%2d s in [%g,%g]: max rel err %.3e" % (J, smin, smax, np.max(np.abs(a/ref - 1))))
    # trapezoid tier
    n = 200000
    x = rng.uniform(0, 12, n)
    y = 10**rng.uniform(-10, 1.1, n)
    ref = wofz(x + 1j*y).real
    for h, N in [(0.5, 13), (0.5, 14), (0.45, 15)]:
        a = trap(x, y, h, N)
        err = np.abs(a/ref - 1)
        i = np.argmax(err)
        print("trap h=%.2f N=%d: max rel err %.3e at x=%.4f y=%.3e" % (h, N, err[i], x[i], y[i]))
