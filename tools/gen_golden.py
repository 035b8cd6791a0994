#!/usr/bin/env python3
"""Generate tests/golden/*.npz -- the build's own golden vectors for the hot path.

The reference (Julia) cannot run here and its tests hold no vectors for this path.  These goldens are
produced by an INDEPENDENT numpy/scipy/mpmath restatement of the reference formulas (scipy.special.wofz as the exact
Faddeeva function), sharing no code with oracle/ or the HIP library; they pin both.  Reference lines restated:
  line_shapes.jl:5,10,18-22,27-48,53-87,107-123,144,160,255-257,273,366-378,467-481; radiation.jl:48-54;
  core/shared.jl:4-21,125-137; core/discretized.jl:2-9,85-87,136-177,249-326; util.jl:19-33; atmospherics.jl:16-26.
Run:  python tools/gen_golden.py        (container only; inputs = tests/golden/hitran/*.par + data/molparam.json)
"""
import json
import math
import os

import mpmath as mp
import numpy as np
from numpy.polynomial.legendre import leggauss
from scipy.special import wofz

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

c, h, k = 299792458.0, 6.62607015e-34, 1.38064852e-23
Rg, atm, Na, Tref = 8.31446262, 101325.0, 6.02214076e23, 296.0
c2 = 100.0 * h * c / k
ISO = {ch: i + 1 for i, ch in enumerate("1234567890ABCDEFGHIJKLMNOPQRSTUVWXYZ")}


def load_par(path):
    rows = [ln for ln in open(path, "rb").read().split(b"\n") if ln.strip()]
    d = dict(M=np.array([int(r[0:2]) for r in rows]), I=np.array([ISO[chr(r[2])] for r in rows]),
             nu=np.array([float(r[3:15]) for r in rows]), S=np.array([float(r[15:25]) for r in rows]),
             ga=np.array([float(r[35:40]) for r in rows]), gs=np.array([float(r[40:45]) for r in rows]),
             Epp=np.array([float(r[45:55]) for r in rows]), na=np.array([float(r[55:59]) for r in rows]))
    idx = np.argsort(d["nu"], kind="stable")
    d = {k_: v[idx] for k_, v in d.items()}
    mol = json.load(open(os.path.join(ROOT, "clearsky.jl_amd", "data", "molparam.json")))["molecules"][int(d["M"][0]) - 1]
    d["mu"] = np.array(mol["mu"])[d["I"] - 1]
    d["cheb"] = [np.array(cf) for cf in mol["cheb"]]
    return d


def qrefq(T, a):
    tau = 2 * (T - 25.0) / (1000.0 - 25.0) - 1
    c1, c2_ = 1.0, tau
    y = a[0] + a[1] * c2_
    for kk in range(2, len(a)):
        c3 = 2 * tau * c2_ - c1
        y += a[kk] * c3
        c1, c2_ = c2_, c3
    return 1.0 / y


def line_params(d, T, P, Pp):
    a = -c2 * d["Epp"]
    b = -c2 * d["nu"]
    n = np.exp(a / T) * (1 - np.exp(b / T))
    dd = np.exp(a / Tref) * (1 - np.exp(b / Tref))
    Q = np.array([qrefq(T, d["cheb"][i - 1]) for i in d["I"]])
    S = d["S"] * Q * (n / dd)
    alpha = (d["nu"] / c) * np.sqrt(2.0 * Rg * T / d["mu"])
    gamma = ((Tref / T) ** d["na"]) * (d["ga"] * (P - Pp) + d["gs"] * Pp) / atm
    return S, alpha, gamma


def chi_phco2(dn, T):
    B1 = 0.0888 - 0.16 * math.exp(-0.0041 * T)
    B2 = 0.0526 * math.exp(-0.00152 * T)
    out = np.ones_like(dn)
    m = (dn >= 3) & (dn < 30); out[m] = np.exp(-B1 * (dn[m] - 3.0))
    m = (dn >= 30) & (dn < 120); out[m] = np.exp(-B1 * 27.0 - B2 * (dn[m] - 30.0))
    m = dn >= 120; out[m] = np.exp(-B1 * 27.0 - B2 * 90.0 - 0.0232 * (dn[m] - 120.0))
    return out


def shape_sigma(shape, nu, d, T, P, Pp, cut):
    """sum over lines with |nu - nul| <= cut and (vector pre-filter) min(nu)-cut < nul < max(nu)+cut"""
    S, al, gm = line_params(d, T, P, Pp)
    keep = (d["nu"] > nu.min() - cut) & (d["nu"] < nu.max() + cut)
    nul, S, al, gm = d["nu"][keep], S[keep], al[keep], gm[keep]
    out = np.zeros(len(nu))
    for i, v in enumerate(nu):
        m = ~(np.abs(v - nul) > cut)
        dv = v - nul[m]
        if shape == "voigt" or shape == "PHCO2":
            g_ = gm[m] * (chi_phco2(np.abs(dv), T) if shape == "PHCO2" else 1.0)
            beta = 1 / al[m]
            dd = math.sqrt(math.log(2.0)) * beta
            f = wofz(dv * dd + 1j * g_ * dd).real
            out[i] = np.sum(S[m] * ((1 / math.sqrt(math.pi / math.log(2.0))) * beta * f))
        elif shape == "lorentz":
            out[i] = np.sum(S[m] * gm[m] / (math.pi * (dv * dv + gm[m] ** 2)))
        else:
            out[i] = np.sum(S[m] * np.exp(-dv ** 2 / al[m] ** 2) / (al[m] * math.sqrt(math.pi)))
    return out


def planck(nu, T):
    num = 100.0 * nu
    return 100.0 * (2 * h * c ** 2 * num ** 3) / (np.exp(h * c * num / (k * T)) - 1.0)


def streamnodes(n):
    x, w = leggauss(n)
    th = (math.pi / 2) * (x + 1) / 2
    return 1 / np.cos(th), 2 * math.pi * ((math.pi / 2) * w / 2) * np.cos(th) * np.sin(th)


def lobatto01(n):
    if n == 2:
        x, w = np.array([-1.0, 1.0]), np.array([1.0, 1.0])
    elif n == 3:
        x, w = np.array([-1.0, 0.0, 1.0]), np.array([1 / 3, 4 / 3, 1 / 3])
    elif n == 4:
        x, w = np.array([-1.0, -1 / math.sqrt(5), 1 / math.sqrt(5), 1.0]), np.array([1 / 6, 5 / 6, 5 / 6, 1 / 6])
    elif n == 5:
        r = math.sqrt(3 / 7)
        x, w = np.array([-1.0, -r, 0.0, r, 1.0]), np.array([0.1, 49 / 90, 32 / 45, 49 / 90, 0.1])
    else:
        raise ValueError(n)
    return (x + 1) / 2, w / 2


def chebygrid(a, b, n):
    return (np.cos(np.pi * np.arange(n - 1, -1, -1) / (n - 1)) + 1) * (b - a) / 2 + a


def interp_logP(P, y, p):
    x, xs = math.log(p), np.log(P)
    i = min(max(int(np.searchsorted(xs, x, side="right")) - 1, 0), len(P) - 2)
    return (x - xs[i]) * (y[i + 1] - y[i]) / (xs[i + 1] - xs[i]) + y[i]


def column(nu, P, g, nlob, Tlevels_vec, mu, sigma_at, fS, fa, theta_s, nstream):
    """fluxes.jl:238-279 + shared.jl:125-137 with sigma_at(k, T, P) -> total cross-section vector at node k"""
    npl, nl = len(P), len(P) - 1
    xs, ws = lobatto01(nlob)
    m, W = streamnodes(nstream)
    C = 1e-4 * Na / g
    fT = lambda p: interp_logP(P, Tlevels_vec, p)
    K_ = nl * (nlob - 1) + 1
    Pk, Tk = np.zeros(K_), np.zeros(K_)
    Pk[0], Tk[0] = P[0], fT(P[0] + (P[1] - P[0]) * xs[0])
    for i in range(nl):
        dP = P[i + 1] - P[i]
        for n in range(1, nlob):
            kk = i * (nlob - 1) + n
            Pn = P[i] + dP * xs[n]
            Tk[kk] = fT(Pn)
            Pk[kk] = P[i + 1] if n == nlob - 1 else Pn
    sig = np.array([sigma_at(kk, Tk[kk], Pk[kk]) for kk in range(K_)])      # (K, nnu)
    beta = C * (sig / mu)
    tau = np.zeros((nl, len(nu)))
    for i in range(nl):
        dP = P[i + 1] - P[i]
        t = (dP * ws[0]) * beta[i * (nlob - 1)]
        for n in range(1, nlob):
            t = t + (dP * ws[n]) * beta[i * (nlob - 1) + n]
        tau[i] = np.maximum(t, 1e-6)
    B = np.array([planck(nu, fT(p)) for p in P])                               # (np, nnu)
    Mup, Mdn = np.zeros((npl, len(nu))), np.zeros((npl, len(nu)))
    cth = math.cos(theta_s)
    lp = lambda B1, B2, t_, tr: B2 * (1 - tr) - (B1 - B2) * tr + (1 - tr) * (B1 - B2) / t_
    for kk in range(nstream):
        I = np.zeros(len(nu))
        for i in range(nl):
            ti = tau[i] * m[kk]
            tr = np.exp(-ti)
            I = I * tr + lp(B[i], B[i + 1], ti, tr)
            Mdn[i + 1] += W[kk] * I
    Mdn[0] += cth * fS
    Ms = Mdn[0].copy()
    for i in range(nl):
        Ms = Ms * np.exp(-tau[i] / cth)
        Mdn[i + 1] += Ms
    Is = Mdn[-1] * fa / math.pi + B[-1]
    Mup[-1] = Is * math.pi
    for kk in range(nstream):
        I = Is.copy()
        for i in range(nl - 1, -1, -1):
            ti = tau[i] * m[kk]
            tr = np.exp(-ti)
            I = I * tr + lp(B[i + 1], B[i], ti, tr)
            Mup[i] += W[kk] * I
    trap = lambda y: float(np.sum((nu[1:] - nu[:-1]) * (y[:-1] + y[1:]) / 2))
    Fup = np.array([trap(Mup[i]) for i in range(npl)])
    Fdn = np.array([trap(Mdn[i]) for i in range(npl)])
    return dict(Pk=Pk, Tk=Tk, sigma=sig, tau=tau, Mup=Mup, Mdn=Mdn, Fup=Fup, Fdn=Fdn)


def main():
    os.makedirs(OUT, exist_ok=True)
    # 1. Faddeeva known answers (40-digit mpmath)
    mp.mp.dps = 40
    rng = np.random.Generator(np.random.PCG64(7))
    xs = np.concatenate([rng.uniform(0, 11, 150), 10 ** rng.uniform(1, 7, 100), [0.0, 0.25, 0.5, 6.0, 10.0, 99.9, 100.1]])
    ys = np.concatenate([10 ** rng.uniform(-12, 1.1, 150), 10 ** rng.uniform(-8, 3, 100), [0.0, 1e-8, 1e-3, 1e-6, 1e-9, 5.0, 5.0]])
    w = np.array([float((mp.exp(-mp.mpc(a, b) ** 2) * mp.erfc(-1j * mp.mpc(a, b))).real) for a, b in zip(xs, ys)])
    np.savez(os.path.join(OUT, "faddeeva.npz"), x=xs, y=ys, w=w)

    # 2. line shapes on the reference's HITRAN fixtures
    co2 = load_par(os.path.join(OUT, "hitran", "CO2.par"))
    h2o = load_par(os.path.join(OUT, "hitran", "H2O.par"))
    states = [(220.0, 50.0, 50.0 * 400e-6), (296.0, 101325.0, 101325.0 * 400e-6), (260.0, 3e3, 3e3 * 0.01)]
    nu_co2 = np.concatenate([np.linspace(640.0, 700.0, 192), np.linspace(2300.0, 2380.0, 64)])
    nu_h2o = np.linspace(1400.0, 1700.0, 256)
    out = dict(states=np.array(states), nu_co2=nu_co2, nu_h2o=nu_h2o)
    for nm, d, nu in (("co2", co2, nu_co2), ("h2o", h2o, nu_h2o)):
        out[f"voigt_{nm}"] = np.array([shape_sigma("voigt", nu, d, *s, 25.0) for s in states])
    out["lorentz_co2"] = np.array([shape_sigma("lorentz", nu_co2, co2, *s, 25.0) for s in states])
    out["doppler_co2"] = np.array([shape_sigma("doppler", nu_co2, co2, *s, 25.0) for s in states])
    out["phco2_co2"] = np.array([shape_sigma("PHCO2", nu_co2, co2, *s, 500.0) for s in states])
    # cut-off edge semantics: a 3-point grid whose ends sit exactly one cut-off away from a line
    l0 = float(co2["nu"][np.argmin(np.abs(co2["nu"] - 667.0))])
    nu_edge = np.array([l0 - 25.0, l0 - 12.5, l0 + 25.0])
    out["nu_edge"] = nu_edge
    out["voigt_edge"] = shape_sigma("voigt", nu_edge, co2, *states[1], 25.0)
    np.savez(os.path.join(OUT, "lineshapes.npz"), **out)

    # 3. columns: gray 20 layers (config 1 plumbing) and CO2 40 layers (config 2 shape, 256 wavenumbers)
    P20 = np.exp(chebygrid(math.log(1e-3), math.log(1e5), 21))
    Tgray = 300.0 * (P20 / 1e5) ** (Rg / (0.01 * 1e3))
    nug = np.concatenate([((10.0 ** np.linspace(0, 4, 400)) - 1) * (1e5 - 1e-6) / (10.0 ** 4 - 1) + 1e-6])
    gray = column(nug, P20, 10.0, 2, Tgray, 0.01, lambda kk, T, P: np.full(len(nug), 1e-26), 0.0, 0.0, 0.841, 5)
    np.savez(os.path.join(OUT, "column_gray.npz"), nu=nug, P=P20, T=Tgray, g=10.0, mu=0.01, sigma=1e-26, nstream=5,
             nlobatto=2, tau=gray["tau"], Mup=gray["Mup"], Mdn=gray["Mdn"], Fup=gray["Fup"], Fdn=gray["Fdn"])
    P40 = np.exp(chebygrid(math.log(1.0), math.log(1e5), 41))
    T40 = np.maximum(288.0 * (P40 / 1e5) ** (Rg / (0.029 * 1040.0)), 200.0)
    nuc = np.linspace(600.0, 760.0, 256)
    for nlob, tag in ((2, "co2"), (4, "co2_lob4")):
        Pp = (P40 if nlob == 2 else P40[::4])
        Tp = (T40 if nlob == 2 else T40[::4])
        # scalar-nu semantics in a column: every line within the cut-off of the point, no end-point pre-filter
        def sig(kk, T, P, _nu=nuc):
            S, al, gm = line_params(co2, T, P, 400e-6 * P)
            o = np.zeros(len(_nu))
            for i, v in enumerate(_nu):
                m = ~(np.abs(v - co2["nu"]) > 25.0)
                dv = v - co2["nu"][m]
                beta = 1 / al[m]
                dd = math.sqrt(math.log(2.0)) * beta
                o[i] = np.sum(S[m] * ((1 / math.sqrt(math.pi / math.log(2.0))) * beta * wofz(dv * dd + 1j * gm[m] * dd).real))
            return 400e-6 * o
        r = column(nuc, Pp, 9.8, nlob, Tp, 0.029, sig, 30.0 if nlob == 4 else 0.0, 0.3 if nlob == 4 else 0.0, 0.841,
                   5 if nlob == 2 else 3)
        np.savez(os.path.join(OUT, f"column_{tag}.npz"), nu=nuc, P=Pp, T=Tp, g=9.8, mu=0.029, conc=400e-6, nlobatto=nlob,
                 nstream=5 if nlob == 2 else 3, fS=30.0 if nlob == 4 else 0.0, fa=0.3 if nlob == 4 else 0.0,
                 sigma=r["sigma"], tau=r["tau"], Mup=r["Mup"], Mdn=r["Mdn"], Fup=r["Fup"], Fdn=r["Fdn"], Pk=r["Pk"], Tk=r["Tk"])
    # 4. quadrature rules
    np.savez(os.path.join(OUT, "quadrature.npz"), **{f"m{n}": streamnodes(n)[0] for n in (1, 2, 3, 5, 8, 16)},
             **{f"W{n}": streamnodes(n)[1] for n in (1, 2, 3, 5, 8, 16)}, **{f"lx{n}": lobatto01(n)[0] for n in (2, 3, 4, 5)},
             **{f"lw{n}": lobatto01(n)[1] for n in (2, 3, 4, 5)})
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
