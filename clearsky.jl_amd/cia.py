"""Collision-induced absorption: .cia reader, CIATables and the cia() cross-section formula (host side).

Mirrors reference src/absorption/collision_induced_absorption.jl: readcia :39-94, CIATables :145-235, the functor :251-276,
cia :295-303,:318-323.  Inside a Column the tables are evaluated by the k_cia kernel; the numpy paths here serve scalar
calls and tests.
"""
import math

import numpy as np

from . import constants as K


def readcia(filename: str):
    """readcia(filename) :39-94 -> list of dicts (symbol, numin, numax, npts, T, maxcia, res, comments, reference, nu, k)"""
    assert filename.endswith(".cia"), "expected file with .cia extension downloaded from https://hitran.org/cia/"
    with open(filename) as f:
        lines = [ln.rstrip("\n").rstrip("\r") for ln in f]
    while lines and lines[-1] == "":
        lines.pop()
    L = [len(ln) for ln in lines]
    assert max(L) == 100, f"unexpected maximum line length in cia file, expected 100 but got {max(L)}"
    hidx = [i for i, n in enumerate(L) if n == 100] + [len(lines)]
    data = []
    for a, b in zip(hidx[:-1], hidx[1:]):
        h = lines[a]
        d = dict(symbol=h[0:20].strip(), numin=float(h[20:30]), numax=float(h[30:40]), npts=int(h[40:47]), T=float(h[47:54]),
                 maxcia=float(h[54:64]), res=float(h[64:70]), comments=h[70:97].strip(), reference=int(h[97:100]))
        rows = [ln.split() for ln in lines[a + 1:b]]
        d["nu"] = np.array([float(r[0]) for r in rows])
        d["k"] = np.array([float(r[1]) for r in rows])
        data.append(d)
    return data


def cia(*args):
    """cia(k, T, Pa, P1, P2) :295-303  or  cia(nu, tables, T, Pa, P1, P2) :318-323 -> cross-section [cm^2/molecule]"""
    if len(args) == 6:
        nu, x, T, Pa, P1, P2 = args
        return cia(x(nu, T), T, Pa, P1, P2)
    k, T, Pa, P1, P2 = args
    rho1 = (P1 / K.atm) * (K.T0 / T)
    rho2 = (P2 / K.atm) * (K.T0 / T)
    rhoa = 1e-6 * Pa / (K.k * T)
    return (k * K.Lo2) * rho1 * rho2 / rhoa


class CIATables:
    """CIATables(data_or_filename; extrapolate=False, singles=False) :145-242.

    `grids`: list of (nu[nb], T[nt], lnk[nt, nb]) -- bilinear interpolation of ln k (BilinearInterpolator, NoBoundaries);
    `single`: list of (nu, lnk, T) single-temperature ranges (LinearInterpolator).  Callable: tables(nu, T) -> k.
    """

    def __init__(self, data, extrapolate: bool = False, singles: bool = False, verbose: bool = False):
        self.filename = data if isinstance(data, str) else None
        if isinstance(data, str):
            data = readcia(data)
        ranges = sorted(set((d["numin"], d["numax"]) for d in data), key=lambda r: r[0])
        self.grids, self.single = [], []
        for lo, hi in ranges:
            sel = [d for d in data if math.isclose(d["numin"], lo) and math.isclose(d["numax"], hi)]
            if len(sel) == 1:
                k = sel[0]["k"].copy()
                k[k <= 0.0] = 0.0
                with np.errstate(divide="ignore"):
                    self.single.append((sel[0]["nu"].copy(), np.log(k), float(sel[0]["T"])))
            else:
                for d in sel[1:]:
                    assert math.isclose(float(np.sum(sel[0]["nu"] - d["nu"])), 0.0, abs_tol=1e-12), \
                        "wavenumber sample within a wavenumber range appear to be different"
                sel = sorted(sel, key=lambda d: d["T"])
                k = np.array([d["k"] for d in sel], dtype=float)         # [nt, nb]
                k[k <= 0.0] = np.finfo(float).tiny
                self.grids.append((sel[0]["nu"].copy(), np.array([d["T"] for d in sel], float), np.log(k)))
        symbols = sorted(set(d["symbol"] for d in data))
        assert len(symbols) == 1
        self.name = symbols[0]
        self.formulae = tuple(self.name.split("-"))
        self.extrapolate, self.singles = bool(extrapolate), bool(singles)
        if verbose:
            print(f"creating CIATables\n  formulae: {self.formulae[0]} & {self.formulae[1]}\n  {len(self.grids) + len(self.single)} absorption region(s)")

    def __call__(self, nu, T):
        """tables(nu, T) :251-276 (scalar nu)"""
        k = 0.0
        for g_nu, g_T, lnk in self.grids:
            if g_nu[0] <= nu <= g_nu[-1]:
                if g_T[0] <= T <= g_T[-1]:
                    k += math.exp(_bilinear(g_nu, g_T, lnk, nu, T))
                elif self.extrapolate:
                    k += math.exp(_bilinear(g_nu, g_T, lnk, nu, g_T[-1] if T > g_T[-1] else g_T[0]))
        if self.singles:
            for s_nu, s_lnk, _ in self.single:
                if s_nu[0] <= nu <= s_nu[-1]:
                    i = min(max(int(np.searchsorted(s_nu, nu, side="right")) - 1, 0), len(s_nu) - 2)
                    with np.errstate(invalid="ignore"):
                        k += math.exp((nu - s_nu[i]) * (s_lnk[i + 1] - s_lnk[i]) / (s_nu[i + 1] - s_nu[i]) + s_lnk[i])
        return k

    def __repr__(self):
        return f"CIATables - {self.name}"


def _bilinear(xg, yg, z, x, y):
    i = min(max(int(np.searchsorted(xg, x, side="right")) - 1, 0), len(xg) - 2)
    j = min(max(int(np.searchsorted(yg, y, side="right")) - 1, 0), len(yg) - 2)
    xx = (x - xg[i]) / (xg[i + 1] - xg[i])
    yy = (y - yg[j]) / (yg[j + 1] - yg[j])
    return (1 - xx) * (1 - yy) * z[j, i] + xx * (1 - yy) * z[j, i + 1] + (1 - xx) * yy * z[j + 1, i] + xx * yy * z[j + 1, i + 1]
