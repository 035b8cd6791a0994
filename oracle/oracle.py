"""ctypes wrapper of the CPU oracle (oracle/cs_oracle.c).  TEST INFRASTRUCTURE ONLY -- see that file's header.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libcs_oracle.so")
CHEB_LD = 16
SHAPES = {"voigt": 0, "lorentz": 1, "doppler": 2, "PHCO2": 3, "phco2": 3}
_dp = C.POINTER(C.c_double)


class _Lines(C.Structure):
    _fields_ = [("L", C.c_int64), ("nu", _dp), ("S", _dp), ("ga", _dp), ("gs", _dp), ("Epp", _dp), ("na", _dp),
                ("mu", _dp), ("iso", C.POINTER(C.c_int16)), ("niso", C.c_int), ("ncheb", C.POINTER(C.c_int32)),
                ("cheb", _dp)]


def build(force=False):
    src = os.path.join(_HERE, "cs_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s", "-B", "libcs_oracle.so"], check=True)
    return LIB


def use_native_build():
    """Rebuild the oracle with -O3 -march=native ON THE HOST IT RUNS ON and load that copy (bench.py's cpu_baseline
    leg, so the CPU number is not handicapped by a generic x86-64 build).  Falls back to the portable build."""
    global _lib
    nat = os.path.join(_HERE, "libcs_oracle_native.so")
    try:
        subprocess.run(["gcc", "-O3", "-march=native", "-fPIC", "-fopenmp", "-std=c99", "-ffp-contract=off", "-shared",
                        "-o", nat, os.path.join(_HERE, "cs_oracle.c"), "-lm"], check=True)
    except Exception:
        return False
    _lib = None
    globals()["LIB"] = nat
    lib()
    return True


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not LIB.endswith("_native.so"):
            build()
        L = C.CDLL(LIB)
        L.cso_faddeeva_re.restype = C.c_double
        L.cso_faddeeva_re.argtypes = [C.c_double, C.c_double]
        L.cso_faddeeva_re_vec.argtypes = [C.c_int64, _dp, _dp, _dp]
        L.cso_chebyQrefQ.restype = C.c_double
        L.cso_chebyQrefQ.argtypes = [C.c_double, C.c_int, _dp, C.POINTER(C.c_int)]
        L.cso_shape_bang.argtypes = [C.c_int, C.c_int, C.c_int64, _dp, C.POINTER(_Lines), C.c_double, C.c_double,
                                     C.c_double, C.c_double, _dp]
        L.cso_planck.restype = C.c_double
        L.cso_planck.argtypes = [C.c_double, C.c_double]
        L.cso_streamnodes.argtypes = [C.c_int, _dp, _dp]
        L.cso_lobattonodes.argtypes = [C.c_int, _dp, _dp]
        L.cso_depth_bang.argtypes = [_dp, C.c_int, _dp, _dp, C.c_int, _dp]
        L.cso_monoflux_bang.argtypes = [_dp, _dp, _dp, C.c_int, _dp, C.c_double, C.c_double, C.c_double, C.c_int, _dp, _dp]
        L.cso_trapz.restype = C.c_double
        L.cso_trapz.argtypes = [C.c_int64, _dp, _dp, C.c_int64]
        L.cso_fluxes_discretized.argtypes = [C.c_int64, _dp, C.c_int, _dp, C.c_double, C.c_int, _dp, _dp, _dp, C.c_int,
                                             C.POINTER(C.POINTER(_Lines)), C.POINTER(C.c_int), _dp, _dp, C.c_double, _dp,
                                             _dp, _dp, C.c_double, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
        L.cso_num_threads.restype = C.c_int
        L.cso_set_faddeeva_backend.argtypes = [C.c_int]
        L.cso_get_faddeeva_backend.restype = C.c_int
        _lib = L
    return _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


class Lines:
    """Holds contiguous copies of a SpectralLines-like object's arrays and the C struct that points at them."""

    def __init__(self, sl):
        self.a = [_f64(getattr(sl, n)) for n in ("nu", "S", "gamma_a", "gamma_s", "Epp", "na", "mu")]
        self.iso = np.ascontiguousarray(sl.I, dtype=np.int16)
        self.ncheb = np.ascontiguousarray(sl.ncheb, dtype=np.int32)
        self.cheb = _f64(sl.cheb)
        assert self.cheb.shape == (len(self.ncheb), CHEB_LD)
        self.c = _Lines(len(self.a[0]), *[_p(x) for x in self.a], self.iso.ctypes.data_as(C.POINTER(C.c_int16)),
                        len(self.ncheb), self.ncheb.ctypes.data_as(C.POINTER(C.c_int32)), _p(self.cheb))


class faddeeva_backend:
    """with faddeeva_backend("alg985"): ...  -- evaluate the oracle with its restatement of ACM TOMS Algorithm 985 (what the
    reference's Faddeyeva985 dependency implements; unverifiable here) instead of the exact function.  Only for quantifying the
    expected disagreement with the Julia reference; every parity test uses the exact back-end."""

    def __init__(self, name):
        self.b = {"exact": 0, "alg985": 1}[name]

    def __enter__(self):
        self.old = lib().cso_get_faddeeva_backend()
        lib().cso_set_faddeeva_backend(self.b)

    def __exit__(self, *a):
        lib().cso_set_faddeeva_backend(self.old)


def faddeeva(x, y):
    x = _f64(np.atleast_1d(x))
    y = _f64(np.atleast_1d(y))
    out = np.zeros_like(x)
    lib().cso_faddeeva_re_vec(len(x), _p(x), _p(y), _p(out))
    return out


def chebyQrefQ(T, a):
    a = _f64(a)
    err = C.c_int(0)
    v = lib().cso_chebyQrefQ(float(T), len(a), _p(a), C.byref(err))
    if err.value:
        raise AssertionError("temperature outside of Qref/Q interpolation range")
    return v


def shape_bang(shape, nu, sl, T, P, Pp, cut=25.0, strict_ends=True):
    """shape!(sigma, nu, sl, T, P, Pp, cut) -> sigma (new array)."""
    nu = _f64(nu)
    L = sl if isinstance(sl, Lines) else Lines(sl)
    out = np.zeros(len(nu))
    rc = lib().cso_shape_bang(SHAPES[shape] if isinstance(shape, str) else shape, int(strict_ends), len(nu), _p(nu),
                              C.byref(L.c), float(T), float(P), float(Pp), float(cut), _p(out))
    if rc:
        raise AssertionError(f"oracle shape! failed with code {rc}")
    return out


def planck(nu, T):
    return np.array([lib().cso_planck(float(v), float(T)) for v in np.atleast_1d(nu)])


def streamnodes(n):
    m, W = np.zeros(n), np.zeros(n)
    lib().cso_streamnodes(n, _p(m), _p(W))
    return m, W


def lobattonodes(n):
    x, w = np.zeros(n), np.zeros(n)
    lib().cso_lobattonodes(n, _p(x), _p(w))
    return x, w


def depth_bang(P, beta, nlobatto):
    P, beta = _f64(P), _f64(beta)
    _, ws = lobattonodes(nlobatto)
    tau = np.zeros(len(P) - 1)
    lib().cso_depth_bang(_p(tau), len(P), _p(P), _p(beta), nlobatto, _p(ws))
    return tau


def monoflux_bang(tau, B, fS, fa, theta_s, nstream):
    tau, B = _f64(tau), _f64(B)
    npl = len(B)
    m, W = streamnodes(nstream)
    Mup, Mdn = np.zeros(npl), np.zeros(npl)
    lib().cso_monoflux_bang(_p(Mup), _p(Mdn), _p(tau), npl, _p(B), float(fS), float(fa), float(theta_s), nstream, _p(m), _p(W))
    return Mup, Mdn


def trapz(x, y):
    x, y = _f64(x), _f64(y)
    return lib().cso_trapz(len(x), _p(x), _p(y), 1)


def fluxes_discretized(nu, P, g, nlobatto, Tn, mun, Tlev, gases, shapes, cuts, conc, sigma_gray=0.0, sigma_extra=None,
                       S_toa=None, albedo=None, theta_s=0.841, nstream=5, want_sigma=False):
    """Whole column on the CPU.  Tn/mun: (nlobatto, nl); conc: (ngas, K).  Returns dict(tau, Mup, Mdn, Fup, Fdn[, sigma])
    with tau (nl, nnu), Mup/Mdn (np, nnu) Fortran order; sigma (K, nnu)."""
    nu, P = _f64(nu), _f64(P)
    nnu, npl = len(nu), len(P)
    nl = npl - 1
    Kn = nl * (nlobatto - 1) + 1
    Tn_ = _f64(np.asarray(Tn).ravel(order="F"))
    mun_ = _f64(np.asarray(mun).ravel(order="F"))
    Tlev = _f64(Tlev)
    Ls = [g_ if isinstance(g_, Lines) else Lines(g_) for g_ in gases]
    ng = len(Ls)
    arr = (C.POINTER(_Lines) * max(ng, 1))(*[C.pointer(l.c) for l in Ls])
    sh = (C.c_int * max(ng, 1))(*[SHAPES[s] if isinstance(s, str) else int(s) for s in shapes])
    cuts_ = _f64(cuts if ng else [0.0])
    conc_ = _f64(np.asarray(conc, float).reshape(ng, Kn).ravel(order="F")) if ng else _f64([0.0])
    ex = None if sigma_extra is None else _f64(sigma_extra)
    S = None if S_toa is None else _f64(S_toa)
    al = None if albedo is None else _f64(albedo)
    tau = np.zeros((nl, nnu), order="F")
    Mup = np.zeros((npl, nnu), order="F")
    Mdn = np.zeros((npl, nnu), order="F")
    Fup, Fdn = np.zeros(npl), np.zeros(npl)
    sig = np.zeros((Kn, nnu)) if want_sigma else None
    rc = lib().cso_fluxes_discretized(nnu, _p(nu), npl, _p(P), float(g), nlobatto, _p(Tn_), _p(mun_), _p(Tlev), ng, arr,
                                      sh, _p(cuts_), _p(conc_), float(sigma_gray), _p(ex), _p(S), _p(al), float(theta_s),
                                      nstream, tau.ctypes.data_as(_dp), Mup.ctypes.data_as(_dp), Mdn.ctypes.data_as(_dp),
                                      _p(Fup), _p(Fdn), _p(sig))
    if rc:
        raise AssertionError(f"oracle fluxes failed with code {rc}")
    out = dict(tau=tau, Mup=Mup, Mdn=Mdn, Fup=Fup, Fdn=Fdn)
    if want_sigma:
        out["sigma"] = sig
    return out


def num_threads():
    return lib().cso_num_threads()


# ---- opacity tables (gases.jl:68-145): numpy restatement, test infrastructure only ------------------------------------

def bake(sl, conc, nu, Tgrid, Pgrid, shape="voigt", cut=25.0):
    """bake + OpacityTable: ln(sigma) of shape [nnu, nT, nP]; conc[i, j] = fC(T_i, P_j).  Zero-row scrub gases.jl:132-142,
    ln(floatmin) for all-zero rows gases.jl:76-79."""
    nu = _f64(nu)
    L = sl if isinstance(sl, Lines) else Lines(sl)
    sig = np.zeros((len(nu), len(Tgrid), len(Pgrid)))
    for i, T in enumerate(Tgrid):
        for j, P in enumerate(Pgrid):
            sig[:, i, j] = shape_bang(shape, nu, L, T, P, conc[i, j] * P, cut)      # shape!(sigma_ij, nu, sl, T, P, C*P, cut) :126
    flat = sig.reshape(len(nu), -1)
    z = (flat.min(axis=1) == 0) & (flat.max(axis=1) > 0)
    sig[z] = 0.0
    tiny = np.finfo(float).tiny
    out = np.empty_like(sig)
    for n in range(len(nu)):
        out[n] = np.log(tiny) if np.all(sig[n] <= tiny) else np.log(sig[n])
    return out


def cheb_basis(x, v):
    """Lagrange basis on Chebyshev extrema x (ascending) at v, barycentric form (the unique interpolating polynomial
    that BichebyshevInterpolator evaluates, gases.jl:80,85)."""
    x = np.asarray(x, float)
    n = len(x)
    hit = np.nonzero(x == v)[0]
    if len(hit):
        l = np.zeros(n)
        l[hit[0]] = 1.0
        return l
    w = np.where(np.arange(n) % 2 == 1, -1.0, 1.0)
    w[0] *= 0.5
    w[-1] *= 0.5
    t = w / (v - x)
    return t / t.sum()


def table_sigma(lnsig, Tgrid, Pgrid, T, P):
    """OpacityTable functor: exp(Phi(T, ln P)) for every wavenumber (gases.jl:85)"""
    a = cheb_basis(Tgrid, T)
    b = cheb_basis(np.log(Pgrid), np.log(P))
    return np.exp(np.einsum("nij,i,j->n", lnsig, a, b))


# ---- collision-induced absorption (collision_induced_absorption.jl:161-303): numpy restatement, test infrastructure -----

def cia_sigma(data, nu, T, Pa, P1, P2, extrapolate=False, singles=False):
    """cia(nu, CIATables(data), T, Pa, P1, P2) for a vector of wavenumbers.  `data` = list of dicts as produced by a .cia
    reader (keys numin, numax, T, nu, k)."""
    nu = np.asarray(nu, float)
    ranges = sorted(set((d["numin"], d["numax"]) for d in data))
    ktot = np.zeros(len(nu))
    for lo, hi in ranges:
        sel = sorted([d for d in data if np.isclose(d["numin"], lo) and np.isclose(d["numax"], hi)], key=lambda d: d["T"])
        g = sel[0]["nu"]
        inside = (nu >= g[0]) & (nu <= g[-1])
        if not inside.any():
            continue
        v = nu[inside]
        i = np.clip(np.searchsorted(g, v, side="right") - 1, 0, len(g) - 2)
        xx = (v - g[i]) / (g[i + 1] - g[i])
        if len(sel) == 1:
            if not singles:
                continue
            k = sel[0]["k"].copy()
            k[k <= 0] = 0.0
            with np.errstate(divide="ignore", invalid="ignore"):
                ln = np.log(k)
                ktot[inside] += np.exp((v - g[i]) * (ln[i + 1] - ln[i]) / (g[i + 1] - g[i]) + ln[i])
            continue
        Tg = np.array([d["T"] for d in sel])
        if Tg[0] <= T <= Tg[-1]:
            Te = T
        elif extrapolate:
            Te = Tg[-1] if T > Tg[-1] else Tg[0]
        else:
            continue
        z = np.array([d["k"] for d in sel], float)
        z[z <= 0] = np.finfo(float).tiny
        z = np.log(z)
        j = int(np.clip(np.searchsorted(Tg, Te, side="right") - 1, 0, len(Tg) - 2))
        yy = (Te - Tg[j]) / (Tg[j + 1] - Tg[j])
        ktot[inside] += np.exp((1 - xx) * (1 - yy) * z[j, i] + xx * (1 - yy) * z[j, i + 1] + (1 - xx) * yy * z[j + 1, i]
                               + xx * yy * z[j + 1, i + 1])
    rho1 = (P1 / 101325.0) * (273.15 / T)
    rho2 = (P2 / 101325.0) * (273.15 / T)
    rhoa = 1e-6 * Pa / (1.38064852e-23 * T)
    return (ktot * 7.21879268e38) * rho1 * rho2 / rhoa
