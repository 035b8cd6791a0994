"""HITRAN .par reader, SpectralLines container and the MOLPARAM table (host side, numpy).

Mirrors reference src/hitran/par.jl:1-286 (readpar :91-193, SpectralLines :224-286, ISOINDEX :6-13) and the
data table src/hitran/molparam.jl (extracted to data/molparam.json by tools/gen_molparam.py).
"""
import json
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "molparam.json")
CHEB_LD = 16

# par.jl:6-13
ISOINDEX = {ch: i + 1 for i, ch in enumerate("1234567890ABCDEFGHIJKLMNOPQRSTUVWXYZ")}


@dataclass
class MolParam:
    """par.jl:18-45"""
    M: int
    formula: str
    name: str
    I: list
    isoform: list
    AFGL: list
    A: np.ndarray
    mu: np.ndarray
    Qref: np.ndarray
    hascheb: np.ndarray
    ncheb: np.ndarray
    maxrelerr: np.ndarray
    cheb: list

    def cheb_table(self) -> np.ndarray:
        """[niso, CHEB_LD] row-major table of Chebyshev coefficients (zero padded)."""
        t = np.zeros((len(self.ncheb), CHEB_LD))
        for i, c in enumerate(self.cheb):
            t[i, : len(c)] = c
        return t

    def ncheb_table(self) -> np.ndarray:
        """[niso] int32; 0 where hascheb is false (line_shapes.jl:115-120 throws for those)."""
        return np.where(self.hascheb, self.ncheb, 0).astype(np.int32)


def _load_molparam():
    with open(_DATA) as f:
        d = json.load(f)
    mols = {}
    for m in d["molecules"]:
        if m is None:
            continue
        mols[m["M"]] = MolParam(
            M=m["M"], formula=m["formula"], name=m["name"], I=m["I"], isoform=m["isoform"], AFGL=m["AFGL"],
            A=np.array(m["A"], float), mu=np.array(m["mu"], float), Qref=np.array(m["Qref"], float),
            hascheb=np.array(m["hascheb"], bool), ncheb=np.array(m["ncheb"], np.int64),
            maxrelerr=np.array(m["maxrelerr"], float), cheb=[np.array(c, float) for c in m["cheb"]])
    return d["TMIN"], d["TMAX"], mols


TMIN, TMAX, MOLPARAM = _load_molparam()

_FIELDS = ("M", "I", "nu", "S", "A", "gamma_a", "gamma_s", "Epp", "na", "delta_a")


def readpar(filename: str, numin: float = 0.0, numax: float = np.inf, Scut: float = 0.0, I: Sequence = (),
            maxlines: int = -1, native: bool = True) -> dict:
    """Read a HITRAN 160-column .par file (par.jl:91-193; columns :131-149).

    Filters (nu range, intensity cut, isotopologues, strongest `maxlines`) and the final stable sort by wavenumber
    follow par.jl:153-191.  Returns a dict of numpy arrays keyed M, I (characters), nu, S, A, gamma_a, gamma_s, Epp,
    na, delta_a.
    """
    if not filename.endswith(".par"):
        raise AssertionError("expected file with .par extension, downloaded from https://hitran.org/lbl/")
    par = _readpar_native(filename) if native else None    # cs_par_parse: mmap + threads (include/clearsky_hip.h)
    if par is not None:
        return _filter_sort(par, len(par["nu"]), numin, numax, Scut, I, maxlines)
    with open(filename, "rb") as f:
        raw = f.read().split(b"\n")
    raw = [ln.rstrip(b"\r") for ln in raw if len(ln.strip()) > 0]
    N = len(raw)
    par = dict(M=np.zeros(N, np.int16), I=np.empty(N, "U1"), nu=np.zeros(N), S=np.zeros(N), A=np.zeros(N),
               gamma_a=np.zeros(N), gamma_s=np.zeros(N), Epp=np.zeros(N), na=np.zeros(N), delta_a=np.zeros(N))
    for i, ln in enumerate(raw):
        par["M"][i] = int(ln[0:2])
        par["I"][i] = chr(ln[2])
        par["nu"][i] = float(ln[3:15])
        par["S"][i] = float(ln[15:25])
        par["A"][i] = float(ln[25:35])
        par["gamma_a"][i] = float(ln[35:40])
        par["gamma_s"][i] = float(ln[40:45])
        par["Epp"][i] = float(ln[45:55])
        par["na"][i] = float(ln[55:59])
        par["delta_a"][i] = float(ln[59:67])
    return _filter_sort(par, N, numin, numax, Scut, I, maxlines)


def _readpar_native(filename):
    """Parsing loop in the native library (cs_par_parse: mmap + threads); None when the library is not built."""
    import ctypes as C
    from . import _lib
    if not os.path.exists(_lib.LIB_PATH):
        return None
    L = _lib.lib()
    n = C.c_int64()
    _lib.check(L.cs_par_count(filename.encode(), C.byref(n)))
    N = n.value
    par = dict(M=np.zeros(N, np.int16), I=np.zeros(N, "S1"), nu=np.zeros(N), S=np.zeros(N), A=np.zeros(N),
               gamma_a=np.zeros(N), gamma_s=np.zeros(N), Epp=np.zeros(N), na=np.zeros(N), delta_a=np.zeros(N))
    d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    _lib.check(L.cs_par_parse(filename.encode(), N, par["M"].ctypes.data_as(C.POINTER(C.c_int16)),
                              par["I"].ctypes.data_as(C.c_char_p), d(par["nu"]), d(par["S"]), d(par["A"]), d(par["gamma_a"]),
                              d(par["gamma_s"]), d(par["Epp"]), d(par["na"]), d(par["delta_a"])))
    par["I"] = par["I"].astype("U1")
    return par


def _filter_sort(par, N, numin, numax, Scut, I, maxlines):
    """par.jl:153-191: masks, strongest-N selection, stable sort by wavenumber"""
    mask = (par["nu"] >= numin) & (par["nu"] <= numax) & (par["S"] >= Scut)
    if len(I) > 0:
        keep = set(I)
        for j in range(N):
            ch = par["I"][j]
            if (ch not in keep) and (ISOINDEX[ch] not in keep):
                mask[j] = False
    if not mask.any():
        raise AssertionError("par information has been filtered to nothing!")
    par = {k: v[mask] for k, v in par.items()}
    if maxlines > 0 and N > maxlines:  # par.jl:177-186 (N is the unfiltered count there too)
        idx = np.argsort(par["S"], kind="stable")[::-1][:maxlines]
        par = {k: v[idx] for k, v in par.items()}
    idx = np.argsort(par["nu"], kind="stable")
    return {k: v[idx] for k, v in par.items()}


class SpectralLines:
    """Line table of a single gas, sorted by wavenumber (par.jl:224-286).

    Fields (reference names in brackets): name, formula, N, M, I [I], mu [mu], A [A], nu [nu], S, gamma_a [gamma_a],
    gamma_s [gamma_s], Epp, na.  `SpectralLines(filename, **kw)` reads a .par file; `SpectralLines(par_dict)` wraps
    the output of readpar.  The pressure-shift column is parsed but dropped, as in the reference (quirk 2).
    """

    def __init__(self, src, **kwargs):
        par = readpar(src, **kwargs) if isinstance(src, str) else src
        nu = np.asarray(par["nu"], float)
        N = len(nu)
        Ms = np.unique(par["M"])
        if len(Ms) != 1:
            raise AssertionError("SpectralLines objects must contain only one molecule's lines")
        M = int(Ms[0])
        mp = MOLPARAM[M]
        I = np.array([ISOINDEX[str(c)] if not isinstance(c, (int, np.integer)) else int(c) for c in par["I"]],
                     dtype=np.int16)
        idx = np.argsort(nu, kind="stable")
        self.name, self.formula, self.N, self.M = mp.name, mp.formula, N, M
        self.I = I[idx]
        self.mu = mp.mu[self.I - 1].astype(float)
        self.A = mp.A[self.I - 1].astype(float)
        self.nu = np.ascontiguousarray(nu[idx])
        self.S = np.ascontiguousarray(np.asarray(par["S"], float)[idx])
        self.gamma_a = np.ascontiguousarray(np.asarray(par["gamma_a"], float)[idx])
        self.gamma_s = np.ascontiguousarray(np.asarray(par["gamma_s"], float)[idx])
        self.Epp = np.ascontiguousarray(np.asarray(par["Epp"], float)[idx])
        self.na = np.ascontiguousarray(np.asarray(par["na"], float)[idx])
        self.ncheb = mp.ncheb_table()
        self.cheb = mp.cheb_table()

    # aliases with the reference's Greek field names
    ν = property(lambda s: s.nu)
    γa = property(lambda s: s.gamma_a)
    γs = property(lambda s: s.gamma_s)
    μ = property(lambda s: s.mu)

    @classmethod
    def synthetic(cls, M: int, L: int, seed: int, numin=0.0, numax=2525.0, iso: int = 1):
        """Seeded synthetic table (SURVEY.md 8d "Lines, C3/C4"): nu ~ U sorted, log10 S ~ U(-28,-19),
        gamma_a ~ U(.05,.10), gamma_s ~ U(.06,.13), E'' ~ U(0,3000), na ~ U(.6,.8), isotopologue `iso`."""
        rng = np.random.Generator(np.random.PCG64(seed))
        par = dict(M=np.full(L, M, np.int16), I=np.full(L, iso, np.int16), nu=np.sort(rng.uniform(numin, numax, L)),
                   S=10.0 ** rng.uniform(-28, -19, L), gamma_a=rng.uniform(0.05, 0.10, L),
                   gamma_s=rng.uniform(0.06, 0.13, L), Epp=rng.uniform(0, 3000, L), na=rng.uniform(0.6, 0.8, L))
        return cls(par)

    def __repr__(self):
        return f"SpectralLines({self.formula}, {self.N} lines, {self.nu[0]:.3f}-{self.nu[-1]:.3f} cm^-1)"
