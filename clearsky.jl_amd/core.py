"""Host-side mirror of ClearSky.jl's operator surface for the Discretized line-by-line hot path.

Same names, argument meaning and error behaviour as the reference (Julia `!` becomes a trailing underscore), so the
parity tests read like the reference's own calls.  Everything numerical runs in the HIP library through the C ABI
(include/clearsky_hip.h); this module only pre-evaluates the closures that cannot cross the ABI, exactly where the
reference evaluates them, and moves arrays.  Citations are reference paths (src/...).

Matrices the reference stores as [level, nu] column-major (core/shared.jl:93-101) are numpy arrays of shape
(np, nnu) in Fortran order here -- identical memory layout, identical indexing M[i, j].
"""
import ctypes as C
import math
from typing import Callable, Optional, Sequence

import numpy as np

from . import constants as K
from ._lib import (CHEB_LD, CS_MAX_ACCEL, CS_MAX_CIA, CS_MAX_GAS, CS_MAX_TABLE, SHAPES, ClearSkyHIPError, as_f64, check, dptr, lib)
from .hitran import TMAX, TMIN, SpectralLines
from .cia import CIATables, cia, readcia

# ----------------------------------------------------------------------------------------------------------------
# small numerical helpers (host)


def chebygrid(*args):
    """BasicInterpolators.chebygrid: Chebyshev extrema, ascending; chebygrid(n) on [-1,1], chebygrid(a,b,n) mapped."""
    if len(args) == 1:
        n = args[0]
        return np.cos(np.pi * np.arange(n - 1, -1, -1) / (n - 1))
    a, b, n = args
    return (chebygrid(n) + 1) * (b - a) / 2 + a


def pressuregrid(Pt, Ps, n):
    """util.jl:19-23"""
    assert Ps > Pt
    assert n >= 3
    return np.exp(chebygrid(math.log(Pt), math.log(Ps), n))


def trapz(x, y):
    """util.jl:26-33 (sequential sum)."""
    x = np.asarray(x, float)
    y = np.asarray(y, float)
    assert len(x) == len(y), "vectors must be equal length"
    s = 0.0
    for i in range(len(x) - 1):
        s += (x[i + 1] - x[i]) * (y[i] + y[i + 1]) / 2
    return s


def trapz_weights(x):
    """Per-point weights w with sum(w*y) == trapz(x, y) (util.jl:26-33); shards of a grid use slices of these."""
    x = np.asarray(x, float)
    w = np.zeros(len(x))
    if len(x) > 1:
        d = np.diff(x)
        w[:-1] += d / 2
        w[1:] += d / 2
    return w


def logrange(a, b, N=101, gamma=1):
    """util.jl:43-45"""
    return ((10.0 ** np.linspace(0, gamma, N)) - 1) * (b - a) / (10.0 ** gamma - 1) + a


def planck(nu, T):
    """radiation.jl:48-54, W/m^2/cm^-1/sr"""
    num = 100.0 * np.asarray(nu, float)
    x = K.h * K.c * num / (K.k * T)
    p = 2 * K.h * K.c ** 2 * num ** 3
    return 100.0 * p / (np.exp(x) - 1.0)


def stefanboltzmann(T):
    """radiation.jl:95"""
    return K.sigma_sb * T ** 4


def dtaudP(sigma, g, mu):
    """radiation.jl:141"""
    return 1e-4 * sigma * K.Na / (mu * g)


def streamnodes(n: int):
    """core/shared.jl:4-21 -> (m = 1/cos(theta_k), W_k)"""
    m = np.zeros(n)
    W = np.zeros(n)
    check(lib().cs_streamnodes(n, dptr(m), dptr(W)))
    return m, W


def lobattonodes(n: int):
    """core/discretized.jl:2-9 -> nodes and weights on [0,1]"""
    x = np.zeros(n)
    w = np.zeros(n)
    check(lib().cs_lobattonodes(n, dptr(x), dptr(w)))
    return x, w


def psatH2O(T):
    """atmospherics.jl:528-541 (Murphy & Koop 2005)"""
    a = math.log(T)
    b = 1 / T
    if T >= 273.15:
        c = 53.878 - 1331.22 * b - 9.44523 * a + 0.014025 * T
        d = c * math.tanh(0.0415 * (T - 218.8))
        return math.exp(54.842763 - 6763.22 * b - 4.21 * a + 3.67e-4 * T + d)
    return math.exp(9.550426 - 5723.265 * b + 3.53068 * a - 0.00728332 * T)


def ozonelayer(P, Cmax=8e-6):
    """atmospherics.jl:567-578"""
    lp = math.log(P)
    P1, P2, P3 = 10.146433731146518, 7.3777589082278725, 4.605170185988092
    if P2 <= lp <= P1:
        return Cmax * (P1 - lp) / (P1 - P2)
    if P3 <= lp <= P2:
        return Cmax * (lp - P3) / (P2 - P3)
    return 0.0


class AtmosphericProfile:
    """Linear interpolation in ln P without boundary checks (atmospherics.jl:6-26; LinearInterpolator+NoBoundaries)."""

    def __init__(self, P, y):
        P = np.asarray(P, float)
        y = np.asarray(y, float)
        assert len(P) == len(y), "cannot form AtmosphericProfile with unequal numbers of points"
        idx = np.argsort(P, kind="stable")
        self.x = np.log(P[idx])
        self.y = y[idx]

    def __call__(self, P):
        x = math.log(P)
        n = len(self.x)
        i = int(np.searchsorted(self.x, x, side="right")) - 1
        i = min(max(i, 0), n - 2)
        return (x - self.x[i]) * (self.y[i + 1] - self.y[i]) / (self.x[i + 1] - self.x[i]) + self.y[i]


def formprofile(P, x):
    """fluxes.jl:13-16"""
    if callable(x):
        return x
    if np.ndim(x) == 0:
        v = float(x)
        return lambda *a: v
    return AtmosphericProfile(P, x)


def lobattoevaluations(P, fT, fmu, nlobatto):
    """core/discretized.jl:11-30 -> T, mu of shape (nlobatto, np-1), Fortran order"""
    npl = len(P)
    T = np.zeros((nlobatto, npl - 1), order="F")
    mu = np.zeros((nlobatto, npl - 1), order="F")
    xs, _ = lobattonodes(nlobatto)
    for i in range(nlobatto):
        for j in range(npl - 1):
            dP = P[j + 1] - P[j]
            Pi = P[j] + dP * xs[i]
            Ti = fT(Pi)
            T[i, j] = Ti
            mu[i, j] = fmu(Ti, Pi)
    return T, mu


def nodepressures(P, nlobatto):
    """Pressure of node k = i*(nlobatto-1)+n as dDepth! uses it (core/discretized.jl:150,162,169)."""
    P = np.asarray(P, float)
    nl = len(P) - 1
    xs, _ = lobattonodes(nlobatto)
    Pk = np.zeros(nl * (nlobatto - 1) + 1)
    Pk[0] = P[0]
    for i in range(nl):
        dP = P[i + 1] - P[i]
        for n in range(1, nlobatto):
            Pk[i * (nlobatto - 1) + n] = P[i + 1] if n == nlobatto - 1 else P[i] + dP * xs[n]
    return Pk


def nodevalues(X, nlobatto):
    """Flatten a (nlobatto, nl) Lobatto array to node order (node 0 = X[0,0]; node of (n>=1, i) = X[n,i])."""
    nl = X.shape[1]
    out = np.zeros(nl * (nlobatto - 1) + 1)
    out[0] = X[0, 0]
    for i in range(nl):
        for n in range(1, nlobatto):
            out[i * (nlobatto - 1) + n] = X[n, i]
    return out


# ----------------------------------------------------------------------------------------------------------------
# device context


def interp_plan(nu, cut=25.0):
    """Interval sizes the far-wing interpolation uses on grid `nu` with cut-off `cut` ([] = every pair evaluated directly)."""
    nu = np.ascontiguousarray(nu, dtype=np.float64)
    out = (C.c_int * 5)()
    n = lib().cs_interp_plan(nu.size, nu.ctypes.data_as(C.POINTER(C.c_double)), float(cut), out)
    if n < 0:
        check(n)
    return [out[i] for i in range(n)]


def phco2_plan(nu, cut=500.0):
    """The levels PHCO2's far wings are interpolated on (default settings): [(interval size, nodes per interval, chi-regions)], regions
    as a tuple out of (1, 2, 3) = 3-30, 30-120, 120-cut-off cm^-1.  [] = every pair evaluated per point.  Host only."""
    nu = np.ascontiguousarray(nu, dtype=np.float64)
    sz, nd, rg = (C.c_int * 16)(), (C.c_int * 16)(), (C.c_int * 16)()
    n = lib().cs_phco2_plan(nu.size, nu.ctypes.data_as(C.POINTER(C.c_double)), float(cut), 16, sz, nd, rg)
    if n < 0:
        check(n)
    return [(sz[i], nd[i], tuple(r + 1 for r in range(3) if (rg[i] >> r) & 1)) for i in range(n)]


class Context:
    """One HIP context (cs_ctx) = one device + stream + resident gas tables.  Not re-entrant."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        check(lib().cs_create(int(device), C.byref(self._h)))
        self.device = device
        self._slots = {}   # id(sl) -> (slot, sl)
        self._next = 0
        self._tables = {}  # id(Gas) -> slot
        self._free_tables = list(range(CS_MAX_TABLE))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().cs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def set_precision(self, mode="fp64", far_s: float = 1e6):
        """"fp64" (default) or "mixed": Voigt far wings with x^2 >= far_s in fp32 (BASELINE configs[4]); far_s >= 1e6 is the
        knob of the tolerance sweep."""
        check(lib().cs_set_precision(self._h, {"fp64": 0, "mixed": 1}[mode], float(far_s)))

    def set_interp(self, on: bool = True):
        """Far wings by Chebyshev interpolation over 256-point intervals (default on); off = every (nu, line) pair."""
        check(lib().cs_set_interp(self._h, int(bool(on))))

    def set_interp_plan(self, first_level: int = -1, size_min: int = 128, size_max: int = 2048):
        """Tuning: first interval level every gas uses (-1 = by line density) and the range of interval sizes considered."""
        check(lib().cs_set_interp_plan(self._h, int(first_level), int(size_min), int(size_max)))

    def set_matrix_cores(self, on=True):
        """Separable far-wing node sums on the matrix cores: True/1 (default) where the grid is long enough to pay, 2 always,
        False/0 every node sum on the vector unit."""
        check(lib().cs_set_matrix_cores(self._h, int(on)))

    def set_merge(self, on: bool = True):
        """One launch set per column (default): Voigt/Lorentz gases with the same cut-off share a merged line table; False = one
        launch set per gas."""
        check(lib().cs_set_merge(self._h, int(bool(on))))

    def set_tuning(self, key: int, value: int):
        """A/B switches (cs_set_tuning, a lab entry point: include/clearsky_hip_dev.h describes each key): where the interpolated
        wings are applied, matrix-core kernels on short grids, side streams, interpolation margin, hipGraph replay, flux sweeps per
        stream, series radius rank, PHCO2 core and node counts, low-order far pieces, level cascade, fused flux tail, ..."""
        check(lib().cs_set_tuning(self._h, int(key), int(value)))

    def slot_of(self, sl: SpectralLines, keep=()) -> int:
        """Upload `sl` (once) and return its gas slot.  `keep`: tables that must stay where they are (the other gases of the call this
        one belongs to): with every slot taken, the oldest table NOT among them makes room."""
        key = id(sl)
        if key in self._slots:
            return self._slots[key][0]
        slot = self._take_slot(keep)
        iso = np.ascontiguousarray(sl.I, dtype=np.int16)
        ncheb = np.ascontiguousarray(sl.ncheb, dtype=np.int32)
        cheb = as_f64(sl.cheb)
        assert cheb.shape[1] == CHEB_LD
        arrs = [as_f64(a) for a in (sl.nu, sl.S, sl.gamma_a, sl.gamma_s, sl.Epp, sl.na, sl.mu)]
        check(lib().cs_gas_upload(self._h, slot, len(arrs[0]), *[dptr(a) for a in arrs],
                                  iso.ctypes.data_as(C.POINTER(C.c_int16)), len(ncheb),
                                  ncheb.ctypes.data_as(C.POINTER(C.c_int32)), dptr(cheb)))
        self._slots[key] = (slot, sl)
        return slot


    def slots_of(self, tables):
        """Slots of all the line tables of ONE call: none of them is evicted to make room for another of them."""
        return [self.slot_of(sl, keep=tables) for sl in tables]

    def _take_slot(self, keep=()):
        if len(self._slots) >= CS_MAX_GAS:   # evict the oldest table that the current call does not use
            pinned = {id(x) for x in keep}
            for old in self._slots:
                if old not in pinned:
                    return self._slots.pop(old)[0]
            raise ClearSkyHIPError(-1, f"all {CS_MAX_GAS} gas slots hold tables of the current call")
        slot = self._next
        self._next += 1
        return slot

    def load_par(self, filename: str, M: int, numin: float = 0.0, numax: float = np.inf, Scut: float = 0.0, I=(),
                 maxlines: int = -1) -> SpectralLines:
        """A HITRAN .par file straight into a gas slot of this context (cs_gas_upload_par: parse, filter, strongest-N, sort,
        MOLPARAM lookup and upload on the native side -- hitran/par.jl:91-286 without host arrays in between).  Returns the
        SpectralLines mirror of what was loaded (arrays copied back from the slot), already bound to the slot, so gases built
        on it do not upload again.  `M`: the molecule the file holds (its MOLPARAM rows go along)."""
        from .hitran import MOLPARAM, ISOINDEX
        if not filename.endswith(".par"):
            raise AssertionError("expected file with .par extension, downloaded from https://hitran.org/lbl/")
        mp = MOLPARAM[M]
        keep = np.array([ISOINDEX[c] if isinstance(c, str) else int(c) for c in I], dtype=np.int32)
        mu = as_f64(mp.mu)
        ncheb = np.ascontiguousarray(mp.ncheb_table(), dtype=np.int32)
        cheb = as_f64(mp.cheb_table())
        slot = self._take_slot()
        L = C.c_int64()
        check(lib().cs_gas_upload_par(self._h, slot, filename.encode(), float(numin), float(min(numax, 1e300)), float(Scut),
                                      keep.ctypes.data_as(C.POINTER(C.c_int)), len(keep), int(maxlines), int(M), dptr(mu), len(mu),
                                      ncheb.ctypes.data_as(C.POINTER(C.c_int32)), dptr(cheb), C.byref(L)))
        n = L.value
        a = {k: np.zeros(n) for k in ("nu", "S", "gamma_a", "gamma_s", "Epp", "na")}
        iso = np.zeros(n, dtype=np.int16)
        check(lib().cs_gas_fetch(self._h, slot, n, dptr(a["nu"]), dptr(a["S"]), dptr(a["gamma_a"]), dptr(a["gamma_s"]), dptr(a["Epp"]),
                                 dptr(a["na"]), None, iso.ctypes.data_as(C.POINTER(C.c_int16))))
        sl = SpectralLines(dict(M=np.full(n, M, np.int16), I=iso, **a))
        self._slots[id(sl)] = (slot, sl)
        return sl

    def cia_slot(self, x: CIATables) -> int:
        """Upload a CIATables object (once) and return its slot."""
        if not hasattr(self, "_cia"):
            self._cia = {}
        key = id(x)
        if key in self._cia:
            return self._cia[key][0]
        if len(self._cia) >= CS_MAX_CIA:
            raise ClearSkyHIPError(-1, "no free CIA slot")
        slot = len(self._cia)
        nb = len(x.grids) + len(x.single)
        check(lib().cs_cia_begin(self._h, slot, nb))
        b = 0
        for g_nu, g_T, lnk in x.grids:
            check(lib().cs_cia_band(self._h, slot, b, len(g_nu), dptr(as_f64(g_nu)), len(g_T), dptr(as_f64(g_T)), dptr(as_f64(lnk))))
            b += 1
        for s_nu, s_lnk, s_T in x.single:
            check(lib().cs_cia_band(self._h, slot, b, len(s_nu), dptr(as_f64(s_nu)), 1, dptr(as_f64([s_T])), dptr(as_f64(s_lnk))))
            b += 1
        self._cia[key] = (slot, x)
        return slot

    def table_slot(self, owner) -> int:
        """Reserve an opacity-table slot for a baked Gas (released when the Gas is garbage collected)."""
        if not self._free_tables:
            raise ClearSkyHIPError(-1, "no free opacity-table slot (CS_MAX_TABLE baked gases per context)")
        slot = self._free_tables.pop(0)
        self._tables[id(owner)] = slot
        return slot

    def release_table(self, owner):
        slot = self._tables.pop(id(owner), None)
        if slot is not None and getattr(self, "_h", None) and self._h.value:
            lib().cs_table_clear(self._h, slot)
            self._free_tables.append(slot)


class MultiContext:
    """N contexts on N devices of one node (several may share a device: rehearsal), holding the same gas tables in the same slots
    -- the `ctxs` of cs_fluxes_discretized_multi, i.e. radiate!/fluxes with `ngpu` GPUs behind them: contiguous
    cost-balanced wavenumber ranges, one per context, band fluxes added on the host in context order.  Line-by-line, gray and
    function absorbers only (baked gases, CIA pairs and accelerated absorbers live on one context)."""

    def __init__(self, devices):
        self.ctxs = [Context(int(d)) for d in devices]
        assert len(self.ctxs) >= 1

    def slot_of(self, sl: SpectralLines, keep=()) -> int:
        slots = [c.slot_of(sl, keep) for c in self.ctxs]
        assert all(s_ == slots[0] for s_ in slots), "contexts of a MultiContext must be used together from the start"
        return slots[0]

    def slots_of(self, tables):
        return [self.slot_of(sl, keep=tables) for sl in tables]

    def handles(self):
        return (C.c_void_p * len(self.ctxs))(*[c.handle.value for c in self.ctxs])

    def set_merge(self, on=True):
        for c in self.ctxs:
            c.set_merge(on)

    def close(self):
        for c in self.ctxs:
            c.close()


def balanced_ranges(nu, line_positions, nparts: int):
    """cs_balanced_ranges: `nparts` contiguous ranges (start, stop) of the grid `nu` with equal estimated device time -- the multi-GPU
    partition of SURVEY.md 8e.  line_positions: one sorted array of line wavenumbers per gas.  Host only."""
    nu = as_f64(nu)
    tabs = [as_f64(a) for a in line_positions]
    n = (C.c_int64 * max(len(tabs), 1))(*[len(a) for a in tabs])
    ptrs = (C.POINTER(C.c_double) * max(len(tabs), 1))(*[dptr(a) for a in tabs])
    out = (C.c_int64 * (2 * nparts))()
    check(lib().cs_balanced_ranges(len(nu), dptr(nu), len(tabs), n, ptrs, int(nparts), out))
    return [(int(out[2 * r]), int(out[2 * r + 1])) for r in range(nparts)]


def rebalance_ranges(nu, line_positions, prev_ranges, prev_time, fixed_time: float = 0.0):
    """cs_rebalance_ranges: the partition `prev_ranges` re-cut from the times its ranges were measured to take (balance the
    work, not the wavenumbers).  Host only."""
    nu = as_f64(nu)
    tabs = [as_f64(a) for a in line_positions]
    nparts = len(prev_ranges)
    n = (C.c_int64 * max(len(tabs), 1))(*[len(a) for a in tabs])
    ptrs = (C.POINTER(C.c_double) * max(len(tabs), 1))(*[dptr(a) for a in tabs])
    prev = (C.c_int64 * (2 * nparts))(*[int(x) for r in prev_ranges for x in r])
    t = as_f64(prev_time)
    out = (C.c_int64 * (2 * nparts))()
    check(lib().cs_rebalance_ranges(len(nu), dptr(nu), len(tabs), n, ptrs, nparts, prev, dptr(t), float(fixed_time), out))
    return [(int(out[2 * r]), int(out[2 * r + 1])) for r in range(nparts)]


_default_ctx = {}


def default_context(device: int = 0) -> Context:
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


# ----------------------------------------------------------------------------------------------------------------
# line shapes (B1)


def shape_batch(sl: SpectralLines, shape, nu, T, P, Pp, dnu_cut=25.0, ctx: Optional[Context] = None):
    """sigma[k, :] = shape!(.., nu, sl, T[k], P[k], Pp[k], dnu_cut) for all states in one launch (bake's inner loop,
    gases.jl:115-126).  Returns an array of shape (K, nnu)."""
    ctx = ctx or default_context()
    nu = as_f64(nu)
    T, P, Pp = (as_f64(np.atleast_1d(a)) for a in (T, P, Pp))
    Kn = len(T)
    assert len(P) == Kn and len(Pp) == Kn
    out = np.zeros((Kn, len(nu)))
    sh = SHAPES[shape] if isinstance(shape, str) else int(shape)
    check(lib().cs_shape_batch(ctx.handle, ctx.slot_of(sl), sh, float(dnu_cut), len(nu), dptr(nu), Kn, dptr(T), dptr(P),
                               dptr(Pp), dptr(out), len(nu)))
    return out


def shape_points(sl: SpectralLines, shape, nu, T, P, Pp, dnu_cut=25.0, ctx: Optional[Context] = None):
    """The scalar-wavenumber methods shape(nu, sl, T, P, Pp, dnu_cut) (line_shapes.jl:399-405 etc.) mapped over `nu` and over
    the states: every line with |nu - nul| <= dnu_cut counts (includedlines(::Real), :12-16).  Returns (K, nnu)."""
    ctx = ctx or default_context()
    nu = as_f64(np.atleast_1d(nu))
    T, P, Pp = (as_f64(np.atleast_1d(a)) for a in (T, P, Pp))
    out = np.zeros((len(T), len(nu)))
    sh = SHAPES[shape] if isinstance(shape, str) else int(shape)
    check(lib().cs_shape_points(ctx.handle, ctx.slot_of(sl), sh, float(dnu_cut), len(nu), dptr(nu), len(T), dptr(T), dptr(P), dptr(Pp),
                                dptr(out), len(nu)))
    return out


def _shape_inplace(name, default_cut):
    def f_(sigma, nu, sl, T, P, Pp, dnu_cut=default_cut, ctx=None):
        r = shape_batch(sl, name, nu, [T], [P], [Pp], dnu_cut, ctx)
        sigma[...] = r[0]   # overwrite, line_shapes.jl:85
        return None

    def f(nu, sl, T, P, Pp, dnu_cut=default_cut, ctx=None):
        if np.ndim(nu) == 0:   # the scalar-nu method: inclusive cut-off, no end-point pre-filter (line_shapes.jl:12-16)
            return float(shape_points(sl, name, [float(nu)], [T], [P], [Pp], dnu_cut, ctx)[0, 0])
        return shape_batch(sl, name, nu, [T], [P], [Pp], dnu_cut, ctx)[0]

    f_.__doc__ = f"{name}!(sigma, nu, sl, T, P, Pp, dnu_cut={default_cut}) -- absorption/line_shapes.jl; fills sigma in place"
    f.__doc__ = f"{name}(nu, sl, T, P, Pp, dnu_cut={default_cut}) -- absorption/line_shapes.jl; returns cross-sections"
    return f_, f


voigt_, voigt = _shape_inplace("voigt", 25.0)        # line_shapes.jl:412-448
lorentz_, lorentz = _shape_inplace("lorentz", 25.0)  # :313-348
doppler_, doppler = _shape_inplace("doppler", 25.0)  # :200-235
PHCO2_, PHCO2 = _shape_inplace("PHCO2", 500.0)       # :527-564


def faddeeva(x, y, ctx: Optional[Context] = None):
    """Re w(x+iy) evaluated by the kernels' device function (stand-in for Faddeyeva985.faddeyeva, line_shapes.jl:375)."""
    ctx = ctx or default_context()
    x = as_f64(np.atleast_1d(x))
    y = as_f64(np.atleast_1d(y))
    out = np.zeros_like(x)
    check(lib().cs_faddeeva_batch(ctx.handle, len(x), dptr(x), dptr(y), dptr(out)))
    return out


def device_function(which: str, x, y=None, z=None, ctx: Optional[Context] = None):
    """The flux kernel's device functions point by point (test hook, cs_devfn_batch): "exp" (its own exp), "planck" (x = nu,
    y = T; radiation.jl:48-54), "layerplanck" (x = B1, y = B2, z = tau; discretized.jl:85-87)."""
    ctx = ctx or default_context()
    x = as_f64(np.atleast_1d(x))
    y = None if y is None else as_f64(np.broadcast_to(y, x.shape))
    z = None if z is None else as_f64(np.broadcast_to(z, x.shape))
    out = np.zeros_like(x)
    check(lib().cs_devfn_batch(ctx.handle, {"exp": 0, "planck": 1, "layerplanck": 2}[which], len(x), dptr(x), dptr(y), dptr(z), dptr(out)))
    return out


# ----------------------------------------------------------------------------------------------------------------
# absorbers (B2)


class AbstractGas:
    pass


class DirectGas(AbstractGas):
    """Line-by-line gas evaluated directly at every (T,P) node ("Mode D").

    Reference semantics: the function absorber  (nu,T,P) -> C*shape(nu, sl, T, P, C*P)  with C = fC(T,P)
    (absorbers.jl:16,24,71,91 + line_shapes.jl:399-405); differs from a baked `Gas` (gases.jl:205-249) only by the
    opacity-table interpolation error (gases.jl:7).  `fC` is a callable fC(T,P) or a number (molar concentration).
    """

    def __init__(self, sl: SpectralLines, fC, nu, shape="voigt", dnu_cut=None):
        nu = np.array(nu, dtype=float)
        assert len(nu) > 0
        assert np.all(np.diff(nu) > 0), "wavenumbers must be unique and in ascending order"
        assert np.all(nu >= 0), "wavenumbers must be positive"
        self.sl = sl
        self.name, self.formula = sl.name, sl.formula
        self.mu = float(np.sum(sl.A * sl.mu) / np.sum(sl.A))   # gases.jl:233
        self.nu = nu
        self.shape = shape
        self.dnu_cut = float(dnu_cut if dnu_cut is not None else (500.0 if SHAPES[shape] == 3 else 25.0))
        if callable(fC):
            self.fC = fC
        else:
            cval = float(fC)
            assert 0 <= cval <= 1.0, f"gas molar concentrations must be in [0,1], not {cval}"
            self.fC = lambda T, P: cval

    def concentration(self, T, P):
        return self.fC(T, P)

    def __call__(self, *a, ctx: Optional["Context"] = None):
        """g(i, T, P) or g(T, P): the function absorber C*shape(nu, sl, T, P, C*P) at wavenumber index i (0-based) or at every
        wavenumber (the scalar access sigma-chain uses, absorbers.jl:84-92)"""
        if len(a) == 3:
            i, T, P = a
            Cv = self.fC(T, P)
            return Cv * float(shape_points(self.sl, self.shape, [self.nu[int(i)]], [T], [P], [Cv * P], self.dnu_cut, ctx)[0, 0])
        T, P = a
        Cv = self.fC(T, P)
        return Cv * shape_points(self.sl, self.shape, self.nu, [T], [P], [Cv * P], self.dnu_cut, ctx)[0]


class GrayGas(AbstractGas):
    """gases.jl:342-360: constant cross-section [cm^2/molecule] for any arguments."""

    def __init__(self, sigma, nu):
        self.name = self.formula = "Gray"
        self.mu = float("nan")
        self.nu = np.array(nu, dtype=float)
        self.sigma = float(sigma)

    def __call__(self, *a):
        return self.sigma


class SemiGrayGas(AbstractGas):
    """gases.jl:366-386: sigma [cm^2/molecule] at the wavenumbers up to nu_cut, zero beyond -- g(i, ...) = (nu[i] <= nu_cut) ? sigma : 0.
    In a column it travels as a per-wavenumber vector added at every node state (the C ABI's sigma_extra)."""

    def __init__(self, sigma, nu, nucut):
        self.name = self.formula = "SemiGray"
        self.mu = float("nan")
        self.nu = np.array(nu, dtype=float)
        self.nucut = float(nucut)
        self.sigma = float(sigma)

    def __call__(self, *a):
        if len(a) == 2:        # U(T, P): every wavenumber
            return self.vector(self.nu)
        return self.sigma if self.nu[int(a[0])] <= self.nucut else 0.0

    def vector(self, nu):
        return np.where(np.asarray(nu, float) <= self.nucut, self.sigma, 0.0)


class AtmosphericDomain:
    """gases.jl:26-61: Chebyshev-extrema grids in T and ln P over which cross-sections are baked."""

    def __init__(self, Trange, nT: int, Prange, nP: int):
        assert all(t > 0 for t in Trange), "temperature range must be positive"
        assert all(p > 0 for p in Prange), "pressure range must be positive"
        assert all(t >= TMIN for t in Trange), f"minimum temperature with Qref/Q accuracy is {TMIN} K"
        assert all(t <= TMAX for t in Trange), f"maximum temperature with Qref/Q accuracy is {TMAX} K"
        assert Trange[0] < Trange[1], f"Trange[1] ({Trange[0]}) can't be greater than Trange[2] ({Trange[1]})"
        assert Prange[0] < Prange[1], f"Prange[1] ({Prange[0]}) can't be greater than Prange[2] ({Prange[1]})"
        self.T = chebygrid(float(Trange[0]), float(Trange[1]), nT)
        self.Tmin, self.Tmax, self.nT = float(Trange[0]), float(Trange[1]), int(nT)
        self.P = np.exp(chebygrid(math.log(Prange[0]), math.log(Prange[1]), nP))
        self.Pmin, self.Pmax, self.nP = float(Prange[0]), float(Prange[1]), int(nP)

    def __repr__(self):
        return (f"AtmosphericDomain:\n  {self.nT} temperature nodes ∈ [{self.Tmin},{self.Tmax}] K\n"
                f"  {self.nP} pressure nodes ∈ [{self.Pmin},{self.Pmax}] Pa")


class Gas(AbstractGas):
    """Gas(sl, fC, nu, Omega, shape='voigt', dnu_cut=25) -- the reference's baked gas object (gases.jl:205-249): cross-sections
    are evaluated once on Omega's (T, ln P) Chebyshev grid (`bake`, gases.jl:97-145 -- all nT*nP states in one device launch)
    and interpolated afterwards (`OpacityTable`, gases.jl:68-85).  The ln(sigma) tables stay resident in HBM."""

    def __init__(self, sl, fC, nu, Omega: AtmosphericDomain, shape="voigt", dnu_cut=25.0, ctx: Optional["Context"] = None,
                 keep_host_tables: bool = False, **readpar_kwargs):
        if isinstance(sl, str):
            sl = SpectralLines(sl, **readpar_kwargs)
        nu = np.array(nu, dtype=float)
        assert len(nu) > 0
        assert np.all(np.diff(nu) > 0), "wavenumbers must be unique and in ascending order"
        assert np.all(nu >= 0), "wavenumbers must be positive"
        self.ctx = ctx or default_context()
        self.sl, self.name, self.formula = sl, sl.name, sl.formula
        self.mu = float(np.sum(sl.A * sl.mu) / np.sum(sl.A))      # gases.jl:233
        self.nu, self.Omega, self.shape, self.dnu_cut = nu, Omega, shape, float(dnu_cut)
        self.fC = fC if callable(fC) else (lambda T, P, _c=float(fC): _c)
        conc = np.zeros((Omega.nT, Omega.nP), order="F")
        for i, T in enumerate(Omega.T):
            for j, P in enumerate(Omega.P):
                Cv = self.fC(T, P)
                assert 0 <= Cv <= 1, f"gas molar concentrations must be in [0,1], not {Cv} (encountered @ {T} K, {P} Pa)"
                conc[i, j] = Cv
        self.slot = self.ctx.table_slot(self)
        out = np.zeros((len(nu), Omega.nT, Omega.nP), order="F") if keep_host_tables else None
        try:
            check(lib().cs_bake(self.ctx.handle, self.ctx.slot_of(sl), self.slot, SHAPES[shape], self.dnu_cut, len(nu), dptr(nu),
                                Omega.nT, dptr(as_f64(Omega.T)), Omega.nP, dptr(as_f64(Omega.P)), dptr(conc.ravel(order="F").copy()),
                                out.ctypes.data_as(C.POINTER(C.c_double)) if out is not None else None))
        except Exception:
            self.ctx.release_table(self)
            raise
        self.lnsigma = out     # [nnu, nT, nP] like the reference's sigma block (only when keep_host_tables)

    def __del__(self):
        try:
            self.ctx.release_table(self)
        except Exception:
            pass

    def concentration(self, T, P):
        """gases.jl:270"""
        return self.fC(T, P)

    def rawsigma(self, T, P, i=None):
        """rawσ(g, T, P) / rawσ(g, i, T, P): interpolated cross-section(s) WITHOUT the concentration factor (gases.jl:256-263)"""
        n = len(self.nu)
        i0, cnt = (0, n) if i is None else (int(i), 1)
        out = np.zeros(cnt)
        check(lib().cs_table_eval(self.ctx.handle, self.slot, float(T), float(P), i0, cnt, dptr(out)))
        return out if i is None else float(out[0])

    def __call__(self, *a):
        """g(i, T, P) or g(T, P): concentration-scaled cross-section(s) (gases.jl:278-281)"""
        if len(a) == 3:
            i, T, P = a
            return self.concentration(T, P) * self.rawsigma(T, P, i)
        T, P = a
        return self.concentration(T, P) * self.rawsigma(T, P)

    def reconcentrate(self, fC):
        """gases.jl:292-320: same tables, new concentration function (self-broadening is NOT recomputed, as in the reference)"""
        f = fC if callable(fC) else (lambda T, P, _c=float(fC): _c)
        for P in self.Omega.P:
            for T in self.Omega.T:
                Cv = f(T, P)
                assert 0 <= Cv <= 1.0, f"gas molar concentrations must be in [0,1], not {Cv}, which was encountered at T={T} P={P}"
        g = object.__new__(Gas)
        g.__dict__.update(self.__dict__)
        g.fC = f
        g._shared_with = self      # keeps the table owner alive; the copy does not own the slot
        g.__class__ = _GasView
        return g


class _GasView(Gas):
    def __del__(self):
        pass


def reconcentrate(g: Gas, fC):
    return g.reconcentrate(fC)


def opacityerror(g: Gas, i: int, N: int = 50, shape=None):
    """opacityerror(Π, Ω, sl, ν, C, shape=voigt, N=50) (gases.jl:152-175) for the table of wavenumber index i (0-based) of a baked Gas
    -- Π = that wavenumber's OpacityTable, Ω = g.Omega, sl = g.sl, ν = g.nu[i], C = the concentration the gas was baked with.
    Returns (T, P, aerr, rerr) on the reference's N x N grid (T linear over [Tmin, Tmax], P logarithmic over [Pmin, Pmax]):
    interpolated minus exact cross-section and that over the exact one.  The N*N exact values come from ONE cs_shape_points call
    (the scalar-wavenumber `shape(ν, sl, T, P, C(T,P)*P)` at N*N states), the interpolated ones from cs_table_eval."""
    Om = g.Omega
    T = np.linspace(Om.Tmin, Om.Tmax, N)
    P = 10.0 ** np.linspace(math.log10(Om.Pmin), math.log10(Om.Pmax), N)
    TT, PP = np.meshgrid(T, P, indexing="ij")
    Pp = np.array([g.fC(t, p) * p for t, p in zip(TT.ravel(), PP.ravel())])
    sex = shape_points(g.sl, shape or g.shape, [g.nu[int(i)]], TT.ravel(), PP.ravel(), Pp, g.dnu_cut, g.ctx)[:, 0].reshape(N, N)
    sop = np.array([[g.rawsigma(t, p, int(i)) for p in P] for t in T])
    aerr = sop - sex
    with np.errstate(divide="ignore", invalid="ignore"):
        rerr = aerr / sex
    return T, P, aerr, rerr


class UnifiedAbsorber:
    """absorbers.jl:18-77: gases + functions sigma(nu,T,P) on one wavenumber grid."""

    def __init__(self, *absorbers):
        if len(absorbers) == 1 and isinstance(absorbers[0], (tuple, list)):
            absorbers = tuple(absorbers[0])
        assert len(absorbers) > 0, "no absorbers... nothing to group"
        assert len(absorbers) == len(set(map(id, absorbers))), "duplicate absorbers"
        for a in absorbers:
            if isinstance(a, (UnifiedAbsorber, AcceleratedAbsorber)) or not (isinstance(a, (AbstractGas, CIATables)) or callable(a)):
                raise TypeError("absorbers must only be gases (<: Gas), CIA objects, or functions in the form σ(ν, T, P)")
        self.gas = tuple(a for a in absorbers if isinstance(a, AbstractGas))
        if not self.gas:
            raise ValueError("must have at least one Gas object, which specifies wavenumber samples")
        realgas = [g_ for g_ in self.gas if isinstance(g_, (Gas, DirectGas))]     # "real gases, ignoring Gray" absorbers.jl:67
        self.cia = tuple(CIA(x, realgas) for x in absorbers if isinstance(x, CIATables))   # absorbers.jl:69
        self.fun = tuple(a for a in absorbers if not isinstance(a, (AbstractGas, CIATables)))
        nu0 = self.gas[0].nu
        assert all(len(g.nu) == len(nu0) and np.array_equal(g.nu, nu0) for g in self.gas), \
            "gases must have identical wavenumber vectors"
        self.nu = nu0
        self.nnu = len(nu0)

    def __call__(self, *a):
        """U(i, T, P) = Sigma(U, i, T, P) (absorbers.jl:95,97) or U(T, P) (all wavenumbers, :99): the sigma-chain over gases, CIA
        objects and functions (absorbers.jl:84-92).  i is 0-based."""
        if len(a) == 3:
            i, T, P = a
            v = self.nu[int(i)]
            return (sum(g_(int(i), T, P) for g_ in self.gas) + sum(x(v, T, P) for x in self.cia) + sum(f(v, T, P) for f in self.fun))
        T, P = a
        out = np.zeros(self.nnu)
        for g_ in self.gas:
            out = out + np.asarray(g_(T, P), float)
        for x in self.cia:
            out = out + np.array([x(v, T, P) for v in self.nu])
        for f in self.fun:
            try:
                out = out + np.asarray(f(self.nu, T, P), float)
            except Exception:
                out = out + np.array([f(v, T, P) for v in self.nu], float)
        return out

    def update_(self, T):
        """blank, for similarity with AcceleratedAbsorber (absorbers.jl:80)"""
        return None


def pressurelimits(gases):
    """absorbers.jl:248-256: (largest Pmin, smallest Pmax) over the baked Gas members; (0, inf) without any"""
    g = [g_ for g_ in gases if isinstance(g_, Gas)]
    if not g:
        return 0.0, float("inf")
    return max(g_.Omega.Pmin for g_ in g), min(g_.Omega.Pmax for g_ in g)


def temperaturelimits(x):
    """absorbers.jl:258-270: accepts a tuple of gases, a UnifiedAbsorber or an AcceleratedAbsorber"""
    gases = x.U.gas if isinstance(x, AcceleratedAbsorber) else (x.gas if isinstance(x, UnifiedAbsorber) else x)
    g = [g_ for g_ in gases if isinstance(g_, Gas)]
    if not g:
        return 0.0, float("inf")
    return max(g_.Omega.Tmin for g_ in g), min(g_.Omega.Tmax for g_ in g)


def checkpressures(x, Ps, Pt):
    """absorbers.jl:101,209,237-246"""
    gases = x.U.gas if isinstance(x, AcceleratedAbsorber) else (x.gas if isinstance(x, UnifiedAbsorber) else x)
    assert Ps > Pt, "Pₛ must be greater than Pₜ"
    Pmin, Pmax = pressurelimits(gases)
    for P in (Ps, Pt):
        assert P >= Pmin, f"Pressure {P} Pa too low, domain minimum is {Pmin}"
        assert P <= Pmax, f"Pressure {P} Pa too low, domain minimum is {Pmax}"


def Sigma(A, i: int, T, P):
    """Σ(𝒜, i, T, P) (absorbers.jl:95,203): total cross-section at wavenumber index i (0-based); an AcceleratedAbsorber ignores T"""
    if isinstance(A, AcceleratedAbsorber):
        return A(i, P)
    return A(i, T, P)


class AcceleratedAbsorber:
    """AcceleratedAbsorber(T, P, absorbers...) (absorbers.jl:114-166): per-wavenumber ln Sigma on the pressure knots P, interpolated
    linearly in ln P.  The knots' cross-sections Sigma(U, i, T_k, P_k) are evaluated by the line kernels for every wavenumber and
    knot at once and stay in HBM (cs_accel_store); `update_(A, T)` = update!(A, T) (:173-200) re-evaluates them."""

    def __new__(cls, T=None, P=None, *absorbers, ctx: Optional[Context] = None):
        # AcceleratedAbsorber(T, P, A::AcceleratedAbsorber) hands A itself back (absorbers.jl:161-164): the SAME object, so that the
        # device slot has exactly one owner
        if len(absorbers) == 1 and isinstance(absorbers[0], AcceleratedAbsorber):
            A = absorbers[0]
            assert np.all(np.asarray(P, float)[np.argsort(P)] == A.P), \
                "cannot change AcceleratedAbsorber's pressure coordinates after construction"
            return A
        return super().__new__(cls)

    def __init__(self, T, P, *absorbers, ctx: Optional[Context] = None):
        if len(absorbers) == 1 and absorbers[0] is self:   # the alias form: __new__ returned the existing object, nothing to build
            return
        self.ctx = ctx or default_context()
        U = absorbers[0] if (len(absorbers) == 1 and isinstance(absorbers[0], UnifiedAbsorber)) else UnifiedAbsorber(*absorbers)
        P = np.asarray(P, float)
        T = np.asarray(T, float)
        assert len(P) == len(T)
        idx = np.argsort(P, kind="stable")       # absorbers.jl:140-142
        self.P, self.T = P[idx].copy(), T[idx].copy()
        self.U, self.nu, self.nnu = U, U.nu, U.nnu
        used = getattr(self.ctx, "_accel_used", set())
        free = [s_ for s_ in range(CS_MAX_ACCEL) if s_ not in used]
        if not free:
            raise ClearSkyHIPError(-1, "no free accelerated-absorber slot (CS_MAX_ACCEL per context)")
        self.slot = free[0]
        used.add(self.slot)
        self.ctx._accel_used = used
        self._knots = None
        self.update_(self.T)

    def __del__(self):
        try:
            if getattr(self.ctx, "_h", None) and self.ctx._h.value and hasattr(self, "slot"):
                res = getattr(self.ctx, "_resident", None)
                if res is not None and getattr(res, "accel", None) is self:
                    self.ctx._resident = None      # (cs_accel_clear drops a resident column that reads this slot)
                lib().cs_accel_clear(self.ctx._h, self.slot)
                self.ctx._accel_used.discard(self.slot)
        except Exception:
            pass

    def update_(self, T, idx=None):
        """update!(A, T) / update!(A, T, idx) (absorbers.jl:173-200).  idx is 0-based; a single-knot update re-evaluates the
        whole set of knots (one pass of the kernels either way)."""
        if idx is not None:
            Tn = self.T.copy()
            Tn[int(idx)] = float(T)
            T = Tn
        T = np.asarray(T, float)
        assert len(T) == len(self.P)
        Pk, Tk = self.P, T
        lnP = np.log(Pk)
        fT = lambda p: float(Tk[int(np.argmin(np.abs(lnP - math.log(p))))])     # the knot's own temperature at the knot
        if self._knots is None:
            # a column whose node k is knot k: nlobatto = 2, levels = knots (g, mu, fS, fa play no role in the cross-sections)
            self._knots = Column(Pk, 1.0, fT, 1.0, None, None, self.U, core=Discretized(1, 2), want_tau=False, want_M=False,
                                 ctx=self.ctx, _warn=False)
        else:
            self._knots.update(fT)
        self._knots._ensure_resident()
        check(lib().cs_accel_store(self.ctx.handle, self.slot))
        self.T = T.copy()
        return None

    def __call__(self, *a):
        """A(i, P) = exp(phi_i(ln P)) and A(P) for all wavenumbers (absorbers.jl:203-207).  i is 0-based."""
        if len(a) == 2:
            i, P = a
            out = np.zeros(1)
            check(lib().cs_accel_eval(self.ctx.handle, self.slot, float(P), int(i), 1, dptr(out)))
            return float(out[0])
        out = np.zeros(self.nnu)
        check(lib().cs_accel_eval(self.ctx.handle, self.slot, float(a[0]), 0, self.nnu, dptr(out)))
        return out

    def __repr__(self):
        return f"AcceleratedAbsorber ({len(self.P)} pressure samples)"


def update_(A, T, idx=None):
    """update!(A, T[, idx]) for either absorber type (absorbers.jl:80,173)"""
    return A.update_(T) if idx is None or isinstance(A, UnifiedAbsorber) else A.update_(T, idx)


class _NoAbsorbers:
    """the (empty) member lists of a column whose cross-sections come from an AcceleratedAbsorber"""

    def __init__(self, nu):
        self.gas, self.cia, self.fun, self.nu, self.nnu = (), (), (), nu, len(nu)


class CIA:
    """CIA(ciatables, gases): a CIATables object paired with the two gases whose partial pressures it needs
    (collision_induced_absorption.jl:431-465).  chi(nu, T, P) = cia(nu, tables, T, P, P*C1(T,P), P*C2(T,P))."""

    def __init__(self, x: CIATables, gases):
        if len(gases) == 0:
            raise ValueError("no Gas objects provided, cannot create CIA object")

        def find(f):
            m = [g_ for g_ in gases if g_.formula == f]
            assert len(m) > 0, f"pairing failed for {x.name} CIA, gas {f} is missing"
            assert len(m) == 1, f"pairing failed for {x.name} CIA, duplicate {f} gases found"
            return m[0]
        self.name, self.formulae, self.x = x.name, x.formulae, x
        self.g1, self.g2 = find(x.formulae[0]), find(x.formulae[1])

    def __call__(self, nu, T, P):
        return cia(nu, self.x, T, P, P * self.g1.concentration(T, P), P * self.g2.concentration(T, P))


def unifyabsorbers(absorbers):
    """absorbers.jl:214-223"""
    if len(absorbers) == 0:
        raise ValueError("no absorbers")
    if len(absorbers) == 1 and isinstance(absorbers[0], (UnifiedAbsorber, AcceleratedAbsorber)):
        U = absorbers[0]
    else:
        U = UnifiedAbsorber(*absorbers)
    return U, U.nu, U.nnu


# ----------------------------------------------------------------------------------------------------------------
# numerical core selector and output container


class Discretized:
    """core/shared.jl:55-66 -- and julia/ClearSkyHIP.jl's HIPDiscretized, whose extra field it carries: `fluxpack` says what
    radiate_ (radiate!, fluxes.jl:357-383) brings back.  "full": tau, M+, M- and the band fluxes, as the reference fills its FluxPack;
    "bands": Fup, Fdn, Fnet only -- all heating! reads (radiative_convective.jl:109-144) -- tau, M+, M- stay in HBM (NULL across the
    ABI: no 146 MB copy per call at 1e5 x 60) and keep whatever the FluxPack held."""

    def __init__(self, nstream: int = 5, nlobatto: int = 2, fluxpack: str = "full"):
        if fluxpack not in ("full", "bands"):
            raise ValueError(f"fluxpack must be 'full' or 'bands', not {fluxpack!r}")
        self.nstream, self.nlobatto, self.fluxpack = int(nstream), int(nlobatto), fluxpack

    def __repr__(self):
        return f"Discretized(nstream={self.nstream}, nlobatto={self.nlobatto}" + (", fluxpack='bands')" if self.fluxpack == "bands" else ")")


HIPDiscretized = Discretized     # the name of the core type in julia/ClearSkyHIP.jl


class FluxPack:
    """core/shared.jl:73-106.  tau (np-1, nnu), Mup/Mdn (np, nnu) Fortran order; Fup, Fdn, Fnet (np)."""

    def __init__(self, npl, nnu):
        if not isinstance(npl, (int, np.integer)):
            npl, nnu = len(npl), len(nnu)
        self.tau = np.zeros((npl - 1, nnu), order="F")
        self.Mup = np.zeros((npl, nnu), order="F")
        self.Mdn = np.zeros((npl, nnu), order="F")
        self.Fup = np.zeros(npl)
        self.Fdn = np.zeros(npl)
        self.Fnet = np.zeros(npl)

    @property
    def size(self):
        return self.Mup.shape


def checkstreams(n):
    if n < 4:
        import warnings
        warnings.warn("careful! using nstream < 4 is likely to be inaccurate!")


# ----------------------------------------------------------------------------------------------------------------
# whole-column evaluation (B3)


class Column:
    """A column resident in HBM: inputs uploaded once, evaluated any number of times (cs_column_* in the C ABI).

    Pre-evaluates the closures exactly where monochromaticfluxes!(…, core::Discretized, …) does (fluxes.jl:250-267):
    lobattoevaluations (T, mu at Lobatto nodes), T at the levels for Planck, fC at the nodes, fS(nu), fa(nu).
    `nu_range=(j0, j1)` restricts the device work to a contiguous shard of the grid with the global trapezoid
    weights, so shards' Fup/Fdn simply add (multi-GPU).
    """

    def __init__(self, P, g, T, mu, fS, fa, *absorbers, core: Optional[Discretized] = None, theta_s: float = 0.841,
                 want_tau: bool = True, want_M: bool = True, nu_range=None, ctx: Optional[Context] = None, _setup: bool = True,
                 _warn: bool = True):
        core = core or Discretized()
        U, nu, nnu = unifyabsorbers(absorbers)
        self.accel = U if isinstance(U, AcceleratedAbsorber) else None
        if self.accel is not None:
            assert ctx is None or ctx is self.accel.ctx, "an AcceleratedAbsorber lives on the context it was built on"
            ctx = self.accel.ctx
            U = _NoAbsorbers(self.accel.nu)      # the slot stands for all absorbers (absorbers.jl:216)
        self.ctx = ctx or default_context()
        P = as_f64(P)
        assert np.all(np.diff(P) >= 0), "pressure coordinates must be in ascending order (sorted)"
        fT, fmu = formprofile(P, T), formprofile(P, mu)
        self._fmu = fmu            # kept: update()/run_batch() re-evaluate it at the new (T, P) nodes when no new mu is given
        self.U, self.core, self.P, self.g, self.theta_s = U, core, P, float(g), float(theta_s)
        assert 0 <= theta_s < math.pi / 2, "azimuth angle θ must be ∈ [0,π/2)"
        if _warn:
            checkstreams(core.nstream)
        self.np, self.nl = len(P), len(P) - 1
        nlob = core.nlobatto
        self.K = self.nl * (nlob - 1) + 1
        nu = as_f64(nu)
        # trapezoid weights on the global grid (util.jl:26-33 rewritten as per-point weights)
        w = trapz_weights(nu)
        j0, j1 = (0, nnu) if nu_range is None else nu_range
        self.j0, self.j1 = int(j0), int(j1)
        self.nu_all = nu
        self.nu = np.ascontiguousarray(nu[j0:j1])
        self.wts = np.ascontiguousarray(w[j0:j1])
        self.nnu = len(self.nu)
        self.Pk = nodepressures(P, nlob)
        self._fS, self._fa = fS, fa

        def evalv(f):   # fS(nu), fa(nu): discretized.jl:299,309
            if f is None:
                return None
            if callable(f):
                return as_f64([f(v) for v in self.nu])
            return np.full(self.nnu, float(f))

        self.S_toa = evalv(fS)
        self.albedo = evalv(fa)
        self.gases = [g_ for g_ in U.gas if isinstance(g_, DirectGas)]
        self.baked = [g_ for g_ in U.gas if isinstance(g_, Gas)]
        for g_ in self.baked:
            assert g_.ctx is self.ctx, "baked Gas objects live on the context they were baked on"
            # checkpressures, absorbers.jl:237-246
            assert P[-1] > P[0], "Pₛ must be greater than Pₜ"
            for p_ in (P[-1], P[0]):
                assert p_ >= g_.Omega.Pmin, f"Pressure {p_} Pa too low, domain minimum is {g_.Omega.Pmin}"
                assert p_ <= g_.Omega.Pmax, f"Pressure {p_} Pa too low, domain minimum is {g_.Omega.Pmax}"
        self.sigma_gray = float(sum(g_.sigma for g_ in U.gas if isinstance(g_, GrayGas)))
        for g_ in U.gas:
            if not isinstance(g_, (DirectGas, GrayGas, SemiGrayGas, Gas)):
                raise TypeError(f"unsupported gas type {type(g_).__name__} for the HIP Discretized core")
        self.slots = np.array(self.ctx.slots_of([g_.sl for g_ in self.gases]), dtype=np.int32)
        self.shapes = np.array([SHAPES[g_.shape] for g_ in self.gases], dtype=np.int32)
        self.cuts = as_f64([g_.dnu_cut for g_ in self.gases])
        self.want_tau, self.want_M = bool(want_tau), bool(want_M)
        self._set = False
        self._state(fT, fmu)
        if _setup:
            self._setup()

    # -- closures -> arrays ---------------------------------------------------------------------------------------
    def _state(self, fT, fmu):
        nlob = self.core.nlobatto
        self.Tn, self.mun = lobattoevaluations(self.P, fT, fmu, nlob)
        self.Tlev = as_f64([fT(p) for p in self.P])          # planckevaluations, discretized.jl:51
        self.Tk = nodevalues(self.Tn, nlob)
        self.muk = nodevalues(self.mun, nlob)
        ng = len(self.gases)
        conc = np.zeros((ng, self.K), order="F")
        for gi, g_ in enumerate(self.gases):
            for k in range(self.K):
                conc[gi, k] = g_.fC(self.Tk[k], self.Pk[k])
        self.conc = conc
        ct = np.zeros((len(self.baked), self.K), order="F")
        for ti, g_ in enumerate(self.baked):
            for k in range(self.K):
                ct[ti, k] = g_.fC(self.Tk[k], self.Pk[k])
        self.conc_tab = ct
        nc = len(self.U.cia)
        self.cia_P1 = np.zeros((nc, self.K), order="F")
        self.cia_P2 = np.zeros((nc, self.K), order="F")
        for ci, x in enumerate(self.U.cia):
            for k in range(self.K):
                self.cia_P1[ci, k] = self.Pk[k] * x.g1.concentration(self.Tk[k], self.Pk[k])   # cia…jl:378-382
                self.cia_P2[ci, k] = self.Pk[k] * x.g2.concentration(self.Tk[k], self.Pk[k])
        semi = [g_ for g_ in self.U.gas if isinstance(g_, SemiGrayGas)]
        if self.U.fun or semi:
            ex = np.zeros((self.K, self.nnu))
            for g_ in semi:                      # (nu[i] <= nu_cut) ? sigma : 0 at every node state, gases.jl:386
                ex += g_.vector(self.nu)[None, :]
            for k in range(self.K):
                for f in self.U.fun:
                    try:
                        ex[k] += np.asarray(f(self.nu, self.Tk[k], self.Pk[k]), float)
                    except Exception:
                        ex[k] += np.array([f(v, self.Tk[k], self.Pk[k]) for v in self.nu], float)
            self.sigma_extra = ex
        else:
            self.sigma_extra = None

    def _setup(self):
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int)) if len(a) else None
        check(lib().cs_column_setup(
            self.ctx.handle, self.nnu, dptr(self.nu), dptr(self.wts), self.np, dptr(self.P), self.g, self.core.nlobatto,
            dptr(np.asfortranarray(self.Tn).ravel(order="F").copy()), dptr(np.asfortranarray(self.mun).ravel(order="F").copy()),
            dptr(self.Tlev), len(self.gases), ip(self.slots), ip(self.shapes), dptr(self.cuts) if len(self.cuts) else None,
            dptr(self.conc.ravel(order="F").copy()) if self.conc.size else None, self.sigma_gray,
            dptr(self.sigma_extra) if self.sigma_extra is not None else None, dptr(self.S_toa), dptr(self.albedo),
            self.theta_s, self.core.nstream, int(self.want_tau), int(self.want_M)))
        if self.baked:
            slots = np.array([g_.slot for g_ in self.baked], dtype=np.int32)
            check(lib().cs_column_set_tables(self.ctx.handle, len(slots), slots.ctypes.data_as(C.POINTER(C.c_int)),
                                             dptr(self.conc_tab.ravel(order="F").copy())))
        self._set_cia()
        if self.accel is not None:
            check(lib().cs_column_set_accel(self.ctx.handle, self.accel.slot))
        self._set = True
        self.ctx._resident = self      # a context holds ONE resident column
        if getattr(self, "_flux_dst", 0):
            check(lib().cs_column_set_flux_dst(self.ctx.handle, C.c_void_p(self._flux_dst)))

    def _set_cia(self):
        if not self.U.cia:
            return
        slots = np.array([self.ctx.cia_slot(x.x) for x in self.U.cia], dtype=np.int32)
        flags = np.array([int(x.x.extrapolate) | (int(x.x.singles) << 1) for x in self.U.cia], dtype=np.int32)
        check(lib().cs_column_set_cia(self.ctx.handle, len(slots), slots.ctypes.data_as(C.POINTER(C.c_int)),
                                      flags.ctypes.data_as(C.POINTER(C.c_int)), dptr(self.cia_P1.ravel(order="F").copy()),
                                      dptr(self.cia_P2.ravel(order="F").copy())))

    def _ensure_resident(self):
        if getattr(self.ctx, "_resident", None) is not self:
            self._setup()

    def _require_resident(self, what):
        """Results live in the context's ONE resident column: reading them after another Column has been set up on the same
        context would return that column's data (and overrun buffers sized for this one).  Re-running setup here would discard
        the results, so this is an error."""
        if getattr(self.ctx, "_resident", None) is not self:
            raise ClearSkyHIPError(-6, f"{what}: this column is no longer resident on its context (another Column was set up on "
                                       "it); call run() again, or give each Column its own Context")

    def update(self, T, mu=None):
        """New temperature (and molar-mass) profile on the same grid: re-evaluates the closures and uploads the node
        states only (the RCM inner loop, radiative_convective.jl:109-113)."""
        fT = formprofile(self.P, T)
        if mu is not None:
            self._fmu = formprofile(self.P, mu)
        self._state(fT, self._fmu)     # mu=None: the column's own mu (number, profile or mu(T,P)) at the new nodes, discretized.jl:19-27
        if self.sigma_extra is not None or getattr(self.ctx, "_resident", None) is not self:
            self._setup()
            return
        check(lib().cs_column_update_state(self.ctx.handle, dptr(self.Tn.ravel(order="F").copy()),
                                           dptr(self.mun.ravel(order="F").copy()), dptr(self.Tlev),
                                           dptr(self.conc.ravel(order="F").copy()) if self.conc.size else None,
                                           dptr(self.conc_tab.ravel(order="F").copy()) if self.conc_tab.size else None))
        self._set_cia()

    def run_batch(self, Ts, mus=None):
        """Band fluxes of B temperature profiles on this column's grid in one device batch (the np+1 perturbed profiles of
        jacobian!, or the profiles of an RCM loop: radiative_convective.jl:109-171).  Ts: [B, np] level temperatures (or a
        list of callables fT(P)); mus: None (keep the molar mass), a number, or per-profile profiles.  Returns (Fup, Fdn) of
        shape [B, np].  Members: line-by-line gases, baked Gas objects, CIA pairs, the gray term, or an AcceleratedAbsorber
        (whose cross-sections are shared by all B states, as in jacobian!); function absorbers cannot be batched."""
        self._ensure_resident()
        B = len(Ts)
        nlob = self.core.nlobatto
        nn = nlob * self.nl
        ng, nt, nc = len(self.gases), len(self.baked), len(self.U.cia)
        Tn_all, mun_all = np.zeros((B, nn)), np.zeros((B, nn))
        Tlev_all = np.zeros((B, self.np))
        conc_all = np.zeros((B, max(ng, 1) * self.K))
        ctab_all = np.zeros((B, max(nt, 1) * self.K))
        P1_all, P2_all = np.zeros((B, max(nc, 1) * self.K)), np.zeros((B, max(nc, 1) * self.K))
        for b in range(B):
            fT = formprofile(self.P, Ts[b])
            mu_b = mus if (mus is None or np.ndim(mus) == 0) else mus[b]
            fmu = formprofile(self.P, mu_b) if mu_b is not None else self._fmu
            Tn, mun = lobattoevaluations(self.P, fT, fmu, nlob)
            Tk = nodevalues(Tn, nlob)
            Tn_all[b], mun_all[b] = Tn.ravel(order="F"), mun.ravel(order="F")
            Tlev_all[b] = [fT(p) for p in self.P]
            for arr, members, n in ((conc_all, self.gases, ng), (ctab_all, self.baked, nt)):
                if n:
                    cc = np.zeros((n, self.K), order="F")
                    for gi, g_ in enumerate(members):
                        for k in range(self.K):
                            cc[gi, k] = g_.fC(Tk[k], self.Pk[k])
                    arr[b] = cc.ravel(order="F")
            if nc:
                p1, p2 = np.zeros((nc, self.K), order="F"), np.zeros((nc, self.K), order="F")
                for ci, x in enumerate(self.U.cia):
                    for k in range(self.K):
                        p1[ci, k] = self.Pk[k] * x.g1.concentration(Tk[k], self.Pk[k])      # cia…jl:378-382
                        p2[ci, k] = self.Pk[k] * x.g2.concentration(Tk[k], self.Pk[k])
                P1_all[b], P2_all[b] = p1.ravel(order="F"), p2.ravel(order="F")
        Fup, Fdn = np.zeros((B, self.np)), np.zeros((B, self.np))
        check(lib().cs_column_batch(self.ctx.handle, B, dptr(Tn_all), dptr(mun_all), dptr(Tlev_all), dptr(conc_all),
                                    dptr(ctab_all) if nt else None, dptr(P1_all) if nc else None, dptr(P2_all) if nc else None,
                                    dptr(Fup), dptr(Fdn)))
        return Fup, Fdn

    # -- execution -------------------------------------------------------------------------------------------------
    def run(self, stream: int = 0):
        """Enqueue one evaluation (asynchronous).  `stream` is a raw hipStream_t (e.g. torch's cuda_stream) or 0."""
        self._ensure_resident()
        check(lib().cs_column_run(self.ctx.handle, C.c_void_p(stream) if stream else None))

    def sigma_run(self, stream: int = 0):
        """Only the cross-section stage: Sigma(absorbers, i, T_k, P_k) for every wavenumber and node (asynchronous)."""
        self._ensure_resident()
        check(lib().cs_column_sigma_run(self.ctx.handle, C.c_void_p(stream) if stream else None))

    def sync(self):
        check(lib().cs_column_sync(self.ctx.handle))

    def profile(self, reps: int = 3, stream: int = 0):
        """HIP-event time per kernel class, ms per evaluation, summed over gases: dict(prep, nodes, apply, far, near, rt, reduce,
        nodes_mx, far_mx, sub) = k_gas_setup (+ k_mxzones), k_cheb_nodes, k_cheb_apply, k_voigt_far (or k_linesum), k_voigt_near, k_rt,
        k_freduce, k_cheb_nodes_mx, k_voigt_edge_mx, k_voigt_sub."""
        ms = np.zeros(10)
        self._ensure_resident()
        check(lib().cs_column_profile(self.ctx.handle, C.c_void_p(stream) if stream else None, reps, dptr(ms)))
        return dict(prep=ms[0], nodes=ms[1], apply=ms[2], far=ms[3], near=ms[4], rt=ms[5], reduce=ms[6], nodes_mx=ms[7], far_mx=ms[8], sub=ms[9])

    def flux_ptr(self) -> int:
        self._require_resident("flux_ptr")
        p = C.c_void_p()
        check(lib().cs_column_flux_ptr(self.ctx.handle, C.byref(p)))
        return p.value

    def flux_to(self, device_ptr: int, stream: int = 0):
        """Async copy of [Fup; Fdn] (2*np doubles) into caller-owned device memory (e.g. a torch tensor's data_ptr())."""
        self._require_resident("flux_to")
        check(lib().cs_column_flux_to(self.ctx.handle, C.c_void_p(device_ptr), C.c_void_p(stream) if stream else None))

    def set_flux_dst(self, device_ptr: int):
        """From now on the column writes [Fup; Fdn] straight into caller-owned device memory (cs_column_set_flux_dst): the buffer a
        collective reduces in place -- no copy per step.  0 / None: back to the column's own."""
        self._flux_dst = int(device_ptr or 0)      # (re-applied whenever the column is set up again: _setup)
        self._ensure_resident()
        check(lib().cs_column_set_flux_dst(self.ctx.handle, C.c_void_p(self._flux_dst) if self._flux_dst else None))

    def counts(self):
        self._require_resident("counts")
        a, b = C.c_int64(), C.c_int64()
        check(lib().cs_column_counts(self.ctx.handle, C.byref(a), C.byref(b)))
        return dict(pair_evals=a.value, lines_in_range=b.value)

    def info(self):
        """dict(groups, launches, lines, merge, max_members): launch groups of the resident column and kernel launches of its last run"""
        self._require_resident("info")
        out = (C.c_int64 * 8)()
        check(lib().cs_column_info(self.ctx.handle, out))
        return dict(groups=out[0], launches=out[1], lines=out[2], merge=out[3], max_members=out[4], flux_form=out[5], near_launches=out[6],
                    line_kernel=out[7])

    def work(self):
        """Evaluations the last run issued for its Voigt gases: per-point, at interpolation nodes; levels in use."""
        self._require_resident("work")
        out = (C.c_int64 * 40)()
        check(lib().cs_column_work(self.ctx.handle, out))
        return dict(direct_evals=out[0], node_evals=out[1], levels=out[2], intervals=out[3],
                    direct_by_body=dict(zip(("t2", "t2_cut", "t3", "t3_cut", "t4_cut", "near_zone"), [out[4 + q] for q in range(6)])),
                    node_by_body=dict(zip(("t2", "t3", "t4"), [out[10 + q] for q in range(3)])), node_evals_matrix=out[13],
                    direct_evals_matrix=out[14], matrix_evals_3term=out[15], sub_evals=out[16], core_tile_states=out[17], matrix_evals_8term=out[18], node_evals_matrix_3term=out[19],
                    near_pairs_tier0=out[20], near_pairs_tier1=out[21], edge_mx_flops_useful=out[22], edge_mx_flops_issued=out[23],
                    nodes_mx_flops_useful=out[24], nodes_mx_flops_issued=out[25], apply_flops=out[26],
                    edge_mx_record_bytes_requested=out[32], nodes_mx_record_bytes_requested=out[33])

    def fetch(self, tau=None, Mup=None, Mdn=None):
        """Copy results to host.  Returns (Fup, Fdn); fills the optional Fortran-order matrices in place."""
        self._require_resident("fetch")
        Fup, Fdn = np.zeros(self.np), np.zeros(self.np)
        bufs = []
        ptrs = []
        for a, rows in ((tau, self.nl), (Mup, self.np), (Mdn, self.np)):
            if a is None:
                ptrs.append(None)
                bufs.append(None)
                continue
            assert a.shape == (rows, self.nnu), f"expected shape {(rows, self.nnu)}, got {a.shape}"
            b = a if (a.flags["F_CONTIGUOUS"] and a.dtype == np.float64) else np.zeros(a.shape, order="F")
            bufs.append(b)
            ptrs.append(b.ctypes.data_as(C.POINTER(C.c_double)))
        check(lib().cs_column_fetch(self.ctx.handle, self.nnu, self.np, ptrs[0], ptrs[1], ptrs[2], dptr(Fup), dptr(Fdn)))
        for a, b in zip((tau, Mup, Mdn), bufs):
            if a is not None and b is not a:
                a[...] = b
        return Fup, Fdn

    def sigma_nodes(self):
        """Total cross-section at the nodes, shape (K, nnu) (test hook)."""
        self._require_resident("sigma_nodes")
        out = np.zeros((self.K, self.nnu))
        check(lib().cs_column_sigma_fetch(self.ctx.handle, self.nnu, self.K, dptr(out)))
        return out


def _fluxes_discretized(col: "Column", tau, Mup, Mdn):
    """One call of cs_fluxes_discretized -- the symbol the Julia method monochromaticfluxes!(…, core::HIPDiscretized, …) binds
    (julia/ClearSkyHIP.jl) -- with the arrays laid out exactly as that ccall passes them: T/mu at the Lobatto nodes
    [nlobatto, np-1] and conc [ngas, K] column-major, tau/M+/M- column-major [level, nu] or NULL.  `col` only carries the
    pre-evaluated closures (Column(..., _setup=False)).  Returns (Fup, Fdn)."""
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int)) if len(a) else None
    fp = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))
    for a, rows in ((tau, col.nl), (Mup, col.np), (Mdn, col.np)):
        if a is not None:
            assert a.shape == (rows, col.nnu) and a.flags["F_CONTIGUOUS"] and a.dtype == np.float64, \
                f"expected a Fortran-order float64 array of shape {(rows, col.nnu)}"
    Fup, Fdn = np.zeros(col.np), np.zeros(col.np)
    if isinstance(col.ctx, MultiContext):      # `ngpu` contexts: cs_fluxes_discretized_multi, same arguments behind the context list
        for c_ in col.ctx.ctxs:
            c_._resident = None
        check(lib().cs_fluxes_discretized_multi(
            col.ctx.handles(), len(col.ctx.ctxs), col.nnu, dptr(col.nu), col.np, dptr(col.P), col.g, col.core.nlobatto,
            dptr(np.asfortranarray(col.Tn).ravel(order="F").copy()), dptr(np.asfortranarray(col.mun).ravel(order="F").copy()),
            dptr(col.Tlev), len(col.gases), ip(col.slots), ip(col.shapes), dptr(col.cuts) if len(col.cuts) else None,
            dptr(col.conc.ravel(order="F").copy()) if col.conc.size else None, col.sigma_gray,
            dptr(col.sigma_extra) if col.sigma_extra is not None else None, dptr(col.S_toa), dptr(col.albedo), col.theta_s,
            col.core.nstream, fp(tau), fp(Mup), fp(Mdn), dptr(Fup), dptr(Fdn)))
        return Fup, Fdn
    col.ctx._resident = None           # the call replaces (or re-uses) the context's resident column on the C side
    # members beyond the line-by-line gases go by slot (cs_fluxes_discretized_members; without any it is cs_fluxes_discretized)
    tslots = np.array([g_.slot for g_ in col.baked], dtype=np.int32)
    cslots = np.array([col.ctx.cia_slot(x.x) for x in col.U.cia], dtype=np.int32)
    cflags = np.array([int(x.x.extrapolate) | (int(x.x.singles) << 1) for x in col.U.cia], dtype=np.int32)
    args = (col.ctx.handle, col.nnu, dptr(col.nu), col.np, dptr(col.P), col.g, col.core.nlobatto,
            dptr(np.asfortranarray(col.Tn).ravel(order="F").copy()), dptr(np.asfortranarray(col.mun).ravel(order="F").copy()),
            dptr(col.Tlev), len(col.gases), ip(col.slots), ip(col.shapes), dptr(col.cuts) if len(col.cuts) else None,
            dptr(col.conc.ravel(order="F").copy()) if col.conc.size else None)
    tail = (col.sigma_gray, dptr(col.sigma_extra) if col.sigma_extra is not None else None, dptr(col.S_toa), dptr(col.albedo),
            col.theta_s, col.core.nstream, fp(tau), fp(Mup), fp(Mdn), dptr(Fup), dptr(Fdn))
    if len(tslots) or len(cslots) or col.accel is not None:
        check(lib().cs_fluxes_discretized_members(
            *args, len(tslots), ip(tslots), dptr(col.conc_tab.ravel(order="F").copy()) if len(tslots) else None,
            len(cslots), ip(cslots), ip(cflags), dptr(col.cia_P1.ravel(order="F").copy()) if len(cslots) else None,
            dptr(col.cia_P2.ravel(order="F").copy()) if len(cslots) else None, col.accel.slot if col.accel is not None else -1, *tail))
    else:
        check(lib().cs_fluxes_discretized(*args, *tail))
    return Fup, Fdn


def _b3(P, g, T, mu, fS, fa, absorbers, core, theta_s, ctx, tau, Mup, Mdn):
    """The B3 boundary, as the Julia method monochromaticfluxes!(…, core::HIPDiscretized, …) calls it: ONE host-pointer call per
    evaluation -- cs_fluxes_discretized for columns of line-by-line / gray / function absorbers, cs_fluxes_discretized_members when
    baked Gas objects, CIA pairs or an AcceleratedAbsorber are among the members (B2)."""
    direct = Column(P, g, T, mu, fS, fa, *absorbers, core=core, theta_s=theta_s, want_tau=tau is not None,
                    want_M=Mup is not None or Mdn is not None, ctx=ctx, _setup=False)
    if isinstance(direct.ctx, MultiContext) and (direct.baked or direct.U.cia or direct.accel is not None):
        raise TypeError("baked Gas objects, CIA pairs and accelerated absorbers live on ONE context: a MultiContext takes line-by-line, "
                        "gray and function absorbers")
    bufs = [None if a is None else (a if (a.flags["F_CONTIGUOUS"] and a.dtype == np.float64) else np.zeros(a.shape, order="F"))
            for a in (tau, Mup, Mdn)]
    F = _fluxes_discretized(direct, *bufs)
    for a, b in zip((tau, Mup, Mdn), bufs):
        if a is not None and b is not a:
            a[...] = b
    return F


def monochromaticfluxes_(Mup, Mdn, tau, core: Discretized, P, g, T, mu, fS, fa, *absorbers, theta_s=0.841, ctx=None):
    """monochromaticfluxes!(M+, M-, tau, core::Discretized, P, g, T, mu, fS, fa, absorbers...; theta_s) fluxes.jl:238-279"""
    _b3(P, g, T, mu, fS, fa, absorbers, core, theta_s, ctx, tau, Mup, Mdn)
    return None


def monochromaticfluxes(P, g, T, mu, fS, fa, *absorbers, core: Optional[Discretized] = None, theta_s=0.841, ctx=None):
    """fluxes.jl:281-306 -> (M+, M-)"""
    U, _, nnu = unifyabsorbers(absorbers)
    npl = len(P)
    Mup = np.zeros((npl, nnu), order="F")
    Mdn = np.zeros((npl, nnu), order="F")
    tau = np.zeros((npl - 1, nnu), order="F")
    monochromaticfluxes_(Mup, Mdn, tau, core or Discretized(), P, g, T, mu, fS, fa, U, theta_s=theta_s, ctx=ctx)
    return Mup, Mdn


def fluxes(P, g, T, mu, fS, fa, *absorbers, core: Optional[Discretized] = None, theta_s=0.841, ctx=None):
    """fluxes.jl:311-340 -> (F+, F-) [W/m^2] at every level; the nu-integral (intF!, shared.jl:125-137) runs on device."""
    return _b3(P, g, T, mu, fS, fa, absorbers, core, theta_s, ctx, None, None, None)


def netfluxes(P, g, T, mu, fS, fa, *absorbers, **kw):
    """fluxes.jl:342-352"""
    Fup, Fdn = fluxes(P, g, T, mu, fS, fa, *absorbers, **kw)
    return Fup - Fdn


# julia/ClearSkyHIP.jl's names for the two (there `fluxes` itself cannot dispatch on its `core` keyword and stays the reference's
# host-integrating body; here `fluxes` already is the band-fluxes-only call)
hipfluxes, hipnetfluxes = fluxes, netfluxes


def radiate_(F: FluxPack, core: Discretized, P, g, T, mu, fS, fa, *absorbers, theta_s=0.841, ctx=None):
    """radiate!(F, core, P, g, T, mu, fS, fa, absorbers...) fluxes.jl:357-383"""
    U, nu, nnu = unifyabsorbers(absorbers)
    assert F.size == (len(P), nnu), "size of FluxPack does not match number of pressure or wavenumber coordinates"
    if getattr(core, "fluxpack", "full") == "bands":     # F+, F-, Fnet from the device's intF!; tau, M+, M- neither copied nor touched
        Fup, Fdn = _b3(P, g, T, mu, fS, fa, (U,), core, theta_s, ctx, None, None, None)
    else:
        Fup, Fdn = _b3(P, g, T, mu, fS, fa, (U,), core, theta_s, ctx, F.tau, F.Mup, F.Mdn)
    F.Fup[:] = Fup
    F.Fdn[:] = Fdn
    F.Fnet[:] = Fup - Fdn
    return None


def radiate(P, g, T, mu, fS, fa, *absorbers, core: Optional[Discretized] = None, theta_s=0.841, ctx=None) -> FluxPack:
    """fluxes.jl:385-404.  OLR = F.Fup[0] (index 0 = top of atmosphere)."""
    U, nu, nnu = unifyabsorbers(absorbers)
    F = FluxPack(len(P), nnu)
    radiate_(F, core or Discretized(), P, g, T, mu, fS, fa, U, theta_s=theta_s, ctx=ctx)
    return F


def opticaldepth(P, g, T, mu, theta, *absorbers, nlobatto: int = 4, ctx=None):
    """opticaldepth(P::Vector, g, T, mu, theta, absorbers...; nlobatto=4) fluxes.jl:68-97: total slant optical depth per
    wavenumber via dDepth (discretized.jl:92-134, no 1e-6 floor).  The node cross-sections come from the device; the
    O(nnu*K) Lobatto sum is done here."""
    assert 0 <= theta < math.pi / 2, "azimuth angle θ must be ∈ [0,π/2)"
    P = np.sort(as_f64(P))
    col = Column(P, g, T, mu, None, None, *absorbers, core=Discretized(5, nlobatto), want_tau=False, want_M=False, ctx=ctx)
    col.sigma_run()
    sig = col.sigma_nodes()
    Cc = 1e-4 * K.Na / g
    _, ws = lobattonodes(nlobatto)
    m = 1 / math.cos(theta)
    beta = Cc * (sig / col.muk[:, None])
    tau = np.zeros(col.nnu)
    nl = len(P) - 1
    for i in range(nl):
        dP = P[i + 1] - P[i]
        ti = np.zeros(col.nnu)
        for n in range(nlobatto):
            ti = ti + (dP * ws[n]) * beta[i * (nlobatto - 1) + n]
        tau = tau + ti * m
    return tau


def transmittance(*args, **kw):
    """fluxes.jl:109"""
    return np.exp(-opticaldepth(*args, **kw))
