"""ctypes binding of the C ABI in include/clearsky_hip.h (libclearsky_hip.so, built in-tree by build_native()).

There is no CPU fallback: if the HIP library is missing or no GPU is visible, every compute entry point raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libclearsky_hip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "clearsky_hip.h")            # the product surface
HEADER_DEV = os.path.join(os.path.dirname(_HERE), "include", "clearsky_hip_dev.h")    # measurement / tuning / test hooks of the same library

CS_MAX_GAS = 16
CS_MAX_TABLE = 16
CS_MAX_CIA = 8
CS_MAX_ACCEL = 4
CHEB_LD = 16
SHAPES = {"voigt": 0, "lorentz": 1, "doppler": 2, "PHCO2": 3, "phco2": 3}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p

# name -> (restype, argtypes); must list every symbol the two headers declare
SIGNATURES = {
    "cs_version": (C.c_int, []),
    "cs_build_id": (C.c_char_p, []),
    "cs_last_error": (C.c_char_p, []),
    "cs_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "cs_destroy": (None, [_vp]),
    "cs_gas_upload": (C.c_int, [_vp, C.c_int, C.c_int64, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_int16),
                                C.c_int, C.POINTER(C.c_int32), _dp]),
    "cs_gas_clear": (C.c_int, [_vp, C.c_int]),
    "cs_set_precision": (C.c_int, [_vp, C.c_int, C.c_double]),
    "cs_set_interp": (C.c_int, [_vp, C.c_int]),
    "cs_set_interp_plan": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "cs_set_matrix_cores": (C.c_int, [_vp, C.c_int]),
    "cs_set_merge": (C.c_int, [_vp, C.c_int]),
    "cs_set_tuning": (C.c_int, [_vp, C.c_int, C.c_int]),
    "cs_column_info": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "cs_shape_batch": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_int64, _dp, C.c_int, _dp, _dp, _dp, _dp,
                                 C.c_int64]),
    "cs_bake": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int64, _dp, C.c_int, _dp, C.c_int, _dp, _dp, _dp]),
    "cs_table_clear": (C.c_int, [_vp, C.c_int]),
    "cs_table_eval": (C.c_int, [_vp, C.c_int, C.c_double, C.c_double, C.c_int64, C.c_int64, _dp]),
    "cs_cia_begin": (C.c_int, [_vp, C.c_int, C.c_int]),
    "cs_cia_band": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp, C.c_int, _dp, _dp]),
    "cs_cia_clear": (C.c_int, [_vp, C.c_int]),
    "cs_column_set_cia": (C.c_int, [_vp, C.c_int, _ip, _ip, _dp, _dp]),
    "cs_column_set_tables": (C.c_int, [_vp, C.c_int, _ip, _dp]),
    "cs_fluxes_discretized": (C.c_int, [_vp, C.c_int64, _dp, C.c_int, _dp, C.c_double, C.c_int, _dp, _dp, _dp, C.c_int,
                                        _ip, _ip, _dp, _dp, C.c_double, _dp, _dp, _dp, C.c_double, C.c_int, _dp, _dp,
                                        _dp, _dp, _dp]),
    "cs_fluxes_discretized_multi": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int64, _dp, C.c_int, _dp, C.c_double, C.c_int, _dp, _dp, _dp, C.c_int,
                                              _ip, _ip, _dp, _dp, C.c_double, _dp, _dp, _dp, C.c_double, C.c_int, _dp, _dp,
                                              _dp, _dp, _dp]),
    "cs_fluxes_discretized_members": (C.c_int, [_vp, C.c_int64, _dp, C.c_int, _dp, C.c_double, C.c_int, _dp, _dp, _dp, C.c_int,
                                                _ip, _ip, _dp, _dp, C.c_int, _ip, _dp, C.c_int, _ip, _ip, _dp, _dp, C.c_int,
                                                C.c_double, _dp, _dp, _dp, C.c_double, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "cs_table_upload": (C.c_int, [_vp, C.c_int, C.c_int64, _dp, C.c_int, _dp, C.c_int, _dp, _dp]),
    "cs_accel_upload": (C.c_int, [_vp, C.c_int, C.c_int64, _dp, C.c_int, _dp, _dp]),
    "cs_accel_fetch": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int, _dp]),
    "cs_balanced_ranges": (C.c_int, [C.c_int64, _dp, C.c_int, C.POINTER(C.c_int64), C.POINTER(_dp), C.c_int, C.POINTER(C.c_int64)]),
    "cs_rebalance_ranges": (C.c_int, [C.c_int64, _dp, C.c_int, C.POINTER(C.c_int64), C.POINTER(_dp), C.c_int, C.POINTER(C.c_int64), _dp, C.c_double,
                                     C.POINTER(C.c_int64)]),
    "cs_column_setup": (C.c_int, [_vp, C.c_int64, _dp, _dp, C.c_int, _dp, C.c_double, C.c_int, _dp, _dp, _dp, C.c_int,
                                  _ip, _ip, _dp, _dp, C.c_double, _dp, _dp, _dp, C.c_double, C.c_int, C.c_int, C.c_int]),
    "cs_column_run": (C.c_int, [_vp, _vp]),
    "cs_column_sync": (C.c_int, [_vp]),
    "cs_column_profile": (C.c_int, [_vp, _vp, C.c_int, _dp]),
    "cs_column_flux_ptr": (C.c_int, [_vp, C.POINTER(_vp)]),
    "cs_column_flux_to": (C.c_int, [_vp, _vp, _vp]),
    "cs_column_set_flux_dst": (C.c_int, [_vp, _vp]),
    "cs_column_fetch": (C.c_int, [_vp, C.c_int64, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "cs_column_sigma_fetch": (C.c_int, [_vp, C.c_int64, C.c_int, _dp]),
    "cs_column_counts": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "cs_column_work": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "cs_interp_plan": (C.c_int, [C.c_int64, _dp, C.c_double, _ip]),
    "cs_phco2_plan": (C.c_int, [C.c_int64, _dp, C.c_double, C.c_int, _ip, _ip, _ip]),
    "cs_column_batch": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "cs_shape_points": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_int64, _dp, C.c_int, _dp, _dp, _dp, _dp, C.c_int64]),
    "cs_column_sigma_run": (C.c_int, [_vp, _vp]),
    "cs_accel_store": (C.c_int, [_vp, C.c_int]),
    "cs_accel_clear": (C.c_int, [_vp, C.c_int]),
    "cs_accel_eval": (C.c_int, [_vp, C.c_int, C.c_double, C.c_int64, C.c_int64, _dp]),
    "cs_column_set_accel": (C.c_int, [_vp, C.c_int]),
    "cs_column_update_state": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _dp]),
    "cs_par_count": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64)]),
    "cs_par_parse": (C.c_int, [C.c_char_p, C.c_int64, C.POINTER(C.c_int16), C.c_char_p, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "cs_gas_upload_par": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_int), C.c_int, C.c_int64,
                                    C.c_int, _dp, C.c_int, C.POINTER(C.c_int32), _dp, C.POINTER(C.c_int64)]),
    "cs_gas_fetch": (C.c_int, [_vp, C.c_int, C.c_int64, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_int16)]),
    "cs_streamnodes": (C.c_int, [C.c_int, _dp, _dp]),
    "cs_lobattonodes": (C.c_int, [C.c_int, _dp, _dp]),
    "cs_faddeeva_batch": (C.c_int, [_vp, C.c_int64, _dp, _dp, _dp]),
    "cs_devfn_batch": (C.c_int, [_vp, C.c_int, C.c_int64, _dp, _dp, _dp, _dp]),
}


class ClearSkyHIPError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[clearsky_hip {code}] {msg}")
        self.code = code


def source_id() -> str:
    """sha256 (16 hex digits) of the sources the library is built from -- compiled into it as cs_build_id(), so that a measurement
    can name the binary it was taken on (bench.py's kernel_source_sha16, profiles/pmc_traffic.json)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(HEADER, "rb").read())
    h.update(open(HEADER_DEV, "rb").read())
    return h.hexdigest()[:16]


def built_id():
    """cs_build_id() of the library file in the tree (None if it is missing or does not load).  Asked of a child process, so that a
    library about to be rebuilt is never mapped into this one."""
    if not os.path.exists(LIB_PATH):
        return None
    import sys
    code = f"import ctypes; L = ctypes.CDLL({LIB_PATH!r}); L.cs_build_id.restype = ctypes.c_char_p; print(L.cs_build_id().decode())"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    return out.stdout.strip() if out.returncode == 0 else None


def build_native(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into csrc/libclearsky_hip.so with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, "cs_api.hip")]
    if not force and os.path.exists(LIB_PATH) and built_id() == source_id():
        return LIB_PATH      # the binary says which sources it was compiled from (cs_build_id): file dates are not consulted
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f'-DCS_BUILD_ID="{source_id()}"', "-o", LIB_PATH] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB_PATH


_lib = None


def lib():
    """Load the shared library (raises if it has not been built -- no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ClearSkyHIPError(-100, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                         "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise ClearSkyHIPError(rc, lib().cs_last_error().decode("utf-8", "replace"))


def dptr(a):
    """double* of a C-contiguous float64 array (or NULL for None)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
