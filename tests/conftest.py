import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
HITRAN = os.path.join(GOLDEN, "hitran")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return lambda name: np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def cs():
    import clearsky_jl_amd
    return clearsky_jl_amd


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def lines(cs):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = cs.SpectralLines(os.path.join(HITRAN, name + ".par"))
        return cache[name]
    return get


def relerr(a, b, floor=0.0):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


def source_rounding_bound(cs, nu, Tlev, tau):
    """How far the monochromatic fluxes of two evaluations may differ when their layer optical depths differ only in the last bit:
    the source term (1 - t)(B1 - B2)/tau of layerplanck (discretized.jl:85-87) divides a difference of order tau by tau, so a
    transmission t = exp(-tau m) that rounds the other way (2^-53) moves it by 2^-53 / tau x |B1 - B2| -- 1e-10 |dB| just above the
    1e-6 floor -- per layer and stream (weights sum to pi).  Rounding of the reference's own formula, not of the line sums."""
    B = cs.planck(np.asarray(nu)[None, :], np.asarray(Tlev)[:, None])
    return float(np.max(np.sum(np.pi * np.abs(np.diff(B, axis=0)) * 2.0 ** -53 / np.asarray(tau), axis=0)))
