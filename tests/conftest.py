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
