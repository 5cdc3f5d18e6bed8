import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hrir():
    """710 x 2 x 128 KEMAR table, libsndfile scaling (int16 / 32768)."""
    t = np.load(os.path.join(GOLD, "kemar_hrir_710x2x128_i16.npy"))
    return (t.astype(np.float32) / np.float32(32768.0)).astype(np.float32)


@pytest.fixture(scope="session")
def castanets():
    """2 s of the reference's default input, libsndfile scaling (int24 / 2^23)."""
    x = np.load(os.path.join(GOLD, "castanets_441_excerpt_i24.npy"))
    return (x.astype(np.float64) / 8388608.0).astype(np.float32)


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLD, "golden_scenarios.npz"))


@pytest.fixture(scope="session")
def jf():
    from jf_load import jf as mod
    return mod


def scenario_positions(azi0, ele0, n_dwell, n_rounds, r=0.5):
    """Position per block of a benchmarkTesting scenario (precision_test.cu:2093-2152)."""
    out = []
    azi = float(azi0)
    out += [(ele0, azi, r)] * n_dwell
    for _ in range(n_rounds):
        azi += 5
        if azi >= 360:
            azi -= 360
        out += [(ele0, azi, r)] * n_dwell
    return out
