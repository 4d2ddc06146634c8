import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Decoded Tsukuba fixtures (reference stereo_matching_cuda/data/*.png, see tools/make_golden.py)."""
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "tsukuba_golden.npz")))


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def tsukuba_gray(golden, orc):
    return orc.gray(golden["tsukuba0"]), orc.gray(golden["tsukuba1"])


@pytest.fixture(scope="session")
def tsukuba_oracle(tsukuba_gray, orc):
    Il, Ir = tsukuba_gray
    return orc.stereo_pair(Il, Ir, 16, dminl=-15, dminr=0, want_cost=True, want_agg=True)
