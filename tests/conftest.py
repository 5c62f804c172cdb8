import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))


@pytest.fixture(scope="session")
def golden_rectify():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_rectify_v1.npz"))


@pytest.fixture(scope="session")
def oracle():
    import oracle_py

    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def rsdsfm():
    import rsdsfm as pkg

    return pkg


GOLDEN_CASES = ["clean_k0", "noisy_k0", "deepflow_k0", "clean_k04", "noisy_k04"]
RECTIFY_CASES = ["k0", "k04"]
