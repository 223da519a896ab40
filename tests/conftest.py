import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def relinf(a, b):
    """||a-b||_inf / ||b||_inf  -- the tolerance metric of SURVEY.md 8d."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.max(np.abs(a - b)) if a.size else 0.0
    s = np.max(np.abs(b)) if b.size else 0.0
    return d / s if s > 0 else d


@pytest.fixture(scope="session")
def conv_golden():
    return np.load(os.path.join(GOLDEN, "conv_golden.npz"))


@pytest.fixture(scope="session")
def model_golden():
    return np.load(os.path.join(GOLDEN, "model_golden.npz"))


@pytest.fixture(scope="session")
def regression_golden():
    return np.load(os.path.join(GOLDEN, "regression_golden.npz"))


@pytest.fixture(scope="session")
def dataset_golden():
    return np.load(os.path.join(GOLDEN, "dataset_golden.npz"))
