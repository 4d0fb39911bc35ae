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


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def olib():
    from oracle import oracle
    return oracle.lib()


def angle_close(a, b, tol):
    """compare angles modulo 2*pi"""
    d = np.abs(((np.asarray(a, np.float64) - np.asarray(b, np.float64)) + np.pi) % (2 * np.pi) - np.pi)
    return np.max(d) if d.size else 0.0
