"""Shared test plumbing: paths, the `gpu` marker, golden-fixture loader."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "1d-burgers-equation-roms_amd")
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip(f"fixture {name} not generated")
    return np.load(path)


def rel_l2(a, b):
    """The reference's own error metric (relative l2 / Frobenius, POD/Results_thesis/max_error.py:31-48)."""
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b)))


def mesh(n, a=0.0, b=100.0):
    X = np.linspace(a, b, n)
    T = np.array([np.arange(1, n), np.arange(2, n + 1)]).T
    return X, T


@pytest.fixture(scope="session")
def hip():
    """The loaded HIP library front end; GPU tests fail loudly if it is missing."""
    import torch
    assert torch.cuda.is_available(), "GPU test selected but no HIP device is visible"
    from burgers_hip import lib
    lib.load()
    return lib
