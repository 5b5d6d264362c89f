"""pytest configuration: the ``gpu`` marker, repo root on sys.path, golden-fixture loaders."""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

#: tests/parameters.py:7-8 of the reference (values, not code)
T_VALUES = [(t1, t2) for t1 in [-0.1, 0.2, 0.3] for t2 in [-0.2, 0.5]]
KPT = [(0.1, 0.2, 0.7), (-0.3, 0.5, 0.2), (0.0, 0.0, 0.0), (0.1, -0.9, -0.7)]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as data:
        return {key: data[key] for key in data.files}


@pytest.fixture(scope="session")
def silicon():
    return load_golden("silicon")


@pytest.fixture(scope="session")
def toy():
    return load_golden("toy")


@pytest.fixture(scope="session")
def synthetic():
    return load_golden("synthetic")


@pytest.fixture(scope="session")
def kdotp_golden():
    return load_golden("kdotp")
