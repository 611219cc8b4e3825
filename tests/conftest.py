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


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a CPU-only box: GPU tests are skipped, not failed (the driver selects with
    -m gpu / -m "not gpu"; on the GPU box nothing is skipped -- a missing library must fail)."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def fx_tiny():
    return load_golden("fx_tiny")


@pytest.fixture(scope="session")
def fx_m16():
    return load_golden("fx_m16")


@pytest.fixture(scope="session")
def fx_container():
    return load_golden("fx_container")


@pytest.fixture(scope="session")
def fx_kmeans():
    return load_golden("fx_kmeans")


@pytest.fixture(scope="session")
def fx_kmeans_fit():
    return load_golden("fx_kmeans_fit")


@pytest.fixture(scope="session")
def fx_cosine():
    return load_golden("fx_cosine")


@pytest.fixture(scope="session")
def fx_residual():
    return load_golden("fx_residual")


@pytest.fixture(scope="session")
def fx_c1():
    return load_golden("fx_c1")


@pytest.fixture(scope="session")
def fx_ties():
    return load_golden("fx_ties")


@pytest.fixture(scope="session")
def fx_tomb():
    return load_golden("fx_tomb")


@pytest.fixture(scope="session")
def fx_layout():
    return load_golden("fx_layout")
