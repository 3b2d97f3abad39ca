"""pytest configuration: the `gpu` marker and shared fixtures.

not-gpu tests: the CPU oracle against the golden vectors (and, in the dev container, against the compiled
reference), the host scene/BVH code, the C-ABI library's symbols.  gpu tests: the HIP path through the C ABI
against the oracle and the golden vectors.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold_unit():
    return np.load(os.path.join(GOLD, "unit_vectors.npz"))


@pytest.fixture(scope="session")
def gold_bvh():
    return np.load(os.path.join(GOLD, "bvh_vectors.npz"))


@pytest.fixture(scope="session")
def gold_small():
    return np.load(os.path.join(GOLD, "default_320x180.npz"))


@pytest.fixture(scope="session")
def gold_720():
    return np.load(os.path.join(GOLD, "default_1280x720_hits.npz"))


@pytest.fixture(scope="session")
def default_scene():
    from raytracer_amd import api
    return api.HostScene.default()


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a
