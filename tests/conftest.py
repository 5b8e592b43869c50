"""pytest configuration: the `gpu` marker and shared fixtures."""
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
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def synth_cloud(n, seed, scale=1.0):
    """BASELINE.md 4 generator: float32-grid coordinates in the unit cube + unit normals."""
    rng = np.random.default_rng(seed)
    p = rng.random((n, 3), dtype=np.float32).astype(np.float64) * scale
    nr = rng.standard_normal((n, 3))
    nr /= np.linalg.norm(nr, axis=1)[:, None]
    return p, nr, rng


def config1_cloud(n, seed):
    """BASELINE config 1 stand-in (tools/gen_golden_next.py): noisy sphere of the Stanford bunny's size, float32-grid
    coordinates, outward unit directions as stored normals."""
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    p = (0.5 + 0.5 * d * (1.0 + 0.01 * rng.standard_normal((n, 1)))).astype(np.float32).astype(np.float64)
    return p, d
