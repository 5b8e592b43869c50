"""pytest configuration: the `gpu` marker and shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


_LAUNCHER = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Tests that run whole programs on the GPU (bench.py with several ranks) must not start them from this process once it
    # holds the GPU itself: a helper that never touches the GPU is started NOW, before any test has run, and starts them.
    global _LAUNCHER
    expr = config.getoption("-m", default="") or ""
    if "gpu" in expr and "not gpu" not in expr:
        import subprocess

        _LAUNCHER = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_launcher.py")], stdin=subprocess.PIPE,
                                     stdout=subprocess.PIPE, text=True, cwd=ROOT)


def pytest_unconfigure(config):
    global _LAUNCHER
    if _LAUNCHER is not None:
        try:
            _LAUNCHER.stdin.close()
            _LAUNCHER.wait(timeout=30)
        except Exception:  # noqa: BLE001
            _LAUNCHER.kill()
        _LAUNCHER = None


def run_program(argv, env=None, timeout=900):
    """Run a program through the GPU-free launcher: {"rc", "stdout", "stderr"}.  Only in `-m gpu` sessions."""
    import json

    if _LAUNCHER is None:
        pytest.skip("no launcher: not a -m gpu session")
    _LAUNCHER.stdin.write(json.dumps({"argv": list(argv), "env": env or {}, "timeout": timeout, "cwd": ROOT}) + "\n")
    _LAUNCHER.stdin.flush()
    return json.loads(_LAUNCHER.stdout.readline())


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


# ---- degenerate cloud families (tests/test_hip_fuzz.py, tools/gen_golden_r2b.py) ----------------------------------
def _unit(v):
    return v / np.linalg.norm(v, axis=1)[:, None]


def _f32grid(a):
    return np.ascontiguousarray(a, dtype=np.float32).astype(np.float64)


def family(name, n, rng):
    """(points, normals, has_distance_ties, is_flat)"""
    if name == "uniform":
        p = rng.random((n, 3))
    elif name == "clustered":  # a few tight blobs in a sparse background: densities two orders of magnitude apart
        centres = rng.random((6, 3))
        p = np.vstack([centres[rng.integers(0, 6, n - n // 5)] + 0.02 * rng.standard_normal((n - n // 5, 3)),
                       rng.random((n // 5, 3))])
    elif name == "lattice":  # spacing 1/16: squared distances are exact multiples of 2^-8 -> ties and on-radius points
        side = int(round(n ** (1 / 3)))
        g = np.arange(side) / 16.0
        p = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    elif name == "plane":  # exactly z = 0.25
        p = np.column_stack([rng.random((n, 2)), np.full(n, 0.25)])
    elif name == "rough_plane":  # a tilted plane with 1e-3 roughness: flat but with a well-defined frame
        uv = rng.random((n, 2))
        p = np.column_stack([uv, 0.3 * uv[:, 0] - 0.2 * uv[:, 1] + 1e-3 * rng.standard_normal(n)])
    elif name == "line":  # collinear points plus a little cloud around them so that lists are not all degenerate
        t = rng.random((n, 1))
        p = np.vstack([(t * np.array([[1.0, 0.5, 0.25]]))[: n // 2], rng.random((n - n // 2, 3))])
    elif name == "duplicates":  # a fifth of the points occur two or three times
        base = rng.random((n - 2 * (n // 5), 3))
        p = np.vstack([base, base[: n // 5], base[: n // 5]])
    elif name == "far_origin":  # 4 decimal digits of the mantissa eaten by the offset
        p = rng.random((n, 3)) + np.array([[4096.0, -2048.0, 1024.0]])
    elif name == "slab":  # one cell thick along z, a few along y
        p = rng.random((n, 3)) * np.array([[1.0, 0.3, 0.004]])
    else:
        raise KeyError(name)
    p = _f32grid(p) if name not in ("far_origin",) else np.ascontiguousarray(p)
    nr = _unit(rng.standard_normal((p.shape[0], 3)))
    if name in ("plane", "rough_plane"):
        nr[: p.shape[0] // 2] = np.array([0.0, 0.0, 1.0])  # half the normals exactly along the plane's: alpha, theta on edges
    return p, nr, name in ("lattice", "duplicates"), name in ("plane", "line", "lattice", "slab")


FAMILIES = ["uniform", "clustered", "lattice", "plane", "rough_plane", "line", "duplicates", "far_origin", "slab"]


def long_list_cloud():
    """1 500 uniform points + a 320-point blob of diameter 0.1 near z = 0.05: at radius 0.15 every blob point has more than 255
    neighbours, and in a z-slab sharding only the first rank owns such a list."""
    p, nr, _ = synth_cloud(1500, 41)
    rng = np.random.default_rng(43)
    blob = (np.array([0.5, 0.5, 0.06]) + (rng.random((320, 3), dtype=np.float32).astype(np.float64) - 0.5) * 0.1)
    bn = rng.standard_normal((320, 3))
    bn /= np.linalg.norm(bn, axis=1)[:, None]
    return np.vstack([p, blob]), np.vstack([nr, bn])
