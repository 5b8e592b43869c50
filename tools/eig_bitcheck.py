#!/usr/bin/env python3
"""Bitwise A/B of everything that goes through the 3x3 eigen-solver (K3 normals / PCA, K4 frames) between two library
builds: `eig_bitcheck.py dump OUT.npz` writes the outputs of the library currently in place, `eig_bitcheck.py cmp A.npz B.npz`
compares.  Used when the solver's control flow is restructured: the arithmetic per matrix must not change by one bit."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def clouds():
    rng = np.random.default_rng(2024)
    yield "uniform", rng.random((300000, 3), dtype=np.float32).astype(np.float64), 0.02
    uv = rng.random((200000, 2))
    yield "rough_plane", np.column_stack([uv, 0.3 * uv[:, 0] - 0.2 * uv[:, 1] + 1e-3 * rng.standard_normal(200000)]), 0.012
    g = np.arange(40) / 64.0
    yield "lattice", np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3), float(np.sqrt(5.0) / 64.0) * 1.0000001
    yield "plane", np.column_stack([rng.random((100000, 2)), np.full(100000, 0.25)]), 0.02
    t = rng.random((50000, 1))
    yield "line+cloud", np.vstack([t * np.array([[1.0, 0.5, 0.25]]), rng.random((50000, 3))]), 0.03
    c = rng.random((6, 3))
    yield "clustered", np.vstack([c[rng.integers(0, 6, 150000)] + 0.02 * rng.standard_normal((150000, 3)), rng.random((50000, 3))]), 0.01


def dump(path):
    import shot_fpfh_amd as s

    eng = s.default_engine()
    out = {}
    for name, p, r in clouds():
        cloud = eng.cloud(p)
        nb = cloud.radius_search(p, r)
        out[name + "_normals_r"] = nb.normals()
        w, v = nb.pca()[:2]
        out[name + "_w"], out[name + "_v"] = w, v
        out[name + "_lrf"] = nb.shot_lrf()
        nk = cloud.knn_search(p[:50000], 12)
        out[name + "_normals_k"] = nk.normals()
    np.savez(path, **out)


def cmp(a, b):
    A, B = np.load(a), np.load(b)
    bad = 0
    for k in A.files:
        same = np.array_equal(A[k], B[k], equal_nan=True)
        print(f"{k:24s} {A[k].shape} {'identical' if same else 'DIFFERENT: %d entries' % int((A[k] != B[k]).sum())}")
        bad += 0 if same else 1
    return bad


if __name__ == "__main__":
    sys.exit(dump(sys.argv[2]) if sys.argv[1] == "dump" else cmp(sys.argv[2], sys.argv[3]))
