#!/usr/bin/env python3
"""Bitwise A/B of the integer SPFH table (K6) and of FPFH between two library builds: `spfh_bitcheck.py dump OUT.npz` writes
the tables of the library currently in place for a set of clouds (uniform, clustered, exact plane with normals along its own,
lattice, non-unit normals, millimetre scale), `spfh_bitcheck.py cmp A.npz B.npz` compares.  Used when K6's feature arithmetic
is restructured: the bins must not change by one count."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def clouds():
    rng = np.random.default_rng(77)

    def unit(v):
        return v / np.linalg.norm(v, axis=1)[:, None]

    p = rng.random((200000, 3), dtype=np.float32).astype(np.float64)
    yield "uniform", p, unit(rng.standard_normal(p.shape)), 0.03
    yield "uniform_long_normals", p[:50000], 3.0 * unit(rng.standard_normal((50000, 3))), 0.05
    c = rng.random((6, 3))
    pc = np.vstack([c[rng.integers(0, 6, 80000)] + 0.02 * rng.standard_normal((80000, 3)), rng.random((20000, 3))])
    yield "clustered", pc, unit(rng.standard_normal(pc.shape)), 0.012
    pl = np.column_stack([rng.random((60000, 2)), np.full(60000, 0.25)])
    npl = unit(rng.standard_normal(pl.shape))
    npl[:30000] = [0.0, 0.0, 1.0]
    yield "plane", pl, npl, 0.02
    g = np.arange(32) / 64.0
    lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    nl = unit(rng.standard_normal(lat.shape))
    nl[::3] = [1.0, 0.0, 0.0]
    yield "lattice", lat, nl, float(np.sqrt(5.0) / 64.0) * 1.0000001
    yield "mm_scale", p[:60000] * 1000.0, unit(rng.standard_normal((60000, 3))), 40.0


def dump(path):
    import shot_fpfh_amd as s

    eng = s.default_engine()
    out = {}
    for name, p, nr, r in clouds():
        cloud = eng.cloud(p, nr)
        cloud.build_grid(r)
        full = cloud.radius_search_self(r)
        for nb in (5, 4, 3):
            sp = eng.spfh(cloud, nb, full.max_count)
            sp.compute(full)
            out[f"{name}_spfh{nb}"] = sp.export()
            out[f"{name}_fpfh{nb}"] = sp.fpfh(full)
    np.savez(path, **out)


def cmp(a, b):
    A, B = np.load(a), np.load(b)
    bad = 0
    for k in A.files:
        same = np.array_equal(A[k], B[k])
        print(f"{k:32s} {A[k].shape} {'identical' if same else 'DIFFERENT: %d entries' % int((A[k] != B[k]).sum())}")
        bad += 0 if same else 1
    return bad


if __name__ == "__main__":
    sys.exit(dump(sys.argv[2]) if sys.argv[1] == "dump" else cmp(sys.argv[2], sys.argv[3]))
