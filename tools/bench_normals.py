#!/usr/bin/env python3
"""compute_normals on the C3 cloud, both branches of pca_based_descriptors.py:45-49: k-NN (the CLI's default, k = 30) and
radius.  Resident timings with the per-kernel breakdown.  Usage: bench_normals.py [n_points] [k] [radius]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shot_fpfh_amd as s

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 30
radius = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
rng = np.random.default_rng(3)
p = rng.random((n, 3), dtype=np.float32).astype(np.float64)
eng = s.default_engine()
cloud = eng.cloud(p)
out = eng.empty((n, 3))
for name, search in (("k-NN k=%d" % k, lambda: cloud.knn_search(p, k)), ("radius %.3f" % radius, lambda: cloud.radius_search(p, radius))):
    nb = search(); nb.normals(out=out); nb.free(); eng.sync()
    eng.profile_reset(); eng.profile(True)
    t0 = time.perf_counter()
    nb = search(); nb.normals(out=out); nb.free(); eng.sync()
    t = time.perf_counter() - t0
    eng.profile(False)
    rep = {kk: round(v[1], 3) for kk, v in eng.profile_report().items() if v[1] > 0.005}
    print(f"{name}: {1e3 * t:.2f} ms for {n} queries (host queries uploaded each time) {rep}")
