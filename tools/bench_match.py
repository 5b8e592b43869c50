#!/usr/bin/env python3
"""Time K8 (sf_match_argmin, device-resident) and K9 on synthetic descriptors.  Usage: bench_match.py M1 M2 D"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shot_fpfh_amd as s

m1, m2, d = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (20000, 20000, 352)
eng = s.default_engine()
rng = np.random.default_rng(0)
a = rng.random((m1, d))
a /= np.linalg.norm(a, axis=1)[:, None]
b = a[rng.permutation(m1)[:m2]] + 1e-3 * rng.standard_normal((m2, d)) if m2 <= m1 else rng.random((m2, d))
da, db = eng.empty((m1, d)).from_host(a), eng.empty((m2, d)).from_host(b)
idx, dist = eng.empty((m1,), np.int64), eng.empty((m1,), np.float64)
eng.match_argmin_device(da, db, idx, dist)
eng.sync()
eng.profile_reset()
eng.profile(True)
t0 = time.perf_counter()
eng.match_argmin_device(da, db, idx, dist)
eng.sync()
t = time.perf_counter() - t0
eng.profile(False)
print(f"K8 {m1}x{m2}x{d}: {t*1e3:.2f} ms, {m1*m2/t/1e9:.2f} G pair-dists/s, {3*m1*m2*d/t/1e12:.2f} TFLOP/s (difference form)")
print(eng.profile_report())
if os.environ.get("SF_BENCH_COMPARE"):  # same problem through the other matrix-core path: results must be identical
    i1, d1 = idx.to_host().copy(), dist.to_host().copy()
    prev = os.environ.get("SF_MATCH_HALF")
    os.environ["SF_MATCH_HALF"] = "0" if prev != "0" else "1"
    eng.profile_reset()
    eng.profile(True)
    t0 = time.perf_counter()
    eng.match_argmin_device(da, db, idx, dist)
    eng.sync()
    t2 = time.perf_counter() - t0
    eng.profile(False)
    print(f"other path (SF_MATCH_HALF={os.environ['SF_MATCH_HALF']}): {t2*1e3:.2f} ms; identical idx {np.array_equal(i1, idx.to_host())}, "
          f"identical dist {np.array_equal(d1, dist.to_host())}")
    print(eng.profile_report())
