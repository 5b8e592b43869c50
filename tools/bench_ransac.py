#!/usr/bin/env python3
"""Time K9 (sf_ransac_score) and the host side of ransac_on_matches.  Usage: bench_ransac.py N_MATCHES N_DRAWS"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shot_fpfh_amd as s
from shot_fpfh_amd.core import solver_point_to_point

m, nd = (int(x) for x in sys.argv[1:3]) if len(sys.argv) > 2 else (1_000_000, 10_000)
eng = s.default_engine()
rng = np.random.default_rng(0)
a = rng.random((m, 3))
q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
q *= np.sign(np.linalg.det(q))
b = a @ q.T + 0.3 + 0.004 * rng.standard_normal((m, 3))
t0 = time.perf_counter()
draws = [rng.choice(m, 4, replace=False, shuffle=False) for _ in range(nd)]
t1 = time.perf_counter()
rec = np.array([solver_point_to_point(a[d], b[d]).as_row12() for d in draws])
t2 = time.perf_counter()
eng.ransac_score(a, b, rec[:16], 0.01)
eng.profile_reset()
eng.profile(True)
t3 = time.perf_counter()
inl = eng.ransac_score(a, b, rec, 0.01)
t4 = time.perf_counter()
eng.profile(False)
print(f"host draws {t1 - t0:.3f} s, host Kabsch {t2 - t1:.3f} s, K9 call (host buffers) {t4 - t3:.3f} s; best {inl.max()} of {m}")
print(eng.profile_report())
ref = [(np.linalg.norm(a @ r[:9].reshape(3, 3).T + r[9:] - b, axis=1) <= 0.01).sum() for r in rec[:20]]
print("first 20 draws equal NumPy:", np.array_equal(inl[:20], ref))
