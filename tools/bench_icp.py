#!/usr/bin/env python3
"""Time one ICP run (icp_point_to_point_with_sampling's nearest-neighbour step = k-NN kernel with k = 1).
Usage: bench_icp.py N_POINTS"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shot_fpfh_amd as s
from shot_fpfh_amd.core import RigidTransform
from shot_fpfh_amd.icp import icp_point_to_point

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
rng = np.random.default_rng(0)
ref = rng.random((n, 3))
ang = 0.02
rot = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
scan = (ref[rng.permutation(n)[: n // 2]] - 0.5) @ rot.T + 0.5 + 0.002
eng = s.default_engine()
t0 = time.perf_counter()
out = icp_point_to_point(scan, ref, RigidTransform(), d_max=0.05, voxel_size=0.004, max_iter=20, rms_threshold=1e-9, disable_progress_bar=True)
t1 = time.perf_counter()
print(f"ICP {scan.shape[0]} -> {n} points: {t1 - t0:.3f} s total")
eng.profile_reset()
eng.profile(True)
t0 = time.perf_counter()
out = icp_point_to_point(scan, ref, RigidTransform(), d_max=0.05, voxel_size=0.004, max_iter=20, rms_threshold=1e-9, disable_progress_bar=True)
t1 = time.perf_counter()
eng.profile(False)
print(f"second run {t1 - t0:.3f} s; device kernels: { {k: (v[0], round(v[1], 2)) for k, v in eng.profile_report().items()} }")
