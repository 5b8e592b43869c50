#!/usr/bin/env python3
"""Time the SURVEY 8(f) rows that moved onto the device in round 2: voxel subsampling (csrc/voxel.hip) and the ICP
iteration (csrc/icp.hip).  Usage: bench_voxel_icp.py [N_POINTS]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shot_fpfh_amd as s
from shot_fpfh_amd.core import RigidTransform, voxel_closest_to_barycentre
from shot_fpfh_amd.icp import icp_point_to_plane, icp_point_to_point

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(0)
p = rng.random((n, 3), dtype=np.float32).astype(np.float64)
eng = s.default_engine()
for voxel in (0.003, 0.01):
    for order in ("numpy", "index"):
        voxel_closest_to_barycentre(p, voxel, within_voxel_order=order)
        eng.profile_reset(); eng.profile(True)
        t0 = time.perf_counter()
        picked, counts = voxel_closest_to_barycentre(p, voxel, within_voxel_order=order)
        t1 = time.perf_counter()
        eng.profile(False)
        dev = {k: round(v[1], 3) for k, v in eng.profile_report().items() if v[1] > 0}
        print(f"grid_subsampling n={n} voxel={voxel} order={order}: {t1 - t0:.4f} s host-to-host, {picked.size} voxels, device ms {dev}")

m = n // 5
ref = p
ang = 0.02
rot = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
scan = (ref[rng.permutation(n)[: n // 2]] - 0.5) @ rot.T + 0.5 + 0.002
nrm = rng.standard_normal((n, 3)); nrm /= np.linalg.norm(nrm, axis=1)[:, None]
for name, fn in (("point_to_point", lambda: icp_point_to_point(scan, ref, RigidTransform(), d_max=0.05, voxel_size=0.004, max_iter=20, rms_threshold=1e-12)),
                 ("point_to_plane", lambda: icp_point_to_plane(scan, ref, nrm, RigidTransform(), d_max=0.05, voxel_size=0.004, max_iter=20, rms_threshold=1e-12))):
    fn()
    eng.profile_reset(); eng.profile(True)
    t0 = time.perf_counter()
    tf, rms, ok = fn()
    t1 = time.perf_counter()
    eng.profile(False)
    dev = {k: (v[0], round(v[1], 2)) for k, v in eng.profile_report().items() if v[1] > 0}
    print(f"ICP {name} {scan.shape[0]} -> {n} points, 20 iterations: {t1 - t0:.3f} s; rot err {np.abs(tf.rotation - rot.T).max():.2e}; device (launches, ms) {dev}")
