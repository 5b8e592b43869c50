#!/usr/bin/env python3
"""BASELINE config 4 at full size on one MI355X: two 1M-point clouds (the second a permuted rigid copy of the
first), SHOT descriptors for every point, brute-force L2 matching 1M x 1M x 352, RANSAC (10 000 draws) on the
matches.  Everything between the upload of the clouds and the read-back of the match indices is device-resident.
Usage: run_config4.py [n_points] [n_draws]
"""
import os
import sys
import time

import numpy as np
from scipy.spatial.transform import Rotation

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shot_fpfh_amd as s
import shot_fpfh_amd.matching.ransac as R
from shot_fpfh_amd.sharding import DescriptorJob, MatchJob

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_draws = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
radius = 0.03 * (1_000_000 / n) ** (1 / 3)
rng = np.random.default_rng(4)
scan = rng.random((n, 3), dtype=np.float32).astype(np.float64)
nrm = rng.standard_normal((n, 3))
nrm /= np.linalg.norm(nrm, axis=1)[:, None]
rot = Rotation.from_euler("xyz", [0.3, -0.2, 0.5]).as_matrix()
t = np.array([0.1, -0.3, 0.2])
perm = rng.permutation(n)
ref, ref_nrm = (scan @ rot.T + t)[perm], (nrm @ rot.T)[perm]

eng = s.default_engine()
t0 = time.perf_counter()
js = DescriptorJob(eng, scan, nrm, radius, min_neighborhood_size=10, do_fpfh=False)
jr = DescriptorJob(eng, ref, ref_nrm, radius, min_neighborhood_size=10, do_fpfh=False)
eng.sync()
t1 = time.perf_counter()
js.step()
jr.step()
eng.sync()
t2 = time.perf_counter()
mj = MatchJob(eng, 352, n, n)
mj.run(js.shot_out, jr.shot_out)
eng.sync()
t3 = time.perf_counter()
# (the same pass once more with the launch timers on: per-kernel times of K8; the wall time above is the untimed run's)
eng.profile_reset()
eng.profile(True)
mj.run(js.shot_out, jr.shot_out)
eng.sync()
eng.profile(False)
k8_report = {k: (v[0], round(v[1], 3)) for k, v in eng.profile_report().items() if k.startswith("k8")}
eng.profile_reset()
eng.profile(True)
rows_s, rows_r = mj.matches()  # cell-sorted numbering of each cloud
scan_idx, ref_idx = js.block_original_indices()[rows_s], jr.block_original_indices()[rows_r]
inv = np.argsort(perm)
correct = float((inv[scan_idx] == ref_idx).mean())
t4 = time.perf_counter()
R.rng = np.random.default_rng(seed=72)
ratio, tf = R.ransac_on_matches(scan_idx, ref_idx, scan, ref, n_draws=n_draws, draw_size=4, distance_threshold=0.01,
                                disable_progress_bar=True)
t5 = time.perf_counter()
print(f"config 4, n = {n}, radius = {radius:.4f}")
print(f"  upload 2 clouds                      {t1 - t0:8.3f} s")
print(f"  SHOT for 2 x {n} points (K1,K2,K4,K5)  {t2 - t1:8.3f} s   ({2 * n / (t2 - t1) / 1e6:.1f} M desc/s)")
print(f"  matching {n} x {n} x 352 (K8)          {t3 - t2:8.3f} s   ({n * n / (t3 - t2) / 1e9:.1f} G pair-dists/s)")
print(f"  matches kept {len(scan_idx)}, equal to the true correspondence: {100 * correct:.1f} %")
print(f"  RANSAC {n_draws} draws x {len(scan_idx)} matches    {t5 - t4:8.3f} s   (host draws + Kabsch, K9 scoring)")
print(f"  inlier ratio {ratio:.4f}, max |R - R_true| = {np.abs(tf.rotation - rot).max():.2e}, "
      f"max |t - t_true| = {np.abs(tf.translation - t).max():.2e}")
eng.profile(False)
print("  K8 kernels (launches, ms) of a second, timed pass:", k8_report)
print("  K9 kernels (launches, ms):", {k: (v[0], round(v[1], 3)) for k, v in eng.profile_report().items() if k.startswith("k9")})
