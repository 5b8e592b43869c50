#!/usr/bin/env python3
"""Max deviation of the HIP path from the CPU oracle (run on the GPU box): SHOT and FPFH on a seeded cloud.
Prints one line per descriptor; the parity tolerance (BASELINE.json) is 1e-5, the kernels sit near 1e-15."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shot_fpfh_amd as s  # noqa: E402
from oracle import oracle as O  # noqa: E402

rng = np.random.default_rng(5)
n, m, r = 60000, 4000, 0.06
p = rng.random((n, 3), dtype=np.float32).astype(np.float64)
nrm = rng.standard_normal((n, 3))
nrm /= np.linalg.norm(nrm, axis=1)[:, None]
kp = np.sort(rng.choice(n, m, replace=False))
with s.ShotMultiprocessor(normalize=True, min_neighborhood_size=10, verbose=False, disable_progress_bar=True) as sm:
    shot = sm.compute_descriptor_single_scale(point_cloud=p, keypoints=p[kp], normals=nrm, radius=r)
shot_o = O.shot_single_scale(p, nrm, p[kp], r, normalize=True, min_neighborhood_size=10)
d = np.abs(shot - shot_o)
print(f"SHOT  {m} x 352: max |gpu - oracle| = {d.max():.3e}, rows with any |d| > 1e-9: {(d.max(axis=1) > 1e-9).sum()}")
fpfh = s.compute_fpfh_descriptor(kp, p, nrm, radius=r, n_bins=5, verbose=False)
fpfh_o = O.compute_fpfh_descriptor(kp, p, nrm, r, 5)
d = np.abs(fpfh - fpfh_o)
print(f"FPFH  {m} x 125: max |gpu - oracle| = {d.max():.3e} (max value {fpfh_o.max():.3f}), "
      f"rows with any |d| > 1e-9: {(d.max(axis=1) > 1e-9).sum()}")
