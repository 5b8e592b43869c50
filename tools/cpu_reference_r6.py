#!/usr/bin/env python3
"""The REFERENCE itself (aubin-tchoi/shot-fpfh, imported from /root/reference -- build container only), timed on the build
container's cores:

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tools/cpu_reference_r6.py

* BASELINE config 2 exactly as written: synthetic 100 000-point uniform cloud, 10 000 keypoints, radius 0.05,
  compute_fpfh_descriptor (fpfh.py:16-117: SPFH of ALL 100 000 points, single process) + ShotMultiprocessor(n_procs = 8,
  min_neighborhood_size = 10).compute_descriptor_single_scale (shot_parallelization.py:135-183).
* a 200 000-point slice of BASELINE config 3's workload -- every point a keypoint, radius 0.03 * 5^(1/3) (config 3's neighbours
  per ball), FPFH (5 bins) + SHOT -- the largest size the reference's whole-cloud Python loops finish in minutes here.
* the calibration of oracle/numpy_shaped.py (what bench.py times on the GPU box, where the reference cannot go) against the
  reference on 12 000 points: same protocol as tools/calibrate_cpu_baseline.py, refreshed.
Writes profiles/r06_cpu_reference.json (versions, core count, seconds, descriptors per second)."""
import json
import os
import platform
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import scipy  # noqa: E402
import sklearn  # noqa: E402


def cloud(n, seed):
    rng = np.random.default_rng(seed)
    p = rng.random((n, 3), dtype=np.float32).astype(np.float64)
    nr = rng.standard_normal((n, 3))
    nr /= np.linalg.norm(nr, axis=1)[:, None]
    return p, nr, rng


def main():
    from oracle import numpy_shaped as NS
    from shot_fpfh.descriptors import ShotMultiprocessor, compute_fpfh_descriptor

    n_procs = min(8, os.cpu_count() or 1)
    out = {"what": "the reference's own functions, imported from /root/reference, on the build container",
           "host": {"cores": os.cpu_count(), "machine": platform.machine(), "python": platform.python_version()},
           "versions": {"numpy": np.__version__, "scipy": scipy.__version__, "sklearn": sklearn.__version__}, "n_procs": n_procs}
    which = sys.argv[1:] or ["config2", "config3_slice", "calibration"]

    if "config2" in which:
        p, nr, rng = cloud(100_000, 2)
        kp = np.sort(rng.choice(100_000, 10_000, replace=False))
        t = time.perf_counter(); f = compute_fpfh_descriptor(kp, p, nr, 0.05, 5, verbose=False); t_f = time.perf_counter() - t
        t = time.perf_counter()
        with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, n_procs=n_procs, disable_progress_bar=True, verbose=False) as sm:
            d = sm.compute_descriptor_single_scale(p, nr, p[kp], 0.05)
        t_s = time.perf_counter() - t
        out["config2_as_written"] = {"points": 100_000, "keypoints": 10_000, "radius": 0.05, "fpfh_s": t_f, "shot_s": t_s,
                                     "descriptors": 20_000, "descriptors_per_s": 20_000 / (t_f + t_s),
                                     "nonzero_shot_rows": int(np.any(d, axis=1).sum()), "fpfh_row_sum_mean": float(f.sum(axis=1).mean())}
        print(json.dumps(out["config2_as_written"]), flush=True)

    if "config3_slice" in which:
        n = 200_000
        r = 0.03 * 5.0 ** (1.0 / 3.0)
        p, nr, rng = cloud(n, 3)
        kp = np.arange(n)
        t = time.perf_counter(); f = compute_fpfh_descriptor(kp, p, nr, r, 5, verbose=False); t_f = time.perf_counter() - t
        t = time.perf_counter()
        with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, n_procs=n_procs, disable_progress_bar=True, verbose=False) as sm:
            d = sm.compute_descriptor_single_scale(p, nr, p, r)
        t_s = time.perf_counter() - t
        out["config3_200k_slice"] = {"points": n, "keypoints": n, "radius": r, "fpfh_s": t_f, "shot_s": t_s, "descriptors": 2 * n,
                                     "descriptors_per_s": 2 * n / (t_f + t_s),
                                     "implied_seconds_for_config3_full": 2e6 / (2 * n / (t_f + t_s)),
                                     "nonzero_shot_rows": int(np.any(d, axis=1).sum())}
        print(json.dumps(out["config3_200k_slice"]), flush=True)
        del f, d

    if "calibration" in which:
        ns = 12000
        r = 0.03 * (1_000_000 / ns) ** (1.0 / 3.0)
        p, nr = NS._cloud(ns, 33)
        kp = np.arange(ns)
        c = {"points": ns, "radius": r, "n_procs": n_procs, "host_cores": os.cpu_count()}
        t = time.perf_counter(); f_ref = compute_fpfh_descriptor(kp, p, nr, r, 5, verbose=False); c["ref_fpfh_s"] = time.perf_counter() - t
        t = time.perf_counter(); f_ns = NS.fpfh_numpy_shaped(kp, p, nr, r, 5); c["shaped_fpfh_s"] = time.perf_counter() - t
        t = time.perf_counter()
        with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, n_procs=n_procs, disable_progress_bar=True, verbose=False) as sm:
            d_ref = sm.compute_descriptor_single_scale(p, nr, p, r)
        c["ref_shot_s"] = time.perf_counter() - t
        t = time.perf_counter(); d_ns = NS.shot_numpy_shaped(p, nr, p, r, True, 10, n_procs); c["shaped_shot_s"] = time.perf_counter() - t
        c["fpfh_max_abs_diff"] = float(np.abs(f_ref - f_ns).max())
        c["shot_max_abs_diff"] = float(np.abs(d_ref - d_ns).max())
        c["ratio_total"] = (c["shaped_fpfh_s"] + c["shaped_shot_s"]) / (c["ref_fpfh_s"] + c["ref_shot_s"])
        c["ref_desc_per_s"] = 2 * ns / (c["ref_fpfh_s"] + c["ref_shot_s"])
        c["shaped_desc_per_s"] = 2 * ns / (c["shaped_fpfh_s"] + c["shaped_shot_s"])
        out["calibration_of_numpy_shaped"] = c
        print(json.dumps(c), flush=True)

    path = os.path.join(ROOT, "profiles", "r06_cpu_reference.json")
    prev = {}
    if os.path.exists(path):
        prev = json.load(open(path))
    prev.update(out)
    json.dump(prev, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
