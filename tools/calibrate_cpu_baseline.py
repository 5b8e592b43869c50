#!/usr/bin/env python3
"""Calibrate oracle/numpy_shaped.py (the reference-SHAPED CPU baseline bench.py times on the GPU box) against the
reference itself, which can only be imported in the build container:

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tools/calibrate_cpu_baseline.py [points] [n_procs]

Both run FPFH (single process, as the reference does) and single-scale SHOT (Pool of n_procs) on the same cloud, all
points keypoints, at the neighbours-per-ball of BASELINE config 3.  Writes profiles/r02_cpu_calibration.json with both
timings, the ratio restatement / reference (SURVEY 8d asks for +-20 %) and the max abs difference of the outputs."""
import json
import os
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    from oracle import numpy_shaped as NS
    from shot_fpfh.descriptors import ShotMultiprocessor, compute_fpfh_descriptor

    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
    n_procs = int(sys.argv[2]) if len(sys.argv) > 2 else min(8, os.cpu_count() or 1)
    r = 0.03 * (1_000_000 / ns) ** (1.0 / 3.0)
    p, nr = NS._cloud(ns, 33)
    kp = np.arange(ns)
    out = {"points": ns, "radius": r, "n_procs": n_procs, "host_cores": os.cpu_count()}
    t = time.perf_counter(); f_ref = compute_fpfh_descriptor(kp, p, nr, r, 5, verbose=False); out["ref_fpfh_s"] = time.perf_counter() - t
    t = time.perf_counter(); f_ns = NS.fpfh_numpy_shaped(kp, p, nr, r, 5); out["shaped_fpfh_s"] = time.perf_counter() - t
    t = time.perf_counter()
    with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, n_procs=n_procs, disable_progress_bar=True, verbose=False) as sm:
        d_ref = sm.compute_descriptor_single_scale(p, nr, p, r)
    out["ref_shot_s"] = time.perf_counter() - t
    t = time.perf_counter(); d_ns = NS.shot_numpy_shaped(p, nr, p, r, True, 10, n_procs); out["shaped_shot_s"] = time.perf_counter() - t
    out["fpfh_max_abs_diff"] = float(np.abs(f_ref - f_ns).max())
    out["shot_max_abs_diff"] = float(np.abs(d_ref - d_ns).max())
    out["ratio_fpfh"] = out["shaped_fpfh_s"] / out["ref_fpfh_s"]
    out["ratio_shot"] = out["shaped_shot_s"] / out["ref_shot_s"]
    out["ratio_total"] = (out["shaped_fpfh_s"] + out["shaped_shot_s"]) / (out["ref_fpfh_s"] + out["ref_shot_s"])
    out["ref_desc_per_s"] = 2 * ns / (out["ref_fpfh_s"] + out["ref_shot_s"])
    out["shaped_desc_per_s"] = 2 * ns / (out["shaped_fpfh_s"] + out["shaped_shot_s"])
    path = os.path.join(ROOT, "profiles", "r02_cpu_calibration.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
