"""Step time of the FPFH + SHOT pass on clouds of equal mean neighbourhood size but different density structure:
the BASELINE uniform cloud, a surface (tests/conftest.py::config1_cloud: a noisy sphere, the stand-in for the Stanford
scans the reference is run on, scripts/parse_args.py:6-22) and a clustered cloud (tests/conftest.py::family("clustered")).
Prints one JSON object; bench.py's `surface_cloud` / `clustered_cloud` keys use the same functions.

    python tools/bench_density.py [--n 1000000] [--kbar 110] [--steps 10] [--clouds uniform,surface,clustered]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make(name: str, n: int, seed: int = 3):
    from conftest import config1_cloud, family, synth_cloud

    if name == "uniform":
        p, nr, _ = synth_cloud(n, seed)
        return p, nr
    if name == "surface":
        return config1_cloud(n, seed)
    if name == "clustered":
        p, nr, _, _ = family("clustered", n, np.random.default_rng(seed))
        return p, nr
    raise KeyError(name)


def radius_for(eng, points, kbar: float, sample: int = 20000, seed: int = 9) -> tuple[float, float]:
    """Radius at which the MEAN neighbourhood size (self included) over the cloud's points is `kbar`: bisection on a
    sample of the points as queries."""
    rng = np.random.default_rng(seed)
    q = points[rng.choice(points.shape[0], min(sample, points.shape[0]), replace=False)]
    cloud = eng.cloud(points)
    try:
        lo, hi = 1e-4, 0.5
        mean = 0.0
        for _ in range(40):
            r = (lo * hi) ** 0.5
            nb = cloud.radius_search(q, r)
            mean = nb.total / nb.m
            nb.free()
            if mean > kbar:
                hi = r
            else:
                lo = r
            if hi / lo < 1.002:
                break
        return (lo * hi) ** 0.5, mean
    finally:
        cloud.free()


def run(eng, name: str, n: int, kbar: float, steps: int, warmup: int = 3, parity_rows: int = 0) -> dict:
    from shot_fpfh_amd.sharding import DescriptorJob

    p, nr = make(name, n)
    radius, _ = radius_for(eng, p, kbar)
    job = DescriptorJob(eng, p, nr, radius, n_bins=5, normalize=True, min_neighborhood_size=10)
    try:
        for _ in range(warmup):
            job.step()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            job.step()
        eng.sync()
        dt = (time.perf_counter() - t0) / steps
        eng.profile_reset()
        eng.profile(True)
        for _ in range(min(steps, 5)):
            job.step()
        eng.sync()
        eng.profile(False)
        rep = eng.profile_report()
        nprof = min(steps, 5)
        nb = job.cloud.radius_search_self(radius)
        cnt = nb.counts()
        nb.free()
        out = {
            "cloud": name, "points": n, "radius": radius, "mean_k": float(cnt.mean()), "max_k": int(cnt.max()),
            "share_over_255": float((cnt > 255).mean()), "ms_per_step": dt * 1e3, "desc_per_s": 2 * n / dt,
            "kernels_ms_per_step": {k: round(v[1] / nprof, 4) for k, v in sorted(rep.items()) if v[1] > 0},
            "launches_per_step": {k: v[0] / nprof for k, v in sorted(rep.items())},
        }
        if parity_rows:
            import bench

            out["parity"] = bench.parity_sample(job, p, nr, radius, parity_rows)
        return out
    finally:
        job.close()


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--kbar", type=float, default=110.0)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--clouds", default="uniform,surface,clustered")
    ap.add_argument("--parity-rows", type=int, default=0)
    a = ap.parse_args()
    from shot_fpfh_amd.engine import Engine

    eng = Engine()
    res = [run(eng, c, a.n, a.kbar, a.steps, parity_rows=a.parity_rows) for c in a.clouds.split(",")]
    print(json.dumps({"build": eng.lib.sf_version().decode(), "results": res}, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
