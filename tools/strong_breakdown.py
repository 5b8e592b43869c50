#!/usr/bin/env python3
"""Where a rank's STRONG-scaling step goes: the 1M-point cloud cut N ways, rank R alone on one GPU (exchange call skipped, as
bench.py --emulate-rank), the step timed plain and then with every launch bracketed.  Prints one JSON object per (N, R).

    python tools/strong_breakdown.py [N R ...]        (default: 1 0  8 0  8 3)
"""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> int:
    from bench import make_cloud
    from shot_fpfh_amd.engine import Engine
    from shot_fpfh_amd.sharding import DescriptorJob

    args = [int(a) for a in sys.argv[1:]] or [1, 0, 8, 0, 8, 3]
    eng = Engine()
    pts, nrm = make_cloud(1_000_000, 3)
    for n, r in zip(args[::2], args[1::2]):
        job = DescriptorJob(eng, pts, nrm, 0.03, n_bins=5, normalize=True, min_neighborhood_size=10, world=n, rank=r,
                            spfh_exchange="neighbor", emulate_peers=n > 1)
        for _ in range(5):
            job.step()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(50):
            job.step()
        eng.sync()
        plain = (time.perf_counter() - t0) / 50
        eng.profile_reset()
        eng.profile(True)
        for _ in range(10):
            job.step()
        eng.sync()
        eng.profile(False)
        rep = eng.profile_report()
        kern = {k: round(v[1] / 10, 4) for k, v in sorted(rep.items()) if v[0] > 0}
        launches = {k: v[0] / 10 for k, v in sorted(rep.items()) if v[0] > 0}
        print(json.dumps({"world": n, "rank": r, "block_keypoints": int(job.m), "ms_per_step": round(plain * 1e3, 4),
                          "sum_of_kernels_ms": round(sum(kern.values()), 4), "launches_per_step": sum(launches.values()),
                          "kernels_ms": kern, "launches": launches}))
        job.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
