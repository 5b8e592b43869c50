"""Where the host-to-host time of the two headline drop-in calls goes (1M uniform points, all keypoints): cProfile of one warm
call each, cumulative.  python tools/prof_dropin_main.py"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bench import make_cloud
from shot_fpfh_amd.descriptors import ShotMultiprocessor, compute_fpfh_descriptor
from shot_fpfh_amd.engine import Engine

eng = Engine()
p, nr = make_cloud(1_000_000, 3)
kp = np.arange(p.shape[0])


def fpfh():
    return compute_fpfh_descriptor(kp, p, nr, 0.03, 5, verbose=False, engine=eng)


def shot():
    with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, verbose=False, engine=eng) as sm:
        return sm.compute_descriptor_single_scale(p, nr, p, 0.03)


for f in (fpfh, shot):
    f(); f()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); a = f(); ts.append(time.perf_counter() - t0); del a
    pr = cProfile.Profile(); pr.enable(); a = f(); pr.disable(); del a
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
    print(f.__name__, "best of 3: %.2f ms" % (1e3 * min(ts)))
    print("\n".join(s.getvalue().splitlines()[:40]))
