"""Where the host time of BASELINE config 2 as written goes (100k points, 10k keypoints, r = 0.05: compute_fpfh_descriptor +
ShotMultiprocessor.compute_descriptor_single_scale): wall time per call over 30 calls, and one call of each under cProfile.
python tools/prof_config2.py"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import synth_cloud
import shot_fpfh_amd as s
from shot_fpfh_amd.descriptors import ShotMultiprocessor

p, nr, rng = synth_cloud(100_000, 5)
kp_idx = np.sort(rng.choice(p.shape[0], 10_000, replace=False))
kp = p[kp_idx]

def fpfh():
    return s.compute_fpfh_descriptor(kp_idx, p, nr, 0.05, 5, verbose=False)

def shot():
    with ShotMultiprocessor(min_neighborhood_size=10, verbose=False) as sm:
        return sm.compute_descriptor_single_scale(p, nr, kp, 0.05)

for f in (fpfh, shot):
    for _ in range(3):
        f()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print(f"{f.__name__}: min {ts.min():.3f} median {np.median(ts):.3f} max {ts.max():.3f} ms")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10):
        f()
    pr.disable()
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(22)
    print("\n".join(st.getvalue().splitlines()[:40]))
