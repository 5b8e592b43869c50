"""First calls of a process and the steady state of BASELINE config 2 as written (100k points, 10k keypoints, r = 0.05:
compute_fpfh_descriptor + ShotMultiprocessor.compute_descriptor_single_scale), with the results of the last two rounds kept
alive as a caller would.  SF_PINNED_THRESHOLD_MB sets the size from which results live in page-locked memory (default 32).
python tools/first_call_config2.py"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import synth_cloud
import shot_fpfh_amd as s
from shot_fpfh_amd.descriptors import ShotMultiprocessor
p, nr, rng = synth_cloud(100_000, 5)
kp_idx = np.sort(rng.choice(p.shape[0], 10_000, replace=False)); kp = p[kp_idx]
s.default_engine().sync()
t0=time.perf_counter(); f = s.compute_fpfh_descriptor(kp_idx, p, nr, 0.05, 5, verbose=False); t1=time.perf_counter()
with ShotMultiprocessor(min_neighborhood_size=10, verbose=False) as sm:
    d = sm.compute_descriptor_single_scale(p, nr, kp, 0.05)
t2=time.perf_counter()
print("first calls: fpfh %.2f ms shot %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3))
keep=[]
ts=[]
for i in range(20):
    t0=time.perf_counter(); f = s.compute_fpfh_descriptor(kp_idx, p, nr, 0.05, 5, verbose=False)
    with ShotMultiprocessor(min_neighborhood_size=10, verbose=False) as sm:
        d = sm.compute_descriptor_single_scale(p, nr, kp, 0.05)
    ts.append(time.perf_counter()-t0); keep.append((f,d)); keep=keep[-2:]
print("steady (results of the last two rounds alive): min %.2f median %.2f ms" % (min(ts)*1e3, np.median(ts)*1e3))
