import sys, time, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
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
