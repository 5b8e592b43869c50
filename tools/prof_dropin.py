import sys, cProfile, pstats, io
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from conftest import config1_cloud
from shot_fpfh_amd.engine import Engine
from shot_fpfh_amd.descriptors import ShotMultiprocessor
eng = Engine()
ps, ns = config1_cloud(1_000_000, 3)
rng = np.random.default_rng(17)
kps = ps[np.sort(rng.choice(ps.shape[0], 100_000, replace=False))]
kw = dict(normalize=True, min_neighborhood_size=10, verbose=False, engine=eng)
def single():
    with ShotMultiprocessor(**kw) as sm:
        return sm.compute_descriptor_single_scale(ps, ns, kps, 0.05, subsampling_voxel_size=0.005)
def multi():
    with ShotMultiprocessor(**kw) as sm:
        return sm.compute_descriptor_multiscale(ps, ns, kps, [0.05, 0.075], voxel_sizes=[0.005, 0.0075])
for f in (single, multi):
    f(); f()
    pr = cProfile.Profile(); pr.enable(); f(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(14); print(f.__name__); print('\n'.join(s.getvalue().splitlines()[:34]))
