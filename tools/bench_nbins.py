import sys, json, time
sys.path.insert(0, '/root/repo')
import numpy as np
from bench import make_cloud
from shot_fpfh_amd.engine import Engine
from shot_fpfh_amd.sharding import DescriptorJob
eng = Engine()
pts, nrm = make_cloud(1_000_000, 3)
for nb in (3, 4, 5, 6, 7, 8, 9, 11, 10):
    job = DescriptorJob(eng, pts, nrm, 0.03, n_bins=nb, normalize=True, min_neighborhood_size=10, do_shot=False)
    for _ in range(3): job.step()
    eng.sync(); eng.profile_reset(); eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(5): job.step()
    eng.sync(); dt = (time.perf_counter() - t0) / 5
    eng.profile(False)
    rep = eng.profile_report()
    print(nb, round(dt * 1e3, 3), {k: round(v[1] / 5, 3) for k, v in sorted(rep.items()) if v[0] and k[:2] in ('k6', 'k7')})
    job.close()
