"""The bench step (FPFH + SHOT, 1M uniform points) at other radii than BASELINE's: a quick look for cliffs between the forms
(1 / 2 / 3 / 4 chunks, the long-list launches).  python tools/bench_radii.py [radius ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_cloud
from shot_fpfh_amd.engine import Engine
from shot_fpfh_amd.sharding import DescriptorJob

eng = Engine()
pts, nrm = make_cloud(1_000_000, 3)
for r in ([float(a) for a in sys.argv[1:]] or (0.015, 0.02, 0.025, 0.03, 0.035, 0.04, 0.045, 0.05)):
    job = DescriptorJob(eng, pts, nrm, r, n_bins=5, normalize=True, min_neighborhood_size=10)
    for _ in range(3):
        job.step()
    eng.sync(); eng.profile_reset(); eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(5):
        job.step()
    eng.sync(); dt = (time.perf_counter() - t0) / 5
    eng.profile(False)
    rep = eng.profile_report()
    kbar = job.last_pairs / 1e6
    print(f"r={r} kbar={kbar:.0f} step {dt * 1e3:.3f} ms  {dt * 1e9 / job.last_pairs:.3f} ns/pair ", {k: round(v[1] / 5, 3) for k, v in sorted(rep.items()) if v[0] and k[:2] in ('k2', 'k5', 'k6', 'k7') and v[1] / 5 > 0.02})
    job.close()
