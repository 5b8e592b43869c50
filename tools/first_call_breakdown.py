"""Where the first call of a process goes (BASELINE config 2's size): wall time of every step of compute_fpfh_descriptor's
sequence, first and second time round.  python tools/first_call_breakdown.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
t00 = time.perf_counter()
import numpy as np
from conftest import synth_cloud
t_np = time.perf_counter()
import shot_fpfh_amd as s
from shot_fpfh_amd.engine import Cloud, Spfh
t_imp = time.perf_counter()
p, nr, rng = synth_cloud(100_000, 5)
kp = np.sort(rng.choice(p.shape[0], 10_000, replace=False)).astype(np.int64)
t0 = time.perf_counter()
eng = s.default_engine()
eng.sync()
t1 = time.perf_counter()
print(f"import numpy+conftest {1e3 * (t_np - t00):.0f} ms, import shot_fpfh_amd {1e3 * (t_imp - t_np):.0f} ms, engine (library load, sf_create, first sync) {1e3 * (t1 - t0):.0f} ms")
for rnd in (1, 2, 3):
    marks = [("start", time.perf_counter())]
    cloud = Cloud(eng, p, nr); eng.sync(); marks.append(("upload", time.perf_counter()))
    nb = cloud.radius_search_self(0.05); eng.sync(); marks.append(("grid + search", time.perf_counter()))
    sp = Spfh(cloud, 5, nb.max_count, 0.05); eng.sync(); marks.append(("table", time.perf_counter()))
    sp.compute(nb); eng.sync(); marks.append(("K6", time.perf_counter()))
    out = sp.fpfh(nb, kp); eng.sync(); marks.append(("K7 + copy", time.perf_counter()))
    sp.free(); nb.free(); cloud.free(); marks.append(("free", time.perf_counter()))
    print(f"round {rnd}: " + ", ".join(f"{n} {1e3 * (b - a):.2f}" for (_, a), (n, b) in zip(marks, marks[1:])) + f"  = {1e3 * (marks[-1][1] - marks[0][1]):.2f} ms")
