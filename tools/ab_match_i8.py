#!/usr/bin/env python3
"""K8 A/B on one GPU: the integer pre-filter (SF_MATCH_I8=1) against the FP16 pre-filter (SF_MATCH_I8=0) on the same resident
descriptors -- identical index / distance vectors required, times and flagged rows printed.
Usage: ab_match_i8.py [n_points of the two SHOT clouds, default 200000] [m of the synthetic pair, default 65536]"""
import os
import sys
import time

import numpy as np
from scipy.spatial.transform import Rotation

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shot_fpfh_amd as s
from shot_fpfh_amd.sharding import DescriptorJob

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
eng = s.default_engine()


def run(da, db, m1, label):
    idx, dist = eng.empty((m1,), np.int64), eng.empty((m1,), np.float64)
    out = {}
    for mode in ("0", "1"):
        os.environ["SF_MATCH_I8"] = mode[0]
        eng.match_argmin_device(da, db, idx, dist)
        eng.sync()
        eng.profile_reset()
        eng.profile(True)
        t0 = time.perf_counter()
        eng.match_argmin_device(da, db, idx, dist)
        eng.sync()
        t = time.perf_counter() - t0
        eng.profile(False)
        rep = {k: (v[0], round(v[1], 3)) for k, v in eng.profile_report().items() if k.startswith("k8")}
        out[mode] = (idx.to_host().copy(), dist.to_host().copy())
        print(f"{label} SF_MATCH_I8={mode}: {t * 1e3:.2f} ms  {rep}", flush=True)
    for mode in ("1",):
        same_i, same_d = np.array_equal(out["0"][0], out[mode][0]), np.array_equal(out["0"][1], out[mode][1])
        print(f"{label} {mode}: identical idx {same_i}, identical dist {same_d}", flush=True)
        if not same_i:
            bad = np.flatnonzero(out["0"][0] != out[mode][0])
            print("  first differing rows", bad[:10], out["0"][0][bad[:5]], out[mode][0][bad[:5]], out["0"][1][bad[:5]], out[mode][1][bad[:5]])


rng = np.random.default_rng(0)
a = rng.random((m, 352))
a /= np.linalg.norm(a, axis=1)[:, None]
b = a[rng.permutation(m)] + 1e-3 * rng.standard_normal((m, 352))
da, db = eng.empty((m, 352)).from_host(a), eng.empty((m, 352)).from_host(b)
run(da, db, m, f"synthetic {m} x {m} x 352")
da.free(); db.free()

radius = 0.03 * (1_000_000 / n) ** (1 / 3)
rng = np.random.default_rng(4)
scan = rng.random((n, 3), dtype=np.float32).astype(np.float64)
nrm = rng.standard_normal((n, 3))
nrm /= np.linalg.norm(nrm, axis=1)[:, None]
rot = Rotation.from_euler("xyz", [0.3, -0.2, 0.5]).as_matrix()
t = np.array([0.1, -0.3, 0.2])
perm = rng.permutation(n)
ref, ref_nrm = (scan @ rot.T + t)[perm], (nrm @ rot.T)[perm]
js = DescriptorJob(eng, scan, nrm, radius, min_neighborhood_size=10, do_fpfh=False)
jr = DescriptorJob(eng, ref, ref_nrm, radius, min_neighborhood_size=10, do_fpfh=False)
js.step(); jr.step(); eng.sync()
run(js.shot_out, jr.shot_out, n, f"SHOT {n} x {n} x 352")
