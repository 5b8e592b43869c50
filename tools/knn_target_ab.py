import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, shot_fpfh_amd as s
from conftest import config1_cloud, synth_cloud
eng = s.default_engine()
for name in ("uniform", "surface"):
    p = synth_cloud(1000000, 3)[0] if name == "uniform" else config1_cloud(1000000, 3)[0]
    cloud = eng.cloud(p); out = eng.empty((p.shape[0], 3))
    for t in ("3.5", "2.5", "2.0", "1.75", "1.5"):
        os.environ["SF_KNN_TARGET"] = t
        best, rep = 1e9, None
        for i in range(4):
            eng.sync(); eng.profile_reset(); eng.profile(True); t0 = time.perf_counter()
            nb = cloud.knn_search(p, 30); nb.normals(out=out); nb.free(); eng.sync(); dt = time.perf_counter() - t0; eng.profile(False)
            if i and dt < best:
                best, rep = dt, eng.profile_report()
        dev = sum(v[1] for v in rep.values())
        print(name, t, "wall %.2f ms  device %.2f ms  k2_knn %.3f (+crowded %.3f)  k1 %.3f  launches of k2_knn %d" % (
            1e3 * best, dev, rep.get("k2_knn", (0, 0))[1], rep.get("k2_knn_crowded", (0, 0))[1],
            sum(v[1] for k, v in rep.items() if k.startswith("k1_")), rep.get("k2_knn", (0, 0))[0]))
    cloud.free()
