#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference (aubin-tchoi/shot-fpfh).

Runs only in the build container, where /root/reference exists:

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py

The reference ships no tests and no golden vectors, so these files are what pins
parity: each stores the inputs, the parameters and the outputs the reference produced for them,
plus the library versions used.  Only data is stored -- no reference source travels.
"""
from __future__ import annotations

import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import scipy  # noqa: E402
import sklearn  # noqa: E402
from sklearn.neighbors import KDTree  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
VERSIONS = np.array([f"numpy {np.__version__}", f"scipy {scipy.__version__}", f"sklearn {sklearn.__version__}"])


def cloud(n, seed, scale=1.0):
    """Synthetic input convention of BASELINE.md 4: float32-grid coordinates, unit normals."""
    rng = np.random.default_rng(seed)
    p = rng.random((n, 3), dtype=np.float32).astype(np.float64) * scale
    nr = rng.standard_normal((n, 3))
    nr /= np.linalg.norm(nr, axis=1)[:, None]
    return p, nr, rng


def surface_cloud(n, seed):
    """Noisy sphere: gives the normals / LRF fixtures a real surface (distinct eigenvalues)."""
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    p = (0.5 + 0.5 * d * (1.0 + 0.01 * rng.standard_normal((n, 1)))).astype(np.float32).astype(np.float64)
    return p, d, rng


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, versions=VERSIONS, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def csr_sorted(lists, dists=None):
    off = np.zeros(len(lists) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in lists])
    order = [np.argsort(x, kind="stable") for x in lists]
    idx = np.concatenate([x[o] for x, o in zip(lists, order)]).astype(np.int32) if len(lists) else np.zeros(0, np.int32)
    if dists is None:
        return off, idx
    return off, idx, np.concatenate([d[o] for d, o in zip(dists, order)])


def main():
    os.makedirs(OUT, exist_ok=True)
    from shot_fpfh.core import grid_subsampling, solver_point_to_point
    from shot_fpfh.descriptors import ShotMultiprocessor, compute_fpfh_descriptor, compute_normals
    from shot_fpfh.descriptors.shot import compute_shot_descriptor, get_local_rf
    from shot_fpfh.matching import basic_matching, match_descriptors, ransac_on_matches, threshold_filter

    # ---- RANSAC first: the module-level default_rng(72) must be fresh (ransac.py:14) -------
    rng = np.random.default_rng(500)
    a = rng.random((700, 3))
    ang = np.array([0.3, -0.2, 0.5])
    from scipy.spatial.transform import Rotation

    R = Rotation.from_euler("xyz", ang).as_matrix()
    t = np.array([0.1, -0.3, 0.2])
    perm = rng.permutation(700)
    b = (a @ R.T + t)[perm]
    scan_idx = rng.choice(700, 500, replace=False).astype(np.int64)
    ref_idx = np.argsort(perm)[scan_idx].astype(np.int64)
    outl = rng.random(500) < 1 / 3
    ref_idx[outl] = rng.integers(0, 700, outl.sum())
    ratio, tr = ransac_on_matches(scan_idx, ref_idx, a, b, n_draws=200, draw_size=4, distance_threshold=0.01,
                                  disable_progress_bar=True)
    draws = np.stack([np.random.default_rng(seed=72).choice(500, 4, replace=False, shuffle=False)])  # first draw only
    g = np.random.default_rng(seed=72)
    draws = np.stack([g.choice(500, 4, replace=False, shuffle=False) for _ in range(200)])
    # per-draw transforms and inlier counts, recomputed with the reference's own solver
    rts = np.zeros((200, 12))
    inl = np.zeros(200, dtype=np.int64)
    for i, d in enumerate(draws):
        tf = solver_point_to_point(a[scan_idx[d]], b[ref_idx[d]])
        rts[i, :9] = tf.rotation.reshape(-1)
        rts[i, 9:] = tf.translation
        inl[i] = (np.linalg.norm(tf[a[scan_idx]] - b[ref_idx], axis=1) <= 0.01).sum()
    save("ransac_500.npz", scan_kp=a, ref_kp=b, scan_idx=scan_idx, ref_idx=ref_idx, n_draws=200, thr=0.01,
         ratio=ratio, rotation=tr.rotation, translation=tr.translation, draws=draws, draw_rt=rts, draw_inliers=inl)

    # ---- neighbour search ------------------------------------------------------------------
    p, nr, rng = cloud(2000, 11)
    r = 0.12
    lists, dists = KDTree(p).query_radius(p, r, return_distance=True)
    off, idx, dist = csr_sorted(lists, dists)
    q_off = rng.random((50, 3)) * 1.4 - 0.2  # queries that are not cloud points, some outside the cube
    lists2 = KDTree(p).query_radius(q_off, r)
    off2, idx2 = csr_sorted(lists2)
    save("nbrs_2k.npz", cloud=p, radius=r, offsets=off, idx=idx, dist=dist, queries=q_off, q_offsets=off2, q_idx=idx2)

    # ---- normals ---------------------------------------------------------------------------
    ps, ds, rng = surface_cloud(2000, 12)
    qn = ps[::4]
    pre = ds[::4]
    save("normals_2k.npz", cloud=ps, queries=qn, pre=pre, radius=0.12, k=30,
         n_radius=compute_normals(qn, ps, radius=0.12),
         n_radius_pre=compute_normals(qn, ps, radius=0.12, pre_computed_normals=pre),
         n_knn=compute_normals(qn, ps, k=30),
         n_knn_pre=compute_normals(qn, ps, k=30, pre_computed_normals=pre))

    # ---- SHOT ------------------------------------------------------------------------------
    p, nr, rng = cloud(6000, 13)
    r = 0.12
    kp_idx = np.sort(rng.choice(6000, 150, replace=False))
    kp = np.vstack([p[kp_idx], [[1.5, 1.5, 1.5]], [[0.5, 0.5, 1.1]]])  # + empty and sparse off-cloud keypoints
    lists = KDTree(p).query_radius(kp, r)
    lrfs = np.array([get_local_rf((k_, p[l_], r)) for k_, l_ in zip(kp, lists)])
    out = dict(cloud=p, normals=nr, keypoints=kp, radius=r, lrf=lrfs)
    for norm in (True, False):
        for mn in (10, 100):
            with ShotMultiprocessor(normalize=norm, min_neighborhood_size=mn, n_procs=2, disable_progress_bar=True,
                                    verbose=False) as sm:
                out[f"single_n{int(norm)}_m{mn}"] = sm.compute_descriptor_single_scale(p, nr, kp, r)
    vox = r / 10
    support = grid_subsampling(p, vox)
    with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, n_procs=2, disable_progress_bar=True,
                            verbose=False) as sm:
        out["voxel"] = vox
        out["support"] = support
        out["single_sub"] = sm.compute_descriptor_single_scale(p, nr, kp, r, subsampling_voxel_size=vox)
        out["bi_scale_sub"] = sm.compute_descriptor_bi_scale(p, nr, kp[:60], local_rf_radius=0.08, shot_radius=r,
                                                             subsampling_voxel_size=vox)
        out["multi_shared"] = sm.compute_descriptor_multiscale(p, nr, kp[:60], radii=[0.08, 0.12], weights=[1.0, 0.5])
    with ShotMultiprocessor(normalize=True, share_local_rfs=False, min_neighborhood_size=10, n_procs=2,
                            disable_progress_bar=True, verbose=False) as sm:
        out["multi_unshared"] = sm.compute_descriptor_multiscale(p, nr, kp[:60], radii=[0.08, 0.12],
                                                                 voxel_sizes=[0.008, 0.012])
        out["support_008"] = grid_subsampling(p, 0.008)
    out["serial"] = compute_shot_descriptor(kp[:60], p, nr, r, min_neighborhood_size=10)
    save("shot_150.npz", **out)

    # duplicates (rho == 0) and tiny neighbourhoods
    pd_, nd, rng = cloud(400, 14, scale=0.4)
    pd_[50:60] = pd_[40:50]  # exact duplicates
    kpd = pd_[35:65]
    listsd = KDTree(pd_).query_radius(kpd, 0.1)
    lrfd = np.array([get_local_rf((k_, pd_[l_], 0.1)) for k_, l_ in zip(kpd, listsd)])
    with ShotMultiprocessor(normalize=True, min_neighborhood_size=5, n_procs=2, disable_progress_bar=True,
                            verbose=False) as sm:
        sd = sm.compute_descriptor_single_scale(pd_, nd, kpd, 0.1)
    fd = compute_fpfh_descriptor(np.arange(35, 65), pd_, nd, 0.1, 5, verbose=False)
    save("edge_dups.npz", cloud=pd_, normals=nd, keypoints=kpd, kp_idx=np.arange(35, 65), radius=0.1, lrf=lrfd,
         shot_m5=sd, fpfh5=fd)

    # ---- FPFH ------------------------------------------------------------------------------
    p, nr, rng = cloud(3000, 15)
    kpi = np.sort(rng.choice(3000, 200, replace=False)).astype(np.int64)
    save("fpfh_200.npz", cloud=p, normals=nr, kp_idx=kpi, radius=0.12,
         fpfh5=compute_fpfh_descriptor(kpi, p, nr, 0.12, 5, verbose=False),
         fpfh4=compute_fpfh_descriptor(kpi, p, nr, 0.12, 4, verbose=False))
    # a surface-like cloud with coherent normals exercises more bins
    ps, ds, rng = surface_cloud(3000, 16)
    kpi = np.sort(rng.choice(3000, 100, replace=False)).astype(np.int64)
    save("fpfh_surface.npz", cloud=ps, normals=ds, kp_idx=kpi, radius=0.15,
         fpfh5=compute_fpfh_descriptor(kpi, ps, ds, 0.15, 5, verbose=False),
         fpfh3=compute_fpfh_descriptor(kpi, ps, ds, 0.15, 3, verbose=False))

    # ---- matching --------------------------------------------------------------------------
    rng = np.random.default_rng(17)
    sa = rng.random((300, 352)) * (rng.random((300, 352)) < 0.2)
    sb = sa[rng.permutation(300)][:280] + 0.01 * rng.standard_normal((280, 352)) * (rng.random((280, 352)) < 0.2)
    sa[[3, 77, 150]] = 0.0
    sb[[5, 200]] = 0.0
    m = {}
    m["basic_s"], m["basic_r"] = basic_matching(sa, sb)
    m["md_s"], m["md_r"] = match_descriptors(sa, sb, verbose=False)
    m["thr_s"], m["thr_r"] = match_descriptors(sa, sb, threshold_filter, verbose=False, threshold_multiplier=10)
    m["rec_s"], m["rec_r"] = match_descriptors(sa, sb, filter_nonreciprocal=True, verbose=False, n_min_matches=100)
    m["recbig_s"], m["recbig_r"] = match_descriptors(sa, sb, filter_nonreciprocal=True, verbose=False,
                                                     n_min_matches=10**6)
    save("match_300.npz", scan=sa, ref=sb, **m)

    # ---- 3-D (multi-scale, "minimum over scales") matching branch: matching.py:77-136 -------
    rng = np.random.default_rng(19)
    s3 = rng.random((2, 200, 352)) * (rng.random((2, 200, 352)) < 0.2)
    r3 = s3[:, rng.permutation(200)][:, :180] + 0.01 * rng.standard_normal((2, 180, 352)) * (rng.random((2, 180, 352)) < 0.2)
    s3[0, [3, 50]] = 0.0   # empty at one scale only
    s3[:, 77] = 0.0        # empty at every scale -> distance stays at max_val and is dropped
    r3[1, [5, 100]] = 0.0
    m3 = {}
    m3["md_s"], m3["md_r"] = match_descriptors(s3, r3, verbose=False)
    m3["thr_s"], m3["thr_r"] = match_descriptors(s3, r3, threshold_filter, verbose=False, threshold_multiplier=3)
    m3["rec_s"], m3["rec_r"] = match_descriptors(s3, r3, filter_nonreciprocal=True, verbose=False, n_min_matches=10**6)
    save("match3d_200.npz", scan=s3, ref=r3, **m3)

    # ---- grid subsampling ------------------------------------------------------------------
    p, nr, rng = cloud(20000, 18)
    save("grid_sub_20k.npz", seed=18, n=20000, voxel=0.05, idx=grid_subsampling(p, 0.05))


if __name__ == "__main__":
    main()
