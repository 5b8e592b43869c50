"""Reference-SHAPED CPU baseline -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (never imported by shot_fpfh_amd/).

oracle/shot_fpfh_oracle.c is a scalar C port: it computes what the reference computes, but ~10x faster than the
reference can, so timing it understates what the GPU path replaces.  SURVEY 8(d) asks for the CPU path "timed beside"
the GPU one to have the reference's own cost structure: an interpreted per-point loop in which every step is a small
NumPy call (sklearn KDTree search; np.cross / einsum / arctan2 / np.histogramdd per cloud point for SPFH, fpfh.py:38-90;
a masked gather-divide-sum per keypoint for FPFH, fpfh.py:101-116; eigh + sign votes per keypoint for the frame,
shot.py:16-48; argsort + ten fancy-index `D[idx] += v` statements per keypoint for SHOT, shot.py:175-306, spread over a
multiprocessing.Pool with the neighbourhoods pickled to the workers, shot_parallelization.py:46-133).

This module restates those loops with that structure -- same third-party calls, same number of NumPy dispatches per
point, same process boundary -- so its throughput tracks the reference's (calibrated in the build container, where
the reference can be imported: tools/calibrate_cpu_baseline.py; ratio recorded in profiles/r02_cpu_calibration.json,
required to be within +-20 %).  Its VALUES are checked against the reference's golden vectors in
tests/test_oracle_golden.py, so it is also a second, independent oracle.

It needs numpy + scikit-learn only (both in the image) and runs in a process that never touches the GPU
(bench.py starts it as a child BEFORE initialising HIP, because it forks a Pool).
"""
from __future__ import annotations

import json
import multiprocessing as mp
import sys
import time

import numpy as np

PI = np.pi


# ---------------------------------------------------------------------------------------------------------------
# FPFH (fpfh.py:16-117, decorrelated=False): single process, single thread -- as the reference runs it
# ---------------------------------------------------------------------------------------------------------------
def fpfh_numpy_shaped(keypoints_indices, cloud, normals, radius, n_bins):
    from sklearn.neighbors import KDTree

    lists, dists = KDTree(cloud).query_radius(cloud, radius, return_distance=True)  # fpfh.py:26-30: ALL points
    n = cloud.shape[0]
    table = np.zeros((n, n_bins, n_bins, n_bins))
    box = [(-1, 1), (-1, 1), (-PI / 2, PI / 2)]
    for i in range(n):  # fpfh.py:38-90
        nbr = lists[i]
        if nbr.shape[0] == 0:
            continue
        offs = cloud[nbr] - cloud[i]
        nn = normals[nbr]
        d = np.linalg.norm(offs, axis=1)
        far = d > 0
        u = normals[i]
        v = np.cross(offs[far], u)
        w = np.cross(u, v)
        alpha = np.einsum("ij,ij->i", v, nn[far])
        phi = offs[far].dot(u) / d[far]
        theta = np.arctan2(np.einsum("ij,ij->i", nn[far], w), nn[far].dot(u))
        table[i] = np.histogramdd(np.vstack((alpha, phi, theta)).T, bins=n_bins, range=box)[0] / nbr.shape[0]
    table = table.reshape(n, -1)
    out = np.zeros((len(keypoints_indices), n_bins**3))
    for row, kp in enumerate(keypoints_indices):  # fpfh.py:101-116
        nbr, d = lists[kp], dists[kp]
        with np.errstate(invalid="ignore", divide="ignore"):
            out[row] = table[kp] + (table[nbr] / d[:, None])[d > 0].sum(axis=0) / nbr.shape[0]
    return out


# ---------------------------------------------------------------------------------------------------------------
# SHOT (shot.py:16-306 through ShotMultiprocessor.compute_descriptor_single_scale, shot_parallelization.py:135-183)
# ---------------------------------------------------------------------------------------------------------------
def _frame(task):
    """get_local_rf, shot.py:16-48."""
    p, pts, radius = task
    if pts.shape[0] == 0:
        return np.eye(3)
    c = pts - p
    wgt = radius - np.linalg.norm(c, axis=1)
    cov = c.T @ (c * wgt[:, None]) / wgt.sum()
    _, vec = np.linalg.eigh(cov)
    for col in (2, 0):
        proj = (pts - p) @ vec[:, col]
        if (proj < 0).sum() > (proj >= 0).sum():
            vec[:, col] *= -1
    vec[:, 1] = np.cross(vec[:, 0], vec[:, 2])
    return np.flip(vec, axis=1)


def _octant(x, y):
    """get_azimuth_idx, shot.py:51-70 (three boolean maps combined into 0..7)."""
    a = (y > 0) | ((y == 0) & (x < 0))
    return 4 * a + 2 * np.logical_xor((x > 0) | ((x == 0) & (y > 0)), a) + np.where(
        (x * y > 0) | (x == 0), np.abs(x) < np.abs(y), np.abs(x) > np.abs(y))


def _shells(d, r):
    """interpolate_on_adjacent_husks, shot.py:73-118."""
    h = r / 2
    cur = (d < h) * (1 - np.abs(d - r / 4) / h) + (d > h) * (1 - np.abs(d - 3 * r / 4) / h)
    return ((d < h) & (d > r / 4)) * (d - r / 4) / h, ((d > h) & (d < 3 * r / 4)) * (3 * r / 4 - d) / h, cur


def _elevations(phi, z):
    """interpolate_vertical_volumes, shot.py:121-171."""
    q, near = PI / 2, np.abs(phi - PI / 2) < 1e-10
    up = (((phi > q) | (near & (z <= 0))) & (phi <= 3 * PI / 4)) * (3 * PI / 4 - phi) / q
    lo = (((phi < q) & (~near | (z > 0))) & (phi >= PI / 4)) * (phi - PI / 4) / q
    cur = (phi < q) * (1 - np.abs(phi - PI / 4) / q) + (phi >= q) * (1 - np.abs(phi - 3 * PI / 4) / q)
    return up, lo, cur


def _descriptor(task):
    """compute_single_shot_descriptor, shot.py:175-306: ten gather-add-scatter statements (last writer wins)."""
    p, pts, nrm, radius, frame, normalize, min_nb = task
    D = np.zeros((11, 8, 2, 2))
    rho = np.linalg.norm(pts - p, axis=1)
    if (rho > 0).sum() <= min_nb:
        return np.zeros(352)
    keep = rho > 0
    loc = (pts[keep] - p) @ frame
    cosv = np.clip(nrm[keep] @ frame[:, 2].T, -1, 1)
    rho = rho[keep]
    order = np.argsort(rho)
    rho, loc, cosv = rho[order], loc[order], cosv[order]
    theta = np.arctan2(loc[:, 1], loc[:, 0])
    phi = np.arccos(np.clip(loc[:, 2] / rho, -1, 1))
    cpos = (cosv + 1.0) * 11 / 2.0 - 0.5
    ci = np.rint(cpos).astype(int)
    ti = _octant(loc[:, 0], loc[:, 1])
    pi_ = (loc[:, 2] > 0).astype(int)
    ri = (rho > radius / 2).astype(int)
    dc = cpos - ci
    sc = np.sign(dc)
    adc = sc * dc
    D[(ci + sc).astype(int) % 11, ti, pi_, ri] += adc * ((ci > -0.5) & (ci < 10.5))
    D[ci, ti, pi_, ri] += 1 - adc
    outer, inner, cur = _shells(rho, radius)
    D[ci, ti, pi_, 1] += outer * (ri == 0)
    D[ci, ti, pi_, 0] += inner * (ri == 1)
    D[ci, ti, pi_, ri] += cur
    up, lo, curv = _elevations(phi, loc[:, 2])
    D[ci, ti, 1, ri] += up * (pi_ == 0)
    D[ci, ti, 0, ri] += lo * (pi_ == 1)
    D[ci, ti, pi_, ri] += curv
    size = 2 * PI / 8
    dt = np.clip((theta - (-PI + ti * size)) / size - 0.5, -0.5, 0.5)
    st = np.sign(dt)
    adt = st * dt
    D[ci, (ti + st).astype(int) % 8, pi_, ri] += adt
    D[ci, ti, pi_, ri] += 1 - adt
    if normalize:
        nr = np.linalg.norm(D)
        return D.ravel() / nr if nr > 0 else D.ravel()
    return D.ravel()


def shot_numpy_shaped(cloud, normals, keypoints, radius, normalize=True, min_neighborhood_size=10, n_procs=8):
    """One KDTree search in the parent, then frames (chunksize 1) and descriptors (chunksize ceil(M / 2 n_procs))
    through a fork Pool, every task carrying its pickled neighbourhood -- shot_parallelization.py:30-44, 65-84, 109-133."""
    from sklearn.neighbors import KDTree

    lists = KDTree(cloud).query_radius(keypoints, radius)
    with mp.get_context("fork").Pool(processes=n_procs) as pool:
        frames = np.array(list(pool.imap(_frame, [(kp, cloud[lists[i]], radius) for i, kp in enumerate(keypoints)])))
        chunk = int(np.ceil(keypoints.shape[0] / (2 * n_procs)))
        desc = np.array(list(pool.imap(
            _descriptor,
            [(kp, cloud[lists[i]], normals[lists[i]], radius, frames[i], normalize, min_neighborhood_size)
             for i, kp in enumerate(keypoints)], chunksize=chunk)))
    return desc


# ---------------------------------------------------------------------------------------------------------------
def _cloud(n, seed):
    rng = np.random.default_rng(seed)
    p = rng.random((n, 3), dtype=np.float32).astype(np.float64)
    nr = rng.standard_normal((n, 3))
    nr /= np.linalg.norm(nr, axis=1)[:, None]
    return p, nr


def timed_sample(points_per_gpu: int, radius: float, sample_points: int, n_procs: int) -> dict:
    """FPFH + SHOT, every point a keypoint, on a `sample_points` cloud at the SAME expected neighbours per ball as the
    GPU workload (radius scaled by (n / sample)^(1/3))."""
    ns = min(sample_points, points_per_gpu)
    r = radius * (points_per_gpu / ns) ** (1.0 / 3.0)
    p, nr = _cloud(ns, 33)
    t0 = time.perf_counter()
    f = fpfh_numpy_shaped(np.arange(ns), p, nr, r, 5)
    t1 = time.perf_counter()
    d = shot_numpy_shaped(p, nr, p, r, True, 10, n_procs)
    t2 = time.perf_counter()
    return {"points": ns, "radius": r, "fpfh_s": t1 - t0, "shot_s": t2 - t1, "n_procs": n_procs,
            "desc_per_s": 2 * ns / (t2 - t0), "fpfh_checksum": float(f.sum()), "shot_checksum": float(d.sum())}


if __name__ == "__main__":
    # python oracle/numpy_shaped.py POINTS_PER_GPU RADIUS SAMPLE_POINTS N_PROCS  -> one JSON line
    a = sys.argv[1:]
    print(json.dumps(timed_sample(int(a[0]), float(a[1]), int(a[2]), int(a[3]))))
