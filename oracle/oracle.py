"""ctypes front-end of the CPU oracle (oracle/shot_fpfh_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of the C file.  Nothing in shot_fpfh_amd/ imports
this module; it is loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

The thin Python layer on top of the C functions restates the host-side glue of the reference
(zero-row filtering in matching, keypoint-by-coordinate SHOT driver, ...) with file:line
citations so parity tests can call functions shaped like the reference's.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc if it is missing or older than its source."""
    src = os.path.join(_HERE, "shot_fpfh_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_radius_search.restype = C.c_int64
        L.orc_radius_search.argtypes = [_f64p, C.c_int64, _f64p, C.c_int64, C.c_double, _i64p, C.c_void_p, C.c_void_p]
        L.orc_radius_search_brute.restype = C.c_int64
        L.orc_radius_search_brute.argtypes = [_f64p, C.c_int64, _f64p, C.c_int64, C.c_double, _i64p, C.c_void_p]
        L.orc_eigh3.restype = None
        L.orc_eigh3.argtypes = [_f64p, _f64p, _f64p]
        L.orc_normals_radius.restype = C.c_int
        L.orc_normals_radius.argtypes = [_f64p, C.c_int64, _f64p, C.c_int64, C.c_double, C.c_void_p, _f64p]
        L.orc_normals_from_lists.restype = None
        L.orc_normals_from_lists.argtypes = [_f64p, _i64p, _i32p, C.c_int64, C.c_void_p, _f64p]
        L.orc_pca_from_lists.restype = None
        L.orc_pca_from_lists.argtypes = [_f64p, _i64p, _i32p, C.c_int64, _f64p, _f64p, C.c_void_p]
        L.orc_shot_lrf.restype = C.c_int
        L.orc_shot_lrf.argtypes = [_f64p, C.c_int64, _f64p, C.c_int64, C.c_double, _f64p]
        L.orc_shot.restype = C.c_int
        L.orc_shot.argtypes = [_f64p, _f64p, C.c_int64, _f64p, C.c_int64, C.c_double, _f64p, C.c_int, C.c_int64, _f64p]
        L.orc_azimuth_idx.restype = C.c_int
        L.orc_azimuth_idx.argtypes = [C.c_double, C.c_double]
        L.orc_lrf_single.restype = None
        L.orc_lrf_single.argtypes = [_f64p, _f64p, _i32p, C.c_int64, C.c_double, _f64p]
        L.orc_shot_single.restype = None
        L.orc_shot_single.argtypes = [_f64p, _f64p, _f64p, _i32p, C.c_int64, C.c_double, _f64p, C.c_int, C.c_int64, _f64p]
        L.orc_shot_serial.restype = C.c_int
        L.orc_shot_serial.argtypes = [_f64p, _f64p, C.c_int64, _f64p, C.c_int64, C.c_double, C.c_int64, _f64p]
        L.orc_fpfh_sample.restype = C.c_int
        L.orc_fpfh_sample.argtypes = [_f64p, _f64p, C.c_int64, _i64p, C.c_int64, C.c_double, C.c_int, _f64p, _f64p]
        L.orc_fpfh.restype = C.c_int
        L.orc_fpfh.argtypes = [_f64p, _f64p, C.c_int64, _i64p, C.c_int64, C.c_double, C.c_int, _f64p, _f64p, C.c_void_p]
        L.orc_match_argmin.restype = None
        L.orc_match_argmin.argtypes = [_f64p, C.c_int64, _f64p, C.c_int64, C.c_int64, _i64p, C.c_void_p, C.c_void_p]
        L.orc_ransac_score.restype = None
        L.orc_ransac_score.argtypes = [_f64p, _f64p, C.c_int64, _f64p, C.c_int64, C.c_double, _i64p]
        _lib = L
    return _lib


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------------------------
# (a1) radius search
# --------------------------------------------------------------------------------------------
def radius_search(cloud, queries, radius, return_distance=False, brute=False):
    """CSR neighbour lists (ascending index inside each list) of KDTree(cloud).query_radius(queries, radius)."""
    cloud, queries = _f64(cloud), _f64(queries)
    m = queries.shape[0]
    off = np.zeros(m + 1, dtype=np.int64)
    L = lib()
    if brute:
        total = L.orc_radius_search_brute(cloud, cloud.shape[0], queries, m, radius, off, None)
        idx = np.zeros(max(total, 1), dtype=np.int32)
        L.orc_radius_search_brute(cloud, cloud.shape[0], queries, m, radius, off, _ptr(idx))
        return off, idx[:total]
    total = L.orc_radius_search(cloud, cloud.shape[0], queries, m, radius, off, None, None)
    idx = np.zeros(max(total, 1), dtype=np.int32)
    dist = np.zeros(max(total, 1), dtype=np.float64) if return_distance else None
    L.orc_radius_search(cloud, cloud.shape[0], queries, m, radius, off, _ptr(idx), _ptr(dist))
    if return_distance:
        return off, idx[:total], dist[:total]
    return off, idx[:total]


def eigh3(a):
    a = _f64(a).reshape(3, 3)
    w = np.zeros(3)
    v = np.zeros((3, 3))
    lib().orc_eigh3(a, w, v)
    return w, v


# --------------------------------------------------------------------------------------------
# (a2) compute_normals, radius branch (pca_based_descriptors.py:29-59)
# --------------------------------------------------------------------------------------------
def compute_normals(query_points, cloud_points, *, k=None, radius=None, pre_computed_normals=None, knn_radius_hint=None):
    assert k is not None or radius is not None, "No parameter provided for the neighborhood search."
    q, p = _f64(query_points), _f64(cloud_points)
    out = np.zeros((q.shape[0], 3))
    pre = None if pre_computed_normals is None else _f64(pre_computed_normals)
    if k is not None:
        off, idx = knn_lists(p, q, k, knn_radius_hint)
        lib().orc_normals_from_lists(p, off, idx, q.shape[0], _ptr(pre), out)
    else:
        lib().orc_normals_radius(p, p.shape[0], q, q.shape[0], radius, _ptr(pre), out)
    return out


def local_pca(query_points, cloud_points, *, radius=None, k=None, moments=False):
    """pca() / compute_local_pca_with_moments (pca_based_descriptors.py:15-26, 75-146): eigenvalues (m,3),
    eigenvectors (m,3,3) as np.linalg.eigh returns them, [moments (m,8)], neighbourhood sizes (m,)."""
    q, p = _f64(query_points), _f64(cloud_points)
    if k is not None:
        off, idx = knn_lists(p, q, k)
    else:
        off, idx = radius_search(p, q, radius)
    m = q.shape[0]
    w, v = np.zeros((m, 3)), np.zeros((m, 9))
    mo = np.zeros((m, 8)) if moments else None
    lib().orc_pca_from_lists(p, np.ascontiguousarray(off, np.int64), np.ascontiguousarray(idx, np.int32), m, w, v, _ptr(mo))
    sizes = np.diff(off)
    return (w, v.reshape(m, 3, 3), mo, sizes) if moments else (w, v.reshape(m, 3, 3), sizes)


def compute_pca_based_features(query_points, cloud_points, radius):
    """The (N, 21) feature matrix of pca_based_descriptors.py:190-244 on top of local_pca."""
    ev, vec, moments, sizes = local_pca(query_points, cloud_points, radius=radius, moments=True)
    lbd3, lbd2, lbd1 = ev[:, 0], ev[:, 1], ev[:, 2]
    lbd1 += 1e-6  # in place, before the sums below (as in the reference)
    normals, principal_axis = vec[:, :, 0], vec[:, :, 2]
    eigensum = ev.sum(axis=-1)
    cols = [
        eigensum, (ev**2).sum(axis=-1), np.cbrt(ev.prod(axis=-1)), (-ev * np.log(ev + 1e-6)).sum(axis=-1),
        1 - lbd2 / lbd1, (lbd2 - lbd3) / lbd1, lbd3 / lbd1, lbd3 / eigensum,
        2 * np.arcsin(np.abs(normals[:, 2])) / np.pi, 2 * np.arcsin(np.abs(principal_axis[:, 2])) / np.pi,
        2 * np.arcsin(np.abs(normals[:, 0])) / np.pi, 2 * np.arcsin(np.abs(normals[:, 1])) / np.pi,
    ]
    return np.hstack([c[:, None] for c in cols] + [moments, sizes[:, None].astype(np.float64)])


def compute_pca_based_basic_features(query_points, cloud_points, radius):
    """(verticality, linearity, planarity, sphericity), pca_based_descriptors.py:150-187."""
    ev, vec, _ = local_pca(query_points, cloud_points, radius=radius)
    lbd3, lbd2, lbd1 = ev[:, 0], ev[:, 1], ev[:, 2] + 1e-6
    return 2 * np.arcsin(np.abs(vec[:, 2, 0])) / np.pi, 1 - lbd2 / lbd1, (lbd2 - lbd3) / lbd1, lbd3 / lbd1


def compute_sphericity(query_points, cloud_points, radius):
    """pca_based_descriptors.py:60-73."""
    ev = local_pca(query_points, cloud_points, radius=radius)[0]
    return ev[:, 0] / (ev[:, 2] + 1e-6)


def knn_lists(cloud, queries, k, radius_hint=None):
    """k nearest neighbours as CSR (KDTree.query(k=k, return_distance=False), pca_based_descriptors.py:46).
    Ties in distance are broken by lower index.  Brute force by default; with `radius_hint` the candidates come
    from the C radius search (any query with fewer than k points inside the hint falls back to brute force)."""
    cloud, queries = _f64(cloud), _f64(queries)
    m = queries.shape[0]
    idx = np.zeros((m, k), dtype=np.int32)
    todo = np.arange(m)
    if radius_hint is not None:
        off, cand = radius_search(cloud, queries, radius_hint)
        short = []
        for i in range(m):
            c = cand[off[i] : off[i + 1]]  # ascending index
            if c.shape[0] < k:
                short.append(i)
                continue
            d = cloud[c] - queries[i]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            idx[i] = c[np.argsort(d2, kind="stable")[:k]]
        todo = np.array(short, dtype=np.int64)
    for i in todo:
        d = cloud - queries[i]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        idx[i] = np.argsort(d2, kind="stable")[:k]
    off = np.arange(0, (m + 1) * k, k, dtype=np.int64)
    return off, np.ascontiguousarray(idx.reshape(-1))


# --------------------------------------------------------------------------------------------
# (a3, a7, a9) SHOT
# --------------------------------------------------------------------------------------------
def azimuth_idx(x, y):
    """get_azimuth_idx (shot.py:51-70), elementwise."""
    L = lib()
    return np.array([L.orc_azimuth_idx(float(a), float(b)) for a, b in zip(np.ravel(x), np.ravel(y))], dtype=np.int64)


def shot_lrf(cloud, keypoints, radius):
    p, q = _f64(cloud), _f64(keypoints)
    out = np.zeros((q.shape[0], 3, 3))
    lib().orc_shot_lrf(p, p.shape[0], q, q.shape[0], radius, out.reshape(-1, 9))
    return out


def shot(cloud, normals, keypoints, radius, lrf, normalize=True, min_neighborhood_size=100):
    p, nrm, q = _f64(cloud), _f64(normals), _f64(keypoints)
    lrf = _f64(lrf).reshape(-1, 9)
    out = np.zeros((q.shape[0], 352))
    lib().orc_shot(p, nrm, p.shape[0], q, q.shape[0], radius, lrf, int(bool(normalize)), int(min_neighborhood_size), out)
    return out


def shot_single(point, neighbors, normals, radius, lrf, normalize=True, min_neighborhood_size=100):
    """compute_single_shot_descriptor on one explicit neighbourhood (shot.py:175-306)."""
    nbh, nrm = _f64(neighbors), _f64(normals)
    out = np.zeros(352)
    lib().orc_shot_single(_f64(point), nbh, nrm, np.arange(nbh.shape[0], dtype=np.int32), nbh.shape[0], float(radius),
                          _f64(lrf).reshape(9), int(bool(normalize)), int(min_neighborhood_size), out)
    return out


def shot_lrf_lists(cloud, keypoints, offsets, idx, radius):
    """ShotMultiprocessor.compute_local_rf on the caller's lists (shot_parallelization.py:46-84): keypoint i's
    neighbourhood is cloud[idx[offsets[i]:offsets[i+1]]]."""
    p, q = _f64(cloud), _f64(keypoints)
    idx32 = np.ascontiguousarray(idx, dtype=np.int32)
    out = np.zeros((q.shape[0], 3, 3))
    L = lib()
    for i in range(q.shape[0]):
        a, b = int(offsets[i]), int(offsets[i + 1])
        L.orc_lrf_single(q[i], p, idx32[a:b] if b > a else np.zeros(1, np.int32), b - a, float(radius), out[i].reshape(9))
    return out


def shot_lists(cloud, normals, keypoints, offsets, idx, radius, lrf, normalize=True, min_neighborhood_size=100):
    """ShotMultiprocessor.compute_descriptor on the caller's lists (shot_parallelization.py:86-133)."""
    p, nrm, q = _f64(cloud), _f64(normals), _f64(keypoints)
    lrf = _f64(lrf).reshape(-1, 9)
    idx32 = np.ascontiguousarray(idx, dtype=np.int32)
    out = np.zeros((q.shape[0], 352))
    L = lib()
    for i in range(q.shape[0]):
        a, b = int(offsets[i]), int(offsets[i + 1])
        L.orc_shot_single(q[i], p, nrm, idx32[a:b] if b > a else np.zeros(1, np.int32), b - a, float(radius), lrf[i],
                          int(bool(normalize)), int(min_neighborhood_size), out[i])
    return out


def shot_single_scale(point_cloud, normals, keypoints, radius, normalize=True, min_neighborhood_size=100, support=None):
    """ShotMultiprocessor.compute_descriptor_single_scale (shot_parallelization.py:135-183);
    `support` is the index array grid_subsampling would return (or None)."""
    pc = _f64(point_cloud) if support is None else _f64(np.asarray(point_cloud)[support])
    nr = _f64(normals) if support is None else _f64(np.asarray(normals)[support])
    lrf = shot_lrf(pc, keypoints, radius)
    return shot(pc, nr, keypoints, radius, lrf, normalize, min_neighborhood_size)


def compute_shot_descriptor(keypoints, cloud_points, normals, radius, min_neighborhood_size=10):
    """The serial / debug variant compute_shot_descriptor (shot.py:310-499): frames from the neighbours at
    non-zero distance only, rows always normalised."""
    p, nrm, q = _f64(cloud_points), _f64(normals), _f64(keypoints)
    out = np.zeros((q.shape[0], 352))
    rc = lib().orc_shot_serial(p, nrm, p.shape[0], q, q.shape[0], radius, int(min_neighborhood_size), out)
    assert rc == 0
    return out


# --------------------------------------------------------------------------------------------
# (a10) FPFH
# --------------------------------------------------------------------------------------------
def fpfh_edges(n_bins):
    """Bin edges exactly as np.histogramdd builds them for fpfh.py:82-87."""
    return np.ascontiguousarray(
        np.stack(
            [
                np.linspace(-1, 1, n_bins + 1),
                np.linspace(-1, 1, n_bins + 1),
                np.linspace(-np.pi / 2, np.pi / 2, n_bins + 1),
            ]
        )
    )


def compute_fpfh_descriptor(keypoints_indices, cloud_points, normals, radius, n_bins, return_spfh=False):
    p, nrm = _f64(cloud_points), _f64(normals)
    kp = np.ascontiguousarray(keypoints_indices, dtype=np.int64)
    out = np.zeros((kp.shape[0], n_bins**3))
    spfh = np.zeros((p.shape[0], n_bins**3)) if return_spfh else None
    rc = lib().orc_fpfh(p, nrm, p.shape[0], kp, kp.shape[0], radius, n_bins, fpfh_edges(n_bins), out, _ptr(spfh))
    assert rc == 0
    return (out, spfh) if return_spfh else out


def compute_fpfh_descriptor_sample(keypoints_indices, cloud_points, normals, radius, n_bins):
    """compute_fpfh_descriptor for a SAMPLE of keypoints of a large cloud: the SPFH rows are evaluated only for the
    sample's keypoints and their neighbours (bit-identical rows; affordable at 1M / 8M points)."""
    p, nrm = _f64(cloud_points), _f64(normals)
    kp = np.ascontiguousarray(keypoints_indices, dtype=np.int64)
    out = np.zeros((kp.shape[0], n_bins**3))
    rc = lib().orc_fpfh_sample(p, nrm, p.shape[0], kp, kp.shape[0], radius, n_bins, fpfh_edges(n_bins), out)
    assert rc == 0
    return out


# --------------------------------------------------------------------------------------------
# (a11) matching
# --------------------------------------------------------------------------------------------
def match_argmin(a, b, want_col=False):
    a, b = _f64(a), _f64(b)
    m1, m2 = a.shape[0], b.shape[0]
    idx = np.zeros(m1, dtype=np.int64)
    dist = np.zeros(m1)
    col = np.zeros(m2, dtype=np.int64) if want_col else None
    lib().orc_match_argmin(a, m1, b, m2, a.shape[1], idx, _ptr(dist), _ptr(col))
    return (idx, dist, col) if want_col else (idx, dist)


def basic_matching(scan_descriptors, ref_descriptors):
    """matching.py:149-169."""
    ne_s = np.any(scan_descriptors, axis=1).nonzero()[0]
    ne_r = np.any(ref_descriptors, axis=1).nonzero()[0]
    idx, _ = match_argmin(scan_descriptors[ne_s], ref_descriptors[ne_r])
    return ne_s, ne_r[idx]


def match_descriptors(scan, ref, filter_callback=None, filter_nonreciprocal=False, n_min_matches=100, **kwargs):
    """2-D branch of matching.py:39-74, 138-146."""
    ne_s = np.any(scan, axis=1).nonzero()[0]
    ne_r = np.any(ref, axis=1).nonzero()[0]
    idx, dist, col = match_argmin(scan[ne_s], ref[ne_r], want_col=True)
    keep = filter_callback(dist, **kwargs) if filter_callback is not None else np.ones(dist.shape[0], dtype=bool)
    if filter_nonreciprocal:
        recip = col[idx] == np.arange(idx.shape[0])
        both = keep & recip
        if both.sum() >= n_min_matches:
            keep = both
    return ne_s[keep], ne_r[idx[keep]]


# --------------------------------------------------------------------------------------------
# (a12) RANSAC scoring
# --------------------------------------------------------------------------------------------
def ransac_score(a, b, rt, thr):
    a, b, rt = _f64(a), _f64(b), _f64(rt).reshape(-1, 12)
    out = np.zeros(rt.shape[0], dtype=np.int64)
    lib().orc_ransac_score(a, b, a.shape[0], rt, rt.shape[0], thr, out)
    return out


def match_descriptors_multiscale(scan, ref, filter_callback=None, max_val=1000, **kwargs):
    """3-D ("minimum over scales") branch of matching.py:77-136, restated with NumPy (small inputs only):
    per scale a distance matrix that is max_val wherever either descriptor is all-zero, element-wise minimum
    over scales, first arg-min per row, matches at max_val dropped."""
    scan, ref = _f64(scan), _f64(ref)
    n_scales, n_points, _ = scan.shape
    inf = np.full((n_points, ref.shape[1]), float(max_val))
    for s in range(n_scales):
        ne_s, ne_r = np.any(scan[s], axis=1), np.any(ref[s], axis=1)
        dm = np.full_like(inf, float(max_val))
        diff = scan[s][ne_s][:, None, :] - ref[s][ne_r][None, :, :]
        acc = np.zeros(diff.shape[:2])
        for t in range(diff.shape[2]):  # left-to-right sum, as scipy's cdist accumulates
            acc += diff[:, :, t] * diff[:, :, t]
        dm[np.ix_(ne_s, ne_r)] = np.sqrt(acc)
        inf = np.minimum(dm, inf)
    idx = inf.argmin(axis=1)
    dist = inf[np.arange(n_points), idx]
    keep = (filter_callback(dist, **kwargs) if filter_callback is not None else np.ones(n_points, dtype=bool)) & (dist < max_val)
    return np.arange(n_points)[keep], np.arange(ref.shape[1])[idx[keep]]
