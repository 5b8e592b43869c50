"""ICP refinement on the GPU -- drop-in for shot_fpfh.icp (icp.py:20-189) and core.solvers.compute_point_to_point_error
(solvers.py:51-62).

How an iteration is split.  The reference does everything in NumPy around a KDTree query: transform the working points,
query their nearest reference points, mask the pairs farther than `d_max`, build the centred 3x3 cross-covariance
(point-to-point) or the 6x6 normal equations (point-to-plane) from the masked arrays, solve, compose.  Here the working
points and the reference cloud live in HBM for the whole run and ONE device call per iteration (`sf_icp_accumulate`,
csrc/icp.hip) does the transform, the nearest-neighbour search (grid k-NN kernel, k = 1), the `d_max` filter and the
reductions; what crosses the bus per iteration is a 12-number transform one way and ~30 sums the other.  The host keeps
only the tiny dense step -- a 3x3 SVD (`_rigid_fit`) or a 6x6 solve (`_plane_fit`) on those sums -- and the bookkeeping of
the running transform.  `_Registration` is that device state; the three public functions differ only in which rows they
feed it, which fit they ask for and what they return.

Deviation, on purpose: the reference's `icp_point_to_point` computes its RMS from `ref[neighbors]` (all queried points,
shape (n, 1, 3)) instead of `ref[inliers_neighbors]` (icp.py:122-124); the broadcast yields an array, and formatting it for
the progress bar raises TypeError on the first iteration, so that function cannot run there at all.  Here the RMS is taken
over the inlier pairs, as in `icp_point_to_point_with_sampling`.
"""
from __future__ import annotations

import ctypes as C
import logging
from typing import Optional

import numpy as np
import numpy.typing as npt
from scipy.spatial.transform import Rotation

from . import _ffi
from .core import RigidTransform, grid_subsampling
from .engine import Cloud, DeviceArray, Engine, default_engine

__all__ = [
    "icp_point_to_point_with_sampling",
    "icp_point_to_point",
    "icp_point_to_plane",
    "compute_point_to_point_error",
    "nearest_within",
]

_POINT, _PLANE = 0, 1
_TRIU = np.triu_indices(6)


class _PairSums:
    """What one device pass returns about the inlier pairs (p = moved working point, q = its nearest reference point)."""

    def __init__(self, raw: np.ndarray, mode: int):
        self.count = int(raw[0])
        self.sum_p, self.sum_q = raw[1:4], raw[4:7]
        if mode == _POINT:
            self.cross_cov = raw[8:17].reshape(3, 3)  # sum (p - pbar)(q - qbar)^T
            self.sq_dist = float(raw[17])             # sum |p - q|^2
        else:
            self.gtg = np.zeros((6, 6))
            self.gtg[_TRIU] = raw[8:29]
            self.gtg = self.gtg + np.triu(self.gtg, 1).T
            self.gth = raw[29:35]
            self.abs_h = float(raw[35])               # sum |(q - p) . n|

    def require_pairs(self) -> None:
        if self.count == 0:
            raise np.linalg.LinAlgError("ICP: no scan point has a reference point within d_max")


def _rigid_fit(s: _PairSums) -> RigidTransform:
    """Kabsch from the centred cross-covariance (core/solvers.py:9-30: same SVD, same reflection rule)."""
    s.require_pairs()
    u, _, vt = np.linalg.svd(s.cross_cov)
    rot = vt.T @ u.T
    if np.linalg.det(rot) < 0:
        ut = u.T.copy()
        ut[-1] *= -1
        rot = vt.T @ ut
    return RigidTransform(rot, s.sum_q / s.count - rot.dot(s.sum_p / s.count))


def _plane_fit(s: _PairSums) -> RigidTransform:
    """Linearised point-to-plane step from G^T G and G^T h (core/solvers.py:33-48)."""
    s.require_pairs()
    sol = np.linalg.solve(s.gtg, s.gth)
    return RigidTransform(Rotation.from_euler("xyz", sol[:3]).as_matrix(), sol[3:6])


class _Registration:
    """Working points + reference cloud resident on one GPU for the length of an ICP run."""

    def __init__(self, points, ref, ref_normals=None, engine: Optional[Engine] = None):
        self.engine = engine or default_engine()
        self.ref = Cloud(self.engine, ref, ref_normals)
        pts = np.ascontiguousarray(points, dtype=np.float64)
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise ValueError(f"expected an (N, 3) array, got shape {pts.shape}")
        self.n = pts.shape[0]
        self.points: DeviceArray = self.engine.empty((max(self.n, 1), 3)).from_host(pts if self.n else np.zeros((1, 3)))
        self.rows: Optional[DeviceArray] = None

    def pairs(self, mode: int, d_max: float, moved_by: Optional[RigidTransform] = None, rows=None) -> _PairSums:
        """Inlier-pair sums of the working points (all of them, or the given `rows`) after `moved_by`."""
        m, sel = self.n, None
        if rows is not None:
            rows = np.ascontiguousarray(rows, dtype=np.int64)
            m = rows.shape[0]
            if self.rows is None or self.rows.shape[0] < m:
                if self.rows is not None:
                    self.rows.free()
                self.rows = self.engine.empty((max(m, 1),), np.int64)
            if m:
                _ffi.check(self.engine.lib.sf_h2d(self.engine.h, self.rows.ptr, rows.ctypes.data_as(C.c_void_p), m * 8), "sf_h2d")
            sel = self.rows.ptr
        rt = None if moved_by is None else np.ascontiguousarray(moved_by.as_row12())
        raw = np.zeros(40)
        _ffi.check(
            self.engine.lib.sf_icp_accumulate(self.engine.h, self.ref.h, self.points.ptr, sel, m,
                                              None if rt is None else rt.ctypes.data_as(C.c_void_p), float(d_max), mode,
                                              raw.ctypes.data_as(C.c_void_p)),
            "sf_icp_accumulate",
        )
        return _PairSums(raw, mode)

    def move(self, transform: RigidTransform) -> None:
        """points <- transform[points], in place in HBM."""
        rt = np.ascontiguousarray(transform.as_row12())
        _ffi.check(self.engine.lib.sf_transform_points(self.engine.h, self.points.ptr, self.n, rt.ctypes.data_as(C.c_void_p)),
                   "sf_transform_points")

    def download(self) -> np.ndarray:
        return self.points.to_host()[: self.n]

    def close(self) -> None:
        for obj in (self.points, self.rows, self.ref):
            if obj is not None:
                obj.free()


def icp_point_to_point_with_sampling(
    scan: npt.NDArray[np.float64],
    ref: npt.NDArray[np.float64],
    d_max: float,
    max_iter: int = 100,
    rms_threshold: float = 1e-2,
    sampling_limit: int = 100,
    disable_progress_bar: bool = False,
) -> tuple[npt.NDArray[np.float64], float, bool]:
    """Point-to-point ICP fitted on a fresh random subset per iteration and applied to ALL points (icp.py:20-78).
    Subsets are drawn from NumPy's global generator exactly as the reference draws them.
    Returns (aligned points, rms of the last fit = sqrt(sum of squared inlier distances), converged)."""
    reg = _Registration(scan, ref)
    limit = min(sampling_limit, reg.n)
    rms = 0.0
    try:
        for _ in range(max_iter):
            subset = np.random.choice(reg.n, limit, replace=False)
            found = reg.pairs(_POINT, d_max, rows=subset)
            reg.move(_rigid_fit(found))
            rms = float(np.sqrt(found.sq_dist))
            if rms < rms_threshold:
                break
    except KeyboardInterrupt:
        logging.info("ICP interrupted by user.")
    try:
        return reg.download(), rms, rms < rms_threshold
    finally:
        reg.close()


def _refine(reg: _Registration, start: RigidTransform, mode: int, d_max: float, max_iter: int, rms_threshold: float):
    """The loop shared by icp_point_to_point and icp_point_to_plane: the working points stay where they are, the
    running transform is what moves (icp.py:103-130, 155-189)."""
    total, rms = start, 0.0
    fit = _rigid_fit if mode == _POINT else _plane_fit
    try:
        for _ in range(max_iter):
            found = reg.pairs(mode, d_max, moved_by=total)
            total = fit(found) @ total
            # residual of the pairs the fit was computed FROM (before this iteration's update), as in the reference
            rms = float(np.sqrt(found.sq_dist)) if mode == _POINT else found.abs_h / found.count
            if rms < rms_threshold:
                logging.info("RMS threshold reached.")
                break
    except KeyboardInterrupt:
        logging.info("ICP interrupted by user.")
    return total, rms, rms < rms_threshold


def icp_point_to_point(
    scan: npt.NDArray[np.float64],
    ref: npt.NDArray[np.float64],
    transformation_init: RigidTransform,
    d_max: float,
    voxel_size: float = 0.2,
    max_iter: int = 100,
    rms_threshold: float = 1e-2,
    disable_progress_bar: bool = False,
) -> tuple[RigidTransform, float, bool]:
    """Point-to-point ICP on the voxel-subsampled scan (icp.py:81-130; see the module note on its RMS)."""
    scan = np.asarray(scan)
    reg = _Registration(scan[grid_subsampling(scan, voxel_size)], ref)
    try:
        return _refine(reg, transformation_init, _POINT, d_max, max_iter, rms_threshold)
    finally:
        reg.close()


def icp_point_to_plane(
    scan: npt.NDArray[np.float64],
    ref: npt.NDArray[np.float64],
    ref_normals: npt.NDArray[np.float64],
    transformation_init: RigidTransform,
    d_max: float,
    voxel_size: float = 0.2,
    max_iter: int = 50,
    rms_threshold: float = 1e-2,
    disable_progress_bar: bool = False,
) -> tuple[RigidTransform, float, bool]:
    """Point-to-plane ICP (icp.py:133-189): the returned rms is the mean |(inlier - neighbour) . normal| of the pairs
    the last step was fitted on."""
    scan = np.asarray(scan)
    reg = _Registration(scan[grid_subsampling(scan, voxel_size)], ref, ref_normals)
    try:
        return _refine(reg, transformation_init, _PLANE, d_max, max_iter, rms_threshold)
    finally:
        reg.close()


def compute_point_to_point_error(
    scan: npt.NDArray[np.float64], ref: npt.NDArray[np.float64], transformation: RigidTransform
) -> tuple[float, npt.NDArray[np.float64]]:
    """RMS nearest-neighbour distance of the transformed scan to the reference cloud, and the transformed scan
    (core/solvers.py:51-62).  Every point counts: no d_max."""
    reg = _Registration(scan, ref)
    try:
        found = reg.pairs(_POINT, np.inf, moved_by=transformation)
    finally:
        reg.close()
    return float(np.sqrt(found.sq_dist / max(found.count, 1))), transformation[np.asarray(scan)]


def nearest_within(points: npt.NDArray[np.float64], against: npt.NDArray[np.float64], distance: float) -> int:
    """How many of `points` have a point of `against` within `distance` (the `KDTree(ref).query(...)[0] <= thr` masks
    of pipeline.py:560-587, summed) -- counted on the device."""
    reg = _Registration(points, against)
    try:
        return reg.pairs(_POINT, distance).count
    finally:
        reg.close()
