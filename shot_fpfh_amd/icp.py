"""ICP refinement, the step after RANSAC in the reference pipeline (mirrors shot_fpfh/icp.py and
core/solvers.py:51-62).

Each iteration's nearest-neighbour query (`KDTree(ref).query(points)`) runs on the MI355X: the reference
cloud is uploaded once and searched with the grid k-NN kernel (k = 1).  The 3x3 Kabsch / 6x6 point-to-plane
solves on the inliers stay NumPy calls, written exactly as the reference writes them, so the iterates are the
reference's iterates.

Deviation, on purpose: the reference's `icp_point_to_point` computes its RMS from `ref[neighbors]` (all
queried points, shape (n, 1, 3)) instead of `ref[inliers_neighbors]` (icp.py:122-124); the broadcast yields an
array, and formatting it for the progress bar raises TypeError on the first iteration, so that function cannot
run there at all.  Here the RMS is taken over the inlier pairs, as in `icp_point_to_point_with_sampling`.
"""
from __future__ import annotations

import logging
from typing import Optional

import numpy as np
import numpy.typing as npt

from .core import RigidTransform, grid_subsampling, solver_point_to_plane, solver_point_to_point
from .engine import Cloud, Engine, default_engine

__all__ = [
    "icp_point_to_point_with_sampling",
    "icp_point_to_point",
    "icp_point_to_plane",
    "compute_point_to_point_error",
]


class _NearestNeighbour:
    """KDTree(ref).query(points) on the device: (distances (n,), indices (n,))."""

    def __init__(self, ref, engine: Optional[Engine] = None):
        self.cloud = Cloud(engine or default_engine(), ref)

    def query(self, points):
        nbrs = self.cloud.knn_search(points, 1)
        try:
            _, idx, dist = nbrs.export(return_distance=True)
        finally:
            nbrs.free()
        return dist, idx.astype(np.int64)

    def close(self):
        self.cloud.free()


def icp_point_to_point_with_sampling(
    scan: npt.NDArray[np.float64],
    ref: npt.NDArray[np.float64],
    d_max: float,
    max_iter: int = 100,
    rms_threshold: float = 1e-2,
    sampling_limit: int = 100,
    disable_progress_bar: bool = False,
) -> tuple[npt.NDArray[np.float64], float, bool]:
    """Point-to-point ICP on a fresh random subset per iteration (icp.py:19-77).  Subsets come from NumPy's
    global generator, as in the reference.  Returns (aligned points, rms, converged)."""
    points_aligned = np.copy(scan)
    nn = _NearestNeighbour(ref)
    sampling_limit = min(sampling_limit, scan.shape[0])
    rms = 0.0
    try:
        for _ in range(max_iter):
            indexes = np.random.choice(scan.shape[0], sampling_limit, replace=False)
            points_aligned_subset = points_aligned[indexes]
            distances, neighbors = nn.query(points_aligned_subset)
            inlier_points = points_aligned_subset[distances <= d_max]
            neighbors = neighbors[distances <= d_max]
            transformation = solver_point_to_point(inlier_points, ref[neighbors])
            rms = np.sqrt((np.linalg.norm(inlier_points - ref[neighbors], axis=1) ** 2).sum(axis=0))
            points_aligned = transformation[points_aligned]
            if rms < rms_threshold:
                break
    except KeyboardInterrupt:
        logging.info("ICP interrupted by user.")
    finally:
        nn.close()
    return points_aligned, rms, rms < rms_threshold


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
    """Point-to-point ICP on a voxel-subsampled scan (icp.py:80-135; see the module note on its RMS)."""
    nn = _NearestNeighbour(ref)
    subsampled_indices = grid_subsampling(scan, voxel_size)
    transformation_icp = transformation_init
    rms = 0.0
    try:
        for _ in range(max_iter):
            points_aligned = transformation_icp[scan[subsampled_indices]]
            distances, neighbors = nn.query(points_aligned)
            inliers = points_aligned[distances <= d_max]
            inliers_neighbors = neighbors[distances <= d_max]
            transformation_aligned_to_ref = solver_point_to_point(inliers, ref[inliers_neighbors])
            rms = np.sqrt((np.linalg.norm(inliers - ref[inliers_neighbors], axis=1) ** 2).sum(axis=0))
            transformation_icp = transformation_aligned_to_ref @ transformation_icp
            if rms < rms_threshold:
                logging.info("RMS threshold reached.")
                break
    except KeyboardInterrupt:
        logging.info("ICP interrupted by user.")
    finally:
        nn.close()
    return transformation_icp, rms, rms < rms_threshold


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
    """Point-to-plane ICP (icp.py:138-189): linearised 6-dof solve per iteration; the returned rms is the mean
    |(inlier - neighbour) . normal| measured BEFORE the iteration's update, as in the reference."""
    nn = _NearestNeighbour(ref)
    subsampled_indices = grid_subsampling(scan, voxel_size)
    transformation_icp = transformation_init
    rms = 0.0
    try:
        for _ in range(max_iter):
            points_aligned = transformation_icp[scan[subsampled_indices]]
            distances, neighbors = nn.query(points_aligned)
            inliers = points_aligned[distances <= d_max]
            inliers_neighbors = neighbors[distances <= d_max]
            transformation_aligned_to_ref = solver_point_to_plane(
                inliers, ref[inliers_neighbors], ref_normals[inliers_neighbors]
            )
            transformation_icp = transformation_aligned_to_ref @ transformation_icp
            rms = np.abs(
                np.einsum("ij, ij->i", inliers - ref[inliers_neighbors], ref_normals[inliers_neighbors])
            ).mean(axis=0)
            if rms < rms_threshold:
                logging.info("RMS threshold reached.")
                break
    except KeyboardInterrupt:
        logging.info("ICP interrupted by user.")
    finally:
        nn.close()
    return transformation_icp, rms, rms < rms_threshold


def compute_point_to_point_error(
    scan: npt.NDArray[np.float64], ref: npt.NDArray[np.float64], transformation: RigidTransform
) -> tuple[float, npt.NDArray[np.float64]]:
    """RMS nearest-neighbour distance of the transformed scan to the reference cloud, and the transformed scan
    (core/solvers.py:51-62)."""
    transformed_data = transformation[scan]
    nn = _NearestNeighbour(ref)
    try:
        distances, _ = nn.query(transformed_data)
    finally:
        nn.close()
    return np.sqrt((distances**2).mean()), transformed_data
