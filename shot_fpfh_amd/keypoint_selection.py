"""Keypoint selection, the step in front of the descriptor kernels (mirrors shot_fpfh/keypoint_selection.py).

Same names, arguments and results as the reference.  Wherever the reference builds a KDTree and calls
query_radius, the lists come from the uniform-grid radius search on the MI355X (kernels K1 + K2); the
remaining logic is index bookkeeping and stays on the host.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import numpy.typing as npt

from .core import grid_subsampling, voxel_closest_to_barycentre
from .engine import default_engine

__all__ = [
    "select_keypoints_iteratively",
    "select_keypoints_subsampling",
    "select_keypoints_randomly",
    "select_query_indices_randomly",
    "select_keypoints_with_density_threshold",
]

# module-level generator with the reference's seed (keypoint_selection.py:8); its state persists across calls
rng = np.random.default_rng(seed=1)

_BLOCK = 1 << 18  # query points per device search when sweeping a whole cloud (bounds the exported CSR)


def select_keypoints_iteratively(points: npt.NDArray[np.float64], radius: float) -> npt.NDArray[np.int64]:
    """Greedy cover (keypoint_selection.py:11-31): take the first point not yet visited, mark its spherical
    neighbourhood visited, repeat.  The neighbourhoods of ALL points are searched on the device, block by
    block; the sweep itself is sequential by definition and runs on the exported lists."""
    points = np.ascontiguousarray(points, dtype=np.float64)
    n = points.shape[0]
    selected = np.zeros(n, dtype=bool)
    visited = np.zeros(n, dtype=bool)
    if n == 0:
        return selected.nonzero()[0]
    cloud = default_engine().cloud(points)
    try:
        for begin in range(0, n, _BLOCK):
            end = min(begin + _BLOCK, n)
            off, idx = cloud.radius_search(points[begin:end], float(radius)).export()
            for i in range(begin, end):  # the first unvisited index only moves forward
                if not visited[i]:
                    selected[i] = True
                    visited[idx[off[i - begin] : off[i - begin + 1]]] = True
    finally:
        cloud.free()
    return selected.nonzero()[0]


def select_keypoints_subsampling(points: npt.NDArray[np.float64], voxel_size: float) -> npt.NDArray[np.int64]:
    """Point closest to the barycentre of every occupied voxel (keypoint_selection.py:34-44)."""
    return grid_subsampling(points, voxel_size)


def select_keypoints_randomly(points: npt.NDArray[np.float64], n_feature_points: int) -> npt.NDArray[np.float64]:
    """Random subset of the POINTS themselves, drawn from the module-level generator (keypoint_selection.py:47-53)."""
    return rng.choice(points, n_feature_points, replace=False, shuffle=False)


def select_query_indices_randomly(n_points: int, n_feature_points: int) -> npt.NDArray[np.int64]:
    """Random subset of indices from NumPy's global generator (keypoint_selection.py:56-62)."""
    return np.random.choice(n_points, n_feature_points, replace=False)


def select_keypoints_with_density_threshold(
    points: npt.NDArray[np.float64],
    voxel_size: float,
    density_threshold_value: int,
    density_threshold_radius: Optional[float] = None,
) -> npt.NDArray[np.int64]:
    """Voxel subsampling that keeps a voxel's representative only where the cloud is dense
    (keypoint_selection.py:65-122): more than `density_threshold_value` points in the voxel itself when the
    radius is the voxel size (or None), else within `density_threshold_radius` of the representative --
    counted by one device radius search over all representatives."""
    points = np.ascontiguousarray(points, dtype=np.float64)
    picked, counts = voxel_closest_to_barycentre(points, voxel_size)
    if density_threshold_radius is None:
        density_threshold_radius = voxel_size
    if density_threshold_radius == voxel_size:
        return picked[counts > density_threshold_value]
    cloud = default_engine().cloud(points)
    try:
        density = cloud.radius_search(points[picked], float(density_threshold_radius)).counts()
    finally:
        cloud.free()
    return picked[density > density_threshold_value]
