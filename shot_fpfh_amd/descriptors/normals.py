"""PCA normals on the GPU -- drop-in for shot_fpfh.descriptors.compute_normals
(pca_based_descriptors.py:29-59): K2 radius or k-nearest-neighbour search of the query points + K3 (covariance, LAPACK-
compatible 3x3 eigensolver, smallest-eigenvalue eigenvector, optional re-orientation).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import numpy.typing as npt

from ..engine import Cloud, Engine, default_engine

__all__ = ["compute_normals"]


def compute_normals(
    query_points: npt.NDArray[np.float64],
    cloud_points: npt.NDArray[np.float64],
    *,
    k: int | None = None,
    radius: float | None = None,
    pre_computed_normals: npt.NDArray[np.float64] | None = None,
    engine: Optional[Engine] = None,
) -> npt.NDArray[np.float64]:
    """(M, 3) unit normals from the k nearest neighbours (`k=`, KDTree.query, pca_based_descriptors.py:46 --
    the reference's precedence when both are given) or from a radius neighbourhood (`radius=`, :48)."""
    assert k is not None or radius is not None, "No parameter provided for the neighborhood search."
    eng = engine or default_engine()
    cloud = Cloud(eng, cloud_points)
    try:
        if k is None:  # the radius branch (:48): one sweep, no lists (sf_normals_radius)
            return cloud.normals_radius(query_points, radius, pre_computed_normals)
        nbrs = cloud.knn_search(query_points, k)
        try:
            return nbrs.normals(pre_computed_normals)
        finally:
            nbrs.free()
    finally:
        cloud.free()
