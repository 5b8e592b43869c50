"""PCA normals on the GPU -- drop-in for shot_fpfh.descriptors.compute_normals
(pca_based_descriptors.py:29-59): K2 radius search of the query points + K3 (covariance, LAPACK-
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
    """(M, 3) unit normals.  Only the radius neighbourhood is implemented on the device so far; the
    k-nearest-neighbour branch (`k=`, pca_based_descriptors.py:46) is the next row of the scope table
     and raises NotImplementedError rather than silently running anywhere else."""
    assert k is not None or radius is not None, "No parameter provided for the neighborhood search."
    if k is not None:
        raise NotImplementedError(
            "compute_normals(k=...) (k-NN neighbourhoods) is not on the device path yet; pass radius=..."
        )
    eng = engine or default_engine()
    cloud = Cloud(eng, cloud_points)
    try:
        nbrs = cloud.radius_search(query_points, radius)
        try:
            return nbrs.normals(pre_computed_normals)
        finally:
            nbrs.free()
    finally:
        cloud.free()
