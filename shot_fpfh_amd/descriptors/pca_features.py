"""PCA-based neighbourhood features on the GPU -- drop-ins for shot_fpfh.descriptors.compute_sphericity,
compute_local_pca_with_moments, compute_pca_based_basic_features and compute_pca_based_features
(pca_based_descriptors.py:60-244).

The per-point loops of the reference (neighbour search, barycentre, covariance, np.linalg.eigh, moments) run in
kernels K2 + K3 (sf_pca); the closed-form feature expressions on top of the eigen-decomposition are evaluated
here exactly as the reference writes them.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import numpy.typing as npt

from ..engine import Cloud, Engine, default_engine

__all__ = [
    "compute_sphericity",
    "compute_local_pca_with_moments",
    "compute_pca_based_basic_features",
    "compute_pca_based_features",
]


def _local_pca(query_points, cloud_points, *, radius=None, k=None, moments=False, engine: Optional[Engine] = None):
    eng = engine or default_engine()
    cloud = Cloud(eng, cloud_points)
    try:
        nbrs = cloud.knn_search(query_points, k) if k is not None else cloud.radius_search(query_points, radius)
        try:
            sizes = nbrs.counts()
            return nbrs.pca(moments=moments) + (sizes,)
        finally:
            nbrs.free()
    finally:
        cloud.free()


def compute_sphericity(
    query_points: npt.NDArray[np.float64], cloud_points: npt.NDArray[np.float64], radius: float
) -> npt.NDArray[np.float64]:
    """lambda_3 / (lambda_1 + 1e-6) per query point (pca_based_descriptors.py:60-73)."""
    eigenvalues = _local_pca(query_points, cloud_points, radius=radius)[0]
    return eigenvalues[:, 0] / (eigenvalues[:, 2] + 1e-6)


def compute_local_pca_with_moments(
    query_points: npt.NDArray[np.float64],
    cloud_points: npt.NDArray[np.float64],
    nghbrd_search: str = "spherical",
    radius: float | None = None,
    k: int | None = None,
    verbose: bool = False,
) -> tuple[npt.NDArray[np.float64], npt.NDArray[np.float64], npt.NDArray[np.float64], list[int]]:
    """(eigenvalues (N,3), eigenvectors (N,3,3), moments (N,8), neighbourhood sizes), pca_based_descriptors.py:75-146.
    `verbose` only drove a matplotlib histogram in the reference and is inert here."""
    mode = nghbrd_search.lower()
    if mode == "spherical":
        w, v, mo, sizes = _local_pca(query_points, cloud_points, radius=radius, moments=True)
    elif mode == "knn":
        w, v, mo, sizes = _local_pca(query_points, cloud_points, k=k, moments=True)
    else:
        raise TypeError("nghbrd_search must be 'spherical' or 'knn'")  # the reference fails on neighborhoods=None
    return w, v, mo, [int(n) for n in sizes]


class _Spectrum:
    """Eigen-decomposition of every neighbourhood in the form the feature formulas consume.  The reference bumps the
    LARGEST eigenvalue by 1e-6 in place before anything else reads the eigenvalue array (pca_based_descriptors.py:172,
    207), so the sums, the product and the entropy below all see the bumped value; that is reproduced by bumping a copy
    once, here."""

    def __init__(self, eigenvalues: np.ndarray, eigenvectors: np.ndarray):
        self.values = np.array(eigenvalues, dtype=np.float64, copy=True)
        self.values[:, 2] += 1e-6
        self.smallest, self.middle, self.largest = self.values[:, 0], self.values[:, 1], self.values[:, 2]
        self.normal = eigenvectors[:, :, 0]  # eigenvector of the smallest eigenvalue
        self.axis = eigenvectors[:, :, 2]    # ... of the largest


def _tilt(component: np.ndarray) -> np.ndarray:
    """|component| of a unit vector as an angle fraction of a right angle."""
    return 2 * np.arcsin(np.abs(component)) / np.pi


# name -> formula; the 12 leading columns of compute_pca_based_features in the reference's order
# (pca_based_descriptors.py:222-243), the four of compute_pca_based_basic_features picked by name (:174-179)
_SHAPE_FEATURES = {
    "eigensum": lambda s: s.values.sum(axis=-1),
    "eigen_square_sum": lambda s: (s.values**2).sum(axis=-1),
    "omnivariance": lambda s: np.cbrt(s.values.prod(axis=-1)),
    "eigenentropy": lambda s: (-s.values * np.log(s.values + 1e-6)).sum(axis=-1),
    "linearity": lambda s: 1 - s.middle / s.largest,
    "planarity": lambda s: (s.middle - s.smallest) / s.largest,
    "sphericity": lambda s: s.smallest / s.largest,
    "curvature_change": lambda s: s.smallest / s.values.sum(axis=-1),
    "verticality": lambda s: _tilt(s.normal[:, 2]),
    "lin_verticality": lambda s: _tilt(s.axis[:, 2]),
    "horizontalityx": lambda s: _tilt(s.normal[:, 0]),
    "horizontalityy": lambda s: _tilt(s.normal[:, 1]),
}


def compute_pca_based_basic_features(
    query_points: npt.NDArray[np.float64], cloud_points: npt.NDArray[np.float64], radius: float
) -> tuple[npt.NDArray[np.float64], npt.NDArray[np.float64], npt.NDArray[np.float64], npt.NDArray[np.float64]]:
    """(verticality, linearity, planarity, sphericity), pca_based_descriptors.py:150-187."""
    eigenvalues, eigenvectors, _ = _local_pca(query_points, cloud_points, radius=radius)
    spectrum = _Spectrum(eigenvalues, eigenvectors)
    return tuple(_SHAPE_FEATURES[name](spectrum) for name in ("verticality", "linearity", "planarity", "sphericity"))


def compute_pca_based_features(
    query_points: npt.NDArray[np.float64], cloud_points: npt.NDArray[np.float64], radius: float
) -> npt.NDArray[np.float64]:
    """(N, 21) feature matrix (pca_based_descriptors.py:187-244): the twelve shape features above, the eight moments of
    compute_local_pca_with_moments, the neighbourhood size."""
    eigenvalues, eigenvectors, moments, sizes = compute_local_pca_with_moments(query_points, cloud_points, radius=radius)
    spectrum = _Spectrum(eigenvalues, eigenvectors)
    out = np.empty((eigenvalues.shape[0], len(_SHAPE_FEATURES) + moments.shape[1] + 1))
    for col, formula in enumerate(_SHAPE_FEATURES.values()):
        out[:, col] = formula(spectrum)
    out[:, len(_SHAPE_FEATURES):-1] = moments
    out[:, -1] = sizes
    return out
