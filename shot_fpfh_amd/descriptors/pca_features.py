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


def compute_pca_based_basic_features(
    query_points: npt.NDArray[np.float64], cloud_points: npt.NDArray[np.float64], radius: float
) -> tuple[npt.NDArray[np.float64], npt.NDArray[np.float64], npt.NDArray[np.float64], npt.NDArray[np.float64]]:
    """(verticality, linearity, planarity, sphericity), pca_based_descriptors.py:150-187."""
    all_eigenvalues, all_eigenvectors, _ = _local_pca(query_points, cloud_points, radius=radius)
    lbd3, lbd2, lbd1 = all_eigenvalues[:, 0], all_eigenvalues[:, 1], all_eigenvalues[:, 2]
    lbd1 += 1e-6
    normals = all_eigenvectors[:, :, 0]
    verticality = 2 * np.arcsin(np.abs(normals[:, 2])) / np.pi
    linearity = 1 - lbd2 / lbd1
    planarity = (lbd2 - lbd3) / lbd1
    sphericity = lbd3 / lbd1
    return verticality, linearity, planarity, sphericity


def compute_pca_based_features(
    query_points: npt.NDArray[np.float64], cloud_points: npt.NDArray[np.float64], radius: float
) -> npt.NDArray[np.float64]:
    """(N, 21) feature matrix, columns in the reference's order (pca_based_descriptors.py:190-244).  As there,
    lambda_1 is bumped by 1e-6 IN PLACE before the eigenvalue sums are formed."""
    all_eigenvalues, all_eigenvectors, moments, neighborhood_sizes = compute_local_pca_with_moments(
        query_points, cloud_points, radius=radius
    )
    lbd3, lbd2, lbd1 = all_eigenvalues[:, 0], all_eigenvalues[:, 1], all_eigenvalues[:, 2]
    lbd1 += 1e-6

    normals = all_eigenvectors[:, :, 0]
    principal_axis = all_eigenvectors[:, :, 2]

    eigensum = all_eigenvalues.sum(axis=-1)
    eigen_square_sum = (all_eigenvalues**2).sum(axis=-1)
    omnivariance = np.cbrt(all_eigenvalues.prod(axis=-1))
    eigenentropy = (-all_eigenvalues * np.log(all_eigenvalues + 1e-6)).sum(axis=-1)

    linearity = 1 - lbd2 / lbd1
    planarity = (lbd2 - lbd3) / lbd1
    sphericity = lbd3 / lbd1
    curvature_change = lbd3 / eigensum

    verticality = 2 * np.arcsin(np.abs(normals[:, 2])) / np.pi
    lin_verticality = 2 * np.arcsin(np.abs(principal_axis[:, 2])) / np.pi
    horizontalityx = 2 * np.arcsin(np.abs(normals[:, 0])) / np.pi
    horizontalityy = 2 * np.arcsin(np.abs(normals[:, 1])) / np.pi

    return np.hstack(
        (
            eigensum[:, None],
            eigen_square_sum[:, None],
            omnivariance[:, None],
            eigenentropy[:, None],
            linearity[:, None],
            planarity[:, None],
            sphericity[:, None],
            curvature_change[:, None],
            verticality[:, None],
            lin_verticality[:, None],
            horizontalityx[:, None],
            horizontalityy[:, None],
            moments,
            np.array(neighborhood_sizes)[:, None],
        )
    )
