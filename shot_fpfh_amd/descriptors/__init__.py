"""GPU drop-ins for shot_fpfh.descriptors (reference descriptors/__init__.py:1-17): the three hot-path
exports (compute_fpfh_descriptor, compute_normals, ShotMultiprocessor) and the PCA feature helpers that
share kernel K3 with the normals."""
from .fpfh import compute_fpfh_descriptor
from .normals import compute_normals
from .pca_features import (
    compute_local_pca_with_moments,
    compute_pca_based_basic_features,
    compute_pca_based_features,
    compute_sphericity,
)
from .shot import ShotMultiprocessor

__all__ = [
    "compute_fpfh_descriptor",
    "compute_normals",
    "compute_pca_based_basic_features",
    "compute_pca_based_features",
    "compute_sphericity",
    "compute_local_pca_with_moments",
    "ShotMultiprocessor",
]
