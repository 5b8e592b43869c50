"""GPU drop-ins for shot_fpfh.descriptors (reference descriptors/__init__.py:1-17).

Hot-path exports only: compute_fpfh_descriptor, compute_normals, ShotMultiprocessor.  The PCA
feature helpers of the reference (compute_sphericity, compute_pca_based_*) are outside the scope
table and are not provided.
"""
from .fpfh import compute_fpfh_descriptor
from .normals import compute_normals
from .shot import ShotMultiprocessor

__all__ = ["compute_fpfh_descriptor", "compute_normals", "ShotMultiprocessor"]
