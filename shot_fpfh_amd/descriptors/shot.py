"""SHOT on the GPU -- drop-in for shot_fpfh.descriptors.ShotMultiprocessor
(shot_parallelization.py:16-312).

The reference forks `n_procs` workers and pickles every keypoint's neighbourhood to them; here one
device launch per stage covers all keypoints (K2 search -> K4 local reference frames -> K5 352-bin
descriptor), so the context manager only pins the GPU engine.  No process is forked -- a HIP context
must not be used across fork() -- and `n_procs` is kept as an inert field for signature parity.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass, field
from types import TracebackType
from typing import Optional, Sequence

import numpy as np
import numpy.typing as npt

from ..core import grid_subsampling
from ..engine import Cloud, Engine, default_engine

__all__ = ["ShotMultiprocessor"]


@dataclass
class ShotMultiprocessor:
    """Fields and defaults as in shot_parallelization.py:22-28."""

    normalize: bool = True
    share_local_rfs: bool = True
    min_neighborhood_size: int = 100

    n_procs: int = 8  # unused: the GPU replaces the worker pool
    disable_progress_bar: bool = False
    verbose: bool = True

    engine: Optional[Engine] = field(default=None, repr=False, compare=False)

    def __enter__(self) -> "ShotMultiprocessor":
        self._engine = self.engine or default_engine()
        return self

    def __exit__(self, exc_type: type | None, exc_val: Exception | None, exc_tb: TracebackType | None) -> None:
        self._engine = None

    # ------------------------------------------------------------------------------------------------
    def _eng(self) -> Engine:
        eng = getattr(self, "_engine", None)
        if eng is None:  # the reference's methods are only valid inside `with` (the pool lives there)
            raise AttributeError("ShotMultiprocessor must be used as a context manager (`with ShotMultiprocessor(...) as sm:`)")
        return eng

    def _support_cloud(self, point_cloud, normals, voxel) -> Cloud:
        """The support the descriptors are computed on: the whole cloud or its voxel-subsampled subset
        (shot_parallelization.py:157-166)."""
        if voxel is None:
            return Cloud(self._eng(), point_cloud, normals)
        keep = grid_subsampling(np.asarray(point_cloud), voxel)
        if self.verbose:
            logging.info(
                f"Keeping a support of {keep.shape[0]} points out of {np.asarray(point_cloud).shape[0]} "
                f"(voxel size: {voxel:.2f})"
            )
        return Cloud(self._eng(), np.asarray(point_cloud)[keep], np.asarray(normals)[keep])

    # ---- pieces (names follow the reference's public methods) ----------------------------------------
    def compute_local_rf(self, keypoints, neighborhoods, support, radius):
        """Local reference frames of `keypoints` over `support` within `radius` (M, 3, 3).
        `neighborhoods` is accepted for signature parity; the lists are rebuilt on the device."""
        cloud = Cloud(self._eng(), support)
        try:
            nb = cloud.radius_search(keypoints, radius)
            try:
                return nb.shot_lrf()
            finally:
                nb.free()
        finally:
            cloud.free()

    def compute_descriptor(self, keypoints, normals, neighborhoods, local_rfs, support, radius):
        cloud = Cloud(self._eng(), support, normals)
        try:
            nb = cloud.radius_search(keypoints, radius)
            try:
                return nb.shot(local_rfs, self.normalize, self.min_neighborhood_size)
            finally:
                nb.free()
        finally:
            cloud.free()

    # ---- drivers ---------------------------------------------------------------------------------------
    def compute_descriptor_single_scale(
        self,
        point_cloud: npt.NDArray[np.float64],
        normals: npt.NDArray[np.float64],
        keypoints: npt.NDArray[np.float64],
        radius: float,
        subsampling_voxel_size: float | None = None,
    ) -> npt.NDArray[np.float64]:
        """(M, 352) descriptors; one search shared by the frames and the descriptor
        (shot_parallelization.py:135-183)."""
        cloud = self._support_cloud(point_cloud, normals, subsampling_voxel_size)
        try:
            nb = cloud.radius_search(keypoints, radius)
            try:
                return nb.shot_single_scale(self.normalize, self.min_neighborhood_size)
            finally:
                nb.free()
        finally:
            cloud.free()

    def compute_descriptor_bi_scale(
        self,
        point_cloud: npt.NDArray[np.float64],
        normals: npt.NDArray[np.float64],
        keypoints: npt.NDArray[np.float64],
        local_rf_radius: float,
        shot_radius: float,
        subsampling_voxel_size: float | None = None,
    ) -> npt.NDArray[np.float64]:
        """Frames at `local_rf_radius`, descriptor at `shot_radius` (shot_parallelization.py:185-239).
        The reference indexes `point_cloud[None]` at :229 when no voxel size is given and crashes; the
        same call is rejected here."""
        if subsampling_voxel_size is None:
            raise IndexError("compute_descriptor_bi_scale needs subsampling_voxel_size (the reference fails without it)")
        cloud = self._support_cloud(point_cloud, normals, subsampling_voxel_size)
        try:
            nb_rf = cloud.radius_search(keypoints, local_rf_radius)
            try:
                lrf = nb_rf.shot_lrf()
            finally:
                nb_rf.free()
            nb = cloud.radius_search(keypoints, shot_radius)
            try:
                return nb.shot(lrf, self.normalize, self.min_neighborhood_size)
            finally:
                nb.free()
        finally:
            cloud.free()

    def compute_descriptor_multiscale(
        self,
        point_cloud: npt.NDArray[np.float64],
        normals: npt.NDArray[np.float64],
        keypoints: npt.NDArray[np.float64],
        radii: Sequence[float] | npt.NDArray[np.float64],
        voxel_sizes: Sequence[float] | npt.NDArray[np.float64] | None = None,
        weights: Sequence[float] | npt.NDArray[np.float64] | None = None,
    ) -> npt.NDArray[np.float64]:
        """One descriptor per radius, scaled by `weights`; frames come from the FIRST radius when
        `share_local_rfs` (shot_parallelization.py:241-312).  The result is the (scales, M, 352) stack
        RESHAPED to (M, 352*scales) exactly as the reference does at :312 (rows interleave scales; it
        is not a per-keypoint concatenation)."""
        n_scales = len(radii)
        if weights is None:
            weights = np.ones(n_scales)
        m = np.asarray(keypoints).shape[0]
        stack = np.zeros((n_scales, m, 352))
        lrf = None
        for s, radius in enumerate(radii):
            cloud = self._support_cloud(point_cloud, normals, None if voxel_sizes is None else voxel_sizes[s])
            try:
                nb = cloud.radius_search(keypoints, radius)
                try:
                    if lrf is None or not self.share_local_rfs:
                        lrf = nb.shot_lrf()
                    stack[s] = nb.shot(lrf, self.normalize, self.min_neighborhood_size) * weights[s]
                finally:
                    nb.free()
            finally:
                cloud.free()
        return stack.reshape(m, 352 * n_scales)
