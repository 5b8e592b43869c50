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
from ..core.geometry import grid_subsampling_many
from ..engine import Cloud, Engine, default_engine

__all__ = ["ShotMultiprocessor", "compute_shot_descriptor", "get_azimuth_idx"]


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

    def _support_cloud(self, point_cloud, normals, voxel, keep=None) -> Cloud:
        """The support the descriptors are computed on: the whole cloud or its voxel-subsampled subset
        (shot_parallelization.py:157-166).  keep: the subset's indices when the caller has them already."""
        if voxel is None:
            return Cloud(self._eng(), point_cloud, normals)
        if keep is None:
            keep = grid_subsampling(np.asarray(point_cloud), voxel, engine=self._eng())
        if self.verbose:
            logging.info(
                f"Keeping a support of {keep.shape[0]} points out of {np.asarray(point_cloud).shape[0]} "
                f"(voxel size: {voxel:.2f})"
            )
        return Cloud(self._eng(), point_cloud, normals, subset=keep)  # (gathered on the device)

    # ---- pieces (names follow the reference's public methods) ----------------------------------------
    def _lists(self, cloud: Cloud, keypoints, neighborhoods, radius):
        """The lists the reference would use: the caller's (`support[neighborhoods[i]]`, shot_parallelization.py:71, 121-122)
        whatever made them -- another radius, KDTree.query, a hand-picked subset.  `neighborhoods=None` (an extension: the
        reference has no default) searches `radius` on the device instead, with no list crossing the host link."""
        if neighborhoods is None:
            return cloud.radius_search(keypoints, radius)
        return cloud.import_neighbors(keypoints, neighborhoods, radius)

    def compute_local_rf(self, keypoints, neighborhoods, support, radius):
        """Local reference frames of `keypoints` over `support[neighborhoods[i]]` with `radius` in the weights (M, 3, 3) --
        shot_parallelization.py:46-84."""
        cloud = Cloud(self._eng(), support)
        try:
            nb = self._lists(cloud, keypoints, neighborhoods, radius)
            try:
                return nb.shot_lrf()
            finally:
                nb.free()
        finally:
            cloud.free()

    def compute_descriptor(self, keypoints, normals, neighborhoods, local_rfs, support, radius):
        """(M, 352) descriptors over `support[neighborhoods[i]]`, `normals[neighborhoods[i]]` -- shot_parallelization.py:86-133."""
        cloud = Cloud(self._eng(), support, normals)
        try:
            nb = self._lists(cloud, keypoints, neighborhoods, radius)
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
        # (the result's own block -- page-locked above 32 MiB -- and every scale's rows copied from the device straight into their
        # place in it: a pageable np.empty + one host copy per scale cost 38 of the 87 ms of a two-radius call on 100 000 keypoints)
        stack = self._eng().host_empty((n_scales, m, 352))
        lrf = None
        # (the supports of all scales at once: their host-side sorts run side by side, core.geometry.grid_subsampling_many)
        keeps = None if voxel_sizes is None else grid_subsampling_many(np.asarray(point_cloud), list(voxel_sizes)[:n_scales], engine=self._eng())
        for s, radius in enumerate(radii):
            cloud = self._support_cloud(point_cloud, normals, None if voxel_sizes is None else voxel_sizes[s],
                                        None if keeps is None else keeps[s])
            try:
                nb = cloud.radius_search(keypoints, radius)
                try:
                    if lrf is None or not self.share_local_rfs:
                        lrf = nb.shot_lrf()
                    nb.shot(lrf, self.normalize, self.min_neighborhood_size, host_out=stack[s])
                    if weights[s] != 1.0:
                        np.multiply(stack[s], weights[s], out=stack[s])
                finally:
                    nb.free()
            finally:
                cloud.free()
        return stack.reshape(m, 352 * n_scales)


def get_azimuth_idx(x: npt.NDArray[np.float64], y: npt.NDArray[np.float64]) -> npt.NDArray[np.int64]:
    """Drop-in for shot.py:51-70: octant index 0..7 of (x, y); a point exactly on a boundary ray falls in the LOWER
    octant.  Evaluated on the GPU by the same device function the SHOT kernel bins its neighbours with."""
    return default_engine().azimuth_idx(x, y)


def compute_shot_descriptor(
    keypoints: npt.NDArray[np.float64],
    cloud_points: npt.NDArray[np.float64],
    normals: npt.NDArray[np.float64],
    radius: float,
    min_neighborhood_size: int = 10,
    n_cosine_bins: int = 11,
    n_azimuth_bins: int = 8,
    n_elevation_bins: int = 2,
    n_radial_bins: int = 2,
    debug_mode: bool = False,
    disable_progress_bars: bool = True,
) -> npt.NDArray[np.float64]:
    """Drop-in for the reference's serial SHOT (shot.py:310-499, "kept for debugging purposes"): not the same function
    as ShotMultiprocessor's -- the frame of a keypoint is computed WITHOUT the keypoint itself (and without any
    duplicate of it) in the support (:361-363), and the rows are always normalised (:496-497).  One K2 search, K4 with
    the zero-distance neighbours left out, K5.  `debug_mode` only adds assertions / warnings in the reference and
    `disable_progress_bars` a tqdm bar; both are accepted and have nothing to act on here."""
    assert n_azimuth_bins == 8, "Generic function for other than 8 azimuth divisions not implemented"
    assert n_elevation_bins == 2, "Generic function for other than 2 elevation divisions not implemented"
    assert n_radial_bins == 2, "Generic function for other than 2 radial divisions not implemented"
    if n_cosine_bins != 11:
        raise NotImplementedError("the device SHOT kernel has the 11 cosine bins of ShotMultiprocessor (352-bin rows)")
    cloud = Cloud(default_engine(), cloud_points, normals)
    try:
        nb = cloud.radius_search(keypoints, radius)
        try:
            return nb.shot_serial(min_neighborhood_size)
        finally:
            nb.free()
    finally:
        cloud.free()
