"""FPFH on the GPU -- drop-in for shot_fpfh.descriptors.compute_fpfh_descriptor (fpfh.py:16-117).

Pipeline (all on device): K1 grid build -> K2 radius search of EVERY cloud point (the reference
computes SPFH for the whole cloud, not just around keypoints, fpfh.py:28-43) -> K6 SPFH integer
histograms -> K7 weighted reduction for the keypoints.
"""
from __future__ import annotations

import logging
from typing import Optional

import numpy as np
import numpy.typing as npt

from .. import _ffi
from ..engine import Cloud, Engine, Spfh, default_engine

__all__ = ["compute_fpfh_descriptor"]


def compute_fpfh_descriptor(
    keypoints_indices: npt.NDArray[np.integer],
    cloud_points: npt.NDArray[np.float64],
    normals: npt.NDArray[np.float64],
    radius: float,
    n_bins: int,
    decorrelated: bool = False,
    verbose: bool = True,
    disable_progress_bars: bool = True,
    *,
    engine: Optional[Engine] = None,
    return_spfh: bool = False,
) -> npt.NDArray[np.float64]:
    """Same positional signature and result as the reference: an (M, n_bins**3) float64 array.

    `decorrelated=True` raises ValueError in the reference (shape mismatch at fpfh.py:59) and is
    rejected here as well.  `disable_progress_bars` is accepted for compatibility (there is no
    per-point host loop to report on).  Keyword-only extras: `engine` picks the GPU context,
    `return_spfh` additionally returns the (N, n_bins**3) SPFH table the reference keeps local.
    """
    if decorrelated:
        raise ValueError("decorrelated=True is not supported (the reference implementation raises for it too)")
    if not 1 <= int(n_bins) <= _ffi.MAX_FPFH_BINS:
        raise ValueError(f"n_bins must be in 1..{_ffi.MAX_FPFH_BINS} on the device path, got {n_bins}")
    eng = engine or default_engine()
    kp = np.asarray(keypoints_indices)
    if kp.dtype == bool:  # NumPy boolean-mask indexing
        kp = np.flatnonzero(kp)
    kp = np.ascontiguousarray(kp, dtype=np.int64)
    cloud = Cloud(eng, cloud_points, normals)
    try:
        if cloud.n == 0:
            out = np.zeros((kp.shape[0], int(n_bins) ** 3))
            return (out, np.zeros((0, int(n_bins) ** 3))) if return_spfh else out
        nbrs = cloud.radius_search_self(radius)
        try:
            spfh = Spfh(cloud, n_bins, nbrs.max_count, radius)
            try:
                spfh.compute(nbrs)
                if verbose:
                    logging.info(
                        f"Mean neighborhood size over the whole point cloud: {nbrs.total / max(cloud.n, 1):.2f}"
                    )
                out = spfh.fpfh(nbrs, kp)
                if return_spfh:
                    return out, spfh.export()
                return out
            finally:
                spfh.free()
        finally:
            nbrs.free()
    finally:
        cloud.free()
