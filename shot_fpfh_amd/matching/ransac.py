"""RANSAC over descriptor matches -- drop-in for shot_fpfh.matching.ransac_on_matches
(ransac.py:17-82).

Host: the draws (same module-level np.random.default_rng(seed=72) stream as ransac.py:14, so draw
k of a process equals the reference's draw k) and one Kabsch fit per draw -- solved as one stack through
the same BLAS / LAPACK routines (bit-identical transforms, checked against the per-draw solver on the
a sample of the draws and every reflected draw of every call).  Device (K9): the O(n_draws x n_matches) inlier count of every candidate
transform in ONE launch, instead of one NumPy pass over all matches per draw.
"""
from __future__ import annotations

import logging
from typing import Optional

import numpy as np
import numpy.typing as npt

from ..core import RigidTransform, solver_point_to_point
from ..core.geometry import solver_point_to_point_batched
from ..engine import Engine, default_engine

__all__ = ["ransac_on_matches", "rng"]

# same seed and same persistence across calls as the reference module's generator
rng = np.random.default_rng(seed=72)


def ransac_on_matches(
    scan_descriptors_indices: npt.NDArray[np.integer],
    ref_descriptors_indices: npt.NDArray[np.integer],
    scan_keypoints: npt.NDArray[np.float64],
    ref_keypoints: npt.NDArray[np.float64],
    n_draws: int = 10000,
    draw_size: int = 4,
    distance_threshold: float = 1,
    verbose: bool = False,
    disable_progress_bar: bool = False,
    *,
    engine: Optional[Engine] = None,
) -> tuple[float, RigidTransform]:
    """Returns (inlier ratio of the best draw, its RigidTransform with a re-normalised rotation).

    The best draw is the FIRST one reaching the maximal inlier count (strict `>` at ransac.py:68).
    """
    eng = engine or default_engine()
    n_matches = scan_descriptors_indices.shape[0]
    scan_pts = np.ascontiguousarray(scan_keypoints[scan_descriptors_indices], dtype=np.float64)
    ref_pts = np.ascontiguousarray(ref_keypoints[ref_descriptors_indices], dtype=np.float64)

    draws = np.empty((n_draws, draw_size), dtype=np.int64)
    for d in range(n_draws):  # one generator call per draw, as ransac.py:50-55 (keeps the stream aligned)
        draws[d] = rng.choice(n_matches, draw_size, replace=False, shuffle=False)
    records = np.empty((n_draws, 12), dtype=np.float64)
    if n_draws:
        rot, tr, reflected = solver_point_to_point_batched(scan_pts[draws], ref_pts[draws], return_reflected=True)
        records[:, :9] = rot.reshape(n_draws, 9)
        records[:, 9:] = tr
        # the stacked solve must reproduce the per-draw one bit for bit: checked on the first draws, on an evenly
        # spread sample and on the draws that took the reflection branch (det < 0: rare, and the likeliest to differ)
        check = set(range(min(n_draws, 8))) | set(np.linspace(0, n_draws - 1, min(n_draws, 56)).astype(int).tolist())
        check |= set(reflected[:256].tolist())
        for d in sorted(check):
            if not np.array_equal(records[d], solver_point_to_point(scan_pts[draws[d]], ref_pts[draws[d]]).as_row12()):
                logging.warning("stacked Kabsch differs from the per-draw solver on this NumPy build: using the per-draw loop")
                for e in range(n_draws):
                    records[e] = solver_point_to_point(scan_pts[draws[e]], ref_pts[draws[e]]).as_row12()
                break

    inliers = eng.ransac_score(scan_pts, ref_pts, records, distance_threshold)
    best = int(np.argmax(inliers))  # first maximum == reference's strict-greater update rule
    if verbose:
        logging.info(f"Best draw {best}: {int(inliers[best])} inliers out of {n_matches}")
    best_transform = RigidTransform(records[best, :9].reshape(3, 3).copy(), records[best, 9:].copy())
    best_transform.normalize_rotation()
    return inliers[best] / n_matches, best_transform
