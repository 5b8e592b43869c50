"""Descriptor matching on the GPU -- drop-ins for shot_fpfh.matching.basic_matching and the 2-D
(Euclidean) branch of match_descriptors (matching.py:39-74, 138-169).

The M1 x M2 distance matrix of the reference (scipy cdist, 8 TB at 1M x 1M) is never formed: kernel
K8 returns, per scan descriptor, the first arg-min over the reference descriptors and its distance,
and per reference descriptor the arg-min over the scan side for the reciprocity test.
"""
from __future__ import annotations

import logging
from typing import Callable, Optional

import numpy as np
import numpy.typing as npt

from ..engine import Engine, default_engine

__all__ = ["basic_matching", "match_descriptors", "double_matching_with_rejects"]


def _non_empty_rows(desc: np.ndarray) -> np.ndarray:
    # descriptors left at zero (too sparse a neighbourhood in SHOT) never take part (matching.py:43-44)
    return np.flatnonzero(np.any(desc, axis=1))


def basic_matching(
    scan_descriptors: npt.NDArray[np.float64],
    ref_descriptors: npt.NDArray[np.float64],
    *,
    engine: Optional[Engine] = None,
) -> tuple[npt.NDArray[np.int64], npt.NDArray[np.int64]]:
    """Nearest reference descriptor of every non-empty scan descriptor (matching.py:149-169).
    Returns (scan row indices, matched ref row indices) in the original row numbering."""
    eng = engine or default_engine()
    scan_rows, ref_rows = _non_empty_rows(scan_descriptors), _non_empty_rows(ref_descriptors)
    idx, _, _ = eng.match_argmin(scan_descriptors[scan_rows], ref_descriptors[ref_rows], want_dist=False)
    return scan_rows, ref_rows[idx]


def match_descriptors(
    scan_descriptors: npt.NDArray[np.float64],
    ref_descriptors: npt.NDArray[np.float64],
    filter_callback: Callable[..., npt.NDArray[np.bool_]] | None = None,
    filter_nonreciprocal: bool = False,
    verbose: bool = True,
    n_min_matches: int = 100,
    *,
    engine: Optional[Engine] = None,
    **kwargs,
) -> tuple[npt.NDArray[np.int64], npt.NDArray[np.int64]]:
    """Arg-min matching with an optional distance filter and reciprocity test (matching.py:39-74).

    `filter_callback(distances, **kwargs)` receives the winners' distances and returns a keep-mask.
    With `filter_nonreciprocal`, a match i -> j is also required to satisfy argmin_i' d(i', j) == i,
    but only if at least `n_min_matches` matches survive; otherwise the reciprocity test is dropped.
    A 3-D input (n_scales, n_points, length) selects the "minimum over scales" distance of
    matching.py:77-136: per pair the smallest per-scale Euclidean distance, 1000 where either descriptor is
    empty at a scale; matches whose distance stays at 1000 are dropped.  (The reference's reciprocity masking
    in that branch writes into a temporary copy and has no effect, matching.py:106-108; its only observable
    consequence -- the fallback call when fewer than n_min_matches survive -- is reproduced.)
    """
    eng = engine or default_engine()
    if np.ndim(scan_descriptors) == 3:
        return _match_multiscale(eng, scan_descriptors, ref_descriptors, filter_callback, filter_nonreciprocal, verbose,
                                 n_min_matches, **kwargs)
    if verbose:
        logging.info("")
        logging.info("-- Matching descriptors based on Euclidian-norm proximity --")
    scan_rows, ref_rows = _non_empty_rows(scan_descriptors), _non_empty_rows(ref_descriptors)
    idx, dist, col = eng.match_argmin(
        scan_descriptors[scan_rows], ref_descriptors[ref_rows], want_dist=True, want_col=filter_nonreciprocal
    )
    keep = filter_callback(dist, **kwargs) if filter_callback is not None else np.ones(dist.shape[0], dtype=bool)
    if filter_nonreciprocal:
        both = keep & (col[idx] == np.arange(idx.shape[0]))
        if both.sum() >= n_min_matches:
            keep = both
        elif verbose:
            logging.warning("Too few reciprocal matches, keeping non-reciprocal matches.")
    if verbose:
        logging.info(f"Kept {keep.sum()} matches out of {scan_descriptors.shape[-2]} descriptors.")
    return scan_rows[keep], ref_rows[idx[keep]]


def _match_multiscale(eng, scan, ref, filter_callback, filter_nonreciprocal, verbose, n_min_matches, **kwargs):
    max_val = 1000
    if verbose:
        logging.info("")
        logging.info("-- Matching descriptors based on infinite-norm proximity --")
    idx, dist = eng.match_argmin_multiscale(scan, ref, max_val)
    keep = filter_callback(dist, **kwargs) if filter_callback is not None else np.ones(dist.shape[0], dtype=bool)
    keep = keep & (dist < max_val)
    if keep.sum() < n_min_matches and filter_nonreciprocal:
        logging.warning("Too few reciprocal matches, keeping non-reciprocal matches.")
        # as the reference does: same call without the reciprocity flag (and with the default n_min_matches)
        return match_descriptors(scan, ref, filter_callback, filter_nonreciprocal=False, verbose=verbose, engine=eng, **kwargs)
    if verbose:
        logging.info(f"Kept {keep.sum()} matches out of {scan.shape[-2]} descriptors.")
    return np.arange(scan.shape[1])[keep], np.arange(ref.shape[1])[idx[keep]]


def double_matching_with_rejects(scan_descriptors, ref_descriptors, threshold, verbose=True):
    """Present in the reference's export list but broken there (matching.py:172-221): whatever it is handed, it
    ends in an exception -- `distance_matrix[np.arange(S), indices]` (:202) does not broadcast unless as many ref rows are
    non-empty as there are scan rows, and the last line (:220) indexes with FLOAT distances.  A drop-in raises what the
    reference raises, from the shapes alone (no distance matrix is formed): the table of
    tests/golden/double_matching_errors.json, generated by calling the reference, is held in tests/test_host_logic.py."""
    scan_descriptors, ref_descriptors = np.asarray(scan_descriptors), np.asarray(ref_descriptors)
    s = scan_descriptors.shape[0]
    s1 = int(np.any(scan_descriptors, axis=1).sum())
    r1 = int(np.any(ref_descriptors, axis=1).sum())
    if 0 < r1 < 3:  # np.argpartition(distance_matrix, kth=2, axis=1), :201 (an empty axis passes)
        raise ValueError(f"kth(=2) out of bounds ({r1})")
    try:  # distance_matrix[np.arange(S), indices] with indices of shape (S', R'), :202
        shape = np.broadcast_shapes((s,), (s1, r1))
    except ValueError:
        raise IndexError(f"shape mismatch: indexing arrays could not be broadcast together with shapes ({s},) ({s1},{r1}) ") from None
    if int(np.prod(shape)) > 0 and s > s1:
        raise IndexError(f"index {s1} is out of bounds for axis 0 with size {s1}")
    if shape[0] != s:  # np.divide(..., out=np.ones(S)), :204-209
        raise ValueError(f"non-broadcastable output operand with shape ({s},) doesn't match the broadcast shape ({shape[0]},)")
    raise IndexError("arrays used as indices must be of integer (or boolean) type")  # :220
