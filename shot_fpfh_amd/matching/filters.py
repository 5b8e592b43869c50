"""Distance-based match filters (host side, O(M) NumPy on the winners' distance vector).

These are the callbacks `match_descriptors(..., filter_callback=...)` accepts; the reference keeps
them on the host too (shot_fpfh/matching/filters.py:12-40) and users may pass their own callable,
so they are not kernels.  Each returns a boolean keep-mask over `distances`.
"""
from __future__ import annotations

from typing import Any, Protocol

import numpy as np
import numpy.typing as npt

__all__ = ["FilterFunction", "threshold_filter", "quantile_filter", "left_median_filter"]


class FilterFunction(Protocol):
    """Signature of a filter callback (reference filters.py:12-16)."""

    def __call__(self, distances: npt.NDArray[np.float64], *args: Any, **kwargs: Any) -> npt.NDArray[np.bool_]: ...


def threshold_filter(distances: npt.NDArray[np.float64], threshold_multiplier: float) -> npt.NDArray[np.bool_]:
    """Keep matches no farther than `threshold_multiplier` x the smallest NON-ZERO distance
    (reference filters.py:19-23)."""
    smallest_positive = distances[np.flatnonzero(distances)].min()
    return distances <= smallest_positive * threshold_multiplier


def quantile_filter(distances: npt.NDArray[np.float64], quantiles: tuple[float, float]) -> npt.NDArray[np.bool_]:
    """Keep matches whose distance lies between the two given quantiles, bounds included
    (reference filters.py:26-31)."""
    low, high = np.quantile(distances, quantiles)
    return (low <= distances) & (distances <= high)


def left_median_filter(distances: npt.NDArray[np.float64]) -> npt.NDArray[np.bool_]:
    """Keep the upper half of the below-median matches (reference filters.py:34-40).

    The lower bound is written there as (median + distances.nonzero()[0].min()) / 2, i.e. it
    averages the median with the smallest INDEX of a non-zero distance, not the smallest non-zero
    distance; that observable behaviour is kept.
    """
    median = np.median(distances)
    first_nonzero_position = np.flatnonzero(distances).min()
    lower = (median + first_nonzero_position) / 2
    return (distances <= median) & (distances >= lower)
