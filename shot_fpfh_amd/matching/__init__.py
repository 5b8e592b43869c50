"""GPU drop-ins for shot_fpfh.matching (reference matching/__init__.py:1-19)."""
from .filters import FilterFunction, left_median_filter, quantile_filter, threshold_filter
from .match import basic_matching, double_matching_with_rejects, match_descriptors
from .ransac import ransac_on_matches

__all__ = [
    "FilterFunction",
    "threshold_filter",
    "quantile_filter",
    "left_median_filter",
    "match_descriptors",
    "basic_matching",
    "double_matching_with_rejects",
    "ransac_on_matches",
]
