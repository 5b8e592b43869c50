#!/usr/bin/env python3
"""Record the call signatures of the reference's public functions on and around the hot path as a data fixture
(tests/golden/api_signatures.json): name -> [[parameter, kind, default-repr], ...].  Runs only in the build
container (imports /root/reference); tests/test_host_logic.py holds shot_fpfh_amd to the recorded list.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tools/gen_api_signatures.py
"""
import inspect
import json
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

TARGETS = {
    "shot_fpfh.descriptors": ["compute_fpfh_descriptor", "compute_normals", "compute_sphericity",
                              "compute_pca_based_basic_features", "compute_pca_based_features",
                              "ShotMultiprocessor.compute_descriptor_single_scale", "ShotMultiprocessor.compute_descriptor_bi_scale",
                              "ShotMultiprocessor.compute_descriptor_multiscale", "ShotMultiprocessor.compute_local_rf",
                              "ShotMultiprocessor.compute_descriptor"],
    "shot_fpfh.descriptors.pca_based_descriptors": ["compute_local_pca_with_moments"],
    "shot_fpfh.descriptors.shot": ["compute_shot_descriptor"],
    "shot_fpfh.matching": ["match_descriptors", "basic_matching", "double_matching_with_rejects", "ransac_on_matches",
                           "threshold_filter", "quantile_filter", "left_median_filter"],
    "shot_fpfh.keypoint_selection": ["select_keypoints_iteratively", "select_keypoints_subsampling", "select_keypoints_randomly",
                                     "select_query_indices_randomly", "select_keypoints_with_density_threshold"],
    "shot_fpfh.icp": ["icp_point_to_point_with_sampling", "icp_point_to_point", "icp_point_to_plane"],
    "shot_fpfh.core": ["solver_point_to_point", "solver_point_to_plane", "grid_subsampling", "compute_point_to_point_error"],
    "shot_fpfh.helpers.io_ply": ["read_ply", "write_ply", "get_data"],
    "shot_fpfh.pipeline": ["RegistrationPipeline." + m for m in (
        "select_keypoints", "compute_shot_descriptor_single_scale", "compute_shot_descriptor_bi_scale",
        "compute_shot_descriptor_multiscale", "compute_descriptors", "find_descriptors_matches", "run_ransac", "run_icp",
        "compute_metrics_post_icp", "write_alignments")],
}


def describe(fn):
    fn = inspect.unwrap(fn)
    return [[name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)]
            for name, p in inspect.signature(fn).parameters.items()]


def main():
    import importlib

    out = {}
    for module, names in TARGETS.items():
        mod = importlib.import_module(module)
        for name in names:
            obj = mod
            for part in name.split("."):
                obj = getattr(obj, part)
            out[f"{module}:{name}"] = describe(obj)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "api_signatures.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"{len(out)} signatures -> {path}")


if __name__ == "__main__":
    main()
