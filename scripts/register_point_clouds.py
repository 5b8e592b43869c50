#!/usr/bin/env python3
"""Register two binary-PLY point clouds end to end on one MI355X (the counterpart of the reference's
scripts/register_point_clouds.py, with plain argparse flags instead of its YAML configuration layer).

    python scripts/register_point_clouds.py scan.ply ref.ply --radius 0.05 --descriptor shot_single_scale \
        --keypoints subsampling --keypoint-size 0.03 --ransac-threshold 0.01 --icp point_to_plane --icp-dmax 0.05

Stages: read PLY (+ PCA normals, k nearest neighbours) -> keypoints -> SHOT / FPFH descriptors -> matching ->
RANSAC -> ICP -> overlap metrics; optionally writes the aligned pair as PLY.
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from shot_fpfh_amd import compute_normals  # noqa: E402
from shot_fpfh_amd.helpers import get_data  # noqa: E402
from shot_fpfh_amd.pipeline import RegistrationPipeline  # noqa: E402


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    p.add_argument("scan"), p.add_argument("ref")
    p.add_argument("--normals-k", type=int, default=30, help="neighbours for the PCA normals (reference default 30)")
    p.add_argument("--keep-stored-normals", action="store_true", help="use the file's normals instead of recomputing them")
    p.add_argument("--keypoints", default="subsampling", choices=["random", "iterative", "subsampling", "subsampling_with_density"])
    p.add_argument("--keypoint-size", type=float, default=None, help="sphere / voxel size of the keypoint selection")
    p.add_argument("--min-n-neighbors", type=int, default=None)
    p.add_argument("--proportion", type=float, default=0.5, help="share of points kept by --keypoints random")
    p.add_argument("--descriptor", default="shot_single_scale", choices=["fpfh", "shot_single_scale", "shot_bi_scale", "shot_multiscale"])
    p.add_argument("--radius", type=float, required=True)
    p.add_argument("--fpfh-bins", type=int, default=5)
    p.add_argument("--phi", type=float, default=3.0), p.add_argument("--rho", type=float, default=10.0)
    p.add_argument("--n-scales", type=int, default=2)
    p.add_argument("--no-support-subsampling", action="store_true")
    p.add_argument("--min-neighborhood-size", type=int, default=100)
    p.add_argument("--matching", default="simple", choices=["simple", "double", "threshold"])
    p.add_argument("--reject-threshold", type=float, default=0.8), p.add_argument("--threshold-multiplier", type=float, default=10)
    p.add_argument("--ransac-draws", type=int, default=10000), p.add_argument("--ransac-draw-size", type=int, default=4)
    p.add_argument("--ransac-threshold", type=float, default=1.0)
    p.add_argument("--icp", default="point_to_plane", choices=["point_to_point", "point_to_plane", "none"])
    p.add_argument("--icp-dmax", type=float, default=0.5), p.add_argument("--icp-voxel", type=float, default=0.2)
    p.add_argument("--icp-max-iter", type=int, default=50), p.add_argument("--icp-rms", type=float, default=1e-3)
    p.add_argument("--metric-threshold", type=float, default=0.1)
    p.add_argument("--write", default=None, help="basename for the aligned clouds (<name>_ransac.ply, <name>_icp.ply)")
    return p.parse_args(argv)


def main(argv=None) -> int:
    args = parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    t0 = time.perf_counter()
    load = dict(recompute_normals=not args.keep_stored_normals, k=args.normals_k, normals_computation_callback=compute_normals)
    scan, scan_normals = get_data(args.scan, **load)
    ref, ref_normals = get_data(args.ref, **load)
    pipe = RegistrationPipeline(scan=scan, scan_normals=scan_normals, ref=ref, ref_normals=ref_normals)
    pipe.select_keypoints(args.keypoints, neighborhood_size=args.keypoint_size, min_n_neighbors=args.min_n_neighbors,
                          proportion_picked=args.proportion)
    pipe.compute_descriptors(radius=args.radius, descriptor_choice=args.descriptor, fpfh_n_bins=args.fpfh_bins, phi=args.phi,
                             rho=args.rho, n_scales=args.n_scales, subsample_support=not args.no_support_subsampling,
                             min_neighborhood_size=args.min_neighborhood_size, disable_progress_bars=True, verbose=False)
    pipe.find_descriptors_matches(args.matching, reject_threshold=args.reject_threshold,
                                  threshold_multiplier=args.threshold_multiplier)
    logging.info(f"{pipe.matches[0].shape[0]} matches")
    transformation, inliers_ratio = pipe.run_ransac(n_draws=args.ransac_draws, draw_size=args.ransac_draw_size,
                                                    max_inliers_distance=args.ransac_threshold, disable_progress_bar=True)
    logging.info(f"RANSAC inlier ratio {inliers_ratio:.3f}\n{transformation}")
    outputs = [(f"{args.write}_ransac.ply", transformation)] if args.write else []
    if args.icp != "none":
        transformation, rms, converged = pipe.run_icp(args.icp, transformation, d_max=args.icp_dmax, voxel_size=args.icp_voxel,
                                                       max_iter=args.icp_max_iter, rms_threshold=args.icp_rms,
                                                       disable_progress_bar=True)
        logging.info(f"ICP rms {rms:.3e}, converged: {bool(converged)}\n{transformation}")
        if args.write:
            outputs.append((f"{args.write}_icp.ply", transformation))
    overlap, keypoint_inliers = pipe.compute_metrics_post_icp(transformation, args.metric_threshold)
    logging.info(f"overlap {overlap:.3f}, keypoint inlier ratio {keypoint_inliers:.3f}, total {time.perf_counter() - t0:.2f} s")
    if outputs:
        pipe.write_alignments(*outputs)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
