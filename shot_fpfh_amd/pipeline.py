"""Descriptor-based registration of two point clouds: keypoints -> descriptors -> matches -> RANSAC -> ICP.

Mirrors the stage methods of the reference's RegistrationPipeline (shot_fpfh/pipeline.py:34-608) -- same method
names, parameters, defaults and stored attributes -- so that a script written against it runs unchanged, with
every stage's heavy step on the MI355X: radius / k-NN search (K1 + K2), SHOT (K4 + K5), FPFH (K6 + K7),
brute-force matching (K8), RANSAC scoring (K9), the ICP nearest-neighbour queries (k-NN kernel).  The plotting
helper of the reference (`analyze_matches` with the 'double' algorithm) is not provided.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Callable, Literal, Optional

import numpy as np
import numpy.typing as npt

from . import keypoint_selection as _ks
from .core import RigidTransform
from .descriptors import ShotMultiprocessor, compute_fpfh_descriptor
from .helpers import write_ply
from .icp import icp_point_to_plane, icp_point_to_point, nearest_within
from .matching import (
    basic_matching,
    double_matching_with_rejects,
    match_descriptors,
    ransac_on_matches,
    threshold_filter,
)

__all__ = ["RegistrationPipeline"]


@dataclass
class RegistrationPipeline:
    """State of one registration: the two clouds with unit normals, then whatever the stages have produced.
    A stage leaves an already-filled attribute alone unless `force_recompute` is set (as in the reference)."""

    scan: npt.NDArray[np.float64]
    scan_normals: npt.NDArray[np.float64]
    ref: npt.NDArray[np.float64]
    ref_normals: npt.NDArray[np.float64]

    scan_keypoints: Optional[np.ndarray] = None
    ref_keypoints: Optional[np.ndarray] = None
    scan_descriptors: Optional[npt.NDArray[np.float64]] = None
    ref_descriptors: Optional[npt.NDArray[np.float64]] = None
    matches: Optional[tuple[np.ndarray, np.ndarray]] = None

    # ---- helpers: run one function on both sides, honouring the cache ---------------------------------
    def _fill(self, attr: str, force: bool, make: Callable[[str], np.ndarray]) -> None:
        for side in ("scan", "ref"):
            if getattr(self, f"{side}_{attr}") is None or force:
                setattr(self, f"{side}_{attr}", make(side))

    def _cloud(self, side: str):
        return getattr(self, side), getattr(self, f"{side}_normals"), getattr(self, f"{side}_keypoints")

    # ---- stage 1: keypoints (pipeline.py:53-130) ---------------------------------------------------------
    def select_keypoints(
        self,
        selection_algorithm: Literal["random", "iterative", "subsampling", "subsampling_with_density"],
        *,
        neighborhood_size: float | None = None,
        min_n_neighbors: int | None = None,
        proportion_picked: float = 0.5,
        force_recompute: bool = False,
    ) -> None:
        if selection_algorithm == "random":
            assert 0 <= proportion_picked <= 1, "Incorrect proportion passed."
        pick = {
            "random": lambda p: _ks.select_query_indices_randomly(p.shape[0], int(p.shape[0] * proportion_picked)),
            "iterative": lambda p: _ks.select_keypoints_iteratively(p, neighborhood_size),
            "subsampling": lambda p: _ks.select_keypoints_subsampling(p, neighborhood_size),
            "subsampling_with_density": lambda p: _ks.select_keypoints_with_density_threshold(p, neighborhood_size, min_n_neighbors),
        }.get(selection_algorithm)
        if pick is None:
            raise ValueError("Incorrect keypoint selection algorithm.")
        logging.info(f"-- Selecting keypoints: {selection_algorithm} --")
        self._fill("keypoints", force_recompute, lambda side: pick(getattr(self, side)))
        for side in ("scan", "ref"):
            logging.info(f"{getattr(self, side + '_keypoints').shape[0]} descriptors selected on {side} out of "
                         f"{getattr(self, side).shape[0]} points.")

    # ---- stage 2: descriptors (pipeline.py:132-349) --------------------------------------------------------
    def _shot(self, force: bool, config: dict, call: Callable[[ShotMultiprocessor, np.ndarray, np.ndarray, np.ndarray], np.ndarray]):
        with ShotMultiprocessor(**config) as shot_multiprocessor:
            def make(side):
                points, normals, kp = self._cloud(side)
                return call(shot_multiprocessor, points, points[kp], normals)
            self._fill("descriptors", force, make)

    def compute_shot_descriptor_single_scale(self, radius: float, subsampling_voxel_size: float | None = None,
                                             force_recompute: bool = False, **shot_multiprocessor_config) -> None:
        logging.info("-- Computing single-scale SHOT descriptors --")
        self._shot(force_recompute, shot_multiprocessor_config, lambda sm, pts, kp, nrm: sm.compute_descriptor_single_scale(
            point_cloud=pts, keypoints=kp, normals=nrm, radius=radius, subsampling_voxel_size=subsampling_voxel_size))

    def compute_shot_descriptor_bi_scale(self, local_rf_radius: float, shot_radius: float,
                                         subsampling_voxel_size: float | None = None, force_recompute: bool = False,
                                         **shot_multiprocessor_config) -> None:
        logging.info("-- Computing SHOT descriptors with two scales (local RF and SHOT) --")
        self._shot(force_recompute, shot_multiprocessor_config, lambda sm, pts, kp, nrm: sm.compute_descriptor_bi_scale(
            point_cloud=pts, keypoints=kp, normals=nrm, local_rf_radius=local_rf_radius, shot_radius=shot_radius,
            subsampling_voxel_size=subsampling_voxel_size))

    def compute_shot_descriptor_multiscale(self, radii, voxel_sizes=None, weights=None, force_recompute: bool = False,
                                           **shot_multiprocessor_config) -> None:
        logging.info("-- Computing multi-scale SHOT descriptors --")
        self._shot(force_recompute, shot_multiprocessor_config, lambda sm, pts, kp, nrm: sm.compute_descriptor_multiscale(
            point_cloud=pts, keypoints=kp, normals=nrm, radii=radii, voxel_sizes=voxel_sizes, weights=weights))

    def compute_descriptors(
        self,
        radius: float,
        descriptor_choice: Literal["fpfh", "shot_single_scale", "shot_bi_scale", "shot_multiscale"] = "shot_single_scale",
        fpfh_n_bins: int = 5,
        phi: float = 3.0,
        rho: float = 10.0,
        n_scales: int = 2,
        subsample_support: bool = True,
        normalize: bool = True,
        share_local_rfs: bool = True,
        min_neighborhood_size: int = 100,
        n_procs: int = 8,
        disable_progress_bars: bool = False,
        verbose: bool = True,
        force_recompute: bool = False,
    ) -> None:
        shot_config = dict(normalize=normalize, share_local_rfs=share_local_rfs, min_neighborhood_size=min_neighborhood_size,
                           n_procs=n_procs, disable_progress_bar=disable_progress_bars, verbose=verbose)
        voxel = radius / rho if subsample_support else None
        if descriptor_choice == "shot_single_scale":
            self.compute_shot_descriptor_single_scale(radius=radius, subsampling_voxel_size=voxel,
                                                      force_recompute=force_recompute, **shot_config)
        elif descriptor_choice == "shot_bi_scale":
            self.compute_shot_descriptor_bi_scale(local_rf_radius=radius, shot_radius=radius * phi, subsampling_voxel_size=voxel,
                                                  force_recompute=force_recompute, **shot_config)
        elif descriptor_choice in ("shot_multi_scale", "shot_multiscale"):  # the reference tests the first spelling (:321)
            scales = radius * phi ** np.arange(n_scales)
            self.compute_shot_descriptor_multiscale(radii=scales, voxel_sizes=scales / rho, force_recompute=force_recompute,
                                                    **shot_config)
        elif descriptor_choice == "fpfh":
            def make(side):
                points, normals, kp = self._cloud(side)
                return compute_fpfh_descriptor(kp, points, normals, radius=radius, n_bins=fpfh_n_bins,
                                               disable_progress_bars=disable_progress_bars, verbose=verbose)
            self._fill("descriptors", force_recompute, make)
        else:
            raise ValueError("Incorrect descriptor choice")

    # ---- stage 3: matching (pipeline.py:351-412) -------------------------------------------------------------
    def find_descriptors_matches(
        self,
        matching_algorithm: Literal["simple", "double", "threshold"],
        *,
        reject_threshold: float,
        threshold_multiplier: float,
        debug_mode: bool = False,
        force_recompute: bool = False,
    ) -> None:
        run = {
            "simple": lambda: basic_matching(self.scan_descriptors, self.ref_descriptors),
            "double": lambda: double_matching_with_rejects(self.scan_descriptors, self.ref_descriptors, reject_threshold),
            "threshold": lambda: match_descriptors(self.scan_descriptors, self.ref_descriptors, threshold_filter,
                                                   threshold_multiplier=threshold_multiplier),
        }.get(matching_algorithm)
        if run is None:
            raise ValueError("Incorrect matching algorithm selection.")
        logging.info(f"-- Matching descriptors: {matching_algorithm} --")
        if self.matches is None or force_recompute:
            self.matches = run()
        if debug_mode:
            gap = np.linalg.norm(self.scan_descriptors[self.matches[0]] - self.ref_descriptors[self.matches[1]], axis=0)
            logging.info(f"Maximum distance between the descriptors matched: {gap.max(initial=0):.2f}")

    # ---- stage 4: coarse registration (pipeline.py:445-486) ----------------------------------------------------
    def run_ransac(self, *, n_draws: int = 10000, draw_size: int = 4, max_inliers_distance: float = 2,
                   exact_transformation: RigidTransform | None = None,
                   disable_progress_bar: bool = False) -> tuple[RigidTransform, float]:
        logging.info(" -- Aligning the point clouds by RANSAC-ing the matches --")
        inliers_ratio, transformation = ransac_on_matches(
            *self.matches, self.scan[self.scan_keypoints], self.ref[self.ref_keypoints], n_draws=n_draws,
            draw_size=draw_size, distance_threshold=max_inliers_distance, disable_progress_bar=disable_progress_bar)
        if exact_transformation is not None:
            cosine = (np.trace(exact_transformation.rotation @ transformation.rotation.T) - 1) / 2
            logging.info(f"Norm of the angle between the two rotations: {abs(np.arccos(cosine)):.2f}\n"
                         f"Norm of the difference between the two translation: "
                         f"{np.linalg.norm(exact_transformation.translation - transformation.translation):.2f}")
        return transformation, inliers_ratio

    # ---- stage 5: fine registration (pipeline.py:488-542) ---------------------------------------------------------
    def run_icp(self, icp_type: Literal["point_to_point", "point_to_plane"], transformation_init: RigidTransform, *,
                d_max: float, voxel_size: float = 0.2, max_iter: int = 30, rms_threshold: float = 1e-2,
                disable_progress_bar: bool = False) -> tuple[RigidTransform, float, bool]:
        common = dict(d_max=d_max, voxel_size=voxel_size, max_iter=max_iter, rms_threshold=rms_threshold,
                      disable_progress_bar=disable_progress_bar)
        if icp_type == "point_to_point":
            return icp_point_to_point(self.scan, self.ref, transformation_init, **common)
        if icp_type == "point_to_plane":
            return icp_point_to_plane(self.scan, self.ref, self.ref_normals, transformation_init, **common)
        raise ValueError("Incorrect ICP type selected.")

    # ---- evaluation / output (pipeline.py:544-608) ---------------------------------------------------------------------
    def compute_metrics_post_icp(self, transformation_icp: RigidTransform, distance_threshold: float) -> tuple[float, float]:
        """(overlap of the aligned scan with ref, ratio of aligned scan keypoints that have a ref keypoint within
        the threshold)."""
        aligned = transformation_icp[self.scan]
        return (nearest_within(aligned, self.ref, distance_threshold) / aligned.shape[0],
                nearest_within(aligned[self.scan_keypoints], self.ref[self.ref_keypoints], distance_threshold)
                / self.scan_keypoints.shape[0])

    def write_alignments(self, *args: tuple[str, RigidTransform]) -> None:
        """One PLY per (file_name, transform): the transformed scan stacked on ref, with an `is_scan` column."""
        is_scan = np.hstack((np.ones(self.scan.shape[0], dtype=bool), np.zeros(self.ref.shape[0], dtype=bool)))[:, None]
        for file_name, transform in args:
            write_ply(file_name, [np.hstack((np.vstack((transform[self.scan], self.ref)), is_scan))], ["x", "y", "z", "is_scan"])
