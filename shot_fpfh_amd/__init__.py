"""shot_fpfh_amd -- MI355X (gfx950) native SHOT / FPFH descriptor engine.

Drop-in for the hot path of aubin-tchoi/shot-fpfh: `shot_fpfh_amd.descriptors` and
`shot_fpfh_amd.matching` mirror `shot_fpfh.descriptors` / `shot_fpfh.matching`; the work runs in
hand-written HIP kernels in libshotfpfh.so (C ABI: include/shotfpfh.h) reached through ctypes.
No PyTorch, no CPU fallback.
"""
from . import _ffi
from ._ffi import ShotFpfhError
from .descriptors import ShotMultiprocessor, compute_fpfh_descriptor, compute_normals
from .engine import Cloud, DeviceArray, Engine, Neighbors, Spfh, default_engine
from .helpers import get_data, read_ply, write_ply
from .matching import basic_matching, match_descriptors, ransac_on_matches
from .pipeline import RegistrationPipeline

__all__ = [
    "ShotFpfhError",
    "Engine",
    "Cloud",
    "Neighbors",
    "Spfh",
    "DeviceArray",
    "default_engine",
    "compute_fpfh_descriptor",
    "compute_normals",
    "ShotMultiprocessor",
    "basic_matching",
    "match_descriptors",
    "ransac_on_matches",
    "RegistrationPipeline",
    "read_ply",
    "write_ply",
    "get_data",
]
