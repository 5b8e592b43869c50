"""ctypes binding of libshotfpfh.so (C ABI: include/shotfpfh.h).

There is deliberately no fallback: if the shared library is missing, or no MI355X is visible,
importing the symbols works (so `-m "not gpu"` tests can check the export table) but creating an
engine raises `ShotFpfhError`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libshotfpfh.so")

SF_HOST, SF_OUT_DEVICE, SF_IN_DEVICE = 0, 1, 2
SHOT_LEN = 352
MAX_FPFH_BINS = 1290  # SF_MAX_FPFH_BINS: n_bins^3 fits an int (n_bins above 8 take the generic kernels; memory is the real bound)


class ShotFpfhError(RuntimeError):
    """Raised for any failure reported by the native library (or its absence)."""


_vp, _i64, _i32, _f64, _int, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_int, C.c_size_t

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "sf_last_error": (C.c_char_p, []),
    "sf_version": (C.c_char_p, []),
    "sf_device_count": (_int, []),
    "sf_create": (_vp, [_int]),
    "sf_destroy": (None, [_vp]),
    "sf_sync": (_int, [_vp]),
    "sf_stream": (_vp, [_vp]),
    "sf_fork": (_int, [_vp]),
    "sf_switch": (_int, [_vp, _int]),
    "sf_join": (_int, [_vp]),
    "sf_mark": (_int, [_vp]),
    "sf_wait_mark": (_int, [_vp]),
    "sf_dev_alloc": (_vp, [_vp, _sz]),
    "sf_dev_free": (_int, [_vp, _vp]),
    "sf_host_alloc": (_vp, [_vp, _sz]),
    "sf_host_free": (_int, [_vp, _vp]),
    "sf_h2d": (_int, [_vp, _vp, _vp, _sz]),
    "sf_d2h": (_int, [_vp, _vp, _vp, _sz]),
    "sf_d2d": (_int, [_vp, _vp, _vp, _sz]),
    "sf_cloud_upload": (_vp, [_vp, _vp, _vp, _i64, _int]),
    "sf_cloud_set_normals": (_int, [_vp, _vp, _vp, _int]),
    "sf_cloud_build_grid": (_int, [_vp, _vp, _f64]),
    "sf_cloud_build_grid_block": (_int, [_vp, _vp, _f64, _i64, _i64, _int, _vp, _vp]),
    "sf_cloud_size": (_i64, [_vp]),
    "sf_cloud_free": (None, [_vp, _vp]),
    "sf_cloud_perm": (_int, [_vp, _vp, _vp]),
    "sf_cloud_halo_range": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "sf_cloud_layer_table": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "sf_radius_search": (_vp, [_vp, _vp, _vp, _i64, _f64, _int]),
    "sf_radius_search_self": (_vp, [_vp, _vp, _f64, _i64, _i64]),
    "sf_knn_search": (_vp, [_vp, _vp, _vp, _i64, _int, _int]),
    "sf_nbrs_import": (_vp, [_vp, _vp, _vp, _i64, _vp, _vp, _f64, _int]),
    "sf_nbrs_slice": (_vp, [_vp, _vp, _i64, _i64]),
    "sf_nbrs_num_queries": (_i64, [_vp]),
    "sf_nbrs_total": (_i64, [_vp]),
    "sf_nbrs_max_count": (_i64, [_vp]),
    "sf_nbrs_max_count_all": (_i64, [_vp]),
    "sf_nbrs_export": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "sf_nbrs_free": (None, [_vp, _vp]),
    "sf_normals": (_int, [_vp, _vp, _vp, _vp, _vp, _int]),
    "sf_normals_radius": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _f64, _vp, _vp, _int]),
    "sf_pca": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int]),
    "sf_shot_lrf": (_int, [_vp, _vp, _vp, _vp, _int]),
    "sf_shot": (_int, [_vp, _vp, _vp, _vp, _int, _i64, _vp, _int]),
    "sf_shot_single_scale": (_int, [_vp, _vp, _vp, _int, _i64, _vp, _vp, _int]),
    "sf_azimuth_idx": (_int, [_vp, _vp, _vp, _i64, _vp, _int]),
    "sf_shot_serial": (_int, [_vp, _vp, _vp, _i64, _vp, _int]),
    "sf_sync_count": (C.c_ulonglong, []),
    "sf_graph_begin": (_int, [_vp]),
    "sf_graph_end": (_vp, [_vp]),
    "sf_graph_launch": (_int, [_vp, _vp]),
    "sf_graph_free": (None, [_vp, _vp]),
    "sf_spfh_create": (_vp, [_vp, _vp, _int, _i64]),
    "sf_spfh_create_for_radius": (_vp, [_vp, _vp, _int, _i64, C.c_double]),
    "sf_spfh_elem_bytes": (_int, [_vp]),
    "sf_spfh_compute": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "sf_spfh_compute_moments": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "sf_shot_from_moments": (_int, [_vp, _vp, _vp, _vp, _int, _i64, _vp, _vp, _int]),
    "sf_lrf_raw_from_moments": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "sf_shot_from_raw_lrf": (_int, [_vp, _vp, _vp, _vp, _int, _i64, _vp]),
    "sf_spfh_allgather": (_int, [_vp, _vp, _i64]),
    "sf_spfh_exchange_rows": (_int, [_vp, _vp, _int, _vp, _vp, _vp, _vp, _vp]),
    "sf_spfh_rows_image": (_int, [_vp, _vp, _i64, _i64, _vp, _sz, _int, _vp]),
    "sf_spfh_export": (_int, [_vp, _vp, _vp, _vp, _int]),
    "sf_spfh_free": (None, [_vp, _vp]),
    "sf_fpfh": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _int]),
    "sf_match_argmin": (_int, [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _int]),
    "sf_rows_nonzero": (_int, [_vp, _vp, _i64, _i64, _vp]),
    "sf_rows_gather": (_int, [_vp, _vp, _i64, _vp, _i64, _i64, _vp]),
    "sf_match_argmin_multiscale": (_int, [_vp, _vp, _vp, _int, _i64, _i64, _i64, _vp, _vp, _f64, _vp, _vp, _int]),
    "sf_match_col_candidates": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "sf_match_stream_begin": (_vp, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _f64, _i64]),
    "sf_match_stream_feed": (_int, [_vp, _vp, _i64, _i64]),
    "sf_match_stream_end": (_int, [_vp, _vp, _vp, _vp]),
    "sf_match_stream_abort": (None, [_vp, _vp]),
    "sf_rows_abs_max": (_int, [_vp, _vp, _i64, _i64, _vp]),
    "sf_ransac_score": (_int, [_vp, _vp, _vp, _i64, _vp, _i64, _f64, _vp, _int]),
    "sf_voxels_build": (_vp, [_vp, _vp, _i64, _f64, _int]),
    "sf_voxels_count": (_i64, [_vp]),
    "sf_voxels_inverse": (_int, [_vp, _vp, _vp]),
    "sf_voxels_select": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "sf_voxels_free": (None, [_vp, _vp]),
    "sf_icp_accumulate": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _f64, _int, _vp]),
    "sf_transform_points": (_int, [_vp, _vp, _i64, _vp]),
    "sf_comm_unique_id": (_int, [_vp]),
    "sf_comm_init": (_int, [_vp, _vp, _int, _int]),
    "sf_comm_allgather": (_int, [_vp, _vp, _vp, _sz]),
    "sf_comm_exchange": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp]),
    "sf_comm_allreduce_min_u64": (_int, [_vp, _vp, _vp, _sz]),
    "sf_comm_collective_stats": (_int, [_vp, _int]),
    "sf_comm_destroy": (_int, [_vp]),
    "sf_profile_enable": (_int, [_vp, _int]),
    "sf_profile_reset": (_int, [_vp]),
    "sf_profile_only": (_int, [_vp, C.c_char_p]),
    "sf_profile_report": (_i64, [_vp, _vp, _i64]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen libshotfpfh.so and declare every prototype.  Raises ShotFpfhError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ShotFpfhError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  shot_fpfh_amd has no CPU fallback."
        )
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as exc:  # e.g. librccl / libamdhip64 not found
        raise ShotFpfhError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().sf_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise ShotFpfhError(f"{what or 'libshotfpfh'} failed ({rc}): {last_error()}")


def check_handle(h, what: str):
    if not h:
        raise ShotFpfhError(f"{what} failed: {last_error()}")
    return h
