"""Geometry that rides with the hot path: rigid transforms, the Kabsch solver RANSAC uses per draw, and voxel-grid
support subsampling.

The small dense solves stay on the host on purpose: a RANSAC run needs 10^4 3x3 SVDs whose results must
equal NumPy/LAPACK's bit for bit for the draw-by-draw inlier counts to match, and they cost microseconds each; only
the O(draws x matches) scoring is a kernel (K9).  Voxel subsampling -- O(N), in front of every subsampled SHOT call --
runs on the device (csrc/voxel.hip).
"""
from __future__ import annotations

import numpy as np
import numpy.typing as npt
from scipy.spatial.transform import Rotation

__all__ = ["RigidTransform", "solver_point_to_point", "solver_point_to_plane", "grid_subsampling", "voxel_closest_to_barycentre"]


class RigidTransform:
    """Rotation + translation acting on row-vector points: p -> p @ R.T + t
    (mirrors shot_fpfh/core/rigid_transform.py:10-106; `transform[points]` applies it)."""

    def __init__(self, rotation: npt.NDArray[np.float64] | None = None, translation: npt.NDArray[np.float64] | None = None):
        self.rotation = np.eye(3) if rotation is None else rotation
        self.translation = np.zeros(3) if translation is None else translation

    def __getitem__(self, points: npt.NDArray[np.float64]) -> npt.NDArray[np.float64]:
        return points.dot(self.rotation.T) + self.translation  # rigid_transform.py:81-88

    def transform(self, points: npt.NDArray[np.float64]) -> npt.NDArray[np.float64]:
        return self[points]

    def normalize_rotation(self) -> None:
        """Re-orthonormalise through a unit quaternion (rigid_transform.py:45-52)."""
        quat = Rotation.from_matrix(self.rotation).as_quat()
        self.rotation = Rotation.from_quat(quat / np.linalg.norm(quat)).as_matrix()

    def __matmul__(self, other: "RigidTransform") -> "RigidTransform":
        composed = RigidTransform(self.rotation @ other.rotation, self.rotation @ other.translation + self.translation)
        composed.normalize_rotation()
        return composed

    def __invert__(self) -> "RigidTransform":
        # the reference returns (R^T, -t) here (rigid_transform.py:72-79), not the true inverse; kept as is
        return RigidTransform(self.rotation.T, -self.translation)

    def inv(self) -> "RigidTransform":
        return ~self

    def as_row12(self) -> npt.NDArray[np.float64]:
        """[R row-major (9), t (3)] -- the per-draw record sf_ransac_score consumes."""
        return np.concatenate([np.asarray(self.rotation, dtype=np.float64).reshape(9), np.asarray(self.translation, dtype=np.float64)])

    def __repr__(self) -> str:
        m = np.vstack((np.hstack((self.rotation, self.translation[:, None])), [0, 0, 0, 1]))
        with np.printoptions(suppress=True):
            return str(m).replace("[", "").replace("]", "")


def solver_point_to_point(scan: npt.NDArray[np.float64], ref: npt.NDArray[np.float64]) -> RigidTransform:
    """Least-squares rigid fit scan -> ref (Kabsch), following shot_fpfh/core/solvers.py:9-30 call for
    call so the SVD sign conventions -- hence the transforms -- are identical."""
    scan_center, ref_center = scan.mean(axis=0), ref.mean(axis=0)
    cross_cov = (scan - scan_center).T.dot(ref - ref_center)
    u, _, vt = np.linalg.svd(cross_cov)
    rot = vt.T @ u.T
    if np.linalg.det(rot) < 0:  # reflection: flip the last singular direction (solvers.py:23-26)
        ut = u.T
        ut[-1] *= -1
        rot = vt.T @ ut
    return RigidTransform(rot, ref_center - rot.dot(scan_center))


def solver_point_to_point_batched(scan: npt.NDArray[np.float64], ref: npt.NDArray[np.float64], return_reflected: bool = False):
    """`solver_point_to_point` for a stack of draws: scan, ref of shape (n, k, 3) -> rotations (n, 3, 3),
    translations (n, 3).

    Same operations on the same operands as the per-draw function, through NumPy's stacked forms of the same
    routines (`matmul` -> the same BLAS call per matrix, `linalg.svd` / `det` -> the same LAPACK call per matrix), so
    every transform comes out bit-identical to the per-draw one; `ransac_on_matches` verifies that on a sample of
    draws and on every reflected draw of every call and falls back to the per-draw loop otherwise.  ~8x less host
    time at 10 000 draws.
    """
    scan_center, ref_center = scan.mean(axis=1), ref.mean(axis=1)
    cross_cov = np.matmul((scan - scan_center[:, None, :]).transpose(0, 2, 1), ref - ref_center[:, None, :])
    u, _, vt = np.linalg.svd(cross_cov)
    ut = u.transpose(0, 2, 1)
    rot = np.matmul(vt.transpose(0, 2, 1), ut)
    neg = np.flatnonzero(np.linalg.det(rot) < 0)
    if neg.size:  # reflections: flip the last singular direction (solvers.py:23-26)
        ut = ut.copy()
        ut[neg, -1] *= -1
        rot[neg] = np.matmul(vt[neg].transpose(0, 2, 1), ut[neg])
    # rot.dot(centre) (solvers.py:28, a dgemv) as a stacked matrix-vector product: NumPy's matmul hands every (3, 3) @ (3, 1)
    # item to the same gemv.  A BLAS build that rounded the two differently would change a draw's translation by an ulp --
    # and its inlier count: the callers hold every stack to the per-draw solver on a sample (ransac._solve_chunk) and take the
    # per-draw loop on any difference.  (The Python loop over 10^4 draws this replaces cost 7 ms.)
    translation = ref_center - np.matmul(rot, scan_center[:, :, None])[:, :, 0]
    return (rot, translation, neg) if return_reflected else (rot, translation)


def solver_point_to_plane(scan: npt.NDArray[np.float64], ref: npt.NDArray[np.float64],
                          normals_ref: npt.NDArray[np.float64]) -> RigidTransform:
    """Linearised point-to-plane fit (small-angle Euler xyz + translation), shot_fpfh/core/solvers.py:33-48:
    rows g_i = [scan_i x n_i, n_i], h_i = (ref_i - scan_i) . n_i, solve (G^T G) s = G^T h."""
    g = np.hstack((np.cross(scan, normals_ref), normals_ref))
    h = np.einsum("ij, ij->i", ref - scan, normals_ref)
    solution = np.linalg.solve(g.T @ g, g.T @ h)
    return RigidTransform(Rotation.from_euler("xyz", solution[:3]).as_matrix(), solution[3:6])


def voxel_closest_to_barycentre(points: npt.NDArray[np.float64], voxel_size: float, *, within_voxel_order: str = "numpy",
                                engine=None):
    """Per occupied voxel (np.unique's lexicographic key order): the index of the point closest to the voxel's
    barycentre, and the number of points in the voxel (shot_fpfh/core/subsampling.py:12-37 and the identical
    loop of keypoint_selection.py:80-101) -- on the GPU: voxel keys, stable sort, run detection, one thread per voxel
    for barycentre / distances / first minimum (csrc/voxel.hip).

    The reference visits a voxel's points in the order `np.argsort(inverse)` gives them -- an UNSTABLE sort -- and a
    two-point voxel is an exact distance tie (both points are equally far from their midpoint), so which of the two it
    returns is decided by NumPy's sort implementation.  `within_voxel_order="numpy"` (default) takes that order from the
    same call on the same array (one host argsort of the device-computed `inverse`) and feeds it to the device
    selection: results identical to the reference's on the same NumPy build, ties included.  `"index"` visits a voxel's
    points by ascending index, entirely on the device and independent of the platform; it differs from the reference
    only on exact ties and last-bit barycentre rounding.
    """
    import ctypes as C

    from .. import _ffi
    from ..engine import default_engine

    if within_voxel_order not in ("numpy", "index"):
        raise ValueError("within_voxel_order must be 'numpy' or 'index'")
    pts = np.ascontiguousarray(points, dtype=np.float64)
    if pts.ndim != 2 or pts.shape[1] != 3:
        raise ValueError(f"expected an (N, 3) array, got shape {pts.shape}")
    n = pts.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    eng = engine or default_engine()
    lib = eng.lib
    vox = _ffi.check_handle(lib.sf_voxels_build(eng.h, pts.ctypes.data_as(C.c_void_p), n, float(voxel_size), _ffi.SF_HOST),
                            "sf_voxels_build")
    try:
        count = int(lib.sf_voxels_count(vox))
        order_ptr = None
        if within_voxel_order == "numpy":
            inverse = np.empty(n, dtype=np.int64)
            _ffi.check(lib.sf_voxels_inverse(eng.h, vox, inverse.ctypes.data_as(C.c_void_p)), "sf_voxels_inverse")
            order = np.ascontiguousarray(np.argsort(inverse), dtype=np.int64)  # the reference's call (subsampling.py:19)
            order_ptr = order.ctypes.data_as(C.c_void_p)
        picked, counts = np.empty(count, dtype=np.int64), np.empty(count, dtype=np.int64)
        _ffi.check(lib.sf_voxels_select(eng.h, vox, order_ptr, picked.ctypes.data_as(C.c_void_p),
                                        counts.ctypes.data_as(C.c_void_p)), "sf_voxels_select")
    finally:
        lib.sf_voxels_free(eng.h, vox)
    return picked, counts


def grid_subsampling_many(points: npt.NDArray[np.float64], voxel_sizes, *, engine=None) -> list:
    """grid_subsampling for several voxel sizes of ONE cloud (the supports of a multi-scale SHOT): the voxel partitions are
    built on the device one after the other, then the host-side `np.argsort(inverse)` calls -- the reference's unstable sort,
    11 ms per million points each and most of what a support costs -- run side by side in threads (NumPy releases the
    interpreter lock while it sorts), then the device selects.  Same indices as one grid_subsampling call per size."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    from .. import _ffi
    from ..engine import default_engine

    sizes = [float(v) for v in voxel_sizes]
    pts = np.ascontiguousarray(points, dtype=np.float64)
    if pts.ndim != 2 or pts.shape[1] != 3:
        raise ValueError(f"expected an (N, 3) array, got shape {pts.shape}")
    n = pts.shape[0]
    if n == 0 or len(sizes) < 2:
        return [grid_subsampling(pts, v, engine=engine) for v in sizes]
    eng = engine or default_engine()
    lib = eng.lib
    voxels, inverses = [], []
    try:
        for v in sizes:
            voxels.append(_ffi.check_handle(lib.sf_voxels_build(eng.h, pts.ctypes.data_as(C.c_void_p), n, v, _ffi.SF_HOST), "sf_voxels_build"))
            inv = np.empty(n, dtype=np.int64)
            _ffi.check(lib.sf_voxels_inverse(eng.h, voxels[-1], inv.ctypes.data_as(C.c_void_p)), "sf_voxels_inverse")
            inverses.append(inv)
        with ThreadPoolExecutor(max_workers=min(len(sizes), 4)) as pool:
            orders = list(pool.map(lambda a: np.ascontiguousarray(np.argsort(a), dtype=np.int64), inverses))  # (subsampling.py:19)
        out = []
        for vox, order in zip(voxels, orders):
            count = int(lib.sf_voxels_count(vox))
            picked, counts = np.empty(count, dtype=np.int64), np.empty(count, dtype=np.int64)
            _ffi.check(lib.sf_voxels_select(eng.h, vox, order.ctypes.data_as(C.c_void_p), picked.ctypes.data_as(C.c_void_p),
                                            counts.ctypes.data_as(C.c_void_p)), "sf_voxels_select")
            out.append(picked)
        return out
    finally:
        for vox in voxels:
            lib.sf_voxels_free(eng.h, vox)


def grid_subsampling(points: npt.NDArray[np.float64], voxel_size: float, *, within_voxel_order: str = "numpy",
                     engine=None) -> npt.NDArray[np.int64]:
    """Voxel subsampling: per occupied voxel keep the point closest to the voxel's barycentre; voxels
    come out in np.unique's lexicographic key order (shot_fpfh/core/subsampling.py:5-39).  Runs on the GPU; see
    `voxel_closest_to_barycentre` for `within_voxel_order`."""
    return voxel_closest_to_barycentre(points, voxel_size, within_voxel_order=within_voxel_order, engine=engine)[0]
