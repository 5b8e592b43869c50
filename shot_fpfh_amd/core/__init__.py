from .geometry import (
    RigidTransform,
    grid_subsampling,
    solver_point_to_plane,
    solver_point_to_point,
    voxel_closest_to_barycentre,
)

__all__ = ["RigidTransform", "solver_point_to_point", "solver_point_to_plane", "grid_subsampling", "voxel_closest_to_barycentre"]
