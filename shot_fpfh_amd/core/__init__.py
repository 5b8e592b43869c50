from .geometry import RigidTransform, grid_subsampling, solver_point_to_point

__all__ = ["RigidTransform", "solver_point_to_point", "grid_subsampling"]
