"""Data formats on either side of the descriptor path (mirrors the I/O part of shot_fpfh.helpers)."""
from .io_ply import NormalsComputationCallback, get_data, read_ply, write_ply

__all__ = ["read_ply", "write_ply", "get_data", "NormalsComputationCallback"]
