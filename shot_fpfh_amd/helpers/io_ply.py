"""Binary PLY point-cloud files and `get_data`, the loader in front of the descriptor path
(same behaviour as shot_fpfh/helpers/io_ply.py:58-301, written independently).

Format handled: the vertex-only binary PLY the reference reads and writes -- a text header (`ply`,
`format binary_{little,big}_endian 1.0`, one `element vertex N`, scalar `property <type> <name>` lines,
`end_header`) followed by N packed records.  ASCII bodies are rejected, as in the reference.
`get_data(..., normals_computation_callback=compute_normals)` plugs the K2 + K3 normals kernel in where the
reference plugs its NumPy loop.
"""
from __future__ import annotations

import logging
import sys
from typing import Optional, Protocol, Sequence

import numpy as np
import numpy.typing as npt

__all__ = ["read_ply", "write_ply", "get_data", "NormalsComputationCallback"]

# PLY scalar type names (both spellings) -> NumPy type codes without byte order
_SCALAR = {
    "int8": "i1", "char": "i1", "uint8": "u1", "uchar": "u1",
    "int16": "i2", "short": "i2", "uint16": "u2", "ushort": "u2",
    "int32": "i4", "int": "i4", "uint32": "u4", "uint": "u4",
    "float32": "f4", "float": "f4", "float64": "f8", "double": "f8",
}
_ORDER = {"binary_little_endian": "<", "binary_big_endian": ">"}


def read_ply(filename: str) -> np.ndarray:
    """Structured array with one field per vertex property (`data["x"]`, ...).  Raises ValueError when the
    file does not start with `ply` or is ASCII (io_ply.py:91-96)."""
    with open(filename, "rb") as f:
        if b"ply" not in f.readline():
            raise ValueError("The file does not start with the word ply")
        fmt = f.readline().split()[1].decode()
        if fmt == "ascii":
            raise ValueError("The file is not binary")
        order = _ORDER[fmt]
        count, fields = None, []
        while True:
            line = f.readline()
            if not line or b"end_header" in line:
                break
            words = line.split()
            if b"element" in line:
                count = int(words[2])  # the last element line wins, as in the reference's header scan
            elif b"property" in line:
                fields.append((words[2].decode(), order + _SCALAR[words[1].decode()]))
        return np.fromfile(f, dtype=fields, count=-1 if count is None else count)


def write_ply(filename: str, field_list, field_names: Sequence[str]) -> bool:
    """Write 1-D arrays / the columns of 2-D arrays as vertex properties, native byte order.  Returns False
    (after a logged warning) on malformed input instead of raising, like the reference (io_ply.py:125-213).
    `.ply` is appended to the name when missing."""
    columns = []
    for block in list(field_list) if isinstance(field_list, (list, tuple)) else [field_list]:
        if block is None:
            logging.warning("WRITE_PLY ERROR: a field is None")
            return False
        if block.ndim > 2:
            logging.warning("WRITE_PLY ERROR: a field have more than 2 dimensions")
            return False
        columns.extend([block] if block.ndim < 2 else list(block.T))
    if len({c.shape[0] for c in columns}) > 1:
        logging.warning("wrong field dimensions")
        return False
    if len(columns) != len(field_names):
        logging.warning("wrong number of field names")
        return False
    if not filename.endswith(".ply"):
        filename += ".ply"
    n = columns[0].shape[0] if columns else 0
    header = ["ply", f"format binary_{sys.byteorder}_endian 1.0", f"element vertex {n}"]
    header += [f"property {c.dtype.name} {name}" for c, name in zip(columns, field_names)]
    header.append("end_header")
    records = np.empty(n, dtype=[(name, c.dtype.str) for c, name in zip(columns, field_names)])
    for c, name in zip(columns, field_names):
        records[name] = c
    with open(filename, "wb") as f:
        f.write(("\n".join(header) + "\n").encode())
        records.tofile(f)
    return True


class NormalsComputationCallback(Protocol):
    """Signature of `compute_normals` (io_ply.py:241-256)."""

    def __call__(self, query_points: npt.NDArray[np.float64], cloud_points: npt.NDArray[np.float64], *,
                 k: Optional[int] = None, radius: Optional[float] = None,
                 pre_computed_normals: Optional[npt.NDArray[np.float64]] = None) -> npt.NDArray[np.float64]: ...


def get_data(
    data_path: str,
    remove_duplicates: bool = False,
    recompute_normals: bool = True,
    k: Optional[int] = None,
    radius: Optional[float] = None,
    normals_computation_callback: Optional[NormalsComputationCallback] = None,
) -> tuple[npt.NDArray[np.float64], npt.NDArray[np.float64]]:
    """(points (N,3), normals (N,3)) of a PLY cloud (io_ply.py:259-301).  Stored normals (`nx ny nz` or
    `n_x n_y n_z`) are used as they are, or -- by default -- only to orient freshly computed ones; without
    stored normals the callback is mandatory.  `remove_duplicates` keeps the first point of every
    1e-4-rounded coordinate triple, in np.unique's order."""
    data = read_ply(data_path)
    points = np.vstack((data["x"], data["y"], data["z"])).T
    stored = next((names for names in (("nx", "ny", "nz"), ("n_x", "n_y", "n_z")) if names[0] in data.dtype.fields), None)
    if stored is not None:
        normals = np.vstack([data[name] for name in stored]).T
        if recompute_normals:
            logging.info(f"Recomputing normals using function {normals_computation_callback.__name__}.")
            normals = normals_computation_callback(points, points, k=k, radius=radius, pre_computed_normals=normals)
    else:
        if normals_computation_callback is None:
            raise ValueError(
                "The function used to compute normals needs to be specified as the ply file does not contain normals."
            )
        normals = normals_computation_callback(points, points, k=k, radius=radius)
    if remove_duplicates:
        keep = np.unique(points.round(decimals=4), axis=0, return_index=True)[1]
        return points[keep], normals[keep]
    return points, normals
