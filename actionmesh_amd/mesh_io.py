"""Deformation export for the vertex tensors the Stage-II loop produces (the array half of it):
`save_deformation` of actionmesh/io/mesh_io.py:43-106 without the detour through T trimesh objects - the (T, V, 3) vertex
stack goes from the device to `{stem}_vertices.npy` in one copy, with the reference's axis convention (columns [2, 0, 1],
new x negated, float32) and `{stem}_faces.npy` (int32).  Per-frame GLB files and the Blender shape-key export
(mesh_io.py:109-118, glb_export.py) need trimesh / bpy and stay on the reference path.
"""
from __future__ import annotations

from pathlib import Path
from typing import Tuple, Union

import numpy as np
import torch


def save_deformation(vertices: Union[torch.Tensor, np.ndarray], faces: Union[torch.Tensor, np.ndarray],
                     path: Union[str, Path]) -> Tuple[Path, Path]:
    """vertices (T, V, 3) of meshes sharing `faces` (F, 3) -> ({stem}_vertices.npy, {stem}_faces.npy) next to `path`."""
    v = vertices.detach().to("cpu", torch.float32).numpy() if isinstance(vertices, torch.Tensor) else np.asarray(vertices, dtype=np.float32)
    f = faces.detach().cpu().numpy() if isinstance(faces, torch.Tensor) else np.asarray(faces)
    if v.ndim != 3 or v.shape[0] == 0 or v.shape[2] != 3:
        raise ValueError(f"Cannot save deformation from a vertex stack of shape {v.shape}: need (T >= 1, V, 3)")
    if f.ndim != 2 or f.shape[1] != 3:
        raise ValueError(f"faces must be (F, 3), got {f.shape}")
    if f.size and (f.min() < 0 or f.max() >= v.shape[1]):
        raise ValueError("faces index vertices that do not exist")
    out = np.ascontiguousarray(v[:, :, [2, 0, 1]])
    out[:, :, 0] = -out[:, :, 0]
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    vertices_path = path.parent / f"{path.stem}_vertices.npy"
    faces_path = path.parent / f"{path.stem}_faces.npy"
    np.save(vertices_path, out)
    np.save(faces_path, f.astype(np.int32))
    return vertices_path, faces_path
