"""Deformation export for the vertex tensors the Stage-II loop produces (the array half of it):
`save_deformation` of actionmesh/io/mesh_io.py:43-106 without the detour through T trimesh objects - the (T, V, 3) vertex
stack goes from the device to `{stem}_vertices.npy` in one copy, with the reference's axis convention (columns [2, 0, 1],
new x negated, float32) and `{stem}_faces.npy` (int32).

The two GLB outputs of the reference are written directly as glTF 2.0 binaries (a 12-byte header, one JSON chunk, one BIN
chunk - the container is a few dozen lines, so neither trimesh nor a Blender subprocess is needed):
  * `save_meshes` (mesh_io.py:109-118, trimesh's `mesh.export("mesh_XX.glb")`): one static triangle mesh per frame;
  * `create_animated_glb` (glb_export.py:18-87, 142-284: Blender shape keys "Frame_i" keyed 1 at frame i and 0 at i +- 1):
    ONE mesh with a morph target per frame and a LINEAR weights animation, i.e. what Blender's exporter writes for those
    shape keys, including its Z-up -> Y-up axis conversion and the reference's blue default material.
Deliberate differences (format level, not geometry): no Draco compression (KHR_draco_mesh_compression needs the Draco
encoder Blender bundles), no vertex normals unless asked for, generator string of this package.  `load_glb` reads back
what these writers (and any uncompressed single-buffer GLB with float32 positions) contain.
"""
from __future__ import annotations

import json
import os
import struct
from pathlib import Path
from typing import Dict, List, Optional, Tuple, Union

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


# ---------------------------------------------------------------------------------------------------------------- glTF 2.0
_GLB_MAGIC, _CHUNK_JSON, _CHUNK_BIN = 0x46546C67, 0x4E4F534A, 0x004E4942
_FLOAT, _UINT32, _ARRAY_BUFFER, _ELEMENT_ARRAY_BUFFER = 5126, 5125, 34962, 34963


def _host(x, dtype) -> np.ndarray:
    a = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
    return np.ascontiguousarray(a, dtype=dtype)


class _GlbBuilder:
    """Accumulates accessors over ONE binary buffer (every view 4-byte aligned) and serialises the container."""

    def __init__(self):
        self.bin = bytearray()
        self.views: List[Dict] = []
        self.accessors: List[Dict] = []

    def add(self, arr: np.ndarray, kind: str, target: Optional[int] = None, minmax: bool = False) -> int:
        arr = np.ascontiguousarray(arr)
        ctype = {np.dtype(np.float32): _FLOAT, np.dtype(np.uint32): _UINT32}[arr.dtype]
        while len(self.bin) % 4:
            self.bin.append(0)
        view = {"buffer": 0, "byteOffset": len(self.bin), "byteLength": arr.nbytes}
        if target is not None:
            view["target"] = target
        self.bin += arr.tobytes()
        self.views.append(view)
        acc = {"bufferView": len(self.views) - 1, "componentType": ctype, "count": int(arr.shape[0] if kind != "SCALAR" else arr.size),
               "type": kind}
        if minmax:
            flat = arr.reshape(acc["count"], -1)
            acc["min"], acc["max"] = [float(v) for v in flat.min(axis=0)], [float(v) for v in flat.max(axis=0)]
        self.accessors.append(acc)
        return len(self.accessors) - 1

    def write(self, path: Union[str, Path], gltf: Dict) -> Path:
        while len(self.bin) % 4:
            self.bin.append(0)
        gltf = dict(gltf)
        gltf["buffers"] = [{"byteLength": len(self.bin)}]
        gltf["bufferViews"], gltf["accessors"] = self.views, self.accessors
        js = json.dumps(gltf, separators=(",", ":")).encode("utf-8")
        js += b" " * (-len(js) % 4)
        total = 12 + 8 + len(js) + 8 + len(self.bin)
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        with open(path, "wb") as f:
            f.write(struct.pack("<III", _GLB_MAGIC, 2, total))
            f.write(struct.pack("<II", len(js), _CHUNK_JSON)); f.write(js)
            f.write(struct.pack("<II", len(self.bin), _CHUNK_BIN)); f.write(bytes(self.bin))
        return path


def _check_mesh(v: np.ndarray, f: np.ndarray) -> None:
    if v.ndim != 2 or v.shape[1] != 3 or v.shape[0] == 0:
        raise ValueError(f"vertices must be (V >= 1, 3), got {v.shape}")
    if f.ndim != 2 or f.shape[1] != 3:
        raise ValueError(f"faces must be (F, 3), got {f.shape}")
    if f.size and int(f.max()) >= v.shape[0]:
        raise ValueError("faces index vertices that do not exist")
    if not np.isfinite(v).all():
        raise ValueError("vertices contain nan or inf")


def vertex_normals(vertices: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """Area-weighted vertex normals (unit length; a vertex no face touches gets +Y)."""
    v, f = vertices.astype(np.float64), faces.astype(np.int64)
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    n = np.zeros_like(v)
    for k in range(3):
        np.add.at(n, f[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    n = np.where(ln > 0, n / np.where(ln > 0, ln, 1), np.array([0.0, 1.0, 0.0]))
    return n.astype(np.float32)


_GENERATOR = "actionmesh_amd.mesh_io"


def save_glb(vertices, faces, path: Union[str, Path], normals: bool = False, name: str = "geometry_0") -> Path:
    """One static triangle mesh -> a .glb (POSITION float32, indices uint32, optional NORMAL)."""
    v, f = _host(vertices, np.float32), _host(faces, np.int64)
    _check_mesh(v, f)
    b = _GlbBuilder()
    attrs = {"POSITION": b.add(v, "VEC3", _ARRAY_BUFFER, minmax=True)}
    if normals:
        attrs["NORMAL"] = b.add(vertex_normals(v, f), "VEC3", _ARRAY_BUFFER)
    prim = {"attributes": attrs, "mode": 4}
    if f.size:
        prim["indices"] = b.add(f.astype(np.uint32).reshape(-1), "SCALAR", _ELEMENT_ARRAY_BUFFER)
    gltf = {"asset": {"version": "2.0", "generator": _GENERATOR}, "scene": 0, "scenes": [{"nodes": [0]}],
            "nodes": [{"name": name, "mesh": 0}], "meshes": [{"name": name, "primitives": [prim]}]}
    return b.write(path, gltf)


def save_meshes(vertices, faces, output_dir: Union[str, Path], normals: bool = False) -> List[Path]:
    """mesh_io.py:109-118 on the Stage-II vertex stack: vertices (T, V, 3) sharing `faces` -> output_dir/mesh_{i:02d}.glb."""
    v = _host(vertices, np.float32)
    if v.ndim != 3 or v.shape[0] == 0:
        raise ValueError(f"need a (T >= 1, V, 3) vertex stack, got {v.shape}")
    os.makedirs(output_dir, exist_ok=True)
    return [save_glb(v[i], faces, Path(output_dir) / f"mesh_{i:02d}.glb", normals=normals) for i in range(v.shape[0])]


def _carry_appearance(b: "_GlbBuilder", input_glb: Union[str, Path], n_vertices: int) -> Tuple[Dict, Optional[int], Optional[int]]:
    """Materials / textures / images / samplers and TEXCOORD_0 of `input_glb`'s first triangle primitive, re-based onto builder `b`
    (what the reference keeps by importing the textured GLB into Blender, glb_export.py:160-185).  Returns (top-level glTF entries,
    TEXCOORD_0 accessor in `b` or None, material index or None)."""
    gltf, blob = read_glb(input_glb)
    prim = next((p for m in gltf.get("meshes", []) for p in m["primitives"] if p.get("mode", 4) == 4 and "POSITION" in p["attributes"]), None)
    if prim is None:
        raise ValueError(f"No mesh found in input GLB {input_glb}")
    if "extensions" in prim and "KHR_draco_mesh_compression" in prim["extensions"]:
        raise ValueError(f"{input_glb}: Draco-compressed geometry is not supported")
    count = gltf["accessors"][prim["attributes"]["POSITION"]]["count"]
    if count != n_vertices:
        raise ValueError(f"Vertex count mismatch. Mesh has {count} vertices, deformations have {n_vertices} vertices")
    uv = None
    if "TEXCOORD_0" in prim["attributes"]:
        uv = b.add(np.ascontiguousarray(read_accessor(gltf, blob, prim["attributes"]["TEXCOORD_0"]), dtype=np.float32), "VEC2", _ARRAY_BUFFER)
    extra: Dict = {}
    images = []
    for img in gltf.get("images", []):
        img = dict(img)
        if "bufferView" in img:                                  # embedded image bytes move into the new buffer
            view = gltf["bufferViews"][img["bufferView"]]
            data = blob[view.get("byteOffset", 0):view.get("byteOffset", 0) + view["byteLength"]]
            while len(b.bin) % 4:
                b.bin.append(0)
            b.views.append({"buffer": 0, "byteOffset": len(b.bin), "byteLength": len(data)})
            b.bin += data
            img["bufferView"] = len(b.views) - 1
        images.append(img)
    if images:
        extra["images"] = images
    for key in ("materials", "textures", "samplers", "extensionsUsed"):
        if key in gltf:
            extra[key] = gltf[key]
    return extra, uv, prim.get("material")


def create_animated_glb(vertices_npy, faces_npy, output_glb: Union[str, Path], blender_path: Optional[str] = None, fps: int = 24,
                        export_normals: bool = False, input_glb: Optional[Union[str, Path]] = None) -> int:
    """glb_export.py:18-87 / 142-284 without Blender; same arguments (`blender_path` is accepted and ignored), returns 0 like a
    successful Blender run.  `vertices_npy` / `faces_npy`: the arrays `save_deformation` wrote (paths or arrays; (T, V, 3) in the
    reference's Blender-frame convention, (F, 3)).  Writes ONE mesh "AnimatedMesh": base geometry = frame 0, a morph target
    "Frame_i" per frame (displacement from the base), weights keyed 1 at frame i and 0 at its neighbours (LINEAR interpolation), time =
    frame / fps, Blender's axis conversion (x, y, z) -> (x, z, -y).  Material: the reference's blue default, or - with `input_glb` -
    the materials / textures / texture coordinates of that file's mesh (vertex counts must match, as in the reference)."""
    v = np.load(vertices_npy) if isinstance(vertices_npy, (str, os.PathLike)) else _host(vertices_npy, np.float32)
    f = np.load(faces_npy) if isinstance(faces_npy, (str, os.PathLike)) else _host(faces_npy, np.int64)
    v = np.ascontiguousarray(v, dtype=np.float32)
    if v.ndim != 3 or v.shape[0] == 0:
        raise ValueError(f"need a (T >= 1, V, 3) vertex stack, got {v.shape}")
    if fps <= 0:
        raise ValueError("fps must be positive")
    T = v.shape[0]
    yup = np.ascontiguousarray(np.stack([v[..., 0], v[..., 2], -v[..., 1]], axis=-1))      # Blender Z-up -> glTF Y-up
    f = np.asarray(f, dtype=np.int64)
    _check_mesh(yup[0], f)
    b = _GlbBuilder()
    attrs = {"POSITION": b.add(yup[0], "VEC3", _ARRAY_BUFFER, minmax=True)}
    if export_normals:
        attrs["NORMAL"] = b.add(vertex_normals(yup[0], f), "VEC3", _ARRAY_BUFFER)
    appearance: Dict = {"materials": [{"name": "BlueMaterial", "pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.4, 0.8, 1.0],
                                                                                     "metallicFactor": 0.1, "roughnessFactor": 0.4}}]}
    material: Optional[int] = 0
    if input_glb is not None:
        appearance, uv, material = _carry_appearance(b, input_glb, v.shape[1])
        if uv is not None:
            attrs["TEXCOORD_0"] = uv
    prim = {"attributes": attrs, "mode": 4,
            "targets": [{"POSITION": b.add(yup[i] - yup[0], "VEC3", _ARRAY_BUFFER, minmax=True)} for i in range(T)]}
    if material is not None and "materials" in appearance:
        prim["material"] = material
    if f.size:
        prim["indices"] = b.add(f.astype(np.uint32).reshape(-1), "SCALAR", _ELEMENT_ARRAY_BUFFER)
    names = [f"Frame_{i}" for i in range(T)]
    w0 = [1.0] + [0.0] * (T - 1)
    times = b.add((np.arange(T, dtype=np.float32) / np.float32(fps)), "SCALAR", minmax=True)
    weights = b.add(np.eye(T, dtype=np.float32).reshape(-1), "SCALAR")
    gltf = {
        "asset": {"version": "2.0", "generator": _GENERATOR}, "scene": 0, "scenes": [{"name": "Scene", "nodes": [0]}],
        "nodes": [{"name": "AnimatedMesh", "mesh": 0}],
        "meshes": [{"name": "AnimatedMesh", "primitives": [prim], "weights": w0, "extras": {"targetNames": names}}],
        "animations": [{"name": "KeyAction", "samplers": [{"input": times, "output": weights, "interpolation": "LINEAR"}],
                        "channels": [{"sampler": 0, "target": {"node": 0, "path": "weights"}}]}],
    }
    gltf.update(appearance)
    b.write(output_glb, gltf)
    return 0


def read_glb(path: Union[str, Path]) -> Tuple[Dict, bytes]:
    """(glTF JSON, BIN chunk) of a .glb; raises ValueError on a malformed container."""
    data = Path(path).read_bytes()
    if len(data) < 20:
        raise ValueError(f"{path}: not a GLB file")
    magic, version, total = struct.unpack_from("<III", data, 0)
    if magic != _GLB_MAGIC or version != 2 or total != len(data):
        raise ValueError(f"{path}: bad GLB header")
    n, kind = struct.unpack_from("<II", data, 12)
    if kind != _CHUNK_JSON:
        raise ValueError(f"{path}: first chunk is not JSON")
    gltf = json.loads(data[20:20 + n].decode("utf-8"))
    off, blob = 20 + n, b""
    if off < len(data):
        m, kind = struct.unpack_from("<II", data, off)
        if kind != _CHUNK_BIN:
            raise ValueError(f"{path}: second chunk is not BIN")
        blob = data[off + 8:off + 8 + m]
    return gltf, blob


def read_accessor(gltf: Dict, blob: bytes, index: int) -> np.ndarray:
    acc = gltf["accessors"][index]
    view = gltf["bufferViews"][acc["bufferView"]]
    dtype = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}[acc["componentType"]]
    ncomp = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}[acc["type"]]
    if "byteStride" in view and view["byteStride"] not in (0, np.dtype(dtype).itemsize * ncomp):
        raise ValueError("interleaved buffer views are not supported")
    start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    a = np.frombuffer(blob, dtype=dtype, count=acc["count"] * ncomp, offset=start)
    return a.reshape(acc["count"], ncomp) if ncomp > 1 else a


def load_glb(path: Union[str, Path]) -> Tuple[np.ndarray, np.ndarray]:
    """mesh_io.py:17-40 as arrays: every triangle primitive of the file concatenated -> (vertices (V, 3) float32, faces (F, 3) int64).
    Node transforms are not applied (the writers above emit none).  Raises ValueError if the file contains no geometry."""
    gltf, blob = read_glb(path)
    vs, fs, base = [], [], 0
    for mesh in gltf.get("meshes", []):
        for prim in mesh["primitives"]:
            if prim.get("mode", 4) != 4 or "POSITION" not in prim["attributes"]:
                continue
            if "extensions" in prim and "KHR_draco_mesh_compression" in prim["extensions"]:
                raise ValueError(f"{path}: Draco-compressed geometry is not supported")
            v = read_accessor(gltf, blob, prim["attributes"]["POSITION"]).astype(np.float32)
            f = (read_accessor(gltf, blob, prim["indices"]).astype(np.int64).reshape(-1, 3) if "indices" in prim
                 else np.arange(len(v), dtype=np.int64).reshape(-1, 3))
            vs.append(v); fs.append(f + base); base += len(v)
    if not vs:
        raise ValueError(f"No mesh geometry found in {path}")
    return np.concatenate(vs), np.concatenate(fs)
