"""Deformation export: bit-exact against what the reference's save_deformation writes."""
import os

import numpy as np
import pytest
import torch

from actionmesh_amd.mesh_io import save_deformation

GOLD = os.path.join(os.path.dirname(__file__), "golden", "deformation.npz")


def test_save_deformation_matches_reference(tmp_path):
    g = np.load(GOLD)
    vp, fp = save_deformation(torch.from_numpy(g["vertices"]), torch.from_numpy(g["faces"]), tmp_path / "run" / "deformations.npy")
    assert vp.name == "deformations_vertices.npy" and fp.name == "deformations_faces.npy"
    v, f = np.load(vp), np.load(fp)
    assert v.dtype == np.float32 and f.dtype == np.int32
    assert np.array_equal(v, g["out_vertices"]) and np.array_equal(f, g["out_faces"])
    # numpy inputs take the same path
    vp2, _ = save_deformation(g["vertices"], g["faces"], tmp_path / "np" / "deformations.npy")
    assert np.array_equal(np.load(vp2), g["out_vertices"])


def test_save_deformation_rejects_bad_input(tmp_path):
    with pytest.raises(ValueError):
        save_deformation(torch.zeros((0, 4, 3)), torch.zeros((2, 3), dtype=torch.int64), tmp_path / "x.npy")
    with pytest.raises(ValueError):
        save_deformation(torch.zeros((2, 4, 3)), torch.tensor([[0, 1, 4]]), tmp_path / "x.npy")
    with pytest.raises(ValueError):
        save_deformation(torch.zeros((2, 4, 2)), torch.tensor([[0, 1, 2]]), tmp_path / "x.npy")


# ------------------------------------------------------------------------------------------------ GLB outputs (glTF 2.0)
from actionmesh_amd import mesh_io as M  # noqa: E402


def _mesh_stack(T=4, V=9, seed=3):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((T, V, 3)).astype(np.float32)
    f = np.array([[0, 1, 2], [2, 3, 4], [4, 5, 6], [6, 7, 8], [8, 0, 4]], dtype=np.int64)
    return v, f


def _validate_container(path):
    """structural rules of the glTF 2.0 / GLB specification the writers must satisfy"""
    raw = open(path, "rb").read()
    assert raw[:4] == b"glTF" and int.from_bytes(raw[4:8], "little") == 2 and int.from_bytes(raw[8:12], "little") == len(raw)
    gltf, blob = M.read_glb(path)
    assert len(raw) % 4 == 0 and len(blob) % 4 == 0
    assert gltf["asset"]["version"] == "2.0" and gltf["buffers"] == [{"byteLength": len(blob)}]
    for view in gltf["bufferViews"]:
        assert view["byteOffset"] % 4 == 0 and view["byteOffset"] + view["byteLength"] <= len(blob)
    for acc in gltf["accessors"]:
        size = {5125: 4, 5126: 4}[acc["componentType"]] * {"SCALAR": 1, "VEC2": 2, "VEC3": 3}[acc["type"]]
        assert acc["count"] * size == gltf["bufferViews"][acc["bufferView"]]["byteLength"]
    for mesh in gltf["meshes"]:
        for prim in mesh["primitives"]:
            pos = gltf["accessors"][prim["attributes"]["POSITION"]]
            assert "min" in pos and "max" in pos and pos["type"] == "VEC3" and pos["componentType"] == 5126      # required by the spec
            assert gltf["bufferViews"][pos["bufferView"]]["target"] == 34962
            if "indices" in prim:
                ia = gltf["accessors"][prim["indices"]]
                assert ia["type"] == "SCALAR" and ia["count"] % 3 == 0 and gltf["bufferViews"][ia["bufferView"]]["target"] == 34963
            for tgt in prim.get("targets", []):
                ta = gltf["accessors"][tgt["POSITION"]]
                assert ta["count"] == pos["count"] and "min" in ta and "max" in ta
    return gltf, blob


def test_save_meshes_roundtrip(tmp_path):
    v, f = _mesh_stack()
    paths = M.save_meshes(torch.from_numpy(v), torch.from_numpy(f), tmp_path / "meshes")
    assert [p.name for p in paths] == [f"mesh_{i:02d}.glb" for i in range(4)]          # mesh_io.py:116 naming
    for i, p in enumerate(paths):
        gltf, _ = _validate_container(p)
        vv, ff = M.load_glb(p)
        assert vv.dtype == np.float32 and np.array_equal(vv, v[i]) and np.array_equal(ff, f)
        pos = gltf["accessors"][gltf["meshes"][0]["primitives"][0]["attributes"]["POSITION"]]
        assert np.allclose(pos["min"], v[i].min(0)) and np.allclose(pos["max"], v[i].max(0))
    p = M.save_glb(v[0], f, tmp_path / "n.glb", normals=True)
    gltf, blob = _validate_container(p)
    n = M.read_accessor(gltf, blob, gltf["meshes"][0]["primitives"][0]["attributes"]["NORMAL"])
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-6)
    # a point cloud without faces is still a valid file; bad input is refused
    M.save_glb(v[0], np.zeros((0, 3), dtype=np.int64), tmp_path / "pts.glb")
    with pytest.raises(ValueError):
        M.save_glb(v[0], np.array([[0, 1, 99]]), tmp_path / "bad.glb")
    with pytest.raises(ValueError):
        M.save_meshes(np.zeros((0, 3, 3), np.float32), f, tmp_path / "none")
    v_bad = v[0].copy(); v_bad[0, 0] = np.nan
    with pytest.raises(ValueError):
        M.save_glb(v_bad, f, tmp_path / "nan.glb")


def test_animated_glb_is_the_shape_key_animation(tmp_path):
    """glb_export.py:234-255: shape key i = frame i, value 1 at frame i and 0 at its neighbours, fps -> key times; the export
    applies Blender's Z-up -> Y-up conversion.  Fed with what save_deformation wrote, the animation must reproduce the
    original vertices in the glTF frame: (x, y, z)_saved = (-v2, v0, v1)  ->  glTF (x, z, -y)_saved = (-v2, v1, -v0)."""
    v, f = _mesh_stack(T=5)
    vp, fp = M.save_deformation(v, f, tmp_path / "deformations.npy")
    out = tmp_path / "animated.glb"
    assert M.create_animated_glb(vertices_npy=str(vp), faces_npy=str(fp), output_glb=out, blender_path="/no/blender/needed", fps=12) == 0
    gltf, blob = _validate_container(out)
    prim = gltf["meshes"][0]["primitives"][0]
    base = M.read_accessor(gltf, blob, prim["attributes"]["POSITION"])
    want = np.stack([-v[..., 2], v[..., 1], -v[..., 0]], axis=-1)
    assert np.array_equal(base, want[0])
    assert len(prim["targets"]) == 5 and gltf["meshes"][0]["extras"]["targetNames"] == [f"Frame_{i}" for i in range(5)]
    anim = gltf["animations"][0]
    assert anim["channels"] == [{"sampler": 0, "target": {"node": 0, "path": "weights"}}] and anim["samplers"][0]["interpolation"] == "LINEAR"
    times = M.read_accessor(gltf, blob, anim["samplers"][0]["input"])
    weights = M.read_accessor(gltf, blob, anim["samplers"][0]["output"]).reshape(5, 5)
    assert np.array_equal(times, np.arange(5, dtype=np.float32) / np.float32(12)) and np.array_equal(weights, np.eye(5, dtype=np.float32))
    deltas = np.stack([M.read_accessor(gltf, blob, t["POSITION"]) for t in prim["targets"]])
    for i in range(5):                           # evaluating the morph at key i gives frame i
        assert np.allclose(base + np.tensordot(weights[i], deltas, axes=1), want[i], atol=1e-6)
    assert np.array_equal(M.read_accessor(gltf, blob, prim["indices"]).reshape(-1, 3), f)
    mat = gltf["materials"][prim["material"]]["pbrMetallicRoughness"]          # glb_export.py:213-222
    assert mat == {"baseColorFactor": [0.2, 0.4, 0.8, 1.0], "metallicFactor": 0.1, "roughnessFactor": 0.4}
    assert gltf["meshes"][0]["weights"] == [1.0, 0.0, 0.0, 0.0, 0.0]
    # arrays instead of paths, normals on request, a single frame
    out2 = tmp_path / "one.glb"
    assert M.create_animated_glb(np.load(vp)[:1], np.load(fp), out2, export_normals=True) == 0
    g2, _ = _validate_container(out2)
    assert "NORMAL" in g2["meshes"][0]["primitives"][0]["attributes"] and len(g2["meshes"][0]["primitives"][0]["targets"]) == 1
    with pytest.raises(ValueError):
        M.create_animated_glb(np.load(vp), np.load(fp), tmp_path / "x.glb", fps=0)
    # input_glb (video_and_3d_to_animated_mesh.py:122-129): the textured anchor's material, image and texture coordinates are kept
    from actionmesh_amd.mesh_io import _GlbBuilder
    b = _GlbBuilder()
    pos = b.add(np.load(vp)[0], "VEC3", 34962, minmax=True)
    uv0 = np.random.default_rng(0).random((v.shape[1], 2)).astype(np.float32)
    uv = b.add(uv0, "VEC2", 34962)
    idx = b.add(f.astype(np.uint32).reshape(-1), "SCALAR", 34963)
    png = bytes(range(37))                                          # opaque image payload (not decoded by anybody here)
    b.views.append({"buffer": 0, "byteOffset": len(b.bin), "byteLength": len(png)}); b.bin += png
    anchor = b.write(tmp_path / "anchor.glb", {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": pos, "TEXCOORD_0": uv}, "indices": idx, "material": 0, "mode": 4}]}],
        "materials": [{"name": "Textured", "pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}],
        "textures": [{"source": 0, "sampler": 0}], "samplers": [{"magFilter": 9729}],
        "images": [{"bufferView": len(b.views) - 1, "mimeType": "image/png"}]})
    out3 = tmp_path / "textured.glb"
    assert M.create_animated_glb(str(vp), str(fp), out3, blender_path=None, fps=8, input_glb=anchor) == 0
    g3, blob3 = _validate_container(out3)
    p3 = g3["meshes"][0]["primitives"][0]
    assert g3["materials"][p3["material"]]["name"] == "Textured" and g3["textures"] == [{"source": 0, "sampler": 0}]
    assert np.array_equal(M.read_accessor(g3, blob3, p3["attributes"]["TEXCOORD_0"]), uv0)
    iv = g3["bufferViews"][g3["images"][0]["bufferView"]]
    assert blob3[iv["byteOffset"]:iv["byteOffset"] + iv["byteLength"]] == png and len(p3["targets"]) == 5
    with pytest.raises(ValueError, match="Vertex count mismatch"):              # glb_export.py:176-184
        M.create_animated_glb(np.load(vp)[:, :-1], np.array([[0, 1, 2]]), tmp_path / "y.glb", input_glb=anchor)


def test_load_glb_errors(tmp_path):
    bad = tmp_path / "bad.glb"
    bad.write_bytes(b"not a glb at all, just bytes......")
    with pytest.raises(ValueError):
        M.load_glb(bad)
    v, f = _mesh_stack()
    p = M.save_glb(v[0], f, tmp_path / "ok.glb")
    gltf, blob = M.read_glb(p)
    gltf["meshes"] = []                                     # "No mesh geometry found" (mesh_io.py:37-38)
    b = M._GlbBuilder(); b.bin = bytearray(blob); b.views, b.accessors = gltf["bufferViews"], gltf["accessors"]
    empty = b.write(tmp_path / "empty.glb", {k: v_ for k, v_ in gltf.items() if k not in ("buffers", "bufferViews", "accessors")})
    with pytest.raises(ValueError, match="No mesh geometry"):
        M.load_glb(empty)
    # two primitives are concatenated with re-based indices (mesh_io.py:32-39)
    gltf, blob = M.read_glb(p)
    gltf["meshes"].append(gltf["meshes"][0])
    b = M._GlbBuilder(); b.bin = bytearray(blob); b.views, b.accessors = gltf["bufferViews"], gltf["accessors"]
    two = b.write(tmp_path / "two.glb", {k: v_ for k, v_ in gltf.items() if k not in ("buffers", "bufferViews", "accessors")})
    vv, ff = M.load_glb(two)
    assert len(vv) == 2 * len(v[0]) and np.array_equal(ff[len(f):], f + len(v[0]))
