"""Generate tests/golden/deformation.npz from the REFERENCE's own save_deformation (actionmesh/io/mesh_io.py:43-106).

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (needs /root/reference):

    python oracle/make_golden_mesh_io.py

`trimesh` is not installed offline; save_deformation only reads `.vertices` / `.faces` of the objects it is handed, so a
stand-in module with a bare Trimesh class carrying those two arrays is registered.  Stored: the input vertex stack and
faces, and the two arrays the reference writes.
"""
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
if "trimesh" not in sys.modules:
    tm = types.ModuleType("trimesh")

    class Trimesh:
        def __init__(self, vertices, faces):
            self.vertices, self.faces = vertices, faces
    tm.Trimesh = Trimesh
    tm.Scene = type("Scene", (), {})
    sys.modules["trimesh"] = tm
import trimesh  # noqa: E402

import importlib.util  # noqa: E402

# the module file itself: actionmesh/io/__init__.py pulls in cv2 (video loading), which is not installed offline either
_spec = importlib.util.spec_from_file_location("ref_mesh_io", "/root/reference/actionmesh/io/mesh_io.py")
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
save_deformation = _mod.save_deformation  # (reference)

rng = np.random.default_rng(7)
T, V, Fc = 5, 37, 60
verts = rng.standard_normal((T, V, 3))                      # float64, like trimesh holds them
faces = rng.integers(0, V, size=(Fc, 3)).astype(np.int64)
with tempfile.TemporaryDirectory() as d:
    vp, fp = save_deformation([trimesh.Trimesh(verts[t], faces) for t in range(T)], os.path.join(d, "sub", "deformations.npy"))
    assert vp.name == "deformations_vertices.npy" and fp.name == "deformations_faces.npy"
    out_v, out_f = np.load(vp), np.load(fp)
path = os.path.join(ROOT, "tests", "golden", "deformation.npz")
np.savez_compressed(path, vertices=verts, faces=faces, out_vertices=out_v, out_faces=out_f)
print("wrote", path, out_v.shape, out_v.dtype, out_f.shape, out_f.dtype)
