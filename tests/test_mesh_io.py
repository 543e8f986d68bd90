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
