"""Generate tests/golden/windows.json from the REFERENCE's own window-chunking and latent-bank code.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (needs /root/reference):

    python oracle/make_golden_windows.py

Pins (the callers of the Stage-I hot path):
  * actionmesh/model/utils/timesteps.py: chunk_right / chunk_left / chunk_from over a sweep of
    (start, total, size, slide), including the shipped (size 16, slide 15) setting;
  * actionmesh/model/utils/storage.py: LatentBank.update / get / get_ordered semantics (first write wins unless
    replace=True, eps-matching of float timesteps, zero latents + mask 0 for missing timesteps).
  * actionmesh/model/utils/embeddings.py: get_scaling / apply_scaling / get_n_subdivisions / interpolate_timesteps, the
    timestep arithmetic of the Stage-II window loop (pipeline.py:553-566).
`trimesh` (imported by storage.py for the MeshBank) is not installed offline: an empty stand-in module is registered.
"""
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
if "trimesh" not in sys.modules:
    tm = types.ModuleType("trimesh")
    tm.Trimesh = type("Trimesh", (), {})
    sys.modules["trimesh"] = tm

from actionmesh.model.utils.embeddings import (apply_scaling, get_n_subdivisions, get_scaling,  # noqa: E402  (reference)
                                               interpolate_timesteps)
from actionmesh.model.utils.storage import LatentBank  # noqa: E402  (reference)
from actionmesh.model.utils.timesteps import chunk_from, chunk_left, chunk_right  # noqa: E402

out = {"chunk_from": [], "chunk_right": [], "chunk_left": [], "bank": []}
for size, slide in ((16, 15), (4, 3), (4, 2), (5, 5), (6, 1), (3, 2)):
    for total in sorted({size, size + 1, 2 * size - 1, 2 * size, 2 * size + 3, 31, 32, 47}):
        if total < size:
            continue
        for start in sorted({0, 1, total // 3, total // 2, total - 2, total - 1}):
            if 0 <= start < total:
                out["chunk_from"].append({"start": start, "total": total, "size": size, "slide": slide,
                                          "chunks": [c.tolist() for c in chunk_from(start, total, size, slide)]})
    for (s, e) in ((0, 10), (3, 17), (0, size), (2, 2 + size + 1), (0, 40)):
        if e - s >= 1:
            out["chunk_right"].append({"start": s, "end": e, "size": size, "slide": slide,
                                       "chunks": [c.tolist() for c in chunk_right(s, e, size, slide)]})
            out["chunk_left"].append({"start": s, "end": e, "size": size, "slide": slide,
                                      "chunks": [c.tolist() for c in chunk_left(s, e, size, slide)]})

# LatentBank scenario: a scripted sequence of operations and what the reference returns after each query
g = torch.Generator().manual_seed(3)
dims = (3, 2)
bank = LatentBank(empty_dims=dims)
script = [
    ("update", [2.0], False), ("get", [0.0, 2.0, 5.0]), ("update", [0.0, 1.0, 2.0, 3.0], False),
    ("get", [3.0, 2.0, 1.0, 0.0, 4.0]), ("update", [2.0, 4.0], True), ("get", [2.0, 4.0, 2.000001, 2.1]),
    ("ordered",), ("update", [7.0, 6.0, 5.0], False), ("ordered",),
]
for op in script:
    if op[0] == "update":
        ts = torch.tensor(op[1])
        lat = torch.randn((len(op[1]),) + dims, generator=g)
        bank.update(ts, lat[None], replace=op[2])             # leading batch dim like the pipeline passes it
        out["bank"].append({"op": "update", "timesteps": op[1], "latents": lat.tolist(), "replace": op[2]})
    elif op[0] == "get":
        lat, mask = bank.get(torch.tensor(op[1]), device="cpu", add_batch_dim=True)
        out["bank"].append({"op": "get", "timesteps": op[1], "latents": lat.tolist(), "mask": mask.tolist()})
    else:
        lat, ts = bank.get_ordered()
        out["bank"].append({"op": "ordered", "latents": lat.tolist(), "timesteps": ts.tolist()})

# Stage-II timestep arithmetic: what pipeline.py:553-566 computes for a window of (possibly unordered) timesteps
out["scaling"] = []
for ts in ([0.0, 1.0, 2.0, 3.0], [5.0, 4.0, 3.0, 2.0, 1.0, 0.0], [7.0, 3.0, 4.0, 5.0, 6.0], [0.0, 2.0, 4.0, 6.0],
           [list(range(16))][0], [15.0] + [float(v) for v in range(15)], [10.0, 11.0], [2.5, 3.5, 4.5]):
    w = torch.tensor([[float(v) for v in ts]])
    for level in (1, 2, 3):
        o = interpolate_timesteps(w, subsampling_level=level, device="cpu", drop_first=True)
        o_all = interpolate_timesteps(w, subsampling_level=level, device="cpu", drop_first=False)
        t_min, t_range = get_scaling(w)
        out["scaling"].append({"timesteps": w[0].tolist(), "level": level, "n": get_n_subdivisions(w.min().item(), w.max().item(), level),
                               "output": o[0].tolist(), "output_all": o_all[0].tolist(), "t_min": t_min.tolist(),
                               "t_range": t_range.tolist(), "source_alpha": apply_scaling(w[:, 0], t_min, t_range).tolist(),
                               "target_alphas": apply_scaling(o, t_min, t_range)[0].tolist()})

path = os.path.join(ROOT, "tests", "golden", "windows.json")
with open(path, "w") as f:
    json.dump(out, f)
print("wrote", path, {k: len(v) for k, v in out.items()}, os.path.getsize(path), "bytes")
