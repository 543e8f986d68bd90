"""Generate tests/golden/actionbench.npz from the REFERENCE's own actionbench/chamfer.py (compute_chamfer_score,
compute_motion_chamfer_score) and scipy's KD-tree, which that file searches with.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (needs /root/reference):

    python oracle/make_golden_actionbench.py

Cases: a deforming blob sequence (T frames of P predicted / Q ground-truth points, float32) with duplicated points (the tie
rule), a sub-sampled and a full-cloud Chamfer call, and an empty-intersection edge (one-point clouds).
"""
import importlib.util
import os
import sys

import numpy as np
import torch
from scipy.spatial import KDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("ref_chamfer", "/root/reference/actionbench/chamfer.py")
ref = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ref)

rng = np.random.default_rng(11)
T, P, Q = 4, 3000, 2600
base = rng.standard_normal((P, 3)).astype(np.float32)
base /= np.linalg.norm(base, axis=1, keepdims=True)
base *= rng.uniform(0.6, 1.0, (P, 1)).astype(np.float32)
preds = np.stack([base * (1 + 0.05 * t) + 0.02 * t * np.sin(3 * base[:, ::-1]) for t in range(T)]).astype(np.float32)
gt0 = rng.standard_normal((Q, 3)).astype(np.float32)
gt0 /= np.linalg.norm(gt0, axis=1, keepdims=True)
gt0 *= rng.uniform(0.55, 1.05, (Q, 1)).astype(np.float32)
gts = np.stack([gt0 * (1 + 0.045 * t) + 0.025 * t * np.cos(2 * gt0[:, ::-1]) for t in range(T)]).astype(np.float32)
# coincident points: a few ground-truth points sit exactly on predicted points (distance 0).  (Exact duplicates INSIDE a
# cloud would make the answer a tie - scipy's choice there is an implementation detail of its tree - so none are planted.)
gts[:, :5] = preds[:, 100:105]

out = {"preds": preds, "gts": gts}
tp, tg = torch.from_numpy(preds), torch.from_numpy(gts)
out["cd_sub"] = np.float64(ref.compute_chamfer_score(pred=tp[1], gt=tg[1], n=1000, seed=44))
out["cd_full"] = np.float64(ref.compute_chamfer_score(pred=tp[2], gt=tg[2], n=0, seed=44))
out["cd_default"] = np.float64(ref.compute_chamfer_score(pred=tp[0], gt=tg[0]))
out["cd_motion"] = np.float64(ref.compute_motion_chamfer_score(preds=tp, gts=tg))
d, i = KDTree(preds[0]).query(gts[0])
out["nn_dist_gt_to_pred"], out["nn_idx_gt_to_pred"] = d, i.astype(np.int64)
d, i = KDTree(gts[0]).query(preds[0])
out["nn_dist_pred_to_gt"], out["nn_idx_pred_to_gt"] = d, i.astype(np.int64)
one_p, one_g = torch.tensor([[0.25, -1.0, 2.0]]), torch.tensor([[1.25, -1.0, 2.0]])
out["cd_one"] = np.float64(ref.compute_chamfer_score(pred=one_p, gt=one_g))

sys.path.insert(0, ROOT)
from oracle import actionbench_oracle as O  # noqa: E402
dd, ii = O.nearest(preds[0], gts[0])
assert np.array_equal(ii, out["nn_idx_gt_to_pred"]) and np.array_equal(dd, out["nn_dist_gt_to_pred"]), "restatement != scipy KD-tree"
assert O.compute_chamfer_score(preds[1], gts[1], n=1000) == out["cd_sub"]
assert O.compute_motion_chamfer_score(preds, gts) == out["cd_motion"]
path = os.path.join(ROOT, "tests", "golden", "actionbench.npz")
np.savez_compressed(path, **out)
print("wrote", path, {k: (v.shape if hasattr(v, "shape") and v.shape else float(v)) for k, v in out.items()})
