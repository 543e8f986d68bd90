"""ActionBench Chamfer metrics on the device (the end-to-end quality gate of the hot path).

Mirrors of the reference's actionbench/chamfer.py with the same names, arguments and sampling (numpy RandomState
permutations with the reference's seeds); the nearest-neighbour searches - scipy KD-trees over 100 000-point clouds in
the reference - run as one exact brute-force kernel (am_nn_search, fp64 arithmetic in the KD-tree's summation order), so
indices and distances are the reference's bit for bit; the final sqrt / mean reductions stay on the host in numpy exactly
as the reference has them.  No CPU fallback: inputs are moved to `device` (default cuda:0) and the library must be loaded.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import ops


def _dev(t, device) -> torch.Tensor:
    t = torch.as_tensor(t)
    device = torch.device(device) if device is not None else (t.device if t.is_cuda else torch.device("cuda:0"))
    return t.to(device=device, dtype=torch.float32).contiguous()


def nearest(points, queries, device=None) -> Tuple[np.ndarray, np.ndarray]:
    """KDTree(points).query(queries) of chamfer.py: (distances float64, indices int64) as numpy arrays."""
    p, q = _dev(points, device), _dev(queries, device)
    idx, d2 = ops.nearest_neighbors(p, q.to(p.device), precise=True)
    return np.sqrt(d2.cpu().numpy()), idx.cpu().numpy().astype(np.int64)


def compute_chamfer_score(pred, gt, n: int = 10_000, seed: int = 44, device=None) -> float:
    """Symmetric Chamfer distance between two point clouds (actionbench/chamfer.py:13-52).  pred (N, 3), gt (M, 3);
    at most `n` query points per direction, drawn with the reference's RandomState(seed) / RandomState(seed + 1)."""
    pred_t, gt_t = torch.as_tensor(pred), torch.as_tensor(gt)
    rng_pred = np.random.RandomState(seed=seed)
    rng_gt = np.random.RandomState(seed=seed + 1)
    indices_pred = rng_pred.permutation(len(pred_t))[:n] if 0 < n < len(pred_t) else np.arange(len(pred_t))
    indices_gt = rng_gt.permutation(len(gt_t))[:n] if 0 < n < len(gt_t) else np.arange(len(gt_t))
    p, g = _dev(pred_t, device), _dev(gt_t, device)
    g = g.to(p.device)
    d1, _ = nearest(p, g[torch.from_numpy(indices_gt).to(p.device)])
    d2, _ = nearest(g, p[torch.from_numpy(indices_pred).to(p.device)])
    return float(np.mean(d1) + np.mean(d2))


def compute_motion_chamfer_score(preds, gts, device=None) -> float:
    """Motion Chamfer distance over a sequence (actionbench/chamfer.py:55-86): correspondences from the first frame,
    distances over all frames.  preds (T, P, 3), gts (T, Q, 3)."""
    preds_t, gts_t = torch.as_tensor(preds), torch.as_tensor(gts)
    assert preds_t.shape[0] == gts_t.shape[0], "Mismatching number of timesteps"
    p, g = _dev(preds_t, device), _dev(gts_t, device)
    g = g.to(p.device)
    idx_gt_to_pred, _ = ops.nearest_neighbors(p[0].contiguous(), g[0].contiguous(), precise=True)
    idx_pred_to_gt, _ = ops.nearest_neighbors(g[0].contiguous(), p[0].contiguous(), precise=True)
    # the gathers and the fp32 differences on the device; norms and means in numpy like the reference (float32 data)
    diff1 = (p[:, idx_gt_to_pred.long(), :] - g).cpu().numpy()
    diff2 = (g[:, idx_pred_to_gt.long(), :] - p).cpu().numpy()
    d1 = np.linalg.norm(diff1, axis=-1).mean(axis=0)
    d2 = np.linalg.norm(diff2, axis=-1).mean(axis=0)
    return float(np.mean(d1) + np.mean(d2))
