"""CPU restatement of the ActionBench Chamfer metrics (reference actionbench/chamfer.py).

TEST INFRASTRUCTURE ONLY (imported by tests/ and the golden generator, never by the product).  The reference searches
with scipy.spatial.KDTree - an EXACT Euclidean nearest-neighbour search in fp64; the restatement is the brute-force
statement of the same result: argmin over fp64 squared distances ((dx*dx) + (dy*dy)) + dz*dz, ties to the lowest index,
distance = sqrt of the minimum.  Pinned: oracle/make_golden_actionbench.py runs the reference's own two functions
(and scipy's KD-tree for the indices) and stores inputs + outputs in tests/golden/actionbench.npz.
"""
import numpy as np


def nearest(points, queries, block: int = 2048):
    """KDTree(points).query(queries) (chamfer.py:45-50, 75-79): (distances float64, indices int64)."""
    p = np.asarray(points, dtype=np.float64)
    q = np.asarray(queries, dtype=np.float64)
    idx = np.empty(len(q), dtype=np.int64)
    d2 = np.empty(len(q), dtype=np.float64)
    for s in range(0, len(q), block):
        qq = q[s:s + block]
        dx = qq[:, None, 0] - p[None, :, 0]
        dy = qq[:, None, 1] - p[None, :, 1]
        dz = qq[:, None, 2] - p[None, :, 2]
        dd = (dx * dx + dy * dy) + dz * dz
        i = dd.argmin(axis=1)                      # first minimum = lowest index
        idx[s:s + block] = i
        d2[s:s + block] = dd[np.arange(len(qq)), i]
    return np.sqrt(d2), idx


def compute_chamfer_score(pred, gt, n: int = 10_000, seed: int = 44) -> float:
    """chamfer.py:13-52"""
    pred, gt = np.asarray(pred), np.asarray(gt)
    rng_pred = np.random.RandomState(seed=seed)
    rng_gt = np.random.RandomState(seed=seed + 1)
    indices_pred = rng_pred.permutation(len(pred))[:n] if 0 < n < len(pred) else np.arange(len(pred))
    indices_gt = rng_gt.permutation(len(gt))[:n] if 0 < n < len(gt) else np.arange(len(gt))
    d1, _ = nearest(pred, gt[indices_gt])
    d2, _ = nearest(gt, pred[indices_pred])
    return float(np.mean(d1) + np.mean(d2))


def compute_motion_chamfer_score(preds, gts) -> float:
    """chamfer.py:55-86 (float32 differences and norms, like the reference's tensors)"""
    preds, gts = np.asarray(preds, dtype=np.float32), np.asarray(gts, dtype=np.float32)
    assert preds.shape[0] == gts.shape[0], "Mismatching number of timesteps"
    _, idx_gt_to_pred = nearest(preds[0], gts[0])
    _, idx_pred_to_gt = nearest(gts[0], preds[0])
    d1 = np.linalg.norm(preds[:, idx_gt_to_pred, :] - gts, axis=-1).mean(axis=0)
    d2 = np.linalg.norm(gts[:, idx_pred_to_gt, :] - preds, axis=-1).mean(axis=0)
    return float(np.mean(d1) + np.mean(d2))
