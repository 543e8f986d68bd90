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


# ---------------------------------------------------------------------------------------------------------------- ICP
# actionbench/icp.py restated on CPU torch.  PARITY UNPINNED for this half: the reference computes with pytorch3d
# (chamfer_distance, rotation_6d_to_matrix, euler_angles_to_matrix, Transform3d), which is neither vendored nor installable
# offline, and the reference ships no test vectors for it; the pytorch3d functions are restated from their published
# definitions (pytorch3d v0.7 docs: chamfer_distance = mean squared nearest-neighbour distance in both directions;
# rotation_6d_to_matrix = Gram-Schmidt rows; euler_angles_to_matrix("XYZ") = Rx Ry Rz).
import torch  # noqa: E402


def chamfer_distance_sq(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """(N, P, 3), (N, Q, 3) -> (N,): autograd differentiates straight through the min."""
    d = (x[:, :, None, :] - y[:, None, :, :]).pow(2).sum(-1)
    return d.min(dim=2).values.mean(1) + d.min(dim=1).values.mean(1)


def _rot(axis, a):
    c, s, o, z = torch.cos(a), torch.sin(a), torch.ones_like(a), torch.zeros_like(a)
    m = {"X": (o, z, z, z, c, -s, z, s, c), "Y": (c, z, s, z, o, z, -s, z, c), "Z": (c, -s, z, s, c, z, z, z, o)}[axis]
    return torch.stack(m, -1).reshape(a.shape + (3, 3))


def canonical_rotation_matrices() -> torch.Tensor:
    """icp.py:19-51"""
    d = torch.pi / 180
    azim = torch.tensor([0] * 4 + [90] * 4 + [180] * 4 + [270] * 4 + [0] * 4 + [90] * 4, dtype=torch.float32) * d
    elev = torch.tensor([0] * 16 + [90] * 2 + [-90] * 2 + [90] * 2 + [-90] * 2, dtype=torch.float32) * d
    roll = torch.tensor([0, 90, 180, 270] * 4 + [0, 90] * 4, dtype=torch.float32) * d
    return _rot("X", azim) @ _rot("Y", elev) @ _rot("Z", roll)


def rotation_6d_to_matrix(d6):
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = a1 / a1.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = b2 / b2.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), dim=-2)


def gradient_icp(pc_pred, pc_gt, lr=0.01, n_iter=200):
    """icp.py:54-111 -> (R (1,3,3), T (1,3), s (1,3), best loss): p' = (s * p) @ R + T"""
    with torch.enable_grad():
        R_init = canonical_rotation_matrices()
        n = len(R_init)
        pred, gt = pc_pred[None].expand(n, -1, -1), pc_gt[None].expand(n, -1, -1)
        T = torch.nn.Parameter(torch.zeros(n, 3))
        R6 = torch.nn.Parameter(torch.tensor([[1.0, 0, 0, 0, 1.0, 0]]).repeat(n, 1))
        s = torch.nn.Parameter(torch.ones(n, 3))
        opt = torch.optim.Adam([T, R6, s], lr=lr)
        best_loss, best = float("inf"), None
        for _ in range(n_iter):
            opt.zero_grad()
            R = R_init @ rotation_6d_to_matrix(R6)
            loss = chamfer_distance_sq(s[:, None] * pred @ R + T[:, None], gt)
            loss.mean().backward()
            opt.step()
            m, i = loss.detach().min(0)
            if m.item() < best_loss:
                best_loss = m.item()
                best = (R[i:i + 1].detach().clone(), T[i:i + 1].detach().clone(), s[i:i + 1].detach().clone())
    return best + (best_loss,)
