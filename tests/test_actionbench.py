"""ActionBench Chamfer metrics: the exact nearest-neighbour kernel (am_nn_search) and the host mirrors of
actionbench/chamfer.py against fixtures produced by the reference's own functions + scipy's KD-tree
(oracle/make_golden_actionbench.py), and the CPU restatement against the same fixtures."""
import ctypes
import os

import numpy as np
import pytest
import torch

from actionmesh_amd import _lib, actionbench as AB, ops
from oracle import actionbench_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "actionbench.npz")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


# ---------------------------------------------------------------- CPU: oracle pinned to the reference, C-ABI argument checks
def test_oracle_reproduces_reference(gold):
    d, i = O.nearest(gold["preds"][0], gold["gts"][0])
    assert np.array_equal(i, gold["nn_idx_gt_to_pred"]) and np.array_equal(d, gold["nn_dist_gt_to_pred"])
    d, i = O.nearest(gold["gts"][0], gold["preds"][0])
    assert np.array_equal(i, gold["nn_idx_pred_to_gt"]) and np.array_equal(d, gold["nn_dist_pred_to_gt"])
    assert O.compute_chamfer_score(gold["preds"][1], gold["gts"][1], n=1000, seed=44) == gold["cd_sub"]
    assert O.compute_chamfer_score(gold["preds"][2], gold["gts"][2], n=0) == gold["cd_full"]
    assert O.compute_chamfer_score(gold["preds"][0], gold["gts"][0]) == gold["cd_default"]
    assert O.compute_motion_chamfer_score(gold["preds"], gold["gts"]) == gold["cd_motion"]
    assert O.compute_chamfer_score(np.float32([[0.25, -1, 2]]), np.float32([[1.25, -1, 2]])) == gold["cd_one"] == 2.0


def test_nn_argument_validation_without_gpu():
    lib = _lib.lib()
    assert lib.am_nn_search(None, None, 0, None) != 0 and b"null" in lib.am_last_error()
    a = _lib.AmNnArgs()
    a.batch, a.n_points, a.n_queries = 1, 0, 4
    assert lib.am_nn_search(ctypes.byref(a), None, 0, None) != 0 and b"empty" in lib.am_last_error()
    assert lib.am_nn_workspace_bytes(0, 5, 1, 1) == 0
    assert lib.am_nn_workspace_bytes(100_000, 1_000_000, 1, 1) == 0               # enough query blocks: one split
    assert lib.am_nn_workspace_bytes(100_000, 1000, 1, 1) % (12 * 1000) == 0       # split over the points: (8 + 4) B per partial
    with pytest.raises(RuntimeError):      # no CPU path
        ops.nearest_neighbors(torch.zeros(4, 3), torch.zeros(2, 3))


# ---------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_nn_matches_kdtree_bit_for_bit(gold, dev):
    p, g = torch.from_numpy(gold["preds"][0]).to(dev), torch.from_numpy(gold["gts"][0]).to(dev)
    idx, d2 = ops.nearest_neighbors(p, g)
    assert idx.dtype == torch.int32 and d2.dtype == torch.float64
    assert np.array_equal(idx.cpu().numpy(), gold["nn_idx_gt_to_pred"])
    assert np.array_equal(np.sqrt(d2.cpu().numpy()), gold["nn_dist_gt_to_pred"])
    d, i = AB.nearest(gold["gts"][0], gold["preds"][0])
    assert np.array_equal(i, gold["nn_idx_pred_to_gt"]) and np.array_equal(d, gold["nn_dist_pred_to_gt"])
    assert d[100:105].max() == 0.0 and np.array_equal(i[100:105], np.arange(5))      # the planted coincident points


@pytest.mark.gpu
def test_chamfer_scores_match_reference(gold, dev):
    tp, tg = torch.from_numpy(gold["preds"]), torch.from_numpy(gold["gts"])
    assert AB.compute_chamfer_score(pred=tp[1], gt=tg[1], n=1000, seed=44) == gold["cd_sub"]
    assert AB.compute_chamfer_score(pred=tp[2], gt=tg[2], n=0, seed=44) == gold["cd_full"]
    assert AB.compute_chamfer_score(pred=tp[0].to(dev), gt=tg[0].to(dev)) == gold["cd_default"]      # device inputs too
    assert AB.compute_motion_chamfer_score(preds=tp, gts=tg) == gold["cd_motion"]
    assert AB.compute_chamfer_score(torch.tensor([[0.25, -1.0, 2.0]]), torch.tensor([[1.25, -1.0, 2.0]])) == 2.0
    with pytest.raises(AssertionError):
        AB.compute_motion_chamfer_score(tp[:2], tg[:3])


@pytest.mark.gpu
@pytest.mark.parametrize("P,Q,B", [(1, 1, 1), (513, 7, 1), (5000, 300, 1), (2049, 1025, 3), (20000, 64, 24), (700, 70000, 1)])
def test_nn_shapes_splits_and_batches(dev, P, Q, B):
    """every launch plan (1 or 4 queries per thread, 1..n point splits, batches, a cloud shared by the batch) against a
    torch fp64 brute force; ties (duplicated points) resolve to the lowest index"""
    g = torch.Generator().manual_seed(P * 31 + Q)
    pts = torch.randn((B, P, 3), generator=g)
    if P > 10:
        pts[:, P // 2] = pts[:, 3]          # an exact duplicate: index 3 must win
    qry = torch.randn((B, Q, 3), generator=g)
    if P > 10:
        qry[:, 0] = pts[:, 3]
    def brute(q, p):            # fp64, the kernel's summation order (torch.cdist switches to a matmul formulation)
        d = q.double()[:, :, None, :] - p.double()[:, None, :, :]
        return ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).min(dim=2)
    idx, d2 = ops.nearest_neighbors(pts.to(dev), qry.to(dev))
    idx, d2 = idx.cpu().long(), d2.cpu()
    assert idx.shape == (B, Q)
    got = (qry.double() - torch.gather(pts.double(), 1, idx[..., None].expand(-1, -1, 3))).pow(2)
    got = (got[..., 0] + got[..., 1]) + got[..., 2]
    assert torch.equal(got, d2)                                   # the reported d2 is the distance to the reported index, bitwise
    sub = torch.arange(0, Q, max(1, (P * Q * B) // 4_000_000))
    ref = brute(qry[:, sub], pts)
    assert torch.equal(d2[:, sub], ref.values) and torch.equal(idx[:, sub], ref.indices)      # torch.min: first minimum, like the kernel
    if P > 10:
        assert (idx[:, 0] == 3).all() and (d2[:, 0] == 0).all()
    # shared cloud: 2-D points beside batched queries
    idx_s, d2_s = ops.nearest_neighbors(pts[0].contiguous().to(dev), qry.to(dev))
    assert torch.equal(idx_s[0].cpu().long(), idx[0]) and torch.equal(d2_s[0].cpu(), d2[0])
    # fp32 path (the ICP inner loop): same neighbour up to fp32 resolution of the distance
    idx_f, d2_f = ops.nearest_neighbors(pts.to(dev), qry.to(dev), precise=False)
    assert d2_f.dtype == torch.float32
    assert torch.allclose(d2_f.cpu().double(), d2, rtol=1e-5, atol=1e-10)


@pytest.mark.gpu
def test_nn_full_size_properties(dev):
    """the reference's metric size (100 000 x 100 000): self-search is the identity with distance 0, a permuted copy finds
    the inverse permutation, and the symmetric Chamfer distance of a cloud with itself is 0"""
    g = torch.Generator().manual_seed(5)
    pts = torch.randn((100_000, 3), generator=g).to(dev)
    idx, d2 = ops.nearest_neighbors(pts, pts)
    assert torch.equal(idx.long().cpu(), torch.arange(100_000)) and float(d2.max()) == 0.0
    perm = torch.randperm(100_000, generator=g).to(dev)
    idx, d2 = ops.nearest_neighbors(pts[perm].contiguous(), pts)
    assert torch.equal(perm[idx.long()], torch.arange(100_000, device=dev)) and float(d2.max()) == 0.0
    assert AB.compute_chamfer_score(pts, pts) == 0.0
    assert AB.compute_motion_chamfer_score(pts[None].repeat(3, 1, 1), pts[None].repeat(3, 1, 1)) == 0.0


@pytest.mark.gpu
def test_nn_rejects_bad_shapes(dev):
    with pytest.raises(ValueError):
        ops.nearest_neighbors(torch.zeros((0, 3), device=dev), torch.zeros((2, 3), device=dev))
    with pytest.raises(ValueError):
        ops.nearest_neighbors(torch.zeros((4, 2), device=dev), torch.zeros((2, 3), device=dev))
    with pytest.raises(ValueError):
        ops.nearest_neighbors(torch.zeros((2, 4, 3), device=dev), torch.zeros((3, 2, 3), device=dev))
    with pytest.raises(TypeError):
        ops.nearest_neighbors(torch.zeros((4, 3), device=dev, dtype=torch.float64), torch.zeros((2, 3), device=dev))


# ---------------------------------------------------------------- ICP half (restated pytorch3d formulas: parity unpinned)
def test_icp_formulas_cpu():
    R = AB.canonical_rotation_matrices()
    assert torch.allclose(R, O.canonical_rotation_matrices(), atol=1e-7)
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(24, 3, 3), atol=1e-6) and torch.allclose(torch.det(R), torch.ones(24))
    assert torch.allclose(R.abs().sum(-1), torch.ones(24, 3), atol=1e-6)           # axis-aligned: signed permutation matrices
    d6 = torch.randn(7, 6, generator=torch.Generator().manual_seed(0))
    M6 = AB.rotation_6d_to_matrix(d6)
    assert torch.allclose(M6, O.rotation_6d_to_matrix(d6), atol=1e-6) and torch.allclose(M6 @ M6.transpose(1, 2), torch.eye(3).expand(7, 3, 3), atol=1e-5)
    assert torch.allclose(AB.rotation_6d_to_matrix(torch.tensor([[1.0, 0, 0, 0, 1.0, 0]])), torch.eye(3)[None])
    t = AB.ScaleRotateTranslate(R[5], torch.tensor([1.0, 2, 3]), torch.tensor([2.0, 1, 0.5]))
    p = torch.randn(11, 3)
    assert torch.allclose(t.transform_points(p), (torch.tensor([2.0, 1, 0.5]) * p) @ R[5] + torch.tensor([1.0, 2, 3]))
    st = t.stack(t, t)
    assert len(st) == 3 and torch.allclose(st.transform_points(p[None].repeat(3, 1, 1))[2], t.transform_points(p))
    pc = torch.arange(5 * 20 * 3, dtype=torch.float32).reshape(5, 20, 3)
    sub = AB.sample_point_cloud(pc, 6, seed=44)                                   # sample_point_cloud.py:34-36
    assert torch.equal(sub, pc[:, torch.from_numpy(np.random.RandomState(44).permutation(20)[:6]).long()])
    assert AB.sample_point_cloud(pc, 20) is pc


@pytest.mark.gpu
def test_chamfer_loss_and_gradient_match_autograd_through_min(dev):
    g = torch.Generator().manual_seed(2)
    x = torch.randn((3, 400, 3), generator=g, requires_grad=True)
    y = torch.randn((3, 350, 3), generator=g, requires_grad=True)
    ref = O.chamfer_distance_sq(x, y)
    ref.sum().backward()
    xd, yd = x.detach().to(dev).requires_grad_(), y.detach().to(dev).requires_grad_()
    got = AB.chamfer_distance_sq(xd, yd)
    got.sum().backward()
    assert torch.allclose(got.cpu(), ref, rtol=1e-5) and torch.allclose(xd.grad.cpu(), x.grad, atol=1e-6) and torch.allclose(yd.grad.cpu(), y.grad, atol=1e-6)


@pytest.mark.gpu
def test_gradient_icp_recovers_a_known_similarity(dev):
    """an asymmetric cloud, rotated by a canonical orientation plus 12 degrees, scaled anisotropically and shifted: the ICP must
    bring the aligned Chamfer distance to the noise floor; the CPU restatement run on the same data lands at the same loss"""
    g = torch.Generator().manual_seed(4)
    pred = torch.randn((500, 3), generator=g) * torch.tensor([1.0, 0.6, 0.3]) + torch.tensor([0.3, 0.0, 0.0]) * torch.randn((500, 1), generator=g).abs()
    R_true = AB.canonical_rotation_matrices()[9] @ AB.euler_angles_to_matrix(torch.tensor([0.21, -0.1, 0.05]))
    gt = (torch.tensor([1.1, 0.95, 1.05]) * pred) @ R_true + torch.tensor([0.05, -0.02, 0.03])
    gt = gt[torch.randperm(500, generator=g)]
    icp = AB.gradient_icp(pc_pred=pred.to(dev), pc_gt=gt.to(dev), lr=0.01, n_iter=120)
    aligned = icp.transform_points(pred.to(dev))
    before = AB.compute_chamfer_score(pred, gt)
    after = AB.compute_chamfer_score(aligned, gt)
    assert after < 0.1 * before and icp.loss < 2e-3, (before, after, icp.loss)
    Rc, Tc, sc, loss_c = O.gradient_icp(pred, gt, lr=0.01, n_iter=120)
    assert abs(icp.loss - loss_c) <= 0.25 * max(loss_c, 1e-4), (icp.loss, loss_c)          # two fp32 runs of a 120-step optimiser
    cd3, cd4, cdm = AB.compute_chamfer_3d_4d(gt[None].repeat(2, 1, 1), pred[None].repeat(2, 1, 1), device=dev, is_4D=True,
                                             pred_pc_4D=pred[None].repeat(2, 1, 1), n_pts_icp=300, n_iter=60)
    assert 0 <= cd3 < before and 0 <= cd4 < before and cdm >= 0


def test_sample_meshes_cpu():
    """surface sampling on the vertex stack (torch only: runs on CPU tensors too): points lie on their triangles, the face
    distribution follows the areas, synchronized draws keep the correspondence"""
    v0 = torch.tensor([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [3, 1, 0]])
    f = torch.tensor([[0, 1, 2], [1, 3, 2], [1, 4, 3]])                 # areas 0.5, 0.5, 1.0 in the z = 0 plane
    v = torch.stack([v0, v0 * 2 + torch.tensor([0.0, 0, 1.0])])          # frame 1: scaled and lifted
    pts = AB.sample_meshes(v, f, n_pts=20000, synchronized=True, seed=3)
    assert pts.shape == (2, 20000, 3) and float(pts[0, :, 2].abs().max()) == 0.0
    assert torch.allclose(pts[1], pts[0] * 2 + torch.tensor([0.0, 0, 1.0]), atol=1e-6)              # same (face, barycentric) per point
    frac_right = float((pts[0, :, 0] > 1.0).float().mean())              # the third triangle (area 1.0 of 2.0) is the part with x > 1
    assert abs(frac_right - 0.5) < 0.03, frac_right
    ind = AB.sample_meshes(v, f, n_pts=1000, synchronized=False, seed=3)
    assert not torch.allclose(ind[1], ind[0] * 2 + torch.tensor([0.0, 0, 1.0]))                      # independent draws per frame
    assert torch.equal(AB.sample_meshes(v, f, 1000, True, 3), AB.sample_meshes(v, f, 1000, True, 3))  # seeded
    with pytest.raises(ValueError):
        AB.sample_meshes(v, torch.zeros((0, 3), dtype=torch.long))
