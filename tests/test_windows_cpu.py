"""Autoregressive window orchestration, host logic on CPU: the product module and the oracle
restatement against fixtures generated from the reference's own timesteps.py / storage.py
(oracle/make_golden_windows.py -> tests/golden/windows.json)."""
import json
import os

import pytest
import torch

from actionmesh_amd import windows as W
from oracle import windows_oracle as WO


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "windows.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("impl", [W, WO], ids=["product", "oracle"])
def test_chunking_matches_reference(gold, impl):
    for c in gold["chunk_from"]:
        got = [x.tolist() for x in impl.chunk_from(c["start"], c["total"], c["size"], c["slide"])]
        assert got == c["chunks"], c
    for name in ("chunk_right", "chunk_left"):
        for c in gold[name]:
            got = [x.tolist() for x in getattr(impl, name)(c["start"], c["end"], c["size"], c["slide"])]
            assert got == c["chunks"], (name, c)


def test_chunk_from_properties():
    """Size-independent properties: every frame is covered, every window has <= size frames, the first window holds
    the anchor first, and every later window overlaps what was generated before it (its conditioning frames)."""
    for size, slide in ((16, 15), (8, 5), (4, 1)):
        for total in (size, size + 1, 3 * size + 2, 100):
            for start in (0, 1, total // 2, total - 1):
                ws = W.chunk_from(start, total, size, slide)
                seen = set()
                for k, w in enumerate(ws):
                    assert len(w) <= size and len(set(w.tolist())) == len(w)
                    if k == 0:
                        assert start in w.tolist()
                    else:
                        assert seen & set(w.tolist()), "a later window must overlap earlier output"
                    seen |= set(w.tolist())
                assert seen == set(range(total))
    with pytest.raises(AssertionError):
        W.chunk_right(0, 10, 4, 5)


def _replay(bank, ops, batch_dim=True):
    for op in ops:
        if op["op"] == "update":
            lat = torch.tensor(op["latents"])
            bank.update(torch.tensor(op["timesteps"]), lat[None] if batch_dim else lat, replace=op["replace"])
        elif op["op"] == "get":
            lat, m = bank.get(torch.tensor(op["timesteps"]), add_batch_dim=True)
            assert torch.equal(lat.cpu(), torch.tensor(op["latents"])), op["timesteps"]
            assert m.cpu().tolist() == op["mask"]
        else:
            lat, ts = bank.get_ordered()
            assert torch.equal(lat.cpu(), torch.tensor(op["latents"]))
            assert ts.cpu().tolist() == op["timesteps"]


def test_latent_bank_matches_reference(gold):
    _replay(W.LatentBank(empty_dims=(3, 2), capacity=2), gold["bank"])      # capacity 2: exercises growth
    _replay(WO.ListLatentBank((3, 2)), gold["bank"])


def test_latent_bank_duplicates_in_one_update():
    """The reference loops over the timesteps of one update call: without `replace` the first occurrence wins,
    with `replace` the last one does."""
    a, b = torch.full((1, 3, 2), 1.0), torch.full((1, 3, 2), 2.0)
    for replace, want in ((False, 1.0), (True, 2.0)):
        bank, ref = W.LatentBank(empty_dims=(3, 2)), WO.ListLatentBank((3, 2))
        for bk in (bank, ref):
            bk.update(torch.tensor([5.0, 5.0]), torch.cat([a, b]), replace=replace)
            lat, m = bk.get(torch.tensor([5.0]))
            assert float(lat[0, 0, 0]) == want and m.tolist() == [1]


def test_generate_3d_latents_drives_sampler_like_the_reference():
    """The window loop against a recording stand-in sampler: window order, seeds, masks and bank updates follow
    pipeline.py:469-506 (anchor frame conditions the first window, overlaps condition the later ones)."""
    calls = []

    class Sched:
        def get_noise(self, latent_shape, batch_size, n_timesteps, device, generator=None):
            return torch.randn([batch_size, n_timesteps] + list(latent_shape), generator=generator, device=device)

        def denoise(self, model, cfg, init_latent, context, mask, framestep, device, disable_prog, step_callback):
            calls.append((framestep[0].tolist(), mask[0].tolist()))
            assert torch.equal(init_latent[0][mask[0] > 0], bank.get(framestep[0][mask[0] > 0])[0])
            if step_callback is not None:
                step_callback(1, 1)
            return init_latent + 1.0

    class Model:
        device = torch.device("cpu")

    T = 9
    ts = torch.arange(T, dtype=torch.float32)
    bank = W.LatentBank(empty_dims=(2, 2))
    bank.update(ts[3:4], torch.full((1, 2, 2), 7.0))
    seen = []
    W.generate_3d_latents(Model(), Sched(), None, ts, torch.zeros(T, 1, 1), bank, anchor_idx=3, window=4, slide=3,
                          latent_shape=(2, 2), seed=5, step_callback=lambda s, t, i, n: seen.append((i, n)))
    want = [w.tolist() for w in WO.chunk_from(3, T, 4, 3)]
    assert [c[0] for c in calls] == [[float(i) for i in w] for w in want]
    assert calls[0][1] == [1.0 if i == 3 else 0.0 for i in want[0]]
    assert all(sum(c[1]) >= 1 for c in calls), "every window is conditioned on at least one known frame"
    assert bank.n_timesteps == T and seen == [(i, len(want)) for i in range(len(want))]
    lat, m = bank.get(ts[3:4])
    assert float(lat[0, 0, 0]) == 7.0, "the anchor latent is never overwritten (first write wins)"


# ---- Stage-II window loop (pipeline.py:510-600) ----------------------------------------------------------------
def test_timestep_scaling_matches_reference(gold):
    """get_scaling / apply_scaling / get_n_subdivisions / interpolate_timesteps against values computed by the reference's
    own functions (embeddings.py:156-245) for ordered, reversed and anchor-first windows and three subsampling levels."""
    from actionmesh_amd import windows as W
    assert len(gold["scaling"]) >= 20
    for c in gold["scaling"]:
        w = torch.tensor([c["timesteps"]])
        assert W.get_n_subdivisions(w.min().item(), w.max().item(), c["level"]) == c["n"]
        assert W.interpolate_timesteps(w, c["level"], drop_first=True)[0].tolist() == c["output"]
        assert W.interpolate_timesteps(w, c["level"], drop_first=False)[0].tolist() == c["output_all"]
        t_min, t_range = W.get_scaling(w)
        assert t_min.tolist() == c["t_min"] and t_range.tolist() == c["t_range"]
        assert W.apply_scaling(w[:, 0], t_min, t_range).tolist() == c["source_alpha"]
        assert W.apply_scaling(W.interpolate_timesteps(w, c["level"], drop_first=True), t_min, t_range)[0].tolist() == c["target_alphas"]


class _RecordingAutoencoder:
    """Stands for the decoder: displacement = a deterministic function of everything it is handed, so that a wrong
    window, source mesh, alpha or latent shows up in the stored vertices."""
    device = torch.device("cpu")

    def __init__(self):
        self.calls = []

    def __call__(self, latent, framestep, source_alpha, target_alphas, query, step_callback=None):
        self.calls.append((framestep.tolist(), source_alpha.tolist(), target_alphas.tolist()))
        T_out = target_alphas.shape[1]
        if step_callback is not None:
            for i in range(T_out):
                step_callback(i + 1, T_out)
        base = query[0, :, :3] * 0.5 + latent.mean() + framestep.sum() * 1e-3
        return torch.stack([base + 0.01 * float(a) - float(source_alpha[0]) for a in target_alphas[0]])[None]

    @staticmethod
    def apply_displacement(vertex, displacement, scale=1.0):
        return torch.clamp(displacement, -scale, scale)


@pytest.mark.parametrize("n_frames,anchor,level", [(16, 0, 1), (31, 0, 1), (31, 12, 1), (20, 19, 2), (7, 0, 1)])
def test_generate_vertex_animation_matches_the_restated_reference_loop(n_frames, anchor, level):
    from actionmesh_amd import windows as W
    from oracle import windows_oracle as WO
    g = torch.Generator().manual_seed(n_frames * 100 + anchor)
    N, D, V, size, slide = 5, 4, 9, 16, 15
    ts = torch.arange(n_frames, dtype=torch.float32)
    lat = torch.randn((n_frames, N, D), generator=g)
    verts = torch.rand((V, 3), generator=g) - 0.5
    feats = lambda v: torch.cat([v, torch.nn.functional.normalize(v + 0.1, dim=-1)], -1)
    ae = _RecordingAutoencoder()
    bank = W.LatentBank(empty_dims=(N, D)); bank.update(ts, lat)
    vbank = W.LatentBank(empty_dims=(V, 3)); vbank.update(ts[anchor:anchor + 1], verts[None])
    seen = []
    W.generate_vertex_animation(ae, bank, vbank, feats, anchor, size, slide, subsampling_level=level,
                                step_callback=lambda s, t, i, n: seen.append((s, t, i, n)))
    obank = WO.ListLatentBank(empty_dims=(N, D)); obank.update(ts, lat)
    ae2 = _RecordingAutoencoder()

    def decode(latents, wts, sa, ta, src):
        d = ae2(latents, wts, sa, ta, feats(src)[None])
        return ae2.apply_displacement(None, d)[0]
    meshes = WO.generate_mesh_animation(decode, obank, {float(anchor): verts}, anchor, size, slide, subsampling_level=level)
    assert ae.calls == ae2.calls and len(ae.calls) == len(W.chunk_from(anchor, n_frames, size, slide))
    got, got_ts = vbank.get_ordered()
    assert got_ts.tolist() == sorted(meshes)
    for t, v in zip(got_ts.tolist(), got):
        assert torch.equal(v, meshes[t]), t
    assert {(i, n) for _, _, i, n in seen} == {(i, len(ae.calls)) for i in range(len(ae.calls))}
    if level == 1 and anchor == 0:
        assert got_ts.tolist() == ts.tolist()          # every input frame gets a mesh, the anchor keeps its own
        assert torch.equal(got[anchor], verts)
    # (with a later anchor the reference's drop_first removes the window's MINIMUM timestep, not its source frame - a
    #  descending window never decodes frame 0; mirrored as is, see the oracle restatement)
