"""Stage II: ActionMeshAutoencoder on the Stage-I kernels.

CPU: the oracle restatement (oracle/autoencoder_oracle.py) against the fixture generated from the reference's own
unmodified module (oracle/make_golden_autoencoder.py -> tests/golden/ae_tiny.npz).
GPU: HipAutoencoder through the C-ABI against the same fixture and the oracle.

Stated tolerance (displacement = 2 sigmoid(logit) - 1 in [-1, 1]; bf16 storage / fp32 accumulation through
num_layers + 1 transformer blocks, the reference's own cuda path runs the self-attention stack under bf16 autocast):
max abs error <= 2e-2, rel-L2 <= 2e-2 against the fp32 reference."""
import os

import numpy as np
import pytest
import torch

from oracle import autoencoder_oracle as AO


def _case(golden_dir):
    g = np.load(os.path.join(golden_dir, "ae_tiny.npz"))
    width, layers, heads, latent = (int(v) for v in g["config"])
    cfg = AO.AEConfig(width=width, num_layers=layers, num_attention_heads=heads, latent_channels=latent)
    sd = AO.synthetic_state_dict(cfg, seed=0)
    assert AO.state_dict_checksum(sd) == pytest.approx(float(g["weights_checksum"]), rel=1e-12)
    t = {k: torch.from_numpy(g[k]) for k in ("latent", "framestep", "source_alpha", "target_alphas", "query", "displacement_fp32")}
    return cfg, sd, t


def test_oracle_matches_reference_fixture(golden_dir):
    cfg, sd, t = _case(golden_dir)
    d = AO.autoencoder_forward(sd, cfg, t["latent"], t["framestep"], t["source_alpha"], t["target_alphas"], t["query"])
    assert d.shape == t["displacement_fp32"].shape
    assert float((d - t["displacement_fp32"]).abs().max()) < 1e-5


def test_oracle_embeddings_known_answers():
    """TimestepEmbedder / FrequencyPositionalEmbedding layouts (embeddings.py:14-130)."""
    e = AO.timestep_embed(4, torch.tensor([0.0, 1.0]), torch.tensor([2.0, 3.0]))
    assert e.shape == (2, 8)
    assert torch.allclose(e[1, :4], torch.tensor([np.cos(1.0), np.cos(0.01), np.sin(1.0), np.sin(0.01)], dtype=torch.float32), atol=1e-6)
    cfg = AO.AEConfig(embed_frequency=2)
    p = AO.point_embed(cfg, torch.tensor([[0.5, -1.0, 2.0]]))
    want = [0.5, -1.0, 2.0] + [np.sin(v * f) for v in (0.5, -1.0, 2.0) for f in (1.0, 2.0)] + \
           [np.cos(v * f) for v in (0.5, -1.0, 2.0) for f in (1.0, 2.0)]
    assert torch.allclose(p[0], torch.tensor(want, dtype=torch.float32), atol=1e-6)


def test_from_pretrained_reads_the_reference_layout(tmp_path, golden_dir):
    """config.json + model.safetensors as ActionMeshAutoencoder.from_pretrained writes them (no GPU needed to load)."""
    import json
    from safetensors.torch import save_file
    from actionmesh_amd.autoencoder import HipAutoencoder
    cfg, sd, _ = _case(golden_dir)
    (tmp_path / "config.json").write_text(json.dumps(dict(width=cfg.width, num_layers=cfg.num_layers,
                                                          num_attention_heads=cfg.num_attention_heads, verbose=False)))
    save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / "model.safetensors"))
    m = HipAutoencoder.from_pretrained(str(tmp_path))
    assert (m.width, m.num_layers, m.heads, m.query_dim, m.query_pad) == (cfg.width, cfg.num_layers, 2, 54, 64)
    with pytest.raises(RuntimeError):      # no CPU path
        m.forward(torch.zeros(1, 2, 4, 64), torch.zeros(1, 2), torch.zeros(1), torch.zeros(1, 1), torch.zeros(1, 3, 6))


@pytest.mark.gpu
def test_hip_autoencoder_matches_reference_fixture_and_oracle(golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from actionmesh_amd.autoencoder import HipAutoencoder
    cfg, sd, t = _case(golden_dir)
    dev = torch.device("cuda:0")
    m = HipAutoencoder(width=cfg.width, num_layers=cfg.num_layers, num_attention_heads=cfg.num_attention_heads,
                       latent_channels=cfg.latent_channels)
    m.load_state_dict(sd)
    m.to(dev).eval()
    seen = []
    d = m(t["latent"].to(dev), t["framestep"], t["source_alpha"], t["target_alphas"], t["query"].to(dev),
          step_callback=lambda i, n: seen.append((i, n)))
    torch.cuda.synchronize()
    d = d.cpu()
    ref = t["displacement_fp32"]
    assert d.shape == ref.shape and seen == [(i + 1, ref.shape[1]) for i in range(ref.shape[1])]
    err = float((d - ref).abs().max())
    rl = float((d - ref).norm() / ref.norm())
    print(f"Stage II: max abs err {err:.3e}, rel-L2 {rl:.3e} vs the reference's fp32 displacement")
    assert err < 2e-2 and rl < 2e-2
    assert float(d.abs().max()) <= 1.0


@pytest.mark.gpu
def test_point_embed_and_displacement_kernels():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from actionmesh_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    q = torch.cat([torch.rand((1000, 3), generator=g) * 2 - 1, torch.randn((1000, 3), generator=g)], -1)
    for include_pi in (False, True):
        cfg = AO.AEConfig(embed_include_pi=include_pi)
        e = ops.point_embed(q.to(dev), 3, 3, 8, include_pi).float().cpu()
        want = torch.cat([AO.point_embed(cfg, q[:, :3]), q[:, 3:]], -1)
        assert e.shape == (1000, 64) and float(e[:, 54:].abs().max()) == 0.0
        # bf16 rounding of values in [-1, 1] plus the fp32 sin/cos of arguments up to 128 pi
        assert float((e[:, :54] - want).abs().max()) < 8e-3
    lg = (torch.randn((777, 8), generator=g) * 3).to(torch.bfloat16)
    out = torch.empty((777, 3), device=dev)
    ops.displacement(lg.to(dev), 3, out)
    want = 2 * torch.sigmoid(-lg.float()[:, :3]) - 1
    assert float((out.cpu() - want).abs().max()) < 1e-5


@pytest.mark.gpu
def test_hip_autoencoder_properties_at_model_width():
    """Size-independent properties at the shipped width (1024 / 8 heads; 2 + 1 blocks, 8 frames x 1023 tokens so that
    the self-attention runs on the long-key-stream kernel): (a) outputs in [-1, 1] and finite; (b) permuting the query
    vertices permutes the displacements (each vertex attends independently); (c) source == target with the same
    weights is deterministic across calls; (d) extra padded query channels do not leak (first 3 output dims only)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from actionmesh_amd.autoencoder import HipAutoencoder
    dev = torch.device("cuda:0")
    cfg = AO.AEConfig(width=1024, num_layers=2, num_attention_heads=8)
    sd = AO.synthetic_state_dict(cfg, seed=1)
    m = HipAutoencoder(width=1024, num_layers=2, num_attention_heads=8)
    m.load_state_dict(sd)
    m.to(dev)
    g = torch.Generator().manual_seed(5)
    B, T, N, V = 1, 8, 1023, 1500
    latent = torch.randn((B, T, N, 64), generator=g)
    fs = torch.arange(T, dtype=torch.float32)[None]
    query = torch.cat([torch.rand((B, V, 3), generator=g) * 2 - 1, torch.randn((B, V, 3), generator=g)], -1)
    src, tgt = torch.tensor([0.5]), torch.tensor([[0.0, 0.5]])
    d = m(latent.to(dev), fs, src, tgt, query.to(dev))
    assert d.shape == (B, 2, V, 3) and bool(torch.isfinite(d).all()) and float(d.abs().max()) <= 1.0
    perm = torch.randperm(V, generator=g)
    dp = m(latent.to(dev), fs, src, tgt, query[:, perm].to(dev))
    assert float((dp - d[:, :, perm.to(dev)]).abs().max()) < 2e-2      # bf16 noise only (different tile membership)
    d2 = m(latent.to(dev), fs, src, tgt, query.to(dev))
    assert torch.equal(d, d2), "same inputs, same bits"
