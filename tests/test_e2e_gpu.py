"""The GPU stages of the video -> 4D pipeline chained end to end (context encoder -> Stage I AR windows -> Stage II windows)
on tiny random-init models: shapes, finiteness, every frame decoded, anchor mesh kept.  (Numerical parity of each stage
against its oracle lives in test_image_encoder.py, test_denoiser_gpu.py and test_autoencoder.py.)"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.gpu
def test_pipeline_stages_chain_on_tiny_models():
    import e2e_synthetic as E
    dev = torch.device("cuda:0")
    out = E.run(frames=6, steps=2, vertices=300, tiny=True, dev=dev)      # 2 AR windows of 4 frames (slide 3)
    assert out["unit"] == "s" and out["value"] > 0
    assert set(out["seconds"]) == {"context_encoder", "stage_I", "stage_II", "model_build_and_upload", "output_files_host_side",
                                   "chamfer_metrics"}
    assert out["output_files_mb"] > 0
    assert "2 AR window(s) of 4" in out["config"]["workload"]


@pytest.mark.gpu
def test_pipeline_chain_matches_the_oracle_chain():
    """Numerical parity of the whole chain on tiny models: pixels -> DINOv2 context -> two dependent Stage-I windows (CFG,
    3 flow steps each, CPU-drawn noise) -> Stage-II decoding of both windows -> vertices, against the same chain through
    the CPU oracles (dinov2_oracle -> windows_oracle.generate_3d_latents over denoiser_oracle -> generate_mesh_animation
    over autoencoder_oracle), everything in fp32 there.  Tolerances: 2e-2 context, 4e-2 latents after dependent windows,
    2e-2 vertices decoded from a shared source mesh (see the note at the assertions for the second window)."""
    from actionmesh_amd import ClassifierFreeGuidance, HipAutoencoder, HipDenoiser, HipImageEncoder, HipSchedulerFlow
    from actionmesh_amd import windows as W
    from oracle import autoencoder_oracle as AO
    from oracle import denoiser_oracle as O
    from oracle import dinov2_oracle as DO
    from oracle import windows_oracle as WO
    dev = torch.device("cuda:0")
    T, N, D, V, steps, window, slide = 6, 48, 64, 200, 3, 4, 3
    dcfg = DO.DinoConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=1, image_size=56)
    dsd = DO.synthetic_state_dict(dcfg, seed=3)
    hp = dict(in_channels=64, num_layers=5, num_attention_heads=2, width=256, mlp_ratio=4.0, cross_attention_dim=64,
              inflated_layers=(0, 1, 2, 3, 4))
    ocfg = O.OracleConfig(**hp)
    osd = O.synthetic_state_dict(ocfg, seed=0)
    acfg = AO.AEConfig(width=256, num_layers=2, num_attention_heads=2, latent_channels=64)
    asd = AO.synthetic_state_dict(acfg, seed=1)
    g = torch.Generator().manual_seed(9)
    pixels = torch.randn((T, 3, 56, 56), generator=g)
    anchor = torch.randn((1, N, D), generator=g)
    verts = torch.rand((V, 3), generator=g) * 1.6 - 0.8
    feats = lambda v: torch.cat([v, torch.nn.functional.normalize(v + 0.05, dim=-1)], dim=-1)
    ts = torch.arange(T, dtype=torch.float32)

    # ---- HIP chain
    enc = HipImageEncoder(config=dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=1, image_size=56), state_dict=dsd).to(dev)
    den = HipDenoiser(num_tokens_nominal=N, temporal_context_size=window, **hp)
    den.load_state_dict(osd)
    den.to(dev).eval()
    vae = HipAutoencoder(temporal_context_size=window, width=256, num_layers=2, num_attention_heads=2, latent_channels=64)
    vae.load_state_dict(asd)
    vae.to(dev)
    context = enc.encode_pixels(pixels.to(dev))
    bank = W.LatentBank(empty_dims=(N, D), device=str(dev))
    bank.update(ts[:1], anchor)
    W.generate_3d_latents(den, HipSchedulerFlow(num_inference_steps=steps, shift=3.0, is_additive=True),
                          ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5]), ts, context, bank, 0, window, slide, (N, D),
                          seed=44, device=dev, noise_device="cpu")
    vbank = W.LatentBank(empty_dims=(V, 3), device=str(dev))
    vbank.update(ts[:1], verts[None])
    W.generate_vertex_animation(vae, bank, vbank, feats, 0, window, slide, device=dev)
    got, got_ts = vbank.get_ordered()

    # ---- oracle chain (fp32, CPU)
    ctx_ref = DO.dinov2_forward(dsd, dcfg, pixels)
    ref_bank = WO.ListLatentBank((N, D))
    ref_bank.update(ts[:1], anchor)
    WO.generate_3d_latents(osd, ocfg, ts, ctx_ref, ref_bank, 0, window, slide, (N, D), steps, seed=44)

    def decode(latents, wts, sa, ta, src):
        d = AO.autoencoder_forward(asd, acfg, latents, wts, sa, ta, feats(src)[None])
        return torch.clamp(d, -1.0, 1.0)[0]
    meshes = WO.generate_mesh_animation(decode, ref_bank, {0.0: verts}, 0, window, slide)

    assert got_ts.tolist() == sorted(meshes) == ts.tolist()
    assert _rel(context.cpu(), ctx_ref) < 2e-2
    lat, _ = bank.get_ordered()
    lat_ref, _ = ref_bank.get_ordered()
    lat_err = [_rel(lat[i].cpu(), lat_ref[i]) for i in range(1, T)]
    v_err = [_rel(got[i].cpu(), meshes[float(i)]) for i in range(1, T)]
    print("chain: latents rel-L2", [f"{e:.2e}" for e in lat_err], "vertices rel-L2", [f"{e:.2e}" for e in v_err])
    assert torch.equal(got[0].cpu(), verts)
    assert max(lat_err) < 4e-2
    # window 1 (frames 1..3) decodes from the anchor mesh both sides share; window 2 (frames 4, 5) decodes from the mesh
    # window 1 produced for frame 2, and the decoder embeds those positions with frequencies up to 2^7: a 6e-3 relative
    # difference of the source vertices is a phase difference of ~0.4 rad in the top band.  The coupled chain is therefore
    # only loosely comparable there (the reference's own bf16 and fp32 runs diverge the same way) ...
    assert max(v_err[:3]) < 2e-2 and max(v_err[3:]) < 2e-1
    # ... and window 2 is checked with the inputs held equal: the HIP decoder on the oracle chain's latents and source mesh
    wts = ts[2:6][None]
    lat2, _ = ref_bank.get(wts[0], add_batch_dim=True)
    src2 = meshes[2.0]
    out_ts = W.interpolate_timesteps(wts, 1, drop_first=True)
    t_min, t_range = W.get_scaling(wts)
    sa, ta = W.apply_scaling(wts[:, 0], t_min, t_range), W.apply_scaling(out_ts, t_min, t_range)
    d_hip = vae(latent=lat2.to(dev), framestep=wts, source_alpha=sa, target_alphas=ta, query=feats(src2)[None].to(dev))
    d_ref = AO.autoencoder_forward(asd, acfg, lat2, wts, sa, ta, feats(src2)[None])
    assert _rel(d_hip.cpu(), d_ref) < 2e-2
    for j, t in enumerate(out_ts[0].tolist()):
        if t >= 4.0:                              # frames 2 and 3 were written by window 1 (first write wins)
            assert torch.equal(torch.clamp(d_ref[0, j], -1.0, 1.0), meshes[t])


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())
