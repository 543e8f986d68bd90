#!/usr/bin/env python
"""BASELINE.json configs[0] as a record: "assets/examples/davis_camel, 8 frames, 10 denoise steps, --fast, CPU reference path
(plumbing, no GPU)" - the REFERENCE's own modules, on the host cores of the build container, chained the way its pipeline chains them
(pipeline.py:655-692), below the CLI: nothing of this repository's product path runs here.

    python tools/config0_cpu.py [--tokens 256] [--vertices 2000] [--out profiles/r06_config0.json]        (needs /root/reference; ~6 min)

  frames   the first 8 davis_camel frames as the reference's ImagePreprocessor + BitImageProcessor deliver them (tests/golden/frames/)
  context  transformers.Dinov2Model (ViT-L/14, 24 layers; image_encoder.py:53), fp32
  Stage I  ActionMeshDenoiser + SchedulerFlow(10 steps) + ClassifierFreeGuidance [[0,1],[1,1]] x 7.5 at the SHIPPED architecture
           (21 layers, width 2048, 16 heads, actionmesh.yaml:33-43), anchor = frame 0, one window of 8 frames
  Stage II ActionMeshAutoencoder (width 1024, 16 + 1 blocks), 7 target frames, `--vertices` query points on a sphere
What is NOT the reference's configuration, and why: the weights are random-init (no checkpoint offline), Stage 0 (TripoSG / RMBG) is replaced
by a seeded anchor latent + a sphere, and the latent token count is `--tokens` instead of 2048 - a full-size step of the shipped model is
~8 minutes of fp32 on these 8 cores (1.8e14 flop at ~0.4 TFLOP/s), 10 of them would not fit the round.  `diffusers` is the
6-class shim of oracle/diffusers_shim.  A plumbing record: every stage of the CPU reference path runs and hands on finite
results of the right shapes; the seconds say what this container's cores do, nothing about the GPU path.
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--vertices", type=int, default=2000)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    sys.path.insert(0, os.path.join(ROOT, "oracle", "diffusers_shim"))
    sys.path.insert(0, "/root/reference")
    if "trimesh" not in sys.modules:
        tm = types.ModuleType("trimesh"); tm.Trimesh = type("Trimesh", (), {}); sys.modules["trimesh"] = tm
    from transformers import Dinov2Config, Dinov2Model                                  # the reference's image_encoder.py:25
    from actionmesh.model.temporal_autoencoder import ActionMeshAutoencoder             # reference
    from actionmesh.model.temporal_denoiser import ActionMeshDenoiser                   # reference
    from actionmesh.model.utils.embeddings import apply_scaling, get_scaling, interpolate_timesteps   # reference
    from actionmesh.scheduler.guidance import ClassifierFreeGuidance                    # reference
    from actionmesh.scheduler.scheduler import SchedulerFlow                            # reference
    from bench import random_state_dict
    from oracle.make_golden_frames import frames_to_pixels
    from oracle import autoencoder_oracle as AO

    T, N, V = a.frames, a.tokens, a.vertices
    torch.manual_seed(0)
    sec = {}
    t0 = time.time()
    rgb = np.load(os.path.join(ROOT, "tests", "golden", "frames", "davis_camel_16x224.npz"))["rgb_u8"][:T]
    pixels = frames_to_pixels(rgb)
    dino = Dinov2Model(Dinov2Config(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, image_size=518, patch_size=14)).eval()
    hp = dict(in_channels=64, num_layers=21, num_attention_heads=16, width=2048, mlp_ratio=4.0, cross_attention_dim=1024,
              inflated_layers=list(range(21)))
    den = ActionMeshDenoiser(num_tokens_nominal=N, temporal_context_size=16, clear_autocast=False, **hp)
    den.load_state_dict(random_state_dict(hp, seed=0))
    den.eval()
    ae_kw = dict(width=1024, num_layers=16, num_attention_heads=8, latent_channels=64)
    vae = ActionMeshAutoencoder(verbose=False, **ae_kw)
    vae.load_state_dict(AO.synthetic_state_dict(AO.AEConfig(**ae_kw), seed=0))
    vae.eval()
    sec["model_build"] = round(time.time() - t0, 1)

    with torch.no_grad():
        t0 = time.time()
        context = dino(pixels).last_hidden_state                                       # (T, 257, 1024)   image_encoder.py:53-55
        sec["context_encoder"] = round(time.time() - t0, 1)
        g = torch.Generator().manual_seed(44)
        sched = SchedulerFlow(num_inference_steps=a.steps, num_train_timesteps=1000, shift=3.0, is_additive=True, split_cfg_batch=False)
        cfgd = ClassifierFreeGuidance(inference_enabled=True, guidance_at_inference=[[0, 1], [1, 1]], guidance_scales=[7.5])
        anchor = torch.randn((1, 1, N, 64), generator=g)                               # Stage 0's latent: seeded noise here
        noise = sched.get_noise(batch_size=1, latent_shape=[N, 64], n_timesteps=T, generator=g, device="cpu")
        mask = torch.zeros(1, T); mask[0, 0] = 1.0
        init = torch.cat([anchor, noise[:, 1:]], dim=1)
        ts = torch.arange(T, dtype=torch.float32)[None]
        t0 = time.time()
        latents = sched.denoise(den, cfgd, init_latent=init, context=context[None], mask=mask, framestep=ts, device="cpu", disable_prog=True)
        sec["stage_I"] = round(time.time() - t0, 1)
        pts = torch.nn.functional.normalize(torch.randn((1, V, 3), generator=g), dim=-1) * 0.8
        query = torch.cat([pts, torch.nn.functional.normalize(pts, dim=-1)], dim=-1)
        out_ts = interpolate_timesteps(ts, subsampling_level=1, device="cpu", drop_first=True)           # pipeline.py:553-566
        t_min, t_range = get_scaling(ts)
        t0 = time.time()
        disp = vae(latents, ts, apply_scaling(ts[:, 0], t_min, t_range), apply_scaling(out_ts, t_min, t_range), query)
        verts = vae.apply_displacement(pts, disp)
        sec["stage_II"] = round(time.time() - t0, 1)
    assert context.shape == (T, 257, 1024) and latents.shape == (1, T, N, 64) and verts.shape == (1, T - 1, V, 3)
    assert all(bool(torch.isfinite(x).all()) for x in (context, latents, verts)) and torch.equal(latents[0, 0], anchor[0, 0])
    rec = {"baseline_config": "configs[0]: assets/examples/davis_camel, 8 frames, 10 denoise steps, --fast, CPU reference path (plumbing, no GPU)",
           "what": "the reference's own modules chained on the host (Dinov2Model -> ActionMeshDenoiser under SchedulerFlow + CFG -> ActionMeshAutoencoder); "
                   "random-init weights, seeded anchor latent and a sphere instead of Stage 0, REDUCED latent token count; nothing of actionmesh_amd's product path runs",
           "value": round(sec["context_encoder"] + sec["stage_I"] + sec["stage_II"], 1), "unit": "s", "n_gpus": 0, "dtype": "f32", "data": "davis_camel frames, random-init weights",
           "seconds": sec, "host_threads": torch.get_num_threads(),
           "config": {"frames": T, "denoise_steps": a.steps, "latent_tokens_per_frame": N, "shipped_latent_tokens_per_frame": 2048, "vertices": V,
                      "stage_I_architecture": "21 layers, width 2048, 16 heads (shipped)", "stage_II_architecture": "width 1024, 16 + 1 blocks (shipped)"},
           "context_rms": round(float(context.pow(2).mean().sqrt()), 4), "latents_rms": round(float(latents[0, 1:].pow(2).mean().sqrt()), 4),
           "vertex_displacement_rms": round(float((verts - pts[:, None]).pow(2).mean().sqrt()), 4)}
    line = json.dumps(rec)
    print(line)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
