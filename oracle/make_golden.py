"""Generate tests/golden/*.npz from the REFERENCE's own unmodified modules.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (needs
/root/reference, which does not exist on the GPU box):

    python oracle/make_golden.py

The reference's python modules are imported from /root/reference with the
un-vendored `diffusers` dependency supplied by oracle/diffusers_shim.  The
weights are the deterministic synthetic weights of
oracle.denoiser_oracle.synthetic_state_dict (no pretrained weights offline);
only inputs, outputs and a weight checksum are stored, so fixtures stay small.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "diffusers_shim"))
sys.path.insert(0, "/root/reference")

from actionmesh.model.temporal_denoiser import ActionMeshDenoiser  # noqa: E402  (reference)
from actionmesh.model.utils.rotary_embedding import (  # noqa: E402
    apply_rotary_embedding,
    compute_rotary_embeddings,
)
from actionmesh.model.utils.timesteps import chunk_from  # noqa: E402
from actionmesh.scheduler.guidance import ClassifierFreeGuidance  # noqa: E402
from actionmesh.scheduler.scheduler import SchedulerFlow  # noqa: E402

from oracle.denoiser_oracle import (  # noqa: E402
    OracleConfig,
    state_dict_checksum,
    synthetic_state_dict,
)

OUT = os.path.join(ROOT, "tests", "golden")

# name -> (OracleConfig kwargs, T, N, S, steps)
CASES = {
    # all layers inflated (as shipped, actionmesh.yaml:43), odd L, mask on frame 0
    "tiny_inflated": (dict(in_channels=64, num_layers=5, num_attention_heads=2, width=256,
                           mlp_ratio=4.0, cross_attention_dim=64,
                           inflated_layers=(0, 1, 2, 3, 4)), 4, 48, 9, 4),
    # one non-inflated layer (layer 2) to pin the per-frame attention branch
    "tiny_mixed": (dict(in_channels=64, num_layers=5, num_attention_heads=2, width=256,
                        mlp_ratio=4.0, cross_attention_dim=64,
                        inflated_layers=(0, 1, 3, 4)), 3, 70, 17, 3),
}


def build_reference_model(cfg_kwargs, sd):
    m = ActionMeshDenoiser(
        num_tokens_nominal=48, temporal_context_size=4, clear_autocast=False, **cfg_kwargs
    )
    ref_keys = set(m.state_dict().keys())
    assert ref_keys == set(sd.keys()), (ref_keys ^ set(sd.keys()))
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    m.load_state_dict(sd)
    return m.eval()


def make_case(name, cfg_kwargs, T, N, S, steps):
    cfg = OracleConfig(**cfg_kwargs)
    sd = synthetic_state_dict(cfg, seed=0)
    model = build_reference_model(cfg_kwargs, sd)

    g = torch.Generator().manual_seed(1234)
    init_latent = torch.randn(1, T, N, cfg.in_channels, generator=g)
    context = torch.randn(1, T, S, cfg.cross_attention_dim, generator=g)
    mask = torch.zeros(1, T)
    mask[0, 0] = 1.0
    framestep = torch.tensor([[1.0, 0.0, 2.0, 3.0, 5.0, 4.0][:T]])

    cfgd = ClassifierFreeGuidance(
        inference_enabled=True, guidance_at_inference=[[0, 1], [1, 1]], guidance_scales=[7.5]
    )
    sched = SchedulerFlow(num_inference_steps=steps, num_train_timesteps=1000, shift=3.0,
                          is_additive=True, split_cfg_batch=False)

    out = {
        "init_latent": init_latent.numpy(), "context": context.numpy(),
        "mask": mask.numpy(), "framestep": framestep.numpy(),
        "weights_checksum": np.float64(state_dict_checksum(sd)),
        "steps": np.int64(steps),
    }

    with torch.no_grad():
        # (1) one CFG-batched forward at t = 700 (fp32 = the reference CPU path)
        x_in, c_in, m_in, f_in = cfgd.cfg_at_inference(init_latent, context, mask, framestep)
        t_in = torch.tensor([700.0]).expand(2)
        v, _ = model.forward(hidden_states=x_in, context=c_in, framestep=f_in,
                             diffusion_time=t_in, mask=m_in, freqs_rot=None)
        out["fwd_t"] = np.float32(700.0)
        out["fwd_velocity_fp32"] = v.numpy().copy()

        # (2) the same forward under CPU autocast(bf16): the closest available
        # stand-in for the reference's cuda autocast dtype flow (informational)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            vb, _ = model.forward(hidden_states=x_in, context=c_in, framestep=f_in,
                                  diffusion_time=t_in, mask=m_in, freqs_rot=None)
        out["fwd_velocity_cpu_autocast_bf16"] = vb.float().numpy().copy()

        # (3) the full sampler loop, per-step latents (fp32)
        lat = init_latent.clone()
        per_step = []
        for latents, t in sched._flow_sample(
            diffusion_model=model, cf_guidance=cfgd, init_latent=lat, context=context,
            device="cpu", disable_prog=True, mask=mask, framestep=framestep,
        ):
            per_step.append(latents.clone().numpy())
        out["loop_latents_fp32"] = np.stack(per_step)

        # (4) split_cfg_batch=True must give the same result (scheduler.py:150-170)
        sched2 = SchedulerFlow(num_inference_steps=steps, shift=3.0, is_additive=True,
                               split_cfg_batch=True)
        final2 = sched2.denoise(model, cfgd, init_latent=init_latent.clone(), context=context,
                                device="cpu", disable_prog=True, mask=mask, framestep=framestep)
        out["loop_final_split_cfg_fp32"] = final2.numpy().copy()

    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"[golden] {name}: v rms {float(v.pow(2).mean().sqrt()):.4f} "
          f"final latent rms {float(torch.from_numpy(per_step[-1]).pow(2).mean().sqrt()):.4f}")


def make_kats():
    """Known-answer vectors from the reference's importable helper modules
    (SURVEY.md App. D), stored so the GPU box can check them without the reference."""
    kat = {}
    for n in (10, 15, 30, 50):
        t, d = SchedulerFlow(num_inference_steps=n, shift=3.0).get_schedule()
        kat[f"sched_t_{n}"] = t.numpy()
        kat[f"sched_d_{n}"] = d.numpy()
    cos, sin = compute_rotary_embeddings(128, torch.arange(16.0))
    kat["rope_cos_128_16"] = cos.numpy()
    kat["rope_sin_128_16"] = sin.numpy()
    x = torch.randn(1, 2, 16, 128, generator=torch.Generator().manual_seed(0))
    kat["rope_apply_in"] = x.numpy()
    kat["rope_apply_out"] = apply_rotary_embedding(x, cos, sin).numpy()
    noise = SchedulerFlow(num_inference_steps=1).get_noise(
        [2048, 64], 1, 16, "cpu", torch.Generator().manual_seed(44))
    kat["noise_seed44_head"] = noise[0, :2, 0, :3].numpy()
    kat["noise_seed44_mean_std"] = np.array([float(noise.mean()), float(noise.std())])
    small = SchedulerFlow(num_inference_steps=1).get_noise(
        [8, 4], 1, 3, "cpu", torch.Generator().manual_seed(7))
    kat["noise_seed7_small"] = small.numpy()
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    kat["cfg_aggregate_1_2"] = cfgd.aggregate_cfg(torch.tensor([[1.0], [2.0]])).numpy()
    for (s, tot) in ((0, 16), (3, 16), (0, 31), (0, 32), (0, 24), (10, 31)):
        ch = chunk_from(s, tot, 16, 15)
        kat[f"chunk_from_{s}_{tot}"] = np.array([[int(i) for i in c] for c in ch], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "kats.npz"), **kat)
    print("[golden] kats:", len(kat), "arrays")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    for name, (kw, T, N, S, steps) in CASES.items():
        make_case(name, kw, T, N, S, steps)
    make_kats()
