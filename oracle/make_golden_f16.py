"""The reference's OWN float16 curve for the BASELINE-architecture fixtures (tests/golden/arch_*_f16.npz).

TEST INFRASTRUCTURE ONLY.  Build container only (needs /root/reference):   python oracle/make_golden_f16.py [arch_headline ...]

The reference CLI's `--dtype float16` (inference/video_to_animated_mesh.py:153,222) runs Stage I under torch.autocast(dtype=float16)
(pipeline.py:671).  The fp32 truth of a case is already in tests/golden/<case>.npz (oracle/make_golden_baseline.py, the reference's
unmodified modules); this script runs the SAME modules on the SAME regenerated inputs under torch.autocast("cpu", dtype=torch.float16)
- the closest stand-in here for the reference's cuda float16 flow - and stores how far that run is from the fp32 one: one forward and
every sampler step.  The float16 build of the library (HipDenoiser(dtype="float16")) is held to that curve in
tests/test_baseline_arch_gpu.py::test_float16_mode_against_the_reference.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.make_golden_baseline import OUT, TOKEN_STRIDE, baseline_case_inputs, rel, tensor_checksum  # noqa: E402


def make(name: str):
    sys.path.insert(0, os.path.join(ROOT, "oracle", "diffusers_shim"))
    sys.path.insert(0, "/root/reference")
    from actionmesh.model.temporal_denoiser import ActionMeshDenoiser      # reference
    from actionmesh.scheduler.guidance import ClassifierFreeGuidance       # reference
    from actionmesh.scheduler.scheduler import SchedulerFlow               # reference
    t00 = time.time()
    g = np.load(os.path.join(OUT, f"{name}.npz"))
    kw, cfg, sd, inp, steps = baseline_case_inputs(name)
    assert np.allclose([tensor_checksum(inp[k]) for k in ("init_latent", "context", "mask", "framestep")], g["inputs_checksum"], rtol=1e-12)
    T = inp["init_latent"].shape[1]
    model = ActionMeshDenoiser(num_tokens_nominal=inp["init_latent"].shape[2], temporal_context_size=T, clear_autocast=False, **kw)
    model.load_state_dict(sd)
    model.eval()
    cfgd = ClassifierFreeGuidance(inference_enabled=True, guidance_at_inference=[[0, 1], [1, 1]], guidance_scales=[7.5])
    ref_fwd = torch.from_numpy(g["fwd_velocity_fp32"])
    ref_steps = torch.from_numpy(g["loop_latents_sub_fp32"])
    out = {"steps": np.int64(steps), "token_stride": np.int64(TOKEN_STRIDE), "inputs_checksum": g["inputs_checksum"],
           "weights_checksum": g["weights_checksum"]}
    with torch.no_grad():
        x_in, c_in, m_in, f_in = cfgd.cfg_at_inference(inp["init_latent"], inp["context"], inp["mask"], inp["framestep"])
        t_in = torch.tensor([float(g["fwd_t"])]).expand(2)
        t0 = time.time()
        with torch.autocast("cpu", dtype=torch.float16):
            vh, _ = model.forward(hidden_states=x_in, context=c_in, framestep=f_in, diffusion_time=t_in, mask=m_in, freqs_rot=None)
        assert vh.dtype == torch.float16 and torch.isfinite(vh.float()).all()
        out["fwd_ref_autocast_f16_vs_fp32"] = np.float64(rel(vh.float(), ref_fwd))
        print(f"[{name}] forward under autocast(float16): {time.time() - t0:.1f} s, rel-L2 vs the reference's fp32 {rel(vh.float(), ref_fwd):.3e} "
              f"(its autocast(bf16): {float(g['fwd_ref_autocast_vs_fp32']):.3e})", flush=True)
        curve = []
        sched = SchedulerFlow(num_inference_steps=steps, num_train_timesteps=1000, shift=3.0, is_additive=True)
        with torch.autocast("cpu", dtype=torch.float16):
            for i, (latents, _t) in enumerate(sched._flow_sample(
                    diffusion_model=model, cf_guidance=cfgd, init_latent=inp["init_latent"].clone(), context=inp["context"],
                    device="cpu", disable_prog=True, mask=inp["mask"], framestep=inp["framestep"])):
                curve.append(rel(latents[:, :, ::TOKEN_STRIDE].float(), ref_steps[i]))
                print(f"[{name}] autocast(float16) step {i + 1}/{steps}: rel-L2 vs fp32 {curve[-1]:.3e}  ({time.time() - t00:.0f} s)", flush=True)
        out["ref_autocast_f16_curve"] = np.array(curve)
    np.savez_compressed(os.path.join(OUT, f"{name}_f16.npz"), **out)
    print(f"[golden] {name}_f16: done in {time.time() - t00:.0f} s", flush=True)


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["arch_headline"]):
        make(name)
