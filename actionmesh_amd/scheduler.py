"""HipSchedulerFlow / ClassifierFreeGuidance: drop-ins for
actionmesh.scheduler.scheduler.SchedulerFlow (reference scheduler.py:23-295) and
actionmesh.scheduler.guidance.ClassifierFreeGuidance (guidance.py:13-118).

Same dataclass fields and method signatures, so a Hydra overlay can swap the
`_target_`s (actionmesh_amd/configs/actionmesh_mi355x.yaml).  `denoise` drives the
HIP denoiser: the CFG batch, context K/V cache and RoPE table are built once per
window; each step is one C-ABI forward plus one fused CFG+Euler kernel.
"""
from __future__ import annotations

import contextlib
import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np
import torch

from . import ops
from .denoiser import HipDenoiser


@dataclass(eq=False)
class ClassifierFreeGuidance:
    """Conditioning order is [image-conditioning | latent0-conditioning] (guidance.py:15-17)."""

    inference_enabled: bool = True
    guidance_at_inference: list = field(default_factory=lambda: [[0, 0], [0, 1], [1, 1]])
    guidance_scales: list = field(default_factory=lambda: [1.0, 1.0])

    def __post_init__(self):
        assert len(self.guidance_at_inference) == len(self.guidance_scales) + 1

    def get_unobserved_mask(self, mask: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        return None if mask is None else mask == 0

    def branches(self) -> List[List[int]]:
        return [list(g) for g in self.guidance_at_inference] if self.inference_enabled else [[1, 1]]

    def cfg_at_inference(self, latent, context, mask, framestep):
        """guidance.py:38-93: one batch row per guidance branch; a 0 flag zeroes that conditioning."""
        if not self.inference_enabled:
            return latent, context, mask, framestep
        n = len(self.guidance_at_inference)
        latent = torch.cat([latent] * n)
        framestep = torch.cat([framestep] * n) if framestep is not None else None
        ctxs, masks = [], []
        for g in self.guidance_at_inference:
            g = list(g)
            if g not in ([1, 1], [0, 1], [1, 0], [0, 0]):
                raise Exception(f"Unknown guidance: {g}")
            ctxs.append(context if g[0] == 1 else torch.zeros_like(context))
            if mask is not None:
                masks.append(mask if g[1] == 1 else torch.zeros_like(mask))
        return latent, torch.cat(ctxs, dim=0), (torch.cat(masks, dim=0) if mask is not None else None), framestep

    def aggregate_cfg(self, aggregated: torch.Tensor) -> torch.Tensor:
        """guidance.py:95-118 (torch glue; the fused device version is ops.flow_step)."""
        if not self.inference_enabled:
            return aggregated
        outs = aggregated.chunk(len(self.guidance_at_inference), dim=0)
        assert len(outs) == len(self.guidance_at_inference)
        output = outs[0]
        for i in range(len(self.guidance_at_inference) - 1):
            output += self.guidance_scales[i] * (outs[i + 1] - outs[i])
        return output


@dataclass(eq=False)
class HipSchedulerFlow:
    num_inference_steps: int
    num_train_timesteps: int = 1000
    shift: float = 3.0
    is_additive: bool = False
    split_cfg_batch: bool = False   # scheduler.py:150-170: one forward per CFG branch (half the activation workspace)
    # Not a reference field.  The reference's GPU path multiplies `distances[i]` (a 0-dim fp32 DEVICE tensor) with the bf16
    # velocity, and torch's type promotion rounds the distance to bf16 first (scheduler.py:238-241 under cuda autocast; its
    # fp32 CPU path does not).  False (default) keeps dt in fp32 = the reference's CPU arithmetic; True reproduces the
    # cuda-autocast rounding bit for bit.
    cuda_autocast_dt: bool = False
    # Not a reference field.  Two EXACT shortcuts of the CFG batch (am_set_branch_hints: bit-identical latents): the unconditional
    # branch's cross-attention is its to_out bias (zero context, bias-free to_k / to_v), and layer 0's self-attention branch is
    # computed once for all guidance branches (they carry the same latents and times until the first cross-attention).
    exact_shortcuts: bool = True
    # Not a reference field.  Only matters with a denoiser that has a process group: `denoise` keeps the latents sharded by frames
    # across the steps and gathers them once at the end; False = gather the velocity on every rank in every step
    # (what the `_flow_sample` generator does when driven directly, where every yielded tensor has to be complete).
    keep_latents_sharded: bool = True

    def get_schedule(self):
        """scheduler.py:43-56."""
        timesteps = self._compute_timesteps(self.num_inference_steps + 1, self.num_train_timesteps, self.shift)
        distances = (timesteps[:-1] - timesteps[1:]) / self.num_train_timesteps
        return timesteps, distances

    @staticmethod
    def _compute_timesteps(num_inference_steps: int, num_train_timesteps: int = 1000,
                           shift: float = 1.0) -> torch.Tensor:
        """scheduler.py:59-98: shifted-sigma schedule, fp64 numpy -> fp32."""
        n = num_train_timesteps
        full = (np.linspace(1, n, n) / n)[::-1]
        full_shifted = shift * full / (1 + (shift - 1) * full)
        t = np.linspace(full_shifted[0] * n, full_shifted[-1] * n, num_inference_steps)
        s = t / n
        s = shift * s / (1 + (shift - 1) * s)
        return torch.from_numpy((s * n).astype(np.float32))

    def get_noise(self, latent_shape, batch_size: int, n_timesteps: int, device,
                  generator: Optional[torch.Generator] = None, corr_noise: float = 0.0) -> torch.Tensor:
        """scheduler.py:100-137: two draws in this order (same, then independent)."""
        assert 0 <= corr_noise <= 1.0
        same = torch.randn([batch_size, 1] + list(latent_shape), generator=generator,
                           device=device).repeat(1, n_timesteps, 1, 1)
        indep = torch.randn([batch_size, n_timesteps] + list(latent_shape), generator=generator, device=device)
        return math.sqrt(corr_noise) * same + math.sqrt(1 - corr_noise) * indep

    def _flow_sample(self, diffusion_model, cf_guidance, init_latent, context, device="cuda:0",
                     disable_prog: bool = True, mask=None, framestep=None):
        """scheduler.py:172-250 on the HIP path (the reference's signature).  Yields (latents, t) like the reference (the same
        tensor object every step, complete on every rank; clone to record)."""
        return self._flow_sample_impl(diffusion_model, cf_guidance, init_latent, context, device, disable_prog, mask, framestep)

    def _flow_sample_impl(self, diffusion_model, cf_guidance, init_latent, context, device="cuda:0",
                          disable_prog: bool = True, mask=None, framestep=None, local_latents: bool = False):
        """`_flow_sample` plus one switch.
        `local_latents` (a denoiser with a process group; what `denoise` uses): the latents stay SHARDED across the steps - every
        rank advances only the frames it computes the velocity of (`forward_host_time(gather=False)`: no velocity collective at
        all when the CFG batch sits on one group, one small all-gather among the same-frame ranks otherwise), and the frames are
        gathered ONCE, behind the last step.  The tensor yielded before the last step then holds current values
        only in this rank's frames."""
        if not isinstance(diffusion_model, HipDenoiser):
            raise TypeError("HipSchedulerFlow drives a HipDenoiser; got "
                            f"{type(diffusion_model).__name__} (there is no torch fallback path)")
        if init_latent.dim() != 4 or init_latent.shape[0] != 1:
            raise ValueError("HipSchedulerFlow expects init_latent of shape (1, T, N, D) as the pipeline passes it")
        dev = diffusion_model.device
        latents = init_latent.to(dev, torch.float32)
        if not latents.is_contiguous():
            latents = latents.contiguous()
        _, T, N, D = latents.shape
        branches = cf_guidance.branches()
        nb = len(branches)
        scales = [float(s) for s in cf_guidance.guidance_scales] if cf_guidance.inference_enabled else []
        ctx = context.to(dev, torch.float32)
        ctx_b = torch.cat([ctx if g[0] == 1 else torch.zeros_like(ctx) for g in branches], dim=0)
        if framestep is None:
            framestep = torch.arange(T, dtype=torch.float32)[None]
        fs_b = torch.cat([framestep.detach().float().cpu()] * nb, dim=0)
        mask_host = None if mask is None else mask.detach().float().cpu().reshape(-1).tolist()
        unobserved = None if mask_host is None else [m == 0 for m in mask_host]
        if unobserved is not None and not any(unobserved):
            raise AssertionError("No unobserved frames found")           # scheduler.py:245
        keep = [[(1.0 - m) if g[1] == 1 else 1.0 for m in mask_host] if mask_host is not None else [1.0] * T
                for g in branches]

        timesteps, distances = self.get_schedule()
        local_latents = local_latents and diffusion_model.process_group is not None
        fsl = diffusion_model.frame_slice(T, 1 if (self.split_cfg_batch and nb > 1) else nb) if local_latents else slice(0, T)
        unobs_step = unobserved if unobserved is None else unobserved[fsl]
        split = self.split_cfg_batch and nb > 1
        zero = [self.exact_shortcuts and g[0] != 1 for g in branches]
        shared = self.exact_shortcuts and nb > 1 and all(k == keep[0] for k in keep)
        if not split:
            diffusion_model.bind_window(ctx_b, fs_b, N, ctx_zero=zero, shared_prefix=shared)
        it = range(self.num_inference_steps)
        if not disable_prog:
            from tqdm import tqdm
            it = tqdm(it, desc="Temporal 3D Denoising (Stage I)", leave=True)
        for i in it:
            t = float(timesteps[i])
            dt = float(distances[i])
            if self.cuda_autocast_dt:          # the 16-bit type the model computes in (bfloat16, or float16 under --dtype float16)
                h16 = torch.float16 if diffusion_model.compute_kind() == "f16" else torch.bfloat16
                dt = float(torch.tensor(dt, dtype=torch.float32).to(h16))
            # every launch of the step goes to the denoiser's device and that device's current stream, whichever device is
            # current in the calling thread (the forward and the CFG+Euler kernel must share a stream to be ordered)
            with (torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext()):   # (a CPU device: the gloo tests' stand-in engine)
                if split:
                    # scheduler.py:159-168: branch by branch, each with its own slice of the context (the window is re-bound
                    # per branch: the K/V cache holds one context at a time, as the reference recomputes K/V every call)
                    vs = []
                    for b in range(nb):
                        diffusion_model.bind_window(ctx_b[b:b + 1], fs_b[b:b + 1], N, ctx_zero=zero[b:b + 1])
                        vb = diffusion_model.forward_host_time(latents, [t * k for k in keep[b]], gather=not local_latents)
                        # a sharded denoiser hands out its (re-used) gather buffer: keep a copy before the next branch overwrites it
                        vs.append(vb.clone() if diffusion_model.process_group is not None else vb)
                    v = torch.cat(vs, dim=0)
                else:
                    t_bt = [t * k for row in keep for k in row]              # temporal_denoiser.py:209-212
                    x_in = latents.expand(nb, T, N, D).contiguous()
                    v = diffusion_model.forward_host_time(x_in, t_bt, gather=not local_latents)
                # (T, N, D) fp32 is contiguous, so a frame range of it is too: the CFG + Euler kernel runs on this rank's frames only
                ops.flow_step(v, latents[0][fsl], scales, dt, self.is_additive, unobs_step)
                if local_latents and i == self.num_inference_steps - 1:
                    diffusion_model.gather_latent_frames(latents[0], 1 if split else nb)
            # copy-engine exchange: the previous step's fault word without a device sync - except BEHIND THE LAST STEP, where the verdict
            # is taken blocking before the final latents are handed out (a caller that drives this generator directly gets it too;
            # the deferred word used to leave the last step of such a caller unchecked)
            diffusion_model.check_exchange(block=(i == self.num_inference_steps - 1))
            yield latents, timesteps[i]

    @torch.no_grad()
    def denoise(self, diffusion_model, cf_guidance, init_latent, context, device="cuda:0",
                disable_prog: bool = True, mask=None, framestep=None,
                step_callback: Optional[Callable[[int, int], None]] = None):
        """scheduler.py:252-295."""
        latents = None
        total = self.num_inference_steps
        for step_idx, (sample, _t) in enumerate(self._flow_sample_impl(
                diffusion_model, cf_guidance, init_latent, context, device, disable_prog, mask, framestep,
                local_latents=self.keep_latents_sharded)):
            latents = sample
            if step_callback is not None:
                step_callback(step_idx + 1, total)
        if isinstance(diffusion_model, HipDenoiser):
            diffusion_model.check_exchange(block=True)       # the blocking verdict before the latents are handed on
        return latents
