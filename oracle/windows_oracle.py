"""CPU restatement of the reference's autoregressive window orchestration (TEST INFRASTRUCTURE ONLY - imported by
tests/ only; never on the product path).

Follows, literally and with Python lists like the reference:
  * actionmesh/model/utils/timesteps.py:10-117  chunk_right / chunk_left / chunk_from
  * actionmesh/model/utils/storage.py:48-186    LatentBank (first write wins, eps matching, zeros + mask 0 if missing)
  * actionmesh/pipeline.py:247-314, 469-506     _denoise_latents / generate_3d_latents
with the per-window sampler restated by oracle.denoiser_oracle.flow_sample.  Pinned against the reference's own
functions by tests/golden/windows.json (oracle/make_golden_windows.py) for the chunking and the bank; the window
loop itself is glue over those and the (fixture-pinned) sampler.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import denoiser_oracle as O


def chunk_right(start: int, end: int, size: int, slide: int) -> List[torch.Tensor]:       # timesteps.py:10-45
    assert 0 < slide <= size
    chunks: List[torch.Tensor] = []
    chunk_end = start
    while chunk_end < end:
        chunk_end = min(start + size, end) if not chunks else min(chunk_end + slide, end)
        chunks.append(torch.arange(max(start, chunk_end - size), chunk_end))
    return chunks


def chunk_left(start: int, end: int, size: int, slide: int) -> List[torch.Tensor]:        # timesteps.py:48-74
    return [c.flip(0) for c in reversed(chunk_right(start, end, size, slide))]


def chunk_from(start: int, total: int, size: int, slide: int) -> List[torch.Tensor]:      # timesteps.py:77-117
    context = size - slide
    if total == size:
        indices = torch.arange(total)
        return [torch.cat([indices[start:start + 1], indices[indices != start]])]
    if start == 0:
        return chunk_right(0, total, size, slide)
    if start == total - 1:
        return chunk_left(0, total, size, slide)
    if start > total - start:
        left = chunk_left(0, start + 1, size, slide)
        right_start = min(max(0, start - context + 1), total - size)
        return left + chunk_right(right_start, total, size, slide)
    right = chunk_right(start, total, size, slide)
    left_end = max(min(start + context, total), size)
    return right + chunk_left(0, left_end, size, slide)


class ListLatentBank:                                                                     # storage.py:19-186
    def __init__(self, empty_dims: Tuple[int, ...]):
        self.items: List[torch.Tensor] = []
        self.timesteps: List[float] = []
        self.empty_dims = tuple(empty_dims)

    def index(self, timestep: float, eps: float = 1e-5) -> Optional[int]:
        for i, ts in enumerate(self.timesteps):
            if abs(ts - timestep) < eps:
                return i
        return None

    def update(self, timesteps: torch.Tensor, latents: torch.Tensor, replace: bool = False) -> None:
        timesteps = timesteps.flatten()
        latents = latents.reshape(timesteps.shape[0], *self.empty_dims)
        for i in range(timesteps.shape[0]):
            t = timesteps[i].item()
            j = self.index(t)
            if j is None:
                self.timesteps.append(t)
                self.items.append(latents[i])
            elif replace:
                self.items[j] = latents[i]

    def get(self, timesteps: torch.Tensor, add_batch_dim: bool = False):
        lat, mask = [], []
        for t in timesteps:
            j = self.index(float(t))
            lat.append(torch.zeros(self.empty_dims) if j is None else self.items[j])
            mask.append(0 if j is None else 1)
        lat, mask = torch.stack(lat), torch.tensor(mask, dtype=torch.int32)
        return (lat[None], mask[None]) if add_batch_dim else (lat, mask)

    def get_ordered(self):
        order = sorted(range(len(self.timesteps)), key=lambda i: self.timesteps[i])
        return torch.stack([self.items[i] for i in order]), torch.tensor([self.timesteps[i] for i in order])


def generate_3d_latents(sd, cfg: O.OracleConfig, timesteps: torch.Tensor, context: torch.Tensor, bank: ListLatentBank,
                        anchor_idx: int, window: int, slide: int, latent_shape, num_inference_steps: int,
                        seed: int = 44, precision: str = "fp32") -> ListLatentBank:
    """pipeline.py:469-506 over pipeline.py:247-314 (noise drawn with a CPU generator seeded seed + i)."""
    for i, idx in enumerate(chunk_from(anchor_idx, int(timesteps.shape[0]), window, slide)):
        ts = timesteps[idx]
        generator = torch.Generator().manual_seed(seed + i)
        cond, mask = bank.get(ts, add_batch_dim=True)
        noise = O.get_noise(list(latent_shape), 1, int(ts.shape[0]), generator)
        keep = mask[..., None, None].to(noise.dtype)
        init_latent = cond * keep + noise * (1.0 - keep)
        lat = O.flow_sample(sd, cfg, init_latent, context[idx][None], mask.to(init_latent.dtype), ts[None],
                            num_inference_steps, precision=precision)[-1]
        bank.update(ts, lat)
    return bank


def generate_mesh_animation(decode, latent_bank: ListLatentBank, meshes: dict, anchor_idx: int, window: int, slide: int,
                            subsampling_level: int = 1) -> dict:
    """pipeline.py:510-600 with the meshes reduced to their vertex tensors: `meshes` maps a timestep to (V, 3) vertices
    and already holds the anchor's; `decode(latents (1, T, N, D), window_timesteps (1, T), source_alpha (1,),
    target_alphas (1, T_out), source_vertices (V, 3)) -> (T_out, V, 3)` stands for `_decode_displacement`
    (pipeline.py:316-385).  Timestep arithmetic as in embeddings.py:156-245, written out with Python floats."""
    ordered = sorted(latent_bank.timesteps)
    for idx in chunk_from(anchor_idx, len(ordered), window, slide):
        wts = [ordered[int(i)] for i in idx]
        latents, _ = latent_bank.get(torch.tensor(wts), add_batch_dim=True)
        src = next(v for t, v in meshes.items() if abs(t - wts[0]) < 1e-5)          # MeshBank.get(window_timesteps[:, 0])
        t_min, t_max = min(wts), max(wts)
        n = int(t_max - t_min + 1)
        for _ in range(1, subsampling_level):
            n += n - 1
        out_ts = torch.linspace(t_min, t_max, n)[1:]                               # drop_first=True
        rng = t_max - t_min
        source_alpha = torch.tensor([(wts[0] - t_min) / rng])
        target_alphas = ((out_ts - t_min) / rng)[None]
        verts = decode(latents, torch.tensor([wts]), source_alpha, target_alphas, src)
        for t, v in zip(out_ts.tolist(), verts):
            if not any(abs(t - k) < 1e-5 for k in meshes):                          # first write wins
                meshes[t] = v
    return meshes
