"""Autoregressive window orchestration around the Stage-I sampler: the callers of the hot path.

Mirrors, with the same names and semantics,
  * `chunk_right` / `chunk_left` / `chunk_from`  - actionmesh/model/utils/timesteps.py:10-117 (window index lists);
  * `LatentBank`                                 - actionmesh/model/utils/storage.py:89-186 (latents keyed by timestep);
  * `ActionMeshPipeline._denoise_latents` / `generate_3d_latents` - pipeline.py:247-314, 469-506 (one window of
    flow-matching conditioned on the bank; the loop over windows);
  * `get_scaling` / `apply_scaling` / `get_n_subdivisions` / `interpolate_timesteps` - model/utils/embeddings.py:156-245,
    and the Stage-II window loop `ActionMeshPipeline.generate_mesh_animation` / `_decode_displacement`
    (pipeline.py:316-385, 510-600) as `generate_vertex_animation`: the same loop on vertex tensors (the trimesh objects,
    their vertex normals and the MeshBank stay with the caller - mesh glue of the reference's CPU path).
MI355X-first differences: the bank is ONE device tensor (capacity x N x D) plus a host-side timestep -> slot table
(video timesteps are host floats, pipeline io/video_input.py:34), so `get` is a single gather and `update` a single
scatter instead of a Python loop of per-frame `.to(device)` / `torch.stack`; the masks the sampler needs are produced
on the host without a device round trip.  Inputs longer than the 16-frame context stay sequentially dependent
(window i conditions on window i-1's output): this is orchestration, not a parallel axis.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch


# ---------------------------------------------------------------------------------------------------------
# window index lists (timesteps.py:10-117)
# ---------------------------------------------------------------------------------------------------------
def chunk_right(start: int, end: int, size: int, slide: int) -> List[torch.Tensor]:
    """Overlapping index windows moving left to right over [start, end): the first is full-sized (or as long as the
    range), each next one ends `slide` further (clamped to `end`) and reaches back `size` (clamped to `start`)."""
    if not 0 < slide <= size:
        raise AssertionError(f"Need slide <= size, got {slide} > {size}")
    ends: List[int] = []
    e = start
    while e < end:
        e = min(start + size, end) if not ends else min(e + slide, end)
        ends.append(e)
    return [torch.arange(max(start, e - size), e) for e in ends]


def chunk_left(start: int, end: int, size: int, slide: int) -> List[torch.Tensor]:
    """`chunk_right` mirrored: windows from the right end towards `start`, each in descending index order."""
    return [c.flip(0) for c in reversed(chunk_right(start, end, size, slide))]


def chunk_from(start: int, total: int, size: int, slide: int) -> List[torch.Tensor]:
    """Windows covering [0, total) that grow outwards from the anchor frame `start` (timesteps.py:77-117): the side
    with more frames first; the other side restarts from a window that overlaps the anchor's neighbourhood."""
    context = size - slide
    if total == size:        # a single window, anchor first
        idx = torch.arange(total)
        return [torch.cat([idx[start:start + 1], idx[idx != start]])]
    if start == 0:
        return chunk_right(0, total, size, slide)
    if start == total - 1:
        return chunk_left(0, total, size, slide)
    if start > total - start:                                    # more frames on the left: go left first
        left = chunk_left(0, start + 1, size, slide)
        right_start = min(max(0, start - context + 1), total - size)
        return left + chunk_right(right_start, total, size, slide)
    right = chunk_right(start, total, size, slide)
    left_end = max(min(start + context, total), size)
    return right + chunk_left(0, left_end, size, slide)


# ---------------------------------------------------------------------------------------------------------
# latent bank (storage.py:89-186)
# ---------------------------------------------------------------------------------------------------------
@dataclass
class LatentBank:
    """Latents indexed by (float) video timestep, resident in one device tensor.

    Same contract as the reference's LatentBank: `update` keeps the first latent written for a timestep unless
    `replace=True`; timesteps match within `eps`; `get` returns zeros and mask 0 for timesteps never written."""
    empty_dims: Tuple[int, ...] = (768, 64)
    device: str = "cpu"
    capacity: int = 64
    eps: float = 1e-5
    timesteps: List[float] = field(default_factory=list)          # insertion order, like the reference's list
    _store: Optional[torch.Tensor] = field(default=None, repr=False)

    def __post_init__(self):
        self.empty_dims = tuple(self.empty_dims)
        self._store = torch.zeros((self.capacity,) + self.empty_dims, dtype=torch.float32, device=self.device)

    @property
    def n_timesteps(self) -> int:
        return len(self.timesteps)

    def get_timestep_index(self, timestep: float) -> Optional[int]:
        for i, ts in enumerate(self.timesteps):
            if abs(ts - timestep) < self.eps:
                return i
        return None

    def _grow(self, need: int) -> None:
        if need <= self._store.shape[0]:
            return
        cap = max(need, 2 * self._store.shape[0])
        new = torch.zeros((cap,) + self.empty_dims, dtype=torch.float32, device=self._store.device)
        new[: self._store.shape[0]] = self._store
        self._store = new

    def update(self, timesteps: torch.Tensor, latents: torch.Tensor, replace: bool = False) -> None:
        """`latents`: any leading shape whose element count matches len(timesteps) x empty_dims (storage.py:104-132)."""
        ts = [float(t) for t in timesteps.detach().flatten().cpu().tolist()]
        lat = latents.reshape((len(ts),) + self.empty_dims)
        src, dst = [], []
        for i, t in enumerate(ts):                               # host-side key matching, no device traffic
            j = self.get_timestep_index(t)
            if j is None:
                self.timesteps.append(t)
                j = len(self.timesteps) - 1
            elif not replace:
                continue
            # a later duplicate inside the same call overrides an earlier one only under `replace`, like the loop
            if j in dst:
                if not replace:
                    continue
                src[dst.index(j)] = i
                continue
            src.append(i); dst.append(j)
        if not dst:
            return
        self._grow(len(self.timesteps))
        dev = self._store.device
        rows = lat.index_select(0, torch.tensor(src, device=lat.device)).to(dev, torch.float32)
        self._store.index_copy_(0, torch.tensor(dst, device=dev), rows)

    def get(self, timesteps: torch.Tensor, device=None, add_batch_dim: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (latents (N, *dims), mask (N,) int32): one gather; zeros / 0 where the timestep was never written."""
        assert timesteps.ndim == 1
        device = self._store.device if device is None else torch.device(device)
        idx, mask = [], []
        for t in timesteps.detach().cpu().tolist():
            j = self.get_timestep_index(float(t))
            idx.append(0 if j is None else j)
            mask.append(0 if j is None else 1)
        m = torch.tensor(mask, dtype=torch.int32, device=device)
        if self.n_timesteps == 0:
            lat = torch.zeros((len(idx),) + self.empty_dims, dtype=torch.float32, device=device)
        else:
            lat = self._store.index_select(0, torch.tensor(idx, device=self._store.device)).to(device)
            lat = lat * m.to(lat.dtype).view((-1,) + (1,) * len(self.empty_dims))
        return (lat[None], m[None]) if add_batch_dim else (lat, m)

    def get_ordered_timesteps(self) -> torch.Tensor:
        """All stored timesteps, ascending (storage.py:78-83)."""
        return torch.tensor(sorted(self.timesteps))

    def get_ordered(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """All stored latents sorted by timestep, with the timesteps (storage.py:171-186)."""
        order = sorted(range(len(self.timesteps)), key=lambda i: self.timesteps[i])
        lat = self._store.index_select(0, torch.tensor(order, device=self._store.device))
        return lat, torch.tensor([self.timesteps[i] for i in order]).to(lat)


# ---------------------------------------------------------------------------------------------------------
# one window / all windows of Stage I (pipeline.py:247-314, 469-506)
# ---------------------------------------------------------------------------------------------------------
def denoise_window(denoiser, scheduler, cf_guidance, timesteps: torch.Tensor, context: torch.Tensor, bank: LatentBank,
                   latent_shape: Sequence[int], seed: int = 44, device=None, noise_device=None,
                   step_callback: Optional[Callable[[int, int], None]] = None) -> torch.Tensor:
    """`ActionMeshPipeline._denoise_latents` for one window: frames already in the bank condition (mask 1, kept
    clean), the others start from noise.  `timesteps` (T,) host float32, `context` (T, S, Dc).
    `noise_device`: where the initial noise is drawn (default: the sampling device, like the reference; parity tests
    draw on the CPU so that the oracle sees the same noise)."""
    device = denoiser.device if device is None else torch.device(device)
    noise_device = device if noise_device is None else torch.device(noise_device)
    generator = torch.Generator(device=noise_device).manual_seed(seed)
    cond, mask = bank.get(timesteps, device=device, add_batch_dim=True)
    noise = scheduler.get_noise(batch_size=1, latent_shape=list(latent_shape), n_timesteps=int(timesteps.shape[0]),
                                generator=generator, device=noise_device).to(device)
    keep = mask[..., None, None].to(noise.dtype)
    init_latent = cond * keep + noise * (1.0 - keep)
    return scheduler.denoise(denoiser, cf_guidance, init_latent=init_latent, context=context[None],
                             mask=mask.to(init_latent.dtype), framestep=timesteps[None], device=device,
                             disable_prog=True, step_callback=step_callback)


def generate_3d_latents(denoiser, scheduler, cf_guidance, timesteps: torch.Tensor, context: torch.Tensor,
                        bank: LatentBank, anchor_idx: int, window: int, slide: int, latent_shape: Sequence[int],
                        seed: int = 44, device=None, noise_device=None,
                        step_callback: Optional[Callable[[int, int, int, int], None]] = None) -> LatentBank:
    """`ActionMeshPipeline.generate_3d_latents`: the bank already holds the anchor frame's latent; windows grow
    outwards from the anchor (`chunk_from`), window i is seeded with `seed + i`, and every window's output is
    added to the bank (first write wins) before the next window starts."""
    windows = chunk_from(start=anchor_idx, total=int(timesteps.shape[0]), size=window, slide=slide)
    for i, idx in enumerate(windows):
        cb = None
        if step_callback is not None:
            def cb(step, total, _i=i, _n=len(windows)):
                step_callback(step, total, _i, _n)
        ts = timesteps[idx]
        lat = denoise_window(denoiser, scheduler, cf_guidance, ts, context[idx], bank, latent_shape, seed + i,
                             device, noise_device, cb)
        bank.update(timesteps=ts, latents=lat)
    return bank


# ---------------------------------------------------------------------------------------------------------
# Stage II window loop (embeddings.py:156-245; pipeline.py:316-385, 510-600)
# ---------------------------------------------------------------------------------------------------------
def get_scaling(timesteps: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(B, T) -> per-row (min, max - min)."""
    t_min = timesteps.min(dim=1).values
    return t_min, timesteps.max(dim=1).values - t_min


def apply_scaling(timesteps: torch.Tensor, t_min: torch.Tensor, t_range: torch.Tensor) -> torch.Tensor:
    """(timesteps - t_min) / t_range for (B,) or (B, T) timesteps."""
    if timesteps.dim() == 1:
        return (timesteps - t_min) / t_range
    return (timesteps - t_min.unsqueeze(1)) / t_range.unsqueeze(1)


def get_n_subdivisions(start, end, level: int = 1) -> int:
    """Points on [start, end] after `level - 1` midpoint subdivisions of the unit-spaced grid."""
    n = int(end - start + 1)
    for _ in range(1, level):
        n += n - 1
    return n


def interpolate_timesteps(timesteps: torch.Tensor, subsampling_level: int, device=None, drop_first: bool = False) -> torch.Tensor:
    """(1, n) linspace between the global min and max of `timesteps` (any shape), optionally without its first point."""
    t_min, t_max = timesteps.min().item(), timesteps.max().item()
    out = torch.linspace(t_min, t_max, get_n_subdivisions(t_min, t_max, level=subsampling_level), device=device).reshape(1, -1)
    return out[:, 1:] if drop_first else out


def generate_vertex_animation(autoencoder, latent_bank: LatentBank, vertex_bank: LatentBank,
                              vertex_features: Callable[[torch.Tensor], torch.Tensor], anchor_idx: int, window: int, slide: int,
                              subsampling_level: int = 1, device=None,
                              step_callback: Optional[Callable[[int, int, int, int], None]] = None) -> LatentBank:
    """`ActionMeshPipeline.generate_mesh_animation` on vertex tensors.

    `latent_bank`: the Stage-I latents; `vertex_bank`: a LatentBank with empty_dims (V, 3) that already holds the anchor
    mesh's vertices at the anchor timestep and receives every output timestep's deformed vertices (first write wins,
    like the MeshBank); `vertex_features(vertices (V, 3)) -> (V, 3 + extra)`: positions + normalised vertex normals of
    the window's source mesh (pipeline.py:351-354 via trimesh).  Windows grow outwards from the anchor (`chunk_from`);
    the source mesh of a window is the one stored for its first timestep."""
    device = autoencoder.device if device is None else torch.device(device)
    windows = chunk_from(start=anchor_idx, total=latent_bank.n_timesteps, size=window, slide=slide)
    all_timesteps = latent_bank.get_ordered_timesteps()
    for i, idx in enumerate(windows):
        window_timesteps = all_timesteps[idx][None]
        latents, _ = latent_bank.get(window_timesteps[0], device=device, add_batch_dim=True)
        source, have = vertex_bank.get(window_timesteps[0, :1], device=device)
        if int(have[0]) != 1:
            raise RuntimeError(f"generate_vertex_animation: no mesh stored for timestep {float(window_timesteps[0, 0])}")
        output_timesteps = interpolate_timesteps(window_timesteps, subsampling_level, drop_first=True)
        t_min, t_range = get_scaling(window_timesteps)
        source_alpha = apply_scaling(window_timesteps[:, 0], t_min, t_range)
        target_alphas = apply_scaling(output_timesteps, t_min, t_range)
        cb = None
        if step_callback is not None:
            def cb(step, total, _i=i, _n=len(windows)):
                step_callback(step, total, _i, _n)
        feats = vertex_features(source[0])[None].to(device)
        disp = autoencoder(latent=latents, framestep=window_timesteps, source_alpha=source_alpha, target_alphas=target_alphas,
                           query=feats, step_callback=cb)
        deformed = autoencoder.apply_displacement(vertex=feats[..., :3], displacement=disp)
        vertex_bank.update(timesteps=output_timesteps[0], latents=deformed[0])
    return vertex_bank
