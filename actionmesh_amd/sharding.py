"""Frame sharding of the denoiser forward across the GPUs of one node.

Everything in the Stage-I denoiser is frame-local except the inflated
self-attention: LayerNorm, the QKV/out/FFN GEMMs,
qk-norm, RoPE, the per-frame cross-attention, the skip linears and proj_in/out
touch one frame at a time.  Rank r of P therefore owns frames
[r*T/P, (r+1)*T/P) of BOTH CFG samples, weights are replicated, and the only
data-path collective is, per inflated layer, one all-gather of the post-RoPE K
and V^T shards (attention output rows stay frame-local, so no reduction).

The gathered buffers are laid out [rank][B][H][...] - softmax is
permutation-invariant over keys, so the attention kernel just walks the P
chunks and no re-ordering is needed after the all-gather.

The driver below is engine-agnostic: the product engine is `HipEngine`
(C-ABI); tests drive the same code with a CPU stand-in over gloo.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Protocol, Tuple

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class FrameShardPlan:
    n_frames: int
    world: int
    rank: int

    def __post_init__(self):
        if self.world < 1 or not (0 <= self.rank < self.world):
            raise ValueError(f"bad rank {self.rank} / world {self.world}")
        if self.n_frames % self.world != 0:
            raise ValueError(
                f"{self.n_frames} frames do not divide over {self.world} ranks "
                "(the reference window is 16 frames: use 1, 2, 4, 8 or 16 GPUs)")

    @property
    def frames_local(self) -> int:
        return self.n_frames // self.world

    @property
    def frame_slice(self) -> slice:
        return slice(self.rank * self.frames_local, (self.rank + 1) * self.frames_local)

    def slice_frames(self, x: torch.Tensor, dim: int = 1) -> torch.Tensor:
        """Local frames of a (B, T, ...) tensor (contiguous copy)."""
        idx = [slice(None)] * x.dim()
        idx[dim] = self.frame_slice
        return x[tuple(idx)].contiguous()


class Engine(Protocol):
    num_layers: int

    def is_inflated(self, layer: int) -> bool: ...
    def begin(self, x_local: torch.Tensor, t_bt_local: List[float]) -> None: ...
    def layer_pre(self, layer: int) -> None: ...
    def layer_post(self, layer: int) -> None: ...
    def end(self) -> torch.Tensor: ...
    def kv_buffers(self) -> Tuple[torch.Tensor, torch.Tensor]: ...


def exchange_kv(kv: Tuple[torch.Tensor, torch.Tensor], plan: FrameShardPlan,
                group: Optional[dist.ProcessGroup]) -> None:
    """All-gather the K and V^T shards in place.  `kv` tensors are (world, chunk)
    views; rank r has written row r."""
    for buf in kv:
        assert buf.shape[0] == plan.world and buf.is_contiguous()
        # flat views: accepted by both RCCL and gloo; input aliases its slot of the output
        dist.all_gather_into_tensor(buf.view(-1), buf[plan.rank].view(-1), group=group)


def sharded_forward(engine: Engine, plan: FrameShardPlan, group: Optional[dist.ProcessGroup],
                    x_local: torch.Tensor, t_bt_local: List[float]) -> torch.Tensor:
    """One denoiser forward over this rank's frames; returns the local velocity."""
    engine.begin(x_local, t_bt_local)
    for i in range(engine.num_layers):
        engine.layer_pre(i)
        if plan.world > 1 and engine.is_inflated(i):
            exchange_kv(engine.kv_buffers(), plan, group)
        engine.layer_post(i)
    return engine.end()


def gather_frames(v_local: torch.Tensor, plan: FrameShardPlan,
                  group: Optional[dist.ProcessGroup]) -> torch.Tensor:
    """(B, T_local, ...) on every rank -> (B, T, ...) on every rank."""
    if plan.world == 1:
        return v_local
    parts = [torch.empty_like(v_local) for _ in range(plan.world)]
    dist.all_gather(parts, v_local.contiguous(), group=group)
    return torch.cat(parts, dim=1)
