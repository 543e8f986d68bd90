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
    """Which slice of a (B, T, ...) CFG batch a rank owns.

    `cfg_groups` > 1 additionally splits the CFG batch: the guidance branches are independent
    denoiser evaluations (the reference can even run them one by one, scheduler.py:150-170), so
    with P ranks and 2 branches, ranks [0, P/2) take branch 0 and ranks [P/2, P) branch 1, and
    the K/V all-gather only spans the P/2 ranks that share a branch (half the peers, half the
    bytes; none at all for P = 2)."""
    n_frames: int
    world: int
    rank: int
    batch: int = 1
    cfg_groups: int = 1

    def __post_init__(self):
        if self.world < 1 or not (0 <= self.rank < self.world):
            raise ValueError(f"bad rank {self.rank} / world {self.world}")
        if self.cfg_groups < 1 or self.world % self.cfg_groups or self.batch % self.cfg_groups:
            raise ValueError(f"cfg_groups={self.cfg_groups} must divide world={self.world} and batch={self.batch}")
        if self.n_frames < 1:
            raise ValueError(f"n_frames={self.n_frames}")

    @property
    def group_size(self) -> int:           # ranks sharing one CFG group
        return self.world // self.cfg_groups

    @property
    def frame_world(self) -> int:
        """Frame shards per CFG group.  The reference's chunk_right / chunk_from produce windows shorter than 16
        frames when the video is shorter than the window (5, 7 ... frames); a frame count the group does not divide
        is not sharded at all: every rank of the group then computes all frames of its CFG branch (replicas, no K/V
        exchange) instead of the run failing on inputs the single-GPU path handles."""
        return self.group_size if self.n_frames % self.group_size == 0 else 1

    @property
    def replicated(self) -> bool:
        return self.frame_world != self.group_size

    @property
    def frame_rank(self) -> int:
        return (self.rank % self.group_size) if not self.replicated else 0

    @property
    def cfg_rank(self) -> int:
        return self.rank // self.group_size

    @property
    def frames_local(self) -> int:
        return self.n_frames // self.frame_world

    @property
    def batch_local(self) -> int:
        return self.batch // self.cfg_groups

    @property
    def frame_slice(self) -> slice:
        return slice(self.frame_rank * self.frames_local, (self.frame_rank + 1) * self.frames_local)

    @property
    def batch_slice(self) -> slice:
        return slice(self.cfg_rank * self.batch_local, (self.cfg_rank + 1) * self.batch_local)

    def frame_group_ranks(self, cfg_rank: int) -> List[int]:
        return [cfg_rank * self.group_size + r for r in range(self.group_size)]

    def slice_frames(self, x: torch.Tensor, dim: int = 1) -> torch.Tensor:
        """Local frames of a (B, T, ...) tensor (contiguous copy); the batch is left alone."""
        idx = [slice(None)] * x.dim()
        idx[dim] = self.frame_slice
        return x[tuple(idx)].contiguous()

    def slice_local(self, x: torch.Tensor) -> torch.Tensor:
        """This rank's (batch rows, frames) block of a (B, T, ...) tensor (contiguous copy)."""
        return x[self.batch_slice, self.frame_slice].contiguous()

    def local_times(self, t_bt: List[float]) -> List[float]:
        """(b t)-ordered per-frame values -> this rank's (b_local t_local)-ordered values."""
        T, tl = self.n_frames, self.frames_local
        b0, f0 = self.cfg_rank * self.batch_local, self.frame_rank * tl
        return [t_bt[(b0 + b) * T + f0 + j] for b in range(self.batch_local) for j in range(tl)]


class Engine(Protocol):
    num_layers: int

    def is_inflated(self, layer: int) -> bool: ...
    def begin(self, x_local: torch.Tensor, t_bt_local: List[float]) -> None: ...
    def layer_pre(self, layer: int) -> None: ...
    def layer_post(self, layer: int) -> None: ...
    def end(self) -> torch.Tensor: ...
    def kv_buffers(self) -> Tuple[torch.Tensor, torch.Tensor]: ...


def exchange_kv(kv: Tuple[torch.Tensor, ...], plan: FrameShardPlan, group: Optional[dist.ProcessGroup],
                async_op: bool = False) -> list:
    """All-gather the K / V^T shards in place over this rank's frame group.  `kv` tensors are
    (frame_world, chunk) views; frame shard r has written row r.  With `async_op` the collectives are only
    enqueued and their Work handles returned (wait() before reading the other ranks' rows)."""
    works = []
    for buf in kv:
        assert buf.shape[0] == plan.frame_world and buf.is_contiguous()
        # flat views: accepted by both RCCL and gloo; input aliases its slot of the output
        w = dist.all_gather_into_tensor(buf.view(-1), buf[plan.frame_rank].view(-1), group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def sharded_forward(engine: Engine, plan: FrameShardPlan, group: Optional[dist.ProcessGroup],
                    x_local: torch.Tensor, t_bt_local: List[float]) -> torch.Tensor:
    """One denoiser forward over this rank's (batch rows, frames); returns the local velocity.
    `group` = the frame group of this rank (ranks that share its CFG branch).
    Engines that offer `layer_attn_local` (HipEngine) overlap the exchange with the attention of the full query
    blocks against the local shard: softmax is order-free over keys, so the kernel saves (O, m, l) after the local
    keys and resumes over the remote ones once they have landed."""
    engine.begin(x_local, t_bt_local)
    attn_local = getattr(engine, "layer_attn_local", None)
    for i in range(engine.num_layers):
        engine.layer_pre(i)
        if plan.frame_world > 1 and engine.is_inflated(i):
            if attn_local is not None:
                works = exchange_kv(engine.kv_buffers(), plan, group, async_op=True)
                attn_local(i)
                for w in works:
                    w.wait()
            else:
                exchange_kv(engine.kv_buffers(), plan, group)
        engine.layer_post(i)
    return engine.end()


def gather_frames(v_local: torch.Tensor, plan: FrameShardPlan,
                  group: Optional[dist.ProcessGroup]) -> torch.Tensor:
    """(B_local, T_local, ...) on every rank -> (B, T, ...) on every rank (`group` = all ranks).
    Rank g*group_size + r holds batch block g, frame shard r."""
    if plan.world == 1:
        return v_local
    parts = [torch.empty_like(v_local) for _ in range(plan.world)]
    dist.all_gather(parts, v_local.contiguous(), group=group)
    gs, fw = plan.group_size, plan.frame_world      # replicated groups: the first rank's copy stands for the group
    rows = [torch.cat(parts[g * gs:g * gs + fw], dim=1) for g in range(plan.cfg_groups)]
    return torch.cat(rows, dim=0)
