"""HipDenoiser: drop-in for actionmesh.model.temporal_denoiser.ActionMeshDenoiser
(reference temporal_denoiser.py:23-249) whose forward runs in libactionmesh_amd.so.

Same constructor fields, same `forward(hidden_states, context, framestep,
diffusion_time, mask=None, freqs_rot=None) -> (velocity, freqs_rot)`, `.device`,
`.eval()`, `.to()`, `load_state_dict` with the reference's state-dict keys and
`from_pretrained(dir)`.  Python here only marshals pointers; there is no torch
arithmetic on the hot path and no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib as L
from .sharding import FrameShardPlan, gather_cfg, gather_frames, gather_latent_frames, sharded_forward

HEAD_DIM = 128


def rope_tables_host(framestep: torch.Tensor, head_dim: int = HEAD_DIM) -> Tuple[torch.Tensor, torch.Tensor]:
    """Host restatement of precompute_freqs_rot (temporal_denoiser.py:114-149): positions =
    framestep - min_t framestep (embeddings.py:135-153), angle_i = pos * 10000^(-2i/hd)
    (rotary_embedding.py:10-69).  Returns cos, sin of shape (B*T, hd/2) fp32: one value per
    interleaved pair; every token of a frame (incl. the time token) shares its frame's angle."""
    fs = framestep.detach().float().cpu()
    pos = (fs - fs.min(dim=1).values[:, None]).reshape(-1)
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    phases = torch.outer(pos, inv_freq)
    return phases.cos().contiguous(), phases.sin().contiguous()


def masked_time(diffusion_time: Sequence[float], mask: Optional[torch.Tensor], B: int, T: int) -> List[float]:
    """temporal_denoiser.py:209-212: t.repeat(T) (b-fastest) * (1 - merged mask) ((b t) order)."""
    t_rep = [float(diffusion_time[i % B]) for i in range(B * T)]
    if mask is None:
        return t_rep
    m = mask.detach().float().cpu().reshape(B * T).tolist()
    return [t * (1.0 - mm) for t, mm in zip(t_rep, m)]


def _tensor_identity(t: torch.Tensor) -> Tuple:
    """Cheap identity of a tensor's current contents: storage address, geometry and torch's in-place version counter."""
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype, str(t.device), t._version)


class WindowCache:
    """Opaque `freqs_rot` object handed back to the sampler (scheduler.py:224-232).

    In the reference `freqs_rot` caches the RoPE table only; here a bound window also holds the cross-attention K/V
    cache of ONE context tensor.  The reference sampler with `split_cfg_batch: true` (actionmesh_lowram.yaml) calls
    forward once per CFG branch with `context[b:b+1]` and hands every branch the `freqs_rot` branch 0 returned, so the
    cache is tied to the context (storage address, geometry, in-place version) and the framesteps it was built from:
    a call with any other context re-binds instead of silently reusing the first branch's K/V."""

    def __init__(self, generation: int, context: Optional[torch.Tensor] = None,
                 framestep: Optional[torch.Tensor] = None, n_tokens: int = 0):
        self.generation = generation
        # identity of the bound context: storage address + geometry + torch's in-place version counter.  Only the STORAGE is kept
        # alive (so the address cannot be handed to another tensor while this window is bound), not the tensor object.
        self._context_storage = None if context is None else context.untyped_storage()
        self.context_id = None if context is None else _tensor_identity(context)
        self._framestep_id = None if framestep is None else _tensor_identity(framestep)
        self._framestep_storage = None if framestep is None else framestep.untyped_storage()
        self.framestep = None if framestep is None else framestep.detach().float().cpu().reshape(-1).tolist()
        self.n_tokens = n_tokens

    def matches(self, context: torch.Tensor, framestep: torch.Tensor, n_tokens: int) -> bool:
        if self.context_id != _tensor_identity(context) or self.n_tokens != n_tokens:
            return False
        if self._framestep_id == _tensor_identity(framestep):       # the same tensor, unmodified: no device-to-host sync per step
            return True
        return self.framestep == framestep.detach().float().cpu().reshape(-1).tolist()


class HipEngine:
    """Owns one am_handle (weights + workspace on one GPU) and exposes the phase protocol
    of sharding.Engine."""

    def __init__(self, hp: Dict, state_dict: Dict[str, torch.Tensor], device: torch.device,
                 max_batch: int, frames_local: int, tokens: int, ctx_tokens: int,
                 world: int = 1, rank: int = 0, attn_defer_log2: int = 8, attn_dtype: str = "bf16",
                 kv_factory=None, use_graph: bool = False, dtype="bfloat16"):
        # the 16-bit storage / MFMA element type: bfloat16 (default) or float16 - the same sources built twice (_lib.lib(kind))
        self.kind = L.kind_of(dtype)
        self.h16 = torch.float16 if self.kind == "f16" else torch.bfloat16
        self.lib = L.lib(self.kind)
        self._check = lambda status, what="": L.check(status, what, self.lib)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("HipEngine needs a ROCm device (torch device type 'cuda'); there is no CPU path")
        self.hp = dict(hp)
        self.num_layers = hp["num_layers"]
        self.inflated = set(hp["inflated_layers"])
        self.bounds = (max_batch, frames_local, tokens, ctx_tokens)
        self.world, self.rank = world, rank
        cfg = L.AmConfig()
        cfg.in_channels = hp["in_channels"]; cfg.num_layers = hp["num_layers"]
        cfg.num_heads = hp["num_attention_heads"]; cfg.width = hp["width"]
        cfg.ff_inner = int(hp["width"] * hp["mlp_ratio"]); cfg.cross_dim = hp["cross_attention_dim"]
        mask = 0
        for i in self.inflated:
            mask |= 1 << i
        cfg.inflated_mask_lo = mask & 0xFFFFFFFF; cfg.inflated_mask_hi = (mask >> 32) & 0xFFFFFFFF
        cfg.max_batch, cfg.max_frames_local, cfg.max_tokens, cfg.max_ctx_tokens = self.bounds
        cfg.world_size, cfg.rank = world, rank
        cfg.attn_defer_log2 = attn_defer_log2
        if attn_dtype not in ("bf16", "fp8", "fp8_fast"):
            raise ValueError(f"attn_dtype must be 'bf16', 'fp8' or 'fp8_fast', got {attn_dtype!r}")
        cfg.attn_fp8 = {"bf16": 0, "fp8": 1, "fp8_fast": 2}[attn_dtype]
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):
            self._check(self.lib.am_create(C.byref(cfg), C.byref(self.handle)), "am_create")
            for name, t in state_dict.items():
                t = t.detach().to("cpu", torch.float32).contiguous()
                self._check(self.lib.am_load_weight(self.handle, name.encode(), t.data_ptr(), t.numel()),
                        f"am_load_weight({name})")
            missing = self.lib.am_weights_missing(self.handle)
            if missing:
                raise RuntimeError(f"HipEngine: {missing} reference state-dict keys were not provided")
            self._kv = None
            self.exchange = None
            # HIP-graph replay of the single-rank forward (am_denoise_forward_graph): operands at fixed addresses, owned here
            self.use_graph = bool(use_graph) and world == 1
            self._gx = self._gv = self._gstream = None
            if world > 1:
                n = C.c_size_t()
                self._check(self.lib.am_kv_chunk_elems(self.handle, C.byref(n)), "am_kv_chunk_elems")
                # one buffer [rank][K chunk | V^T chunk]: a single in-place all-gather per layer moves both operands.  An fp8 handle
                # exchanges the QUANTISED shards (am_bind_kv8_buffers): one byte per element, half the traffic.
                esz = 1 if attn_dtype.startswith("fp8") else 2
                if kv_factory is not None:      # copy-engine back-end: the buffer is an IPC-shared hipMalloc, not a torch tensor
                    self.exchange = kv_factory(2 * n.value * esz)
                    base = self.exchange.kv_ptr()
                else:
                    kv = torch.zeros((world, 2 * n.value), dtype=torch.uint8 if esz == 1 else self.h16, device=self.device)
                    base = kv.data_ptr()
                    self._kv = (kv,)
                if esz == 1:
                    self._check(self.lib.am_bind_kv8_buffers(self.handle, base, base + n.value, 2 * n.value), "am_bind_kv8_buffers")
                else:
                    self._check(self.lib.am_bind_kv_buffers(self.handle, base, base + n.value * 2, 2 * n.value), "am_bind_kv_buffers")
        self._shape = None

    def close(self, collective: bool = True):
        if getattr(self, "handle", None) is not None and self.handle:
            self.lib.am_destroy(self.handle)
            self.handle = None
        ex = getattr(self, "exchange", None)
        if ex is not None:
            self.exchange = None
            ex.close(collective=collective)

    def __del__(self):
        try:
            self.close(collective=False)      # no collectives from a finalizer
        except Exception:
            pass

    def fits(self, B: int, T: int, N: int, S: int) -> bool:
        b = self.bounds
        return B <= b[0] and T <= b[1] and N <= b[2] and S <= b[3]

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def is_inflated(self, layer: int) -> bool:
        return layer in self.inflated

    def kv_buffers(self):
        return self._kv

    def set_context(self, ctx_local: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                    ctx_zero: Optional[Sequence[bool]] = None, shared_prefix: bool = False) -> None:
        """ctx_local (B, T_local, S, Dc) fp32 device; cos/sin (B*T_local, 64) fp32 host.
        `ctx_zero[b]`: the context of batch row b is identically zero; `shared_prefix`: every batch row carries the same
        hidden_states and t_bt - the two exact shortcuts of am_set_branch_hints (include/actionmesh_amd.h)."""
        B, T, S, _ = ctx_local.shape
        ctx_local = ctx_local.to(self.device, torch.float32).contiguous()
        cos = cos.contiguous(); sin = sin.contiguous()
        assert cos.shape == (B * T, HEAD_DIM // 2) and not cos.is_cuda
        with torch.cuda.device(self.device):
            self._check(self.lib.am_set_context(self.handle, ctx_local.data_ptr(), B, T, S,
                                            cos.data_ptr(), sin.data_ptr(), self._stream()), "am_set_context")
            if (ctx_zero is not None and any(ctx_zero)) or shared_prefix:
                z = (C.c_uint8 * B)(*[1 if (ctx_zero is not None and ctx_zero[b]) else 0 for b in range(B)])
                self._check(self.lib.am_set_branch_hints(self.handle, z, 1 if shared_prefix else 0), "am_set_branch_hints")
        self._ctx_keepalive = ctx_local

    def begin(self, x_local: torch.Tensor, t_bt_local: List[float]) -> None:
        B, T, N, D = x_local.shape
        x_local = x_local.to(self.device, torch.float32).contiguous()
        t = (C.c_float * (B * T))(*t_bt_local)
        self._shape = (B, T, N, D)
        self._x_keepalive = x_local
        with torch.cuda.device(self.device):
            self._check(self.lib.am_forward_begin(self.handle, x_local.data_ptr(), t, B, T, N, self._stream()),
                    "am_forward_begin")

    def layer_pre(self, layer: int) -> None:
        with torch.cuda.device(self.device):
            self._check(self.lib.am_layer_pre_attn(self.handle, layer, self._stream()), "am_layer_pre_attn")

    def layer_attn_local(self, layer: int) -> None:
        """Optional overlap step: attention against the local K/V shard while the all-gather is in flight."""
        with torch.cuda.device(self.device):
            self._check(self.lib.am_layer_attn_local(self.handle, layer, self._stream()), "am_layer_attn_local")

    def layer_post(self, layer: int) -> None:
        with torch.cuda.device(self.device):
            self._check(self.lib.am_layer_post_attn(self.handle, layer, self._stream()), "am_layer_post_attn")

    def end(self) -> torch.Tensor:
        B, T, N, D = self._shape
        v = torch.empty((B, T, N, D), dtype=self.h16, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.am_forward_end(self.handle, v.data_ptr(), self._stream()), "am_forward_end")
        return v

    def forward_peer(self, x_local: torch.Tensor, t_bt_local: List[float]) -> torch.Tensor:
        """The frame-sharded forward of this rank over the copy-engine exchange (`self.exchange`, a sharding.PeerExchange) in ONE C
        call: am_forward_sharded_peer (include/actionmesh_amd_sharded.h) - begin, per layer pre / push + local attention / wait / post /
        consumed, end.  What sharding.sharded_forward(exchange=...) does from Python, launch for launch."""
        ex = self.exchange
        if ex is None or self.world <= 1:
            raise RuntimeError("HipEngine.forward_peer: no copy-engine exchange is bound (world > 1 with a PeerExchange kv_factory)")
        B, T, N, D = x_local.shape
        x_local = x_local.to(self.device, torch.float32).contiguous()
        t = (C.c_float * (B * T))(*t_bt_local)
        v = torch.empty((B, T, N, D), dtype=self.h16, device=self.device)
        infl = (C.c_uint8 * self.num_layers)(*[1 if self.is_inflated(i) else 0 for i in range(self.num_layers)])
        ring = ex.ring()
        ok = False
        try:
            with torch.cuda.device(self.device):
                self._check(self.lib.am_forward_sharded_peer(self.handle, x_local.data_ptr(), t, B, T, N, v.data_ptr(), C.byref(ring), infl,
                                                             self.num_layers, self._stream()), "am_forward_sharded_peer")
            ok = True
        finally:
            # ring->seq has advanced by every exchange the call got through, whether it returned an error or not: the Python counter
            # follows it in every case; after an error the flag state of the peers is unknown, so the exchange is poisoned
            ex.sync_seq()
            if not ok:
                ex.poisoned = True
        self._x_keepalive = x_local
        return v

    def forward(self, x_local: torch.Tensor, t_bt_local: List[float]) -> torch.Tensor:
        """Single-rank convenience: the whole forward in one C call."""
        B, T, N, D = x_local.shape
        x_local = x_local.to(self.device, torch.float32).contiguous()
        t = (C.c_float * (B * T))(*t_bt_local)
        if self.use_graph:
            # the captured forward reads / writes THESE buffers (fixed addresses); the caller gets a copy of the result
            # PyTorch's default stream is the null stream, which cannot be captured: the forward runs on a stream of the engine,
            # ordered after the caller's current stream and before whatever the caller enqueues next
            if self._gx is None or self._gx.shape != x_local.shape:
                self._gx = torch.empty_like(x_local)
                self._gv = torch.empty((B, T, N, D), dtype=self.h16, device=self.device)
                self._gstream = torch.cuda.Stream(self.device)
            with torch.cuda.device(self.device):
                cur = torch.cuda.current_stream(self.device)
                self._gstream.wait_stream(cur)
                x_local.record_stream(self._gstream)
                with torch.cuda.stream(self._gstream):
                    self._gx.copy_(x_local)
                    self._check(self.lib.am_denoise_forward_graph(self.handle, self._gx.data_ptr(), t, B, T, N, self._gv.data_ptr(),
                                                              self._gstream.cuda_stream), "am_denoise_forward_graph")
                cur.wait_stream(self._gstream)
            # the graph writes ONE persistent buffer: hand out a copy, or a caller that keeps several results (split_cfg_batch
            # collects one velocity per guidance branch and concatenates them) would see them all alias the last forward
            return self._gv.clone()
        v = torch.empty((B, T, N, D), dtype=self.h16, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.am_denoise_forward(self.handle, x_local.data_ptr(), t, B, T, N, v.data_ptr(),
                                                self._stream()), "am_denoise_forward")
        return v

    def attention_counters(self) -> Tuple[int, int]:
        """(fp8, bf16) inflated self-attention launches of this engine so far: the arithmetic type that really ran."""
        c = (C.c_uint64 * 2)()
        self._check(self.lib.am_attention_counters(self.handle, c), "am_attention_counters")
        return int(c[0]), int(c[1])

    def graph_stats(self) -> Tuple[int, int, int, int]:
        """(replays, captures, eager forwards, capture failed) of am_denoise_forward_graph."""
        c = (C.c_uint64 * 4)()
        self._check(self.lib.am_graph_stats(self.handle, c), "am_graph_stats")
        return tuple(int(v) for v in c)

    def step_flops(self, B: int, T_total: int, N: int, S: int) -> float:
        return float(self.lib.am_step_flops(self.handle, B, T_total, N, S))


class HipDenoiser(nn.Module):
    """Drop-in replacement of ActionMeshDenoiser (see module docstring).

    `process_group`: when given (world > 1) each call is sharded across the group's ranks (one
    process per GPU): the CFG branches first (`cfg_parallel`, when the batch and the world are
    even), then the frames; the K/V all-gather of the inflated layers runs on RCCL inside each
    frame group.  Every rank passes the full (B, T, ...) tensors and receives the full velocity,
    like the reference.
    """

    def __init__(self, num_tokens_nominal: int = 2048, temporal_context_size: int = 16,
                 in_channels: int = 64, num_layers: int = 21, num_attention_heads: int = 16,
                 width: int = 2048, mlp_ratio: float = 4.0, cross_attention_dim: int = 1024,
                 inflated_layers: Optional[Sequence[int]] = None, clear_autocast: bool = True,
                 compile_blocks: bool = False, compile_mode: str = "default",
                 process_group: Optional[dist.ProcessGroup] = None, attn_defer_log2: int = 8,
                 cfg_parallel: bool = True, attn_dtype: str = "bf16", use_graph: Optional[bool] = None, dtype=None):
        super().__init__()
        # 16-bit storage / MFMA type.  None (default): follow the caller's autocast region like the reference module does - the reference
        # pipeline runs Stage I under torch.autocast("cuda", dtype) (pipeline.py:671) with dtype from the CLI's --dtype {bfloat16,
        # float16}; bfloat16 outside any autocast region.  "float16" / "bfloat16" (or the torch dtypes) pin it.
        self.dtype_pinned = None if dtype is None else L.kind_of(dtype)
        # HIP-graph replay of the single-rank forward (None: the ACTIONMESH_AMD_GRAPH environment variable, default off)
        self.use_graph = (os.environ.get("ACTIONMESH_AMD_GRAPH", "0") == "1") if use_graph is None else bool(use_graph)
        self.attn_dtype = attn_dtype        # "fp8": inflated self-attention on the e4m3 MFMA kernel (BASELINE configs[4])
        if width != num_attention_heads * HEAD_DIM:
            raise ValueError("HipDenoiser supports head_dim 128 only (width = heads * 128), as the reference ships")
        self.num_tokens_nominal = num_tokens_nominal
        self.temporal_context_size = temporal_context_size
        self.in_channels = in_channels
        self.out_channels = in_channels
        self.num_layers = num_layers
        self.num_attention_heads = num_attention_heads
        self.width = width
        self.width_per_head = HEAD_DIM
        self.mlp_ratio = mlp_ratio
        self.cross_attention_dim = cross_attention_dim
        self.inflated_layers = tuple(range(num_layers)) if inflated_layers is None else tuple(inflated_layers)
        self.attn_defer_log2 = attn_defer_log2
        self.process_group = process_group
        self.cfg_parallel = cfg_parallel
        self._frame_groups: Dict[int, List] = {}     # cfg_groups -> [ProcessGroup per CFG branch]
        self._cfg_peer_groups: Dict[int, List] = {}  # cfg_groups -> [ProcessGroup per position inside a CFG group]
        self.register_buffer("_device_probe", torch.zeros(1), persistent=False)
        self._host_sd: Optional[Dict[str, torch.Tensor]] = None
        self._engine: Optional[HipEngine] = None
        self._generation = 0
        self._window: Optional[WindowCache] = None

    # ---- reference-compatible surface ------------------------------------------------
    @property
    def device(self) -> torch.device:
        return self._device_probe.device

    def hyper_params(self) -> Dict:
        return dict(in_channels=self.in_channels, num_layers=self.num_layers,
                    num_attention_heads=self.num_attention_heads, width=self.width,
                    mlp_ratio=self.mlp_ratio, cross_attention_dim=self.cross_attention_dim,
                    inflated_layers=list(self.inflated_layers))

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = {k.replace("_orig_mod.", ""): v.detach().to("cpu", torch.float32) for k, v in state_dict.items()}
        self._host_sd = sd
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        return nn.modules.module._IncompatibleKeys([], [])

    @classmethod
    def from_pretrained(cls, path: str, **kwargs) -> "HipDenoiser":
        """Reads the PyTorchModelHubMixin layout the reference uses (pipeline.py:180-184):
        <path>/config.json + <path>/model.safetensors."""
        from safetensors.torch import load_file
        with open(os.path.join(path, "config.json")) as f:
            cfg = json.load(f)
        fields = ("num_tokens_nominal", "temporal_context_size", "in_channels", "num_layers",
                  "num_attention_heads", "width", "mlp_ratio", "cross_attention_dim", "inflated_layers")
        model = cls(**{k: cfg[k] for k in fields if k in cfg}, **kwargs)
        model.load_state_dict(load_file(os.path.join(path, "model.safetensors")))
        return model

    # ---- engine management ---------------------------------------------------------------
    def _plan(self, T: int, B: int = 1) -> FrameShardPlan:
        if self.process_group is None:
            return FrameShardPlan(T, 1, 0, B, 1)
        world = dist.get_world_size(self.process_group)
        rank = dist.get_rank(self.process_group)
        groups = 2 if (self.cfg_parallel and world % 2 == 0 and B % 2 == 0) else 1
        return FrameShardPlan(T, world, rank, B, groups)

    def _frame_group(self, plan: FrameShardPlan):
        """Process group of the ranks that share this rank's CFG branch (created collectively once)."""
        if plan.cfg_groups == 1:
            return self.process_group
        if plan.cfg_groups not in self._frame_groups:
            base = dist.get_process_group_ranks(self.process_group)
            be = dist.get_backend(self.process_group)      # sub-groups on the SAME backend as the group handed in, not the default group's
            self._frame_groups[plan.cfg_groups] = [
                dist.new_group([base[r] for r in plan.frame_group_ranks(g)], backend=be) for g in range(plan.cfg_groups)]
        return self._frame_groups[plan.cfg_groups][plan.cfg_rank]

    def _cfg_peer_group(self, plan: FrameShardPlan):
        """Process group of the ranks that hold the SAME frame shard of the other CFG branches (created collectively once): the
        only ranks a sampler that keeps its latents sharded has to hear from in a step (forward_host_time(gather=False))."""
        if plan.cfg_groups not in self._cfg_peer_groups:
            base = dist.get_process_group_ranks(self.process_group)
            be = dist.get_backend(self.process_group)
            self._cfg_peer_groups[plan.cfg_groups] = [
                dist.new_group([base[r] for r in plan.cfg_peer_ranks(i)], backend=be) for i in range(plan.group_size)]
        return self._cfg_peer_groups[plan.cfg_groups][plan.rank % plan.group_size]

    def compute_kind(self) -> str:
        """'bf16' or 'f16': the pinned dtype, else the autocast dtype of the calling region (float16 only when the caller asked for it)."""
        return L.autocast_kind(self.dtype_pinned)

    def _ensure_engine(self, B: int, T_local: int, N: int, S: int, plan: FrameShardPlan) -> HipEngine:
        if self._host_sd is None:
            raise RuntimeError("HipDenoiser: no weights loaded (load_state_dict / from_pretrained first)")
        e = self._engine
        kind = self.compute_kind()
        if e is not None and e.device == self.device and e.fits(B, T_local, N, S) and e.world == plan.frame_world \
                and e.rank == plan.frame_rank and e.kind == kind:
            return e
        if e is not None:
            e.close()
        kv_factory = None
        if plan.frame_world > 1 and os.environ.get("ACTIONMESH_AMD_EXCHANGE", "rccl") == "peer":
            from .sharding import PeerExchange
            group = self._frame_group(plan)
            kv_factory = lambda chunk_bytes: PeerExchange(group, plan, chunk_bytes, self.device)
        self._engine = HipEngine(self.hyper_params(), self._host_sd, self.device, B, T_local, N, S,
                                 world=plan.frame_world, rank=plan.frame_rank, attn_defer_log2=self.attn_defer_log2,
                                 attn_dtype=self.attn_dtype, kv_factory=kv_factory, use_graph=self.use_graph, dtype=kind)
        self._window = None
        return self._engine

    def bind_window(self, context: torch.Tensor, framestep: torch.Tensor, n_tokens: int,
                    ctx_zero: Optional[Sequence[bool]] = None, shared_prefix: bool = False) -> WindowCache:
        """Build the step-invariant state for one window: RoPE table and cross-attention K/V.
        `ctx_zero` (per batch row; None = find out with one device reduction) and `shared_prefix` enable the exact shortcuts of
        am_set_branch_hints; ACTIONMESH_AMD_NO_SHORTCUTS=1 turns both off."""
        B, T, S, _ = context.shape
        if os.environ.get("ACTIONMESH_AMD_NO_SHORTCUTS", "0") == "1":
            ctx_zero, shared_prefix = [False] * B, False
        elif ctx_zero is None:
            ctx_zero = [not bool(f) for f in context.reshape(B, -1).ne(0).any(dim=1).tolist()]
        plan = self._plan(T, B)
        e = self._ensure_engine(plan.batch_local, plan.frames_local, n_tokens, S, plan)
        cos, sin = rope_tables_host(framestep, HEAD_DIM)        # from the FULL window's framesteps
        cos = plan.slice_local(cos.view(B, T, -1)).reshape(-1, HEAD_DIM // 2)
        sin = plan.slice_local(sin.view(B, T, -1)).reshape(-1, HEAD_DIM // 2)
        e.set_context(plan.slice_local(context), cos, sin, ctx_zero=list(ctx_zero)[plan.batch_slice],
                      shared_prefix=shared_prefix and plan.world == 1)
        self._generation += 1
        self._window = WindowCache(self._generation, context, framestep, n_tokens)
        return self._window

    def _apply(self, fn, recurse: bool = True):
        """`.to()` / `.cpu()` / `.cuda()`: the engine (weights, workspace, K/V caches in HBM) belongs to one device.
        Leaving it - the reference's `--low_ram` unload moves the denoiser to the CPU between stages
        (pipeline.py:171-184) - frees every byte the engine holds; the host copy of the weights stays, and the next
        forward on a GPU builds a fresh engine there."""
        out = super()._apply(fn, recurse)
        e = self._engine
        if e is not None and e.device != self.device:
            e.close()
            self._engine = None
            self._window = None
        return out

    def forward_host_time(self, hidden_states: torch.Tensor, t_bt: List[float], gather: bool = True) -> torch.Tensor:
        """Forward with the masked per-(b,t) diffusion times already on the host.
        `gather=False` (a sampler that keeps its latents sharded across the steps): returns the velocity of ALL batch
        rows for THIS rank's frames only - (B, T_local, N, D), frames `frame_slice(T, B)` - and reads only those frames of
        `hidden_states`.  With the batch on one CFG group that is the local result and the step has no velocity collective at
        all; with the CFG branches split over groups it is one all-gather among the ranks that hold the same frames
        (1 / frame_world of the bytes of the full gather)."""
        B, T, N, _ = hidden_states.shape
        plan = self._plan(T, B)
        e = self._engine
        if e is None or self._window is None:
            raise RuntimeError("HipDenoiser: bind_window() must precede forward_host_time()")
        if plan.world == 1:
            return e.forward(hidden_states, t_bt)
        x_local, t_local = plan.slice_local(hidden_states), plan.local_times(t_bt)
        if plan.frame_world == 1:                       # pure CFG split: no K/V exchange at all
            v_local = e.forward(x_local, t_local)
        else:
            v_local = sharded_forward(e, plan, self._frame_group(plan), x_local, t_local, exchange=getattr(e, "exchange", None))
        if not gather:
            if plan.cfg_groups == 1:
                return v_local
            self._gather_cfg_out = gather_cfg(v_local, plan, self._cfg_peer_group(plan), out=getattr(self, "_gather_cfg_out", None))
            return self._gather_cfg_out
        self._gather_out = gather_frames(v_local, plan, self.process_group, out=getattr(self, "_gather_out", None))
        return self._gather_out[1]

    def frame_slice(self, T: int, B: int = 1) -> slice:
        """The frames forward_host_time(gather=False) covers on this rank."""
        return self._plan(T, B).frame_slice

    def gather_latent_frames(self, latents: torch.Tensor, B: int = 1) -> torch.Tensor:
        """latents (T, ...) with this rank's frames current -> all frames current on every rank (in place, one all-gather over
        the frame group): what a sampler that kept its latents sharded calls once, after the last step."""
        plan = self._plan(latents.shape[0], B)
        if plan.world > 1 and plan.frame_world > 1:
            gather_latent_frames(latents, plan, self._frame_group(plan))
        return latents

    def check_exchange(self, block: bool = True) -> None:
        """Raise if a flag wait of the copy-engine exchange gave up (a peer died or fell > 20 s behind: the attention then read
        stale or partial shards).  `block=False` looks at the copy of the fault word the LAST forward left in pinned host memory
        (no device sync: HipSchedulerFlow polls it once per step, a step late at worst - the word is sticky); `block=True`
        synchronises the stream first (the end of a sampling loop, and every forward of the reference-driven S2 seam, whose own
        loop syncs once per step anyway, scheduler.py:245)."""
        ex = getattr(self._engine, "exchange", None) if self._engine is not None else None
        if ex is not None and ex.faulted(block=block):
            raise RuntimeError("HipDenoiser: the copy-engine exchange timed out waiting for a peer's K/V shard; the result is invalid")

    def forward(self, hidden_states: torch.Tensor, context: torch.Tensor, framestep: torch.Tensor,
                diffusion_time: torch.Tensor, mask: Optional[torch.Tensor] = None,
                freqs_rot: Optional[WindowCache] = None):
        B, T, N, _ = hidden_states.shape
        if not (isinstance(freqs_rot, WindowCache) and self._window is not None
                and freqs_rot.generation == self._window.generation
                and self._window.matches(context, framestep, N)):
            freqs_rot = self.bind_window(context, framestep, N)
        # The C ABI takes the per-frame times from the host.  The reference sampler hands over device tensors every step
        # (scheduler.py:151-168): the mask of a window never changes, so it is downloaded once per window and the step's time is
        # ONE device-to-host read (the reference's own loop syncs on it as well, scheduler.py:245).
        w = self._window
        mkey = None if mask is None else _tensor_identity(mask)
        if getattr(w, "_mask_key", "unset") != mkey:
            w._mask_key, w._mask_host = mkey, (None if mask is None else mask.detach().float().cpu())
            # keep the STORAGE alive while its address is part of the key: the caching allocator could otherwise hand the same address
            # (with _version 0) to another mask of the same shape and the stale host copy would match it
            w._mask_storage = None if mask is None else mask.untyped_storage()
        t_bt = masked_time(diffusion_time.detach().float().cpu().tolist(), w._mask_host, B, T)
        v = self.forward_host_time(hidden_states, t_bt)
        self.check_exchange(block=True)
        if self.process_group is not None:
            v = v.clone()          # the sharded path hands out its re-used gather buffer; the reference sampler may keep several results
        return v, freqs_rot
