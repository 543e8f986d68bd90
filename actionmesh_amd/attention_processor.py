"""HipAttentionProcessor: drop-in for actionmesh.model.utils.attention_processor.AttentionProcessor
(reference attention_processor.py:19-168) - seam S3 of SURVEY.md section 8(b), the finest-grain plug-in point: a
diffusers `Attention` module keeps owning the parameters, `Attention.forward` keeps filtering kwargs by this
`__call__`'s signature (block.py:137-149), and the arithmetic of the call - Q/K/V projections, the interleaved per-head
split, qk-RMSNorm, RoPE, non-causal SDPA, out-projection - runs in libactionmesh_amd.so through the kernel-level C-ABI
(am_gemm_bf16 / am_head_post / am_attention_bf16).  Same name, argument meaning and error behaviour as the reference
class; everything else of the block (LayerNorm, residual adds, FeedForward) stays PyTorch at this seam.

Usage (reference side):  `Attention(..., processor=HipAttentionProcessor())`  in block.py:67-96.
There is no torch fallback: a CPU tensor or a missing library raises.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import _lib as L
from . import ops

HEAD_DIM = 128


def _version_key(*tensors) -> Tuple:
    return tuple((t.data_ptr(), t._version, tuple(t.shape), str(t.device)) if t is not None else None for t in tensors)


class HipAttentionProcessor:
    r"""
    Processor for implementing the scaled dot-product attention (MI355X / gfx950 HIP kernels).
    """

    def __init__(self):
        # the reference checks that a flash / memory-efficient SDP backend exists (attention_processor.py:24-34);
        # the equivalent here is that the HIP library is built - fail at construction, not at the first call
        L.lib()
        self._weights: Dict[int, Tuple[Tuple, Dict[str, torch.Tensor]]] = {}
        self._rope: Optional[Tuple[Tuple, Tuple[torch.Tensor, torch.Tensor]]] = None

    # ---- parameter packing (cached per Attention module, refreshed when a parameter changes) --------------------
    def _packed(self, attn, device) -> Dict[str, torch.Tensor]:
        ws = [attn.to_q.weight, attn.to_k.weight, attn.to_v.weight, attn.to_out[0].weight, attn.to_out[0].bias,
              None if attn.norm_q is None else attn.norm_q.weight, None if attn.norm_k is None else attn.norm_k.weight]
        key = _version_key(*ws) + (str(device),)
        hit = self._weights.get(id(attn))
        if hit is not None and hit[0] == key:
            return hit[1]
        for lin in (attn.to_q, attn.to_k, attn.to_v):
            if lin.bias is not None:
                raise NotImplementedError("HipAttentionProcessor: q/k/v projections with bias (the reference uses bias=False)")
        bf = lambda w: w.detach().to(device, torch.bfloat16).contiguous()
        f32 = lambda w: None if w is None else w.detach().to(device, torch.float32).contiguous()
        p: Dict[str, torch.Tensor] = {}
        if not attn.is_cross_attention:
            # attention_processor.py:106-110: head h reads columns [3hd*h, 3hd*(h+1)) of cat(q, k, v)
            # => the fused weight is the plain row concatenation [Wq; Wk; Wv]
            p["w_qkv"] = torch.cat([bf(attn.to_q.weight), bf(attn.to_k.weight), bf(attn.to_v.weight)], dim=0).contiguous()
        else:
            p["w_q"] = bf(attn.to_q.weight)
            p["w_kv"] = torch.cat([bf(attn.to_k.weight), bf(attn.to_v.weight)], dim=0).contiguous()   # :111-115
        p["w_o"] = bf(attn.to_out[0].weight)
        b = attn.to_out[0].bias
        # autocast casts the bias to bf16 together with the weight
        p["b_o"] = None if b is None else b.detach().to(device, torch.bfloat16).to(torch.float32).contiguous()
        p["n_q"] = f32(None if attn.norm_q is None else attn.norm_q.weight)
        p["n_k"] = f32(None if attn.norm_k is None else attn.norm_k.weight)
        self._weights[id(attn)] = (key, p)
        return p

    def _rope_tables(self, freqs_rot, frames: int, tokens: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
        """(cos, sin) each (frames, tokens, 128) with every frequency repeated twice (rotary_embedding.py:10-69), constant
        over the tokens of a frame (temporal_denoiser.py:114-149) -> the (frames, 64) fp32 tables am_head_post reads."""
        cos, sin = freqs_rot
        key = _version_key(cos, sin)
        if self._rope is not None and self._rope[0] == key:
            return self._rope[1]
        if cos.shape[-1] != HEAD_DIM or cos.shape[0] != frames:
            raise ValueError(f"freqs_rot: expected ({frames}, {tokens}, {HEAD_DIM}) tables, got {tuple(cos.shape)}")
        for t in (cos, sin):
            if t.dim() == 3 and not bool((t == t[:, :1]).all()):
                raise NotImplementedError("HipAttentionProcessor: RoPE angles that vary inside a frame (the reference "
                                          "rotates by the frame index only)")
        pick = lambda t: (t[:, 0] if t.dim() == 3 else t)[:, 0::2].to(device, torch.float32).contiguous()
        tabs = (pick(cos), pick(sin))
        self._rope = (key, tabs)
        return tabs

    def __call__(
        self,
        attn,
        hidden_states: torch.Tensor,
        encoder_hidden_states: Optional[torch.Tensor] = None,
        attention_mask: Optional[torch.Tensor] = None,
        temb: Optional[torch.Tensor] = None,
        inflate_self_attention: bool = False,
        freqs_rot: Optional[torch.Tensor] = None,
        n_frames: Optional[int] = None,
    ) -> torch.Tensor:
        if inflate_self_attention:
            assert n_frames is not None
        if attention_mask is not None:
            raise NotImplementedError("HipAttentionProcessor: attention_mask (the reference never passes one)")
        if attn.spatial_norm is not None or attn.group_norm is not None or getattr(attn, "norm_cross", None):
            raise NotImplementedError("HipAttentionProcessor: spatial_norm / group_norm / norm_cross are not on the reference path")
        if hidden_states.dim() != 3:
            raise NotImplementedError("HipAttentionProcessor: (batch, tokens, channels) inputs only")
        if not hidden_states.is_cuda:
            raise RuntimeError("HipAttentionProcessor needs device tensors (actionmesh_amd has no CPU path)")
        dev = hidden_states.device
        BT, Ltok, Cq = hidden_states.shape
        H = attn.heads
        if attn.to_q.weight.shape[0] != H * HEAD_DIM:
            raise ValueError(f"HipAttentionProcessor supports head_dim {HEAD_DIM} only (inner dim {attn.to_q.weight.shape[0]}, heads {H})")
        p = self._packed(attn, dev)
        residual = hidden_states
        with torch.cuda.device(dev):
            x = hidden_states.detach().reshape(BT * Ltok, Cq)
            x = ops.f32_to_bf16(x.contiguous()) if x.dtype == torch.float32 else x.to(torch.bfloat16).contiguous()
            if not attn.is_cross_attention:
                if encoder_hidden_states is not None:
                    raise NotImplementedError("HipAttentionProcessor: encoder_hidden_states on a self-attention module")
                seq = n_frames * Ltok if inflate_self_attention else Ltok      # flat_batch_to_flat_seq, tensor_ops.py:89-100
                rope = None if freqs_rot is None else self._rope_tables(freqs_rot, BT, Ltok, dev)
                # :92-103 + :106-130 in one launch (am_gemm_headpost_bf16; small shapes run the two kernels inside the entry point)
                q, k, vt = ops.gemm_head_post(x, p["w_qkv"], H, (0, 1, 2), seq, Ltok, w_q=p["n_q"], w_k=p["n_k"], rope=rope,
                                              eps=float(getattr(attn.norm_q, "eps", 1e-6) or 1e-6))
                o = ops.attention(q, k, vt, seq, seq)                                           # :133-139
            else:
                ctx = encoder_hidden_states
                if ctx is None:
                    raise ValueError("HipAttentionProcessor: a cross-attention module needs encoder_hidden_states")
                S, Dc = ctx.shape[1], ctx.shape[2]
                c = ctx.detach().reshape(BT * S, Dc)
                c = ops.f32_to_bf16(c.contiguous()) if c.dtype == torch.float32 else c.to(torch.bfloat16).contiguous()
                q, _, _ = ops.gemm_head_post(x, p["w_q"], H, (0,), Ltok, Ltok, w_q=p["n_q"],
                                             eps=float(getattr(attn.norm_q, "eps", 1e-6) or 1e-6))
                kv = ops.gemm(c, p["w_kv"])                                                     # :102-103, 111-115
                _, k, vt = ops.head_post(kv, H, (1, 2), S, S, w_k=p["n_k"], eps=float(getattr(attn.norm_k, "eps", 1e-6) or 1e-6))
                o = ops.attention(q, k, vt, Ltok, S)
            out = ops.gemm(o, p["w_o"], bias=p["b_o"])                                          # :147 (dropout p = 0)
        out = out.view(BT, Ltok, -1)
        if hidden_states.dtype != torch.bfloat16 and not torch.is_autocast_enabled():
            out = out.to(hidden_states.dtype)           # the reference returns the autocast dtype; fp32 outside autocast
        if attn.residual_connection:
            out = out + residual
        if attn.rescale_output_factor != 1.0:
            out = out / attn.rescale_output_factor
        return out
