"""CPU restatement of the reference's Stage-II decoder, ActionMeshAutoencoder.forward (TEST INFRASTRUCTURE ONLY -
imported by tests/ and tools/ checkers only; never on the product path).

Follows actionmesh/model/temporal_autoencoder.py:80-267 with
  * FlowMatchingBlock (block.py:110-154) in its two Stage-II forms: self-attention + FF (qk_norm None, no attention
    bias) and cross-attention + FF (cross_attention_norm "layer_norm": nn.LayerNorm on the kv cache);
  * AttentionProcessor (attention_processor.py:36-168): the per-head [q|k|v] (resp. [k|v]) split of the CONCATENATED
    projections, RoPE on q and k of the self-attention, plain SDPA;
  * TimestepEmbedder / FrequencyPositionalEmbedding (embeddings.py:14-130), scale_timestep (embeddings.py:135-153),
    compute_rotary_embeddings / apply_rotary_embedding (rotary_embedding.py:10-124, shared with denoiser_oracle).
Everything in fp32 (the reference's cuda autocast is inert on CPU; its cross-attention part is fp32 by construction).
Pinned against the reference's own unmodified module by tests/golden/ae_tiny.npz (oracle/make_golden_autoencoder.py).
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

from . import denoiser_oracle as O

Tensor = torch.Tensor


@dataclass(frozen=True)
class AEConfig:
    in_channels: int = 3
    in_extra_channels: int = 3
    out_dim: int = 3
    latent_channels: int = 64
    width: int = 1024
    num_layers: int = 16
    num_attention_heads: int = 8
    embed_frequency: int = 8
    embed_include_pi: bool = False

    @property
    def head_dim(self) -> int:
        return self.width // self.num_attention_heads

    @property
    def query_dim(self) -> int:             # FrequencyPositionalEmbedding.out_dim + extra channels (:77)
        return self.in_channels * (2 * self.embed_frequency + 1) + self.in_extra_channels


def state_dict_spec(cfg: AEConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) of every parameter, in the reference module's state-dict order."""
    C, Fi = cfg.width, 4 * cfg.width
    spec: List[Tuple[str, Tuple[int, ...]]] = []
    for i in range(cfg.num_layers):
        p = f"blocks.{i}."
        spec += [(p + "norm_s_attn.weight", (C,)), (p + "norm_s_attn.bias", (C,)),
                 (p + "s_attn.to_q.weight", (C, C)), (p + "s_attn.to_k.weight", (C, C)), (p + "s_attn.to_v.weight", (C, C)),
                 (p + "s_attn.to_out.0.weight", (C, C)), (p + "s_attn.to_out.0.bias", (C,)),
                 (p + "norm_ff.weight", (C,)), (p + "norm_ff.bias", (C,)),
                 (p + "ff.net.0.proj.weight", (Fi, C)), (p + "ff.net.0.proj.bias", (Fi,)),
                 (p + "ff.net.2.weight", (C, Fi)), (p + "ff.net.2.bias", (C,))]
    p = f"blocks.{cfg.num_layers}."
    spec += [(p + "norm_x_attn.weight", (C,)), (p + "norm_x_attn.bias", (C,)),
             (p + "x_attn.norm_cross.weight", (C,)), (p + "x_attn.norm_cross.bias", (C,)),
             (p + "x_attn.to_q.weight", (C, C)), (p + "x_attn.to_k.weight", (C, C)), (p + "x_attn.to_v.weight", (C, C)),
             (p + "x_attn.to_out.0.weight", (C, C)), (p + "x_attn.to_out.0.bias", (C,)),
             (p + "norm_ff.weight", (C,)), (p + "norm_ff.bias", (C,)),
             (p + "ff.net.0.proj.weight", (Fi, C)), (p + "ff.net.0.proj.bias", (Fi,)),
             (p + "ff.net.2.weight", (C, Fi)), (p + "ff.net.2.bias", (C,))]
    spec += [("proj_query.weight", (C, cfg.query_dim)), ("proj_query.bias", (C,)),
             ("norm_out.weight", (C,)), ("norm_out.bias", (C,)),
             ("proj_out.weight", (cfg.out_dim, C)), ("proj_out.bias", (cfg.out_dim,)),
             ("post_quant.weight", (C, cfg.latent_channels)), ("post_quant.bias", (C,))]
    return spec


def synthetic_state_dict(cfg: AEConfig, seed: int = 0) -> Dict[str, Tensor]:
    """Deterministic stand-in weights (no pretrained weights offline): every tensor from its own crc32-seeded generator,
    linear weights ~ N(0, 1/fan_in), norm gains around 1, biases small."""
    sd: Dict[str, Tensor] = {}
    for name, shape in state_dict_spec(cfg):
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        if name.endswith(".weight") and len(shape) == 2:
            t = torch.randn(shape, generator=g) / math.sqrt(shape[1])
        elif name.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = 0.05 * torch.randn(shape, generator=g)
        sd[name] = t
    return sd


def state_dict_checksum(sd: Dict[str, Tensor]) -> float:
    return float(sum(float(v.double().abs().sum()) * (1 + (zlib.crc32(k.encode()) % 97) / 97.0) for k, v in sd.items()))


def timestep_embed(freq_size: int, *ts: Tensor, max_period: int = 10_000) -> Tensor:
    """TimestepEmbedder.forward (embeddings.py:55-130): per input [cos | sin] of t * exp(-ln(max_period) i / half)."""
    half = freq_size // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    outs = []
    for t in ts:
        args = t[..., None].float() * freqs
        outs += [torch.cos(args), torch.sin(args)]
    return torch.cat(outs, dim=-1)


def point_embed(cfg: AEConfig, x: Tensor) -> Tensor:
    """FrequencyPositionalEmbedding.forward (embeddings.py:14-52), logspace, include_input."""
    freqs = 2.0 ** torch.arange(cfg.embed_frequency, dtype=torch.float32)
    if cfg.embed_include_pi:
        freqs = freqs * torch.pi
    embed = (x[..., None] * freqs).view(*x.shape[:-1], -1)
    return torch.cat((x, embed.sin(), embed.cos()), dim=-1)


def _heads(x: Tensor, H: int) -> Tensor:
    B, Lx, C = x.shape
    return x.view(B, Lx, H, C // H).transpose(1, 2)


def self_block(sd, cfg: AEConfig, i: int, h: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    p = f"blocks.{i}."
    H, hd = cfg.num_attention_heads, cfg.head_dim
    z = O.fp32_layer_norm(h, sd[p + "norm_s_attn.weight"], sd[p + "norm_s_attn.bias"])
    qkv = torch.cat([F.linear(z, sd[p + f"s_attn.{n}.weight"]) for n in ("to_q", "to_k", "to_v")], -1)
    B, Lx, _ = qkv.shape
    q, k, v = torch.split(qkv.view(B, Lx, H, 3 * hd), hd, dim=-1)                # attention_processor.py:105-110
    q, k, v = (t.transpose(1, 2) for t in (q, k, v))
    q, k = O.apply_rope(q, cos, sin), O.apply_rope(k, cos, sin)                  # :128-131 (no qk norm)
    a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, Lx, H * hd)
    h = h + F.linear(a, sd[p + "s_attn.to_out.0.weight"], sd[p + "s_attn.to_out.0.bias"])
    z = O.fp32_layer_norm(h, sd[p + "norm_ff.weight"], sd[p + "norm_ff.bias"])
    z = F.gelu(F.linear(z, sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"]))
    return h + F.linear(z, sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"])


def cross_block(sd, cfg: AEConfig, queries: Tensor, kv_cache: Tensor) -> Tensor:
    p = f"blocks.{cfg.num_layers}."
    H, hd = cfg.num_attention_heads, cfg.head_dim
    z = O.fp32_layer_norm(queries, sd[p + "norm_x_attn.weight"], sd[p + "norm_x_attn.bias"])
    q = _heads(F.linear(z, sd[p + "x_attn.to_q.weight"]), H)
    e = F.layer_norm(kv_cache, (cfg.width,), sd[p + "x_attn.norm_cross.weight"], sd[p + "x_attn.norm_cross.bias"], 1e-5)
    kv = torch.cat([F.linear(e, sd[p + "x_attn.to_k.weight"]), F.linear(e, sd[p + "x_attn.to_v.weight"])], -1)
    B, S, _ = kv.shape
    k, v = torch.split(kv.view(B, S, H, 2 * hd), hd, dim=-1)                     # attention_processor.py:111-115
    a = F.scaled_dot_product_attention(q, k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2).reshape(B, -1, H * hd)
    h = queries + F.linear(a, sd[p + "x_attn.to_out.0.weight"], sd[p + "x_attn.to_out.0.bias"])
    z = O.fp32_layer_norm(h, sd[p + "norm_ff.weight"], sd[p + "norm_ff.bias"])
    z = F.gelu(F.linear(z, sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"]))
    return h + F.linear(z, sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"])


def autoencoder_forward(sd: Dict[str, Tensor], cfg: AEConfig, latent: Tensor, framestep: Tensor, source_alpha: Tensor,
                        target_alphas: Tensor, query: Tensor, return_kv_cache: bool = False):
    """temporal_autoencoder.py:160-267.  latent (B,T,N,D), framestep (B,T), source_alpha (B,), target_alphas (B,T_out),
    query (B,V,3|6) -> displacement (B,T_out,V,out_dim) in [-1,1]."""
    B, T, N, _ = latent.shape
    T_out = target_alphas.shape[1]
    hd = cfg.head_dim
    lat = F.linear(latent, sd["post_quant.weight"], sd["post_quant.bias"]).reshape(B, T * N, cfg.width)   # :203
    cos, sin = O.rope_tables(framestep, hd)          # scale_timestep(center) + merge (:198-200, :206-209): (B*T, hd)
    cos, sin = cos.reshape(B, T, hd), sin.reshape(B, T, hd)
    cos = torch.cat([cos.repeat_interleave(N, dim=1), cos], dim=1)               # :214-229: latent tokens, then alpha tokens
    sin = torch.cat([sin.repeat_interleave(N, dim=1), sin], dim=1)
    src = source_alpha[:, None].expand_as(target_alphas)
    alpha = timestep_embed(cfg.width // 2, src, target_alphas)[:, None].repeat(1, T, 1, 1)     # (B, T, T_out, C) :232-235
    qe = point_embed(cfg, query[..., :3])
    if cfg.in_extra_channels > 0:
        qe = torch.cat([qe, query[..., 3:]], dim=-1)
    out = torch.empty((B, T_out, query.shape[1], cfg.out_dim))
    caches = []
    for i in range(T_out):
        h = torch.cat([lat, alpha[:, :, i]], dim=1)                              # :254
        for li in range(cfg.num_layers):
            h = self_block(sd, cfg, li, h, cos, sin)
        caches.append(h)
        qh = F.linear(qe, sd["proj_query.weight"], sd["proj_query.bias"])        # :149-157
        lg = cross_block(sd, cfg, qh, h)
        lg = F.linear(F.layer_norm(lg, (cfg.width,), sd["norm_out.weight"], sd["norm_out.bias"], 1e-5),
                      sd["proj_out.weight"], sd["proj_out.bias"]) * -1
        out[:, i] = lg
    disp = 2 * torch.sigmoid(out) - 1.0
    return (disp, caches) if return_kv_cache else disp
