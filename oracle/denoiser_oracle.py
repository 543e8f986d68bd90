"""CPU oracle for the Stage-I denoise hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch, CPU, functional *restatement* of the reference
algorithm (facebookresearch/actionmesh), written from the reference's source
with every function citing the file:line it follows (paths relative to
/root/reference/).  It is the checker the HIP path is compared against; it is
never the thing shipped or measured.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import it.

Parity pin: the reference has no tests or golden vectors for this path
(SURVEY.md section 4).  The pin is therefore made here: `oracle/make_golden.py`
runs the reference's OWN unmodified modules (with the un-vendored, unpinned
`diffusers` dependency supplied by `oracle/diffusers_shim`) in the build
container and commits inputs/outputs under `tests/golden/`;
`tests/test_oracle_golden.py` checks this restatement against those fixtures
and against the known-answer vectors of SURVEY.md App. D.

Two precision policies:
  * "fp32"  - what the reference computes on CPU (its `torch.autocast("cuda")`
              at pipeline.py:671 is inert on CPU).
  * "bf16"  - emulation of the reference's GPU dtype flow under
              autocast(bf16): every op that autocast makes
              return bf16 has its result rounded to bf16 here (arithmetic is
              done in fp32 on the rounded operands, which is what a bf16 GEMM
              with fp32 accumulation does).
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------
# configuration + deterministic synthetic weights
# ----------------------------------------------------------------------------
@dataclass
class OracleConfig:
    """Hyper-parameters of ActionMeshDenoiser (temporal_denoiser.py:29-48)."""

    in_channels: int = 64
    num_layers: int = 21
    num_attention_heads: int = 16
    width: int = 2048
    mlp_ratio: float = 4.0
    cross_attention_dim: int = 1024
    inflated_layers: Tuple[int, ...] = field(default_factory=lambda: tuple(range(21)))

    @property
    def head_dim(self) -> int:
        return self.width // self.num_attention_heads

    @property
    def ff_inner(self) -> int:
        return int(self.width * self.mlp_ratio)

    def has_skip(self, layer: int) -> bool:
        # temporal_denoiser.py:92  skip=layer > num_layers // 2
        return layer > self.num_layers // 2


def state_dict_spec(cfg: OracleConfig) -> Dict[str, Tuple[int, ...]]:
    """Parameter names and shapes of ActionMeshDenoiser.state_dict()
    (confirmed against the reference class by make_golden.py)."""
    C, F_, Dc, Din, hd = cfg.width, cfg.ff_inner, cfg.cross_attention_dim, cfg.in_channels, cfg.head_dim
    spec: Dict[str, Tuple[int, ...]] = {
        "time_proj.linear_1.weight": (4 * C, C), "time_proj.linear_1.bias": (4 * C,),
        "time_proj.linear_2.weight": (C, 4 * C), "time_proj.linear_2.bias": (C,),
        "proj_in.weight": (C, Din), "proj_in.bias": (C,),
    }
    for i in range(cfg.num_layers):
        p = f"blocks.{i}."
        spec[p + "norm_s_attn.weight"] = (C,); spec[p + "norm_s_attn.bias"] = (C,)
        spec[p + "s_attn.norm_q.weight"] = (hd,); spec[p + "s_attn.norm_k.weight"] = (hd,)
        for n in ("to_q", "to_k", "to_v"):
            spec[p + f"s_attn.{n}.weight"] = (C, C)
        spec[p + "s_attn.to_out.0.weight"] = (C, C); spec[p + "s_attn.to_out.0.bias"] = (C,)
        spec[p + "norm_x_attn.weight"] = (C,); spec[p + "norm_x_attn.bias"] = (C,)
        spec[p + "x_attn.norm_q.weight"] = (hd,); spec[p + "x_attn.norm_k.weight"] = (hd,)
        spec[p + "x_attn.to_q.weight"] = (C, C)
        spec[p + "x_attn.to_k.weight"] = (C, Dc); spec[p + "x_attn.to_v.weight"] = (C, Dc)
        spec[p + "x_attn.to_out.0.weight"] = (C, C); spec[p + "x_attn.to_out.0.bias"] = (C,)
        spec[p + "norm_ff.weight"] = (C,); spec[p + "norm_ff.bias"] = (C,)
        spec[p + "ff.net.0.proj.weight"] = (F_, C); spec[p + "ff.net.0.proj.bias"] = (F_,)
        spec[p + "ff.net.2.weight"] = (C, F_); spec[p + "ff.net.2.bias"] = (C,)
        if cfg.has_skip(i):
            spec[p + "norm_skip.weight"] = (C,); spec[p + "norm_skip.bias"] = (C,)
            spec[p + "linear_skip.weight"] = (C, 2 * C); spec[p + "linear_skip.bias"] = (C,)
    spec["norm_out.weight"] = (C,); spec["norm_out.bias"] = (C,)
    spec["proj_out.weight"] = (Din, C); spec["proj_out.bias"] = (Din,)
    return spec


def synthetic_state_dict(cfg: OracleConfig, seed: int = 0) -> Dict[str, Tensor]:
    """Deterministic random weights (no pretrained weights are reachable offline).

    Each tensor is drawn from its own CPU generator seeded by crc32(name)+seed,
    so the values do not depend on module construction order.  Linear weights
    and biases ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (the nn.Linear default
    scale); norm weights 1 + 0.1 N(0,1), norm biases 0.1 N(0,1) so that every
    affine term is exercised.
    """
    out: Dict[str, Tensor] = {}
    for name, shape in state_dict_spec(cfg).items():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + seed) & 0x7FFFFFFF)
        is_norm = ".norm_" in name or name.startswith("norm_out") or "norm_q" in name or "norm_k" in name
        if is_norm:
            r = torch.randn(shape, generator=g)
            out[name] = (1.0 + 0.1 * r) if name.endswith("weight") else 0.1 * r
        else:
            fan_in = shape[1] if len(shape) == 2 else None
            if fan_in is None:  # bias: fan_in of the matching weight
                wname = name[: -len("bias")] + "weight"
                fan_in = state_dict_spec(cfg)[wname][1]
            bound = 1.0 / math.sqrt(fan_in)
            out[name] = (torch.rand(shape, generator=g) * 2.0 - 1.0) * bound
    return out


def state_dict_checksum(sd: Dict[str, Tensor]) -> float:
    """Order-independent fp64 checksum used to pin regenerated weights."""
    tot = 0.0
    for k in sorted(sd):
        v = sd[k].double()
        tot += float((v * torch.arange(1, v.numel() + 1, dtype=torch.float64).reshape(v.shape).remainder(7.0).add(1.0)).sum())
    return tot


# ----------------------------------------------------------------------------
# precision policy
# ----------------------------------------------------------------------------
class Precision:
    def __init__(self, mode: str = "fp32"):
        assert mode in ("fp32", "bf16")
        self.mode = mode

    def rb(self, x: Tensor) -> Tensor:
        """Round to the autocast dtype and return as fp32 (identity in fp32 mode)."""
        if self.mode == "fp32":
            return x
        return x.to(torch.bfloat16).to(torch.float32)

    def linear(self, x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
        """nn.Linear as autocast runs it: operands cast to bf16, fp32 accumulate,
        result (incl. bias) rounded to bf16."""
        if self.mode == "fp32":
            return F.linear(x, w, b)
        y = F.linear(self.rb(x), self.rb(w), None if b is None else self.rb(b))
        return self.rb(y)


# ----------------------------------------------------------------------------
# L0: the diffusers ops (semantics: SURVEY.md App. A.7)
# ----------------------------------------------------------------------------
def timestep_sinusoid(t: Tensor, dim: int) -> Tensor:
    """diffusers Timesteps(num_channels=dim, flip_sin_to_cos=False,
    downscale_freq_shift=0) as constructed at temporal_denoiser.py:57-61:
    [sin(t f_i), cos(t f_i)], f_i = exp(-ln(1e4) i / half)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)


def fp32_layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """diffusers FP32LayerNorm (block.py:64,83,98,107) / nn.LayerNorm
    (temporal_denoiser.py:107): fp32 statistics, biased variance."""
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps)


def rms_norm(x: Tensor, w: Tensor, eps: float = 1e-6) -> Tensor:
    """diffusers RMSNorm(dim_head, eps=1e-6) created by Attention(qk_norm="rms_norm")
    (block.py:49,72,91): x * rsqrt(mean(x^2)+eps) * w in fp32."""
    var = x.float().pow(2).mean(-1, keepdim=True)
    return x.float() * torch.rsqrt(var + eps) * w.float()


# ----------------------------------------------------------------------------
# L1: RoPE (rotary_embedding.py)
# ----------------------------------------------------------------------------
def rope_tables(framestep: Tensor, head_dim: int) -> Tuple[Tensor, Tensor]:
    """precompute_freqs_rot (temporal_denoiser.py:114-149) without the (N+1)
    broadcast: positions = framestep - min_t framestep (embeddings.py:135-153,
    center=True, scale=False), merged (b t); compute_rotary_embeddings
    (rotary_embedding.py:10-69) -> cos, sin of shape (B*T, head_dim) with each
    frequency repeated twice (interleaved pairs)."""
    fs = framestep.float()
    pos = (fs - fs.min(dim=1).values[:, None]).reshape(-1)
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    phases = torch.outer(pos, inv_freq)
    cos = phases.cos().repeat_interleave(2, dim=1).float()
    sin = phases.sin().repeat_interleave(2, dim=1).float()
    return cos, sin


def apply_rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """apply_rotary_embedding (rotary_embedding.py:72-124) for x (B,H,S,D) and
    cos/sin (B,S,D): out[2i] = x[2i] c - x[2i+1] s ; out[2i+1] = x[2i+1] c + x[2i] s."""
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return x.float() * cos[:, None] + rot.float() * sin[:, None]


# ----------------------------------------------------------------------------
# L1: attention (attention_processor.py:36-168)
# ----------------------------------------------------------------------------
def _sdpa(q: Tensor, k: Tensor, v: Tensor, P: Precision) -> Tensor:
    """F.scaled_dot_product_attention(q,k,v) non-causal, scale 1/sqrt(hd)
    (attention_processor.py:133-139).  Under autocast q,k,v enter as bf16 and
    the result is bf16."""
    q, k, v = P.rb(q), P.rb(k), P.rb(v)
    return P.rb(F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False))


def self_attention(z: Tensor, sd: Dict[str, Tensor], pfx: str, H: int, T: int,
                   inflate: bool, cos: Tensor, sin: Tensor, P: Precision) -> Tensor:
    """Self branch of AttentionProcessor.__call__.  z: (B*T, L, C) normalised
    hidden states; cos/sin: (B*T, hd)."""
    BT, L, C = z.shape
    hd = C // H
    cos_t = cos[:, None, :].expand(BT, L, hd)
    sin_t = sin[:, None, :].expand(BT, L, hd)
    if inflate:  # attention_processor.py:49-65, tensor_ops.py:89-100
        z = z.reshape(BT // T, T * L, C)
        cos_t = cos_t.reshape(BT // T, T * L, hd)
        sin_t = sin_t.reshape(BT // T, T * L, hd)
    Bq, S, _ = z.shape
    q = P.linear(z, sd[pfx + "to_q.weight"], None)          # :92
    k = P.linear(z, sd[pfx + "to_k.weight"], None)          # :102
    v = P.linear(z, sd[pfx + "to_v.weight"], None)          # :103
    # :106-110  cat -> view (.., H, 3*hd) -> split: head h takes columns
    # [3hd*h, 3hd*h+hd) etc. of the *concatenated* q|k|v row
    qkv = torch.cat((q, k, v), dim=-1).view(Bq, S, H, 3 * hd)
    q, k, v = torch.split(qkv, hd, dim=-1)
    q, k, v = (t.transpose(1, 2) for t in (q, k, v))        # :117-119 (B,H,S,hd)
    q = rms_norm(q, sd[pfx + "norm_q.weight"])              # :121-124
    k = rms_norm(k, sd[pfx + "norm_k.weight"])
    q = apply_rope(q, cos_t, sin_t)                         # :127-130
    k = apply_rope(k, cos_t, sin_t)
    o = _sdpa(q, k, v, P)                                   # :133-139
    o = o.transpose(1, 2).reshape(Bq, S, C)                 # :141-144
    o = P.linear(o, sd[pfx + "to_out.0.weight"], sd[pfx + "to_out.0.bias"])  # :147
    if inflate:                                             # :161-166
        o = o.reshape(BT, L, C)
    return o


def cross_attention(z: Tensor, ctx: Tensor, sd: Dict[str, Tensor], pfx: str, H: int,
                    P: Precision) -> Tensor:
    """Cross branch (attention_processor.py:94-115): z (B*T,L,C), ctx (B*T,S,Dc);
    kv = cat(k, v) viewed (.., H, 2*hd); qk RMSNorm, no RoPE."""
    BT, L, C = z.shape
    hd = C // H
    S = ctx.shape[1]
    q = P.linear(z, sd[pfx + "to_q.weight"], None)
    k = P.linear(ctx, sd[pfx + "to_k.weight"], None)
    v = P.linear(ctx, sd[pfx + "to_v.weight"], None)
    kv = torch.cat((k, v), dim=-1).view(BT, S, H, 2 * hd)   # :111-115
    k, v = torch.split(kv, hd, dim=-1)
    q = q.view(BT, L, H, hd).transpose(1, 2)
    k, v = k.transpose(1, 2), v.transpose(1, 2)
    q = rms_norm(q, sd[pfx + "norm_q.weight"])
    k = rms_norm(k, sd[pfx + "norm_k.weight"])
    o = _sdpa(q, k, v, P)
    o = o.transpose(1, 2).reshape(BT, L, C)
    return P.linear(o, sd[pfx + "to_out.0.weight"], sd[pfx + "to_out.0.bias"])


# ----------------------------------------------------------------------------
# L1: block (block.py:110-154)
# ----------------------------------------------------------------------------
def block_forward(h: Tensor, ctx: Tensor, sd: Dict[str, Tensor], i: int, cfg: OracleConfig,
                  T: int, cos: Tensor, sin: Tensor, skip: Optional[Tensor], P: Precision) -> Tensor:
    p = f"blocks.{i}."
    H = cfg.num_attention_heads
    if cfg.has_skip(i):                                     # block.py:131-133
        cat = torch.cat([skip, h], dim=-1)
        h = P.linear(cat, sd[p + "linear_skip.weight"], sd[p + "linear_skip.bias"])
        h = P.rb(fp32_layer_norm(h, sd[p + "norm_skip.weight"], sd[p + "norm_skip.bias"]))
    z = P.rb(fp32_layer_norm(h, sd[p + "norm_s_attn.weight"], sd[p + "norm_s_attn.bias"]))
    h = P.rb(h + self_attention(z, sd, p + "s_attn.", H, T, i in cfg.inflated_layers, cos, sin, P))  # :137-142
    z = P.rb(fp32_layer_norm(h, sd[p + "norm_x_attn.weight"], sd[p + "norm_x_attn.bias"]))
    h = P.rb(h + cross_attention(z, ctx, sd, p + "x_attn.", H, P))                                 # :146-149
    z = P.rb(fp32_layer_norm(h, sd[p + "norm_ff.weight"], sd[p + "norm_ff.bias"]))
    # diffusers FeedForward(activation_fn="gelu"): Linear -> exact-erf GELU -> Linear (block.py:99-104)
    u = P.linear(z, sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"])
    u = P.rb(F.gelu(u, approximate="none"))
    u = P.linear(u, sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"])
    return P.rb(h + u)                                      # :152


# ----------------------------------------------------------------------------
# L2: denoiser forward (temporal_denoiser.py:151-249)
# ----------------------------------------------------------------------------
def denoiser_forward(sd: Dict[str, Tensor], cfg: OracleConfig, hidden_states: Tensor,
                     context: Tensor, framestep: Tensor, diffusion_time: Tensor,
                     mask: Optional[Tensor] = None, precision: str = "fp32",
                     return_intermediates: bool = False):
    """hidden_states (B,T,N,Din), context (B,T,S,Dc), framestep (B,T),
    diffusion_time (B,), mask (B,T) -> velocity (B,T,N,Din)."""
    P = Precision(precision)
    B, T, N, Din = hidden_states.shape
    C = cfg.width
    cos, sin = rope_tables(framestep, cfg.head_dim)                     # :196-202
    h = P.linear(hidden_states.reshape(B * T, N, Din), sd["proj_in.weight"], sd["proj_in.bias"])  # :205-206
    # :209-212  NOTE: repeat(T) is b-fastest while the merged mask is t-fastest.
    t_bt = diffusion_time.repeat(T)
    if mask is not None:
        t_bt = t_bt * (1 - mask.reshape(B * T))
    e = P.rb(timestep_sinusoid(t_bt, C))                                # :213  .to(hidden dtype)
    e = P.linear(e, sd["time_proj.linear_1.weight"], sd["time_proj.linear_1.bias"])
    e = P.rb(F.gelu(e, approximate="none"))
    e = P.linear(e, sd["time_proj.linear_2.weight"], sd["time_proj.linear_2.bias"])  # :214
    h = torch.cat([e[:, None, :], h], dim=1)                            # :217  (BT, L, C)
    ctx = context.reshape(B * T, context.shape[2], context.shape[3])    # :221
    skips: List[Tensor] = []
    inter = {}
    for i in range(cfg.num_layers):                                     # :222-236
        skip = None if i <= cfg.num_layers // 2 else skips.pop()
        h = block_forward(h, ctx, sd, i, cfg, T, cos, sin, skip, P)
        if i < cfg.num_layers // 2:
            skips.append(h)
        if return_intermediates:
            inter[f"block{i}"] = h.clone()
    # :239-242  nn.LayerNorm (fp32 under autocast) -> drop time token -> proj_out
    h = fp32_layer_norm(h, sd["norm_out.weight"], sd["norm_out.bias"])
    h = h[:, -N:]
    v = P.linear(h, sd["proj_out.weight"], sd["proj_out.bias"])
    v = v.reshape(B, T, N, Din)                                         # :245-247
    return (v, inter) if return_intermediates else v


def denoiser_forward_rows(sd: Dict[str, Tensor], cfg: OracleConfig, hidden_states: Tensor, context: Tensor,
                          framestep: Tensor, diffusion_time: Tensor, mask: Optional[Tensor], rows: Tensor,
                          frame_chunk: int = 4) -> Tensor:
    """denoiser_forward (fp32) restricted to SELECTED latent tokens of a ONE-layer model - the same statements in the same order,
    evaluated only where the selected rows need them, so that a sequence of 524 352 tokens (BASELINE configs[4]: 64 frames x 8192
    tokens) is checkable on host cores in seconds instead of hours: in a one-layer model everything except the inflated
    self-attention is per token (temporal_denoiser.py:205-217, block.py:137-152, temporal_denoiser.py:239-247), and the
    self-attention of a row needs the K / V of every token of its sample (attention_processor.py:92-139) but the Q of that row only.
    rows: int64 (R, 3) of (b, t, n), n indexing the N latent tokens of frame t.  Returns the velocity (R, Din).
    Checked against denoiser_forward itself in tests/test_oracle_golden.py::test_forward_rows_is_the_full_forward_on_those_rows."""
    assert cfg.num_layers == 1 and not cfg.has_skip(0), "one layer, no skip: the only block whose inputs are per-token"
    P = Precision("fp32")
    B, T, N, Din = hidden_states.shape
    C, H, hd, L = cfg.width, cfg.num_attention_heads, cfg.head_dim, N + 1
    p = "blocks.0."
    inflate = 0 in cfg.inflated_layers
    cos, sin = rope_tables(framestep, hd)                                   # (B*T, hd)  temporal_denoiser.py:196-202
    t_bt = diffusion_time.repeat(T)                                         # :209-212
    if mask is not None:
        t_bt = t_bt * (1 - mask.reshape(B * T))
    e = timestep_sinusoid(t_bt, C)                                          # :213
    e = P.linear(e, sd["time_proj.linear_1.weight"], sd["time_proj.linear_1.bias"])
    e = F.gelu(e, approximate="none")
    e = P.linear(e, sd["time_proj.linear_2.weight"], sd["time_proj.linear_2.bias"])   # :214  (B*T, C)

    def frames_h0(b: int, t0: int, t1: int) -> Tensor:
        """rows of `h` (:217) for frames [t0, t1) of sample b: (t1 - t0, L, C)."""
        hh = P.linear(hidden_states[b, t0:t1], sd["proj_in.weight"], sd["proj_in.bias"])   # :205-206
        return torch.cat([e[b * T + t0:b * T + t1, None, :], hh], dim=1)

    def heads_of(z: Tensor, cs: Tensor, sn: Tensor):
        """q, k, v (rows, H, hd) of normalised tokens z (rows, C) with their frames' RoPE rows cs / sn (rows, hd):
        attention_processor.py:92-130 - the per-head split of the CONCATENATED projection, qk RMSNorm, RoPE."""
        q = P.linear(z, sd[p + "s_attn.to_q.weight"], None)
        k = P.linear(z, sd[p + "s_attn.to_k.weight"], None)
        v = P.linear(z, sd[p + "s_attn.to_v.weight"], None)
        q, k, v = torch.split(torch.cat((q, k, v), dim=-1).view(-1, H, 3 * hd), hd, dim=-1)
        q = rms_norm(q, sd[p + "s_attn.norm_q.weight"])
        k = rms_norm(k, sd[p + "s_attn.norm_k.weight"])
        # apply_rope takes (B, H, S, D) with cos / sin (B, S, D): one "batch", S = rows
        q = apply_rope(q.transpose(0, 1)[None], cs[None], sn[None])[0].transpose(0, 1)
        k = apply_rope(k.transpose(0, 1)[None], cs[None], sn[None])[0].transpose(0, 1)
        return q, k, v

    out = torch.empty((rows.shape[0], Din), dtype=torch.float32)
    for b in sorted(set(int(x) for x in rows[:, 0])):
        sel = (rows[:, 0] == b).nonzero()[:, 0]
        rt, rn = rows[sel, 1], rows[sel, 2]
        R = sel.numel()
        # ---- the selected rows' own hidden state (token index 1 + n of frame t) -------------------------------------------
        h = torch.stack([frames_h0(b, int(t), int(t) + 1)[0, 1 + int(n)] for t, n in zip(rt, rn)])           # (R, C)
        z = fp32_layer_norm(h, sd[p + "norm_s_attn.weight"], sd[p + "norm_s_attn.bias"])                      # block.py:138
        q, _, _ = heads_of(z, cos[b * T + rt], sin[b * T + rt])                                             # (R, H, hd)
        # ---- K / V of every token this row attends to: the whole sample when inflated, else its own frame ----------------
        o = torch.zeros((R, H, hd))
        if inflate:
            K = torch.empty((H, T * L, hd)); V = torch.empty((H, T * L, hd))
            for t0 in range(0, T, frame_chunk):
                t1 = min(T, t0 + frame_chunk)
                zz = fp32_layer_norm(frames_h0(b, t0, t1), sd[p + "norm_s_attn.weight"], sd[p + "norm_s_attn.bias"]).reshape(-1, C)
                cs = cos[b * T + t0:b * T + t1].repeat_interleave(L, dim=0)
                sn = sin[b * T + t0:b * T + t1].repeat_interleave(L, dim=0)
                _, k, v = heads_of(zz, cs, sn)
                K[:, t0 * L:t1 * L] = k.transpose(0, 1); V[:, t0 * L:t1 * L] = v.transpose(0, 1)
            s = torch.einsum("rhd,hkd->hrk", q, K) * hd ** -0.5                                               # :133-139
            o = torch.einsum("hrk,hkd->rhd", torch.softmax(s, dim=-1), V)
        else:
            for j in range(R):
                t = int(rt[j])
                zz = fp32_layer_norm(frames_h0(b, t, t + 1), sd[p + "norm_s_attn.weight"], sd[p + "norm_s_attn.bias"]).reshape(-1, C)
                _, k, v = heads_of(zz, cos[b * T + t].expand(L, hd), sin[b * T + t].expand(L, hd))
                s = torch.einsum("hd,khd->hk", q[j], k) * hd ** -0.5
                o[j] = torch.einsum("hk,khd->hd", torch.softmax(s, dim=-1), v)
        h = h + P.linear(o.reshape(R, C), sd[p + "s_attn.to_out.0.weight"], sd[p + "s_attn.to_out.0.bias"])   # :147, block.py:137
        z = fp32_layer_norm(h, sd[p + "norm_x_attn.weight"], sd[p + "norm_x_attn.bias"])
        h = h + cross_attention(z[:, None], context[b, rt], sd, p + "x_attn.", H, P)[:, 0]                    # block.py:146-149
        z = fp32_layer_norm(h, sd[p + "norm_ff.weight"], sd[p + "norm_ff.bias"])
        u = F.gelu(P.linear(z, sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"]), approximate="none")
        h = h + P.linear(u, sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"])                               # block.py:152
        h = fp32_layer_norm(h, sd["norm_out.weight"], sd["norm_out.bias"])                                    # :239-242
        out[sel] = P.linear(h, sd["proj_out.weight"], sd["proj_out.bias"])
    return out


# ----------------------------------------------------------------------------
# L3: sampler (scheduler.py, guidance.py)
# ----------------------------------------------------------------------------
def compute_timesteps(num_inference_steps: int, num_train_timesteps: int = 1000,
                      shift: float = 1.0) -> np.ndarray:
    """SchedulerFlow._compute_timesteps (scheduler.py:59-98): numpy fp64 math,
    result cast to fp32."""
    full = np.linspace(1, num_train_timesteps, num_train_timesteps) / num_train_timesteps
    full = full[::-1]
    full_shift = shift * full / (1 + (shift - 1) * full)
    smax, smin = full_shift[0], full_shift[-1]
    ts = np.linspace(smax * num_train_timesteps, smin * num_train_timesteps, num_inference_steps)
    sig = ts / num_train_timesteps
    sig = shift * sig / (1 + (shift - 1) * sig)
    return (sig * num_train_timesteps).astype(np.float32)


def get_schedule(num_inference_steps: int, num_train_timesteps: int = 1000,
                 shift: float = 3.0) -> Tuple[Tensor, Tensor]:
    """SchedulerFlow.get_schedule (scheduler.py:43-56)."""
    t = torch.from_numpy(compute_timesteps(num_inference_steps + 1, num_train_timesteps, shift))
    d = (t[:-1] - t[1:]) / num_train_timesteps
    return t, d


def get_noise(latent_shape, batch_size: int, n_timesteps: int, generator=None,
              corr_noise: float = 0.0) -> Tensor:
    """SchedulerFlow.get_noise (scheduler.py:100-137): TWO draws in fixed order."""
    same = torch.randn([batch_size, 1] + list(latent_shape), generator=generator).repeat(1, n_timesteps, 1, 1)
    indep = torch.randn([batch_size, n_timesteps] + list(latent_shape), generator=generator)
    return math.sqrt(corr_noise) * same + math.sqrt(1 - corr_noise) * indep


def cfg_at_inference(latent, context, mask, framestep, guidance_at_inference):
    """ClassifierFreeGuidance.cfg_at_inference (guidance.py:38-93)."""
    n = len(guidance_at_inference)
    latent = torch.cat([latent] * n)
    framestep = torch.cat([framestep] * n) if framestep is not None else None
    ctxs, masks = [], []
    for g in guidance_at_inference:
        img, lat = g
        ctxs.append(context if img == 1 else torch.zeros_like(context))
        if mask is not None:
            masks.append(mask if lat == 1 else torch.zeros_like(mask))
    return latent, torch.cat(ctxs, 0), (torch.cat(masks, 0) if mask is not None else None), framestep


def aggregate_cfg(v: Tensor, n_branches: int, scales, P: Precision) -> Tensor:
    """ClassifierFreeGuidance.aggregate_cfg (guidance.py:95-118):
    out = v0 + sum_i s_i (v_{i+1} - v_i), evaluated in the velocity dtype."""
    chunks = v.chunk(n_branches, dim=0)
    out = chunks[0]
    for i in range(n_branches - 1):
        out = P.rb(out + P.rb(scales[i] * P.rb(chunks[i + 1] - chunks[i])))
    return out


def flow_sample(sd: Dict[str, Tensor], cfg: OracleConfig, init_latent: Tensor, context: Tensor,
                mask: Optional[Tensor], framestep: Tensor, num_inference_steps: int,
                guidance_at_inference=((0, 1), (1, 1)), guidance_scales=(7.5,),
                shift: float = 3.0, is_additive: bool = True, precision: str = "fp32",
                step_callback: Optional[Callable[[int, int], None]] = None) -> List[Tensor]:
    """SchedulerFlow._flow_sample / denoise (scheduler.py:172-295).  Returns the
    list of per-step latents (clones; the reference yields the same tensor)."""
    P = Precision(precision)
    latents = init_latent.clone()
    timesteps, distances = get_schedule(num_inference_steps, shift=shift)
    unobserved = None if mask is None else (mask == 0)          # guidance.py:28-36
    gai = [list(g) for g in guidance_at_inference]
    out: List[Tensor] = []
    for i, t in enumerate(timesteps[:-1]):
        x_in, c_in, m_in, f_in = cfg_at_inference(latents, context, mask, framestep, gai)   # :215-217
        dt = torch.tensor([float(t)], dtype=latents.dtype).expand(x_in.shape[0])              # :219-221
        v = denoiser_forward(sd, cfg, x_in, c_in, f_in, dt, m_in, precision)                # :224-232
        v = aggregate_cfg(v, len(gai), list(guidance_scales), P)                             # :235
        step = P.rb(distances[i] * v)                  # 0-dim fp32 * bf16 tensor -> bf16 (App. C)
        flow = latents + step if is_additive else latents - step                            # :238-241
        if unobserved is not None:                                                           # :244-248
            assert bool(unobserved.any()), "No unobserved frames found"
            latents[unobserved] = flow[unobserved]
        else:
            latents = flow
        out.append(latents.clone())
        if step_callback is not None:
            step_callback(i + 1, num_inference_steps)
    return out


# ----------------------------------------------------------------------------
# algorithmic work - used by bench.py's cpu_baseline leg
# ----------------------------------------------------------------------------
def step_flops(B: int, T: int, N: int, cfg: OracleConfig, S: int) -> float:
    C, F_, Dc, Din, NL = cfg.width, cfg.ff_inner, cfg.cross_attention_dim, cfg.in_channels, cfg.num_layers
    TL = T * (N + 1)
    tot = 0.0
    for i in range(NL):
        attn = 4.0 * TL * TL * C if i in cfg.inflated_layers else 4.0 * T * (N + 1) ** 2 * C
        per = 6.0 * TL * C * C + attn + 2.0 * TL * C * C
        per += 2.0 * TL * C * C + 4.0 * T * S * Dc * C + 4.0 * TL * S * C + 2.0 * TL * C * C
        per += 4.0 * TL * C * F_
        if cfg.has_skip(i):
            per += 4.0 * TL * C * C
        tot += per
    tot += 4.0 * T * N * Din * C + 16.0 * T * C * C
    return B * tot
