"""CPU restatement of the context encoder the reference runs before Stage I (TEST INFRASTRUCTURE ONLY - imported by
tests/ and tools/ checkers only; never on the product path).

The reference's ImageEncoder (actionmesh/model/image_encoder.py:16-55) is a thin wrapper:
    pixel_values = BitImageProcessor.preprocess(images)            (:48-51, CPU / PIL glue, not restated)
    context      = Dinov2Model(pixel_values).last_hidden_state      (:53-55)  -> (T, S, Dc) fp32
called once per video OUTSIDE the cuda autocast region (pipeline.py:657-667), i.e. in fp32.
The arithmetic lives in a third-party dependency, `transformers` (requirements.txt:10 `transformers<5`, no exact pin;
this container ships 5.15.0): models/dinov2/modeling_dinov2.py.  Its published algorithm, restated here:
    Dinov2PatchEmbeddings   Conv2d(3, C, kernel = stride = patch)  -> (T, n_h * n_w, C), row-major over the patch grid
    Dinov2Embeddings        [cls | patches] + position embeddings; when the patch grid differs from the trained one
                            the patch part of the table is resampled: bicubic, align_corners=False, in fp32
    Dinov2Layer (x NL)      h += ls1 * Wo . MHA(LN(h));  h += ls2 * fc2(GELU_erf(fc1(LN(h))));  LN eps = layer_norm_eps
    Dinov2Model             final LayerNorm -> last_hidden_state
Pinned against `transformers.Dinov2Model` itself (same weights, same pixels) by tests/golden/dinov2_tiny.npz, written
by oracle/make_golden_dinov2.py.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass(frozen=True)
class DinoConfig:
    """The fields of transformers' Dinov2Config this path reads (defaults = DINOv2 ViT-L/14 as TripoSG ships it)."""
    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    mlp_ratio: int = 4
    patch_size: int = 14
    image_size: int = 518            # size the position table was trained at (37 x 37 patches)
    num_channels: int = 3
    layer_norm_eps: float = 1e-6
    qkv_bias: bool = True

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def num_positions(self) -> int:
        return (self.image_size // self.patch_size) ** 2


def state_dict_spec(cfg: DinoConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) of every Dinov2Model parameter, in the module's state-dict order."""
    C, Fi, p = cfg.hidden_size, cfg.hidden_size * cfg.mlp_ratio, cfg.patch_size
    spec: List[Tuple[str, Tuple[int, ...]]] = [
        ("embeddings.cls_token", (1, 1, C)), ("embeddings.mask_token", (1, C)),
        ("embeddings.position_embeddings", (1, cfg.num_positions + 1, C)),
        ("embeddings.patch_embeddings.projection.weight", (C, cfg.num_channels, p, p)),
        ("embeddings.patch_embeddings.projection.bias", (C,))]
    for i in range(cfg.num_hidden_layers):
        q = f"encoder.layer.{i}."
        spec += [(q + "norm1.weight", (C,)), (q + "norm1.bias", (C,))]
        for n in ("query", "key", "value"):
            spec += [(q + f"attention.attention.{n}.weight", (C, C)), (q + f"attention.attention.{n}.bias", (C,))]
        spec += [(q + "attention.output.dense.weight", (C, C)), (q + "attention.output.dense.bias", (C,)),
                 (q + "layer_scale1.lambda1", (C,)),
                 (q + "norm2.weight", (C,)), (q + "norm2.bias", (C,)),
                 (q + "mlp.fc1.weight", (Fi, C)), (q + "mlp.fc1.bias", (Fi,)),
                 (q + "mlp.fc2.weight", (C, Fi)), (q + "mlp.fc2.bias", (C,)),
                 (q + "layer_scale2.lambda1", (C,))]
    spec += [("layernorm.weight", (C,)), ("layernorm.bias", (C,))]
    return spec


def synthetic_state_dict(cfg: DinoConfig, seed: int = 0) -> Dict[str, Tensor]:
    """Deterministic stand-in weights (no pretrained weights offline): every tensor from its own crc32-seeded
    generator; linear / conv weights ~ N(0, 1/fan_in), norm gains and layer scales around 1, biases and tables small."""
    sd: Dict[str, Tensor] = {}
    for name, shape in state_dict_spec(cfg):
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        if name.endswith(".weight") and len(shape) >= 2:
            t = torch.randn(shape, generator=g) / math.sqrt(math.prod(shape[1:]))
        elif name.endswith(".weight") or name.endswith("lambda1"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif "embeddings." in name and not name.endswith(".bias"):
            t = 0.5 * torch.randn(shape, generator=g)
        else:
            t = 0.05 * torch.randn(shape, generator=g)
        sd[name] = t
    return sd


def state_dict_checksum(sd: Dict[str, Tensor]) -> float:
    return float(sum(float(v.double().abs().sum()) * (1 + (zlib.crc32(k.encode()) % 97) / 97.0) for k, v in sd.items()))


def position_table(sd: Dict[str, Tensor], cfg: DinoConfig, height: int, width: int) -> Tensor:
    """Dinov2Embeddings.interpolate_pos_encoding: (1 + n_h * n_w, C); row 0 = the class position."""
    pos = sd["embeddings.position_embeddings"][0].float()
    n_h, n_w = height // cfg.patch_size, width // cfg.patch_size
    if n_h * n_w == cfg.num_positions and height == width:
        return pos
    side = int(cfg.num_positions ** 0.5)
    grid = pos[1:].reshape(1, side, side, -1).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(n_h, n_w), mode="bicubic", align_corners=False)
    return torch.cat([pos[:1], grid.permute(0, 2, 3, 1).reshape(n_h * n_w, -1)], dim=0)


def patch_rows(pixel_values: Tensor, patch: int) -> Tensor:
    """(T, Cin, H, W) -> (T, n_h * n_w, Cin * patch * patch): the im2col of a stride = kernel convolution, columns in
    the (channel, ky, kx) order of a flattened Conv2d weight."""
    T, Cin, H, W = pixel_values.shape
    n_h, n_w = H // patch, W // patch
    x = pixel_values[:, :, : n_h * patch, : n_w * patch].reshape(T, Cin, n_h, patch, n_w, patch)
    return x.permute(0, 2, 4, 1, 3, 5).reshape(T, n_h * n_w, Cin * patch * patch)


def embeddings(sd: Dict[str, Tensor], cfg: DinoConfig, pixel_values: Tensor) -> Tensor:
    T, _, H, W = pixel_values.shape
    w = sd["embeddings.patch_embeddings.projection.weight"].float()
    x = patch_rows(pixel_values.float(), cfg.patch_size) @ w.reshape(w.shape[0], -1).T \
        + sd["embeddings.patch_embeddings.projection.bias"].float()
    cls = sd["embeddings.cls_token"].float().expand(T, -1, -1)
    return torch.cat([cls, x], dim=1) + position_table(sd, cfg, H, W)[None]


def layer(sd: Dict[str, Tensor], cfg: DinoConfig, i: int, h: Tensor) -> Tensor:
    q, C, H, hd = f"encoder.layer.{i}.", cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim
    lin = lambda x, n: F.linear(x, sd[q + n + ".weight"].float(), sd[q + n + ".bias"].float() if q + n + ".bias" in sd else None)
    T, S, _ = h.shape
    z = F.layer_norm(h, (C,), sd[q + "norm1.weight"].float(), sd[q + "norm1.bias"].float(), cfg.layer_norm_eps)
    heads = lambda x: x.view(T, S, H, hd).transpose(1, 2)
    qh, kh, vh = (heads(lin(z, f"attention.attention.{n}")) for n in ("query", "key", "value"))
    p = torch.softmax(qh @ kh.transpose(2, 3) * hd ** -0.5, dim=-1)
    a = (p @ vh).transpose(1, 2).reshape(T, S, C)
    h = lin(a, "attention.output.dense") * sd[q + "layer_scale1.lambda1"].float() + h
    z = F.layer_norm(h, (C,), sd[q + "norm2.weight"].float(), sd[q + "norm2.bias"].float(), cfg.layer_norm_eps)
    m = lin(F.gelu(lin(z, "mlp.fc1")), "mlp.fc2")
    return m * sd[q + "layer_scale2.lambda1"].float() + h


@torch.no_grad()
def dinov2_forward(sd: Dict[str, Tensor], cfg: DinoConfig, pixel_values: Tensor) -> Tensor:
    """pixel_values (T, 3, H, W) fp32 -> last_hidden_state (T, 1 + (H/p)(W/p), C) fp32."""
    h = embeddings(sd, cfg, pixel_values)
    for i in range(cfg.num_hidden_layers):
        h = layer(sd, cfg, i, h)
    return F.layer_norm(h, (cfg.hidden_size,), sd["layernorm.weight"].float(), sd["layernorm.bias"].float(),
                        cfg.layer_norm_eps)
