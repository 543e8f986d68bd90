"""Context encoder: the DINOv2 oracle against transformers' own output (golden fixture), the host-side
weight packing on CPU, and the HIP path against the oracle on the GPU."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from actionmesh_amd import image_encoder as IE
from actionmesh_amd import ops
from oracle import dinov2_oracle as DO

GOLD = os.path.join(os.path.dirname(__file__), "golden", "dinov2_tiny.npz")


def _case(name):
    g = np.load(GOLD)
    C, NL, H, side = (int(v) for v in g[name + ".cfg"])
    cfg = DO.DinoConfig(hidden_size=C, num_hidden_layers=NL, num_attention_heads=H, image_size=side)
    sd = DO.synthetic_state_dict(cfg, seed=0)
    assert DO.state_dict_checksum(sd) == pytest.approx(float(g[name + ".checksum"]), rel=1e-12)
    return cfg, sd, torch.from_numpy(g[name + ".pixels"]), torch.from_numpy(g[name + ".last_hidden_state"])


def _cfg_dict(cfg: DO.DinoConfig):
    return dict(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                mlp_ratio=cfg.mlp_ratio, patch_size=cfg.patch_size, image_size=cfg.image_size, layer_norm_eps=cfg.layer_norm_eps)


@pytest.mark.parametrize("name", ["native", "resampled"])
def test_oracle_matches_transformers_golden(name):
    """The fixture holds transformers.Dinov2Model's last_hidden_state for these weights and pixels (fp32 CPU; its SDPA
    kernel and the explicit softmax of the restatement differ in rounding only)."""
    cfg, sd, pixels, ref = _case(name)
    out = DO.dinov2_forward(sd, cfg, pixels)
    assert out.shape == ref.shape
    assert float((out - ref).abs().max()) <= 2e-5


def test_patch_rows_is_the_strided_convolution():
    g = torch.Generator().manual_seed(1)
    x = torch.randn((2, 3, 28, 42), generator=g)
    w = torch.randn((5, 3, 14, 14), generator=g)
    conv = F.conv2d(x, w, stride=14).flatten(2).transpose(1, 2)
    mine = DO.patch_rows(x, 14) @ w.reshape(5, -1).T
    assert torch.allclose(conv, mine, atol=1e-4)


def _packed_forward_fp32(w, cfg, pixels):
    """The op sequence of HipImageEncoder.encode_pixels in plain fp32 torch on the PACKED operands: checks the packing
    (head padding, head-major [q|k|v] order, folded LayerScale, padded patch projection, position rows), not kernels."""
    C, H, p, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["patch_size"], cfg["layer_norm_eps"]
    T, _, Hi, Wi = pixels.shape
    n_h, n_w = Hi // p, Wi // p
    S = n_h * n_w + 1
    rows = IE.position_rows(w["pos"], w["cls"], cfg["image_size"] // p, n_h, n_w)
    A = DO.patch_rows(pixels, p)
    A = F.pad(A, (0, w["patch.w"].shape[1] - A.shape[-1]))
    h = torch.cat([rows[None, :1].expand(T, -1, -1), A @ w["patch.w"].T + w["patch.b"] + rows[None, 1:]], dim=1)
    for i in range(cfg["num_hidden_layers"]):
        q = f"l{i}."
        z = F.layer_norm(h, (C,), w[q + "norm1.w"], w[q + "norm1.b"], eps)
        qkv = (z @ w[q + "qkv.w"].T + w[q + "qkv.b"]).view(T, S, H, 3, ops.HEAD_DIM)
        Q, K, V = (qkv[:, :, :, j].transpose(1, 2) for j in range(3))
        a = torch.softmax(Q @ K.transpose(2, 3) * (C // H) ** -0.5, -1) @ V
        h = a.transpose(1, 2).reshape(T, S, H * ops.HEAD_DIM) @ w[q + "o.w"].T + w[q + "o.b"] + h
        z = F.layer_norm(h, (C,), w[q + "norm2.w"], w[q + "norm2.b"], eps)
        h = F.gelu(z @ w[q + "fc1.w"].T + w[q + "fc1.b"]) @ w[q + "fc2.w"].T + w[q + "fc2.b"] + h
    return F.layer_norm(h, (C,), w["norm.w"], w["norm.b"], eps)


@pytest.mark.parametrize("name", ["native", "resampled"])
def test_weight_packing_reproduces_the_oracle(name):
    cfg, sd, pixels, ref = _case(name)
    c = dict(IE._CFG_DEFAULTS, **_cfg_dict(cfg))
    out = _packed_forward_fp32(IE.pack_weights(sd, c), c, pixels)
    assert float((out - ref).abs().max()) <= 5e-5


def test_flop_count_of_the_shipped_shape():
    enc = IE.HipImageEncoder.__new__(IE.HipImageEncoder)
    enc.cfg = dict(IE._CFG_DEFAULTS)
    fl = enc.step_flops(16, 224, 224)
    S, C = 257, 1024
    assert fl == pytest.approx(16 * (24 * (8 * S * C * C + 4 * S * S * C + 16 * S * C * C) + 2 * 256 * 588 * C))
    assert 2.4e12 < fl < 2.6e12


# ---- GPU: the HIP path --------------------------------------------------------------------------------------------
def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.gpu
def test_patchify_bit_exact():
    g = torch.Generator().manual_seed(3)
    x = torch.randn((3, 3, 56, 70), generator=g)
    got = ops.patchify(x.cuda(), 14, 640).cpu()
    want = DO.patch_rows(x, 14).reshape(-1, 588).to(torch.bfloat16)
    assert got.shape == (3 * 4 * 5, 640)
    assert torch.equal(got[:, :588], want) and float(got[:, 588:].float().abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["native", "resampled"])
def test_hip_encoder_matches_golden(name):
    """Tolerance 2e-2 rel-L2 / 8e-2 max abs on unit-variance outputs (final LayerNorm): bf16 storage of the residual
    stream and operands against the reference's fp32 evaluation."""
    cfg, sd, pixels, ref = _case(name)
    enc = IE.HipImageEncoder(config=_cfg_dict(cfg), state_dict=sd).to("cuda:0")
    out = enc.encode_pixels(pixels.cuda()).cpu()
    assert out.shape == ref.shape and out.dtype == torch.float32
    assert _rel(out, ref) <= 2e-2, _rel(out, ref)
    assert float((out - ref).abs().max()) <= 8e-2


@pytest.mark.gpu
def test_hip_encoder_shipped_width_against_oracle():
    """DINOv2-L geometry (width 1024, 16 heads of 64, 224 x 224 crops, table trained at 518 -> resampled 37 -> 16), 4 of
    the 24 layers and 2 frames so the CPU oracle finishes in seconds."""
    cfg = DO.DinoConfig(num_hidden_layers=4)
    sd = DO.synthetic_state_dict(cfg, seed=1)
    g = torch.Generator().manual_seed(5)
    pixels = torch.randn((2, 3, 224, 224), generator=g)
    ref = DO.dinov2_forward(sd, cfg, pixels)
    enc = IE.HipImageEncoder(config=dict(num_hidden_layers=4), state_dict=sd).to("cuda:0")
    out = enc.encode_pixels(pixels.cuda())
    assert out.shape == (2, 257, 1024)
    assert _rel(out.cpu(), ref) <= 2e-2, _rel(out.cpu(), ref)
    ctx16 = enc.encode_pixels(pixels.cuda(), out_dtype=torch.bfloat16)
    assert ctx16.dtype == torch.bfloat16 and torch.equal(ctx16.float(), out)
    # frames are independent: encoding them one by one gives the same rows
    one = enc.encode_pixels(pixels[1:].cuda())
    assert _rel(one[0], out[1]) <= 2e-3


def test_load_state_dict_validates_keys_and_shapes():
    cfg, sd, _, _ = _case("native")
    with pytest.raises(KeyError, match="missing"):
        IE.HipImageEncoder(config=_cfg_dict(cfg), state_dict={k: v for k, v in sd.items() if "layer_scale2" not in k})
    with pytest.raises(ValueError, match="shape mismatch"):
        IE.HipImageEncoder(config=dict(_cfg_dict(cfg), image_size=70), state_dict=sd)
    assert set(IE.state_dict_shapes(dict(IE._CFG_DEFAULTS, **_cfg_dict(cfg)))) == set(sd) - {"embeddings.mask_token"}


def test_no_cpu_path():
    cfg, sd, pixels, _ = _case("native")
    enc = IE.HipImageEncoder(config=_cfg_dict(cfg), state_dict=sd)
    with pytest.raises(RuntimeError, match="no CPU path"):
        enc.encode_pixels(pixels)
