"""Generate tests/golden/dinov2_tiny.npz from `transformers.Dinov2Model` itself (the third-party module the reference's
ImageEncoder calls, actionmesh/model/image_encoder.py:25-27, 53-55).

TEST INFRASTRUCTURE ONLY.  Run where `transformers` is importable (it is in this image):

    python oracle/make_golden_dinov2.py

Weights = oracle.dinov2_oracle.synthetic_state_dict (no pretrained weights offline).  Two cases: the position table
used as trained, and resampled (bicubic) to a smaller patch grid - what the shipped model does for its 224 x 224 crops
(table trained at 518 x 518).  Stored: the pixels, the module's last_hidden_state (fp32, CPU) and a weight checksum.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from transformers import Dinov2Config, Dinov2Model  # noqa: E402  (third-party reference implementation)

from oracle import dinov2_oracle as DO  # noqa: E402

CASES = {
    # name: (config kwargs, T, image side fed)
    "native": (dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, image_size=56), 2, 56),
    "resampled": (dict(hidden_size=256, num_hidden_layers=3, num_attention_heads=4, image_size=98), 3, 56),
}
# round 5 (VERDICT r04 next #1c): the SHIPPED geometry - facebook/dinov2-large = ViT-L/14: width 1024, 24 layers, 16 heads of 64,
# position table trained at 518 x 518, fed the 224 x 224 crops of the reference's BitImageProcessor (image_encoder.py:48-51) -> 257
# tokens per frame; 2 frames.  `python oracle/make_golden_dinov2.py vitl` writes tests/golden/dinov2_vitl.npz (every 2nd token of
# the fp32 last_hidden_state, CLS included, plus transformers' own autocast(bf16) distance for scale).
VITL = ("vitl", dict(), 2, 224)
TOKEN_STRIDE = 2


def build(kw):
    cfg = DO.DinoConfig(**kw)
    sd = DO.synthetic_state_dict(cfg, seed=0)
    m = Dinov2Model(Dinov2Config(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                                 num_attention_heads=cfg.num_attention_heads, mlp_ratio=cfg.mlp_ratio,
                                 patch_size=cfg.patch_size, image_size=cfg.image_size,
                                 layer_norm_eps=cfg.layer_norm_eps, qkv_bias=cfg.qkv_bias))
    assert [k for k, _ in DO.state_dict_spec(cfg)] == list(m.state_dict().keys()), "state-dict order"
    m.load_state_dict(sd)
    m.eval()
    return cfg, sd, m


if sys.argv[1:] == ["vitl"]:
    name, kw, T, side = VITL
    cfg, sd, m = build(kw)
    g = torch.Generator().manual_seed(29)
    pixels = torch.randn((T, 3, side, side), generator=g) * 1.2
    with torch.no_grad():
        ref = m(pixels).last_hidden_state
        with torch.autocast("cpu", dtype=torch.bfloat16):
            ref16 = m(pixels).last_hidden_state.float()
    mine = DO.dinov2_forward(sd, cfg, pixels)
    err = float((ref - mine).abs().max())
    r16 = float((ref16.double() - ref.double()).norm() / ref.double().norm())
    print(f"vitl: out {tuple(ref.shape)}  max |transformers - oracle| = {err:.3e};  transformers autocast(bf16) vs its fp32 rel-L2 {r16:.3e}")
    assert ref.shape == (T, 257, 1024) and err <= 1e-4, err
    path = os.path.join(ROOT, "tests", "golden", "dinov2_vitl.npz")
    np.savez_compressed(path, seed=np.int64(29), frames=np.int64(T), side=np.int64(side), pixel_scale=np.float64(1.2),
                        pixels_checksum=np.float64(pixels.double().sum().item()), token_stride=np.int64(TOKEN_STRIDE),
                        last_hidden_state_sub=ref[:, ::TOKEN_STRIDE].numpy(), checksum=np.float64(DO.state_dict_checksum(sd)),
                        ref_autocast_bf16_rel=np.float64(r16),
                        cfg=np.array([cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.image_size]))
    print("wrote", path, os.path.getsize(path), "bytes")
    sys.exit(0)

out = {}
for name, (kw, T, side) in CASES.items():
    cfg = DO.DinoConfig(**kw)
    sd = DO.synthetic_state_dict(cfg, seed=0)
    m = Dinov2Model(Dinov2Config(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                                 num_attention_heads=cfg.num_attention_heads, mlp_ratio=cfg.mlp_ratio,
                                 patch_size=cfg.patch_size, image_size=cfg.image_size,
                                 layer_norm_eps=cfg.layer_norm_eps, qkv_bias=cfg.qkv_bias))
    assert [k for k, _ in DO.state_dict_spec(cfg)] == list(m.state_dict().keys()), "state-dict order"
    m.load_state_dict(sd)
    m.eval()
    g = torch.Generator().manual_seed(23)
    pixels = torch.randn((T, 3, side, side), generator=g) * 1.2
    with torch.no_grad():
        ref = m(pixels).last_hidden_state
    mine = DO.dinov2_forward(sd, cfg, pixels)
    err = float((ref - mine).abs().max())
    print(f"{name}: out {tuple(ref.shape)}  max |transformers - oracle| = {err:.3e}")
    assert err <= 2e-5, err
    out[name + ".pixels"] = pixels.numpy()
    out[name + ".last_hidden_state"] = ref.numpy()
    out[name + ".checksum"] = np.float64(DO.state_dict_checksum(sd))
    out[name + ".cfg"] = np.array([cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.image_size])

path = os.path.join(ROOT, "tests", "golden", "dinov2_tiny.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
