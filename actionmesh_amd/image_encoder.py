"""Context encoder on the Stage-I kernels: host-side mirror of the reference's `ImageEncoder`
(actionmesh/model/image_encoder.py:16-55) - same constructor fields, `.device`, `.eval()`, `.to()`,
`encode_images(images) -> context (T, S, Dc) fp32` - with the DINOv2 ViT (`transformers.Dinov2Model`, the third-party
module the reference calls at :53) evaluated through the C-ABI.

The reference is Python; so is this orchestration.  Every arithmetic step is a call through the C-ABI (`am_patchify`,
`am_gemm_bf16`, `am_layernorm_bf16`, `am_head_post`, `am_attention_bf16`); torch owns device memory and copies the T
class-token rows.  There is no CPU / torch fallback.  The image preprocessing (`BitImageProcessor`: resize, crop,
normalise PIL images; image_encoder.py:48-51) is CPU glue and stays on the reference's own dependency.

How the ViT maps onto kernels that were built for head_dim 128:
  * patch embedding: kernel = stride convolution = im2col (`am_patchify`, K = 3 p^2 padded to a multiple of 64) + one
    GEMM whose row map writes behind each frame's class token and whose residual operand is the position table;
  * the position table is resampled (bicubic, fp32, on the host - a 37 x 37 -> 16 x 16 table, once per image size) exactly
    as Dinov2Embeddings.interpolate_pos_encoding does on every call;
  * heads of 64 channels are zero-padded to 128 in the packed QKV weight (the padded q/k channels add 0 to every
    score, the padded v channels produce zero output columns that meet zero columns of the packed out-projection), so
    `am_head_post` and the attention kernels apply unchanged with scale = 64^-1/2.  This doubles the cost of three small
    GEMM operands of a component that runs once per video (~0.4 % of one 50-step window) - not worth a second kernel;
  * LayerScale is folded into the out-projection / fc2 weights and biases at load time; the residual add is the GEMM
    epilogue's.
Precision: 16-bit MFMA operands / fp32 accumulation, and - round 6, `residual_fp32=True`, the default - an **fp32 residual stream**:
the reference runs this model in fp32, outside its autocast region (pipeline.py:665-667), so what the 16-bit path loses is the rounding
of the GEMM operands / outputs, not 24 layers of re-rounding the stream (`am_add_layernorm_f32` adds each branch's 16-bit output into
the fp32 stream and emits the next LayerNorm's 16-bit output; `residual_fp32=False` is the round-5 all-16-bit stream).  Tolerances
stated in tests/test_image_encoder.py.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from . import ops
from ._lib import lib

_CFG_DEFAULTS = dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, mlp_ratio=4, patch_size=14,
                     image_size=518, num_channels=3, layer_norm_eps=1e-6, qkv_bias=True, use_swiglu_ffn=False)


def state_dict_shapes(cfg: Dict) -> Dict[str, Tuple[int, ...]]:
    """Parameter shapes of the Dinov2Model this encoder evaluates (load-time validation; tools/encoder_bench.py)."""
    C, Fi, p = cfg["hidden_size"], cfg["hidden_size"] * cfg["mlp_ratio"], cfg["patch_size"]
    shapes = {"embeddings.cls_token": (1, 1, C), "embeddings.position_embeddings": (1, (cfg["image_size"] // p) ** 2 + 1, C),
              "embeddings.patch_embeddings.projection.weight": (C, cfg["num_channels"], p, p),
              "embeddings.patch_embeddings.projection.bias": (C,), "layernorm.weight": (C,), "layernorm.bias": (C,)}
    for i in range(cfg["num_hidden_layers"]):
        q = f"encoder.layer.{i}."
        for n in ("query", "key", "value"):
            shapes[q + f"attention.attention.{n}.weight"] = (C, C)
            if cfg["qkv_bias"]:
                shapes[q + f"attention.attention.{n}.bias"] = (C,)
        shapes.update({q + "attention.output.dense.weight": (C, C), q + "attention.output.dense.bias": (C,),
                       q + "mlp.fc1.weight": (Fi, C), q + "mlp.fc1.bias": (Fi,), q + "mlp.fc2.weight": (C, Fi),
                       q + "mlp.fc2.bias": (C,)})
        for n in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "layer_scale1.lambda1", "layer_scale2.lambda1"):
            shapes[q + n] = (C,)
    return shapes


def pack_weights(sd: Dict[str, torch.Tensor], cfg: Dict) -> Dict[str, torch.Tensor]:
    """Dinov2Model state dict (fp32, CPU) -> the operands the kernels take (still fp32, CPU; `_upload` casts):
    head-padded [q | k | v]-per-head projection, LayerScale folded into the out-projection and fc2, the flattened patch
    projection padded to a multiple of 64 columns."""
    C, H = cfg["hidden_size"], cfg["num_attention_heads"]
    hd, HP = C // H, ops.HEAD_DIM
    f = lambda k: sd[k].detach().to("cpu", torch.float32)
    w: Dict[str, torch.Tensor] = {}
    wp = f("embeddings.patch_embeddings.projection.weight").reshape(C, -1)
    kp = ops.round_up(wp.shape[1], 64)
    w["patch.w"] = torch.zeros((C, kp)); w["patch.w"][:, : wp.shape[1]] = wp
    w["patch.b"] = f("embeddings.patch_embeddings.projection.bias")
    w["cls"] = f("embeddings.cls_token").reshape(C)
    w["pos"] = f("embeddings.position_embeddings")[0]
    for i in range(cfg["num_hidden_layers"]):
        q, p = f"encoder.layer.{i}.", f"l{i}."
        # am_head_post reads the GEMM output as (head, part, 128) - the layout of the reference denoiser's fused
        # projection - so the rows of the packed weight are ordered head-major with each 64-row block padded to 128
        wq = torch.zeros((H, 3, HP, C)); bq = torch.zeros((H, 3, HP))
        for j, n in enumerate(("query", "key", "value")):
            wq[:, j, :hd] = f(q + f"attention.attention.{n}.weight").reshape(H, hd, C)
            if q + f"attention.attention.{n}.bias" in sd:
                bq[:, j, :hd] = f(q + f"attention.attention.{n}.bias").reshape(H, hd)
        w[p + "qkv.w"], w[p + "qkv.b"] = wq.reshape(H * 3 * HP, C), bq.reshape(-1)
        ls1, ls2 = f(q + "layer_scale1.lambda1"), f(q + "layer_scale2.lambda1")
        wo = torch.zeros((C, H, HP)); wo[:, :, :hd] = (f(q + "attention.output.dense.weight") * ls1[:, None]).reshape(C, H, hd)
        w[p + "o.w"], w[p + "o.b"] = wo.reshape(C, H * HP), f(q + "attention.output.dense.bias") * ls1
        w[p + "fc1.w"], w[p + "fc1.b"] = f(q + "mlp.fc1.weight"), f(q + "mlp.fc1.bias")
        w[p + "fc2.w"], w[p + "fc2.b"] = f(q + "mlp.fc2.weight") * ls2[:, None], f(q + "mlp.fc2.bias") * ls2
        for n in ("norm1", "norm2"):
            w[p + n + ".w"], w[p + n + ".b"] = f(q + n + ".weight"), f(q + n + ".bias")
    w["norm.w"], w["norm.b"] = f("layernorm.weight"), f("layernorm.bias")
    return w


def position_rows(pos: torch.Tensor, cls: torch.Tensor, trained_side: int, n_h: int, n_w: int) -> torch.Tensor:
    """(1 + n_h n_w, C) fp32: row 0 = class token + class position, rows 1.. = the (resampled) patch positions
    (Dinov2Embeddings.interpolate_pos_encoding + the cat / add of Dinov2Embeddings.forward)."""
    patch = pos[1:]
    if (n_h, n_w) != (trained_side, trained_side):
        grid = patch.reshape(1, trained_side, trained_side, -1).permute(0, 3, 1, 2)
        grid = torch.nn.functional.interpolate(grid, size=(n_h, n_w), mode="bicubic", align_corners=False)
        patch = grid.permute(0, 2, 3, 1).reshape(n_h * n_w, -1)
    return torch.cat([(pos[0] + cls)[None], patch], dim=0)


class HipImageEncoder:
    def __init__(self, pretrained_dino_feature_extractor: Optional[str] = None, pretrained_dino_model: Optional[str] = None,
                 config: Optional[Dict] = None, state_dict: Optional[Dict[str, torch.Tensor]] = None, dtype=None,
                 residual_fp32: bool = True, **_ignored):
        lib()      # fail loudly here if libactionmesh_amd.so is missing
        self.residual_fp32 = bool(residual_fp32)
        # 16-bit storage type.  The reference encodes OUTSIDE its autocast region, in fp32 (pipeline.py:665-667): there is no caller dtype
        # to follow, so the default is bfloat16 (fp32's exponent range: safe for DINOv2's outlier tokens whatever the checkpoint) and
        # dtype="float16" selects the float16 build - 8x finer rounding, measured closer to the fp32 reference on the ViT-L/14 fixture
        # (tests/test_image_encoder.py::test_hip_encoder_vitl_against_transformers); anything else is refused (there is no fp32 library).
        self.kind = "bf16" if dtype is None else L.kind_of(dtype)
        if self.kind == "f16":
            lib("f16")
        self.dt16 = L.torch_dtype(self.kind)
        self.pretrained_dino_feature_extractor = pretrained_dino_feature_extractor
        self.pretrained_dino_model = pretrained_dino_model
        self._device = torch.device("cpu")
        self._packed: Optional[Dict[str, torch.Tensor]] = None
        self._w: Dict[str, torch.Tensor] = {}
        self._pos_cache: Dict[Tuple[int, int, int], torch.Tensor] = {}
        self._processor = None
        self.cfg = dict(_CFG_DEFAULTS)
        if pretrained_dino_model is not None and state_dict is None:
            config, state_dict = self._read_pretrained(pretrained_dino_model)
        if config is not None:
            self.cfg.update({k: config[k] for k in _CFG_DEFAULTS if k in config})
        C, H = self.cfg["hidden_size"], self.cfg["num_attention_heads"]
        if self.cfg["use_swiglu_ffn"]:
            raise ValueError("HipImageEncoder: the SwiGLU FFN variant (DINOv2 giant) is not supported")
        if C % H or C // H > ops.HEAD_DIM or C % 64:
            raise ValueError(f"HipImageEncoder: hidden_size={C}, heads={H}: need head_dim <= 128 and width % 64 == 0")
        if state_dict is not None:
            self.load_state_dict(state_dict)

    @staticmethod
    def _read_pretrained(path: str):
        """<path>/config.json + model.safetensors (or pytorch_model.bin): the layout Dinov2Model.from_pretrained reads
        (image_encoder.py:25-27)."""
        import json
        import os
        with open(os.path.join(path, "config.json")) as fh:
            cfg = json.load(fh)
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        return cfg, {k[len("dinov2."):] if k.startswith("dinov2.") else k: v for k, v in sd.items()}

    # ---- nn.Module-like surface ---------------------------------------------------------------------------
    @property
    def device(self) -> torch.device:
        return self._device

    def eval(self):
        return self

    def to(self, device):
        self._device = torch.device(device)
        if self._packed is not None and self._device.type == "cuda":
            self._upload()
        return self

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        want = state_dict_shapes(self.cfg)
        missing = [k for k in want if k not in sd]
        if missing:
            raise KeyError(f"HipImageEncoder.load_state_dict: missing {missing[:4]}{' ...' if len(missing) > 4 else ''}")
        bad = [(k, tuple(sd[k].shape), v) for k, v in want.items() if tuple(sd[k].shape) != v]
        if bad:
            raise ValueError(f"HipImageEncoder.load_state_dict: shape mismatch (key, got, expected): {bad[:4]}")
        self._packed = pack_weights(sd, self.cfg)
        self._pos_cache.clear()
        if self._device.type == "cuda":
            self._upload()
        return self

    def _upload(self) -> None:
        dev = self._device
        self._w = {k: v.to(dev, torch.float32 if (k.endswith(".b") or k.startswith("norm") or ".norm" in k) else self.dt16).contiguous()
                   for k, v in self._packed.items() if k not in ("pos", "cls")}
        self._pos_cache.clear()

    def _pos_rows(self, T: int, n_h: int, n_w: int) -> torch.Tensor:
        key = (T, n_h, n_w)
        if key not in self._pos_cache:
            side = self.cfg["image_size"] // self.cfg["patch_size"]
            rows = position_rows(self._packed["pos"], self._packed["cls"], side, n_h, n_w)
            self._pos_cache = {key: rows.to(self._device, self.dt16)[None].expand(T, -1, -1).reshape(T * rows.shape[0], -1).contiguous()}
        return self._pos_cache[key]

    def _pos_rows32(self, T: int, n_h: int, n_w: int) -> torch.Tensor:
        """The same rows in fp32 (the fp32 residual stream starts from them)."""
        key = (T, n_h, n_w, 32)
        if key not in self._pos_cache:
            side = self.cfg["image_size"] // self.cfg["patch_size"]
            rows = position_rows(self._packed["pos"], self._packed["cls"], side, n_h, n_w)
            self._pos_cache[key] = rows.to(self._device, torch.float32)[None].expand(T, -1, -1).reshape(T * rows.shape[0], -1).contiguous()
        return self._pos_cache[key]

    # ---- forward ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_pixels(self, pixel_values: torch.Tensor, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """Dinov2Model(pixel_values).last_hidden_state: (T, 3, H, W) -> (T, 1 + (H/p)(W/p), C).
        `out_dtype=torch.bfloat16` (the encoder's own 16-bit type) hands the context over in the dtype `am_set_context` stores it in."""
        if self._device.type != "cuda" or not self._w:
            raise RuntimeError("HipImageEncoder: load_state_dict(...) and .to('cuda:N') first (there is no CPU path)")
        cfg, w, dev = self.cfg, self._w, self._device
        C, H, p, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["patch_size"], cfg["layer_norm_eps"]
        T, Cin, Hi, Wi = pixel_values.shape
        if Cin != cfg["num_channels"]:
            raise ValueError(f"HipImageEncoder: expected {cfg['num_channels']} channels, got {Cin}")
        n_h, n_w = Hi // p, Wi // p
        npatch, S = n_h * n_w, n_h * n_w + 1
        with torch.cuda.device(dev):
            pix = pixel_values.to(dev, torch.float32).contiguous()
            pos = self._pos_rows(T, n_h, n_w)
            dt16 = self.dt16
            HP = ops.HEAD_DIM
            Q = torch.zeros((T, H, ops.round_up(S, 256), HP), dtype=dt16, device=dev)
            K = torch.zeros((T, H, ops.round_up(S, 64), HP), dtype=dt16, device=dev)
            Vt = torch.zeros((T, H, HP, ops.round_up(S, 64)), dtype=dt16, device=dev)
            scale = float(C // H) ** -0.5
            NL = cfg["num_hidden_layers"]
            if self.residual_fp32:
                # fp32 stream: starts as the fp32 position rows (class token + its position in row 0 of every frame); the patch
                # embedding's 16-bit output is the first branch added into it (its class-token rows are zero)
                h32 = self._pos_rows32(T, n_h, n_w).clone()
                y = torch.zeros((T * S, C), dtype=dt16, device=dev)
                ops.gemm(ops.patchify(pix, p, w["patch.w"].shape[1], dtype=dt16), w["patch.w"], bias=w["patch.b"], out=y,
                         c_map=(npatch, S, 1), M=T * npatch)
                for i in range(NL):
                    q = f"l{i}."
                    z = ops.add_layernorm_f32(h32, y, w[q + "norm1.w"], w[q + "norm1.b"], eps=eps)
                    qkv = ops.gemm(z, w[q + "qkv.w"], bias=w[q + "qkv.b"])
                    ops.head_post(qkv, H, (0, 1, 2), S, S, out_q=Q, out_k=K, out_vt=Vt)
                    a = ops.attention(Q, K, Vt, S, S, scale=scale)
                    y = ops.gemm(a, w[q + "o.w"], bias=w[q + "o.b"])
                    z = ops.add_layernorm_f32(h32, y, w[q + "norm2.w"], w[q + "norm2.b"], eps=eps)
                    f = ops.gemm(z, w[q + "fc1.w"], bias=w[q + "fc1.b"], gelu=True)
                    y = ops.gemm(f, w[q + "fc2.w"], bias=w[q + "fc2.b"])
                out = ops.add_layernorm_f32(h32, y, w["norm.w"], w["norm.b"], eps=eps).view(T, S, C)
                if dt16 == torch.float16 and not bool(torch.isfinite(out).all()):
                    raise FloatingPointError("HipImageEncoder: the float16 forward produced non-finite features (IEEE half overflows above "
                                             "65504: DINOv2 outlier tokens); use dtype='bfloat16'")
                return out if out_dtype == out.dtype else out.to(out_dtype)
            h = torch.empty((T * S, C), dtype=dt16, device=dev)
            ops.gemm(ops.patchify(pix, p, w["patch.w"].shape[1], dtype=dt16), w["patch.w"], bias=w["patch.b"], residual=pos, out=h,
                     c_map=(npatch, S, 1), M=T * npatch)
            h.view(T, S, C)[:, 0] = pos[0]                                        # class token + its position
            for i in range(NL):
                q = f"l{i}."
                z = ops.layernorm(h, w[q + "norm1.w"], w[q + "norm1.b"], eps=eps)
                qkv = ops.gemm(z, w[q + "qkv.w"], bias=w[q + "qkv.b"])
                ops.head_post(qkv, H, (0, 1, 2), S, S, out_q=Q, out_k=K, out_vt=Vt)
                a = ops.attention(Q, K, Vt, S, S, scale=scale)
                h = ops.gemm(a, w[q + "o.w"], bias=w[q + "o.b"], residual=h)
                z = ops.layernorm(h, w[q + "norm2.w"], w[q + "norm2.b"], eps=eps)
                f = ops.gemm(z, w[q + "fc1.w"], bias=w[q + "fc1.b"], gelu=True)
                h = ops.gemm(f, w[q + "fc2.w"], bias=w[q + "fc2.b"], residual=h)
            y = ops.layernorm(h, w["norm.w"], w["norm.b"], eps=eps).view(T, S, C)
            return y if out_dtype == y.dtype else y.to(out_dtype)

    def encode_images(self, images: List) -> torch.Tensor:
        """image_encoder.py:38-55: T PIL images -> context (T, S, Dc)."""
        if self._processor is None:
            if self.pretrained_dino_feature_extractor is None:
                raise RuntimeError("HipImageEncoder.encode_images needs pretrained_dino_feature_extractor "
                                   "(the BitImageProcessor config); use encode_pixels for preprocessed input")
            from transformers import BitImageProcessor           # CPU glue of the reference path, imported lazily
            self._processor = BitImageProcessor.from_pretrained(self.pretrained_dino_feature_extractor)
        pixel_values = self._processor.preprocess(images, return_tensors="pt").pixel_values
        return self.encode_pixels(pixel_values)

    def step_flops(self, T: int, height: int, width: int) -> float:
        """Algorithmic flops of one call (un-padded head_dim, MACs x 2)."""
        cfg = self.cfg
        C, Fi, p = cfg["hidden_size"], cfg["hidden_size"] * cfg["mlp_ratio"], cfg["patch_size"]
        npatch = (height // p) * (width // p)
        S = npatch + 1
        per_layer = 8 * S * C * C + 4 * S * S * C + 4 * S * C * Fi
        return float(T * (cfg["num_hidden_layers"] * per_layer + 2 * npatch * cfg["num_channels"] * p * p * C))
