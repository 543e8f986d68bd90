#!/usr/bin/env python
"""Context encoder (DINOv2 ViT-L/14 on the Stage-I kernels) on synthetic data of the shipped shape: 16 frames of
224 x 224 pixels -> context (16, 257, 1024).  Prints one JSON line (secondary metric; bench.py stays the headline)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float16"])
    a = ap.parse_args()
    from actionmesh_amd import image_encoder as IE
    dev = torch.device("cuda:0")
    cfg = dict(IE._CFG_DEFAULTS)
    g = torch.Generator().manual_seed(0)
    sd = {}
    for name, shape in IE.state_dict_shapes(cfg).items():
        if name.endswith(".weight") and len(shape) >= 2:
            fan = 1
            for s in shape[1:]:
                fan *= s
            sd[name] = torch.randn(shape, generator=g) / fan ** 0.5
        elif name.endswith(".weight") or name.endswith("lambda1"):
            sd[name] = torch.ones(shape)
        elif name.startswith("embeddings.") and not name.endswith(".bias"):
            sd[name] = 0.5 * torch.randn(shape, generator=g)
        else:
            sd[name] = torch.zeros(shape)
    enc = IE.HipImageEncoder(state_dict=sd, dtype=a.dtype, residual_fp32=os.environ.get("ACTIONMESH_AMD_RESIDUAL_FP32", "1") != "0").to(dev)
    pix = torch.randn((a.frames, 3, 224, 224), generator=g).to(dev)
    out = enc.encode_pixels(pix)
    torch.cuda.synchronize()
    assert out.shape == (a.frames, 257, 1024) and bool(torch.isfinite(out).all())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        enc.encode_pixels(pix, out_dtype=enc.dt16)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    fl = enc.step_flops(a.frames, 224, 224)
    print(json.dumps({"metric": "context-encoder calls/sec (16 frames x 224x224, DINOv2 ViT-L/14)", "value": round(1e3 / ms, 2),
                      "unit": "calls/s", "ms_per_call": round(ms, 3), "n_gpus": 1, "dtype": "bf16" if a.dtype == "bfloat16" else "f16", "data": "synthetic",
                      "algorithmic_flops": fl, "tflops": round(fl / ms / 1e9, 1),
                      "frac_of_bf16_peak": round(fl / ms / 1e9 / 2500.0, 4),
                      "config": {"workload": f"T={a.frames} 224x224 patch 14 width 1024 heads 16x64 layers 24"}}))


if __name__ == "__main__":
    main()
