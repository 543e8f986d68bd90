#!/usr/bin/env python
"""Stage II (ActionMeshAutoencoder on the Stage-I kernels) on synthetic data of the shipped shape: one 16-frame
window, N = 2048 latent tokens, width 1024 / 8 heads / 16 self-attention layers + 1 cross-attention layer, V query
vertices, T_out target timesteps.  Prints one JSON line (secondary metric; bench.py stays the headline bench)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--targets", type=int, default=15)
    ap.add_argument("--vertices", type=int, default=50000)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--tokens", type=int, default=2048)
    ap.add_argument("--layers", type=int, default=16)
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float16"], help="the 16-bit type (round 5: float16 = libactionmesh_amd_f16.so)")
    a = ap.parse_args()
    from actionmesh_amd.autoencoder import HipAutoencoder
    from oracle.autoencoder_oracle import AEConfig, state_dict_spec      # shapes only (random weights below)
    dev = torch.device("cuda:0")
    C, H = 1024, 8
    cfg = AEConfig(width=C, num_layers=a.layers, num_attention_heads=H)
    g = torch.Generator().manual_seed(0)
    sd = {}
    for name, shape in state_dict_spec(cfg):
        if name.endswith(".weight") and len(shape) == 2:
            sd[name] = torch.randn(shape, generator=g) / shape[1] ** 0.5
        elif name.endswith(".weight"):
            sd[name] = torch.ones(shape)
        else:
            sd[name] = torch.zeros(shape)
    m = HipAutoencoder(width=C, num_layers=a.layers, num_attention_heads=H, dtype=a.dtype,
                       residual_fp32=os.environ.get("ACTIONMESH_AMD_RESIDUAL_FP32", "1") != "0")   # A/B of the round-6 fp32 residual stream
    m.load_state_dict(sd)
    m.to(dev)
    T, N, V, To = a.frames, a.tokens, a.vertices, a.targets
    latent = torch.randn((1, T, N, 64), generator=g).to(dev)
    framestep = torch.arange(T, dtype=torch.float32)[None]
    query = torch.cat([torch.rand((1, V, 3), generator=g) * 2 - 1, torch.randn((1, V, 3), generator=g)], -1).to(dev)
    src, tgt = torch.tensor([0.0]), torch.linspace(0, 1, To)[None]
    m(latent, framestep, src, tgt[:, :1], query)                 # warm-up: one target
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d = m(latent, framestep, src, tgt, query)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert bool(torch.isfinite(d).all())
    TL = T * (N + 1)
    per_layer = 4.0 * TL * TL * C + 2.0 * TL * 12 * C * C
    cross = 4.0 * V * TL * C + 2.0 * TL * 2 * C * C + 2.0 * V * 10 * C * C
    flops = To * (a.layers * per_layer + cross)
    print(json.dumps({"metric": "stage-II decode windows/sec (16f x 2048tok, 15 targets, 50k vertices)",
                      "value": round(1.0 / dt, 4), "unit": "windows/s", "seconds_per_window": round(dt, 3),
                      "n_gpus": 1, "dtype": "bf16" if a.dtype == "bfloat16" else "f16", "data": "synthetic", "algorithmic_flops": flops,
                      "tflops": round(flops / dt / 1e12, 1), "frac_of_bf16_peak": round(flops / dt / 2.5e15, 4),
                      "config": {"workload": f"T={T} N={N} width={C} heads={H} layers={a.layers}+1 targets={To} vertices={V}"}}))


if __name__ == "__main__":
    main()
