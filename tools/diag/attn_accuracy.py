#!/usr/bin/env python
"""Distance of the bf16 self-attention launch from an fp64 softmax over ALL keys, on sampled query rows of the headline shape
(16 x 4097 keys, 8 heads, padded keys in the last tile), plain and peaky scores, one and four key chunks.  Two builds of the kernel
that differ in a rounding (round 6: row sums over the rounded probabilities, AM_A64_DOTSUM) print their distances side by side:
    ACTIONMESH_AMD_LIB=build/variants/libam_<v>.so python tools/diag/attn_accuracy.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from actionmesh_amd import ops

dev = torch.device("cuda:0")
T, N, H = 16, 4096, 8
S = T * (N + 1)
g = torch.Generator(device=dev).manual_seed(11)
rows = torch.randint(0, S, (192,), device=dev, generator=g)
for qs, chunks in ((1.0, 1), (4.0, 1), (1.0, 4), (8.0, 1)):
    skc = S // chunks
    Q = torch.zeros((2, H, ops.round_up(S, 256), 128), dtype=torch.bfloat16, device=dev)
    Q[:, :, :S] = (torch.randn((2, H, S, 128), device=dev, generator=g) * qs).to(torch.bfloat16)
    K = torch.zeros((chunks, 2, H, ops.round_up(skc, 64), 128), dtype=torch.bfloat16, device=dev)
    K[:, :, :, :skc] = torch.randn((chunks, 2, H, skc, 128), device=dev, generator=g).to(torch.bfloat16)
    Vt = torch.zeros((chunks, 2, H, 128, ops.round_up(skc, 64)), dtype=torch.bfloat16, device=dev)
    Vt[..., :skc] = torch.randn((chunks, 2, H, 128, skc), device=dev, generator=g).to(torch.bfloat16)
    out = ops.attention(Q, K, Vt, S, skc, nchunks=chunks).view(2, S, H, 128)
    torch.cuda.synchronize()
    num = den = 0.0
    worst = 0.0
    for b in range(2):
        for h in range(H):
            q = Q[b, h, rows].double()                                                # (r, 128)
            k = torch.cat([K[c, b, h, :skc] for c in range(chunks)]).double()       # (chunks * skc, 128)
            pos = torch.arange(ops.round_up(skc, 64), device=dev)      # V^T keeps its keys permuted inside every 16 (bit 2 <-> bit 3)
            pos = ((pos & ~12) | ((pos & 4) << 1) | ((pos & 8) >> 1))[:skc]
            v = torch.cat([Vt[c, b, h][:, pos] for c in range(chunks)], dim=1).double().t()
            ref = torch.softmax(q @ k.t() * 128 ** -0.5, dim=-1) @ v
            d = out[b, rows, h].double() - ref
            num += float((d * d).sum()); den += float((ref * ref).sum())
            worst = max(worst, float(d.abs().max() / ref.pow(2).mean().sqrt()))
    print(f"qscale {qs} chunks {chunks}: rel-L2 vs fp64 {(num / den) ** 0.5:.4e}  max-abs / rms {worst:.4e}  fallback workgroups {ops.attention_fallback_count()}")
