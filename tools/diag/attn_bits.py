#!/usr/bin/env python
"""Position-weighted integer checksum of the bf16 self-attention output on fixed operands (plain and peaky scores, one and four key
chunks, the float16 build too): two builds of the kernel that claim bit-identical results must print the same lines
(round 5: the packed row sums, ACTIONMESH_AMD_LIB=build/variants/libam_nopk.so vs the product)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from actionmesh_amd import ops

dev = torch.device("cuda:0")
T, N, H = 16, 4096, 8
S = T * (N + 1)
g = torch.Generator(device=dev).manual_seed(7)
for dt in (torch.bfloat16, torch.float16):
    for qs, chunks in ((1.0, 1), (4.0, 1), (1.0, 4)):
        skc = S // chunks
        Q = torch.zeros((2, H, ops.round_up(S, 256), 128), dtype=dt, device=dev)
        Q[:, :, :S] = (torch.randn((2, H, S, 128), device=dev, generator=g) * qs).to(dt)
        K = torch.zeros((chunks, 2, H, ops.round_up(skc, 64), 128), dtype=dt, device=dev)
        K[:, :, :, :skc] = torch.randn((chunks, 2, H, skc, 128), device=dev, generator=g).to(dt)
        Vt = torch.zeros((chunks, 2, H, 128, ops.round_up(skc, 64)), dtype=dt, device=dev)
        Vt[..., :skc] = torch.randn((chunks, 2, H, 128, skc), device=dev, generator=g).to(dt)
        out = ops.attention(Q, K, Vt, S, skc, nchunks=chunks)
        torch.cuda.synchronize()
        w = out.view(torch.int16).to(torch.int64)
        idx = torch.arange(w.numel(), device=dev, dtype=torch.int64).view_as(w)
        print(f"{str(dt)[6:]:9s} qscale {qs} chunks {chunks}: checksum {int((w * (2 * idx + 1) % 1000003).sum())}  finite {bool(torch.isfinite(out.float()).all())}")
