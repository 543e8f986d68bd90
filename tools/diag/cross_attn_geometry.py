#!/usr/bin/env python3
"""Cross-attention shape (257 keys, 4097 queries per frame): the 8-wave / 128 KiB workgroup (product) against the two-4-wave-
workgroups-per-CU geometry (64 KiB each; defer code 58), interleaved, headline and nominal head counts."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from actionmesh_amd import ops
dev = torch.device("cuda:0")


def bench(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for label, nseq, heads, sq in (("headline, both CFG rows", 32, 8, 4097), ("headline, conditional row", 16, 8, 4097), ("nominal, both rows", 32, 16, 2049)):
    sk = 257
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(nseq, heads, ops.round_up(sq, 256), 128, device=dev, generator=g).bfloat16()
    k = torch.randn(nseq, heads, ops.round_up(sk, 64), 128, device=dev, generator=g).bfloat16()
    vt = torch.randn(nseq, heads, 128, ops.round_up(sk, 64), device=dev, generator=g).bfloat16()
    k[:, :, sk:] = 0; vt[..., sk:] = 0
    out = torch.empty((nseq * sq, heads * 128), dtype=torch.bfloat16, device=dev)
    ref = ops.attention(q, k, vt, sq, sk, out=out.clone(), defer_log2=8)
    alt = ops.attention(q, k, vt, sq, sk, out=out.clone(), defer_log2=58)
    d = float((alt.float() - ref.float()).abs().max())
    for rnd in range(2):
        t8 = bench(lambda: ops.attention(q, k, vt, sq, sk, out=out, defer_log2=8))
        t58 = bench(lambda: ops.attention(q, k, vt, sq, sk, out=out, defer_log2=58))
        print(f"{label}: 8-wave / 128 KiB {t8:.4f} ms   2 x 4-wave / 64 KiB {t58:.4f} ms   (max |diff| {d:.2e})", flush=True)
