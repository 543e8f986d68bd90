#!/usr/bin/env python
"""Bitwise repeatability of every kernel of the frame-sharded forward, ONE OP AT A TIME, while other processes hammer the same
GPU with the same ops (launch several copies side by side; no exchange, no IPC, no torch.distributed):
    for i in 0 1; do python tools/divergence/op_determinism.py --reps 300 & done; wait
Each op runs `reps` times on fixed seeded inputs into the same output buffer; every result is compared on the device with
the first one.  Shapes = the peer selftest's (tools/peer_selftest.py: width 256, 2 heads, 4 local frames x 512 rows, 2 shards)
and, with --big, one headline-layer-sized attention.  Separates "a kernel's result depends on what else the chip is doing" from
"the exchange protocol delivered something else"."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from actionmesh_amd import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--defer", type=int, default=8)
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    rn = lambda *s: torch.randn(*s, generator=g)
    B, T, L, C, H, P = 2, 4, 512, 256, 2, 2
    R = B * T * L
    x = rn(R, C).bfloat16().to(dev)
    w_qkv = (rn(3 * C, C) * C ** -0.5).bfloat16().to(dev)
    w_o = (rn(C, C) * C ** -0.5).bfloat16().to(dev)
    w_1 = (rn(4 * C, C) * C ** -0.5).bfloat16().to(dev)
    w_2 = (rn(C, 4 * C) * (4 * C) ** -0.5).bfloat16().to(dev)
    b_o, b_1 = rn(C).to(dev), rn(4 * C).to(dev)
    lw, lb = (1 + 0.1 * rn(C)).to(dev), (0.1 * rn(C)).to(dev)
    nq, nk = torch.ones(128, device=dev), torch.ones(128, device=dev)
    cos = torch.cos(torch.arange(B * T)[:, None] * torch.arange(64)[None] * 0.01).float().to(dev)
    sin = torch.sin(torch.arange(B * T)[:, None] * torch.arange(64)[None] * 0.01).float().to(dev)
    z = ops.layernorm(x, lw, lb)
    qkv = ops.gemm(z, w_qkv)
    ffh = ops.gemm(z, w_1, bias=b_1, gelu=True)
    q, k, vt = ops.head_post(qkv, H, (0, 1, 2), T * L, L, w_q=nq, w_k=nk, rope=(cos, sin))
    # two shards: [chunk][seq][H][...]
    k2 = torch.stack([k, torch.roll(k, 3, dims=2)]).contiguous()
    vt2 = torch.stack([vt, torch.roll(vt, 16, dims=3)]).contiguous()
    sq = T * L
    state = torch.zeros((B * H, q.shape[2], ops.STATE_LD), dtype=torch.float32, device=dev)
    ao = torch.empty((B * sq, C), dtype=torch.bfloat16, device=dev)
    torch.cuda.synchronize()

    def two_pass(rank):
        ops.attention(q, k2, vt2, sq, sq, out=ao, nchunks=1, defer_log2=a.defer, rows=1, state_mode=1, state=state,
                      chunk_first=rank, chunk_total=P)
        ops.attention(q, k2, vt2, sq, sq, out=ao, nchunks=P - 1, defer_log2=a.defer, rows=1, state_mode=2, state=state,
                      chunk_first=(rank + 1) % P, chunk_total=P)
        return ao

    # cross-attention shape: 9 keys per frame
    S = 9
    kx = torch.zeros((B * T, H, 64, 128), dtype=torch.bfloat16, device=dev); kx[:, :, :S] = rn(B * T, H, S, 128).bfloat16().to(dev)
    vx = torch.zeros((B * T, H, 128, 64), dtype=torch.bfloat16, device=dev); vx[..., :S] = rn(B * T, H, 128, S).bfloat16().to(dev)
    qx, _, _ = ops.head_post(qkv[:, :C].contiguous(), H, (0,), L, L, w_q=nq)
    aox = torch.empty((B * T * L, C), dtype=torch.bfloat16, device=dev)
    out_c = torch.empty((R, C), dtype=torch.bfloat16, device=dev)
    out_3c = torch.empty((R, 3 * C), dtype=torch.bfloat16, device=dev)
    out_4c = torch.empty((R, 4 * C), dtype=torch.bfloat16, device=dev)
    oq, ok_, ov = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(vt)

    def hp():
        ops.head_post(qkv, H, (0, 1, 2), T * L, L, w_q=nq, w_k=nk, rope=(cos, sin), out_q=oq, out_k=ok_, out_vt=ov)
        return torch.cat([oq.flatten(), ok_.flatten(), ov.flatten()])

    table = {
        "layernorm": lambda: ops.layernorm(x, lw, lb, out=out_c),
        "gemm qkv (128^2 kernel)": lambda: ops.gemm(z, w_qkv, out=out_3c),
        "gemm out+res": lambda: ops.gemm(z, w_o, bias=b_o, residual=x, out=out_c),
        "gemm ff1+gelu": lambda: ops.gemm(z, w_1, bias=b_1, gelu=True, out=out_4c),
        "gemm ff2+res": lambda: ops.gemm(ffh, w_2, bias=b_o, residual=x, out=out_c),
        "gemm qkv (256^2 pp forced)": lambda: ops.gemm(z, w_qkv, out=out_3c, force_big=True),
        "head_post": hp,
        "attn one-pass 2 chunks": lambda: ops.attention(q, k2, vt2, sq, sq, out=ao, nchunks=P, defer_log2=a.defer),
        "attn two-pass rank0": lambda: two_pass(0),
        "attn two-pass rank1": lambda: two_pass(1),
        "attn cross (8-wave)": lambda: ops.attention(qx, kx, vx, L, S, out=aox, defer_log2=a.defer),
    }
    if a.big:
        Hb, sqb = 8, 16 * 1025
        gq = torch.Generator().manual_seed(7)
        qb = torch.zeros((1, Hb, ops.round_up(sqb, 256), 128), dtype=torch.bfloat16, device=dev)
        qb[:, :, :sqb] = torch.randn((1, Hb, sqb, 128), generator=gq).bfloat16().to(dev)
        kb = torch.zeros((1, Hb, ops.round_up(sqb, 64), 128), dtype=torch.bfloat16, device=dev)
        kb[:, :, :sqb] = torch.randn((1, Hb, sqb, 128), generator=gq).bfloat16().to(dev)
        vb = torch.zeros((1, Hb, 128, ops.round_up(sqb, 64)), dtype=torch.bfloat16, device=dev)
        vb[..., :sqb] = torch.randn((1, Hb, 128, sqb), generator=gq).bfloat16().to(dev)
        aob = torch.empty((sqb, Hb * 128), dtype=torch.bfloat16, device=dev)
        table["attn big one-pass (16400 x 16400, 8 heads)"] = lambda: ops.attention(qb, kb, vb, sqb, sqb, out=aob, defer_log2=a.defer)
        xb = rn(16 * 1025, 1024).bfloat16().to(dev); wb = (rn(1024, 1024) / 32).bfloat16().to(dev)
        ob = torch.empty((16 * 1025, 1024), dtype=torch.bfloat16, device=dev)
        table["gemm big 16400x1024x1024 +res (pp)"] = lambda: ops.gemm(xb, wb, bias=torch.zeros(1024, device=dev), residual=xb, out=ob)
    pid = os.getpid()
    for name, fn in table.items():
        if a.only and a.only not in name:
            continue
        t0 = time.time()
        first = fn().clone()
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        worst = torch.zeros((), dtype=torch.int64, device=dev)
        for _ in range(a.reps):                      # no host sync inside: the processes' launches stay interleaved
            o = fn()
            n = (o.view(torch.int16) != first.view(torch.int16)).sum()
            bad += (n > 0)
            worst = torch.maximum(worst, n)
        bad, worst = int(bad), int(worst)
        print(f"[op_determinism pid {pid}] {name}: {bad}/{a.reps} repeats differ from the first"
              + (f" (up to {worst} of {first.numel()} elements)" if bad else "") + f"  [{time.time() - t0:.1f} s]", flush=True)


if __name__ == "__main__":
    main()
