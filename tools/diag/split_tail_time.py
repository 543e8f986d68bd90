#!/usr/bin/env python3
"""Time of the inflated self-attention's split tail alone (ops.attention(rows=2): the short last query block cut 16 ways over the
keys + the merge) at the headline shape; the rocprofv3 traces up to r04_cccf5cdf show 143 us + 9 us for it in the 256-row geometry."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from actionmesh_amd import ops
dev = torch.device("cuda:0")
T, N, H, B = 16, 4096, 8, 2
sq = T * (N + 1)
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(B, H, ops.round_up(sq, 256), 128, device=dev, generator=g).bfloat16()
k = torch.randn(B, H, ops.round_up(sq, 64), 128, device=dev, generator=g).bfloat16()
vt = torch.randn(B, H, 128, ops.round_up(sq, 64), device=dev, generator=g).bfloat16()
out = torch.zeros((B * sq, H * 128), dtype=torch.bfloat16, device=dev)
full = ops.attention(q, k, vt, sq, sq, out=out.clone())
ops.attention(q, k, vt, sq, sq, out=out, rows=1)
ops.attention(q, k, vt, sq, sq, out=out, rows=2)
print("rows=1 + rows=2 equals the one-call result:", bool(torch.equal(out, full)))
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        ops.attention(q, k, vt, sq, sq, out=out, rows=2)
    e1.record(); torch.cuda.synchronize()
    print(f"split tail + merge: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call", flush=True)
