#!/usr/bin/env python
"""ONE process, two streams: does head_post's result move when a GEMM runs beside it on another stream?  (r03c: it moves when a GEMM of
ANOTHER PROCESS runs beside it.)  Aggressors: our 128x128 kernel, our 256x256 kernel, torch.matmul (the vendor library), nothing."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from actionmesh_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
rn = lambda *s: torch.randn(*s, generator=g)
B, T, Lr, Cw, H = 2, 4, 512, 256, 2
R = B * T * Lr
x = rn(R, Cw).bfloat16().to(dev)
w_qkv = (rn(3 * Cw, Cw) * Cw ** -0.5).bfloat16().to(dev)
lw, lb = (1 + 0.1 * rn(Cw)).to(dev), (0.1 * rn(Cw)).to(dev)
nq, nk = torch.ones(128, device=dev), torch.ones(128, device=dev)
ang = torch.arange(B * T)[:, None] * (10000.0 ** (-torch.arange(64) * 2 / 128))[None]
cos, sin = torch.cos(ang).float().to(dev), torch.sin(ang).float().to(dev)
z = ops.layernorm(x, lw, lb)
qkv = ops.gemm(z, w_qkv)
q, k, vt = ops.head_post(qkv, H, (0, 1, 2), T * Lr, Lr, w_q=nq, w_k=nk, rope=(cos, sin))
oq, ok_, ov = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(vt)
out_3c = torch.empty((R, 3 * Cw), dtype=torch.bfloat16, device=dev)
torch.cuda.synchronize()
ref = [q.clone(), k.clone(), vt.clone()]
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
work = {
    "none": None,
    "our 128x128 GEMM": lambda: ops.gemm(z, w_qkv, out=out_3c),
    "our 256x256 GEMM": lambda: ops.gemm(z, w_qkv, out=out_3c, force_big=True),
    "torch.matmul (vendor)": lambda: torch.matmul(z, w_qkv.t(), out=out_3c),
}
for name, fn in work.items():
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    n = 0
    for it in range(300):
        if fn is not None:
            with torch.cuda.stream(sb):
                for _ in range(12):
                    fn()
        with torch.cuda.stream(sa):
            for _ in range(10):
                ops.head_post(qkv, H, (0, 1, 2), T * Lr, Lr, w_q=nq, w_k=nk, rope=(cos, sin), out_q=oq, out_k=ok_, out_vt=ov)
                bad += ((oq.view(torch.int16) != ref[0].view(torch.int16)).any() | (ok_.view(torch.int16) != ref[1].view(torch.int16)).any()
                        | (ov.view(torch.int16) != ref[2].view(torch.int16)).any())
                n += 1
    torch.cuda.synchronize()
    print(f"[overlap_probe] one process, second stream runs {name}: {int(bad)}/{n} head_post launches differ", flush=True)
