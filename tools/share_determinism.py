#!/usr/bin/env python
"""Is the unsharded forward bitwise repeatable while ANOTHER process computes on the same GPU?  (No exchange, no
torch.distributed: launch two copies side by side.)  Isolates GPU-sharing effects from the copy-engine exchange protocol.
    python tools/share_determinism.py [forwards] [tokens] & python tools/share_determinism.py ... ; wait"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from actionmesh_amd import ClassifierFreeGuidance
from actionmesh_amd.denoiser import HipEngine, masked_time, rope_tables_host
from oracle import denoiser_oracle as O     # synthetic weights only

n_fwd = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N = int(sys.argv[2]) if len(sys.argv) > 2 else 511
dev = torch.device("cuda:0")
hp = dict(in_channels=64, num_layers=3, num_attention_heads=2, width=256, mlp_ratio=4.0, cross_attention_dim=64, inflated_layers=[0, 1, 2])
sd = O.synthetic_state_dict(O.OracleConfig(**{**hp, "inflated_layers": (0, 1, 2)}), seed=3)
T, S, B = 8, 9, 2
g = torch.Generator().manual_seed(11)
x = torch.randn((1, T, N, 64), generator=g); ctx = torch.randn((1, T, S, 64), generator=g)
mask = torch.zeros(1, T); mask[0, 0] = 1
fs = torch.arange(T, dtype=torch.float32)[None]
x_in, c_in, m_in, f_in = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5]).cfg_at_inference(x, ctx, mask, fs)
t_bt = masked_time([640.0, 640.0], m_in, B, T)
cos, sin = rope_tables_host(f_in, 128)
eng = HipEngine(hp, sd, dev, B, T, N, S)
eng.set_context(c_in.to(dev), cos, sin)
xd = x_in.to(dev)
outs = []
for _ in range(n_fwd):
    outs.append(eng.forward(xd, t_bt).float().cpu())
bad = [k for k, o in enumerate(outs) if not torch.equal(o, outs[0])]
for k in bad[:4]:
    d = (outs[k] - outs[0]).abs()
    print(f"[share_determinism pid {os.getpid()}] forward {k}: {int((d > 0).sum())}/{d.numel()} differ, max {float(d.max()):.3e}, "
          f"tokens hit {int((d > 0).any(-1).sum())}", flush=True)
print(f"[share_determinism pid {os.getpid()}] {n_fwd} forwards, {len(bad)} differ from the first", flush=True)
