#!/usr/bin/env python3
"""Where does tests/test_f16_gpu.py::test_gemm_f16[4128-1024-1024] exceed its bound?  Prints the worst elements with the rounded
linear output and the residual beside them (cancellation in `lin + res` puts the linear's own rounding step next to a tiny result)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from actionmesh_amd import ops
dev = torch.device("cuda:0")
M, N, K = 4128, 1024, 1024
g = torch.Generator().manual_seed(M + N)
a = torch.randn(M, K, generator=g).half(); w = (torch.randn(N, K, generator=g) * K ** -0.5).half()
bias = torch.randn(N, generator=g); res = torch.randn(M, N, generator=g).half()
for name, kw in (("default", {}), ("force_small", dict(force_small=True)), ("force_big", dict(force_big=True))):
    out = ops.gemm(a.to(dev), w.to(dev), bias=bias.to(dev), residual=res.to(dev), **kw).double().cpu()
    lin64 = a.double() @ w.double().T + bias.double()
    lin = lin64.float().half().double()
    ref = (lin + res.double()).float().half().double()
    err = (out - ref).abs()
    bound = 2.0 * ref.abs() * 2.0 ** -10 + 3e-4
    bound2 = 2.0 * torch.maximum(ref.abs(), lin.abs()) * 2.0 ** -10 + 3e-4
    r = err / bound
    idx = torch.topk(r.flatten(), 4).indices
    print(name, "worst err/bound", float(r.max()), "with the linear's magnitude in the bound", float((err / bound2).max()),
          "elements over the old bound", int((r > 1).sum()))
    for i in idx.tolist():
        m, n = divmod(i, N)
        print(f"   ({m},{n}) out {out[m, n]:.6f} ref {ref[m, n]:.6f} lin64 {lin64[m, n]:.6f} lin16 {lin[m, n]:.6f} res {res[m, n]:.6f} ulp(lin) {abs(lin[m, n].item()) * 2 ** -10:.2e}")
