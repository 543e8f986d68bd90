#!/usr/bin/env python3
"""CPU emulation (oracle, bf16 policy) of e4m3 GEMM operands for the linears that would run on the fp8 MFMA - which ones can
afford it?  Activations quantised at unit scale (LayerNorm outputs / GELU outputs are O(1)), weights per output channel
(amax / 448).  Prints the one-forward rel-L2 vs the reference's fp32 velocity at the arch_headline fixture for several sets."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import denoiser_oracle as O
from oracle.make_golden_baseline import baseline_case_inputs

E4 = torch.float8_e4m3fn


def q8(x):
    return x.clamp(-448.0, 448.0).to(E4).to(torch.float32)


class P8(O.Precision):
    def __init__(self, ids):
        super().__init__("bf16")
        self.ids = ids

    def linear(self, x, w, b):
        if id(w) not in self.ids:
            return super().linear(x, w, b)
        s = w.abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / 448.0
        w8 = q8(self.rb(w) / s) * s
        y = F.linear(q8(self.rb(x)), w8, None if b is None else self.rb(b))
        return self.rb(y)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "arch_headline"
    g = np.load(f"tests/golden/{name}.npz")
    kw, cfg, sd, inp, steps = baseline_case_inputs(name)
    x_in, c_in, m_in, f_in = O.cfg_at_inference(inp["init_latent"], inp["context"], inp["mask"], inp["framestep"], [[0, 1], [1, 1]])
    t = float(g["fwd_t"]) if "fwd_t" in g else 640.0
    tt = torch.tensor([t]).expand(2)
    ref32 = torch.from_numpy(g["fwd_velocity_fp32"]) if "fwd_velocity_fp32" in g else O.denoiser_forward(sd, cfg, x_in, c_in, f_in, tt, m_in, "fp32")
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    sets = {
        "none (bf16 policy)": [],
        "ff2": ["ff.net.2.weight"],
        "ff1+ff2": ["ff.net.0.proj.weight", "ff.net.2.weight"],
        "qkv": ["s_attn.to_q.weight", "s_attn.to_k.weight", "s_attn.to_v.weight"],
        "qkv+xq+ff1+ff2": ["s_attn.to_q.weight", "s_attn.to_k.weight", "s_attn.to_v.weight", "x_attn.to_q.weight", "ff.net.0.proj.weight", "ff.net.2.weight"],
        "all block linears": ["s_attn.to_q.weight", "s_attn.to_k.weight", "s_attn.to_v.weight", "s_attn.to_out.0.weight", "x_attn.to_q.weight",
                              "x_attn.to_out.0.weight", "ff.net.0.proj.weight", "ff.net.2.weight"],
    }
    for label, suffixes in sets.items():
        ids = {id(v) for k, v in sd.items() if any(k.endswith(s) for s in suffixes)}
        P = P8(ids)
        orig = O.Precision
        O.Precision = lambda mode="fp32": P          # denoiser_forward builds its Precision from the mode string
        try:
            v = O.denoiser_forward(sd, cfg, x_in, c_in, f_in, tt, m_in, "bf16")
        finally:
            O.Precision = orig
        print(f"{name}: e4m3 operands in [{label}]: forward rel-L2 vs fp32 reference {rel(v, ref32):.3e}", flush=True)


if __name__ == "__main__":
    torch.set_num_threads(16)
    main()
