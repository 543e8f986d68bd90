#!/usr/bin/env python3
"""Diagnostic: the 2-rank emulated forward of tests/test_denoiser_gpu.py vs the unsharded one, folded / un-folded LayerNorms, each
also against the fp32 reference fixture - is the larger sharded-vs-unsharded distance of the folded form an error or decorrelation?"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

from oracle import denoiser_oracle as O
from actionmesh_amd import ClassifierFreeGuidance, HipDenoiser
from actionmesh_amd.denoiser import HipEngine, masked_time, rope_tables_host
from actionmesh_amd.sharding import FrameShardPlan

CASE = dict(in_channels=64, num_layers=5, num_attention_heads=2, width=256, mlp_ratio=4.0, cross_attention_dim=64, inflated_layers=(0, 1, 2, 3, 4))
dev = torch.device("cuda:0")


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


def run(env):
    for k, v in env.items():
        os.environ[k] = v
    g = np.load("tests/golden/tiny_inflated.npz")
    cfg = O.OracleConfig(**CASE)
    sd = O.synthetic_state_dict(cfg, seed=0)
    model = HipDenoiser(num_tokens_nominal=48, temporal_context_size=4, **CASE)
    model.load_state_dict(sd)
    model.to(dev).eval()
    t = {k: torch.from_numpy(g[k]) for k in ("init_latent", "context", "mask", "framestep")}
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    x_in, c_in, m_in, f_in = cfgd.cfg_at_inference(t["init_latent"], t["context"], t["mask"], t["framestep"])
    B, T, N, _ = x_in.shape
    S = c_in.shape[2]
    tt = [float(g["fwd_t"])] * B
    ref, _ = model.forward(x_in.to(dev), c_in.to(dev), f_in.to(dev), torch.tensor(tt, device=dev), m_in.to(dev), None)
    t_bt = masked_time(tt, m_in, B, T)
    cos, sin = rope_tables_host(f_in, 128)
    outs = {}
    for world in (1, 2):
        engines = []
        for r in range(world):
            plan = FrameShardPlan(T, world, r)
            e = HipEngine(model.hyper_params(), sd, dev, B, plan.frames_local, N, S, world=world, rank=r)
            e.set_context(plan.slice_frames(c_in.to(dev)), cos.view(B, T, -1)[:, plan.frame_slice].reshape(-1, 64),
                          sin.view(B, T, -1)[:, plan.frame_slice].reshape(-1, 64))
            tl = plan.frames_local
            e.begin(plan.slice_frames(x_in.to(dev)), [t_bt[b * T + r * tl + j] for b in range(B) for j in range(tl)])
            engines.append(e)
        for i in range(cfg.num_layers):
            for e in engines:
                e.layer_pre(i)
            if world == 2 and engines[0].is_inflated(i):
                (kv0,), (kv1,) = engines[0].kv_buffers(), engines[1].kv_buffers()
                kv0[1].copy_(kv1[1]); kv1[0].copy_(kv0[0])
            for e in engines:
                e.layer_post(i)
        outs[world] = torch.cat([e.end() for e in engines], dim=1).float().cpu()
    torch.cuda.synchronize()
    for k in env:
        del os.environ[k]
    return ref.float().cpu(), outs[1], outs[2], torch.from_numpy(g["fwd_velocity_fp32"])


for name, env in (("folded", {}), ("un-folded", {"ACTIONMESH_AMD_LN_FOLD": "0"})):
    ref, w1, w2, ref32 = run(env)
    print(f"{name:10s}: model.forward vs fp32 ref {rel(ref, ref32):.3e} | engine world 1 vs model.forward {rel(w1, ref):.3e} | "
          f"world 2 vs fp32 ref {rel(w2, ref32):.3e} | world 2 vs world 1 {rel(w2, w1):.3e}", flush=True)
