#!/usr/bin/env python
"""Multi-rank self-test of the sharded denoiser under torch.distributed (RCCL): every rank runs the sharded forward
(CFG-branch split x frame shards, async [K|V^T] all-gather overlapped with the local-shard attention pass), rank 0
also runs the unsharded forward and compares.  Launch:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/mgpu_selftest.py
`--same-device` puts every rank on cuda:0 (only useful where the collective library tolerates it)."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--same-device", action="store_true")
    ap.add_argument("--tokens", type=int, default=511)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"])
    a = ap.parse_args()
    from actionmesh_amd import ClassifierFreeGuidance, HipDenoiser
    from oracle import denoiser_oracle as O     # synthetic weights only
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    local = 0 if a.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl")
    hp = dict(in_channels=64, num_layers=3, num_attention_heads=2, width=256, mlp_ratio=4.0, cross_attention_dim=64,
              inflated_layers=(0, 1, 2))
    sd = O.synthetic_state_dict(O.OracleConfig(**hp), seed=3)
    T, N, S = a.frames, a.tokens, 9
    g = torch.Generator().manual_seed(11)
    x = torch.randn((1, T, N, 64), generator=g)
    ctx = torch.randn((1, T, S, 64), generator=g)
    mask = torch.zeros(1, T); mask[0, 0] = 1
    fs = torch.arange(T, dtype=torch.float32)[None]
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    x_in, c_in, m_in, f_in = cfgd.cfg_at_inference(x, ctx, mask, fs)
    tt = torch.tensor([640.0, 640.0])

    def run(group):
        m = HipDenoiser(num_tokens_nominal=N, temporal_context_size=T, process_group=group, attn_dtype=a.dtype, **hp)
        m.load_state_dict(sd)
        m.to(dev).eval()
        v, _ = m.forward(x_in.to(dev), c_in.to(dev), f_in.to(dev), tt.to(dev), m_in.to(dev), None)
        torch.cuda.synchronize(dev)
        n8, n16 = m._engine.attention_counters()
        assert (n8 > 0 and n16 == 0) if a.dtype == "fp8" else (n8 == 0 and n16 > 0), f"--dtype {a.dtype} but launches fp8 {n8} / bf16 {n16}"
        return v.float().cpu()

    v = run(dist.group.WORLD)
    if rank == 0:
        ref = run(None)
        r = float((v - ref).norm() / ref.norm())
        print(f"[mgpu_selftest] world {world}: sharded vs unsharded rel-L2 {r:.3e}", flush=True)
        assert torch.isfinite(v).all() and r < (3e-2 if a.dtype == "fp8" else 1e-2), r
        print("[mgpu_selftest] ok", flush=True)
    dist.barrier(device_ids=[local])
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
