"""tests/golden/ar_configs2_yardstick.json: how far REDUCED PRECISION ITSELF is from fp32 on the configs[2] window loop of
tests/test_denoiser_gpu.py::test_autoregressive_windows_configs2_at_the_headline_architecture.

TEST INFRASTRUCTURE ONLY.  python oracle/make_golden_ar_configs2.py   (~2.5 min of host time; no /root/reference needed)

The case: 32 frames, window 16, slide 15, anchor 0 (three dependent windows), the headline architecture (21 layers, width 1024, 8 heads,
Dc 1024, S 257), 47 latent tokens per frame, 3 sampler steps per window, seeded weights / context / anchor, CPU-drawn noise (seed 44 + i).
Stored: per frame, the rel-L2 distance between oracle/windows_oracle.py run with the bf16 POLICY of oracle/denoiser_oracle.py (every
rounding point of the reference's cuda-autocast flow restated; pinned to the reference's own modules under CPU autocast(bf16) by
tests/test_oracle_golden.py::test_bf16_policy_is_close_to_fp32_and_to_cpu_autocast) and its fp32 run.  With only 3 coarse steps per
window (dt ~ 0.33) that distance is 2.7e-2 .. 3.0e-2 per frame - twice the 30-step figure - and the GPU test holds the HIP path to
1.15 x it + 2e-3, frame by frame, against the fp32 oracle it recomputes at run time.
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import denoiser_oracle as O  # noqa: E402
from oracle import windows_oracle as WO  # noqa: E402

HP = dict(in_channels=64, num_layers=21, num_attention_heads=8, width=1024, mlp_ratio=4.0, cross_attention_dim=1024,
          inflated_layers=tuple(range(21)))
T, N, D, S, STEPS = 32, 47, 64, 257, 3


def case():
    cfg = O.OracleConfig(**HP)
    sd = O.synthetic_state_dict(cfg, seed=2)
    g = torch.Generator().manual_seed(32)
    ts = torch.arange(T, dtype=torch.float32)
    context = torch.randn((T, S, 1024), generator=g)
    anchor = torch.randn((1, N, D), generator=g)
    return cfg, sd, ts, context, anchor


if __name__ == "__main__":
    cfg, sd, ts, context, anchor = case()
    out = {}
    for prec in ("fp32", "bf16"):
        bank = WO.ListLatentBank((N, D))
        bank.update(ts[0:1], anchor)
        WO.generate_3d_latents(sd, cfg, ts, context, bank, 0, 16, 15, (N, D), STEPS, seed=44, precision=prec)
        out[prec] = bank.get_ordered()[0]
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    per_frame = [0.0] + [rel(out["bf16"][i], out["fp32"][i]) for i in range(1, T)]
    path = os.path.join(ROOT, "tests", "golden", "ar_configs2_yardstick.json")
    with open(path, "w") as f:
        json.dump({"what": "bf16-policy oracle vs fp32 oracle, rel-L2 per frame (frame 0 = the anchor, untouched)", "frames": T, "tokens": N,
                   "steps": STEPS, "weights_checksum": O.state_dict_checksum(sd),
                   "fp32_checksum": float(out["fp32"].double().sum()), "bf16_policy_vs_fp32_per_frame": per_frame}, f, indent=1)
    print("wrote", path, "max", max(per_frame))
