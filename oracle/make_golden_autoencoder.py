"""Generate tests/golden/ae_tiny.npz / ae_arch.npz from the REFERENCE's own unmodified ActionMeshAutoencoder (Stage II).

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (needs /root/reference):

    python oracle/make_golden_autoencoder.py [ae_tiny ae_arch]

The reference module is imported from /root/reference with the un-vendored `diffusers` dependency supplied by
oracle/diffusers_shim; weights = oracle.autoencoder_oracle.synthetic_state_dict (no pretrained weights offline).
Stored: the inputs, the displacement the reference returns (fp32, CPU) and a weight checksum.

`ae_arch` (round 5, VERDICT r04 next #1b) is the SHIPPED architecture (actionmesh.yaml: width 1024, 8 heads of 128, 16 self-attention
blocks + 1 cross-attention block) at a reduced token count (T = 8 frames x N = 256 latent tokens = 2056-token sequences, V = 2000 query
vertices, 3 targets).  It also stores the reference's OWN reduced-precision distances: the same module under autocast(bfloat16) and
autocast(float16) (what pipeline.py:679 runs on a GPU with `--dtype bfloat16 | float16`) against its fp32 run.  The reference's
`torch.amp.autocast(device_type="cuda", enabled=False)` regions (temporal_autoencoder.py:236, 265: the query embedding and the
cross-attention block run in fp32) are honoured on the CPU by mapping device_type "cuda" -> "cpu" for the duration of those runs
(`_cuda_autocast_on_cpu`): without it the CPU stand-in would run the fp32 region in reduced precision and overstate the reference's error.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "diffusers_shim"))
sys.path.insert(0, "/root/reference")

from actionmesh.model.temporal_autoencoder import ActionMeshAutoencoder  # noqa: E402  (reference)

from oracle import autoencoder_oracle as AO  # noqa: E402

CASES = {
    # name: (config, B, T, N, V, T_out)
    "ae_tiny": (dict(width=256, num_layers=3, num_attention_heads=2, latent_channels=64), 1, 4, 48, 300, 3),
    "ae_arch": (dict(width=1024, num_layers=16, num_attention_heads=8, latent_channels=64), 1, 8, 256, 2000, 3),
}


class _cuda_autocast_on_cpu:
    """While active, torch.amp.autocast(device_type="cuda", ...) acts on the CPU autocast state (see the module docstring)."""
    def __enter__(self):
        self._orig = torch.amp.autocast
        orig = self._orig

        class _Mapped(orig):
            def __init__(self, device_type, *a, **k):
                super().__init__("cpu" if device_type == "cuda" else device_type, *a, **k)
        torch.amp.autocast = _Mapped
        return self

    def __exit__(self, *exc):
        torch.amp.autocast = self._orig


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


names = sys.argv[1:] or ["ae_tiny"]
for name in names:
    kw, B, T, N, V, T_out = CASES[name]
    cfg = AO.AEConfig(**kw)
    sd = AO.synthetic_state_dict(cfg, seed=0)
    m = ActionMeshAutoencoder(verbose=False, **kw)
    assert set(m.state_dict().keys()) == set(sd.keys())
    assert [k for k, _ in AO.state_dict_spec(cfg)] == list(m.state_dict().keys()), "state-dict order"
    m.load_state_dict(sd)
    m.eval()
    g = torch.Generator().manual_seed(17)
    latent = torch.randn((B, T, N, cfg.latent_channels), generator=g)
    framestep = (torch.tensor([[3.0, 0.0, 2.0, 1.0]]) if T <= 4 else torch.arange(T, dtype=torch.float32)[None]).repeat(B, 1)
    source_alpha = torch.tensor([0.25] * B)
    target_alphas = torch.linspace(0.0, 1.0, T_out)[None].repeat(B, 1)
    pts = torch.rand((B, V, 3), generator=g) * 1.6 - 0.8
    nrm = torch.nn.functional.normalize(torch.randn((B, V, 3), generator=g), dim=-1)
    query = torch.cat([pts, nrm], dim=-1)
    with torch.no_grad():
        disp = m(latent, framestep, source_alpha, target_alphas, query)
        mine = AO.autoencoder_forward(sd, cfg, latent, framestep, source_alpha, target_alphas, query)
    err = float((disp - mine).abs().max())
    print(f"{name}: reference displacement {tuple(disp.shape)} range [{float(disp.min()):.3f}, {float(disp.max()):.3f}]; "
          f"oracle restatement max abs diff {err:.2e}")
    assert err < 2e-5
    extra = {}
    if name != "ae_tiny":       # the reference's own reduced-precision distances (ae_tiny predates them and stays byte-identical)
        for tag, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
            with torch.no_grad(), _cuda_autocast_on_cpu(), torch.autocast("cpu", dtype=dt):
                dr = m(latent, framestep, source_alpha, target_alphas, query)
            assert torch.isfinite(dr.float()).all()
            extra[f"ref_autocast_{tag}_rel"] = np.float64(rel(dr.float(), disp))
            extra[f"ref_autocast_{tag}_maxabs"] = np.float64(float((dr.float() - disp).abs().max()))
            print(f"{name}: reference under autocast({tag}) vs its fp32: rel-L2 {extra[f'ref_autocast_{tag}_rel']:.3e}, "
                  f"max abs {extra[f'ref_autocast_{tag}_maxabs']:.3e}", flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"{name}.npz"),
                        latent=latent.numpy(), framestep=framestep.numpy(), source_alpha=source_alpha.numpy(),
                        target_alphas=target_alphas.numpy(), query=query.numpy(), displacement_fp32=disp.numpy(),
                        weights_checksum=np.float64(AO.state_dict_checksum(sd)),
                        config=np.array([kw["width"], kw["num_layers"], kw["num_attention_heads"], kw["latent_channels"]]), **extra)
    print("wrote", name, os.path.getsize(os.path.join(ROOT, "tests", "golden", f"{name}.npz")), "bytes")
