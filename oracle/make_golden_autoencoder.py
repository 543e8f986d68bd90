"""Generate tests/golden/ae_tiny.npz from the REFERENCE's own unmodified ActionMeshAutoencoder (Stage II).

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (needs /root/reference):

    python oracle/make_golden_autoencoder.py

The reference module is imported from /root/reference with the un-vendored `diffusers` dependency supplied by
oracle/diffusers_shim; weights = oracle.autoencoder_oracle.synthetic_state_dict (no pretrained weights offline).
Stored: the inputs, the displacement the reference returns (fp32, CPU) and a weight checksum.
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
}

for name, (kw, B, T, N, V, T_out) in CASES.items():
    cfg = AO.AEConfig(**kw)
    sd = AO.synthetic_state_dict(cfg, seed=0)
    m = ActionMeshAutoencoder(verbose=False, **kw)
    assert set(m.state_dict().keys()) == set(sd.keys())
    assert [k for k, _ in AO.state_dict_spec(cfg)] == list(m.state_dict().keys()), "state-dict order"
    m.load_state_dict(sd)
    m.eval()
    g = torch.Generator().manual_seed(17)
    latent = torch.randn((B, T, N, cfg.latent_channels), generator=g)
    framestep = torch.tensor([[3.0, 0.0, 2.0, 1.0][:T]]).repeat(B, 1)
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
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"{name}.npz"),
                        latent=latent.numpy(), framestep=framestep.numpy(), source_alpha=source_alpha.numpy(),
                        target_alphas=target_alphas.numpy(), query=query.numpy(), displacement_fp32=disp.numpy(),
                        weights_checksum=np.float64(AO.state_dict_checksum(sd)),
                        config=np.array([kw["width"], kw["num_layers"], kw["num_attention_heads"], kw["latent_channels"]]))
    print("wrote", name, os.path.getsize(os.path.join(ROOT, "tests", "golden", f"{name}.npz")), "bytes")
