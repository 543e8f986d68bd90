"""Seam S3: HipAttentionProcessor is a drop-in for the reference's AttentionProcessor.

CPU: when /root/reference is importable (build container) the `__call__` signature is the reference's, so diffusers'
`Attention.forward` filters kwargs identically.  GPU: driven through the (shim) diffusers `Attention` module exactly as
block.py:137-149 drives the reference processor, compared with the fp32 CPU oracle of the same call
(oracle/denoiser_oracle.py self_attention / cross_attention, themselves pinned to the reference fixtures).
Tolerance: rel-L2 <= 2e-2 (bf16 operands, fp32 accumulation / softmax; the model-level tolerance)."""
import inspect
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "oracle", "diffusers_shim")
REF = "/root/reference"


def _shim_attention():
    if SHIM not in sys.path:
        sys.path.insert(0, SHIM)
    from diffusers.models.attention_processor import Attention
    return Attention


def test_call_signature_is_the_references():
    from actionmesh_amd.attention_processor import HipAttentionProcessor
    ours = inspect.signature(HipAttentionProcessor.__call__)
    names = list(ours.parameters)
    assert names == ["self", "attn", "hidden_states", "encoder_hidden_states", "attention_mask", "temb",
                     "inflate_self_attention", "freqs_rot", "n_frames"]
    defaults = {k: v.default for k, v in ours.parameters.items() if v.default is not inspect._empty}
    assert defaults == dict(encoder_hidden_states=None, attention_mask=None, temb=None, inflate_self_attention=False,
                            freqs_rot=None, n_frames=None)
    if os.path.isdir(REF):
        _shim_attention()
        sys.path.insert(0, REF)
        from actionmesh.model.utils.attention_processor import AttentionProcessor
        ref = inspect.signature(AttentionProcessor.__call__)
        assert list(ref.parameters) == names
        assert {k: v.default for k, v in ref.parameters.items() if v.default is not inspect._empty} == defaults


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from actionmesh_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


@pytest.mark.gpu
@pytest.mark.parametrize("inflate", [True, False])
def test_self_attention_through_the_diffusers_module(dev, inflate):
    from actionmesh_amd.attention_processor import HipAttentionProcessor
    from oracle import denoiser_oracle as O
    Attention = _shim_attention()
    torch.manual_seed(0)
    B, T, Ltok, H = 2, 4, 300, 2
    C = H * 128
    attn = Attention(query_dim=C, heads=H, dim_head=128, bias=False, qk_norm="rms_norm", eps=1e-6,
                     processor=HipAttentionProcessor())
    with torch.no_grad():
        attn.norm_q.weight.mul_(1.0 + 0.1 * torch.randn(128)); attn.norm_k.weight.mul_(1.0 + 0.1 * torch.randn(128))
    z = torch.randn(B * T, Ltok, C)
    cos, sin = O.rope_tables(torch.arange(T, dtype=torch.float32)[None].expand(B, T), 128)        # (B*T, 128)
    freqs = (cos[:, None, :].expand(B * T, Ltok, 128).contiguous(), sin[:, None, :].expand(B * T, Ltok, 128).contiguous())
    sd = {"p.to_q.weight": attn.to_q.weight.detach(), "p.to_k.weight": attn.to_k.weight.detach(),
          "p.to_v.weight": attn.to_v.weight.detach(), "p.norm_q.weight": attn.norm_q.weight.detach(),
          "p.norm_k.weight": attn.norm_k.weight.detach(), "p.to_out.0.weight": attn.to_out[0].weight.detach(),
          "p.to_out.0.bias": attn.to_out[0].bias.detach()}
    want = O.self_attention(z, sd, "p.", H, T, inflate, cos, sin, O.Precision("fp32"))
    attn = attn.to(dev)
    got = attn(z.to(dev), n_frames=T, inflate_self_attention=inflate,
               freqs_rot=(freqs[0].to(dev), freqs[1].to(dev)), some_kwarg_the_processor_does_not_take=1)
    torch.cuda.synchronize()
    assert got.shape == z.shape and got.dtype == z.dtype
    r = rel(got.cpu(), want)
    print(f"S3 self-attention (inflate={inflate}): rel-L2 vs fp32 oracle {r:.3e}")
    assert r < 2e-2
    # second call reuses the packed weights; an in-place parameter update refreshes them
    got2 = attn(z.to(dev), n_frames=T, inflate_self_attention=inflate, freqs_rot=(freqs[0].to(dev), freqs[1].to(dev)))
    assert torch.equal(got2, got)
    with torch.no_grad():
        attn.to_out[0].bias.add_(1.0)
    got3 = attn(z.to(dev), n_frames=T, inflate_self_attention=inflate, freqs_rot=(freqs[0].to(dev), freqs[1].to(dev)))
    assert rel(got3.cpu(), want + 1.0) < 2e-2


@pytest.mark.gpu
def test_cross_attention_through_the_diffusers_module(dev):
    from actionmesh_amd.attention_processor import HipAttentionProcessor
    from oracle import denoiser_oracle as O
    Attention = _shim_attention()
    torch.manual_seed(1)
    BT, Ltok, H, S, Dc = 6, 130, 2, 37, 192
    C = H * 128
    attn = Attention(query_dim=C, cross_attention_dim=Dc, heads=H, dim_head=128, bias=False, qk_norm="rms_norm",
                     eps=1e-6, processor=HipAttentionProcessor())
    z, ctx = torch.randn(BT, Ltok, C), torch.randn(BT, S, Dc)
    sd = {"p.to_q.weight": attn.to_q.weight.detach(), "p.to_k.weight": attn.to_k.weight.detach(),
          "p.to_v.weight": attn.to_v.weight.detach(), "p.norm_q.weight": attn.norm_q.weight.detach(),
          "p.norm_k.weight": attn.norm_k.weight.detach(), "p.to_out.0.weight": attn.to_out[0].weight.detach(),
          "p.to_out.0.bias": attn.to_out[0].bias.detach()}
    want = O.cross_attention(z, ctx, sd, "p.", H, O.Precision("fp32"))
    attn = attn.to(dev)
    got = attn(z.to(dev), encoder_hidden_states=ctx.to(dev))
    torch.cuda.synchronize()
    r = rel(got.cpu(), want)
    print(f"S3 cross-attention: rel-L2 vs fp32 oracle {r:.3e}")
    assert r < 2e-2
    # zero context + bias-free to_k / to_v => the output is exactly the out-projection bias
    got0 = attn(z.to(dev), encoder_hidden_states=torch.zeros_like(ctx).to(dev))
    b = attn.to_out[0].bias.detach().to(torch.bfloat16).float()
    assert torch.allclose(got0, b.expand_as(got0), atol=1e-6)
    with pytest.raises(RuntimeError):
        attn.cpu()(z, encoder_hidden_states=ctx)
