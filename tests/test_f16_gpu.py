"""The float16 build of the library (libactionmesh_amd_f16.so: the same sources with -DAM_F16; the reference CLI's `--dtype float16`,
inference/video_to_animated_mesh.py:153,222; pipeline.py:671) through the C-ABI: kernels against fp32 statements of the same ops, the
denoiser against the reference-generated fixtures, and the autocast-driven dtype selection of HipDenoiser.  IEEE half carries 10
mantissa bits (bf16: 7), so every tolerance here is TIGHTER than its bf16 twin; what float16 lacks is range, which is why the 4x64
attention runs its exact (running-max) form in this build."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from actionmesh_amd import _lib
    _lib.lib("f16")
    return torch.device("cuda:0")


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("M,N,K,kw", [(4096 + 32, 1024, 1024, dict(bias=True, res=True)), (2048, 4096, 1024, dict(bias=True, gelu=True)),
                                      (1024 * 33, 1024, 2048, dict()), (300, 192, 256, dict(bias=True))])
def test_gemm_f16(dev, M, N, K, kw):
    from actionmesh_amd import ops
    g = torch.Generator().manual_seed(M + N)
    a = (torch.randn(M, K, generator=g)).half(); w = (torch.randn(N, K, generator=g) * K ** -0.5).half()
    bias = torch.randn(N, generator=g) if kw.get("bias") else None
    res = torch.randn(M, N, generator=g).half() if kw.get("res") else None
    out = ops.gemm(a.to(dev), w.to(dev), bias=None if bias is None else bias.to(dev), residual=None if res is None else res.to(dev),
                   gelu=bool(kw.get("gelu")))
    assert out.dtype == torch.float16
    ref = a.double() @ w.double().T
    if bias is not None:
        ref = ref + bias.double()                      # the C-ABI takes the bias in fp32 as given (the model rounds its biases when it loads them)
    ref = ref.float().half().double()                   # the linear's output is rounded ...
    lin = ref.clone()
    if kw.get("gelu"):
        ref = F.gelu(ref.float()).half().double()
    if res is not None:
        ref = (ref + res.double()).float().half().double()
    err = (out.double().cpu() - ref).abs()
    # one f16 rounding of an fp32-accumulated sum vs the fp64 statement rounded at the same points: within 2 f16 ulp of the VALUE, plus
    # the fp32 accumulation noise of a K-term dot product of O(1) terms (an absolute floor: outputs near zero have tiny ulps)
    # - of the value OR of the rounded linear output it was formed from: where `linear + residual` cancels, a linear output that sat on a
    # rounding tie in fp64 (-1.0649415 between two halves, tools/diag/f16_gemm_worst.py) lands one f16 step of ITS magnitude away
    bound = 2.0 * torch.maximum(ref.abs(), lin.abs()) * 2.0 ** -10 + 3e-4
    worst = float((err / bound).max())
    print(f"f16 gemm {M}x{N}x{K} {kw}: worst error / bound {worst:.2f}, rel-L2 {rel(out.cpu(), ref):.2e}")
    assert worst <= 1.0 and rel(out.cpu(), ref) < 5e-4


@pytest.mark.parametrize("sq,sk,nchunks", [(300, 300, 1), (2320, 2320, 1), (520, 1100, 3)])
def test_attention_f16(dev, sq, sk, nchunks):
    """Short streams run the 8-wave kernel, long ones the 4x64 kernel in its EXACT form (the float16 build never launches the lazy one)."""
    from actionmesh_amd import ops
    g = torch.Generator().manual_seed(sq + sk)
    nseq, H = 1, 2
    q = torch.randn(nseq, H, sq, 128, generator=g).half(); k = torch.randn(nseq, H, nchunks * sk, 128, generator=g).half()
    v = torch.randn(nseq, H, nchunks * sk, 128, generator=g).half()
    sq_pad, sk_pad = ops.round_up(sq, 256), ops.round_up(sk, 64)
    Q = torch.zeros(nseq, H, sq_pad, 128, dtype=torch.float16); Q[:, :, :sq] = q
    K = torch.zeros(nchunks, nseq, H, sk_pad, 128, dtype=torch.float16)
    Vt = torch.zeros(nchunks, nseq, H, 128, sk_pad, dtype=torch.float16)
    idx = ops.perm16_index(sk_pad)
    for c in range(nchunks):
        K[c, :, :, :sk] = k[:, :, c * sk:(c + 1) * sk]
        vt = torch.zeros(nseq, H, 128, sk_pad, dtype=torch.float16)
        vt[..., :sk] = v[:, :, c * sk:(c + 1) * sk].transpose(-1, -2)
        Vt[c] = vt[..., idx]
    f0 = ops.attention_fallback_count()
    out = ops.attention(Q.to(dev), K.to(dev), Vt.to(dev), sq, sk, nchunks=nchunks)
    torch.cuda.synchronize()
    assert out.dtype == torch.float16
    ref = F.scaled_dot_product_attention(q.float(), k.float(), v.float()).transpose(1, 2).reshape(nseq * sq, H * 128)
    r = rel(out.float().cpu(), ref)
    print(f"f16 attention sq={sq} sk={sk}x{nchunks}: rel-L2 vs fp32 SDPA {r:.3e}")
    assert torch.isfinite(out.float()).all() and r < 1.5e-3            # bf16 kernels: 2.8e-3 (tests/test_kernels_gpu.py)
    assert torch.equal(out, ops.attention(Q.to(dev), K.to(dev), Vt.to(dev), sq, sk, nchunks=nchunks)), "run-to-run bits"


def test_layernorm_and_flow_step_f16(dev):
    from actionmesh_amd import ops
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(1000, 1024, generator=g) * 3 + 1).half()
    w = torch.randn(1024, generator=g); b = torch.randn(1024, generator=g)
    y = ops.layernorm(x.to(dev), w.to(dev), b.to(dev))
    ref = F.layer_norm(x.float(), (1024,), w, b, 1e-5)
    assert y.dtype == torch.float16 and float((y.float().cpu() - ref).abs().max()) <= 2.0 ** -9 * float(ref.abs().max())
    # CFG + Euler in the 16-bit type, as autocast evaluates it: v0 + 7.5 (v1 - v0) rounded at every op, dt * v rounded, fp32 add
    T, N, D = 3, 17, 64
    v = torch.randn(2, T, N, D, generator=g).half()
    lat = torch.randn(T, N, D, generator=g)
    want = lat.clone()
    d = (v[1] - v[0])                                   # half arithmetic
    vv = v[0] + (7.5 * d.float()).half()
    upd = (torch.tensor(0.0371, dtype=torch.float32) * vv.float()).half().float()
    want[1:] = lat[1:] + upd[1:]
    got = lat.clone().to(dev)
    ops.flow_step(v.to(dev), got, [7.5], 0.0371, True, [False, True, True])
    assert torch.allclose(got.cpu(), want, rtol=0, atol=2e-3) and torch.equal(got.cpu()[0], lat[0])


def test_denoiser_float16_tiny_and_autocast_selection(dev, golden_dir):
    """HipDenoiser(dtype="float16") on the reference-generated toy fixture, and the reference pipeline's way of asking for it: no dtype
    argument, the sampler called inside torch.autocast("cuda", dtype=torch.float16) (pipeline.py:671)."""
    from actionmesh_amd import ClassifierFreeGuidance, HipDenoiser, HipSchedulerFlow
    from oracle import denoiser_oracle as O
    kw = dict(in_channels=64, num_layers=5, num_attention_heads=2, width=256, mlp_ratio=4.0, cross_attention_dim=64,
              inflated_layers=(0, 1, 2, 3, 4))
    g = np.load(os.path.join(golden_dir, "tiny_inflated.npz"))
    sd = O.synthetic_state_dict(O.OracleConfig(**kw), seed=0)
    t = {k: torch.from_numpy(g[k]) for k in ("init_latent", "context", "mask", "framestep")}
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    x_in, c_in, m_in, f_in = cfgd.cfg_at_inference(t["init_latent"], t["context"], t["mask"], t["framestep"])
    tt = torch.tensor([float(g["fwd_t"])]).expand(2)
    ref = torch.from_numpy(g["fwd_velocity_fp32"])
    res = {}
    for name, dtype in (("float16", "float16"), ("bfloat16", None)):
        model = HipDenoiser(num_tokens_nominal=48, temporal_context_size=4, dtype=dtype, **kw)
        model.load_state_dict(sd)
        model.to(dev).eval()
        v, _ = model.forward(x_in.to(dev), c_in.to(dev), f_in.to(dev), tt.to(dev), m_in.to(dev), None)
        torch.cuda.synchronize()
        assert v.dtype == (torch.float16 if dtype else torch.bfloat16) and model._engine.kind == ("f16" if dtype else "bf16")
        res[name] = rel(v.float().cpu(), ref)
        model.cpu()
    print(f"tiny denoiser forward vs the reference's fp32: float16 {res['float16']:.3e}, bfloat16 {res['bfloat16']:.3e}")
    assert res["float16"] < 2.5e-3 and res["float16"] < 0.4 * res["bfloat16"]
    # autocast-driven: the same object switches engines with the calling region
    model = HipDenoiser(num_tokens_nominal=48, temporal_context_size=4, **kw)
    model.load_state_dict(sd)
    model.to(dev).eval()
    sched = HipSchedulerFlow(num_inference_steps=int(g["steps"]), shift=3.0, is_additive=True)
    ref_loop = torch.from_numpy(g["loop_latents_fp32"][-1])
    with torch.autocast("cuda", dtype=torch.float16):
        out16 = sched.denoise(model, cfgd, init_latent=t["init_latent"].clone().to(dev), context=t["context"].to(dev), device=dev,
                              mask=t["mask"].to(dev), framestep=t["framestep"].to(dev))
        assert model._engine.kind == "f16"
    with torch.autocast("cuda", dtype=torch.bfloat16):
        outbf = sched.denoise(model, cfgd, init_latent=t["init_latent"].clone().to(dev), context=t["context"].to(dev), device=dev,
                              mask=t["mask"].to(dev), framestep=t["framestep"].to(dev))
        assert model._engine.kind == "bf16"
    r16, rbf = rel(out16.cpu(), ref_loop), rel(outbf.cpu(), ref_loop)
    print(f"tiny sampler loop under autocast: float16 {r16:.3e}, bfloat16 {rbf:.3e} vs the reference's fp32 latents")
    assert r16 < 5e-3 and r16 < 0.5 * rbf
