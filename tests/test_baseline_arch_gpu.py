"""Model-level parity at the BASELINE architectures (GPU): 21 layers (skip depth 10), head_dim 128, Dc 1024, S 257,
width 1024 / 8 heads (the headline "16f x 4096tok x 1024-dim" architecture) and width 2048 / 16 heads (the shipped
one), many sampler steps, through the C-ABI, against fixtures the reference's own unmodified modules produced
(oracle/make_golden_baseline.py -> tests/golden/arch_*.npz).  Weights and inputs are regenerated from seeded CPU
generators and pinned by the checksums stored in the fixtures.

Stated tolerance (per-step latents on the stored token subset, rel-L2 vs the reference's fp32 run): the fixtures carry the
reference's OWN reduced-precision curve (the same modules under CPU autocast(bf16) vs their fp32 run,
`ref_autocast_curve`), and the HIP path must stay within  1.15 x that curve + 2e-3  at every step (measured on MI355X,
profiles/r02a_parity_arch_*.json: the HIP curve tracks it within 2 % - headline architecture 5.3e-4 after step 1 growing
to 1.38e-2 after 30 steps, reference autocast 1.40e-2; nominal 1.75e-2 after 10 steps vs 1.76e-2; one forward 9.8e-3 vs
9.95e-3), under the absolute cap  min(2e-2 + 2.5e-3 * step, 8e-2).
Peaky / spiky weights (qk-norm gains x4 / x7: scores ~ N(0, 16^2), softmax close to one-hot) are chaotic in ANY reduced
precision (the reference under autocast(bf16) is 0.51 rel-L2 from its own fp32 forward on these cases, and so is the HIP
path), so those cases assert what is well defined - see test_peaky_attention_inside_the_full_model.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from actionmesh_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def _case(name, golden_dir, dev, **model_kw):
    from actionmesh_amd import HipDenoiser
    from oracle.denoiser_oracle import state_dict_checksum
    from oracle.make_golden_baseline import baseline_case_inputs, tensor_checksum
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    kw, cfg, sd, inp, steps = baseline_case_inputs(name)
    assert state_dict_checksum(sd) == pytest.approx(float(g["weights_checksum"]), rel=1e-12)
    got = [tensor_checksum(inp[k]) for k in ("init_latent", "context", "mask", "framestep")]
    assert np.allclose(got, g["inputs_checksum"], rtol=1e-12)
    model = HipDenoiser(num_tokens_nominal=inp["init_latent"].shape[2], temporal_context_size=inp["init_latent"].shape[1],
                        **kw, **model_kw)
    model.load_state_dict(sd)
    model.to(dev).eval()
    return g, cfg, sd, model, inp, steps


def _forward(model, inp, t, dev):
    from actionmesh_amd import ClassifierFreeGuidance
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    x_in, c_in, m_in, f_in = cfgd.cfg_at_inference(inp["init_latent"], inp["context"], inp["mask"], inp["framestep"])
    tt = torch.tensor([t]).expand(2)
    v, _ = model.forward(x_in.to(dev), c_in.to(dev), f_in.to(dev), tt.to(dev), m_in.to(dev), None)
    torch.cuda.synchronize()
    return v.float().cpu()


def _record(name, payload):
    """Curves go to gpurun_out/ (scratch) so the numbers quoted in DESIGN.md can be regenerated."""
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, f"parity_{name}.json"), "w") as f:
            json.dump(payload, f, indent=1)
    except OSError:
        pass


def tol_plain(step):
    return min(2e-2 + 2.5e-3 * step, 8e-2)


# Stated tolerance of the fp8 (e4m3) self-attention variant at the BASELINE architectures, relative to the reference's OWN
# reduced-precision curve like the bf16 one:  rel-L2 <= K8 x ref_autocast_curve[step] + EPS8.
# MEASURED on MI355X (profiles/r04a_parity_*_fp8.json): the fp8 curve sits 3-7 % above the reference's own autocast(bf16) curve at
# every step of every case - arch_headline 30 steps 1.44e-2 vs 1.40e-2, arch_headline_50 1.34e-2 vs 1.30e-2 after step 50,
# arch_nominal 1.82e-2 vs 1.76e-2, one forward at the FULL headline / nominal shapes 1.020e-2 / 1.026e-2 vs 0.987e-2 / 0.993e-2 -
# i.e. the e4m3 noise of the self-attention is diluted by the bf16 residual stream to a few per cent of the bf16 rounding error
# the reference itself carries, through 21 layers, depth-10 skips and 50 sampler steps alike (worst ratio 1.065, early steps of
# arch_nominal).  So the fp8 path is held to the SAME statement as the bf16 path: 1.15 x the reference's curve + 2e-3.
K8, EPS8 = 1.15, 2e-3


def tol_curve(dtype, ref, step=None):
    if dtype.startswith("fp8"):          # fp8_fast (exponent-field probabilities): held to the same statement, measured in profiles/r04h_*
        return K8 * ref + EPS8
    return 1.15 * ref + 2e-3


def tol_cap(dtype, step):
    return tol_plain(step) * (K8 / 1.15 if dtype.startswith("fp8") else 1.0)


def _tag(name, dtype):
    return name if dtype == "bf16" else f"{name}_{dtype}"


@pytest.mark.parametrize("dtype", ["bf16", "fp8", "fp8_fast"])
@pytest.mark.parametrize("name", ["full_headline", "full_nominal"])
def test_full_shape_forward(dev, golden_dir, name, dtype):
    """ONE forward at the FULL benchmarked shapes (VERDICT r02 missing #2) against the reference's own modules: the headline workload
    of bench.py (16 frames x 4096 tokens, width 1024: 65 552-token inflated sequences, 1025 key tiles) and the shipped architecture
    at its shipped size (16 x 2048, width 2048).  The fixture (oracle/make_golden_baseline.py make_full; ~30 min of host time per
    fp32 forward) keeps every 64th token of the fp32 velocity and the reference's own autocast(bf16) distance.  Stated tolerance:
    rel-L2 <= 1.15 x that distance + 2e-3, and <= 2e-2."""
    path = os.path.join(golden_dir, f"{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{name}.npz not generated (oracle/make_golden_baseline.py {name})")
    g, cfg, sd, model, inp, _ = _case(name, golden_dir, dev, attn_dtype=dtype)
    stride = int(g["token_stride"])
    v = _forward(model, inp, float(g["fwd_t"]), dev)
    n8, n16 = model._engine.attention_counters()
    assert (n8 > 0 and n16 == 0) if dtype.startswith("fp8") else (n8 == 0 and n16 > 0), (dtype, n8, n16)      # the arithmetic that really ran
    model.cpu()
    ref = torch.from_numpy(g["fwd_velocity_fp32_sub"])
    got = v[:, :, ::stride]
    r = rel(got, ref)
    ref_ac = float(g["fwd_ref_autocast_vs_fp32"]) if "fwd_ref_autocast_vs_fp32" in g.files else None
    rms = float(g["fwd_velocity_rms"])
    print(f"{name} [{dtype}]: full-shape forward rel-L2 vs reference fp32 {r:.3e} on {ref.numel()} sampled values (reference autocast vs its fp32: "
          f"{ref_ac}); velocity rms {float(v.double().pow(2).mean().sqrt()):.4f} (reference {rms:.4f}); max abs {float((got - ref).abs().max()):.3e}")
    _record(_tag(name, dtype), dict(forward=r, ref_autocast=ref_ac, max_abs=float((got - ref).abs().max()), rms=rms, attn_dtype=dtype))
    assert torch.isfinite(v).all() and r < tol_cap(dtype, 0)
    if ref_ac is not None:
        assert r < tol_curve(dtype, ref_ac), (r, ref_ac)
    assert abs(float(v.double().pow(2).mean().sqrt()) - rms) < 2e-2 * rms


@pytest.mark.parametrize("dtype", ["bf16", "fp8", "fp8_fast"])
@pytest.mark.parametrize("name", ["arch_headline", "arch_nominal", "arch_headline_50"])
def test_baseline_arch_forward_and_per_step_latents(dev, golden_dir, name, dtype):
    """bf16: the product path.  fp8: `attn_dtype="fp8"` - BASELINE configs[4]'s arithmetic - through the same 21-layer / depth-10-skip
    architectures and the same 10 / 30 / 50-step sampler loops, against the same reference fixtures; the e4m3 noise of the
    self-attention is diluted by the bf16 residual stream (DESIGN.md section 2 holds the measured curves)."""
    from actionmesh_amd import ClassifierFreeGuidance, HipSchedulerFlow
    g, cfg, sd, model, inp, steps = _case(name, golden_dir, dev, attn_dtype=dtype)
    stride = int(g["token_stride"])
    v = _forward(model, inp, float(g["fwd_t"]), dev)
    r_fwd = rel(v, torch.from_numpy(g["fwd_velocity_fp32"]))
    print(f"{name} [{dtype}]: forward rel-L2 vs reference fp32 {r_fwd:.3e} (reference autocast vs fp32: {float(g['fwd_ref_autocast_vs_fp32']):.3e})")
    assert torch.isfinite(v).all() and r_fwd < tol_cap(dtype, 0)
    assert r_fwd < tol_curve(dtype, float(g["fwd_ref_autocast_vs_fp32"]))

    sched = HipSchedulerFlow(num_inference_steps=steps, shift=3.0, is_additive=True)
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    ref = torch.from_numpy(g["loop_latents_sub_fp32"])
    ref_curve = g["ref_autocast_curve"]
    init = inp["init_latent"].clone().to(dev)
    curve, last = [], None
    for i, (lat, _t) in enumerate(sched._flow_sample(model, cfgd, init, inp["context"].to(dev), device=dev,
                                                     mask=inp["mask"].to(dev), framestep=inp["framestep"].to(dev))):
        sub = lat[:, :, ::stride].cpu()
        curve.append(rel(sub, ref[i]))
        assert torch.equal(sub[0, 0], inp["init_latent"][0, 0, ::stride]), "conditioning frame must stay untouched"
        last = lat
    assert len(curve) == steps
    final = rel(last.cpu(), torch.from_numpy(g["loop_final_fp32"]))
    print(f"{name} [{dtype}]: per-step latents rel-L2 vs reference fp32: " + " ".join(f"{c:.2e}" for c in curve))
    print(f"{name}: reference autocast(bf16) vs its fp32:       " + " ".join(f"{c:.2e}" for c in ref_curve))
    print(f"{name} [{dtype}]: final full latents rel-L2 {final:.3e}")
    n8, n16 = model._engine.attention_counters()
    assert (n8 > 0 and n16 == 0) if dtype.startswith("fp8") else (n8 == 0 and n16 > 0), (dtype, n8, n16)
    worst = max(c / float(ref_curve[i]) for i, c in enumerate(curve) if i < len(ref_curve) and i >= 2)
    _record(_tag(name, dtype), dict(forward=r_fwd, curve=curve, ref_autocast_curve=[float(c) for c in ref_curve], final=final,
                                    attn_dtype=dtype, worst_ratio_to_ref_autocast_from_step_3=worst))
    for i, c in enumerate(curve):
        assert c < tol_cap(dtype, i), (i, c, tol_cap(dtype, i))
        if i < len(ref_curve):       # measured (r02a): the bf16 HIP curve tracks the reference's own autocast curve within 2 %
            assert c < tol_curve(dtype, float(ref_curve[i])), (i, c, float(ref_curve[i]))
    assert final < tol_cap(dtype, steps - 1)


def test_float16_mode_against_the_reference(dev, golden_dir):
    """`--dtype float16` of the reference CLI (inference/video_to_animated_mesh.py:153,222): HipDenoiser(dtype="float16") - the float16
    build of the library, exact (running-max) 4x64 attention - at the headline architecture (21 layers, depth-10 skips) for 30 sampler steps,
    against the fp32 run of the reference's own modules; the tolerance is RELATIVE to the reference's own float16 curve (the same modules
    under torch.autocast("cpu", dtype=float16), oracle/make_golden_f16.py -> tests/golden/arch_headline_f16.npz): 1.25 x that curve + 3e-4
    per step.  IEEE half carries 10 mantissa bits, so both curves sit ~8x below their bf16 twins."""
    from actionmesh_amd import ClassifierFreeGuidance, HipSchedulerFlow
    name = "arch_headline"
    path = os.path.join(golden_dir, f"{name}_f16.npz")
    if not os.path.exists(path):
        pytest.skip("arch_headline_f16.npz not generated (oracle/make_golden_f16.py)")
    g16 = np.load(path)
    g, cfg, sd, model, inp, steps = _case(name, golden_dir, dev, dtype="float16")
    assert np.allclose(g16["inputs_checksum"], g["inputs_checksum"], rtol=1e-12)
    stride = int(g["token_stride"])
    v = _forward(model, inp, float(g["fwd_t"]), dev)
    assert model._engine.kind == "f16"
    r_fwd, ref_fwd = rel(v, torch.from_numpy(g["fwd_velocity_fp32"])), float(g16["fwd_ref_autocast_f16_vs_fp32"])
    print(f"{name} [float16]: forward rel-L2 vs reference fp32 {r_fwd:.3e} (reference autocast(float16) vs its fp32: {ref_fwd:.3e})")
    assert torch.isfinite(v).all() and r_fwd < 1.25 * ref_fwd + 3e-4
    sched = HipSchedulerFlow(num_inference_steps=steps, shift=3.0, is_additive=True)
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    ref = torch.from_numpy(g["loop_latents_sub_fp32"])
    ref_curve = g16["ref_autocast_f16_curve"]
    init = inp["init_latent"].clone().to(dev)
    curve = []
    for i, (lat, _t) in enumerate(sched._flow_sample(model, cfgd, init, inp["context"].to(dev), device=dev,
                                                     mask=inp["mask"].to(dev), framestep=inp["framestep"].to(dev))):
        sub = lat[:, :, ::stride].cpu()
        curve.append(rel(sub, ref[i]))
        assert torch.equal(sub[0, 0], inp["init_latent"][0, 0, ::stride])
    print(f"{name} [float16]: per-step latents rel-L2 vs reference fp32: " + " ".join(f"{c:.2e}" for c in curve))
    print(f"{name}: reference autocast(float16) vs its fp32:          " + " ".join(f"{c:.2e}" for c in ref_curve))
    _record(name + "_float16", dict(forward=r_fwd, ref_forward=ref_fwd, curve=curve, ref_autocast_f16_curve=[float(c) for c in ref_curve]))
    for i, c in enumerate(curve):
        assert c < 1.25 * float(ref_curve[i]) + 3e-4, (i, c, float(ref_curve[i]))


@pytest.mark.parametrize("attn", ["fp8", "fp8_fast"])
def test_float16_mode_with_fp8_attention(dev, golden_dir, attn):
    """`python -m actionmesh_amd.cli --attn-dtype fp8 -- --dtype float16` (built, reachable from the CLI, never
    pinned): the float16 build of the library with the e4m3 self-attention, at the headline architecture for the 30-step sampler,
    against the reference's fp32 run.  The e4m3 noise of the attention (~3e-3 at the latents, measured under bf16 in round 4) is now the
    LARGEST term - float16's own rounding is 1.6e-3 - so the stated tolerance is the bf16 product statement, 1.15 x the reference's
    autocast(bf16) curve + 2e-3 per step: fp8 attention under float16 is no further from the reference than the bf16 default; and it
    must be at least as close as fp8 attention under bfloat16 was (1.44e-2 after 30 steps).  Measured values are printed."""
    from actionmesh_amd import ClassifierFreeGuidance, HipSchedulerFlow
    name = "arch_headline"
    g, cfg, sd, model, inp, steps = _case(name, golden_dir, dev, dtype="float16", attn_dtype=attn)
    stride = int(g["token_stride"])
    v = _forward(model, inp, float(g["fwd_t"]), dev)
    assert model._engine.kind == "f16"
    n8, n16 = model._engine.attention_counters()
    assert n8 > 0 and n16 == 0, (n8, n16)
    r_fwd = rel(v, torch.from_numpy(g["fwd_velocity_fp32"]))
    assert torch.isfinite(v).all() and r_fwd < tol_curve(attn, float(g["fwd_ref_autocast_vs_fp32"]))
    sched = HipSchedulerFlow(num_inference_steps=steps, shift=3.0, is_additive=True)
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    ref = torch.from_numpy(g["loop_latents_sub_fp32"])
    ref_curve = g["ref_autocast_curve"]
    curve = []
    for i, (lat, _t) in enumerate(sched._flow_sample(model, cfgd, inp["init_latent"].clone().to(dev), inp["context"].to(dev), device=dev,
                                                     mask=inp["mask"].to(dev), framestep=inp["framestep"].to(dev))):
        curve.append(rel(lat[:, :, ::stride].cpu(), ref[i]))
    print(f"{name} [float16 + {attn}]: forward {r_fwd:.3e}; per-step latents rel-L2 vs reference fp32: " + " ".join(f"{c:.2e}" for c in curve))
    _record(f"{name}_float16_{attn}", dict(forward=r_fwd, curve=curve, ref_autocast_bf16_curve=[float(c) for c in ref_curve]))
    for i, c in enumerate(curve):
        assert c < tol_curve(attn, float(ref_curve[i])), (i, c, float(ref_curve[i]))
    assert curve[-1] < 1.44e-2


@pytest.mark.parametrize("name", ["arch_headline_peaky", "arch_headline_spiky"])
def test_peaky_attention_inside_the_full_model(dev, golden_dir, name):
    """Trained qk-norm gains make attention peaky; the lazy re-base and the exact fallback of the product attention kernel must
    fire inside the 21-layer model.  With scores ~ N(0, 16^2) the 21-layer map is chaotic in bf16: the reference's own
    autocast(bf16) forward is 0.51 rel-L2 from its fp32 forward, and so is every bf16 execution (measured: lazy kernel 0.512,
    exact kernel 0.512, the two 0.48 apart from each other) - so the model-level assertions are: finite, the data-dependent
    branches fired, and the distance to fp32 is the reference's own reduced-precision distance, not more.  That the lazy and
    the exact kernel compute the same attention is asserted where it is well defined: on ONE peaky layer (no amplification)."""
    from actionmesh_amd import _lib
    import ctypes as C
    g, cfg, sd, model, inp, steps = _case(name, golden_dir, dev)
    lib = _lib.lib()

    def fallbacks():
        n = C.c_uint64()
        _lib.check(lib.am_attention_fallback_count(C.byref(n)), "am_attention_fallback_count")
        return n.value

    f0 = fallbacks()
    v_lazy = _forward(model, inp, float(g["fwd_t"]), dev)
    fired = fallbacks() - f0
    assert torch.isfinite(v_lazy).all()
    model.cpu()
    g2, _, _, exact, _, _ = _case(name, golden_dir, dev, attn_defer_log2=0)       # running row max, immediate re-base
    v_exact = _forward(exact, inp, float(g["fwd_t"]), dev)
    exact.cpu()
    ref = torch.from_numpy(g["fwd_velocity_fp32"])
    r_lazy, r_exact, r_pair = rel(v_lazy, ref), rel(v_exact, ref), rel(v_lazy, v_exact)
    ref_ac = float(g["fwd_ref_autocast_vs_fp32"])
    print(f"{name}: forward rel-L2 vs reference fp32: lazy {r_lazy:.3e}, exact {r_exact:.3e}; lazy vs exact {r_pair:.3e}; "
          f"reference autocast(bf16) vs its fp32 {ref_ac:.3e}; exact-fallback workgroups {fired}")
    assert fired > 0, "built to send workgroups through the exact fallback"
    assert r_lazy < 1.1 * ref_ac + 1e-2 and r_exact < 1.1 * ref_ac + 1e-2

    # one peaky layer: the same weights of block 0 (largest gain of the case), 1-layer model - lazy == exact to rounding
    from actionmesh_amd import HipDenoiser
    from oracle.make_golden_baseline import baseline_case_inputs
    kw, _, sd_full, _, _ = baseline_case_inputs(name)
    hot = 3 if name.endswith("spiky") else 0
    sd1 = {k.replace(f"blocks.{hot}.", "blocks.0."): v for k, v in sd_full.items()
           if not k.startswith("blocks.") or k.startswith(f"blocks.{hot}.")}
    kw1 = dict(kw, num_layers=1, inflated_layers=(0,))
    outs = {}
    for defer in (8, 0):
        m1 = HipDenoiser(num_tokens_nominal=256, temporal_context_size=8, attn_defer_log2=defer, **kw1)
        m1.load_state_dict(sd1)
        m1.to(dev).eval()
        f1 = fallbacks()
        outs[defer] = _forward(m1, inp, float(g["fwd_t"]), dev)
        fired1 = fallbacks() - f1
        m1.cpu()
        if defer == 8 and name.endswith("spiky"):
            assert fired1 > 0
    r1 = rel(outs[8], outs[0])
    print(f"{name}: one peaky layer, lazy vs exact rel-L2 {r1:.3e}")
    _record(name, dict(lazy=r_lazy, exact=r_exact, lazy_vs_exact=r_pair, ref_autocast=ref_ac, fallback_workgroups=fired,
                       one_layer_lazy_vs_exact=r1))
    assert r1 < 5e-3


@pytest.mark.parametrize("name", ["arch_headline_peaky", "arch_headline_spiky"])
def test_peaky_attention_fp8_forms(dev, golden_dir, name):
    """the e4m3 forms - `fp8` (exp2, rounded to e4m3) and `fp8_fast` (exponent-field probabilities, p = 2^n
    (1 + f)) - through the peaky / spiky fixtures (qk-norm gains x4 / x7: scores ~ N(0, 16^2), softmax close to one-hot).  The 21-layer
    map is chaotic there in ANY reduced precision: the reference's own autocast(bf16) forward is 0.51 / 0.52 rel-L2 from its fp32 one, the
    bf16 HIP path 0.51.  MEASURED on MI355X (round 5, profiles/r05b_parity_arch_headline_{peaky,spiky}_fp8*.json): both e4m3 forms sit at
    0.604 / 0.609 - 18 % further out than bf16 (three mantissa bits on one-hot-like probabilities cost more than they do on the smooth
    ones of the unit-gain fixtures, where fp8 is 3-7 % above the reference's curve) - and within 0.1 % of EACH OTHER.
    Stated: finite; each form <= 1.25 x the reference's own reduced-precision distance + 1e-2; and fp8_fast within 3 % of fp8 - the
    exponent-field probabilities change nothing the e4m3 rounding had not already changed, in the regime that stresses them most."""
    r = {}
    for dtype in ("fp8", "fp8_fast"):
        g, cfg, sd, model, inp, steps = _case(name, golden_dir, dev, attn_dtype=dtype)
        v = _forward(model, inp, float(g["fwd_t"]), dev)
        n8, n16 = model._engine.attention_counters()
        assert n8 > 0 and n16 == 0 and bool(torch.isfinite(v).all())
        model.cpu()
        r[dtype], ref_ac = rel(v, torch.from_numpy(g["fwd_velocity_fp32"])), float(g["fwd_ref_autocast_vs_fp32"])
        print(f"{name} [{dtype}]: forward rel-L2 vs reference fp32 {r[dtype]:.3e}; reference autocast(bf16) vs its fp32 {ref_ac:.3e}")
        _record(_tag(name, dtype), dict(forward=r[dtype], ref_autocast=ref_ac))
        assert r[dtype] < 1.25 * ref_ac + 1e-2
    assert abs(r["fp8_fast"] - r["fp8"]) <= 0.03 * r["fp8"], r


def test_peaky_loop_stays_finite_and_anchored(dev, golden_dir):
    from actionmesh_amd import ClassifierFreeGuidance, HipSchedulerFlow
    name = "arch_headline_peaky"
    g, cfg, sd, model, inp, steps = _case(name, golden_dir, dev)
    stride = int(g["token_stride"])
    sched = HipSchedulerFlow(num_inference_steps=steps, shift=3.0, is_additive=True)
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    ref = torch.from_numpy(g["loop_latents_sub_fp32"])
    init = inp["init_latent"].clone().to(dev)
    curve = []
    for i, (lat, _t) in enumerate(sched._flow_sample(model, cfgd, init, inp["context"].to(dev), device=dev,
                                                     mask=inp["mask"].to(dev), framestep=inp["framestep"].to(dev))):
        sub = lat[:, :, ::stride].cpu()
        assert torch.isfinite(sub).all()
        assert torch.equal(sub[0, 0], inp["init_latent"][0, 0, ::stride])
        curve.append(rel(sub, ref[i]))
    ref_curve = [float(c) for c in g["ref_autocast_curve"]]
    print(f"{name}: per-step latents rel-L2 vs reference fp32: " + " ".join(f"{c:.2e}" for c in curve))
    print(f"{name}: reference autocast(bf16) vs its fp32:       " + " ".join(f"{c:.2e}" for c in ref_curve))
    _record(name + "_loop", dict(curve=curve, ref_autocast_curve=ref_curve))
    for i, c in enumerate(curve):
        assert c < 1.1 * ref_curve[i] + 1e-2, (i, c, ref_curve[i])
