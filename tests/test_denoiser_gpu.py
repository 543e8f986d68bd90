"""End-to-end parity (GPU): HipDenoiser / HipSchedulerFlow through the C-ABI against
(1) the committed fixtures generated from the reference's own modules (tests/golden) and
(2) the CPU oracle, on the same seeded inputs.

Stated tolerance (floating-point path, bf16 storage / MFMA with fp32 accumulation and fp32
softmax/norm statistics, i.e. the reference's own cuda-autocast dtype flow):
  * one forward:           rel-L2(velocity) <= 2e-2 vs the fp32 reference, <= 1.5e-2 vs the
                           bf16-policy oracle
  * per-step latents:      rel-L2 <= 2e-2 vs the fp32 reference at every step
For scale: the reference itself under CPU autocast(bf16) sits at ~1e-2 of its fp32 run.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {
    "tiny_inflated": dict(in_channels=64, num_layers=5, num_attention_heads=2, width=256,
                          mlp_ratio=4.0, cross_attention_dim=64, inflated_layers=(0, 1, 2, 3, 4)),
    "tiny_mixed": dict(in_channels=64, num_layers=5, num_attention_heads=2, width=256,
                       mlp_ratio=4.0, cross_attention_dim=64, inflated_layers=(0, 1, 3, 4)),
}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from actionmesh_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


def _setup(name, golden_dir, dev):
    from actionmesh_amd import HipDenoiser
    from oracle import denoiser_oracle as O
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    cfg = O.OracleConfig(**CASES[name])
    sd = O.synthetic_state_dict(cfg, seed=0)
    assert O.state_dict_checksum(sd) == pytest.approx(float(g["weights_checksum"]), rel=1e-12)
    model = HipDenoiser(num_tokens_nominal=48, temporal_context_size=4, **CASES[name])
    model.load_state_dict(sd)
    model.to(dev).eval()
    t = {k: torch.from_numpy(g[k]) for k in ("init_latent", "context", "mask", "framestep")}
    return g, cfg, sd, model, t


@pytest.mark.parametrize("name", list(CASES))
def test_forward_matches_reference_fixture_and_oracle(dev, golden_dir, name):
    from actionmesh_amd import ClassifierFreeGuidance
    from oracle import denoiser_oracle as O
    g, cfg, sd, model, t = _setup(name, golden_dir, dev)
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    x_in, c_in, m_in, f_in = cfgd.cfg_at_inference(t["init_latent"], t["context"], t["mask"], t["framestep"])
    tt = torch.tensor([float(g["fwd_t"])]).expand(2)
    c_dev = c_in.to(dev)
    v, cache = model.forward(x_in.to(dev), c_dev, f_in.to(dev), tt.to(dev), m_in.to(dev), None)
    torch.cuda.synchronize()
    assert v.shape == x_in.shape and v.dtype == torch.bfloat16
    v = v.float().cpu()
    ref32 = torch.from_numpy(g["fwd_velocity_fp32"])
    vb = O.denoiser_forward(sd, cfg, x_in, c_in, f_in, tt, m_in, "bf16")
    r32, rbf = rel(v, ref32), rel(v, vb)
    print(f"{name}: forward rel-L2 vs reference fp32 {r32:.3e}, vs bf16-policy oracle {rbf:.3e}")
    assert r32 < 2e-2 and rbf < 1.5e-2
    # passing the returned cache back with the SAME context tensor reuses the bound window and gives the identical result;
    # another tensor (even with equal values) re-binds: the cache follows the context, not the cache object
    v2, cache2 = model.forward(x_in.to(dev), c_dev, f_in.to(dev), tt.to(dev), m_in.to(dev), cache)
    assert cache2 is cache and torch.equal(v2.float().cpu(), v)
    v3, cache3 = model.forward(x_in.to(dev), c_in.to(dev), f_in.to(dev), tt.to(dev), m_in.to(dev), cache)
    assert cache3 is not cache and torch.equal(v3.float().cpu(), v)


@pytest.mark.parametrize("name", list(CASES))
def test_loop_per_step_latents(dev, golden_dir, name):
    from actionmesh_amd import ClassifierFreeGuidance, HipSchedulerFlow
    g, cfg, sd, model, t = _setup(name, golden_dir, dev)
    steps = int(g["steps"])
    sched = HipSchedulerFlow(num_inference_steps=steps, shift=3.0, is_additive=True)
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    ref = torch.from_numpy(g["loop_latents_fp32"])
    init = t["init_latent"].clone().to(dev)
    got = []
    for lat, _t in sched._flow_sample(model, cfgd, init, t["context"].to(dev), device=dev,
                                      mask=t["mask"].to(dev), framestep=t["framestep"].to(dev)):
        got.append(lat.clone().cpu())
    assert len(got) == steps
    for i in range(steps):
        r = rel(got[i], ref[i])
        print(f"{name}: step {i} latents rel-L2 vs reference {r:.3e}")
        assert r < 2e-2
        assert torch.equal(got[i][0, 0], t["init_latent"][0, 0]), "conditioning frame must stay untouched"
    # denoise(): same result, callback contract, init_latent mutated in place like the reference
    calls = []
    init2 = t["init_latent"].clone().to(dev)
    out = sched.denoise(model, cfgd, init_latent=init2, context=t["context"].to(dev), device=dev,
                        mask=t["mask"].to(dev), framestep=t["framestep"].to(dev),
                        step_callback=lambda i, n: calls.append((i, n)))
    assert calls == [(i + 1, steps) for i in range(steps)]
    assert out.data_ptr() == init2.data_ptr()
    assert torch.equal(out.cpu(), got[-1])


def test_reference_style_loop_over_hipdenoiser(dev, golden_dir):
    """Seam S2: the reference's own sampler logic (restated with torch glue) driving HipDenoiser.forward
    with the opaque freqs_rot cache gives the same latents as HipSchedulerFlow."""
    from actionmesh_amd import ClassifierFreeGuidance, HipSchedulerFlow
    g, cfg, sd, model, t = _setup("tiny_inflated", golden_dir, dev)
    steps = int(g["steps"])
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    sched = HipSchedulerFlow(num_inference_steps=steps, shift=3.0, is_additive=True)
    ts, ds = sched.get_schedule()
    lat = t["init_latent"].clone().to(dev)
    ctx, mask, fs = t["context"].to(dev), t["mask"].to(dev), t["framestep"].to(dev)
    unobs = cfgd.get_unobserved_mask(mask)
    cache = None
    for i in range(steps):
        x_in, c_in, m_in, f_in = cfgd.cfg_at_inference(lat, ctx, mask, fs)
        dt = torch.tensor([float(ts[i])], device=dev).expand(2)
        v, cache = model.forward(x_in, c_in, f_in, dt, m_in, cache)
        v = cfgd.aggregate_cfg(v)
        # NOTE: the reference writes `distances[i] * output_pred` with a 0-dim fp32 DEVICE tensor,
        # which torch type promotion rounds to bf16 before the multiply (a cuda-autocast artefact
        # absent from its fp32 CPU path).  The HIP path keeps dt in fp32 (closer to the fp32
        # reference; DESIGN.md "dtype flow"), so the glue here does the same.
        flow = lat + (float(ds[i]) * v.float()).to(torch.bfloat16)
        lat[unobs] = flow[unobs]
    a = sched.denoise(model, cfgd, init_latent=t["init_latent"].clone().to(dev), context=ctx, device=dev,
                      mask=mask, framestep=fs)
    assert rel(lat.cpu(), a.cpu()) < 1e-3
    assert rel(a.cpu(), torch.from_numpy(g["loop_latents_fp32"][-1])) < 2e-2


def test_fails_loudly_without_hip_path(dev):
    from actionmesh_amd import ClassifierFreeGuidance, HipDenoiser, HipSchedulerFlow
    sched = HipSchedulerFlow(num_inference_steps=2)
    with pytest.raises(TypeError):
        sched.denoise(torch.nn.Linear(2, 2), ClassifierFreeGuidance(), torch.zeros(1, 2, 4, 64),
                      torch.zeros(1, 2, 3, 64))
    m = HipDenoiser(num_layers=1, num_attention_heads=2, width=256, cross_attention_dim=64)
    with pytest.raises(RuntimeError):
        m.to(dev).forward(torch.zeros(2, 2, 4, 64, device=dev), torch.zeros(2, 2, 3, 64, device=dev),
                          torch.zeros(2, 2, device=dev), torch.zeros(2, device=dev))


@pytest.mark.parametrize("name", list(CASES))
def test_two_rank_frame_sharding_emulated_on_one_gpu(dev, golden_dir, name):
    """The N>1 data path on real hardware without a second GPU: two engines (rank 0 / rank 1 of a
    world of 2) on the same device run the phase protocol; the K/V all-gather is emulated by copying
    each rank's chunk into the other's gather buffer.  Must match the unsharded forward."""
    from actionmesh_amd import ClassifierFreeGuidance
    from actionmesh_amd.denoiser import HipEngine, masked_time, rope_tables_host
    from actionmesh_amd.sharding import FrameShardPlan
    g, cfg, sd, model, t = _setup(name, golden_dir, dev)
    if t["init_latent"].shape[1] % 2:
        pytest.skip("odd frame count")
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    x_in, c_in, m_in, f_in = cfgd.cfg_at_inference(t["init_latent"], t["context"], t["mask"], t["framestep"])
    B, T, N, _ = x_in.shape
    S = c_in.shape[2]
    tt = [float(g["fwd_t"])] * B
    ref, _ = model.forward(x_in.to(dev), c_in.to(dev), f_in.to(dev), torch.tensor(tt, device=dev), m_in.to(dev), None)
    t_bt = masked_time(tt, m_in, B, T)
    cos, sin = rope_tables_host(f_in, 128)
    engines, plans = [], []
    for r in range(2):
        plan = FrameShardPlan(T, 2, r)
        e = HipEngine(model.hyper_params(), sd, dev, B, plan.frames_local, N, S, world=2, rank=r)
        e.set_context(plan.slice_frames(c_in.to(dev)), cos.view(B, T, -1)[:, plan.frame_slice].reshape(-1, 64),
                      sin.view(B, T, -1)[:, plan.frame_slice].reshape(-1, 64))
        tl = plan.frames_local
        e.begin(plan.slice_frames(x_in.to(dev)), [t_bt[b * T + r * tl + j] for b in range(B) for j in range(tl)])
        engines.append(e); plans.append(plan)
    for i in range(cfg.num_layers):
        for e in engines:
            e.layer_pre(i)
        if engines[0].is_inflated(i):          # emulated all-gather: rank r contributes chunk r
            (kv0,), (kv1,) = engines[0].kv_buffers(), engines[1].kv_buffers()     # [rank][K chunk | V^T chunk]
            kv0[1].copy_(kv1[1]); kv1[0].copy_(kv0[0])
        for e in engines:
            e.layer_post(i)
    v = torch.cat([e.end() for e in engines], dim=1)
    torch.cuda.synchronize()
    r = rel(v, ref)
    ref32 = torch.from_numpy(g["fwd_velocity_fp32"])
    e_sh, e_un = rel(v.float().cpu(), ref32), rel(ref.float().cpu(), ref32)
    print(f"{name}: 2-rank emulated vs unsharded rel-L2 {r:.3e}; vs the fp32 reference: sharded {e_sh:.3e}, unsharded {e_un:.3e}")
    # Two bf16 computations of the same function (the two-pass attention meets other running maxima).  With the LayerNorms folded
    # into their linears there is no bf16 rounding of the normalised activation to re-align the two runs, so they sit further
    # apart (6.2e-3; 2.6e-3 with ACTIONMESH_AMD_LN_FOLD=0) while each is exactly as close to the reference's fp32 result
    # (8.41e-3 / 8.47e-3): the second statement is the one that matters.
    assert r < 8e-3
    assert e_sh < 1.05 * e_un
    assert rel(v.float().cpu(), torch.from_numpy(g["fwd_velocity_fp32"])) < 2e-2


def test_cfg_parallel_emulated_on_one_gpu(dev, golden_dir):
    """P = 2 splits the two CFG branches across ranks (no K/V exchange): each 'rank' is an engine
    with one batch row and all frames; stacking their velocities must reproduce the batched forward."""
    from actionmesh_amd import ClassifierFreeGuidance
    from actionmesh_amd.denoiser import HipEngine, masked_time, rope_tables_host
    from actionmesh_amd.sharding import FrameShardPlan
    g, cfg, sd, model, t = _setup("tiny_inflated", golden_dir, dev)
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    x_in, c_in, m_in, f_in = cfgd.cfg_at_inference(t["init_latent"], t["context"], t["mask"], t["framestep"])
    B, T, N, _ = x_in.shape
    S = c_in.shape[2]
    tt = [float(g["fwd_t"])] * B
    ref, _ = model.forward(x_in.to(dev), c_in.to(dev), f_in.to(dev), torch.tensor(tt, device=dev), m_in.to(dev), None)
    t_bt = masked_time(tt, m_in, B, T)
    cos, sin = rope_tables_host(f_in, 128)
    outs = []
    for r in range(2):
        plan = FrameShardPlan(T, 2, r, B, 2)
        assert plan.frame_world == 1 and plan.batch_local == 1
        e = HipEngine(model.hyper_params(), sd, dev, 1, T, N, S, world=1, rank=0)
        e.set_context(plan.slice_local(c_in.to(dev)), plan.slice_local(cos.view(B, T, -1)).reshape(-1, 64),
                      plan.slice_local(sin.view(B, T, -1)).reshape(-1, 64))
        outs.append(e.forward(plan.slice_local(x_in.to(dev)), plan.local_times(t_bt)))
    v = torch.cat(outs, dim=0)
    torch.cuda.synchronize()
    assert rel(v, ref) < 2e-3


def test_full_size_properties_headline_shape(dev):
    """BASELINE.json configs[1] in its synthetic form (16 frames x 4096 tokens, width 1024, 21 layers): the CPU
    oracle would need ~15 min per forward, so parity at full size is checked through size-independent
    properties of the denoiser:
      (a) frame-permutation equivariance: permuting frames together with their framesteps, contexts and mask
          permutes the velocity (frames only talk through attention, which is permutation-equivariant, and
          RoPE depends on framestep differences only);
      (b) CFG rows are independent: changing the context of row 1 leaves row 0's velocity bit-identical;
      (c) a second identical call is bit-identical (no races / uninitialised reads at full size)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import random_state_dict
    from actionmesh_amd import HipDenoiser
    T, N, C, H, NL, S, Dc, Din = 16, 4096, 1024, 8, 21, 257, 1024, 64
    hp = dict(in_channels=Din, num_layers=NL, num_attention_heads=H, width=C, mlp_ratio=4.0,
              cross_attention_dim=Dc, inflated_layers=list(range(NL)))
    model = HipDenoiser(num_tokens_nominal=N, temporal_context_size=T, **hp)
    model.load_state_dict(random_state_dict(hp, seed=0))
    model.to(dev).eval()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, T, N, Din, generator=g).to(dev)
    x[1] = x[0]
    ctx = torch.randn(2, T, S, Dc, generator=g).to(dev)
    ctx[0] = 0
    fs = torch.arange(T, dtype=torch.float32)[None].repeat(2, 1).to(dev)
    mask = torch.zeros(2, T, device=dev); mask[:, 0] = 1
    t = torch.tensor([640.0, 640.0], device=dev)
    v, _ = model.forward(x, ctx, fs, t, mask, None)
    v_again, _ = model.forward(x, ctx, fs, t, mask, None)
    assert torch.equal(v, v_again), "(c) non-deterministic at full size"
    assert bool(torch.isfinite(v.float()).all())
    perm = torch.randperm(T, generator=torch.Generator().manual_seed(5)).to(dev)
    vp, _ = model.forward(x[:, perm].contiguous(), ctx[:, perm].contiguous(), fs[:, perm].contiguous(), t,
                          mask[:, perm].contiguous(), None)
    r = rel(vp, v[:, perm])
    print(f"headline shape: frame-permutation equivariance rel-L2 {r:.3e}")
    assert r < 1e-2, "(a)"
    ctx2 = ctx.clone(); ctx2[1] = torch.randn(T, S, Dc, generator=g).to(dev)
    v2, _ = model.forward(x, ctx2, fs, t, mask, None)
    assert torch.equal(v2[0], v[0]), "(b) CFG rows must not interact"
    assert rel(v2[1], v[1]) > 1e-3


@pytest.mark.parametrize("world", [2, 4])
def test_overlap_path_emulated_on_one_gpu(dev, world):
    """The multi-GPU overlap protocol (sharding.sharded_forward with an engine that offers layer_attn_local) on one
    device: `world` engines run pre -> attention against the local K/V shard -> [emulated all-gather] -> post
    (resume over the remote shards).  Needs >= 16 key tiles per shard, so a larger sequence than the fixtures:
    T = 8 frames x (N+1) = 512 tokens, random weights; must match the unsharded engine on the same inputs."""
    from actionmesh_amd.denoiser import HipEngine, rope_tables_host
    from actionmesh_amd.sharding import FrameShardPlan
    from oracle import denoiser_oracle as O
    hp = dict(in_channels=64, num_layers=3, num_attention_heads=2, width=256, mlp_ratio=4.0, cross_attention_dim=64,
              inflated_layers=(0, 1, 2))
    sd = O.synthetic_state_dict(O.OracleConfig(**hp), seed=3)
    B, T, N, S = 2, 8, 511, 9
    g = torch.Generator().manual_seed(11)
    x = torch.randn((B, T, N, 64), generator=g)
    ctx = torch.randn((B, T, S, 64), generator=g)
    frames = torch.arange(T).repeat(B, 1)
    t_bt = [0.37] * (B * T)
    cos, sin = rope_tables_host(frames, 128)

    def run(world, overlap=True):
        engines = []
        for r in range(world):
            plan = FrameShardPlan(T, world, r)
            e = HipEngine(hp, sd, dev, B, plan.frames_local, N, S, world=world, rank=r)
            e.set_context(plan.slice_frames(ctx.to(dev)), cos.view(B, T, -1)[:, plan.frame_slice].reshape(-1, 64),
                          sin.view(B, T, -1)[:, plan.frame_slice].reshape(-1, 64))
            tl = plan.frames_local
            e.begin(plan.slice_frames(x.to(dev)), [t_bt[b * T + r * tl + j] for b in range(B) for j in range(tl)])
            engines.append(e)
        for i in range(hp["num_layers"]):
            for e in engines:
                e.layer_pre(i)
            if world > 1:
                for e in engines:
                    if overlap:
                        e.layer_attn_local(i)      # before the "all-gather": only the local shard is in place
                bufs = [e.kv_buffers()[0] for e in engines]
                for r, dst in enumerate(bufs):     # emulated all-gather: rank s contributes row s
                    for s_, src in enumerate(bufs):
                        if s_ != r:
                            dst[s_].copy_(src[s_])
            for e in engines:
                e.layer_post(i)
        v = torch.cat([e.end() for e in engines], dim=1)
        torch.cuda.synchronize()
        for e in engines:
            e.close()
        return v.float()

    ref = run(1)
    v = run(world)
    v1 = run(world, overlap=False)
    r, r1, r01 = rel(v, ref), rel(v1, ref), rel(v, v1)
    print(f"world {world}: rel-L2 vs unsharded: overlap {r:.3e}, one-pass sharded {r1:.3e}; overlap vs one-pass {r01:.3e}")
    assert torch.isfinite(v).all()
    # different key orders round P to bf16 against different running maxima: both sharded variants sit at the same
    # distance from the unsharded run
    assert r < 1e-2 and r1 < 1e-2 and r01 < 1e-2


def test_autoregressive_windows_match_oracle(dev):
    """SURVEY 8(f) N3 on the HIP path: 7 frames, window 4, slide 3, anchor in the middle - three dependent windows
    (each conditions on the previous ones' output through the device-resident LatentBank) against the CPU oracle's
    restatement of pipeline.py:247-314/469-506 with the same CPU-drawn noise.  Tolerance: the per-step latent
    tolerance of the sampler (2e-2 rel-L2 vs fp32), with head-room for the error carried through the conditioning
    frames of later windows."""
    from actionmesh_amd import ClassifierFreeGuidance, HipDenoiser, HipSchedulerFlow
    from actionmesh_amd import windows as W
    from oracle import denoiser_oracle as O
    from oracle import windows_oracle as WO
    hp = CASES["tiny_inflated"]
    cfg = O.OracleConfig(**hp)
    sd = O.synthetic_state_dict(cfg, seed=0)
    model = HipDenoiser(num_tokens_nominal=48, temporal_context_size=4, **hp)
    model.load_state_dict(sd)
    model.to(dev).eval()
    T, N, D, S, steps = 7, 48, 64, 9, 3
    g = torch.Generator().manual_seed(21)
    ts = torch.arange(T, dtype=torch.float32)
    context = torch.randn((T, S, 64), generator=g)
    anchor = torch.randn((1, N, D), generator=g)
    sched = HipSchedulerFlow(num_inference_steps=steps, shift=3.0, is_additive=True)
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])

    bank = W.LatentBank(empty_dims=(N, D), device=str(dev))
    bank.update(ts[2:3], anchor)
    W.generate_3d_latents(model, sched, cfgd, ts, context.to(dev), bank, anchor_idx=2, window=4, slide=3,
                          latent_shape=(N, D), seed=44, device=dev, noise_device="cpu")
    ref = WO.ListLatentBank((N, D))
    ref.update(ts[2:3], anchor)
    WO.generate_3d_latents(sd, cfg, ts, context, ref, 2, 4, 3, (N, D), steps, seed=44)
    torch.cuda.synchronize()

    lat, t_sorted = bank.get_ordered()
    lat_ref, t_ref = ref.get_ordered()
    assert t_sorted.cpu().tolist() == t_ref.tolist() == list(range(T))
    assert torch.equal(lat[2].cpu(), anchor[0]), "the anchor latent is conditioning only"
    per_frame = [rel(lat[i].cpu(), lat_ref[i]) for i in range(T) if i != 2]
    print("AR windows: per-frame rel-L2 vs fp32 oracle", [f"{r:.2e}" for r in per_frame])
    assert max(per_frame) < 3e-2


def test_autoregressive_windows_configs2_at_the_headline_architecture(dev, golden_dir):
    """BASELINE configs[2]: 32 frames, window 16, slide 15, anchor frame 0 - the THREE sequentially
    dependent windows [0..15], [15..30], [16..31] of the reference (timesteps.py:77-117) - through the device-resident
    LatentBank at the HEADLINE architecture (21 layers, depth-10 skips, width 1024, 8 heads, Dc 1024, S 257) at a reduced token count
    (47 latent tokens per frame -> 768-token inflated sequences) and 3 sampler steps per window, with the same CPU-drawn noise.
    Window 2 conditions on window 1's frame 15, window 3 on window 2's frames 16..30.
    Round 6 (VERDICT r05 next #4c): BOTH halves of the statement now come from the REFERENCE's own modules (tests/golden/ar_configs2_ref.npz,
    oracle/make_golden_ar_configs2_ref.py: the reference's chunk_from, LatentBank, SchedulerFlow, ClassifierFreeGuidance and ActionMeshDenoiser
    under the window loop of pipeline.py:247-314 / 469-506): the fp32 latents of all 32 frames, and the reference's OWN reduced-precision
    distance per frame (the same loop under autocast("cpu", bfloat16) vs its fp32 run: 2.72e-2 .. 2.98e-2 - with only 3 coarse steps per
    window, dt ~ 0.33, twice the 30-step figure).  Round 5 used the oracle's fp32 run and the oracle's bf16-policy distance instead; the
    two agree (same fp32 checksum to the last bit, yardsticks within 4 %: tests/test_oracle_golden.py).
    Stated tolerance, frame by frame: rel-L2 vs the reference's fp32 latents <= 1.15 x (the reference's own autocast distance) + 2e-3.
    MEASURED on MI355X (round 5, against the oracle's figures): 2.91e-2 / 2.94e-2 / 2.94e-2 (window 1 max, frames 16..30 max, frame 31)."""
    import numpy as np
    from actionmesh_amd import ClassifierFreeGuidance, HipDenoiser, HipSchedulerFlow
    from actionmesh_amd import windows as W
    from oracle import denoiser_oracle as O
    from oracle.make_golden_ar_configs2 import HP as hp, N, D, S, STEPS as steps, T, case
    fx = np.load(os.path.join(golden_dir, "ar_configs2_ref.npz"))
    cfg, sd, ts, context, anchor = case()
    assert O.state_dict_checksum(sd) == pytest.approx(float(fx["weights_checksum"]), rel=1e-12)
    assert (int(fx["frames"]), int(fx["tokens"]), int(fx["steps"])) == (T, N, steps)
    model = HipDenoiser(num_tokens_nominal=N, temporal_context_size=16, **hp)
    model.load_state_dict(sd)
    model.to(dev).eval()
    sched = HipSchedulerFlow(num_inference_steps=steps, shift=3.0, is_additive=True)
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    assert [w.tolist() for w in W.chunk_from(0, T, 16, 15)] == [list(range(16)), list(range(15, 31)), list(range(16, 32))]

    bank = W.LatentBank(empty_dims=(N, D), device=str(dev))
    bank.update(ts[0:1], anchor)
    W.generate_3d_latents(model, sched, cfgd, ts, context.to(dev), bank, anchor_idx=0, window=16, slide=15,
                          latent_shape=(N, D), seed=int(fx["seed"]), device=dev, noise_device="cpu")
    torch.cuda.synchronize()
    lat, t_sorted = bank.get_ordered()
    lat_ref = torch.from_numpy(fx["latents_fp32"])
    assert t_sorted.cpu().tolist() == list(range(T))
    assert torch.equal(lat[0].cpu(), anchor[0]) and torch.equal(lat_ref[0], anchor[0]), "the anchor latent is conditioning only"
    per_frame = [rel(lat[i].cpu(), lat_ref[i]) for i in range(1, T)]
    yd = fx["ref_autocast_bf16_vs_fp32_per_frame"][1:].tolist()
    print("configs[2] AR windows at the headline architecture: per-frame rel-L2 vs the REFERENCE's fp32 latents: window 1 max "
          f"{max(per_frame[:15]):.2e}, frames 16..30 max {max(per_frame[15:30]):.2e}, frame 31 {per_frame[30]:.2e}; "
          f"the reference's own autocast(bf16) distance max {max(yd):.2e}; worst ratio {max(p / y for p, y in zip(per_frame, yd)):.3f}")
    for i, (p_, y_) in enumerate(zip(per_frame, yd)):
        assert p_ <= 1.15 * y_ + 2e-3, (i + 1, p_, y_)


def test_split_cfg_batch_style_loop_rebinds_the_context(dev, golden_dir):
    """the reference SchedulerFlow with split_cfg_batch=True (actionmesh_lowram.yaml) calls forward once per
    CFG branch with context[b:b+1] and hands every call the freqs_rot branch 0 returned.  Branch 0 is the zeroed context;
    the window cache must follow the context, not the cache object, or the image conditioning is silently lost."""
    from actionmesh_amd import ClassifierFreeGuidance, HipSchedulerFlow
    g, cfg, sd, model, t = _setup("tiny_inflated", golden_dir, dev)
    steps = int(g["steps"])
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    sched = HipSchedulerFlow(num_inference_steps=steps, shift=3.0, is_additive=True)
    ts, ds = sched.get_schedule()
    lat = t["init_latent"].clone().to(dev)
    ctx, mask, fs = t["context"].to(dev), t["mask"].to(dev), t["framestep"].to(dev)
    unobs = cfgd.get_unobserved_mask(mask)
    cache = None
    for i in range(steps):
        x_in, c_in, m_in, f_in = cfgd.cfg_at_inference(lat, ctx, mask, fs)
        dt = torch.tensor([float(ts[i])], device=dev).expand(2)
        outs = []
        for b in range(2):                                  # scheduler.py:159-168
            o, cache = model.forward(hidden_states=x_in[b:b + 1], context=c_in[b:b + 1], framestep=f_in[b:b + 1],
                                     mask=m_in[b:b + 1], diffusion_time=dt[b:b + 1], freqs_rot=cache)
            outs.append(o)
        v = cfgd.aggregate_cfg(torch.cat(outs, dim=0))
        flow = lat + (float(ds[i]) * v.float()).to(torch.bfloat16)
        lat[unobs] = flow[unobs]
    ref = torch.from_numpy(g["loop_final_split_cfg_fp32"])
    r = rel(lat.cpu(), ref)
    print(f"split_cfg_batch-style loop: final latents rel-L2 vs the reference's split run {r:.3e}")
    assert r < 2e-2
    # the sampler's own split mode: one forward per branch, same result as its batched mode to bf16 rounding
    a = HipSchedulerFlow(num_inference_steps=steps, shift=3.0, is_additive=True, split_cfg_batch=True).denoise(
        model, cfgd, init_latent=t["init_latent"].clone().to(dev), context=ctx, device=dev, mask=mask, framestep=fs)
    b = sched.denoise(model, cfgd, init_latent=t["init_latent"].clone().to(dev), context=ctx, device=dev, mask=mask,
                      framestep=fs)
    assert rel(a.cpu(), ref) < 2e-2 and rel(a.cpu(), b.cpu()) < 5e-3


def test_cuda_autocast_dt_switch(dev, golden_dir):
    """`cuda_autocast_dt=True` reproduces the reference GPU path's bf16 rounding of distances[i] (scheduler.py:238-241
    under cuda autocast); the default keeps it fp32 like the reference's CPU path.  One step: the two differ by exactly
    the rounding of dt."""
    from actionmesh_amd import ClassifierFreeGuidance, HipSchedulerFlow
    g, cfg, sd, model, t = _setup("tiny_inflated", golden_dir, dev)
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    outs = {}
    for flag in (False, True):
        s = HipSchedulerFlow(num_inference_steps=1, shift=3.0, is_additive=True, cuda_autocast_dt=flag)
        outs[flag] = s.denoise(model, cfgd, init_latent=t["init_latent"].clone().to(dev), context=t["context"].to(dev),
                               device=dev, mask=t["mask"].to(dev), framestep=t["framestep"].to(dev)).cpu()
    d = float(HipSchedulerFlow(num_inference_steps=1, shift=3.0).get_schedule()[1][0])
    d_bf = float(torch.tensor(d).to(torch.bfloat16))
    assert d != d_bf
    x0 = t["init_latent"]
    step32, stepbf = outs[False] - x0, outs[True] - x0
    ratio = float((stepbf[0, 1:] * step32[0, 1:]).sum() / (step32[0, 1:] ** 2).sum())
    assert ratio == pytest.approx(d_bf / d, rel=2e-3)
    assert torch.equal(outs[True][0, 0], x0[0, 0])


def test_moving_to_cpu_releases_the_engine(dev, golden_dir):
    """`--low_ram` semantics (pipeline.py:171-184): the reference unloads a stage with `.to("cpu")`; for HipDenoiser that
    must free the engine's HBM (weights, workspace, K/V caches), and a later `.to(device)` + forward must work again."""
    g, cfg, sd, model, t = _setup("tiny_inflated", golden_dir, dev)
    from actionmesh_amd import ClassifierFreeGuidance
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    x_in, c_in, m_in, f_in = cfgd.cfg_at_inference(t["init_latent"], t["context"], t["mask"], t["framestep"])
    tt = torch.tensor([float(g["fwd_t"])]).expand(2)
    args = lambda: (x_in.to(dev), c_in.to(dev), f_in.to(dev), tt.to(dev), m_in.to(dev), None)
    v0, _ = model.forward(*args())
    torch.cuda.synchronize()
    assert model._engine is not None
    free_with, _ = torch.cuda.mem_get_info(dev)
    model.to("cpu")
    assert model._engine is None and model._window is None and model.device.type == "cpu"
    torch.cuda.synchronize()
    free_without, _ = torch.cuda.mem_get_info(dev)
    assert free_without > free_with, (free_with, free_without)
    with pytest.raises(RuntimeError):
        model.forward(x_in, c_in, f_in, tt, m_in, None)
    model.to(dev)
    v1, _ = model.forward(*args())
    torch.cuda.synchronize()
    assert torch.equal(v0, v1)


@pytest.mark.parametrize("name", list(CASES))
def test_exact_shortcuts_are_bit_identical(dev, golden_dir, name):
    """am_set_branch_hints: the unconditional branch's cross-attention replaced by its to_out bias, and layer 0's self-attention
    branch computed once for all guidance branches, must not change a single bit of the latents - per step, in the batched and
    in the split sampler, and against the 3-branch default guidance (where the branches do NOT share the time token)."""
    from actionmesh_amd import ClassifierFreeGuidance, HipSchedulerFlow
    g, cfg, sd, model, t = _setup(name, golden_dir, dev)
    steps = int(g["steps"])
    args = dict(context=t["context"].to(dev), device=dev, mask=t["mask"].to(dev), framestep=t["framestep"].to(dev))
    for guidance, scales in (([[0, 1], [1, 1]], [7.5]), ([[0, 0], [0, 1], [1, 1]], [2.0, 5.0])):
        cfgd = ClassifierFreeGuidance(True, guidance, scales)
        for split in (False, True):
            outs = []
            for shortcuts in (True, False):
                s = HipSchedulerFlow(num_inference_steps=steps, shift=3.0, is_additive=True, split_cfg_batch=split,
                                     exact_shortcuts=shortcuts)
                got = [lat.clone() for lat, _ in s._flow_sample(model, cfgd, t["init_latent"].clone().to(dev), **args)]
                outs.append(torch.stack(got).cpu())
            assert torch.equal(outs[0], outs[1]), (name, guidance, split, float((outs[0] - outs[1]).abs().max()))
    # the reference-driven path finds the zero rows itself (one reduction per bind) and must agree with the plain forward
    import os
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    x_in, c_in, m_in, f_in = cfgd.cfg_at_inference(t["init_latent"], t["context"], t["mask"], t["framestep"])
    tt = torch.tensor([float(g["fwd_t"])]).expand(2)
    v1, _ = model.forward(x_in.to(dev), c_in.to(dev), f_in.to(dev), tt.to(dev), m_in.to(dev), None)
    os.environ["ACTIONMESH_AMD_NO_SHORTCUTS"] = "1"
    try:
        v0, _ = model.forward(x_in.to(dev), c_in.to(dev), f_in.to(dev), tt.to(dev), m_in.to(dev), None)
    finally:
        del os.environ["ACTIONMESH_AMD_NO_SHORTCUTS"]
    assert torch.equal(v0, v1)


@pytest.mark.parametrize("name", list(CASES))
def test_folded_layernorms(dev, golden_dir, name, monkeypatch):
    """norm_s_attn / norm_x_attn / norm_ff live inside the linears behind them (am_model.hip).  (1) The row statistics a
    producer GEMM leaves behind are the same BITS as a read-back of the rows (ACTIONMESH_AMD_LN_STATS=recompute) on a whole
    forward; (2) the round-3 sequence - LayerNorm kernel, bf16 activation, plain linear: ACTIONMESH_AMD_LN_FOLD=0 - differs only
    by the rounding of that activation, and is no closer to the reference than the folded form.  Round 6: norm_out is folded into
    proj_out as well (temporal_denoiser.py:239-242), so the two sequences now differ in one more rounding per row - the stated distance
    between them goes from 8e-3 to 9e-3 (measured 8.1e-3 on tiny_inflated; both are 8.4-8.5e-3 from the reference's fp32)."""
    from actionmesh_amd import ClassifierFreeGuidance, HipDenoiser
    g, cfg, sd, model, t = _setup(name, golden_dir, dev)
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    x_in, c_in, m_in, f_in = cfgd.cfg_at_inference(t["init_latent"], t["context"], t["mask"], t["framestep"])
    tt = torch.tensor([float(g["fwd_t"])]).expand(2)

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = HipDenoiser(num_tokens_nominal=48, temporal_context_size=4, **CASES[name])
        m.load_state_dict(sd)
        m.to(dev).eval()
        v, _ = m.forward(x_in.to(dev), c_in.to(dev), f_in.to(dev), tt.to(dev), m_in.to(dev), None)
        for k in env:
            monkeypatch.delenv(k)
        return v.float().cpu()

    v = run({})
    assert torch.equal(v, run({"ACTIONMESH_AMD_LN_STATS": "recompute"}))
    v0 = run({"ACTIONMESH_AMD_LN_FOLD": "0"})
    ref32 = torch.from_numpy(g["fwd_velocity_fp32"])
    print(f"{name}: folded vs un-folded rel-L2 {rel(v, v0):.3e}; vs reference fp32: folded {rel(v, ref32):.3e}, un-folded {rel(v0, ref32):.3e}")
    assert rel(v, v0) < 9e-3
    assert rel(v, ref32) < 1.05 * rel(v0, ref32)


@pytest.mark.parametrize("use_graph", [False, True])
def test_reloaded_weights_rebuild_the_folded_linears(dev, golden_dir, use_graph):
    """am_load_weight on a live handle (a C-ABI user swapping a LayerNorm's affine or the linear behind it): the folded linears are
    rebuilt by the next forward's entry point - in front of a graph replay too - and the result is the one of a fresh handle."""
    import ctypes as C
    from actionmesh_amd import ClassifierFreeGuidance, HipDenoiser
    name = "tiny_mixed"
    g, cfg, sd, model, t = _setup(name, golden_dir, dev)
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    x_in, c_in, m_in, f_in = cfgd.cfg_at_inference(t["init_latent"], t["context"], t["mask"], t["framestep"])
    tt = torch.tensor([float(g["fwd_t"])]).expand(2)

    def build(state):
        m = HipDenoiser(num_tokens_nominal=48, temporal_context_size=4, use_graph=use_graph, **CASES[name])
        m.load_state_dict(state)
        return m.to(dev).eval()

    def fwd(m, n=1):
        cache, v = None, None
        for _ in range(n):                       # graph engines: eager, captured, replayed
            v, cache = m.forward(x_in.to(dev), c_in.to(dev), f_in.to(dev), tt.to(dev), m_in.to(dev), cache)
        torch.cuda.synchronize()
        return v.float().cpu(), cache

    sd2 = {k: v.clone() for k, v in sd.items()}
    changed = ("blocks.1.norm_ff.weight", "blocks.0.norm_s_attn.bias", "blocks.2.norm_x_attn.weight", "blocks.3.ff.net.0.proj.weight",
               "blocks.3.ff.net.0.proj.bias", "blocks.4.s_attn.to_k.weight")
    gen = torch.Generator().manual_seed(3)
    for k in changed:
        sd2[k] = sd2[k] * (1.0 + 0.3 * torch.randn(sd2[k].shape, generator=gen)) + 0.05
    m = build(sd)
    v_old, cache = fwd(m, 3)
    e = m._engine
    for k in changed:
        w = sd2[k].detach().to("cpu", torch.float32).contiguous()
        e._check(e.lib.am_load_weight(e.handle, k.encode(), C.c_void_p(w.data_ptr()), w.numel()), f"am_load_weight({k})")
    v_new = None
    for _ in range(2):
        v_new, cache = m.forward(x_in.to(dev), c_in.to(dev), f_in.to(dev), tt.to(dev), m_in.to(dev), cache)
    torch.cuda.synchronize()
    want, _ = fwd(build(sd2), 3)
    assert not torch.equal(v_old, want)
    assert torch.equal(v_new.float().cpu(), want)


@pytest.mark.parametrize("name", list(CASES))
def test_graph_replay_is_bit_identical(dev, golden_dir, name):
    """am_denoise_forward_graph (HipDenoiser(use_graph=True)): the sampler through a captured forward - first step eager, second
    captured, the rest replayed with new per-frame times - gives the same latents bit for bit as the eager launches; a new window
    (context) drops the graph and captures again."""
    from actionmesh_amd import ClassifierFreeGuidance, HipDenoiser, HipSchedulerFlow
    g, cfg, sd, model, t = _setup(name, golden_dir, dev)
    steps = max(int(g["steps"]), 4)
    args = dict(device=dev, mask=t["mask"].to(dev), framestep=t["framestep"].to(dev))
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    gm = HipDenoiser(num_tokens_nominal=model.num_tokens_nominal, temporal_context_size=model.temporal_context_size,
                     use_graph=True, **model.hyper_params())
    gm.load_state_dict(sd)
    gm.to(dev).eval()
    for k, ctx in enumerate((t["context"], t["context"] * 0.5)):                # two windows
        outs = []
        for m in (model, gm):
            s = HipSchedulerFlow(num_inference_steps=steps, shift=3.0, is_additive=True)
            got = [lat.clone() for lat, _ in s._flow_sample(m, cfgd, t["init_latent"].clone().to(dev), context=ctx.to(dev), **args)]
            outs.append(torch.stack(got).cpu())
        assert torch.equal(outs[0], outs[1]), (name, k, float((outs[0] - outs[1]).abs().max()))
        replays, captures, eager, failed = gm._engine.graph_stats()
        assert failed == 0 and captures == k + 1 and replays == (k + 1) * (steps - 2) and eager == k + 1, (replays, captures, eager, failed)
    assert model._engine.graph_stats() == (0, 0, 0, 0)


@pytest.mark.gpu
def test_graph_with_split_cfg_batch_keeps_the_branches_apart(dev, golden_dir):
    """with use_graph the engine used to hand out its one persistent result buffer, so split_cfg_batch (one forward
    per guidance branch, results concatenated afterwards) saw every branch alias the last one and guidance collapsed silently.
    The split + graph loop must equal the split eager loop bit for bit (the result is a copy now)."""
    from actionmesh_amd import ClassifierFreeGuidance, HipDenoiser, HipSchedulerFlow
    g, cfg, sd, model, t = _setup("tiny_inflated", golden_dir, dev)
    steps = max(int(g["steps"]), 3)
    args = dict(device=dev, mask=t["mask"].to(dev), framestep=t["framestep"].to(dev), context=t["context"].to(dev))
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    gm = HipDenoiser(num_tokens_nominal=model.num_tokens_nominal, temporal_context_size=model.temporal_context_size,
                     use_graph=True, **model.hyper_params())
    gm.load_state_dict(sd)
    gm.to(dev).eval()
    outs = []
    for m in (model, gm):
        s = HipSchedulerFlow(num_inference_steps=steps, shift=3.0, is_additive=True, split_cfg_batch=True)
        outs.append(s.denoise(m, cfgd, init_latent=t["init_latent"].clone().to(dev), **args).cpu())
    assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())
    # and the two branches really differ (a collapsed guidance would make v_cond == v_uncond)
    x = t["init_latent"].clone().to(dev)
    x_in, c_in, m_in, f_in = cfgd.cfg_at_inference(x, args["context"], args["mask"], args["framestep"])
    tt = torch.full((1,), 500.0, device=dev)
    v0, fr = gm(x_in[:1], c_in[:1], f_in[:1], tt, mask=m_in[:1])
    v1, _ = gm(x_in[1:], c_in[1:], f_in[1:], tt, mask=m_in[1:], freqs_rot=None)
    assert v0.data_ptr() != v1.data_ptr() and not torch.equal(v0, v1)
