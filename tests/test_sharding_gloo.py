"""N>1 path on CPU: world_size-2 gloo run of the frame-sharded forward driver
(actionmesh_amd/sharding.py) with an oracle-backed stand-in engine that follows the same phase
protocol as the HIP engine (begin / layer_pre -> K/V all-gather -> layer_post / end).  Checks the
shard plan, the chunked K/V gather layout and the exchange against the unsharded oracle."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from actionmesh_amd.sharding import FrameShardPlan, gather_frames, sharded_forward
from oracle import denoiser_oracle as O

KW = dict(in_channels=64, num_layers=5, num_attention_heads=2, width=256, mlp_ratio=4.0,
          cross_attention_dim=64, inflated_layers=(0, 1, 3, 4))


class OracleEngine:
    """CPU stand-in for HipEngine (tests only): same phases, K/V shards in (world, chunk) buffers."""

    def __init__(self, sd, cfg, plan, ctx_local, cos_local, sin_local, B, N):
        self.sd, self.cfg, self.plan = sd, cfg, plan
        self.num_layers = cfg.num_layers
        self.P = O.Precision("fp32")
        self.ctx = ctx_local.reshape(-1, ctx_local.shape[2], ctx_local.shape[3])
        self.cos, self.sin = cos_local, sin_local
        Tl, L, C = plan.frames_local, N + 1, cfg.width
        self.chunk = B * Tl * L * C
        self.k = torch.zeros(plan.frame_world, self.chunk)
        self.v = torch.zeros(plan.frame_world, self.chunk)

    def is_inflated(self, i):
        return i in self.cfg.inflated_layers

    def kv_buffers(self):
        return self.k, self.v

    def begin(self, x, t_bt):
        sd, cfg, P = self.sd, self.cfg, self.P
        B, T, N, D = x.shape
        self.B, self.T, self.N, self.L = B, T, N, N + 1
        h = P.linear(x.reshape(B * T, N, D), sd["proj_in.weight"], sd["proj_in.bias"])
        e = O.timestep_sinusoid(torch.tensor(t_bt), cfg.width)
        e = P.linear(e, sd["time_proj.linear_1.weight"], sd["time_proj.linear_1.bias"])
        e = P.linear(F.gelu(e), sd["time_proj.linear_2.weight"], sd["time_proj.linear_2.bias"])
        self.h = torch.cat([e[:, None], h], 1)
        self.skips = []

    def layer_pre(self, i):
        sd, cfg, P = self.sd, self.cfg, self.P
        p = f"blocks.{i}."
        H, hd = cfg.num_attention_heads, cfg.head_dim
        if cfg.has_skip(i):
            cat = torch.cat([self.skips.pop(), self.h], -1)
            self.h = O.fp32_layer_norm(P.linear(cat, sd[p + "linear_skip.weight"], sd[p + "linear_skip.bias"]),
                                       sd[p + "norm_skip.weight"], sd[p + "norm_skip.bias"])
        z = O.fp32_layer_norm(self.h, sd[p + "norm_s_attn.weight"], sd[p + "norm_s_attn.bias"])
        BT, L, C = z.shape
        qkv = torch.cat([P.linear(z, sd[p + f"s_attn.{n}.weight"], None) for n in ("to_q", "to_k", "to_v")], -1)
        q, k, v = torch.split(qkv.view(BT, L, H, 3 * hd), hd, dim=-1)           # (BT, L, H, hd)
        q, k, v = (t.transpose(1, 2) for t in (q, k, v))                         # (BT, H, L, hd)
        cos = self.cos[:, None, :].expand(BT, L, hd); sin = self.sin[:, None, :].expand(BT, L, hd)
        q = O.apply_rope(O.rms_norm(q, sd[p + "s_attn.norm_q.weight"]), cos, sin)
        k = O.apply_rope(O.rms_norm(k, sd[p + "s_attn.norm_k.weight"]), cos, sin)
        self.q = q
        r = self.plan.frame_rank if self.is_inflated(i) else 0
        self.k[r] = k.reshape(-1)
        self.v[r] = v.reshape(-1)

    def layer_post(self, i):
        sd, cfg, P = self.sd, self.cfg, self.P
        p = f"blocks.{i}."
        H, hd, B, T, L = cfg.num_attention_heads, cfg.head_dim, self.B, self.T, self.L
        C = cfg.width
        shp = (B * T, H, L, hd)
        if self.is_inflated(i):
            # every chunk is (B, T_local, H, L, hd); keys of all chunks are simply concatenated
            ks = [self.k[c].view(B, T, H, L, hd).permute(0, 2, 1, 3, 4).reshape(B, H, T * L, hd) for c in range(self.plan.frame_world)]
            vs = [self.v[c].view(B, T, H, L, hd).permute(0, 2, 1, 3, 4).reshape(B, H, T * L, hd) for c in range(self.plan.frame_world)]
            q = self.q.view(B, T, H, L, hd).permute(0, 2, 1, 3, 4).reshape(B, H, T * L, hd)
            o = F.scaled_dot_product_attention(q, torch.cat(ks, 2), torch.cat(vs, 2))
            o = o.view(B, H, T, L, hd).permute(0, 2, 3, 1, 4).reshape(B * T, L, C)
        else:
            o = F.scaled_dot_product_attention(self.q, self.k[0].view(shp), self.v[0].view(shp))
            o = o.transpose(1, 2).reshape(B * T, L, C)
        h = self.h + P.linear(o, sd[p + "s_attn.to_out.0.weight"], sd[p + "s_attn.to_out.0.bias"])
        z = O.fp32_layer_norm(h, sd[p + "norm_x_attn.weight"], sd[p + "norm_x_attn.bias"])
        h = h + O.cross_attention(z, self.ctx, sd, p + "x_attn.", H, P)
        z = O.fp32_layer_norm(h, sd[p + "norm_ff.weight"], sd[p + "norm_ff.bias"])
        u = F.gelu(P.linear(z, sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"]))
        self.h = h + P.linear(u, sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"])
        if i < cfg.num_layers // 2:
            self.skips.append(self.h)

    def end(self):
        sd = self.sd
        h = O.fp32_layer_norm(self.h, sd["norm_out.weight"], sd["norm_out.bias"])[:, -self.N:]
        return self.P.linear(h, sd["proj_out.weight"], sd["proj_out.bias"]).reshape(self.B, self.T, self.N, -1)


class OverlapOracleEngine(OracleEngine):
    """Stand-in that also offers HipEngine's optional overlap hook: sharded_forward must then enqueue the all-gather
    asynchronously, call the hook once per exchanged layer, and wait before layer_post."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.local_calls, self.pending = [], None

    def layer_attn_local(self, i):
        assert self.is_inflated(i) and self.plan.frame_world > 1
        self.local_calls.append(i)
        self.pending = i

    def layer_post(self, i):
        if self.is_inflated(i) and self.plan.frame_world > 1:
            assert self.pending == i, "layer_attn_local must run between layer_pre and layer_post"
        self.pending = None
        super().layer_post(i)


def _inputs():
    g = torch.Generator().manual_seed(5)
    B, T, N, S = 2, 4, 20, 7
    x = torch.randn(B, T, N, 64, generator=g)
    ctx = torch.randn(B, T, S, 64, generator=g); ctx[0] = 0
    fs = torch.tensor([[2.0, 0.0, 1.0, 3.0]]).repeat(B, 1)
    mask = torch.zeros(B, T); mask[:, 0] = 1
    t = torch.tensor([640.0, 640.0])
    return x, ctx, fs, mask, t


def _worker(rank, world, port, q, cfg_groups=1, overlap=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from actionmesh_amd.denoiser import masked_time, rope_tables_host
        torch.set_num_threads(2)
        cfg = O.OracleConfig(**KW)
        sd = O.synthetic_state_dict(cfg, seed=0)
        x, ctx, fs, mask, t = _inputs()
        B, T, N, _ = x.shape
        plan = FrameShardPlan(T, world, rank, B, cfg_groups)
        groups = [dist.new_group(plan.frame_group_ranks(g)) for g in range(cfg_groups)] if cfg_groups > 1 else [None]
        cos, sin = rope_tables_host(fs, 128)                         # from the FULL window's framesteps
        cos = plan.slice_local(cos.repeat_interleave(2, -1).view(B, T, -1)).reshape(-1, 128)
        sin = plan.slice_local(sin.repeat_interleave(2, -1).view(B, T, -1)).reshape(-1, 128)
        eng = (OverlapOracleEngine if overlap else OracleEngine)(sd, cfg, plan, plan.slice_local(ctx), cos, sin, plan.batch_local, N)
        t_local = plan.local_times(masked_time(t.tolist(), mask, B, T))
        v_local = sharded_forward(eng, plan, groups[plan.cfg_rank], plan.slice_local(x), t_local)
        if overlap and plan.frame_world > 1:
            assert eng.local_calls == [i for i in range(cfg.num_layers) if eng.is_inflated(i)]
        _buf, v = gather_frames(v_local, plan, None)
        if rank == 0:
            q.put(v)
    finally:
        dist.destroy_process_group()


def _run_world(world, cfg_groups, overlap=False):
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29400 + (os.getpid() * 7 + world * 13 + cfg_groups + 100 * overlap) % 500
    procs = [ctxm.Process(target=_worker, args=(r, world, port, q, cfg_groups, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    v = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cfg = O.OracleConfig(**KW)
    sd = O.synthetic_state_dict(cfg, seed=0)
    x, ctx, fs, mask, t = _inputs()
    ref = O.denoiser_forward(sd, cfg, x, ctx, fs, t, mask, "fp32")
    assert torch.allclose(v, ref, rtol=1e-4, atol=2e-5), float((v - ref).abs().max())


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,cfg_groups", [(2, 2), (4, 2)])
def test_cfg_parallel_times_frame_shard_equals_unsharded_oracle(world, cfg_groups):
    """CFG branches split first (world 2: no K/V exchange at all), then frames (world 4: 2 x 2)."""
    _run_world(world, cfg_groups)


@pytest.mark.timeout(300)
def test_world2_sharded_forward_equals_unsharded_oracle():
    _run_world(2, 1)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,cfg_groups", [(2, 1), (4, 2)])
def test_async_exchange_with_local_attention_hook(world, cfg_groups):
    """Engines offering layer_attn_local get the overlapped protocol: async all-gather, hook, wait, post."""
    _run_world(world, cfg_groups, overlap=True)


def test_world1_driver_is_identity_plan():
    cfg = O.OracleConfig(**KW)
    sd = O.synthetic_state_dict(cfg, seed=0)
    from actionmesh_amd.denoiser import masked_time, rope_tables_host
    x, ctx, fs, mask, t = _inputs()
    B, T, N, _ = x.shape
    plan = FrameShardPlan(T, 1, 0)
    cos, sin = rope_tables_host(fs, 128)
    eng = OracleEngine(sd, cfg, plan, ctx, cos.repeat_interleave(2, -1), sin.repeat_interleave(2, -1), B, N)
    v = sharded_forward(eng, plan, None, x, masked_time(t.tolist(), mask, B, T))
    ref = O.denoiser_forward(sd, cfg, x, ctx, fs, t, mask, "fp32")
    assert torch.allclose(v, ref, rtol=1e-4, atol=2e-5)


# ---------------------------------------------------------------------------------------------
# HipDenoiser's own multi-rank plumbing (_plan, _frame_group/new_group, bind_window slicing,
# forward_host_time, gather) with the HIP engine swapped for the oracle-backed stand-in.
# ---------------------------------------------------------------------------------------------
class FakeHipEngine(OracleEngine):
    """Same constructor / methods as actionmesh_amd.denoiser.HipEngine, computing with the oracle."""

    def __init__(self, hp, state_dict, device, max_batch, frames_local, tokens, ctx_tokens,
                 world=1, rank=0, attn_defer_log2=8, attn_dtype="bf16", kv_factory=None, use_graph=False, dtype="bfloat16"):
        self.device = torch.device(device)
        self.kind = "f16" if "16" in str(dtype) and "bf" not in str(dtype) else "bf16"
        self.world, self.rank = world, rank
        self.bounds = (max_batch, frames_local, tokens, ctx_tokens)
        self._cfg = O.OracleConfig(in_channels=hp["in_channels"], num_layers=hp["num_layers"],
                                   num_attention_heads=hp["num_attention_heads"], width=hp["width"],
                                   mlp_ratio=hp["mlp_ratio"], cross_attention_dim=hp["cross_attention_dim"],
                                   inflated_layers=tuple(hp["inflated_layers"]))
        self._sd = state_dict

    def close(self):
        pass

    def fits(self, B, T, N, S):
        b = self.bounds
        return B <= b[0] and T <= b[1] and N <= b[2] and S <= b[3]

    def set_context(self, ctx_local, cos, sin, ctx_zero=None, shared_prefix=False):
        # the branch hints are exact shortcuts: the oracle stand-in computes the long way
        B, T, S, _ = ctx_local.shape
        plan = FrameShardPlan(T * self.world, self.world, self.rank)     # only frame_world/frame_rank are used
        OracleEngine.__init__(self, self._sd, self._cfg, plan, ctx_local, cos.repeat_interleave(2, -1),
                              sin.repeat_interleave(2, -1), B, self.bounds[2])

    def forward(self, x_local, t_bt_local):
        return sharded_forward(self, FrameShardPlan(x_local.shape[1], 1, 0), None, x_local, t_bt_local)


def _denoiser_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import actionmesh_amd.denoiser as D
        torch.set_num_threads(2)
        D.HipEngine = FakeHipEngine                      # the only substitution
        cfg = O.OracleConfig(**KW)
        sd = O.synthetic_state_dict(cfg, seed=0)
        x, ctx, fs, mask, t = _inputs()
        model = D.HipDenoiser(num_tokens_nominal=20, temporal_context_size=4, process_group=dist.group.WORLD, **KW)
        model.load_state_dict(sd)
        v, cache = model.forward(x, ctx, fs, t, mask, None)
        v2, cache2 = model.forward(x, ctx, fs, t, mask, cache)
        assert cache2 is cache and torch.equal(v, v2)
        if rank == 0:
            q.put(v)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])
def test_hipdenoiser_multirank_plumbing(world):
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29400 + (os.getpid() * 11 + world * 17) % 500
    procs = [ctxm.Process(target=_denoiser_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    v = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cfg = O.OracleConfig(**KW)
    sd = O.synthetic_state_dict(cfg, seed=0)
    x, ctx, fs, mask, t = _inputs()
    ref = O.denoiser_forward(sd, cfg, x, ctx, fs, t, mask, "fp32")
    assert torch.allclose(v, ref, rtol=1e-4, atol=2e-5), float((v - ref).abs().max())


# ---------------------------------------------------------------------------------------------
# The sampler with its latents kept SHARDED across the steps (HipSchedulerFlow.keep_latents_sharded):
# every rank advances its own frames, no velocity gather per step (one small all-gather among the same-frame ranks when the
# CFG branches are split over groups), frames gathered once behind the last step.  The CFG + Euler kernel (ops.flow_step) is a
# HIP kernel; on CPU it is stood in for by the same arithmetic in torch - test infrastructure, like FakeHipEngine.
# ---------------------------------------------------------------------------------------------
def _flow_step_torch(v, latents, scales, dt, is_additive, unobserved):
    r = lambda x: x.to(torch.bfloat16).float()
    vb = [r(v[b].float()) for b in range(v.shape[0])]
    out, prev = vb[0], vb[0]
    for b in range(1, len(vb)):
        out = r(out + r(float(scales[b - 1]) * r(vb[b] - prev)))
        prev = vb[b]
    upd = (1.0 if is_additive else -1.0) * r(dt * out)
    for f in range(latents.shape[0]):
        if unobserved is None or unobserved[f]:
            latents[f] += upd[f]


def _sampler_worker(rank, world, port, q, cfg_parallel, split):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import actionmesh_amd.denoiser as D
        import actionmesh_amd.scheduler as S
        torch.set_num_threads(2)
        D.HipEngine = FakeHipEngine
        S.ops.flow_step = _flow_step_torch
        calls = {"all": 0}
        orig = dist.all_gather_into_tensor

        def counting(out, inp, *a, **k):
            calls["all"] += 1
            calls["bytes"] = calls.get("bytes", 0) + out.numel() * out.element_size()
            return orig(out, inp, *a, **k)
        dist.all_gather_into_tensor = counting
        cfg = O.OracleConfig(**KW)
        sd = O.synthetic_state_dict(cfg, seed=0)
        x, ctx, fs, mask, _t = _inputs()
        outs, ncoll = [], []
        for keep in (True, False):
            model = D.HipDenoiser(num_tokens_nominal=20, temporal_context_size=4, process_group=dist.group.WORLD,
                                  cfg_parallel=cfg_parallel, **KW)
            model.load_state_dict(sd)
            sched = S.HipSchedulerFlow(num_inference_steps=3, shift=3.0, is_additive=True, split_cfg_batch=split,
                                       keep_latents_sharded=keep, exact_shortcuts=False)
            cfgd = S.ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
            calls["all"], calls["bytes"] = 0, 0
            lat = sched.denoise(model, cfgd, x[:1].clone(), ctx[1:2], device="cpu", mask=mask[:1], framestep=fs[:1])
            outs.append(lat.clone())
            ncoll.append((calls["all"], calls["bytes"]))
        q.put((rank, outs[0], outs[1], ncoll))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,cfg_parallel,split", [(2, False, False), (2, True, False), (4, True, False), (2, False, True),
                                                      (6, True, False)])      # 6 = 2 CFG groups x 3: 4 frames do not divide -> replicas
def test_sampler_keeps_latents_sharded_across_steps(world, cfg_parallel, split):
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29400 + (os.getpid() * 5 + world * 19 + 3 * cfg_parallel + 7 * split) % 500
    procs = [ctxm.Process(target=_sampler_worker, args=(r, world, port, q, cfg_parallel, split)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = got[0][2]
    for rank, sharded, gathered, ncoll in got:
        assert torch.equal(gathered, ref), f"rank {rank}: gather-every-step latents differ between ranks"
        assert torch.equal(sharded, gathered), f"rank {rank}: sharded-latents sampler differs from the gather-every-step one"
        (n_sh, b_sh), (n_ga, b_ga) = ncoll
        assert n_sh <= n_ga + 1 and b_sh <= b_ga, (rank, ncoll)      # (+1: the one frame gather behind the last step)
        if world in (2, 4) and (world > 2 or not cfg_parallel):   # frames are sharded: strictly fewer gathered bytes
            assert b_sh < b_ga, (rank, ncoll)
    if not cfg_parallel:          # one CFG group: the only velocity-side collective left is the final frame gather
        steps, layers_inflated = 3, sum(1 for i in range(KW["num_layers"]) if i in KW["inflated_layers"])
        per_forward_kv = 2 * layers_inflated                     # K and V^T shards, one all-gather each per inflated layer
        forwards = steps * (2 if split else 1)
        assert got[0][3][0][0] == forwards * per_forward_kv + 1, got[0][3]


# ---- bench.py --gpus N: both exchange back-ends in one invocation, fallback, watchdog ------------------------
def _legs_worker(rank, world, port, outdir, scenario):
    import json
    import time
    from datetime import timedelta
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from actionmesh_amd.sharding import run_exchange_legs

    def write(payload):
        with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
            json.dump(payload, f)

    def run_leg(name, ctl):
        if scenario == "rccl_raises_on_rank1" and name == "rccl" and rank == 1:
            raise RuntimeError("ncclSystemError: stand-in failure")
        if scenario == "second_leg_hangs" and name == "peer":
            time.sleep(600)                               # a collective that never returns
        if scenario == "first_leg_hangs" and name == "rccl":
            time.sleep(600)
        if scenario == "rank1_raises_mid_leg" and name == "rccl":
            # rank 1 leaves the leg while rank 0 is inside the leg's OWN collectives (bench's barrier + MAX all-reduce on
            # the control group): the verdict must not pair with them, and rank 0 must come back out (the group's timeout)
            if rank == 1:
                raise RuntimeError("out of memory on one device: stand-in failure")
            dist.barrier(group=ctl)
            t = torch.tensor([1.0]); dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctl)
        if name == "peer":                                # the next leg's collectives work on ITS fresh group whatever happened before
            t = torch.tensor([float(rank)]); dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctl)
            assert float(t) == world - 1
        return {"elapsed": 1.0 + (0.5 if name == "rccl" else 0.0) + 0.01 * rank}

    def on_watchdog(name, legs, report):
        report[name] = {"ok": False, "error": "watchdog"}
        write({"watchdog": name, "legs": sorted(legs), "report": report})
        os._exit(0 if legs else 3)

    legs, report = run_exchange_legs(["rccl", "peer"], run_leg, rank, world, 8.0 if scenario == "rank1_raises_mid_leg" else 3.0, on_watchdog,
                                     describe=lambda r: {"ms_per_step": r["elapsed"] * 1e3},
                                     make_ctl=lambda: dist.new_group(backend="gloo", timeout=timedelta(seconds=3)))
    write({"legs": sorted(legs), "report": report})
    os._exit(0)          # (a group whose collective timed out does not always tear down cleanly; the verdict is on disk)


@pytest.mark.parametrize("scenario", ["both_ok", "rccl_raises_on_rank1", "second_leg_hangs", "first_leg_hangs", "rank1_raises_mid_leg"])
def test_exchange_ab_fallback_and_watchdog(tmp_path, scenario):
    """sharding.run_exchange_legs between two gloo processes with stand-in legs: (a) both complete - both reported; (b) the RCCL leg
    raises on ONE rank - every rank agrees it failed (error text where it happened, "failed on another rank" elsewhere) and the
    copy-engine leg runs and is the result; (c) the second leg never returns - the watchdog of every rank reports the first leg and
    leaves with exit code 0; (d) the FIRST leg never returns - the watchdog is armed there too and leaves with exit code 3 (round 6);
    (e) one rank raises while the other is inside the leg's own control-plane collectives - the verdicts travel on the store, the
    stuck rank's collective times out, both agree and the next leg runs on a fresh group."""
    import json
    ctxm = mp.get_context("spawn")
    port = 29400 + (os.getpid() * 11 + len(scenario) * 17) % 500
    procs = [ctxm.Process(target=_legs_worker, args=(r, 2, port, str(tmp_path), scenario)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == (3 if scenario == "first_leg_hangs" else 0), p.exitcode
    out = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    if scenario == "both_ok":
        for o in out:
            assert o["legs"] == ["peer", "rccl"] and o["report"]["rccl"]["ok"] and o["report"]["peer"]["ok"]
            assert o["report"]["rccl"]["ms_per_step"] > o["report"]["peer"]["ms_per_step"]
    elif scenario == "rccl_raises_on_rank1":
        for o in out:
            assert o["legs"] == ["peer"] and o["report"]["peer"]["ok"] and not o["report"]["rccl"]["ok"]
        assert "failed on another rank" in out[0]["report"]["rccl"]["error"] and "stand-in failure" in out[0]["report"]["rccl"]["error"]
        assert "stand-in failure" in out[1]["report"]["rccl"]["error"]
    elif scenario == "rank1_raises_mid_leg":
        for o in out:
            assert o["legs"] == ["peer"] and o["report"]["peer"]["ok"] and not o["report"]["rccl"]["ok"], o
        assert "stand-in failure" in out[1]["report"]["rccl"]["error"]
        assert out[0]["report"]["rccl"]["failed_ranks"] == [0, 1]          # rank 0's own collective timed out: it failed there too
    elif scenario == "first_leg_hangs":
        for o in out:
            assert o["watchdog"] == "rccl" and o["legs"] == [] and not o["report"]["rccl"]["ok"]
    else:
        for o in out:
            assert o["watchdog"] == "peer" and o["legs"] == ["rccl"] and o["report"]["rccl"]["ok"] and not o["report"]["peer"]["ok"]


def test_bench_self_launch_argument_passing_on_cpu():
    """`python bench.py --gpus N` without a launcher, on a box with no GPU: (a) fewer devices than ranks - ONE JSON
    error line, exit code 2, nothing launched; (b) --same-device (no device count to check): the command re-executes itself under
    torch.distributed.run with the SAME arguments, the ranks die here for want of a GPU, and the launcher still ends with one JSON line
    (the attempts with their arguments and exit codes) and a non-zero exit code."""
    import json
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("the no-GPU behaviour of the launcher")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 2 and rec["value"] is None and rec["visible_devices"] == 0 and "--gpus 2" in rec["error"]
    argv = ["--gpus", "2", "--same-device", "--steps", "2", "--warmup", "1", "--shape", "small", "--no-cpu-baseline"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], capture_output=True, text=True, env=env, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode != 0 and len(lines) == 1 and lines[0] == r.stdout.strip().splitlines()[-1], r.stdout[-500:]
    rec = json.loads(lines[0])
    att = rec["launcher"]["attempts"]
    assert rec["launcher"]["self_launched"] and len(att) == 1 and att[0]["argv"] == argv and att[0]["returncode"] != 0


def test_launcher_command_is_the_drivers():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    cmd = bench.launcher_argv(["--gpus", "8", "--steps", "20", "--warmup", "5"], 8, 29511)
    assert cmd[1:9] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port", "29511"]
    assert cmd[9].endswith("bench.py") and cmd[10:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert bench.preflight_plan(8, 1) == 8 and bench.preflight_plan(8, 0) == 16 and bench.preflight_plan(2, 1) == 4
