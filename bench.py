#!/usr/bin/env python
"""bench.py - denoise-steps/sec of the Stage-I hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one CFG-batched (B=2) denoiser forward + CFG recombination + Euler update
(BASELINE.json metric).  Workload = BASELINE.json configs[1] shape in its synthetic form
(SURVEY.md 8(d) "headline"): T=16 frames x N=4096 latent tokens, width 1024 (8 heads x 128),
21 layers all inflated, S=257 context tokens, random-init weights, seeded N(0,1) inputs already
resident in HBM.  N>1: frames are sharded across ranks (strong scaling: the problem is fixed),
one K/V all-gather per layer over RCCL.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  "roofline":     dominant kernel (inflated self-attention) vs the dense bf16 MFMA peak, timed
                  live with HIP events on the launch stream,
  "cpu_baseline": the CPU oracle (a port of the reference path; oracle/denoiser_oracle.py)
                  timed on this box's host cores on a bounded sample (N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_FP8_TFLOPS = 5000.0       # dense fp8 (MX-scaled K=128 MFMA) peak, same guide

SHAPES = {
    # name: (T, N, width, heads, layers, S, Dc, Din)
    "headline": (16, 4096, 1024, 8, 21, 257, 1024, 64),     # "16f x 4096tok x 1024-dim"
    "nominal": (16, 2048, 2048, 16, 21, 257, 1024, 64),     # the shipped model (actionmesh.yaml:33-43)
    "small": (4, 512, 256, 2, 5, 17, 64, 64),               # plumbing check
    # BASELINE.json configs[4] in bf16: 64 frames x 8192 tokens (T*L = 524 352-token attention, 64x the headline's
    # attention flops: ~45 s per step on one GPU - meant for 8 GPUs; fits one MI355X: ~40 GB of activations)
    "long64": (64, 8192, 1024, 8, 21, 257, 1024, 64),
}


def random_state_dict(hp, seed=0):
    """Random-init weights of the reference architecture (nn.Linear-scale uniform; norms = 1/0)."""
    C, F_, Dc, Din, NL = hp["width"], int(hp["width"] * hp["mlp_ratio"]), hp["cross_attention_dim"], hp["in_channels"], hp["num_layers"]
    g = torch.Generator().manual_seed(seed)

    def lin(n_out, n_in, bias=True, name=""):
        b = n_in ** -0.5
        out = {name + ".weight": (torch.rand((n_out, n_in), generator=g) * 2 - 1) * b}
        if bias:
            out[name + ".bias"] = (torch.rand((n_out,), generator=g) * 2 - 1) * b
        return out

    def norm(n, name, bias=True):
        out = {name + ".weight": torch.ones(n)}
        if bias:
            out[name + ".bias"] = torch.zeros(n)
        return out

    sd = {}
    sd.update(lin(4 * C, C, name="time_proj.linear_1")); sd.update(lin(C, 4 * C, name="time_proj.linear_2"))
    sd.update(lin(C, Din, name="proj_in"))
    for i in range(NL):
        p = f"blocks.{i}."
        sd.update(norm(C, p + "norm_s_attn")); sd.update(norm(C, p + "norm_x_attn")); sd.update(norm(C, p + "norm_ff"))
        for a in ("s_attn", "x_attn"):
            sd.update(norm(128, p + a + ".norm_q", bias=False)); sd.update(norm(128, p + a + ".norm_k", bias=False))
            sd.update(lin(C, C, bias=False, name=p + a + ".to_q"))
            kin = C if a == "s_attn" else Dc
            sd.update(lin(C, kin, bias=False, name=p + a + ".to_k")); sd.update(lin(C, kin, bias=False, name=p + a + ".to_v"))
            sd.update(lin(C, C, name=p + a + ".to_out.0"))
        sd.update(lin(F_, C, name=p + "ff.net.0.proj")); sd.update(lin(C, F_, name=p + "ff.net.2"))
        if i > NL // 2:
            sd.update(norm(C, p + "norm_skip")); sd.update(lin(C, 2 * C, name=p + "linear_skip"))
    sd.update(norm(C, "norm_out")); sd.update(lin(Din, C, name="proj_out"))
    return sd


def source_sha():
    """sha256 over the kernel sources + header: ties a committed PMC summary to the build it was measured on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "actionmesh_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".h", ".inc")):
            h.update(fn.encode()); h.update(open(os.path.join(csrc, fn), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "actionmesh_amd.h"), "rb").read())
    return h.hexdigest()[:16]


def attention_roofline(T, N, H, dev, world=1, reps=3, dtype="bf16", groups=None):
    """Time the dominant kernel (inflated self-attention, both CFG samples) with HIP events on the
    launch stream.  One launch = one layer on one rank: this rank's T/world frames of queries
    against all T frames of keys; algorithmic flops per launch = 4 * (T*L/world) * (T*L) * (H*128) * B
    (SURVEY.md 8(d)).
    Two samples are reported: `plain` Q, K ~ N(0, 1) (scores ~ N(0, 1): what random-init weights with unit qk-norm gains
    give - the step's own regime, and the headline `achieved`), and `peaky` with Q scaled x4 (scores ~ N(0, 16^2), what
    trained qk-norm gains look like): the lazy re-base branch of the kernel then fires on most rows and a few workgroups
    go through the exact fallback, so the pair brackets the data-dependent cost of the product kernel."""
    from actionmesh_amd import ops
    if groups is None:
        groups = 2 if world % 2 == 0 else 1      # CFG branches split first (sharding.FrameShardPlan)
    fw = world // groups                         # frame shards per branch = key chunks
    B, L = 2 // groups, N + 1
    Sq = (T // fw) * L               # local query rows == keys per chunk
    sq_pad, sk_pad = ops.round_up(Sq, 256), ops.round_up(Sq, 64)
    g = torch.Generator(device=dev).manual_seed(0)
    Qf = torch.randn((B, H, sq_pad, 128), device=dev, generator=g)
    K = torch.randn((fw, B, H, sk_pad, 128), device=dev, generator=g).to(torch.bfloat16)
    Vt = torch.randn((fw, B, H, 128, sk_pad), device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty((B * Sq, H * 128), dtype=torch.bfloat16, device=dev)
    flops = 4.0 * Sq * (Sq * fw) * (H * 128) * B
    fast = dtype == "fp8_fast"
    if fast:
        dtype = "fp8"
    attn = ops.attention_fp8 if dtype == "fp8" else ops.attention
    if fast:
        attn = lambda *a, **k: ops.attention_fp8(*a, ablate=400, **k)        # the exponent-field probabilities (attn_dtype "fp8_fast")
    peak = PEAK_FP8_TFLOPS if dtype == "fp8" else PEAK_BF16_TFLOPS

    def sample(qscale, K=K, Vt=Vt):
        Q = (Qf * qscale).to(torch.bfloat16)          # fp8: one launch = quantise pass + attention kernel, as the model runs a layer
        attn(Q, K, Vt, Sq, Sq, out=out, nchunks=fw)
        torch.cuda.synchronize()
        f0 = ops.attention_fallback_count()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            attn(Q, K, Vt, Sq, Sq, out=out, nchunks=fw)
        e1.record()
        torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) / 1e3 / reps
        ach = flops / sec / 1e12
        assert bool(torch.isfinite(out.float()).all())
        return {"achieved": round(ach, 1), "frac": round(ach / peak, 4), "launch_ms": round(sec * 1e3, 3),
                "fallback_workgroups_per_launch": (ops.attention_fallback_count() - f0) / reps}

    plain, peaky = sample(1.0), sample(4.0)
    # Decomposition of `frac`: the firmware's own telemetry (gpu_metrics through amdsmi) sampled at ~20 Hz while the
    # plain launch runs back to back for ~1.5 s - the delivered gfx clock, the socket power, and how long the PPT (socket power) limiter
    # was active.  frac = pipe_busy x effective_clock / 2.4 GHz: only the first factor is the kernel's schedule.
    telemetry = None
    try:
        telemetry = clock_telemetry(lambda: attn((Qf * 1.0).to(torch.bfloat16), K, Vt, Sq, Sq, out=out, nchunks=fw), seconds=1.5)
    except Exception as e:          # no amdsmi / no sysfs in this container: the decomposition is simply absent
        telemetry = {"error": repr(e)[:200]}
    # the same launch on all-zero operands: nothing toggles, the chip keeps its full clock, and what is left is the instruction
    # stream's own rate (= the MFMA-busy fraction of the counters); the distance from `plain` to it is the power budget
    # (tools/clock_probe.py), not the schedule.  Diagnostic only: never `achieved`.
    zeros = sample(0.0, torch.zeros_like(K), torch.zeros_like(Vt)) if dtype == "bf16" else None      # (fp8: no scale for an all-zero head)
    # HBM traffic per launch: rocprofv3 --pmc passes of tools/gpu_profile.sh on the SAME launch shape; the committed summary
    # is quoted only when it names the kernel sources it was measured on and they are the ones built now - otherwise null.
    traffic, traffic_src = None, None
    import glob
    if world == 1 and dtype == "bf16":
        for tj in sorted(glob.glob(os.path.join(ROOT, "profiles", "*attention_traffic.json")), reverse=True):
            with open(tj) as f:
                rec = json.load(f)
            if rec.get("shape") == [T, N, H] and rec.get("source_sha") == source_sha():
                traffic, traffic_src = rec.get("traffic_bytes_per_launch"), "profiles/" + os.path.basename(tj)
                break
    name = "attn_fwd64_kernel" if dtype == "bf16" else ("attn_fp8p_kernel<FAST=1>" if fast else "attn_fp8p_kernel")
    return {"bound": "mfma", "kernel": f"{name} (inflated self-attention, 1 launch = 1 layer on this rank; timed with its "
                                       "exact-fallback grid and split-tail kernels)",
            "achieved": plain["achieved"], "peak": peak, "unit": "TFLOP/s",
            "frac": plain["frac"], "traffic": traffic, "traffic_source": traffic_src,
            "launch_ms": plain["launch_ms"], "flops_per_launch": flops,
            "effective_clock_ghz": (telemetry or {}).get("gfxclk_ghz_median"),
            "pipe_busy": (round(plain["frac"] * 2.4 / telemetry["gfxclk_ghz_median"], 4)
                          if telemetry and telemetry.get("gfxclk_ghz_median") else None),
            "energy_j": (telemetry or {}).get("energy_j_per_launch"),
            "pj_per_flop": (round(telemetry["energy_j_per_launch"] / flops * 1e12, 4)
                            if telemetry and telemetry.get("energy_j_per_launch") else None),
            "clock_telemetry": telemetry,
            "samples": {"plain (scores ~ N(0,1))": plain, "peaky (Q x4: scores ~ N(0,16^2))": peaky,
                        "zero operands (diagnostic: same launch, nothing toggles - the schedule's rate at the full clock)": zeros}}


def reference_full_shape(shape):
    """A MEASUREMENT (not a fit) of the reference's own unmodified modules at the full benchmark shape: the forward that produced
    tests/golden/full_<shape>.npz in the build container (oracle/make_golden_baseline.py make_full: ActionMeshDenoiser.forward, fp32,
    CPU, B = 2 CFG batch) stored its wall time and thread count in the fixture.  Another host than this box's - which is why it is
    reported beside the fit, not instead of it."""
    path = os.path.join(ROOT, "tests", "golden", f"full_{shape}.npz")
    if not os.path.exists(path):
        return None
    import numpy as np
    g = np.load(path)
    if "fwd_seconds_fp32" not in g.files:
        return None
    rec = {"seconds": round(float(g["fwd_seconds_fp32"]), 1), "threads": int(g["host_threads"]) if "host_threads" in g.files else None,
           "source": f"tests/golden/full_{shape}.npz",
           "what": "ONE fp32 forward (B=2) of the reference's own modules (+ diffusers shim) at this exact shape, timed when the parity "
                   "fixture was generated in the build container (not on this box)"}
    if "fwd_seconds_autocast" in g.files:
        rec["seconds_autocast_bf16"] = round(float(g["fwd_seconds_autocast"]), 1)
    rec["value"] = round(1.0 / float(g["fwd_seconds_fp32"]), 6)
    return rec


def clock_telemetry(fn, seconds=1.5):
    """Run `fn` back to back for `seconds` while a thread samples the firmware's gpu_metrics blob (amdsmi): the median delivered gfx
    clock over the XCDs, the socket power, and the share of the interval the PPT (socket power), thermal and PROCHOT limiters were
    active (residency accumulators of gpu_metrics v1.6+; tools/limiter_probe.py is the long form)."""
    import threading
    import amdsmi
    try:
        amdsmi.amdsmi_init()
    except Exception:
        pass
    h = amdsmi.amdsmi_get_processor_handles()[0]
    rows, stop = [], [False]

    def loop():
        while not stop[0]:
            try:
                rows.append(amdsmi.amdsmi_get_gpu_metrics_info(h))
            except Exception:
                pass
            time.sleep(0.05)
    fn(); torch.cuda.synchronize()
    first = amdsmi.amdsmi_get_gpu_metrics_info(h)
    th = threading.Thread(target=loop, daemon=True)
    th.start()
    t0, n = time.time(), 0
    while time.time() - t0 < seconds:
        for _ in range(4):
            fn()
        n += 4
        torch.cuda.synchronize()
    t1 = time.time()
    stop[0] = True
    th.join()
    last = amdsmi.amdsmi_get_gpu_metrics_info(h)
    rows = rows[len(rows) // 4:]
    num = lambda v: isinstance(v, (int, float)) and not isinstance(v, bool)
    med = lambda v: sorted(v)[len(v) // 2] if v else None
    clk = [sum(x for x in r["current_gfxclks"] if num(x)) / max(1, sum(1 for x in r["current_gfxclks"] if num(x)))
           for r in rows if isinstance(r.get("current_gfxclks"), list) and any(num(x) for x in r["current_gfxclks"])]
    out = {"seconds": seconds, "launches": n, "samples": len(rows),
           "gfxclk_ghz_median": round(med(clk) / 1e3, 3) if clk else None,
           "socket_power_w_median": med([r["current_socket_power"] for r in rows if num(r.get("current_socket_power"))])}
    dacc = (last.get("accumulation_counter") or 0) - (first.get("accumulation_counter") or 0) if num(last.get("accumulation_counter")) and num(first.get("accumulation_counter")) else 0
    if dacc > 0:
        for k in ("ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc", "prochot_residency_acc"):
            if num(first.get(k)) and num(last.get(k)):
                out[k.replace("_acc", "_share")] = round((last[k] - first[k]) / dacc, 4)
    try:
        cap = amdsmi.amdsmi_get_power_cap_info(h)
        out["power_cap_w"] = cap.get("power_cap", 0) / 1e6 if num(cap.get("power_cap")) else None
    except Exception:
        pass
    # energy per launch from the firmware's energy accumulator (15.259 uJ units) over the interval: what a power-capped kernel is
    # priced in - time = energy / cap (rank variants by joules, not cycles; tools/limiter_probe.py --energy-table)
    if num(first.get("energy_accumulator")) and num(last.get("energy_accumulator")) and n > 0 and t1 > t0:
        joules = (last["energy_accumulator"] - first["energy_accumulator"]) * 15.259e-6
        out["energy_j_per_launch"] = round(joules / n, 4)
        out["average_power_w"] = round(joules / (t1 - t0), 1)
    out["limiter"] = ("socket power (PPT)" if out.get("ppt_residency_share", 0) > 0.05 else
                      "thermal" if max(out.get("socket_thm_residency_share", 0), out.get("hbm_thm_residency_share", 0), out.get("vr_thm_residency_share", 0)) > 0.05
                      else "none seen")
    return out


def cpu_baseline(hp, sd, step_flops_full, S, T_full, N_full, deep=False, shape=None):
    """The reference CPU path (fp32 - the reference's cuda autocast is inert on CPU) timed on this box's host cores on a
    bounded sample and extrapolated to the full step the way SURVEY 8(d) prescribes: full 21-layer forwards at
    N in {256, 512, 1024} latent tokens per frame at the workload's own T = 16 and C (TL = T (N + 1) = 4112 / 8208 / 16400 tokens per
    sample, ~2 min of host time; round 2 sampled at T = 8 and extrapolated x8), a least-squares fit
    of  seconds = a * TL^2 + b * TL  (the attention term and the GEMM / elementwise term), evaluated at the workload's
    TL.  kind = "reference" when the reference's own modules are importable (build container: /root/reference + the
    diffusers shim), otherwise "port" (oracle/denoiser_oracle.py, the restatement pinned to the reference fixtures)."""
    from oracle import denoiser_oracle as O   # checker / baseline only, never on the product path
    cfg = O.OracleConfig(in_channels=hp["in_channels"], num_layers=hp["num_layers"],
                         num_attention_heads=hp["num_attention_heads"], width=hp["width"],
                         mlp_ratio=hp["mlp_ratio"], cross_attention_dim=hp["cross_attention_dim"],
                         inflated_layers=tuple(hp["inflated_layers"]))
    # pick the thread count that gives the best fp32 GEMM rate on this host (big multi-socket
    # boxes get slower when every hardware thread is used)
    best, cores = 0.0, 1
    a = torch.randn(2048, 1024); b = torch.randn(1024, 4096)
    for n in (8, 16, 32, 64, 128, os.cpu_count() or 1):
        if n > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(n)
        a @ b
        t0 = time.perf_counter()
        for _ in range(3):
            a @ b
        r = 3 * 2 * 2048 * 1024 * 4096 / (time.perf_counter() - t0)
        if r > best:
            best, cores = r, n
    torch.set_num_threads(cores)
    kind, ref_model = "port", None
    if os.path.isdir("/root/reference/actionmesh"):
        try:
            sys.path.insert(0, os.path.join(ROOT, "oracle", "diffusers_shim")); sys.path.insert(0, "/root/reference")
            from actionmesh.model.temporal_denoiser import ActionMeshDenoiser
            ref_model = ActionMeshDenoiser(num_tokens_nominal=N_full, temporal_context_size=T_full, clear_autocast=False, **hp)
            ref_model.load_state_dict(sd)
            ref_model.eval()
            kind = "reference"
        except Exception:
            ref_model = None
    Ts = 16          # the workload's own frame count: TL = 4112, 8208, 16400 - the fit is evaluated x4 beyond the largest sample
    pts = []
    with torch.no_grad():
        for Ns in (64, 256, 512, 1024) + ((2048,) if deep else ()):          # 64: untimed warm-up of the thread pool / allocator
            g = torch.Generator().manual_seed(0)
            x = torch.randn(2, Ts, Ns, hp["in_channels"], generator=g)
            c = torch.randn(2, Ts, S, hp["cross_attention_dim"], generator=g)
            c[0] = 0
            fs = torch.arange(Ts, dtype=torch.float32)[None].repeat(2, 1)
            m = torch.zeros(2, Ts); m[:, 0] = 1
            t = torch.tensor([700.0, 700.0])
            t0 = time.perf_counter()
            if ref_model is not None:
                ref_model.forward(hidden_states=x, context=c, framestep=fs, diffusion_time=t, mask=m, freqs_rot=None)
            else:
                O.denoiser_forward(sd, cfg, x, c, fs, t, m, "fp32")
            sec = time.perf_counter() - t0
            if Ns > 64:
                pts.append((Ts * (Ns + 1), sec, O.step_flops(2, Ts, Ns, cfg, S)))
    # least squares for sec = a TL^2 + b TL
    s40 = sum(p[0] ** 4 for p in pts); s30 = sum(p[0] ** 3 for p in pts); s20 = sum(p[0] ** 2 for p in pts)
    y2 = sum(p[1] * p[0] ** 2 for p in pts); y1 = sum(p[1] * p[0] for p in pts)
    det = s40 * s20 - s30 * s30
    qa, qb = (y2 * s20 - y1 * s30) / det, (s40 * y1 - s30 * y2) / det
    TLf = T_full * (N_full + 1)
    sec_full = qa * TLf * TLf + qb * TLf
    flat = sum(p[2] for p in pts) / sum(p[1] for p in pts)
    resid = max(abs(qa * p[0] ** 2 + qb * p[0] - p[1]) / p[1] for p in pts)      # worst relative misfit at the samples
    quad_share = qa * TLf * TLf / sec_full
    ns_txt = ", ".join(str(p[0] // Ts - 1) for p in pts)
    tl_txt = " / ".join(str(p[0]) for p in pts)
    sample_txt = (f"{'reference modules + diffusers shim' if kind == 'reference' else 'oracle (port)'} fp32, full "
                  f"{hp['num_layers']}-layer width-{hp['width']} forwards at B=2, T={Ts}, N in ({ns_txt}) (TL = {tl_txt}) = "
                  f"{sum(p[1] for p in pts):.1f} s of CPU work at {flat / 1e12:.2f} TFLOP/s on {cores} threads")
    # ---- ONE stated number (round 5).  The reference's own modules were timed ONCE at this exact shape - in the
    # build container, when the parity fixture was generated (reference_full_shape) - and the same port forwards timed above were timed
    # there too (oracle/cpu_rate_build_container.json).  value = 1 / (that measurement x container rate / this box's rate at the
    # largest common sample): a measurement of the full step moved to this box's cores by a measured ratio.  The a TL^2 + b TL fit
    # (x4 extrapolation in TL, 14 % residual in round 4) stays as a secondary field; the two agree to a few per cent.
    ref_full = reference_full_shape(shape) if shape else None
    value, derivation = 1.0 / sec_full, "fit"
    rate_path = os.path.join(ROOT, "oracle", "cpu_rate_build_container.json")
    scaled = None
    if ref_full and kind == "port" and os.path.exists(rate_path):
        with open(rate_path) as f:
            cr = json.load(f)
        common = [(q, p) for q in cr["points"] for p in pts if q["TL"] == p[0]]
        if cr.get("shape") == shape and common:
            q, p_here = max(common, key=lambda c: c[0]["TL"])
            ratio = (q["seconds"] / p_here[1])              # this box is `ratio` x faster than the build container on the same forward
            scaled = {"reference_seconds_build_container": ref_full["seconds"], "build_container_threads": cr["threads"],
                      "rate_ratio_this_box_over_build_container": round(ratio, 3), "at_TL": q["TL"],
                      "build_container_seconds_at_TL": q["seconds"], "this_box_seconds_at_TL": round(p_here[1], 3),
                      "seconds_per_step_on_this_box": round(ref_full["seconds"] / ratio, 1)}
            value, derivation = ratio / ref_full["seconds"], "reference forward at the exact shape, scaled by the measured rate ratio"
    return {"value": value, "unit": "denoise-steps/s", "cores": cores, "kind": kind, "derivation": derivation,
            "reference_full_shape": ref_full, "scaled_reference": scaled,
            "fit": {"model": "seconds = a*TL^2 + b*TL per CFG-batched forward, TL = T*(N+1)", "a": qa, "b": qb, "value": 1.0 / sec_full,
                    "max_relative_residual": round(resid, 4), "quadratic_share_at_workload": round(quad_share, 3),
                    "points": [{"TL": p[0], "seconds": round(p[1], 3), "tflops": round(p[2] / p[1] / 1e12, 3)} for p in pts]},
            "sample": sample_txt + (
                f"; value = the reference's own modules' forward at the exact shape ({ref_full['seconds']:.0f} s on {scaled['build_container_threads']} "
                f"threads of the build container) / {scaled['rate_ratio_this_box_over_build_container']} (this box's rate over the container's on the "
                f"TL = {scaled['at_TL']} forward, both measured with the port) = {scaled['seconds_per_step_on_this_box'] / 60:.1f} min per step; "
                f"secondary: a*TL^2+b*TL fit at TL={TLf}: {sec_full / 60:.1f} min (x{TLf / pts[-1][0]:.0f} extrapolation)"
                if scaled else
                f"; a*TL^2+b*TL fit evaluated at TL={TLf}: {sec_full / 60:.1f} min per step - an extrapolation (x{TLf / pts[-1][0]:.0f} in TL "
                f"beyond the largest sample), not a measurement of the full step")}


XGMI_LINK_GBS = 153.0          # one direction of one xGMI link (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU)


def emulate_world(P, shape, dtype, dev, steps=3):
    """A PROJECTION, not a measurement of N GPUs: rank 0's share of one step of a P-rank run, timed on ONE device - its CFG branch,
    its frames, the Python / ctypes phase loop of sharding.sharded_forward, the two-pass attention over `frame_world` key chunks
    with the remote shards pre-filled (copies of the local shard: realistic values, no exchange), the flow step on its frames.
    Beside it, the modelled link time of the per-layer exchange (every peer pushes its shard over its own xGMI link, so the
    all-gather costs one shard at one link's rate) - what the overlap has to hide.  Exposes the host-loop and small-M GEMM costs
    a real 8-GPU node would see, without one."""
    from actionmesh_amd import ops
    from actionmesh_amd.denoiser import HipEngine, masked_time, rope_tables_host
    from actionmesh_amd.sharding import FrameShardPlan
    T, N, C, H, NL, S, Dc, Din = SHAPES[shape]
    hp = dict(in_channels=Din, num_layers=NL, num_attention_heads=H, width=C, mlp_ratio=4.0, cross_attention_dim=Dc,
              inflated_layers=list(range(NL)))
    plan = FrameShardPlan(T, P, 0, batch=2, cfg_groups=2 if P % 2 == 0 else 1)
    fw, tl, bl = plan.frame_world, plan.frames_local, plan.batch_local
    eng = HipEngine(hp, random_state_dict(hp, seed=0), dev, bl, tl, N, S, world=fw, rank=0, attn_dtype=dtype)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(bl, tl, N, Din, generator=g).to(dev)
    lat = x[0].clone()
    ctx = torch.randn(bl, tl, S, Dc, generator=g).to(dev)
    cos, sin = rope_tables_host(torch.arange(tl, dtype=torch.float32).repeat(bl, 1), 128)
    eng.set_context(ctx, cos, sin)
    t_bt = [640.0] * (bl * tl)
    kv = eng.kv_buffers()[0] if fw > 1 else None

    def step(fill):
        eng.begin(x, t_bt)
        for i in range(NL):
            eng.layer_pre(i)
            if fw > 1:
                if fill:                                   # once: the "remote" shards = copies of the local one
                    for r in range(1, fw):
                        kv[r].copy_(kv[0])
                eng.layer_attn_local(i)
            eng.layer_post(i)
        v = eng.end()
        vv = torch.cat([v, v], 0) if bl == 1 else v        # the partner branch's velocity would arrive by a 2 MB exchange
        ops.flow_step(vv.reshape(2, tl, N, Din), lat, [7.5], 0.02, True, [True] * tl)

    step(True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step(False)
    host_s = (time.perf_counter() - t0) / steps            # host time to ENQUEUE a step (the GPU is still running)
    torch.cuda.synchronize(dev)
    sec = (time.perf_counter() - t0) / steps
    L = N + 1
    shard_bytes = bl * tl * L * C * 2 * (1 if dtype.startswith("fp8") else 2)
    link_ms = shard_bytes / (XGMI_LINK_GBS * 1e9) * 1e3 if fw > 1 else 0.0
    n8, n16 = eng.attention_counters()
    eng.close()
    return {"projection": True, "world": P, "cfg_groups": plan.cfg_groups, "frame_shards": fw, "frames_per_rank": tl,
            "rank0_ms_per_step": round(sec * 1e3, 2), "rank0_host_enqueue_ms_per_step": round(host_s * 1e3, 2),
            "exchange_bytes_per_layer_per_rank": shard_bytes, "modelled_link_ms_per_layer": round(link_ms, 3),
            "modelled_link_ms_per_step_if_not_hidden": round(link_ms * NL, 2),
            "projected_steps_per_s_if_exchange_hidden": round(1.0 / sec, 3), "attention_launches": {"fp8": n8, "bf16": n16},
            "what": "rank 0's share of a step on ONE device (remote K/V shards pre-filled, no exchange); link time modelled at one "
                    f"xGMI link ({XGMI_LINK_GBS:.0f} GB/s) per peer push; not a multi-GPU measurement"}


def nominal_record(dev, dtype, steps=3):
    """The shipped architecture (actionmesh.yaml:33-43: 16 frames x 2048 tokens, width 2048, 16 heads) for a few steps, every
    algorithmic operation executed - reported beside the headline line, never as `value`."""
    from actionmesh_amd import ClassifierFreeGuidance, HipDenoiser, HipSchedulerFlow
    T, N, C, H, NL, S, Dc, Din = SHAPES["nominal"]
    hp = dict(in_channels=Din, num_layers=NL, num_attention_heads=H, width=C, mlp_ratio=4.0, cross_attention_dim=Dc,
              inflated_layers=list(range(NL)))
    model = HipDenoiser(num_tokens_nominal=N, temporal_context_size=T, attn_dtype=dtype, **hp)
    model.load_state_dict(random_state_dict(hp, seed=0))
    model.to(dev).eval()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, T, N, Din, generator=g).to(dev)
    ctx = torch.randn(1, T, S, Dc, generator=g).to(dev)
    mask = torch.zeros(1, T); mask[0, 0] = 1.0
    sched = HipSchedulerFlow(num_inference_steps=50, shift=3.0, is_additive=True, exact_shortcuts=False)
    loop = sched._flow_sample(model, ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5]), x, ctx, device=dev, mask=mask.to(dev),
                              framestep=torch.arange(T, dtype=torch.float32)[None])
    next(loop)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        next(loop)
    torch.cuda.synchronize(dev)
    sec = (time.perf_counter() - t0) / steps
    flops = model._engine.step_flops(2, T, N, S)
    rec = {"workload": f"nominal: B=2 x T={T} x N={N}, width {C} ({H} heads), {NL} layers", "steps": steps, "ms_per_step": round(sec * 1e3, 2),
           "value": round(1.0 / sec, 4), "step_flops": flops, "step_frac_of_bf16_peak": round(flops / sec / 1e12 / PEAK_BF16_TFLOPS, 4)}
    model.to("cpu")
    del model
    torch.cuda.empty_cache()
    return rec


PREFLIGHT_HP = dict(in_channels=64, num_layers=2, num_attention_heads=2, width=256, mlp_ratio=4.0, cross_attention_dim=64,
                    inflated_layers=[0, 1])          # the `small` plumbing width, two layers: the exchange's sequence flags turn once


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launcher_argv(argv, n, port):
    """The command `python bench.py --gpus N ...` turns itself into when nobody wrapped it in a launcher: the driver's own N > 1
    command (task statement), same arguments."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) started WITHOUT torch.distributed.run: re-execute under it - one rank
    per GPU - hand rank 0's ONE JSON line through as the LAST line of stdout (everything else the ranks print goes to stderr), and
    return the launcher's exit code.  Fewer than N visible devices: a one-line JSON error, exit code 2.  With the default back-end
    choice (`--exchange ab`) a run that died without a line is repeated once per single back-end (a hung RCCL cannot be recovered
    inside the process it hung; a fresh set of processes can run the copy-engine exchange) and the line says which attempt it is."""
    import subprocess
    n = args.gpus
    if not args.same_device:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print(json.dumps({"error": f"bench.py --gpus {n}: {have} visible device(s)", "metric": "denoise-steps/sec", "value": None,
                              "n_gpus": n, "visible_devices": have}), flush=True)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    attempts = [list(argv)]
    if args.exchange in (None, "ab") and not os.environ.get("ACTIONMESH_AMD_EXCHANGE") and not args.same_device:
        attempts += [list(argv) + ["--exchange", "peer"], list(argv) + ["--exchange", "rccl"]]
    history, rc = [], 1
    for k, av in enumerate(attempts):
        cmd = launcher_argv(av, n, free_port())
        t0 = time.time()
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
        line = None
        for out in proc.stdout:                      # the ranks' stderr is inherited: it streams through untouched
            if out.startswith("{") and '"metric"' in out:
                line = out.strip()
            else:
                sys.stderr.write(out)
        rc = proc.wait()
        history.append({"argv": av, "returncode": rc, "seconds": round(time.time() - t0, 1), "line": line is not None})
        if line is not None:
            try:
                rec = json.loads(line)
                rec["launcher"] = {"self_launched": True, "command": " ".join(cmd[:cmd.index(os.path.abspath(__file__))] + ["bench.py"] + av),
                                   "attempts": history}
                line = json.dumps(rec)
            except ValueError:
                pass
            print(line, flush=True)
            return rc
    print(json.dumps({"error": "bench.py: no attempt produced a result line", "metric": "denoise-steps/sec", "value": None, "n_gpus": n,
                      "launcher": {"self_launched": True, "attempts": history}}), flush=True)
    return rc or 1


def preflight_plan(world, cfg_parallel):
    """Frames of the pre-flight problem: two per frame shard, so that every rank of an N-rank run exchanges with its peers."""
    groups = 2 if (cfg_parallel and world % 2 == 0) else 1
    fw = world // groups
    return max(4, 2 * fw)


def preflight_child(args):
    """One rank of the PRE-FLIGHT of one exchange back-end (`bench.py --preflight-child rccl|peer`, started by every rank of the real
    run as a killable child process with its own rendezvous port): the sharded sampler on a seconds-long problem - plumbing width,
    two layers, two frames per shard, three steps - through exactly the classes the legs use (HipDenoiser + process group,
    HipSchedulerFlow, the per-layer [K | V^T] exchange), compared on rank 0 with the unsharded run of the same steps.  Prints one
    JSON verdict line.  A back-end that hangs here is killed with the child; the real run then does not spend a headline warm-up on it."""
    import torch.distributed as dist
    from datetime import timedelta
    from actionmesh_amd import ClassifierFreeGuidance, HipDenoiser, HipSchedulerFlow
    name = args.preflight_child
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    t0 = time.time()
    os.environ["ACTIONMESH_AMD_EXCHANGE"] = name
    if name == "rccl" and args.same_device:
        print(json.dumps({"preflight": name, "ok": False, "rank": rank, "error": "RCCL refuses two ranks on one device (--same-device)"}), flush=True)
        return 0
    backend = "nccl" if name == "rccl" else "gloo"          # the copy-engine back-end makes no RCCL call at all
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{args.pf_port}", rank=rank, world_size=world,
                            timeout=timedelta(seconds=max(30.0, args.preflight_timeout)))
    hp = dict(PREFLIGHT_HP)
    T, N, S = preflight_plan(world, args.cfg_parallel), 512, 17
    sd = random_state_dict(hp, seed=1)
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(1, T, N, hp["in_channels"], generator=g)
    ctx = torch.randn(1, T, S, hp["cross_attention_dim"], generator=g).to(dev)
    mask = torch.zeros(1, T); mask[0, 0] = 1.0
    fs = torch.arange(T, dtype=torch.float32)[None]
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])

    def run(group):
        model = HipDenoiser(num_tokens_nominal=N, temporal_context_size=T, process_group=group, attn_dtype=args.dtype,
                            cfg_parallel=bool(args.cfg_parallel), **hp)
        model.load_state_dict(sd)
        model.to(dev).eval()
        lat = x0.clone().to(dev)
        sched = HipSchedulerFlow(num_inference_steps=50, shift=3.0, is_additive=True, exact_shortcuts=False)
        loop = sched._flow_sample_impl(model, cfgd, lat, ctx, device=dev, mask=mask.to(dev), framestep=fs, local_latents=group is not None)
        for _ in range(3):
            next(loop)
        torch.cuda.synchronize(dev)
        model.check_exchange(block=True)
        if group is not None:
            model.gather_latent_frames(lat[0], 2)
        out = lat.detach().float().cpu().clone()
        ex = getattr(model._engine, "exchange", None)
        fine = getattr(ex, "flags_fine_grained", None)
        model._engine.close()
        return out, fine

    rec = {"preflight": name, "rank": rank, "world": world, "frames": T}
    try:
        sharded, fine = run(dist.group.WORLD)
        rec["ok"] = bool(torch.isfinite(sharded).all())
        if fine is not None:
            rec["flags_fine_grained"] = bool(fine)
        if rank == 0:
            single, _ = run(None)
            d = float((sharded.double() - single.double())[0, 1:].norm() / single.double()[0, 1:].norm())
            rec["rel_l2_vs_single_rank"], rec["tol"] = round(d, 6), 3e-2
            rec["ok"] = rec["ok"] and d <= 3e-2
    except Exception as e:                           # noqa: BLE001 - the verdict
        rec["ok"], rec["error"] = False, f"{type(e).__name__}: {str(e)[:300]}"
    rec["seconds"] = round(time.time() - t0, 2)
    print(json.dumps(rec), flush=True)
    sys.stdout.flush()
    # the ranks leave TOGETHER: rank 0 is still busy with its single-rank comparison when the others are done, and a communicator whose
    # peers have vanished is one more thing that could go wrong on first contact with a real node; the wait is bounded
    try:
        from torch.distributed import distributed_c10d as c10d
        st = c10d._get_default_store()
        if rank == 0:
            st.set("am_preflight_done", "1")
        else:
            st.wait(["am_preflight_done"], timedelta(seconds=60))
    except Exception:                                # noqa: BLE001 - the verdict is already out
        pass
    os._exit(0)          # no destroy_process_group: a half-dead back-end must not keep the verdict from leaving


def run_preflight(order, args, rank, world, store, argv):
    """Every rank of the real run starts ONE child per back-end (`--preflight-child`, rendezvous on a fresh port that rank 0 picks and
    publishes on the store), waits at most --preflight-timeout seconds, kills what has not answered (process group and all), and the
    ranks agree over the store: a back-end passes only if every rank's child said ok.  Returns name -> verdict record."""
    import signal
    import subprocess
    from actionmesh_amd.sharding import store_gather
    verdicts = {}
    for k, name in enumerate(order):
        if rank == 0:
            store.set(f"pf_port/{k}", str(free_port()))
        port = int(store.get(f"pf_port/{k}").decode())
        env = {e: v for e, v in os.environ.items() if not e.startswith("TORCHELASTIC")}
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--preflight-child", name, "--pf-port", str(port),
               "--dtype", args.dtype, "--cfg-parallel", str(args.cfg_parallel), "--preflight-timeout", str(args.preflight_timeout)]
        if args.same_device:
            cmd.append("--same-device")
        t0 = time.time()
        rec = None
        try:
            proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
            try:
                out, err = proc.communicate(timeout=args.preflight_timeout)
                for ln in out.splitlines():
                    if ln.startswith("{") and '"preflight"' in ln:
                        rec = json.loads(ln)
                if rec is None:
                    rec = {"ok": False, "error": f"child exited {proc.returncode} without a verdict: {err.strip()[-300:]}"}
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    proc.kill()
                proc.communicate()
                rec = {"ok": False, "error": f"no verdict after {args.preflight_timeout:.0f} s: child killed"}
        except Exception as e:                       # noqa: BLE001
            rec = {"ok": False, "error": f"{type(e).__name__}: {str(e)[:200]}"}
        rec["seconds"] = round(time.time() - t0, 1)
        mine = json.dumps(rec)
        everyone = store_gather(store, f"pf/{k}/{name}", rank, world, mine, args.preflight_timeout + 30.0)
        recs = [json.loads(v) if v else {"ok": False, "error": "no verdict"} for v in everyone]
        bad = [r for r, v in enumerate(recs) if not v.get("ok")]
        verdicts[name] = {"ok": not bad, "seconds": max(v.get("seconds", 0) for v in recs),
                          **({"rel_l2_vs_single_rank": recs[0].get("rel_l2_vs_single_rank")} if recs[0].get("rel_l2_vs_single_rank") is not None else {}),
                          **({"flags_fine_grained": recs[0]["flags_fine_grained"]} if "flags_fine_grained" in recs[0] else {}),
                          **({"failed_ranks": bad, "error": str(recs[bad[0]].get("error"))[:300]} if bad else {})}
    return verdicts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shape", default="headline", choices=list(SHAPES))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8", "fp8_fast"],
                    help="fp8: the inflated self-attention (QK^T, P.V) on the e4m3 MX-scaled MFMA kernel - BASELINE.json "
                         "configs[4] (use with --shape long64); GEMMs, norms and the residual stream stay bf16.  The headline "
                         "metric is bf16.  fp8_fast: the same with the exponent-field form of the probabilities (no v_exp; reported "
                         "as dtype fp8 with `attention_probabilities` saying so)")
    ap.add_argument("--graph", action="store_true", help="single GPU: the forward through a captured HIP graph (am_denoise_forward_graph)")
    ap.add_argument("--emulate-world", type=int, default=0, metavar="P", help="single GPU only: print a PROJECTION record instead - rank 0's share of "
                    "a step of a P-rank run timed on this device, beside the modelled xGMI time of the per-layer exchange")
    ap.add_argument("--no-nominal", action="store_true", help="skip the `nominal` sub-record (the shipped architecture, 3 steps) of the headline N=1 line")
    ap.add_argument("--same-device", action="store_true",
                    help="DRY RUN of the N > 1 path on a box with ONE GPU: every rank drives cuda:0, the control plane is gloo and the "
                         "per-layer K/V exchange is the copy-engine back-end (ACTIONMESH_AMD_EXCHANGE=peer; RCCL refuses two ranks on one "
                         "device).  Executes every line of the world > 1 branch - groups, the sharded HipDenoiser, barriers, the MAX "
                         "all-reduce of the timing, the JSON - so that a first run on an 8-GPU node cannot fail on a typo; the numbers "
                         "are NOT a multi-GPU measurement and the line says so (tests/test_multi_gpu.py)")
    ap.add_argument("--gather-every-step", action="store_true",
                    help="N > 1: all-gather the velocity on every rank in every step (the round-3 sampler) instead of keeping the latents sharded")
    ap.add_argument("--cpu-baseline-deep", action="store_true",
                    help="cpu_baseline: add the N = 2048 sample (TL = 32 784; minutes of host time) so the fit is evaluated x2 instead of "
                         "x4 beyond its largest sample.  Off by default: the default run has to finish within a few minutes")
    ap.add_argument("--exchange", default=None, choices=["ab", "rccl", "peer"],
                    help="N > 1: the per-layer [K | V^T] exchange back-end.  Default `ab`: BOTH back-ends inside this one invocation - first the "
                         "RCCL (`rccl`: all_gather_into_tensor on the nccl backend), then the copy-engine one (`peer`: SDMA pushes between IPC-mapped "
                         "buffers, gloo control plane: no RCCL call at all) - `value` is the faster leg that completed, both timings are in "
                         "`exchange_ab`, and a leg that raises or outlives --leg-timeout is reported there instead of killing the run "
                         "(VERDICT r04 next #2).  The ACTIONMESH_AMD_EXCHANGE environment variable, when set, selects one leg.")
    ap.add_argument("--cfg-parallel", type=int, default=1, choices=[0, 1],
                    help="N > 1: 1 (default) = the two guidance branches go to the two halves of the ranks, frames are sharded inside a half; "
                         "0 = north_star's pure frame sharding (every rank computes both branches of T / N frames)")
    ap.add_argument("--leg-timeout", type=float, default=420.0,
                    help="N > 1, second leg of --exchange ab: seconds before a watchdog prints the line with the completed leg and exits")
    ap.add_argument("--no-fingerprint-check", action="store_true",
                    help="N > 1: skip the single-rank re-run of the same steps on rank 0 that the sharded latents are compared with")
    ap.add_argument("--no-preflight", action="store_true",
                    help="N > 1: skip the seconds-long pre-flight of the exchange back-ends (killable child processes on a plumbing-size "
                         "problem, verdicts in `exchange_ab.preflight`) that keeps a dead back-end from costing a headline warm-up")
    ap.add_argument("--preflight-timeout", type=float, default=120.0, help="seconds a pre-flight child may take before it is killed")
    ap.add_argument("--preflight-child", default=None, choices=["rccl", "peer"], help=argparse.SUPPRESS)
    ap.add_argument("--pf-port", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.graph:
        os.environ["ACTIONMESH_AMD_GRAPH"] = "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # started bare (`python bench.py --gpus N`): become the launcher of N ranks of this same command
        raise SystemExit(self_launch(args, sys.argv[1:]))
    if args.same_device or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # multi-process GPU work on these hosts needs dmabuf IPC (RCCL and the copy-engine exchange alike); exported on the boxes already,
        # set here too - before the first HIP call of the process - so that a launcher with a scrubbed environment does not lose it
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    if args.preflight_child:
        return preflight_child(args)

    import torch.distributed as dist
    from datetime import timedelta
    from actionmesh_amd import ClassifierFreeGuidance, HipDenoiser, HipSchedulerFlow

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.same_device:
        local_rank = 0
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    # ---- process groups.  WORLD: nccl (= RCCL; lazy communicator creation on the current device - no device_id: eager init would create
    # the CFG-branch sub-groups by communicator split, which not every RCCL build supports), gloo for --same-device (RCCL refuses two
    # ranks on one device).  `ctl`: a gloo group of all ranks for everything that is not the data path - barriers, the MAX of the
    # timings, the per-leg verdicts - so that a broken RCCL cannot take the control plane with it.
    # The per-leg VERDICTS travel on neither: they are keys of the rendezvous store (sharding.run_exchange_legs), and every leg gets a
    # fresh gloo group with an explicit timeout for its own barriers (`ctl` below is re-bound per leg) - ADVICE r05.
    ctl, ctl_main, store = None, None, None
    if world > 1:
        dist.init_process_group("gloo" if args.same_device else "nccl")
        ctl_main = dist.new_group(backend="gloo")          # barriers outside the legs (default timeout: rank 0's single-rank re-run is long)
        ctl = ctl_main
        from actionmesh_amd.sharding import leg_store
        store = leg_store()

    def barrier():
        dist.barrier(group=ctl)

    def max_over_ranks(seconds):
        tmax = torch.tensor([seconds], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=ctl)
        return float(tmax.item())

    T, N, C, H, NL, S, Dc, Din = SHAPES[args.shape]
    if args.emulate_world:
        assert world == 1, "--emulate-world runs on one device"
        print(json.dumps({"metric": f"PROJECTION: rank-0 share of a {args.emulate_world}-rank denoise step ({T}f x {N}tok)", "shape": args.shape,
                          "dtype": args.dtype, **emulate_world(args.emulate_world, args.shape, args.dtype, dev, steps=args.steps)}), flush=True)
        return
    hp = dict(in_channels=Din, num_layers=NL, num_attention_heads=H, width=C, mlp_ratio=4.0,
              cross_attention_dim=Dc, inflated_layers=list(range(NL)))
    sd = random_state_dict(hp, seed=0)
    g = torch.Generator().manual_seed(0)
    init_latent0 = torch.randn(1, T, N, Din, generator=g)
    context = torch.randn(1, T, S, Dc, generator=g).to(dev)
    mask = torch.zeros(1, T); mask[0, 0] = 1.0
    framestep = torch.arange(T, dtype=torch.float32)[None]
    total = args.warmup + args.steps
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    cfg_groups = 2 if (args.cfg_parallel and world % 2 == 0) else 1
    # N > 1: the sampler keeps the latents sharded by frames across the steps, as HipSchedulerFlow.denoise does (no velocity gather
    # per step; --gather-every-step restores it for A/B); the frames are gathered once, outside the timed region, for the fingerprint
    local_latents = world > 1 and not args.gather_every_step

    def sync(collective=True):
        torch.cuda.synchronize(dev)
        if world > 1 and collective:
            barrier()
            torch.cuda.synchronize(dev)

    def run_leg(exchange, group):
        """warmup + `steps` timed steps of the sampler on a fresh HipDenoiser (process group `group`, exchange back-end `exchange`), then
        the same with the product's exact shortcuts.  Returns the timings, the final latents (host) and the engine's own counters."""
        coll = group is not None                    # rank 0's single-rank re-run (group None) must not enter a collective
        if exchange is not None:
            os.environ["ACTIONMESH_AMD_EXCHANGE"] = exchange
        if exchange == "rccl" and args.same_device:
            raise RuntimeError("RCCL refuses two ranks on one device (--same-device): this leg cannot run here")
        model = HipDenoiser(num_tokens_nominal=N, temporal_context_size=T, process_group=group, attn_dtype=args.dtype,
                            cfg_parallel=bool(args.cfg_parallel), **hp)
        model.load_state_dict(sd)
        model.to(dev).eval()
        init_latent = init_latent0.clone().to(dev)
        # `value` is measured with every algorithmic operation executed (exact_shortcuts=False); the product default - two exact,
        # bit-identical shortcuts of the CFG batch (include/actionmesh_amd.h am_set_branch_hints) - is timed behind it and reported
        # next to it as `with_exact_shortcuts`, never as `value`
        sched = HipSchedulerFlow(num_inference_steps=max(50, total), shift=3.0, is_additive=True, exact_shortcuts=False)
        loop = sched._flow_sample_impl(model, cfgd, init_latent, context, device=dev, mask=mask.to(dev),
                                       framestep=framestep, local_latents=local_latents and group is not None)
        for _ in range(args.warmup):
            next(loop)
        sync(coll)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            next(loop)
        sync(coll)
        elapsed = time.perf_counter() - t0
        if world > 1 and group is not None:
            elapsed = max_over_ranks(elapsed)
        model.check_exchange(block=True)
        if local_latents and group is not None:
            model.gather_latent_frames(init_latent[0], 2)
        assert bool(torch.isfinite(init_latent).all()), "non-finite latents"
        final = init_latent.detach().float().cpu().clone()
        sched2 = HipSchedulerFlow(num_inference_steps=max(50, total), shift=3.0, is_additive=True, exact_shortcuts=True)
        loop2 = sched2._flow_sample_impl(model, cfgd, init_latent, context, device=dev, mask=mask.to(dev), framestep=framestep,
                                         local_latents=local_latents and group is not None)
        next(loop2)
        sync(coll)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            next(loop2)
        sync(coll)
        elapsed2 = time.perf_counter() - t0
        if world > 1 and group is not None:
            elapsed2 = max_over_ranks(elapsed2)
        model.check_exchange(block=True)
        n_fp8, n_bf16 = model._engine.attention_counters()
        ex = getattr(model._engine, "exchange", None)
        out = {"elapsed": elapsed, "elapsed2": elapsed2, "final": final, "step_flops": model._engine.step_flops(2, T, N, S),
               "n_fp8": n_fp8, "n_bf16": n_bf16, "flags_fine_grained": getattr(ex, "flags_fine_grained", None)}
        model._engine.close()
        del model
        torch.cuda.empty_cache()
        return out

    def fingerprint_of(final):
        # what the sampler has made of the latents after warmup + steps steps: every rank holds the full tensor (like the reference)
        fp_lat = final[0, 1:].double()
        return {"after_steps": total, "rms": float(fp_lat.pow(2).mean().sqrt()), "mean": float(fp_lat.mean()),
                "sample": [round(float(x), 6) for x in final[0, -1, ::max(1, N // 8), 0].double()[:8]]}

    def build_result(legs, exchange_ab, fingerprint_check):
        best = min(legs.values(), key=lambda r: r["elapsed"]) if legs else None
        return _result(best, legs, exchange_ab, fingerprint_check)

    def _result(best, legs, exchange_ab, fingerprint_check):
        elapsed, elapsed2, step_flops = best["elapsed"], best["elapsed2"], best["step_flops"]
        steps_per_s = args.steps / elapsed
        # the arithmetic type the inflated self-attention REALLY ran in (am_attention_counters), not the one that was asked for
        n_fp8, n_bf16 = best["n_fp8"], best["n_bf16"]
        ran = "fp8" if (n_fp8 > 0 and n_bf16 == 0) else "bf16" if n_fp8 == 0 else f"mixed (fp8 x{n_fp8}, bf16 x{n_bf16})"
        if ran != ("fp8" if args.dtype.startswith("fp8") else args.dtype):
            raise SystemExit(f"bench.py: --dtype {args.dtype} was requested but the engine's self-attention ran in {ran}")
        result = {
            "metric": f"denoise-steps/sec ({T}f x {N}tok)", "value": round(steps_per_s, 4),
            "unit": "denoise-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": ran, "data": "synthetic",
            "config": {"workload": f"{args.shape}: Stage-I denoise step, B=2 (CFG) x T={T} frames x N={N} tokens, "
                                   f"width {C} ({H} heads x 128), {NL} layers all inflated, S={S} ctx tokens, "
                                   "random-init weights, seeded N(0,1) latents/context resident in HBM"
                                   + ("; self-attention in fp8 e4m3 (everything else bf16)" if ran == "fp8" else ""),
                       "parallelism": ("single GPU" if world == 1 else f"cfg-branch x{cfg_groups} * frame-shard x{world // cfg_groups}"),
                       "step_flops": step_flops},
            "step_tflops_per_gpu": round(step_flops * steps_per_s / world / 1e12, 1),
            "step_frac_of_bf16_peak": round(step_flops * steps_per_s / world / 1e12 / PEAK_BF16_TFLOPS, 4),
            "step_frac_of_dtype_peak": round(step_flops * steps_per_s / world / 1e12 / (PEAK_FP8_TFLOPS if ran == "fp8" else PEAK_BF16_TFLOPS), 4),
            "attention_launches": {"fp8": n_fp8, "bf16": n_bf16},
            "attention_probabilities": ("exponent-field e4m3 bytes, p = 2^n (1 + f) (fp8_fast: no transcendental instruction)" if args.dtype == "fp8_fast"
                                        else "exp2, rounded to e4m3" if ran == "fp8" else "exp2, rounded to bf16"),
            # second half of BASELINE.json's metric: needs the pretrained checkpoints (facebook/ActionMesh, TripoSG, RMBG) and a
            # real video, none reachable offline - not measured here, and nothing in `value` stands in for it
            "with_exact_shortcuts": {"ms_per_step": round(elapsed2 / args.steps * 1e3, 2), "value": round(args.steps / elapsed2, 4),
                                     "what": "the product default: the unconditional CFG branch's cross-attention is its to_out bias "
                                             "(zero context), layer 0's self-attention branch is computed once for both branches "
                                             "(identical inputs) - bit-identical latents (tests/test_denoiser_gpu.py::"
                                             "test_exact_shortcuts_are_bit_identical); `value` above executes every operation"},
            "hip_graph": bool(args.graph and world == 1),
            "latents_fingerprint": fingerprint_of(best["final"]),
            "latents_sharded_across_steps": bool(local_latents),
            "end_to_end_video_to_4d_s": None,
            "end_to_end_note": "unmeasured: pretrained weights / assets unreachable offline; the GPU stages chained on synthetic "
                               "weights are timed by tools/e2e_synthetic.py (profiles/), which is not this metric",
        }
        if args.same_device:
            result["same_device_dry_run"] = True
            result["metric"] = "DRY RUN (all ranks on ONE device, gloo control plane, copy-engine exchange) of: " + result["metric"]
        if world > 1:
            name = next(k for k, v in legs.items() if v is best)
            result["exchange_backend"] = ("peer (copy engines, IPC-mapped gather buffers)" if name == "peer" else "rccl")
            result["exchange_ab"] = exchange_ab
            result["fingerprint_check"] = fingerprint_check
            result["fingerprint_ok"] = None if fingerprint_check is None else bool(fingerprint_check["ok"])
        return result

    # ---- the legs -------------------------------------------------------------------------------------------------------------------
    exchange_ab, legs = None, {}
    if world == 1:
        best = run_leg(None, None)
    else:
        want = os.environ.get("ACTIONMESH_AMD_EXCHANGE") or args.exchange or ("peer" if args.same_device else "ab")
        # rccl first: torch.distributed's bread-and-butter collective is the leg least likely to hang on first contact with a real node;
        # the copy-engine leg (custom IPC + flag protocol, bounded waits) then runs under the watchdog
        order = ["rccl", "peer"] if want == "ab" else [want]
        from actionmesh_amd.sharding import run_exchange_legs
        state = {"printed": False}
        # ---- pre-flight: each back-end on a seconds-long problem in killable child processes --------------
        preflight = None
        if not args.no_preflight:
            preflight = run_preflight(order, args, rank, world, store, sys.argv[1:])
            alive = [n for n in order if preflight[n]["ok"]]
            # nothing passed: the pre-flight itself may be what is broken (a slow first import, a port) - the legs still get their
            # chance, under the watchdog; otherwise only what passed spends a headline warm-up
            skipped = [n for n in order if n not in alive] if alive else []
            order = alive or order
        else:
            skipped = []

        def on_watchdog(leg, legs_so_far, report):
            # a leg hung (a device-side collective that never returns cannot be caught): report the completed leg(s) and leave; with
            # nothing to report the exit code says so (the self-launcher then tries the back-ends one by one)
            report[leg] = {"ok": False, "error": f"no result after {args.leg_timeout:.0f} s (watchdog)"}
            if preflight is not None:
                report["preflight"] = preflight
            if rank == 0 and not state["printed"] and legs_so_far:
                state["printed"] = True
                print(json.dumps(build_result(legs_so_far, report, fingerprint_check=None)), flush=True)
            os._exit(0 if legs_so_far else 3)

        def make_ctl():
            # a fresh control group per leg: its barriers time out instead of waiting for a rank that has left the leg
            return dist.new_group(backend="gloo", timeout=timedelta(seconds=max(20.0, min(180.0, args.leg_timeout / 2))))

        def leg(name, leg_ctl):
            nonlocal ctl
            ctl = leg_ctl
            try:
                # peer: the model's own small collectives (velocity gather among same-frame ranks, the final frame gather) ride on gloo,
                # so the leg makes no RCCL call at all; rccl: everything on the nccl WORLD
                r = run_leg(name, leg_ctl if name == "peer" else dist.group.WORLD)
                torch.cuda.synchronize(dev)
                return r
            finally:
                ctl = ctl_main

        def describe(r):
            d = {"ms_per_step": round(r["elapsed"] / args.steps * 1e3, 2),
                 "with_exact_shortcuts_ms_per_step": round(r["elapsed2"] / args.steps * 1e3, 2)}
            if r.get("flags_fine_grained") is not None:
                d["flags_fine_grained"] = bool(r["flags_fine_grained"])
            return d

        legs, exchange_ab = run_exchange_legs(order, leg, rank, world, args.leg_timeout, on_watchdog, describe, store=store, make_ctl=make_ctl)
        for n in skipped:
            exchange_ab[n] = {"ok": False, "skipped": True,
                              "error": f"failed the pre-flight, no leg was run: {preflight[n].get('error', 'see exchange_ab.preflight')}"}
        if preflight is not None:
            exchange_ab["preflight"] = preflight
        if not legs:
            if rank == 0:
                print(json.dumps({"error": "bench.py: every exchange leg failed", "metric": f"denoise-steps/sec ({T}f x {N}tok)", "value": None,
                                  "n_gpus": world, "exchange_ab": exchange_ab}), flush=True)
            raise SystemExit(3)
        best = min(legs.values(), key=lambda r: r["elapsed"])

    # ---- N > 1: is the sharded result the single-rank result?  Rank 0 re-runs the SAME warmup + steps steps on an unsharded engine (its
    # own device, outside every timed region) and compares the final latents of every completed leg with it: a wrong-but-finite sharded
    # run must not be published as a scaling point (VERDICT r04 weak #5c).  Stated tolerance 3e-2 rel-L2: two bf16 executions with
    # different reduction orders decorrelate step by step (one forward: <= 8e-3, tests/test_denoiser_gpu.py), both stay ~1e-2 of fp32.
    fingerprint_check = None
    if world > 1 and not args.no_fingerprint_check:
        if rank == 0:
            try:
                single = run_leg(None, None)
                ref = single["final"]
                fingerprint_check = {"tol_rel_l2": 3e-2, "single_rank_fingerprint": fingerprint_of(ref), "legs": {}}
                for k, r in legs.items():
                    d = float((r["final"].double() - ref.double())[0, 1:].norm() / ref.double()[0, 1:].norm())
                    fingerprint_check["legs"][k] = round(d, 6)
                fingerprint_check["ok"] = all(v <= 3e-2 for v in fingerprint_check["legs"].values())
                fingerprint_check["single_rank_ms_per_step"] = round(single["elapsed"] / args.steps * 1e3, 2)
            except Exception as e:                       # noqa: BLE001
                fingerprint_check = {"ok": False, "error": f"{type(e).__name__}: {str(e)[:300]}"}
        barrier()

    result = _result(best, legs if world > 1 else {"single": best}, exchange_ab, fingerprint_check)
    if rank == 0 and not args.no_roofline:
        result["roofline"] = attention_roofline(T, N, H, dev, world, dtype=args.dtype, groups=cfg_groups)
    if rank == 0 and world == 1:
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and args.shape == "headline" and not args.no_nominal:
        result["nominal"] = nominal_record(dev, args.dtype)             # the shipped architecture next to the headline
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(hp, sd, best["step_flops"], S, T, N, deep=args.cpu_baseline_deep, shape=args.shape)
    if world > 1:
        barrier()
    if rank == 0:
        print(json.dumps(result), flush=True)
        if fingerprint_check is not None and not fingerprint_check["ok"]:
            print(f"bench.py: the sharded latents differ from the single-rank run beyond the stated tolerance: {fingerprint_check}", file=sys.stderr, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
