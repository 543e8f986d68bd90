#!/usr/bin/env python3
"""WHICH limiter takes the clock from the hot kernels.

tools/clock_probe.py read hwmon's power1_input / freq1_input: 300 W and 2.397 GHz in EVERY leg - evidently not the signal.  The
firmware's own telemetry is the `gpu_metrics` blob (amdgpu sysfs, read here through the amdsmi python binding that ships with
ROCm): instantaneous per-XCD gfx clocks, the socket power the firmware regulates on, hotspot / HBM temperatures, the throttle
status words, and - gpu_metrics v1.6+ - RESIDENCY ACCUMULATORS: how long the PPT (socket power), socket-thermal, VR-thermal,
HBM-thermal and PROCHOT limiters were active, and per XCD how long the gfx clock sat below the host limit because of power
(`gfx_below_host_limit_ppt_acc`) or temperature (`..._thm_acc`).  A limiter's share of a leg = delta(accumulator) /
delta(accumulation_counter).

Legs (same launches as clock_probe.py, 3 s each, sampled at ~20 Hz): idle, bf16 attention on N(0,1) / zero operands, fp8 attention,
qkv GEMM on N(0,1) / zero operands.  Prints one line per leg and writes the full log as JSON (argv: --out).
"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from actionmesh_amd import ops

SCALARS = ("current_socket_power", "average_socket_power", "average_gfxclk_frequency", "current_gfxclk", "temperature_hotspot",
           "temperature_mem", "throttle_status", "indep_throttle_status", "average_gfx_activity", "gfxclk_lock_status",
           "accumulation_counter", "prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc",
           "hbm_thm_residency_acc", "energy_accumulator", "firmware_timestamp", "voltage_gfx")
LISTS = ("current_gfxclks", "temperature_hbm")
XCP = ("xcp_stats.gfx_below_host_limit_acc", "xcp_stats.gfx_below_host_limit_ppt_acc", "xcp_stats.gfx_below_host_limit_thm_acc",
       "xcp_stats.gfx_below_host_limit_total_acc", "xcp_stats.gfx_low_utilization_acc", "xcp_stats.gfx_busy_acc")


def num(v):
    return v if isinstance(v, (int, float)) and not isinstance(v, bool) else None


class Telemetry:
    def __init__(self):
        import amdsmi
        self.smi = amdsmi
        amdsmi.amdsmi_init()
        self.h = amdsmi.amdsmi_get_processor_handles()[0]

    def sample(self):
        m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        row = {"t": time.time()}
        for k in SCALARS:
            if k in m and num(m[k]) is not None:
                row[k] = m[k]
        for k in LISTS:
            if k in m and isinstance(m[k], list):
                row[k] = [x for x in m[k] if num(x) is not None]
        for k in XCP:                       # [partition][xcd]: partition 0 in SPX mode
            if k in m and isinstance(m[k], list) and m[k] and isinstance(m[k][0], list):
                row[k] = [x for x in m[k][0] if num(x) is not None]
        return row

    def header(self):
        m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        return {k: m.get(k) for k in ("common_header.structure_size", "common_header.format_revision", "common_header.content_revision")}

    def violations(self):
        try:
            v = self.smi.amdsmi_get_violation_status(self.h)
            return {k: v[k] for k in v if k.startswith(("active_", "per_")) and not isinstance(v[k], list)}
        except Exception as e:      # older firmware / binding
            return {"error": repr(e)}

    def power_cap(self):
        try:
            return self.smi.amdsmi_get_power_cap_info(self.h)
        except Exception as e:
            return {"error": repr(e)}


class Sampler(threading.Thread):
    def __init__(self, tel, dt=0.05):
        super().__init__(daemon=True)
        self.tel, self.dt, self.rows, self.stop = tel, dt, [], False

    def run(self):
        while not self.stop:
            try:
                self.rows.append(self.tel.sample())
            except Exception as e:
                self.rows.append({"t": time.time(), "error": repr(e)})
            time.sleep(self.dt)


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else float("nan")


def leg(name, fn, seconds, tel, flops=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = Sampler(tel)
    first = tel.sample()
    s.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t0 = 0, time.time()
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(4):
            fn()
        n += 4
        if n % 16 == 0:
            torch.cuda.current_stream().synchronize()
    viol = tel.violations() if flops else None          # taken UNDER load (the call itself spans ~0.1 s)
    e1.record()
    torch.cuda.synchronize()
    s.stop = True
    s.join()
    last = tel.sample()
    ms = e0.elapsed_time(e1) / max(n, 1)
    rows = [r for r in s.rows[len(s.rows) // 4:] if "error" not in r]
    rec = {"leg": name, "ms_per_launch": round(ms, 3), "launches": n, "samples": len(rows)}
    if flops:
        rec["tflops"] = round(flops / ms / 1e9, 1)
    clk = [sum(r["current_gfxclks"]) / len(r["current_gfxclks"]) for r in rows if r.get("current_gfxclks")]
    if clk:
        rec["gfxclk_mhz_median_over_xcds"] = round(med(clk), 1)
        rec["gfxclk_mhz_min_max"] = [round(min(clk), 1), round(max(clk), 1)]
    for k, out in (("current_socket_power", "socket_power_w_median"), ("average_gfxclk_frequency", "average_gfxclk_mhz_median"),
                   ("temperature_hotspot", "hotspot_c_median"), ("temperature_mem", "mem_c_median"), ("voltage_gfx", "voltage_gfx_mv_median")):
        v = [r[k] for r in rows if k in r]
        if v:
            rec[out] = med(v)
    hb = [max(r["temperature_hbm"]) for r in rows if r.get("temperature_hbm")]
    if hb:
        rec["hbm_c_max"] = max(hb)
    for k in ("throttle_status", "indep_throttle_status"):
        v = sorted({r[k] for r in rows if k in r})
        if v:
            rec[k + "_values_seen"] = v[:8]
    dacc = last.get("accumulation_counter", 0) - first.get("accumulation_counter", 0)
    rec["accumulation_counter_delta"] = dacc
    if dacc > 0:
        for k in ("prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc"):
            if k in first and k in last:
                rec[k.replace("_acc", "_share")] = round((last[k] - first[k]) / dacc, 4)
        for k in XCP:
            if k in first and k in last and len(first[k]) == len(last[k]) and first[k]:
                d = [(b - a) / dacc for a, b in zip(first[k], last[k])]
                rec[k.split(".")[1].replace("_acc", "_share_per_xcd")] = [round(x, 3) for x in d]
    if "energy_accumulator" in first and "energy_accumulator" in last and last["t"] > first["t"]:
        # the firmware's energy accumulator (15.259 uJ units) over the leg: average socket power, and - round 5 - JOULES PER LAUNCH,
        # the quantity a power-capped kernel is really priced in (time = energy / cap): total, and per algorithmic flop
        joules = (last["energy_accumulator"] - first["energy_accumulator"]) * 15.259e-6
        rec["energy_accumulator_w"] = round(joules / (last["t"] - first["t"]), 1)
        rec["joules_per_launch"] = round(rec["energy_accumulator_w"] * ms * 1e-3, 4)
        if flops:
            rec["pj_per_flop"] = round(rec["energy_accumulator_w"] * ms * 1e-3 / flops * 1e12, 4)
    if viol is not None:
        rec["violation_status_under_load"] = viol
    short = {k: rec[k] for k in rec if k not in ("violation_status_under_load",)}
    print(json.dumps(short), flush=True)
    return rec, s.rows


def energy_table(a, tel):
    """Round 5: the bf16 attention as an ENERGY problem.  The same launch at the headline shape on N(0,1) operands,
    product form and the timing ablations of an -DAM_ATTN_ABLATIONS build (ACTIONMESH_AMD_LIB=build/variants/libam_abl.so), each with its
    time, average socket power, joules per launch and pJ per algorithmic flop; the differences between rows price the pieces:
    product - (no softmax) = the softmax VALU; (no softmax) - (no softmax, no LDS reads) = the fragment reads; the pure MFMA + DMA
    stream is the floor the schedule sits on.  Also: the fp8 forms and the ff1 GEMM with / without the GELU table."""
    dev = torch.device("cuda:0")
    T, N, C, H = 16, 4096, 1024, 8
    B, L = 2, N + 1
    Sq = T * L
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(s, device=dev, generator=g).to(torch.bfloat16)
    Q = rnd(B, H, ops.round_up(Sq, 256), 128); K = rnd(B, H, ops.round_up(Sq, 64), 128); Vt = rnd(B, H, 128, ops.round_up(Sq, 64))
    out = torch.empty((B * Sq, C), dtype=torch.bfloat16, device=dev)
    fl = 4.0 * Sq * Sq * C * B
    have_abl = True
    try:
        ops.attention(Q, K, Vt, Sq, Sq, out=out, defer_log2=3008)
        torch.cuda.synchronize()
    except RuntimeError as e:
        print("ablation legs skipped:", str(e)[:300], flush=True)
        have_abl = False
    ops.attention_fp8(Q, K, Vt, Sq, Sq, out=out); qz = ops.attention_fp8.last_quantized
    R = B * Sq
    A = rnd(R, C); W1 = rnd(4 * C, C) * 0.03; b1 = torch.zeros(4 * C, device=dev); Cc = torch.empty((R, 4 * C), dtype=torch.bfloat16, device=dev)
    legs = [("idle (sleep)", lambda: time.sleep(0.01), None, 1.5),
            ("attention bf16 product (lazy re-base)", lambda: ops.attention(Q, K, Vt, Sq, Sq, out=out), fl, a.seconds)]
    if have_abl:
        for code, what in ((3001, "no exp (x0.5 instead)"), (3002, "no row max"), (3004, "no tile barrier / DMA drain"),
                           (3008, "no softmax at all (MFMA + LDS fragment reads + DMA)"), (3016, "no LDS fragment reads (stale registers)"),
                           (3030, "pure MFMA stream (no softmax, no reads, no barrier, no row max)")):
            legs.append((f"attention bf16 ablation {code}: {what}", lambda c=code: ops.attention(Q, K, Vt, Sq, Sq, out=out, defer_log2=c), fl, a.seconds))
    legs += [("attention fp8 (attend only)", lambda: ops.attention_fp8(Q, K, Vt, Sq, Sq, out=out, quantized=qz), fl, a.seconds),
             ("attention fp8_fast (attend only)", lambda: ops.attention_fp8(Q, K, Vt, Sq, Sq, out=out, quantized=qz, ablate=400), fl, a.seconds),
             ("GEMM ff1 + GELU (table) 131104 x 4096 x 1024", lambda: ops.gemm(A, W1, bias=b1, gelu=True, out=Cc), 2.0 * R * C * 4 * C, a.seconds),
             ("GEMM ff1 + GELU (arithmetic)", lambda: ops.gemm(A, W1, bias=b1, gelu=True, out=Cc, gelu_table=False), 2.0 * R * C * 4 * C, a.seconds),
             ("GEMM ff1, no activation", lambda: ops.gemm(A, W1, bias=b1, out=Cc), 2.0 * R * C * 4 * C, a.seconds)]
    log = {"what": "energy per launch, headline shape, N(0,1) operands", "ablation_build": have_abl, "legs": []}
    idle_w = None
    for name, fn, flops, sec in legs:
        rec, _rows = leg(name, fn, sec, tel, flops)
        if flops is None:
            idle_w = rec.get("energy_accumulator_w")
        elif idle_w is not None and "energy_accumulator_w" in rec:
            rec["joules_per_launch_above_idle"] = round((rec["energy_accumulator_w"] - idle_w) * rec["ms_per_launch"] * 1e-3, 4)
        log["legs"].append(rec)
    print("\n%-78s %9s %8s %9s %9s %8s" % ("leg", "ms", "W", "J/launch", "J>idle", "pJ/flop"))
    for r in log["legs"]:
        print("%-78s %9.3f %8.1f %9.3f %9s %8s" % (r["leg"][:78], r["ms_per_launch"], r.get("energy_accumulator_w", float("nan")),
                                                  r.get("joules_per_launch", float("nan")), r.get("joules_per_launch_above_idle", "-"),
                                                  r.get("pj_per_flop", "-")))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(log, f, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--energy-table", action="store_true", help="per-variant joules per launch (see energy_table)")
    a = ap.parse_args()
    if a.energy_table:
        return energy_table(a, Telemetry())
    tel = Telemetry()
    dev = torch.device("cuda:0")
    log = {"gpu_metrics_header": tel.header(), "power_cap": {k: (v if num(v) is not None else str(v)) for k, v in tel.power_cap().items()},
           "first_sample": tel.sample(), "legs": []}
    print("gpu_metrics header:", log["gpu_metrics_header"], "power cap:", log["power_cap"], flush=True)
    T, N, C, H = 16, 4096, 1024, 8
    B, L = 2, N + 1
    Sq = T * L
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(s, device=dev, generator=g).to(torch.bfloat16)
    Q = rnd(B, H, ops.round_up(Sq, 256), 128); K = rnd(B, H, ops.round_up(Sq, 64), 128); Vt = rnd(B, H, 128, ops.round_up(Sq, 64))
    out = torch.empty((B * Sq, C), dtype=torch.bfloat16, device=dev)
    fl = 4.0 * Sq * Sq * C * B
    Qz, Kz, Vz = torch.zeros_like(Q), torch.zeros_like(K), torch.zeros_like(Vt)
    ops.attention_fp8(Q, K, Vt, Sq, Sq, out=out); qz = ops.attention_fp8.last_quantized
    R = B * Sq
    A = rnd(R, C); W = rnd(3 * C, C); Cc = torch.empty((R, 3 * C), dtype=torch.bfloat16, device=dev)
    Az, Wz = torch.zeros_like(A), torch.zeros_like(W)
    legs = [("idle (sleep)", lambda: time.sleep(0.01), None, min(a.seconds, 1.5)),
            ("attention bf16, N(0,1) operands", lambda: ops.attention(Q, K, Vt, Sq, Sq, out=out), fl, a.seconds),
            ("attention bf16, zero operands", lambda: ops.attention(Qz, Kz, Vz, Sq, Sq, out=out), fl, a.seconds),
            ("attention fp8 (attend only), N(0,1)", lambda: ops.attention_fp8(Q, K, Vt, Sq, Sq, out=out, quantized=qz), fl, a.seconds),
            ("GEMM qkv 131104 x 3072 x 1024, N(0,1)", lambda: ops.gemm(A, W, out=Cc), 2.0 * R * C * 3 * C, a.seconds),
            ("GEMM qkv, zero operands", lambda: ops.gemm(Az, Wz, out=Cc), 2.0 * R * C * 3 * C, a.seconds),
            ("attention bf16, N(0,1) again (box warm)", lambda: ops.attention(Q, K, Vt, Sq, Sq, out=out), fl, a.seconds)]
    for name, fn, flops, sec in legs:
        rec, rows = leg(name, fn, sec, tel, flops)
        rec["trace_20hz"] = [{k: r[k] for k in r if k in ("t", "current_socket_power", "current_gfxclks", "temperature_hotspot", "throttle_status",
                                                           "indep_throttle_status", "ppt_residency_acc", "accumulation_counter")} for r in rows]
        log["legs"].append(rec)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(log, f, indent=1)


if __name__ == "__main__":
    main()
