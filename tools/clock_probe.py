#!/usr/bin/env python3
"""What clock and board power the hot kernels sustain: runs one kernel back to back for a few seconds while a thread samples the driver's
sysfs (hwmon freq1_input = sclk, power1_average / power1_input) - the attention kernel's share of the dense MFMA peak is (MFMA pipe busy)
x (sustained clock / 2.4 GHz), and only the first factor is the kernel's schedule.  Legs: the product bf16 attention launch on random,
small-magnitude and zero inputs (the data dependence of the power draw), the fp8 launch, the qkv GEMM, idle.

    python tools/clock_probe.py [--seconds 3]
"""
import argparse, glob, os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from actionmesh_amd import ops


def sysfs_sources():
    src = {}
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for key, names in (("sclk_hz", ("freq1_input",)), ("power_uw", ("power1_average", "power1_input")), ("mclk_hz", ("freq2_input",)),
                           ("temp_mc", ("temp1_input", "temp2_input"))):
            for n in names:
                p = os.path.join(h, n)
                if key not in src and os.path.exists(p):
                    try:
                        open(p).read(); src[key] = p
                    except OSError:
                        pass
    return src


def read(p):
    try:
        return float(open(p).read().strip())
    except (OSError, ValueError):
        return float("nan")


class Sampler(threading.Thread):
    def __init__(self, src, dt=0.02):
        super().__init__(daemon=True); self.src, self.dt, self.rows, self.stop = src, dt, [], False
    def run(self):
        while not self.stop:
            self.rows.append({k: read(p) for k, p in self.src.items()}); time.sleep(self.dt)


def leg(name, fn, seconds, src, flops=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = Sampler(src); s.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t0 = 0, time.time()
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(4):
            fn()
        n += 4
        torch.cuda.current_stream().synchronize() if n % 16 == 0 else None
    e1.record(); torch.cuda.synchronize()
    s.stop = True; s.join()
    ms = e0.elapsed_time(e1) / max(n, 1)
    rows = s.rows[len(s.rows) // 4:]            # skip the ramp
    def stat(k, scale):
        v = sorted(r[k] * scale for r in rows if k in r and r[k] == r[k])
        return (v[len(v) // 2], v[0], v[-1]) if v else (float("nan"),) * 3
    ck, pw = stat("sclk_hz", 1e-9), stat("power_uw", 1e-6)
    tf = f"{flops / ms / 1e9:8.1f} TFLOP/s" if flops else " " * 16
    print(f"{name:34s} {ms:8.3f} ms {tf}  sclk median {ck[0]:.3f} GHz [{ck[1]:.3f}, {ck[2]:.3f}]  power median {pw[0]:.0f} W [{pw[1]:.0f}, {pw[2]:.0f}]  ({len(rows)} samples)", flush=True)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=3.0); a = ap.parse_args()
    src = sysfs_sources()
    print("sysfs:", src if src else "no hwmon files readable - times only", flush=True)
    dev = torch.device("cuda:0")
    T, N, C, H = 16, 4096, 1024, 8
    B, L = 2, N + 1
    Sq = T * L
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(s, device=dev, generator=g).to(torch.bfloat16)
    Q = rnd(B, H, ops.round_up(Sq, 256), 128); K = rnd(B, H, ops.round_up(Sq, 64), 128); Vt = rnd(B, H, 128, ops.round_up(Sq, 64))
    out = torch.empty((B * Sq, C), dtype=torch.bfloat16, device=dev)
    fl = 4.0 * Sq * Sq * C * B
    leg("idle (sleep)", lambda: time.sleep(0.01), min(a.seconds, 1.0), src)
    leg("attention bf16, N(0,1) inputs", lambda: ops.attention(Q, K, Vt, Sq, Sq, out=out), a.seconds, src, fl)
    Qs, Ks, Vs = Q * 0.05, K * 0.05, Vt * 0.05
    leg("attention bf16, N(0,0.05^2) inputs", lambda: ops.attention(Qs, Ks, Vs, Sq, Sq, out=out), a.seconds, src, fl)
    Qz, Kz, Vz = torch.zeros_like(Q), torch.zeros_like(K), torch.zeros_like(Vt)
    leg("attention bf16, zero inputs", lambda: ops.attention(Qz, Kz, Vz, Sq, Sq, out=out), a.seconds, src, fl)
    leg("attention bf16, zero V only", lambda: ops.attention(Q, K, Vz, Sq, Sq, out=out), a.seconds, src, fl)
    ops.attention_fp8(Q, K, Vt, Sq, Sq, out=out); qz = ops.attention_fp8.last_quantized
    leg("attention fp8 (attend only)", lambda: ops.attention_fp8(Q, K, Vt, Sq, Sq, out=out, quantized=qz), a.seconds, src, fl)
    R = B * Sq
    A = rnd(R, C); W = rnd(3 * C, C); Cc = torch.empty((R, 3 * C), dtype=torch.bfloat16, device=dev)
    leg("GEMM qkv 131104 x 3072 x 1024", lambda: ops.gemm(A, W, out=Cc), a.seconds, src, 2.0 * R * C * 3 * C)
    Az, Wz = torch.zeros_like(A), torch.zeros_like(W)
    leg("GEMM qkv, zero inputs", lambda: ops.gemm(Az, Wz, out=Cc), a.seconds, src, 2.0 * R * C * 3 * C)
    leg("attention bf16 again (box warm)", lambda: ops.attention(Q, K, Vt, Sq, Sq, out=out), a.seconds, src, fl)


if __name__ == "__main__":
    main()
