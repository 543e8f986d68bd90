#!/usr/bin/env python
"""Per-kernel timing at the headline (or nominal) layer shapes, HIP events on the launch stream.
    python tools/kernel_bench.py [--shape headline|nominal] [--only attn,gemm,ln,post] [--reps 3]
Prints one line per kernel: ms, TFLOP/s (MFMA kernels) or GB/s (HBM-bound kernels)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from actionmesh_amd import ops  # noqa: E402


def timeit(fn, reps):
    for _ in range(max(3, reps)):       # warm: the first timed loop behind freshly made operands used to read ~8 % slow (clock ramp)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="headline")
    ap.add_argument("--only", default="attn,xattn,gemm,ln,post")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--defer", type=int, default=8)
    ap.add_argument("--ablate", action="store_true", help="time the AM_ATTN_ABLATIONS variants (needs that build)")
    ap.add_argument("--ablate64", action="store_true", help="time the 4x64 kernel's ablations (AM_ATTN_ABLATIONS build)")
    ap.add_argument("--variants", action="store_true", help="also time the experimental schedules (AM_ATTN_ABLATIONS build)")
    ap.add_argument("--fp8", action="store_true", help="attn: the fp8 kernel too when --product-only")
    ap.add_argument("--ablate-fp8", action="store_true", help="attn: time the fp8 kernel's ablations")
    ap.add_argument("--skew-gemm", action="store_true", help="gemm: time the per-XCD start-skew experiment (act bits 13-15)")
    ap.add_argument("--ablate-gemm", action="store_true", help="gemm: time the epilogue ablations (no C stores / no residual loads)")
    ap.add_argument("--product-only", action="store_true", help="attention: the product launch only (PMC passes)")
    ap.add_argument("--gemm-names", default="", help="gemm: only the named layer shapes (comma list of qkv, attn-out+res, ff1+gelu, ff2+res, skip(2C)): "
                    "one shape per counter pass, so that a PMC record belongs to ONE launch shape")
    ap.add_argument("--blas", action="store_true", help="gemm: also time torch.matmul (hipBLASLt) on the same operands - the "
                    "vendor library as a same-box reference point; never on the product path")
    a = ap.parse_args()
    T, N, C, H, S = (16, 4096, 1024, 8, 257) if a.shape == "headline" else (16, 2048, 2048, 16, 257)
    B, L = 2, N + 1
    R = B * T * L
    dev = torch.device("cuda:0")
    only = set(a.only.split(","))
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(s, device=dev, generator=g).to(torch.bfloat16)
    if "attn" in only:
        Sq = T * L
        Q = rnd(B, H, ops.round_up(Sq, 256), 128); K = rnd(B, H, ops.round_up(Sq, 64), 128)
        Vt = rnd(B, H, 128, ops.round_up(Sq, 64)); out = torch.empty((B * Sq, C), dtype=torch.bfloat16, device=dev)
        fl = 4.0 * Sq * Sq * C * B
        for d, nm in ((a.defer, "product"), (a.defer + 60, "4x64 lazy"), (28, "4x64 exact"), (a.defer + 90, "8-wave"), (a.defer + 70, "balanced"), (a.defer + 50, "2x4-wave WGs"), (a.defer + 400, "lean64"), (a.defer + 200, "lockstep"), (a.defer + 300, "pipelined"),
                      (a.defer + 100, "staggered")):
            if (d >= 100 and not a.variants) or (a.product_only and nm != "product"):
                continue
            ms = timeit(lambda: ops.attention(Q, K, Vt, Sq, Sq, out=out, defer_log2=d), a.reps)
            print(f"self-attn  B={B} H={H} S={Sq} variant={d:3d} ({nm:12s}): {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s")
        if not a.product_only or a.fp8:
            ms = timeit(lambda: ops.attention_fp8(Q, K, Vt, Sq, Sq, out=out), a.reps)
            print(f"self-attn  B={B} H={H} S={Sq} fp8 e4m3 (quantise + attend)  : {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s")
            qz = ops.attention_fp8.last_quantized
            ms = timeit(lambda: ops.attention_fp8(Q, K, Vt, Sq, Sq, out=out, quantized=qz), a.reps)
            print(f"self-attn  B={B} H={H} S={Sq} fp8 e4m3 (attend only)        : {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s")
        if not a.product_only or a.fp8:           # round 4: the 4 x 64 product kernel against the round-3 8-wave kernel, interleaved
            for rnd_i in range(2):
                for k, nm in ((0, "free-running 8w (product)"), (400, "same, exponent-field P"), (200, "ping-pong 8w (round 3)"), (100, "4x64")):
                    ms = timeit(lambda: ops.attention_fp8(Q, K, Vt, Sq, Sq, out=out, quantized=qz, ablate=k), a.reps)
                    print(f"  fp8 attend only, {nm:26s} round {rnd_i}: {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s")
            for k, nm in ((301, "product, no exp"), (302, "product, no LDS-DMA"), (308, "product, no softmax steps"), (101, "4x64 no exp"), (108, "4x64 no softmax steps")):
                ms = timeit(lambda: ops.attention_fp8(Q, K, Vt, Sq, Sq, out=out, quantized=qz, ablate=k), a.reps)
                print(f"  fp8 ablation {nm:24s}: {ms:8.3f} ms")
        if a.ablate_fp8:            # round 4: where the product (free-running) kernel's tile time goes (needs tools/build_fp8_prof.sh + ACTIONMESH_AMD_LIB)
            ops.attention_fp8(Q, K, Vt, Sq, Sq, out=out)
            qz = ops.attention_fp8.last_quantized
            for k, nm in ((300, "full"), (308, "no softmax steps"), (304, "no barrier / vmcnt wait"), (316, "no row max"), (332, "no fragment reads"),
                          (312, "no steps, no barrier"), (324, "no steps, no row max"), (340, "no steps, no fragment reads"), (310, "no steps, no LDS-DMA"),
                          (362, "MFMAs only (no steps / barrier / row max / reads / DMA)")):
                try:
                    ms = timeit(lambda: ops.attention_fp8(Q, K, Vt, Sq, Sq, out=out, quantized=qz, ablate=k), a.reps)
                    print(f"  fp8 product ablation {nm:56s}: {ms:8.3f} ms")
                except RuntimeError as e:
                    print(f"  fp8 product ablation {nm}: unavailable in this build ({str(e)[:60]})")
            for k, nm in {1: "no exp", 16: "no row max", 17: "no exp, no row max", 2: "no LDS-DMA", 4: "no fragment reads", 6: "no DMA, no reads", 8: "no MFMAs", 32: "no s_setprio (full kernel)", 64: "s_setprio 1 on the softmax interval (full)", 40: "no setprio, no MFMAs"}.items():
                ms = timeit(lambda: ops.attention_fp8(Q, K, Vt, Sq, Sq, out=out, quantized=qz, ablate=k), a.reps)
                print(f"  fp8 ablation {nm:24s}: {ms:8.3f} ms")
        if a.ablate64:
            for k, nm in {1: "no exp", 2: "no row max", 4: "no barrier / DMA drain", 8: "no exp/sum/pack", 10: "no softmax VALU",
                          14: "no softmax VALU, no barrier", 16: "no LDS fragment addressing (same stage)", 30: "MFMA + fragment reads only", 32: "no exp/sum/pack beside P.V (phase 1b)", 64: "no exp/sum/pack beside QK^T (phase 2b)", 128: "no LDS fragment reads", 132: "no LDS reads, no barrier",
                          142: "no LDS reads, no barrier, no softmax VALU"}.items():
                ms = timeit(lambda: ops.attention(Q, K, Vt, Sq, Sq, out=out, defer_log2=3000 + k), a.reps)
                print(f"  4x64 ablation {nm:42s}: {ms:8.3f} ms  ({fl / ms / 1e9:7.1f} TF-equivalent)")
        if a.ablate:
            for k, nm in {1: "no exp", 2: "no row max", 3: "no row sum", 4: "no cvt", 5: "V frags not prefetched",
                          6: "no softmax VALU at all"}.items():
                ms = timeit(lambda: ops.attention(Q, K, Vt, Sq, Sq, out=out, defer_log2=2000 + k), a.reps)
                print(f"  lean ablation {nm:30s}: {ms:8.3f} ms  ({fl / ms / 1e9:7.1f} TF-equivalent)")
            names = {1: "no softmax math", 3: "no MFMA (softmax+LDS only)", 4: "MFMA from regs (no LDS reads)",
                     5: "MFMA only (no LDS, no softmax, no staging)", 6: "no global loads / LDS writes"}
            for base, lab in ((1000, "lockstep"), (1100, "staggered")):
                for k, nm in names.items():
                    ms = timeit(lambda: ops.attention(Q, K, Vt, Sq, Sq, out=out, defer_log2=base + k), a.reps)
                    print(f"  ablation {lab:9s} {nm:45s}: {ms:8.3f} ms  ({fl / ms / 1e9:7.1f} TF-equivalent)")
        del Q, K, Vt, out
    if "xattn" in only:
        BT = B * T
        Q = rnd(BT, H, ops.round_up(L, 256), 128); K = rnd(BT, H, ops.round_up(S, 64), 128)
        Vt = rnd(BT, H, 128, ops.round_up(S, 64)); out = torch.empty((BT * L, C), dtype=torch.bfloat16, device=dev)
        fl = 4.0 * BT * L * S * C
        for d, nm in ((a.defer, "product"), (a.defer + 60, "4x64"), (a.defer + 90, "8-wave")):
            try:
                ms = timeit(lambda: ops.attention(Q, K, Vt, L, S, out=out, defer_log2=d), a.reps)
            except RuntimeError:
                continue
            print(f"cross-attn BT={BT} H={H} L={L} S={S} variant={d:3d} ({nm:8s}): {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s")
        del Q, K, Vt, out
    if "xsweep" in only:       # cross-attention launch vs the number of context tokens: what is per-workgroup overhead, what is per key tile
        BT = B * T
        Q = rnd(BT, H, ops.round_up(L, 256), 128); out = torch.empty((BT * L, C), dtype=torch.bfloat16, device=dev)
        for Sx in (1, 64, 128, 192, 256, 257, 320, 512):
            K = rnd(BT, H, ops.round_up(Sx, 64), 128); Vt = rnd(BT, H, 128, ops.round_up(Sx, 64))
            ms = timeit(lambda: ops.attention(Q, K, Vt, L, Sx, out=out, defer_log2=a.defer), a.reps)
            print(f"cross-attn sweep BT={BT} H={H} L={L} S={Sx:4d} ({ops.round_up(Sx, 64) // 64} key tiles): {ms:8.3f} ms")
        del Q, out
    if "fused" in only:          # the QKV linear + head split: one launch (am_gemm_headpost_bf16) vs two
        L_ = N + 1
        z = rnd(R, C); wqkv = rnd(3 * C, C)
        nq = torch.ones(128, device=dev); nk = torch.ones(128, device=dev)
        ang = torch.arange(2 * T, device=dev)[:, None].float() * (10000.0 ** (-torch.arange(64, device=dev).float() * 2 / 128))[None]
        rope = (torch.cos(ang).contiguous(), torch.sin(ang).contiguous())
        x = torch.empty((R, 3 * C), dtype=torch.bfloat16, device=dev)
        q_, k_, v_ = ops.head_post(ops.gemm(z, wqkv, out=x), H, (0, 1, 2), T * L_, L_, w_q=nq, w_k=nk, rope=rope)
        two = timeit(lambda: ops.head_post(ops.gemm(z, wqkv, out=x), H, (0, 1, 2), T * L_, L_, w_q=nq, w_k=nk, rope=rope, out_q=q_, out_k=k_, out_vt=v_), a.reps)
        one = timeit(lambda: ops.gemm_head_post(z, wqkv, H, (0, 1, 2), T * L_, L_, w_q=nq, w_k=nk, rope=rope, out_q=q_, out_k=k_, out_vt=v_, x=x), a.reps)
        print(f"qkv linear + head split  R={R} 3C={3 * C}: two launches {two:8.3f} ms, fused {one:8.3f} ms")
        # the launch the MODEL makes (round 6): LayerNorm folded in (statistics + column sums + d), beside the plain linear of the same
        # shape and the epilogue's timing ablations (no Q / K rows, no V^T read-back, neither = main loop + staging pass only)
        stats = torch.stack([torch.zeros(R, device=dev), torch.ones(R, device=dev)], 1).contiguous()
        colsum = wqkv.float().sum(1).contiguous(); dvec = torch.zeros(3 * C, device=dev)
        kw = dict(w_q=nq, w_k=nk, rope=rope, out_q=q_, out_k=k_, out_vt=v_, x=x)
        plain = timeit(lambda: ops.gemm(z, wqkv, out=x), a.reps)
        fold = timeit(lambda: ops.gemm_head_post(z, wqkv, H, (0, 1, 2), T * L_, L_, bias=dvec, ln=(stats, colsum), **kw), a.reps)
        print(f"qkv in-model (LN fold + head split + qk-norm + RoPE) R={R} 3C={3 * C}: {fold:8.3f} ms  {2.0 * R * 3 * C * C / fold / 1e9:7.1f} TFLOP/s"
              f"   | plain linear {plain:8.3f} ms, fused without the fold {one:8.3f} ms")
        for ab, nm in ((0x800, "no Q / K rows"), (0x1000, "no V^T read-back"), (0x1800, "neither (main loop + staging)")):
            ms = timeit(lambda: ops.gemm_head_post(z, wqkv, H, (0, 1, 2), T * L_, L_, bias=dvec, ln=(stats, colsum), ablate=ab, **kw), a.reps)
            print(f"  qkv in-model ablation {nm:30s}: {ms:8.3f} ms")
        wxq = rnd(C, C)
        xq = torch.empty((R, C), dtype=torch.bfloat16, device=dev)
        qx, _, _ = ops.head_post(ops.gemm(z, wxq, out=xq), H, (0,), L_, L_, w_q=nq)
        two = timeit(lambda: ops.head_post(ops.gemm(z, wxq, out=xq), H, (0,), L_, L_, w_q=nq, out_q=qx), a.reps)
        one = timeit(lambda: ops.gemm_head_post(z, wxq, H, (0,), L_, L_, w_q=nq, out_q=qx, x=xq), a.reps)
        print(f"cross to_q + head split  R={R} C={C}: two launches {two:8.3f} ms, fused {one:8.3f} ms")
    if "gemm" in only:
        F_ = 4 * C
        for name, Nn, Kk, kw in (("qkv", 3 * C, C, {}), ("attn-out+res", C, C, {"res": True, "bias": True}),
                                 ("ff1+gelu", F_, C, {"bias": True, "gelu": True}),
                                 ("ff2+res", C, F_, {"bias": True, "res": True}),
                                 ("skip(2C)", C, 2 * C, {"bias": True})):
            if a.gemm_names and name not in a.gemm_names.split(","):
                continue
            A = rnd(R, Kk); W = rnd(Nn, Kk); out = torch.empty((R, Nn), dtype=torch.bfloat16, device=dev)
            if kw.get("gelu"):
                W = (W.float() * Kk ** -0.5).to(torch.bfloat16)      # unit-variance pre-activations (a linear behind a LayerNorm): the
                                                                     # GELU table's domain; N(0, 32^2) ones would all take its repair path
            bias = torch.randn(Nn, device=dev) if kw.get("bias") else None
            res = rnd(R, Nn) if kw.get("res") else None
            for small, leg, nm in ((False, False, "256sq-pingpong"), (False, True, "256sq-lockstep"), (True, False, "128sq-regstage")):
                if a.product_only and (small or leg):
                    continue
                ms = timeit(lambda: ops.gemm(A, W, bias=bias, residual=res, gelu=kw.get("gelu", False), out=out,
                                             force_small=small, legacy=leg), a.reps)
                print(f"gemm {name:13s} M={R} N={Nn} K={Kk} {nm}: {ms:8.3f} ms  {2.0 * R * Nn * Kk / ms / 1e9:8.1f} TFLOP/s")
            if kw.get("gelu"):     # round 5: the GELU epilogue by LDS table (product) vs arithmetic, and the same linear without the activation
                for nm, k2 in (("256sq-pingpong, arithmetic GELU", dict(gelu=True, gelu_table=False)), ("256sq-pingpong, no activation", dict(gelu=False))):
                    ms = timeit(lambda: ops.gemm(A, W, bias=bias, residual=res, out=out, **k2), a.reps)
                    print(f"gemm {name:13s} M={R} N={Nn} K={Kk} {nm}: {ms:8.3f} ms  {2.0 * R * Nn * Kk / ms / 1e9:8.1f} TFLOP/s")
            if a.skew_gemm:
                for units in (7, 1, 2, 3, 4, 6):
                    ms = timeit(lambda: ops.gemm(A, W, bias=bias, residual=res, gelu=kw.get("gelu", False), out=out, ablate=units << 13), a.reps)
                    print(f"  gemm {name:13s} per-XCD start skew {'off' if units == 7 else f'{units * 1.5:4.1f} us x xcd'}: {ms:8.3f} ms  {2.0 * R * Nn * Kk / ms / 1e9:8.1f} TFLOP/s")
            if a.ablate_gemm:
                for ab, nm in ((0x800, "no C stores"), (0x1000, "no residual loads"), (0x1800, "neither")):
                    ms = timeit(lambda: ops.gemm(A, W, bias=bias, residual=res, gelu=kw.get("gelu", False), out=out, ablate=ab), a.reps)
                    print(f"  gemm {name:13s} ablation {nm:18s}: {ms:8.3f} ms")
            if a.blas:
                import torch.nn.functional as F
                Wt = W.t()
                ms = timeit(lambda: torch.matmul(A, Wt, out=out), a.reps)
                print(f"gemm {name:13s} M={R} N={Nn} K={Kk} hipBLASLt(plain): {ms:8.3f} ms  {2.0 * R * Nn * Kk / ms / 1e9:8.1f} TFLOP/s")
                # the op sequence the reference runs for this linear under autocast (vendor GEMM + separate elementwise passes):
                # F.linear(+bias), F.gelu, the residual add of block.py:137-152, the torch.cat of block.py:131
                bb = bias.to(torch.bfloat16) if bias is not None else None
                if name.startswith("skip"):
                    h1, h2 = A[:, :Kk // 2].contiguous(), A[:, Kk // 2:].contiguous()
                    fn = lambda: F.linear(torch.cat([h1, h2], dim=-1), W, bb)
                elif kw.get("gelu"):
                    fn = lambda: F.gelu(F.linear(A, W, bb))
                elif kw.get("res"):
                    fn = lambda: res + F.linear(A, W, bb)
                else:
                    fn = lambda: F.linear(A, W, bb)
                ms = timeit(fn, a.reps)
                print(f"gemm {name:13s} M={R} N={Nn} K={Kk} torch op sequence (library GEMM + elementwise): {ms:8.3f} ms  {2.0 * R * Nn * Kk / ms / 1e9:8.1f} TFLOP/s")
            del A, W, out, res
    if "nn" in only:      # ActionBench nearest-neighbour search (am_nn_search): metric shape and the ICP inner-loop shape
        for P, Q, Bn, precise in ((100_000, 100_000, 1, True), (100_000, 10_000, 1, True), (100_000, 100_000, 1, False), (10_000, 10_000, 24, False)):
            pts = torch.randn((Bn, P, 3), device=dev, generator=g); qry = torch.randn((Bn, Q, 3), device=dev, generator=g)
            ms = timeit(lambda: ops.nearest_neighbors(pts, qry, precise=precise), a.reps)
            # 8 arithmetic lane-operations per pair (3 sub, 3 mul, 2 add) + compare / select
            print(f"nn_search  P={P} Q={Q} batch={Bn} {'fp64' if precise else 'fp32'}: {ms:8.3f} ms  {Bn * P * Q / ms / 1e6:8.1f} G pairs/s  "
                  f"{8.0 * Bn * P * Q / ms / 1e9:7.2f} TFLOP/s")
    if "ln" in only:
        x = rnd(R, C); w = torch.ones(C, device=dev); b = torch.zeros(C, device=dev); y = torch.empty_like(x)
        ms = timeit(lambda: ops.layernorm(x, w, b, out=y), a.reps)
        print(f"layernorm  rows={R} C={C}: {ms:8.3f} ms  {2.0 * R * C * 2 / ms / 1e6:8.1f} GB/s")
    if "post" in only:
        x = rnd(R, 3 * C); wq = torch.ones(128, device=dev)
        cos = torch.randn(B * T, 64, device=dev); sin = torch.randn(B * T, 64, device=dev)
        q, k, vt = ops.head_post(x, H, (0, 1, 2), T * L, L, w_q=wq, w_k=wq, rope=(cos, sin))
        ms = timeit(lambda: ops.head_post(x, H, (0, 1, 2), T * L, L, w_q=wq, w_k=wq, rope=(cos, sin),
                                          out_q=q, out_k=k, out_vt=vt), a.reps)
        print(f"head_post  rows={R} 3C={3 * C}: {ms:8.3f} ms  {2.0 * R * 3 * C * 2 / ms / 1e6:8.1f} GB/s")


if __name__ == "__main__":
    main()
