#!/usr/bin/env python
"""Per-phase cycle stamps of the lean attention kernel (needs an AM_ATTN_ABLATIONS build)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from actionmesh_amd import ops, _lib as L
dev = torch.device("cuda:0")
B, H, S = 2, 8, 16 * 4097
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(s, device=dev, generator=g).to(torch.bfloat16)
Q = rnd(B, H, ops.round_up(S, 256), 128); K = rnd(B, H, ops.round_up(S, 64), 128); Vt = rnd(B, H, 128, ops.round_up(S, 64))
out = torch.empty((B * S, H * 128), dtype=torch.bfloat16, device=dev)
a = L.AmAttnArgs()
a.Q, a.K, a.Vt, a.O = Q.data_ptr(), K.data_ptr(), Vt.data_ptr(), out.data_ptr()
a.nseq, a.heads, a.sq, a.sq_pad, a.sk, a.sk_pad = B, H, S, Q.shape[2], S, K.shape[2]
a.nchunks = 1; a.chunk_stride = 0; a.ldo = H * 128; a.scale = 128 ** -0.5; a.defer_log2 = 8
if "--fp8p" in sys.argv:            # round 4: the free-running 8-wave fp8 kernel (product); needs tools/build_fp8_prof.sh + ACTIONMESH_AMD_LIB
    ops.attention_fp8(Q, K, Vt, S, S, out=out)
    q8, k8, vt8 = ops.attention_fp8.last_quantized
    prof = torch.zeros(8 * 8 * 8, dtype=torch.int64, device=dev)
    lib = C.CDLL(L.LIB_PATH)
    lib.am_attention_fp8p_profile.argtypes = [C.POINTER(L.AmAttnArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for _ in range(2):
        rc = lib.am_attention_fp8p_profile(C.byref(a), q8.data_ptr(), k8.data_ptr(), vt8.data_ptr(), prof.data_ptr(), None)
    torch.cuda.synchronize()
    assert rc == 0
    p = prof.cpu().view(8, 8, 8)
    print("fp8 free-running kernel, workgroup (0,0), cycles (s_memtime): barrier wait | a (2 PV, 2 DMA, row max) | b (2 PV + 4 QK, exp/sum/pack) | tail")
    for w in range(8):
        print(f"wave {w}:")
        for t in range(0, 7):
            r = p[w, t]; nxt = p[w, t + 1, 0]
            print(f"  tile {64 + t}: barrier {int(r[1]-r[0]):5d} | a {int(r[2]-r[1]):5d} | b {int(r[3]-r[2]):5d} | tail {int(nxt-r[3]):4d} | total {int(nxt-r[0]):5d}")
    print("barrier-exit skew (tile 66):", [int(p[w, 2, 1] - p[0, 2, 1]) for w in range(8)])
    sys.exit(0)
if "--fp8x64" in sys.argv:          # round 4: the 4 x 64 fp8 kernel (needs tools/build_fp8_prof.sh + ACTIONMESH_AMD_LIB)
    ops.attention_fp8(Q, K, Vt, S, S, out=out)
    q8, k8, vt8 = ops.attention_fp8.last_quantized
    prof = torch.zeros(4 * 8 * 8, dtype=torch.int64, device=dev)
    lib = C.CDLL(L.LIB_PATH)
    lib.am_attention_fp8x64_profile.argtypes = [C.POINTER(L.AmAttnArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for _ in range(2):
        rc = lib.am_attention_fp8x64_profile(C.byref(a), q8.data_ptr(), k8.data_ptr(), vt8.data_ptr(), prof.data_ptr(), None)
    torch.cuda.synchronize()
    assert rc == 0
    p = prof.cpu().view(4, 8, 8)
    print("fp8 4x64 kernel, workgroup (0,0), cycles (s_memtime): barrier wait | 1a (2 PV, 2 DMA, row max 0) | 1b (6 PV, exp/sum/pack 0) | "
          "2a (2 QK, 2 DMA, row max 1) | 2b (6 QK, exp/sum/pack 1) | tail")
    for w in range(4):
        print(f"wave {w}:")
        for t in range(0, 7):
            r = p[w, t]; nxt = p[w, t + 1, 0]
            print(f"  tile {64 + t}: barrier {int(r[1]-r[0]):5d} | 1a {int(r[2]-r[1]):5d} | 1b {int(r[3]-r[2]):5d} | 2a {int(r[4]-r[3]):5d} | "
                  f"2b {int(r[5]-r[4]):5d} | tail {int(nxt-r[5]):4d} | total {int(nxt-r[0]):5d}")
    print("barrier-exit skew (tile 66):", [int(p[w, 2, 1] - p[0, 2, 1]) for w in range(4)])
    sys.exit(0)
if "--k64" in sys.argv:
    if "--exact" in sys.argv:
        a.defer_log2 = 28
    prof = torch.zeros(4 * 8 * 8, dtype=torch.int64, device=dev)
    lib = C.CDLL(L.LIB_PATH)
    lib.am_attention64_profile.argtypes = [C.POINTER(L.AmAttnArgs), C.c_void_p, C.c_void_p]
    for _ in range(2):
        rc = lib.am_attention64_profile(C.byref(a), prof.data_ptr(), None)
    torch.cuda.synchronize()
    assert rc == 0
    p = prof.cpu().view(4, 8, 8)
    print("4x64 kernel (lazy; --exact: exact re-base), workgroup (0,0), cycles (s_memtime): barrier | dma issue | 1a (first 8 PV, DMA) | "
          "1b (24 PV) | 2a (first 8 QK, DMA) | 2b (24 QK) | loop tail")
    for w in range(4):
        print(f"wave {w}:")
        for t in range(0, 7):
            r = p[w, t]; nxt = p[w, t + 1, 0]
            print(f"  tile {64 + t}: barrier {int(r[1]-r[0]):5d} | dma {int(r[2]-r[1]):4d} | 1a {int(r[3]-r[2]):5d} | 1b {int(r[4]-r[3]):5d} | "
                  f"2a {int(r[5]-r[4]):5d} | 2b {int(r[6]-r[5]):5d} | tail {int(nxt-r[6]):4d} | total {int(nxt-r[0]):5d}")
    print("barrier-exit skew (tile 66):", [int(p[w, 2, 1] - p[0, 2, 1]) for w in range(4)])
    sys.exit(0)
prof = torch.zeros(8 * 8 * 6, dtype=torch.int64, device=dev)
lib = C.CDLL(L.LIB_PATH)
lib.am_attention_profile.argtypes = [C.POINTER(L.AmAttnArgs), C.c_void_p, C.c_void_p]
for _ in range(2):
    rc = lib.am_attention_profile(C.byref(a), prof.data_ptr(), None)
torch.cuda.synchronize()
assert rc == 0
p = prof.cpu().view(8, 8, 6)
print("slots: 0 pre-barrier, 1 post-barrier, 2 QK^T drained, 3 softmax done, 4 PV drained   (cycles, s_memtime)")
for w in (0, 1, 4, 5):
    print(f"wave {w}:")
    for t in range(1, 7):
        r = p[w, t]
        nxt = p[w, t + 1, 0]
        print(f"  tile {64 + t}: barrier-wait {int(r[1]-r[0]):5d} | QK^T {int(r[2]-r[1]):5d} | softmax {int(r[3]-r[2]):5d} | "
              f"PV {int(r[4]-r[3]):5d} | tail {int(nxt - r[4]):5d} | total {int(nxt - r[0]):5d}")
print("cross-wave skew at barrier exit (tile 66):", [int(p[w, 2, 1] - p[0, 2, 1]) for w in range(8)])
