"""Kernel-level Python wrappers over the C-ABI (torch tensors in, raw pointers out).

These are thin marshalling helpers: torch provides device memory and the current
HIP stream, the arithmetic is entirely in libactionmesh_amd.so.  Used by the
parity tests and by the attention-processor seam (S3).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import _lib as L

HEAD_DIM = 128


def _stream(device=None) -> int:
    """torch's current stream of `device` (default: the current device)."""
    return torch.cuda.current_stream(device).cuda_stream


def _launch(t: torch.Tensor, fn, what: str, *args) -> None:
    """Call a C-ABI kernel entry point on the device that owns `t`: the device is made current for the call (the
    library launches on the current device) and the stream is torch's current stream OF THAT DEVICE, so a tensor on a
    non-current device is neither launched on device 0's stream nor unordered with the work that produced it."""
    with torch.cuda.device(t.device):
        L.check(fn(*args, _stream(t.device)), what, getattr(fn, "_am_lib", None))


H16 = (torch.bfloat16, torch.float16)       # the two 16-bit storage types: each has its own build of the library (_lib.lib(kind))


def _fn(t_or_dtype, name: str):
    """Entry point `name` of the library whose 16-bit type is that of the tensor (or dtype) given: bfloat16 -> libactionmesh_amd.so,
    float16 -> libactionmesh_amd_f16.so (the same sources built with -DAM_F16)."""
    dt = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    l = L.lib("f16" if dt == torch.float16 else "bf16")
    f = getattr(l, name)
    f._am_lib = l
    return f


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: actionmesh_amd kernels need a device tensor (no CPU path)")
    if (t.dtype not in dtype) if isinstance(dtype, tuple) else (t.dtype != dtype):
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def perm16_index(n: int, device=None) -> torch.Tensor:
    """position -> key index map of the V^T layout (bit2 <-> bit3 inside each 16-group)."""
    k = torch.arange(n, device=device)
    return (k & ~0xC) | ((k & 4) << 1) | ((k & 8) >> 1)


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
         residual: Optional[torch.Tensor] = None, gelu: bool = False,
         a2: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
         a_map: Tuple[int, int, int] = (0, 0, 0), c_map: Tuple[int, int, int] = (0, 0, 0),
         M: Optional[int] = None, force_small: bool = False, legacy: bool = False,
         force_big: bool = False, ablate: int = 0,
         ln: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, ln_part: Optional[torch.Tensor] = None,
         gelu_table: bool = True) -> torch.Tensor:
    """out = act(cat(a, a2) @ w.T + bias) + residual   (bf16, fp32 accumulate).

    a (Ma, K1), a2 (Ma, K2) optional, w (N, K1+K2), bias fp32 (N,), residual/out (Mc, N).
    a_map / c_map = (G, group_stride, offset) row maps (G=0: identity).
    ln = (stats (M, 2) fp32, colsum (N,) fp32): the LayerNorm of `a` folded into the linear (w, bias from ln_fold_weight; see
    am_gemm_args in include/actionmesh_amd.h).  ln_part (M, ceil(N / 256), 2) fp32: receives the per-slice (mean, M2) of the output rows."""
    _need(a, H16, "a"); _need(w, a.dtype, "w")
    K1 = a.shape[1]
    K = w.shape[1]
    N = w.shape[0]
    if M is None:
        M = a.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=a.dtype, device=a.device)
    _need(out, a.dtype, "out")
    g = L.AmGemmArgs()
    g.A1 = a.data_ptr(); g.lda1 = a.stride(0); g.K1 = K1
    g.A2 = _p(a2); g.lda2 = a2.stride(0) if a2 is not None else 0
    if a2 is not None:
        _need(a2, a.dtype, "a2")
        assert K1 + a2.shape[1] == K
    else:
        assert K1 == K
    g.W = w.data_ptr(); g.ldw = w.stride(0)
    g.bias = _p(_need(bias, torch.float32, "bias")) if bias is not None else None
    g.residual = _p(_need(residual, a.dtype, "residual")) if residual is not None else None
    g.C = out.data_ptr(); g.ldc = out.stride(0)
    g.M, g.N, g.K = M, N, K
    # 0x100: force the 128x128 register-staged kernel (the small-problem path; tests compare the two tilings);
    # 0x200: the round-1 lockstep main loop of the 256x256 tile (same-box A/B); 0x400: the 256x256 tile at any grid size (tests)
    # 0x10000: the arithmetic GELU epilogue in the 256x256 tile too (default: the bit-identical LDS table, am_gemm.hip GT_LO)
    g.act = (1 if gelu else 0) | (0x100 if force_small else 0) | (0x200 if legacy else 0) | (0x400 if force_big else 0) | (ablate & 0xF800) \
        | (0 if gelu_table else 0x10000)
    g.a_G, g.a_gs, g.a_off = a_map
    g.c_G, g.c_gs, g.c_off = c_map
    if ln is not None:
        g.ln_stats = _need(ln[0], torch.float32, "ln stats").data_ptr()
        g.ln_colsum = _need(ln[1], torch.float32, "ln colsum").data_ptr()
        assert ln[0].numel() >= 2 * M and ln[1].numel() == N
    if ln_part is not None:
        _need(ln_part, torch.float32, "ln_part")
        assert ln_part.numel() >= 2 * M * ((N + 255) // 256)
        g.ln_part = ln_part.data_ptr()
    _launch(a, _fn(a, "am_gemm_bf16"), "am_gemm_bf16", C.byref(g))
    return out


def row_stats(x: torch.Tensor, eps: float = 1e-5, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(mean, rstd) per row of x (rows, C): fp32 (rows, 2) - the statistics layernorm() uses, for gemm(..., ln=...)."""
    _need(x, H16, "x")
    Cdim = x.shape[-1]
    rows = x.numel() // Cdim
    if out is None:
        out = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    _launch(x, _fn(x, "am_row_stats_bf16"), "am_row_stats_bf16", x.data_ptr(), _need(out, torch.float32, "out").data_ptr(), rows, Cdim, eps)
    return out


def row_stats_finalize(part: torch.Tensor, Cdim: int, eps: float = 1e-5, out: Optional[torch.Tensor] = None, kind=torch.bfloat16) -> torch.Tensor:
    """part (rows, ceil(Cdim / 256), 2) fp32 pairs (mean, M2) of 256-column slices (gemm(..., ln_part=...)) -> (rows, 2) (mean, rstd)."""
    _need(part, torch.float32, "part")
    rows, nparts = part.shape[0], part.shape[1]
    if out is None:
        out = torch.empty((rows, 2), dtype=torch.float32, device=part.device)
    _launch(part, _fn(kind, "am_row_stats_finalize"), "am_row_stats_finalize", part.data_ptr(), nparts, Cdim, out.data_ptr(), rows, eps)
    return out


def ln_fold_weight(w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, bias: Optional[torch.Tensor] = None):
    """One-time preparation of a linear that absorbs the LayerNorm in front of it: returns (wf, colsum, d) for gemm(a, wf, bias=d,
    ln=(row_stats(a), colsum)) == gemm(layernorm(a, gamma, beta), w, bias) up to the bf16 rounding of the normalised activation."""
    _need(w, H16, "w"); _need(gamma, torch.float32, "gamma"); _need(beta, torch.float32, "beta")
    N, K = w.shape
    wf = torch.empty_like(w)
    colsum = torch.empty((N,), dtype=torch.float32, device=w.device)
    d = torch.empty((N,), dtype=torch.float32, device=w.device)
    _launch(w, _fn(w, "am_ln_fold_weight"), "am_ln_fold_weight", w.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
            _p(_need(bias, torch.float32, "bias")) if bias is not None else None, wf.data_ptr(), colsum.data_ptr(), d.data_ptr(), N, K)
    return wf, colsum, d


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5,
              out: Optional[torch.Tensor] = None, stats_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """stats_out (rows, 2) fp32: also receives (mean, rstd) of the OUTPUT rows (for a linear that folds the next LayerNorm)."""
    _need(x, H16, "x"); _need(w, torch.float32, "w"); _need(b, torch.float32, "b")
    Cdim = x.shape[-1]
    rows = x.numel() // Cdim
    if out is None:
        out = torch.empty_like(x)
    if stats_out is not None:
        _launch(x, _fn(x, "am_layernorm_stats_bf16"), "am_layernorm_stats_bf16", x.data_ptr(), out.data_ptr(), w.data_ptr(), b.data_ptr(),
                rows, Cdim, eps, _need(stats_out, torch.float32, "stats_out").data_ptr())
        return out
    _launch(x, _fn(x, "am_layernorm_bf16"), "am_layernorm_bf16", x.data_ptr(), out.data_ptr(), w.data_ptr(), b.data_ptr(),
                                      rows, Cdim, eps)
    return out


def add_layernorm_f32(h32: torch.Tensor, y: Optional[torch.Tensor] = None, w: Optional[torch.Tensor] = None,
                      b: Optional[torch.Tensor] = None, eps: float = 1e-5, out: Optional[torch.Tensor] = None,
                      dtype: Optional[torch.dtype] = None) -> Optional[torch.Tensor]:
    """am_add_layernorm_f32 - the fp32 residual stream of the reference's Stage II / DINOv2 encoder in one pass:
    h32 (rows, C) fp32 += y (rows, C) 16-bit (None: nothing to add), IN PLACE; returns LayerNorm(h32) * w + b rounded to the 16-bit type
    (`dtype`, or y's) for the next linear - or None when w is None (accumulate only)."""
    _need(h32, torch.float32, "h32")
    Cdim = h32.shape[-1]
    rows = h32.numel() // Cdim
    if y is not None:
        _need(y, H16, "y")
        assert y.numel() == h32.numel()
        dtype = y.dtype
    if w is None:
        assert y is not None, "add_layernorm_f32: nothing to do"
        _launch(h32, _fn(dtype, "am_add_layernorm_f32"), "am_add_layernorm_f32", h32.data_ptr(), y.data_ptr(), None, None, None, rows, Cdim, eps)
        return None
    assert dtype in H16, "add_layernorm_f32: the 16-bit output type must be given when there is no branch to take it from"
    _need(w, torch.float32, "w"); _need(b, torch.float32, "b")
    if out is None:
        out = torch.empty(h32.shape, dtype=dtype, device=h32.device)
    _launch(h32, _fn(dtype, "am_add_layernorm_f32"), "am_add_layernorm_f32", h32.data_ptr(), _p(y), _need(out, dtype, "out").data_ptr(),
            w.data_ptr(), b.data_ptr(), rows, Cdim, eps)
    return out


def head_post(x: torch.Tensor, heads: int, kinds: Sequence[int], seq_len: int, rows_per_frame: int,
              w_q: Optional[torch.Tensor] = None, w_k: Optional[torch.Tensor] = None,
              rope: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, eps: float = 1e-6,
              out_q: Optional[torch.Tensor] = None, out_k: Optional[torch.Tensor] = None,
              out_vt: Optional[torch.Tensor] = None):
    """Split heads of x (rows, heads*len(kinds)*128), apply qk-RMSNorm (+RoPE), and write the
    attention operand layouts.  Returns (Q, K, Vt) (None for absent kinds):
      Q  (nseq, H, sq_pad, 128), K (nseq, H, sk_pad, 128), Vt (nseq, H, 128, sk_pad)."""
    _need(x, H16, "x")
    rows = x.shape[0]
    nseq = rows // seq_len
    sq_pad, sk_pad = round_up(seq_len, 256), round_up(seq_len, 64)
    dev = x.device
    a = L.AmHeadPostArgs()
    a.X = x.data_ptr(); a.ldx = x.stride(0)
    a.rows = rows; a.seq_len = seq_len; a.rows_per_frame = rows_per_frame
    a.heads = heads; a.nparts = len(kinds)
    for i, k in enumerate(kinds):
        a.kinds[i] = k
    a.w_q = _p(w_q); a.w_k = _p(w_k); a.eps = eps
    if rope is not None:
        a.rope_cos = _need(rope[0], torch.float32, "rope_cos").data_ptr()
        a.rope_sin = _need(rope[1], torch.float32, "rope_sin").data_ptr()
    if 0 in kinds and out_q is None:
        out_q = torch.zeros((nseq, heads, sq_pad, HEAD_DIM), dtype=x.dtype, device=dev)
    if 1 in kinds and out_k is None:
        out_k = torch.zeros((nseq, heads, sk_pad, HEAD_DIM), dtype=x.dtype, device=dev)
    if 2 in kinds and out_vt is None:
        out_vt = torch.zeros((nseq, heads, HEAD_DIM, sk_pad), dtype=x.dtype, device=dev)
    a.out_q = _p(out_q); a.sq_pad = out_q.shape[2] if out_q is not None else 0
    a.out_k = _p(out_k); a.out_vt = _p(out_vt)
    a.sk_pad = out_k.shape[2] if out_k is not None else (out_vt.shape[3] if out_vt is not None else 0)
    _launch(x, _fn(x, "am_head_post"), "am_head_post", C.byref(a))
    return out_q, out_k, out_vt


def gemm_head_post(a: torch.Tensor, w: torch.Tensor, heads: int, kinds: Sequence[int], seq_len: int, rows_per_frame: int,
                   w_q: Optional[torch.Tensor] = None, w_k: Optional[torch.Tensor] = None,
                   rope: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, eps: float = 1e-6,
                   out_q: Optional[torch.Tensor] = None, out_k: Optional[torch.Tensor] = None,
                   out_vt: Optional[torch.Tensor] = None, x: Optional[torch.Tensor] = None,
                   bias: Optional[torch.Tensor] = None, ln: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, ablate: int = 0):
    """am_gemm_headpost_bf16: (a @ w.T) -> head split / qk-RMSNorm / RoPE / attention layouts in ONE launch; the arguments of `gemm`
    (no bias, no activation) and of `head_post`.  `x` (rows, N) is the linear's output buffer the un-fused pair would use (only the
    tile grid's remainder rows are written to it).  Returns (Q, K, Vt) like head_post."""
    _need(a, H16, "a"); _need(w, a.dtype, "w")
    rows, K = a.shape
    N = w.shape[0]
    assert N == heads * len(kinds) * HEAD_DIM and w.shape[1] == K
    if x is None:
        x = torch.empty((rows, N), dtype=a.dtype, device=a.device)
    nseq = rows // seq_len
    sq_pad, sk_pad = round_up(seq_len, 256), round_up(seq_len, 64)
    dev = a.device
    g = L.AmGemmArgs()
    g.A1 = a.data_ptr(); g.lda1 = a.stride(0); g.K1 = K
    g.W = w.data_ptr(); g.ldw = w.stride(0)
    g.C = x.data_ptr(); g.ldc = x.stride(0)
    g.M, g.N, g.K = rows, N, K
    g.act = ablate & 0x1800      # timing ablations of the fused epilogue (0x800: no Q / K rows, 0x1000: no V^T read-back); never on the product path
    if bias is not None:
        g.bias = _need(bias, torch.float32, "bias").data_ptr()
    if ln is not None:       # LayerNorm folded into the projection (gemm's `ln`)
        g.ln_stats = _need(ln[0], torch.float32, "ln stats").data_ptr()
        g.ln_colsum = _need(ln[1], torch.float32, "ln colsum").data_ptr()
    h = L.AmHeadPostArgs()
    h.X = x.data_ptr(); h.ldx = x.stride(0)
    h.rows = rows; h.seq_len = seq_len; h.rows_per_frame = rows_per_frame
    h.heads = heads; h.nparts = len(kinds)
    for i, k in enumerate(kinds):
        h.kinds[i] = k
    h.w_q = _p(w_q); h.w_k = _p(w_k); h.eps = eps
    if rope is not None:
        h.rope_cos = _need(rope[0], torch.float32, "rope_cos").data_ptr()
        h.rope_sin = _need(rope[1], torch.float32, "rope_sin").data_ptr()
    if 0 in kinds and out_q is None:
        out_q = torch.zeros((nseq, heads, sq_pad, HEAD_DIM), dtype=a.dtype, device=dev)
    if 1 in kinds and out_k is None:
        out_k = torch.zeros((nseq, heads, sk_pad, HEAD_DIM), dtype=a.dtype, device=dev)
    if 2 in kinds and out_vt is None:
        out_vt = torch.zeros((nseq, heads, HEAD_DIM, sk_pad), dtype=a.dtype, device=dev)
    h.out_q = _p(out_q); h.sq_pad = out_q.shape[2] if out_q is not None else 0
    h.out_k = _p(out_k); h.out_vt = _p(out_vt)
    h.sk_pad = out_k.shape[2] if out_k is not None else (out_vt.shape[3] if out_vt is not None else 0)
    _launch(a, _fn(a, "am_gemm_headpost_bf16"), "am_gemm_headpost_bf16", C.byref(g), C.byref(h))
    return out_q, out_k, out_vt


STATE_LD = 132   # floats per row of a two-pass attention state: O[128], m, l, pad


def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, sq: int, sk: int,
              out: Optional[torch.Tensor] = None, nchunks: int = 1, defer_log2: int = 8,
              scale: Optional[float] = None, rows: int = 0, state_mode: int = 0,
              state: Optional[torch.Tensor] = None, chunk_first: int = 0, chunk_total: int = 0) -> torch.Tensor:
    """softmax(q k^T * scale) v on pre-laid-out operands.
      q (nseq, H, sq_pad, 128); k ([chunks,] nseq, H, sk_pad, 128); vt ([chunks,] nseq, H, 128, sk_pad)
      -> out (nseq * sq, H * 128)
    Two-pass form (am_attn_args in include/actionmesh_amd.h): `rows` selects the query blocks (1 = the full blocks,
    2 = the rest), `state_mode` 1 saves / 2 resumes the (O, m, l) of the full blocks in `state`
    (nseq * H, sq_pad, STATE_LD) fp32, and the chunks walked are (chunk_first + i) % chunk_total."""
    _need(q, H16, "q"); _need(k, q.dtype, "k"); _need(vt, q.dtype, "vt")
    nseq, H, sq_pad, _ = q.shape
    sk_pad = k.shape[-2]
    if out is None:
        out = torch.empty((nseq * sq, H * HEAD_DIM), dtype=q.dtype, device=q.device)
    a = L.AmAttnArgs()
    a.Q, a.K, a.Vt, a.O = q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr()
    a.nseq, a.heads, a.sq, a.sq_pad, a.sk, a.sk_pad = nseq, H, sq, sq_pad, sk, sk_pad
    a.nchunks = nchunks
    a.chunk_stride = nseq * H * sk_pad * HEAD_DIM
    a.ldo = out.stride(0)
    a.scale = scale if scale is not None else HEAD_DIM ** -0.5
    a.defer_log2 = defer_log2
    a.rows, a.state_mode, a.chunk_first, a.chunk_total = rows, state_mode, chunk_first, chunk_total
    if state is not None:
        _need(state, torch.float32, "state")
        assert state.numel() >= nseq * H * sq_pad * STATE_LD
        a.state = state.data_ptr()
    _launch(q, _fn(q, "am_attention_bf16"), "am_attention_bf16", C.byref(a))
    return out


def attention_fp8(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, sq: int, sk: int,
                  out: Optional[torch.Tensor] = None, nchunks: int = 1, scale: Optional[float] = None,
                  quantized: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None, ablate: int = 0,
                  rows: int = 0, state_mode: int = 0, state: Optional[torch.Tensor] = None, chunk_first: int = 0,
                  chunk_total: int = 0, **_ignored) -> torch.Tensor:
    """fp8 (e4m3) attention on the bf16 operand layouts of `attention`: quantise (unless `quantized` = (q8, k8, vt8) from an
    earlier call is passed), then QK^T / P.V on the MX-scaled fp8 MFMA.  Returns out (nseq * sq, H * 128) bf16.
    rows / state_mode / state / chunk_first / chunk_total: the two-pass forms, as in `attention`."""
    _need(q, H16, "q"); _need(k, q.dtype, "k"); _need(vt, q.dtype, "vt")
    nseq, H, sq_pad, _ = q.shape
    sk_pad = k.shape[-2]
    if out is None:
        out = torch.empty((nseq * sq, H * HEAD_DIM), dtype=q.dtype, device=q.device)
    a = L.AmAttnArgs()
    a.Q, a.K, a.Vt, a.O = q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr()
    a.nseq, a.heads, a.sq, a.sq_pad, a.sk, a.sk_pad = nseq, H, sq, sq_pad, sk, sk_pad
    a.nchunks = nchunks
    a.chunk_stride = nseq * H * sk_pad * HEAD_DIM
    a.ldo = out.stride(0)
    a.scale = scale if scale is not None else HEAD_DIM ** -0.5
    a.defer_log2 = 5000 + ablate if ablate else 0        # timing ablations of the kernel (tools/kernel_bench.py)
    a.rows, a.state_mode, a.chunk_first, a.chunk_total = rows, state_mode, chunk_first, chunk_total
    if state is not None:
        _need(state, torch.float32, "state")
        assert state.numel() >= nseq * H * sq_pad * STATE_LD
        a.state = state.data_ptr()
    if quantized is None:
        q8 = torch.empty(q.shape, dtype=torch.uint8, device=q.device)
        k8 = torch.empty(k.shape, dtype=torch.uint8, device=q.device)
        vt8 = torch.empty(vt.shape, dtype=torch.uint8, device=q.device)
        _launch(q, _fn(q, "am_attention_quantize_fp8"), "am_attention_quantize_fp8", C.byref(a), q8.data_ptr(), k8.data_ptr(),
                vt8.data_ptr())
    else:
        q8, k8, vt8 = quantized
    _launch(q, _fn(q, "am_attention_fp8"), "am_attention_fp8", C.byref(a), q8.data_ptr(), k8.data_ptr(), vt8.data_ptr())
    attention_fp8.last_quantized = (q8, k8, vt8)
    return out


def attention_fallback_count() -> int:
    """Workgroups the exact fallback of the lazy attention kernel has recomputed so far on the current device."""
    n = C.c_uint64(0)
    L.check(L.lib().am_attention_fallback_count(C.byref(n)), "am_attention_fallback_count")
    return int(n.value)


def f32_to_bf16(x: torch.Tensor, dtype=torch.bfloat16) -> torch.Tensor:
    """fp32 -> the 16-bit type (round to nearest even); `dtype=torch.float16` runs the float16 build of the library."""
    _need(x, torch.float32, "x")
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    _launch(x, _fn(dtype, "am_f32_to_bf16"), "am_f32_to_bf16", x.data_ptr(), y.data_ptr(), x.numel())
    return y


def timestep_sinusoid(t: torch.Tensor, width: int) -> torch.Tensor:
    _need(t, torch.float32, "t")
    y = torch.empty((t.numel(), width), dtype=torch.bfloat16, device=t.device)
    _launch(t, L.lib().am_timestep_sinusoid, "am_timestep_sinusoid", t.data_ptr(), y.data_ptr(), t.numel(), width)
    return y


def point_embed(query: torch.Tensor, in_channels: int, extra_channels: int, num_freqs: int, include_pi: bool,
                ld_out: int = 64, dtype=torch.bfloat16) -> torch.Tensor:
    """query (rows, >= in+extra) fp32 -> 16-bit (rows, ld_out): FrequencyPositionalEmbedding + extras, zero padded."""
    _need(query, torch.float32, "query")
    rows = query.shape[0]
    out = torch.empty((rows, ld_out), dtype=dtype, device=query.device)
    _launch(query, _fn(dtype, "am_point_embed"), "am_point_embed", query.data_ptr(), query.stride(0), rows, in_channels, extra_channels, num_freqs,
                                   int(include_pi), out.data_ptr(), ld_out)
    return out


def patchify(pixels: torch.Tensor, patch: int, ld_out: int, dtype=torch.bfloat16) -> torch.Tensor:
    """pixels (T, C, H, W) fp32 -> 16-bit (T * (H // patch) * (W // patch), ld_out): rows of the kernel = stride patch
    convolution in flattened-Conv2d-weight column order, zero padded."""
    _need(pixels, torch.float32, "pixels")
    T, Cin, H, W = pixels.shape
    out = torch.empty((T * (H // patch) * (W // patch), ld_out), dtype=dtype, device=pixels.device)
    _launch(pixels, _fn(dtype, "am_patchify"), "am_patchify", pixels.data_ptr(), T, Cin, H, W, patch, out.data_ptr(), ld_out)
    return out


def displacement(logits: torch.Tensor, out_dim: int, out: torch.Tensor) -> torch.Tensor:
    """out (rows, out_dim) fp32 = 2 sigmoid(-logits[:, :out_dim]) - 1."""
    _need(logits, H16, "logits"); _need(out, torch.float32, "out")
    _launch(logits, _fn(logits, "am_displacement"), "am_displacement", logits.data_ptr(), logits.stride(0), logits.shape[0], out_dim, out.data_ptr())
    return out


def flow_step(v: torch.Tensor, latents: torch.Tensor, scales: Sequence[float], dt: float,
              is_additive: bool, unobserved: Optional[Sequence[bool]]) -> None:
    """In place: latents (T, N, D) fp32 += sign * bf16(dt * cfg(v)); v (n_branches, T, N, D) bf16."""
    _need(v, H16, "v"); _need(latents, torch.float32, "latents")
    nb, T, N, D = v.shape
    sc = (C.c_float * max(1, len(scales)))(*[float(s) for s in scales])
    un = None
    if unobserved is not None:
        un = (C.c_uint8 * T)(*[1 if u else 0 for u in unobserved])
    _launch(v, _fn(v, "am_flow_step"), "am_flow_step", v.data_ptr(), latents.data_ptr(), nb, sc, float(dt), 1 if is_additive else 0,
                                 un, T, N, D)


def nearest_neighbors(points: torch.Tensor, queries: torch.Tensor, precise: bool = True,
                      check: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """am_nn_search: for every query its nearest point (exact, brute force).  points (P, 3) or (B, P, 3) fp32, queries
    (Q, 3) or (B, Q, 3) fp32 (a 2-D operand beside a 3-D one is shared by the batch).  Returns (index int32, SQUARED
    distance: float64 when `precise` - the KD-tree arithmetic of actionbench/chamfer.py - else float32), shaped like the
    queries minus the coordinate axis.
    `check` (default): one device-to-host read behind the search that turns "no finite distance" (index -1) into an error.
    Callers inside a loop of searches (the ICP: 4800 of them) pass check=False and validate once per batch instead
    (`nn_indices_valid`), so the stream is not serialised per search."""
    _need(points, torch.float32, "points"); _need(queries, torch.float32, "queries")
    if points.shape[-1] != 3 or queries.shape[-1] != 3 or points.dim() not in (2, 3) or queries.dim() not in (2, 3):
        raise ValueError(f"nearest_neighbors: expected (..., n, 3) operands, got {tuple(points.shape)} / {tuple(queries.shape)}")
    batch = max(points.shape[0] if points.dim() == 3 else 1, queries.shape[0] if queries.dim() == 3 else 1)
    for name, t in (("points", points), ("queries", queries)):
        if t.dim() == 3 and t.shape[0] != batch:
            raise ValueError(f"nearest_neighbors: {name} batch {t.shape[0]} != {batch}")
    P, Q = points.shape[-2], queries.shape[-2]
    if P == 0 or Q == 0:
        raise ValueError("nearest_neighbors: empty point cloud")
    out_shape = (batch, Q) if (points.dim() == 3 or queries.dim() == 3) else (Q,)
    idx = torch.empty(out_shape, dtype=torch.int32, device=queries.device)
    d2 = torch.empty(out_shape, dtype=torch.float64 if precise else torch.float32, device=queries.device)
    lib = L.lib()
    need = lib.am_nn_workspace_bytes(P, Q, batch, 1 if precise else 0)
    ws = torch.empty((max(need, 1),), dtype=torch.uint8, device=queries.device)
    a = L.AmNnArgs()
    a.points, a.n_points, a.points_bstride = points.data_ptr(), P, (P * 3 if points.dim() == 3 else 0)
    a.queries, a.n_queries, a.queries_bstride = queries.data_ptr(), Q, (Q * 3 if queries.dim() == 3 else 0)
    a.batch, a.precise = batch, 1 if precise else 0
    a.out_index, a.out_d2 = idx.data_ptr(), d2.data_ptr()
    _launch(queries, lib.am_nn_search, "am_nn_search", C.byref(a), ws.data_ptr(), need)
    if check and bool((idx < 0).any()):
        # the kernel's strict `<` never fires for a query whose distances are all NaN: index -1 would wrap in a later gather
        raise ValueError("nearest_neighbors: a query has no finite distance to any point (NaN / inf coordinates in the inputs)")
    return idx, d2


def nn_indices_valid(*indices: torch.Tensor) -> None:
    """The deferred form of nearest_neighbors' check: ONE device-to-host read for any number of index tensors."""
    bad = None
    for i in indices:
        b = (i < 0).any()
        bad = b if bad is None else (bad | b)
    if bad is not None and bool(bad):
        raise ValueError("nearest_neighbors: a query has no finite distance to any point (NaN / inf coordinates in the inputs)")
