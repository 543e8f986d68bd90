"""Per-kernel parity tests (GPU): every HIP kernel against a plain PyTorch fp32 statement of
the same op, called through the C-ABI.  Inputs are asymmetric random data so that operand /
output transposes are detected."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from actionmesh_amd import _lib
    _lib.lib()   # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def rb(x):
    return x.to(torch.bfloat16).float()


def _randn(shape, seed, dev, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def _close(out, ref, ulps=2.0, atol=0.0, what="", mag=None):
    """bf16 result within `ulps` bf16 ulps of the fp32 reference (relative 2^-8 each) + atol.
    `mag`: magnitude that sets the ulp when an intermediate (rounded) value is larger than the
    final result (residual adds can cancel)."""
    out = out.float()
    scale = ref.abs() if mag is None else torch.maximum(ref.abs(), mag)
    tol = ulps * (2.0 ** -8) * scale + atol
    bad = (out - ref).abs() > tol
    assert not bool(bad.any()), (
        f"{what}: {int(bad.sum())}/{bad.numel()} elements off; max abs err "
        f"{float((out - ref).abs().max()):.4e}, ref max {float(ref.abs().max()):.3e}")


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (1000, 384, 256), (128, 64, 64), (37, 128, 64),
                                   (4097, 1024, 1024), (2500, 3072, 512)])
@pytest.mark.parametrize("mode", ["plain", "bias", "bias_gelu", "bias_res", "bias_res_small"])
def test_gemm(dev, M, N, K, mode):
    force_small = mode.endswith("_small")      # 128x128 register-staged kernel at every shape
    mode = mode.replace("_small", "")
    from actionmesh_amd import ops
    a = _randn((M, K), 1, dev).to(torch.bfloat16)
    w = _randn((N, K), 2, dev, 1.0 / math.sqrt(K)).to(torch.bfloat16)
    bias = rb(_randn((N,), 3, dev, 0.5)) if mode != "plain" else None
    res = _randn((M, N), 4, dev).to(torch.bfloat16) if mode == "bias_res" else None
    out = ops.gemm(a, w, bias=bias, residual=res, gelu=(mode == "bias_gelu"), force_small=force_small)
    torch.cuda.synchronize()
    ref = a.float() @ w.float().T
    if bias is not None:
        ref = ref + bias
    ref = rb(ref)
    mag = ref.abs()
    if mode == "bias_gelu":
        ref = rb(F.gelu(ref))
    if res is not None:
        mag = torch.maximum(mag, res.float().abs())
        ref = ref + res.float()
    _close(out, ref, ulps=2.0, atol=2e-3 * math.sqrt(K / 64), what=f"gemm {M}x{N}x{K} {mode}", mag=mag)


def test_gemm_split_a_and_row_maps(dev):
    """cat([skip, h]) @ W^T without materialising the cat (block.py:131-132) and the two row
    maps used for proj_in / time token / proj_out (temporal_denoiser.py:206,217,240-242)."""
    from actionmesh_amd import ops
    M, N, K1, K2 = 517, 256, 128, 192
    a1 = _randn((M, K1), 1, dev).to(torch.bfloat16)
    a2 = _randn((M, K2), 2, dev).to(torch.bfloat16)
    w = _randn((N, K1 + K2), 3, dev, 0.05).to(torch.bfloat16)
    bias = rb(_randn((N,), 4, dev, 0.5))
    out = ops.gemm(a1, w, bias=bias, a2=a2)
    ref = rb(torch.cat([a1, a2], 1).float() @ w.float().T + bias)
    _close(out, ref, atol=4e-3, what="split-A gemm")

    # c_map: logical row r -> frame r//G, physical row frame*gs + off + r%G
    frames, G, L = 5, 48, 49
    a = _randn((frames * G, 64), 5, dev).to(torch.bfloat16)
    w = _randn((256, 64), 6, dev, 0.1).to(torch.bfloat16)
    dst = torch.full((frames * L, 256), 7.0, dtype=torch.bfloat16, device=dev)
    ops.gemm(a, w, out=dst, c_map=(G, L, 1))
    ref = rb(a.float() @ w.float().T).view(frames, G, 256)
    got = dst.view(frames, L, 256)
    _close(got[:, 1:], ref, atol=2e-3, what="c_map rows")
    assert bool((got[:, 0] == 7.0).all()), "row 0 of every frame must be untouched"
    # a_map: read rows 1..G of every frame
    src = _randn((frames * L, 128), 8, dev).to(torch.bfloat16)
    w2 = _randn((64, 128), 9, dev, 0.1).to(torch.bfloat16)
    out2 = ops.gemm(src, w2, a_map=(G, L, 1), M=frames * G)
    ref2 = rb(src.view(frames, L, 128)[:, 1:].reshape(-1, 128).float() @ w2.float().T)
    _close(out2, ref2, atol=2e-3, what="a_map rows")


def test_gemm256_split_a_row_maps_and_tails(dev):
    """The 256x256 DMA-staged kernel (M >= 1024, N >= 256): split A, both row maps, M/N tails."""
    from actionmesh_amd import ops
    frames, G, L = 30, 48, 49
    M, N, K1, K2 = frames * G, 328, 128, 64        # M = 1440 (tail 160 of 256), N = 328 (tail)
    a1 = _randn((frames * L, K1), 1, dev).to(torch.bfloat16)
    a2 = _randn((frames * L, K2), 2, dev).to(torch.bfloat16)
    w = _randn((N, K1 + K2), 3, dev, 0.07).to(torch.bfloat16)
    bias = rb(_randn((N,), 4, dev, 0.5))
    res = _randn((frames * L, N), 5, dev).to(torch.bfloat16)
    dst = res.clone()
    ops.gemm(a1, w, bias=bias, a2=a2, residual=dst, out=dst, a_map=(G, L, 1), c_map=(G, L, 1), M=M, force_big=True)
    torch.cuda.synchronize()
    cat = torch.cat([a1, a2], 1).view(frames, L, K1 + K2)[:, 1:].reshape(M, K1 + K2).float()
    lin = rb(cat @ w.float().T + bias)
    r3 = res.view(frames, L, N).float()
    ref = lin.view(frames, G, N) + r3[:, 1:]
    got = dst.view(frames, L, N)
    _close(got[:, 1:], ref, atol=4e-3, what="gemm256 maps", mag=torch.maximum(lin.view(frames, G, N).abs(), r3[:, 1:].abs()))
    assert torch.equal(got[:, 0], res.view(frames, L, N)[:, 0]), "row 0 of every frame must be untouched"


@pytest.mark.parametrize("M,N,K", [(1024, 512, 64), (1300, 256, 128), (777, 328, 192), (4097, 1024, 1024), (2500, 3072, 512),
                                   (256, 256, 4096), (5000, 520, 320)])
@pytest.mark.parametrize("mode", ["plain", "bias_gelu", "bias_res"])
def test_gemm256_pingpong_main_loop(dev, M, N, K, mode):
    """The 256x256 tile with the two-group ping-pong main loop (the product path of every large linear), forced at modest
    grid sizes: 1 .. 64 k-tiles (prologue / steady state / clamped tail stages), M and N tails, every epilogue.  Against the
    fp32 statement (<= 2 bf16 ulp) and against the round-1 lockstep loop on the same tile (the same products summed in a
    different MFMA shape: 16x16x32 vs 32x32x16 - equal up to fp32 summation order, i.e. <= 1 bf16 ulp of the linear)."""
    from actionmesh_amd import ops
    a = _randn((M, K), 1, dev).to(torch.bfloat16)
    w = _randn((N, K), 2, dev, 1.0 / math.sqrt(K)).to(torch.bfloat16)
    bias = rb(_randn((N,), 3, dev, 0.5)) if mode != "plain" else None
    res = _randn((M, N), 4, dev).to(torch.bfloat16) if mode == "bias_res" else None
    kw = dict(bias=bias, residual=res, gelu=(mode == "bias_gelu"))
    out = ops.gemm(a, w, force_big=True, **kw)
    old = ops.gemm(a, w, force_big=True, legacy=True, **kw)
    torch.cuda.synchronize()
    ref = a.float() @ w.float().T
    if bias is not None:
        ref = ref + bias
    ref = rb(ref)
    mag = ref.abs()
    if mode == "bias_gelu":
        ref = rb(F.gelu(ref))
    if res is not None:
        mag = torch.maximum(mag, res.float().abs())
        ref = ref + res.float()
    # `ulps` are relative (2^-8 |x|: half to one true bf16 spacing).  GELU mode: the bf16-rounded linear may sit one TRUE ulp from
    # the rounded fp32 statement (a rounding-boundary case), GELU' ~ 1 carries that to the activation, and the activation's own
    # rounding can flip once more: two true ulps = up to four relative ones (seen: 1 element in 5e5 at |x| ~ 1.5, error 2^-6).
    _close(out, ref, ulps=4.0 if mode == "bias_gelu" else 2.0, atol=2e-3 * math.sqrt(K / 64), what=f"gemm256pp {M}x{N}x{K} {mode}", mag=mag)
    # GELU is evaluated on the bf16-rounded linear: a 1-ulp difference of the linear moves the activation by up to 1 ulp of ITS input
    _close(out, old.float(), ulps=2.0 if mode == "bias_gelu" else 1.0, atol=1e-3 * math.sqrt(K / 64),
           what=f"gemm256pp vs lockstep {M}x{N}x{K} {mode}", mag=mag)
    assert torch.equal(out, ops.gemm(a, w, force_big=True, **kw)), "run-to-run bits"


@pytest.mark.parametrize("scale,ln", [(1.0, False), (3.5, False), (0.02, False), (1.0, True)])
def test_gelu_table_is_bit_identical(dev, scale, ln):
    """Round 5: the 256x256 tile evaluates F.gelu by a 10 KiB LDS table of f2bf(gelu_erf(x)) over the bf16 values with 2^-17 <= |x| < 8
    (am_gemm.hip GT_LO) instead of ~28 VALU slots per element; values outside take the arithmetic form.  Same bits as the arithmetic
    epilogue (gelu_table=False) on: unit-scale pre-activations, wide ones (|x| >= 8 in ~2 % of the entries), tiny ones, exact zeros
    (zero rows x zero bias), -0.0 / huge / NaN-free extremes planted through the bias, and under a folded LayerNorm."""
    from actionmesh_amd import ops
    M, N, K = 2048 + 256, 1024, 512
    a = _randn((M, K), 11, dev, scale).to(torch.bfloat16)
    a[5] = 0                                                    # exact-zero rows
    a[300:310] = 0
    w = _randn((N, K), 12, dev, K ** -0.5).to(torch.bfloat16)
    bias = _randn((N,), 13, dev, 0.5 * scale)
    bias[:8] = torch.tensor([0.0, -0.0, 9.0, -9.0, 300.0, -300.0, 1e-6, -1e-6], device=dev)
    kw = {}
    if ln:
        gamma = torch.rand(K, device=dev) + 0.5
        beta = torch.randn(K, device=dev) * 0.2
        wf, colsum, d = ops.ln_fold_weight(w, gamma, beta, bias)
        w, bias, kw = wf, d, dict(ln=(ops.row_stats(a), colsum))
    tab = ops.gemm(a, w, bias=bias, gelu=True, force_big=True, **kw)
    ari = ops.gemm(a, w, bias=bias, gelu=True, force_big=True, gelu_table=False, **kw)
    torch.cuda.synchronize()
    assert torch.equal(tab.view(torch.int16), ari.view(torch.int16)), f"{int((tab.view(torch.int16) != ari.view(torch.int16)).sum())} elements differ"
    pre = ops.gemm(a, w, bias=bias, force_big=True, **kw).float()
    frac_out = float(((pre.abs() >= 8) | (pre.abs() < 2.0 ** -17)).float().mean())
    print(f"gelu table scale {scale} ln {ln}: {frac_out:.2e} of the pre-activations outside the table")
    ref = torch.nn.functional.gelu(pre)
    assert float((tab.float() - ref).abs().max()) <= 2.0 ** -8 * max(1.0, float(ref.abs().max()))


def test_gemm_in_place_residual(dev):
    from actionmesh_amd import ops
    M, N, K = 700, 256, 256
    a = _randn((M, K), 1, dev).to(torch.bfloat16)
    w = _randn((N, K), 2, dev, 0.06).to(torch.bfloat16)
    h = _randn((M, N), 3, dev).to(torch.bfloat16)
    ref = rb(a.float() @ w.float().T) + h.float()
    ops.gemm(a, w, residual=h, out=h)
    _close(h, ref, atol=4e-3, what="in-place residual")


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,C", [(50, 256), (4097, 1024), (1031, 2048), (64, 4096)])
def test_layernorm(dev, rows, C):
    from actionmesh_amd import ops
    x = (_randn((rows, C), 1, dev) * 2.0 + 0.5).to(torch.bfloat16)
    w = _randn((C,), 2, dev) * 0.2 + 1.0
    b = _randn((C,), 3, dev) * 0.2
    out = ops.layernorm(x, w, b, 1e-5)
    ref = F.layer_norm(x.float(), (C,), w, b, 1e-5)
    _close(out, ref, ulps=1.5, atol=1e-5, what=f"layernorm {rows}x{C}")


# ------------------------------------------------------------------------------------------
def _headpost_ref(x, heads, nparts, part, w, rope, seq_len, rpf):
    rows = x.shape[0]
    xs = x.float().view(rows, heads, nparts, 128)[:, :, part]          # (rows, H, 128)
    if w is not None:
        xs = xs * torch.rsqrt(xs.pow(2).mean(-1, keepdim=True) + 1e-6) * w
    if rope is not None:
        cos, sin = rope
        fr = torch.arange(rows, device=x.device) // rpf
        c = cos[fr].repeat_interleave(2, dim=-1)[:, None]
        s = sin[fr].repeat_interleave(2, dim=-1)[:, None]
        xr, xi = xs.reshape(rows, heads, 64, 2).unbind(-1)
        rot = torch.stack([-xi, xr], -1).flatten(2)
        xs = xs * c + rot * s
    nseq = rows // seq_len
    return xs.view(nseq, seq_len, heads, 128).permute(0, 2, 1, 3)      # (nseq, H, S, 128)


@pytest.mark.parametrize("nseq,frames_per_seq,L,heads", [(2, 4, 49, 2), (1, 3, 70, 3), (6, 1, 130, 2)])
def test_head_post_self(dev, nseq, frames_per_seq, L, heads):
    from actionmesh_amd import ops
    seq_len = frames_per_seq * L
    rows = nseq * seq_len
    x = _randn((rows, heads * 3 * 128), 1, dev).to(torch.bfloat16)
    wq = _randn((128,), 2, dev) * 0.2 + 1.0
    wk = _randn((128,), 3, dev) * 0.2 + 1.0
    nfr = nseq * frames_per_seq
    ang = _randn((nfr, 64), 4, dev) * 3.0
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    q, k, vt = ops.head_post(x, heads, (0, 1, 2), seq_len, L, w_q=wq, w_k=wk, rope=(cos, sin))
    torch.cuda.synchronize()
    qr = _headpost_ref(x, heads, 3, 0, wq, (cos, sin), seq_len, L)
    kr = _headpost_ref(x, heads, 3, 1, wk, (cos, sin), seq_len, L)
    vr = _headpost_ref(x, heads, 3, 2, None, None, seq_len, L)
    _close(q[:, :, :seq_len], qr, ulps=1.5, atol=1e-5, what="head_post Q")
    _close(k[:, :, :seq_len], kr, ulps=1.5, atol=1e-5, what="head_post K")
    assert bool((q[:, :, seq_len:] == 0).all()) and bool((k[:, :, seq_len:] == 0).all())
    sk_pad = vt.shape[-1]
    idx = ops.perm16_index(sk_pad, dev)
    vpad = torch.zeros((nseq, heads, sk_pad, 128), device=dev)
    vpad[:, :, :seq_len] = vr
    assert torch.equal(vt.float(), vpad[:, :, idx].transpose(-1, -2).contiguous()), "V^T layout / perm16"


def test_head_post_cross(dev):
    """cross branch: q only (norm, no rope) and [k|v] pairs from the context (attention_processor.py:111-115)."""
    from actionmesh_amd import ops
    heads, L, S, BT = 2, 49, 9, 5
    xq = _randn((BT * L, heads * 128), 1, dev).to(torch.bfloat16)
    wq = _randn((128,), 2, dev) * 0.2 + 1.0
    q, _, _ = ops.head_post(xq, heads, (0,), L, L, w_q=wq)
    _close(q[:, :, :L], _headpost_ref(xq, heads, 1, 0, wq, None, L, L), ulps=1.5, atol=1e-5, what="cross Q")
    xkv = _randn((BT * S, heads * 2 * 128), 3, dev).to(torch.bfloat16)
    wk = _randn((128,), 4, dev) * 0.2 + 1.0
    _, k, vt = ops.head_post(xkv, heads, (1, 2), S, S, w_k=wk)
    _close(k[:, :, :S], _headpost_ref(xkv, heads, 2, 0, wk, None, S, S), ulps=1.5, atol=1e-5, what="cross K")
    vr = _headpost_ref(xkv, heads, 2, 1, None, None, S, S)
    idx = ops.perm16_index(vt.shape[-1], dev)
    vpad = torch.zeros((BT, heads, vt.shape[-1], 128), device=dev)
    vpad[:, :, :S] = vr
    assert torch.equal(vt.float(), vpad[:, :, idx].transpose(-1, -2).contiguous())


@pytest.mark.parametrize("case", ["self_qkv_tail_and_straddle", "cross_q", "no_norm_no_rope", "misaligned_falls_back", "small_falls_back"])
def test_gemm_headpost_fused_is_bit_identical(dev, case):
    """am_gemm_headpost_bf16 (round 3: north_star's fused RMSNorm + RoPE + QKV - the head split in the GEMM's epilogue) against
    am_gemm_bf16 followed by am_head_post: every byte of Q, K and V^T - including the pad rows / columns, and nothing written outside
    (the outputs start from a sentinel).  Shapes: (a) two sequences of 8208 rows whose boundary falls inside a 256-row tile, 32
    remainder rows behind the tile grid, q | k | v slices of two heads (tiles hold q|k, v|q, k|v); (b) the cross-attention q form
    (one part, odd sequence length); (c) no qk-norm, no RoPE; (d, e) shapes that must take the un-fused fallback inside the entry point."""
    from actionmesh_amd import ops
    if case == "cross_q":
        heads, kinds, T, Lr, nseq_frames = 4, (0,), 48, 513, True
    elif case == "misaligned_falls_back":
        heads, kinds, T, Lr, nseq_frames = 2, (0, 1, 2), 15, 513, False          # seq_len = 7695: key groups would straddle sequences
    elif case == "small_falls_back":
        heads, kinds, T, Lr, nseq_frames = 2, (0, 1, 2), 4, 512, False
    else:
        heads, kinds, T, Lr, nseq_frames = 2, (0, 1, 2), 16, 513, False
    Cw = 256
    B = 1 if nseq_frames else 2
    seq_len = Lr if nseq_frames else T * Lr
    rows = B * T * Lr
    N = heads * len(kinds) * 128
    a = _randn((rows, Cw), 1, dev).to(torch.bfloat16)
    w = _randn((N, Cw), 2, dev, Cw ** -0.5).to(torch.bfloat16)
    norm = case != "no_norm_no_rope"
    wq = (_randn((128,), 3, dev) * 0.2 + 1.0) if norm else None
    wk = (_randn((128,), 4, dev) * 0.2 + 1.0) if norm else None
    rope = None
    if norm and not nseq_frames:
        ang = torch.arange(B * T, device=dev)[:, None].float() * (10000.0 ** (-torch.arange(64, device=dev).float() * 2 / 128))[None]
        rope = (torch.cos(ang).contiguous(), torch.sin(ang).contiguous())
    nseq = rows // seq_len
    sq_pad, sk_pad = ops.round_up(seq_len, 256), ops.round_up(seq_len, 64)

    def outs():
        s_ = torch.full((1,), -3.0, dtype=torch.bfloat16, device=dev)
        return (s_.expand(nseq, heads, sq_pad, 128).contiguous() if 0 in kinds else None,
                s_.expand(nseq, heads, sk_pad, 128).contiguous() if 1 in kinds else None,
                s_.expand(nseq, heads, 128, sk_pad).contiguous() if 2 in kinds else None)

    q0, k0, v0 = outs()
    x = ops.gemm(a, w)
    ops.head_post(x, heads, kinds, seq_len, Lr, w_q=wq, w_k=wk, rope=rope, out_q=q0, out_k=k0, out_vt=v0)
    q1, k1, v1 = outs()
    ops.gemm_head_post(a, w, heads, kinds, seq_len, Lr, w_q=wq, w_k=wk, rope=rope, out_q=q1, out_k=k1, out_vt=v1)
    torch.cuda.synchronize()
    for nm, r, f in (("Q", q0, q1), ("K", k0, k1), ("V^T", v0, v1)):
        if r is not None:
            bad = (r.view(torch.int16) != f.view(torch.int16))
            assert not bool(bad.any()), f"{case}: {nm} differs in {int(bad.sum())} elements, first at {bad.nonzero()[0].tolist()}"
    if 1 in kinds and sk_pad > seq_len:
        assert bool((k1[:, :, seq_len:] == 0).all()) and bool((v1[..., seq_len:] == 0).all()), "pad rows / columns must be zero"


# ------------------------------------------------------------------------------------------
def _layout(q, k, v, nchunks=1):
    """(nseq,H,S,128) tensors -> padded kernel operands; keys split in `nchunks` equal chunks."""
    from actionmesh_amd import ops
    nseq, H, sq, _ = q.shape
    sk = k.shape[2]
    assert sk % nchunks == 0
    skc = sk // nchunks
    sq_pad, sk_pad = ops.round_up(sq, 256), ops.round_up(skc, 64)
    dev = q.device
    Q = torch.zeros((nseq, H, sq_pad, 128), dtype=torch.bfloat16, device=dev)
    Q[:, :, :sq] = q
    K = torch.zeros((nchunks, nseq, H, sk_pad, 128), dtype=torch.bfloat16, device=dev)
    Vt = torch.zeros((nchunks, nseq, H, 128, sk_pad), dtype=torch.bfloat16, device=dev)
    idx = ops.perm16_index(sk_pad, dev)
    for c in range(nchunks):
        K[c, :, :, :skc] = k[:, :, c * skc:(c + 1) * skc]
        vp = torch.zeros((nseq, H, sk_pad, 128), dtype=torch.bfloat16, device=dev)
        vp[:, :, :skc] = v[:, :, c * skc:(c + 1) * skc]
        Vt[c] = vp[:, :, idx].transpose(-1, -2)
    return Q, K, Vt, skc


def _sdpa_ref(q, k, v):
    s = (q.float() @ k.float().transpose(-1, -2)) * (128 ** -0.5)
    return torch.softmax(s, dim=-1) @ v.float()


def _attn_close(out, ref, what="attention", rel_tol=1e-2, abs_frac=0.25):
    """Stated tolerance of the bf16 attention kernels against the fp32 statement: rel-L2 <= 1e-2 AND max-abs <= 0.25 x the output rms
    (an absolute 2e-2 is blind to a lost key tile on long streams, where the output rms is ~ sqrt(e / sk):
    0.025 at 4200 keys, 0.006 at 65 552 - losing one of 66 tiles moves the output by ~0.003).  Measured: rel-L2 ~3e-3."""
    out, ref = out.float(), ref.float()
    rms = ref.pow(2).mean().sqrt().item()
    rel = ((out - ref).norm() / ref.norm()).item()
    mx = (out - ref).abs().max().item()
    assert rel <= rel_tol and mx <= abs_frac * rms, f"{what}: rel-L2 {rel:.3e} (tol {rel_tol}), max-abs {mx:.3e} vs {abs_frac} x rms {rms:.3e}"
    return rel


@pytest.mark.parametrize("nseq,H,sq,sk,nchunks", [(1, 2, 300, 300, 1), (2, 2, 196, 196, 1), (5, 2, 49, 9, 1),
                                                  (3, 2, 70, 17, 1), (1, 1, 1000, 64, 1), (2, 2, 196, 196, 2),
                                                  (1, 2, 520, 1040, 4), (2, 8, 2049, 257, 1),
                                                  # short last query block -> split-KV tail path (+ chunks)
                                                  (1, 2, 2320, 4200, 1), (2, 1, 2305, 4224, 2), (1, 1, 2432, 4097, 1)])
# 0/8: product dispatch; 60/68: forced 4 waves x 64 rows (68 = lazy re-base + exact fallback, 28 = exact deferred re-base);
# 90/98: forced 8-wave kernel; 58: 2 x 4-wave geometry; 78: balanced two-phase
@pytest.mark.parametrize("defer", [0, 8, 28, 60, 68, 90, 98, 58, 78])
def test_attention(dev, nseq, H, sq, sk, nchunks, defer):
    from actionmesh_amd import ops
    q = _randn((nseq, H, sq, 128), 1, dev).to(torch.bfloat16)
    k = _randn((nseq, H, sk, 128), 2, dev).to(torch.bfloat16)
    v = _randn((nseq, H, sk, 128), 3, dev).to(torch.bfloat16)
    Q, K, Vt, skc = _layout(q, k, v, nchunks)
    out = ops.attention(Q, K, Vt, sq, skc, nchunks=nchunks, defer_log2=defer)
    torch.cuda.synchronize()
    ref = _sdpa_ref(q, k, v).permute(0, 2, 1, 3).reshape(nseq * sq, H * 128)
    _attn_close(out, ref, f"attention defer={defer}")


@pytest.mark.parametrize("defer", [0, 8])
@pytest.mark.parametrize("nseq,H,sq,skc,P", [(2, 2, 2320, 1100, 4), (1, 2, 2304, 1024, 2), (1, 1, 2432, 1030, 3)])
def test_attention_two_pass_matches_one_pass(dev, nseq, H, sq, skc, P, defer):
    """Multi-GPU overlap path on one GPU: the full query blocks attend to the local key chunk first (state saved),
    then resume over the other chunks in ring order; the short last block runs one pass.  Every rank's view must
    reproduce the one-pass result (softmax is order-free over keys) and the fp32 reference."""
    from actionmesh_amd import ops
    q = _randn((nseq, H, sq, 128), 1, dev).to(torch.bfloat16)
    k = _randn((nseq, H, skc * P, 128), 2, dev).to(torch.bfloat16)
    v = _randn((nseq, H, skc * P, 128), 3, dev).to(torch.bfloat16)
    Q, K, Vt, skc_ = _layout(q, k, v, P)
    assert skc_ == skc
    ref = _sdpa_ref(q, k, v).permute(0, 2, 1, 3).reshape(nseq * sq, H * 128)
    one = ops.attention(Q, K, Vt, sq, skc, nchunks=P, defer_log2=defer).float()
    _attn_close(one, ref, "one pass")
    state = torch.full((nseq * H, Q.shape[2], ops.STATE_LD), float("nan"), device=dev)
    for r in range(P):
        out = torch.full((nseq * sq, H * 128), 768.0, dtype=torch.bfloat16, device=dev)
        ops.attention(Q, K, Vt, sq, skc, out=out, nchunks=1, defer_log2=defer, rows=1, state_mode=1, state=state,
                      chunk_first=r, chunk_total=P)
        assert (out.float() == 768.0).all(), "the first pass must not write the output"
        ops.attention(Q, K, Vt, sq, skc, out=out, nchunks=P - 1, defer_log2=defer, rows=1, state_mode=2, state=state,
                      chunk_first=(r + 1) % P, chunk_total=P)
        ops.attention(Q, K, Vt, sq, skc, out=out, nchunks=P, defer_log2=defer, rows=2)
        torch.cuda.synchronize()
        o = out.float()
        assert not (o == 768.0).any(), f"rank {r}: rows left unwritten"
        _attn_close(o, ref, f"rank {r} vs fp32 reference")
        _attn_close(o, one, f"rank {r} vs one pass")


def test_attention_forced_rescale_branch(dev):
    """A key that dominates late in the stream forces the online-softmax rescale branch with the
    deferred-rescale threshold active (cdna guide section 5.4 rule 26); threshold 0 and 8 must agree."""
    from actionmesh_amd import ops
    nseq, H, sq, sk = 1, 1, 256, 640
    q = _randn((nseq, H, sq, 128), 1, dev)
    k = _randn((nseq, H, sk, 128), 2, dev) * 0.3
    v = _randn((nseq, H, sk, 128), 3, dev)
    for (row, key, gain) in ((7, 333, 6.0), (100, 500, 12.0), (255, 639, 20.0), (31, 70, 9.0)):
        k[0, 0, key] = q[0, 0, row] * gain / q[0, 0, row].norm() * 11.3 / 3.0
    q, k, v = (t.to(torch.bfloat16) for t in (q, k, v))
    Q, K, Vt, skc = _layout(q, k, v, 1)
    ref = _sdpa_ref(q, k, v).permute(0, 2, 1, 3).reshape(sq, 128)
    for base in (0, 60, 90):          # product dispatch, forced 4x64 kernel, forced 8-wave kernel
        outs = []
        for defer in (base, base + 8):
            o = ops.attention(Q, K, Vt, sq, skc, defer_log2=defer).float()
            assert (o - ref).abs().max().item() < 3e-2, f"defer={defer}"
            outs.append(o)
        assert (outs[0] - outs[1]).abs().max().item() < 3e-2


def _plant(q, k, row, key, score):
    """Make key `key` score ~`score` (natural-log units, after the 1/sqrt(128) scale) against query row `row` of head 0."""
    k[0, 0, key] = q[0, 0, row] / q[0, 0, row].norm() ** 2 * score * 128 ** 0.5


def test_attention_lazy_rebase_and_exact_fallback(dev):
    """The long-key-stream kernel keeps no running row max in its loop: a tile's row sums tell afterwards that m_run
    has fallen behind, and O / l / the pending P are multiplied by an exact power of two (jumps up to 2^60); a bigger
    single-tile jump marks the workgroup and the exact kernel launched behind recomputes it.  Both paths must agree with
    the fp32 softmax, the fallback must fire only when needed, and its marks must be cleared for the next launch."""
    from actionmesh_amd import ops
    nseq, H, sq, sk = 1, 2, 512, 2048           # 32 key tiles -> the 4x64 kernel under the product dispatch
    q = _randn((nseq, H, sq, 128), 1, dev)
    v = _randn((nseq, H, sk, 128), 3, dev)

    def run(k, defer=8):
        qb, kb, vb = (t.to(torch.bfloat16) for t in (q, k, v))
        Q, K, Vt, skc = _layout(qb, kb, vb, 1)
        out = ops.attention(Q, K, Vt, sq, skc, defer_log2=defer).float()
        ref = _sdpa_ref(qb, kb, vb).permute(0, 2, 1, 3).reshape(nseq * sq, H * 128)
        return (out - ref).abs().max().item()

    base = ops.attention_fallback_count()
    # (a) moderate jumps, several per row, in both 32-row blocks of a wave and in both half-lanes' key halves
    k = _randn((nseq, H, sk, 128), 2, dev) * 0.3
    for (row, key, score) in ((7, 300, 12.0), (7, 900, 25.0), (7, 1990, 38.0), (40, 70, 9.0), (40, 100, 20.0),
                              (100, 1500, 30.0), (300, 2047, 35.0), (511, 64, 15.0), (470, 1000, 22.0)):
        _plant(q, k, row, key, score)
    assert run(k) < 3e-2
    assert ops.attention_fallback_count() == base, "jumps below 2^60 are the lazy re-base's job"
    # (b) a single-tile jump of ~2^108 (75 nats): only the exact kernel can represent it
    k2 = _randn((nseq, H, sk, 128), 2, dev) * 0.3
    _plant(q, k2, 100, 1500, 75.0)
    assert run(k2) < 3e-2
    n1 = ops.attention_fallback_count()
    assert n1 == base + 1, f"one workgroup (rows 0..255 of head 0) had to be recomputed, got {n1 - base}"
    # (c) marks are cleared: an ordinary launch afterwards recomputes nothing, and the exact variants never mark
    assert run(_randn((nseq, H, sk, 128), 4, dev)) < 2e-2
    assert run(k2, defer=0) < 3e-2 and run(k2, defer=28) < 3e-2
    assert ops.attention_fallback_count() == n1
    # (d) two-pass form: the jump sits in the remote chunk of rank 0 (resume pass) and in the local chunk of rank 1
    qb, kb, vb = (t.to(torch.bfloat16) for t in (q, k2, v))
    Q, K, Vt, skc = _layout(qb, kb, vb, 2)
    ref = _sdpa_ref(qb, kb, vb).permute(0, 2, 1, 3).reshape(nseq * sq, H * 128)
    state = torch.zeros((nseq * H, Q.shape[2], ops.STATE_LD), device=dev)
    for r in range(2):
        out = torch.zeros((nseq * sq, H * 128), dtype=torch.bfloat16, device=dev)
        ops.attention(Q, K, Vt, sq, skc, out=out, nchunks=1, rows=1, state_mode=1, state=state, chunk_first=r, chunk_total=2)
        ops.attention(Q, K, Vt, sq, skc, out=out, nchunks=1, rows=1, state_mode=2, state=state, chunk_first=(r + 1) % 2,
                      chunk_total=2)
        ops.attention(Q, K, Vt, sq, skc, out=out, nchunks=2, rows=2)
        assert (out.float() - ref).abs().max().item() < 3e-2, f"rank {r}"
    assert ops.attention_fallback_count() == n1 + 2
    # (e) un-normalised operands (Stage II has no qk-norm): scores tens of octaves away from 0 everywhere.  m_run starts at
    #     tile 0's row max, so this is ordinary work for the lazy kernel, not a job for the fallback
    qs = q * 4.0
    ks_ = _randn((nseq, H, sk, 128), 5, dev) * 4.0
    qb, kb, vb = (t.to(torch.bfloat16) for t in (qs, ks_, v))
    Q, K, Vt, skc = _layout(qb, kb, vb, 1)
    smax = ((qb.float() @ kb.float().transpose(-1, -2)) * (128 ** -0.5) * 1.4427).amax(-1)
    assert float(smax.min()) > 25.0 and float(smax.max()) > 60.0, (float(smax.min()), float(smax.max()))   # log2 units
    out = ops.attention(Q, K, Vt, sq, skc).float()
    exact = ops.attention(Q, K, Vt, sq, skc, defer_log2=28).float()
    ref = _sdpa_ref(qb, kb, vb).permute(0, 2, 1, 3).reshape(nseq * sq, H * 128)
    # scores of +-80 octaves amplify the bf16 rounding of the pre-scaled Q fragments (both kernel forms share it): the two
    # forms must agree tightly, the fp32 softmax of the unscaled bf16 operands only loosely
    assert (out - exact).abs().max().item() < 4e-2          # one bf16 ulp at |o| in [4, 8)
    assert (out - ref).abs().max().item() < 0.3 and (exact - ref).abs().max().item() < 0.3
    # of the 4 workgroups at most one meets a row whose later tiles top tile 0's max by more than 60 octaves (scores have a
    # standard deviation of 23 octaves here); from m_run = 0 all four would have gone to the fallback
    assert ops.attention_fallback_count() <= n1 + 2 + 1


@pytest.mark.parametrize("defer", [8, 28, 98])
def test_attention_bitwise_repeatable(dev, defer):
    """Timing-dependent faults (an in-flight MFMA result read early, a DMA piece not yet landed, a register the compiler
    parked where the asm keeps its own data) show up as run-to-run differences long before they show up as large errors:
    24 launches of the same problem - more workgroups than CUs, other kernels in between - must agree bit for bit."""
    from actionmesh_amd import ops
    nseq, H, sq, sk = 2, 8, 4352, 4224
    q = _randn((nseq, H, sq, 128), 1, dev).to(torch.bfloat16)
    k = _randn((nseq, H, sk, 128), 2, dev).to(torch.bfloat16)
    v = _randn((nseq, H, sk, 128), 3, dev).to(torch.bfloat16)
    Q, K, Vt, skc = _layout(q, k, v, 2)
    first = ops.attention(Q, K, Vt, sq, skc, nchunks=2, defer_log2=defer).clone()
    filler_a = _randn((4096, 1024), 4, dev).to(torch.bfloat16)
    filler_w = _randn((1024, 1024), 5, dev, 0.03).to(torch.bfloat16)
    for i in range(23):
        if i % 3 == 0:
            ops.gemm(filler_a, filler_w)               # perturb the clock / cache state between launches
        out = ops.attention(Q, K, Vt, sq, skc, nchunks=2, defer_log2=defer)
        assert torch.equal(out, first), f"launch {i + 2} differs from the first"


def test_attention_properties_full_size(dev):
    """BASELINE headline sequence (16 frames x 4097 tokens), one head: size-independent properties.
    (a) V == 1 -> output == 1 (softmax rows sum to 1), (b) permuting key chunks leaves the output
    unchanged, (c) sampled query rows equal the fp32 reference on the full 65 552 keys."""
    from actionmesh_amd import ops
    T, L = 16, 4097
    S = T * L
    q = _randn((1, 1, S, 128), 1, dev).to(torch.bfloat16)
    k = _randn((1, 1, S, 128), 2, dev).to(torch.bfloat16)
    v = _randn((1, 1, S, 128), 3, dev).to(torch.bfloat16)
    Q, K, Vt, skc = _layout(q, k, v, 4)
    out = ops.attention(Q, K, Vt, S, skc, nchunks=4).float()
    rows = torch.tensor([0, 1, 31, 32, 255, 256, 4096, 4097, 40000, S - 17, S - 1], device=dev)
    ref = _sdpa_ref(q[:, :, rows], k, v)[0, 0]
    _attn_close(out[rows], ref, "sampled rows at 65 552 keys")
    Kp = K[[2, 0, 3, 1]].contiguous(); Vp = Vt[[2, 0, 3, 1]].contiguous()
    outp = ops.attention(Q, Kp, Vp, S, skc, nchunks=4).float()
    _attn_close(outp, out, "permuted key chunks")
    ones = torch.ones_like(Vt)
    o1 = ops.attention(Q, K, ones, S, skc, nchunks=4).float()
    assert (o1 - 1.0).abs().max().item() < 8e-3


def _coverage_case(dev, sq, skc, P, H=1, score_by_tile=False):
    """Operands that make every 64-key tile of the stream visible in the output: V = indicator of the key's tile (channel = global
    tile index mod 128), and either Q = 0 (uniform scores: channel c of every output row = the share of the keys that sit in tiles
    = c mod 128 - EXACT in bf16 / fp8 arithmetic: p = 1, V = 1, integer sums) or scores that depend on the key's tile only.
    A dropped, duplicated or mis-weighted tile moves its channel by >= 1/9 of its value at 1025 tiles."""
    tiles_c = (skc + 63) // 64
    key = torch.arange(skc, device=dev)
    q = torch.zeros((1, H, sq, 128), device=dev)
    k = torch.zeros((1, H, skc * P, 128), device=dev)
    v = torch.zeros((1, H, skc * P, 128), device=dev)
    gt = torch.cat([c * tiles_c + key // 64 for c in range(P)])               # global tile of every key
    v[0, :, torch.arange(skc * P, device=dev), gt % 128] = 1.0
    s_tile = torch.zeros(P * tiles_c, dtype=torch.float64, device=dev)
    if score_by_tile:
        q[..., 0] = 8.0
        b = ((torch.arange(P * tiles_c, device=dev) % 7) - 3).double() * 0.5
        k[0, :, :, 0] = b[gt].float()
        s_tile = 8.0 * b * 128 ** -0.5
    w = torch.exp(s_tile - s_tile.max())[gt]                                   # fp64 softmax weight of every key (same for all rows)
    expect = torch.zeros(128, dtype=torch.float64, device=dev).index_add_(0, gt % 128, w) / w.sum()
    return q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16), expect


@pytest.mark.parametrize("form", ["one_pass", "two_pass", "forced_8wave"])
@pytest.mark.parametrize("sq,skc,P", [(300, 2100, 1), (2320, 1100, 3), (700, 16388, 4)])
def test_attention_key_coverage(dev, sq, skc, P, form):
    """Every key tile is counted exactly once, with its own weight: 33 tiles; 18 x 3 chunk tiles with a
    partial last tile per chunk and a split tail; 65 552 keys in 4 chunks (1028 tiles)."""
    from actionmesh_amd import ops
    if form == "two_pass" and P == 1:
        pytest.skip("two-pass needs more than one chunk")
    for by_tile in (False, True):
        q, k, v, expect = _coverage_case(dev, sq, skc, P, H=1, score_by_tile=by_tile)
        Q, K, Vt, skc_ = _layout(q, k, v, P)
        if form == "two_pass":
            outs = []
            state = torch.zeros((1, Q.shape[2], ops.STATE_LD), device=dev)
            for r in range(P):
                out = torch.zeros((sq, 128), dtype=torch.bfloat16, device=dev)
                ops.attention(Q, K, Vt, sq, skc, out=out, nchunks=1, rows=1, state_mode=1, state=state, chunk_first=r, chunk_total=P)
                ops.attention(Q, K, Vt, sq, skc, out=out, nchunks=P - 1, rows=1, state_mode=2, state=state, chunk_first=(r + 1) % P,
                              chunk_total=P)
                ops.attention(Q, K, Vt, sq, skc, out=out, nchunks=P, rows=2)
                outs.append(out)
        else:
            outs = [ops.attention(Q, K, Vt, sq, skc, nchunks=P, defer_log2=98 if form == "forced_8wave" else 8)]
        for out in outs:
            o = out.double()
            if not by_tile:          # exact arithmetic: one rounding of count / sk to bf16 (computed as O * (1 / l): <= 1 bf16 ulp)
                err = ((o - expect[None]).abs() / expect[None].clamp_min(1e-30)).max().item()
                assert err <= 2.0 ** -7, f"{form} uniform scores: a key tile is mis-counted (max relative error {err:.3e})"
            else:                    # bf16 probabilities: <= 2^-9 relative each, systematic within a tile
                nz = expect > 0
                err = ((o[:, nz] - expect[None, nz]).abs() / expect[None, nz]).max().item()
                assert err <= 1.5e-2, f"{form} tile-dependent scores: max relative error {err:.3e}"


# ------------------------------------------------------------------------------------------
def test_timestep_sinusoid(dev):
    from actionmesh_amd import ops
    t = torch.tensor([1000.0, 0.0, 8.9285717, 523.25, 964.40027], device=dev)
    out = ops.timestep_sinusoid(t, 256).float()
    half = 128
    f = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=dev) / half)
    ref = torch.cat([torch.sin(t[:, None] * f), torch.cos(t[:, None] * f)], -1)
    assert (out - ref).abs().max().item() < 6e-3    # bf16 output + fp32 range reduction at |arg| <= 1000


def test_flow_step(dev):
    from actionmesh_amd import ops
    T, N, D = 4, 48, 64
    v = _randn((2, T, N, D), 1, dev).to(torch.bfloat16)
    lat = _randn((T, N, D), 2, dev)
    lat0 = lat.clone()
    ops.flow_step(v, lat, [7.5], 0.0356, True, [False, True, True, False])
    v0, v1 = v[0].float(), v[1].float()
    agg = rb(v0 + rb(7.5 * rb(v1 - v0)))
    ref = lat0 + rb(0.0356 * agg)
    assert torch.equal(lat[0], lat0[0]) and torch.equal(lat[3], lat0[3])
    assert torch.allclose(lat[1:3], ref[1:3], rtol=0, atol=1e-6)
    # 3-branch guidance + subtractive flow, all frames
    v3 = _randn((3, T, N, D), 3, dev).to(torch.bfloat16)
    lat = lat0.clone()
    ops.flow_step(v3, lat, [2.0, 3.0], 0.1, False, None)
    a, b, c = (v3[i].float() for i in range(3))
    agg = rb(rb(a + rb(2.0 * rb(b - a))) + rb(3.0 * rb(c - b)))
    assert torch.allclose(lat, lat0 - rb(0.1 * agg), rtol=0, atol=1e-6)


def test_f32_to_bf16(dev):
    from actionmesh_amd import ops
    x = _randn((1000003,), 1, dev) * 100
    assert torch.equal(ops.f32_to_bf16(x), x.to(torch.bfloat16))



@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C", [1024, 320, 2048])
def test_add_layernorm_f32(dtype, C):
    """am_add_layernorm_f32 (round 6: the fp32 residual stream of the reference's Stage II / DINOv2 encoder, temporal_autoencoder.py:258):
    h32 += y16 in place, z16 = LayerNorm(h32) rounded to 16 bits - against torch in fp32; the accumulate-only and LayerNorm-only forms."""
    from actionmesh_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(C)
    rows = 777
    h = torch.randn((rows, C), device=dev, generator=g) * 3 + 2
    y = torch.randn((rows, C), device=dev, generator=g).to(dtype)
    w = torch.rand(C, device=dev, generator=g) + 0.5
    b = torch.randn(C, device=dev, generator=g) * 0.3
    h_ref = h + y.float()
    z_ref = torch.nn.functional.layer_norm(h_ref, (C,), w, b, 1e-5)
    h1 = h.clone()
    z = ops.add_layernorm_f32(h1, y, w, b)
    assert torch.equal(h1, h_ref)                                                   # one fp32 add per element: exact
    assert z.dtype == dtype and float((z.float() - z_ref).abs().max()) <= 2.0 ** (-7 if dtype == torch.bfloat16 else -10) * float(z_ref.abs().max())
    assert float((z.float() - z_ref).norm() / z_ref.norm()) < (2.5e-3 if dtype == torch.bfloat16 else 3.5e-4)
    h2 = h.clone()
    assert ops.add_layernorm_f32(h2, y) is None and torch.equal(h2, h_ref)          # accumulate only
    z3 = ops.add_layernorm_f32(h_ref.clone(), None, w, b, dtype=dtype)              # LayerNorm only: the same bits
    assert torch.equal(z3, z)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("defer", [0, 8])
@pytest.mark.parametrize("nseq,H,sq,sk", [(16, 8, 1100, 257),      # the cross-attention's context: 4 full tiles + ONE key
                                          (32, 4, 1024, 256),      # no tail at all
                                          (16, 8, 1500, 65),       # one tile + one key
                                          (16, 8, 1030, 144),      # two tiles + the longest short tail (16 keys)
                                          (64, 2, 2049, 257)])
def test_cross_attention_with_the_resident_key_stream(dev, nseq, H, sq, sk, defer, dtype):
    """Round 6: launches of the cross-attention family (one key chunk of <= 4 full tiles + a tail of <= 16 keys, >= 4 query blocks per
    (sequence, head), >= 128 pairs) run attn_resident_kernel - the key stream loaded into LDS once per (sequence, head), every query
    block walked against it without a barrier, the tail as 8 + 4 MFMAs.  Against the fp32 statement (the attention kernels' own
    tolerance), and against the tile-streaming kernel forced through its geometry code (defer 50 / 58: never the resident path): the two
    differ only in the tail's padded-key correction, i.e. by rounding."""
    from actionmesh_amd import ops
    q = (_randn((nseq, H, sq, 128), 11, dev) * 1.3).to(dtype)
    k = _randn((nseq, H, sk, 128), 12, dev).to(dtype)
    v = _randn((nseq, H, sk, 128), 13, dev).to(dtype)
    sq_pad, sk_pad = ops.round_up(sq, 256), ops.round_up(sk, 64)
    Q = torch.zeros((nseq, H, sq_pad, 128), dtype=dtype, device=dev); Q[:, :, :sq] = q
    K = torch.zeros((1, nseq, H, sk_pad, 128), dtype=dtype, device=dev); K[0, :, :, :sk] = k
    vp = torch.zeros((nseq, H, sk_pad, 128), dtype=dtype, device=dev); vp[:, :, :sk] = v
    Vt = vp[:, :, ops.perm16_index(sk_pad, dev)].transpose(-1, -2).contiguous()[None]
    out = ops.attention(Q, K, Vt, sq, sk, defer_log2=defer)
    streamed = ops.attention(Q, K, Vt, sq, sk, defer_log2=50 + defer)
    torch.cuda.synchronize()
    ref = _sdpa_ref(q, k, v).permute(0, 2, 1, 3).reshape(nseq * sq, H * 128)
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-3
    _attn_close(out, ref, f"resident cross-attention defer={defer}", rel_tol=tol)
    d = ((out.float() - streamed.float()).norm() / streamed.float().norm()).item()
    assert d <= (3e-3 if dtype == torch.bfloat16 else 4e-4), d
    assert torch.equal(out, ops.attention(Q, K, Vt, sq, sk, defer_log2=defer)), "same inputs, same bits"
