"""BASELINE configs[4] AT ITS OWN LENGTH (VERDICT r04 weak #1 / next #1a): 64 frames x 8192 tokens x width 1024 = 524 352-token inflated
sequences (8193 key tiles of 64), B = 2 guidance branches x 8 heads: Q, K, V^T and the output are 2.1 GB EACH, i.e. every operand
crosses 2^31 bytes inside one launch.  `bench.py --shape long64` only asserts finiteness; these tests put numbers behind it:

  * test_attention_long64_sampled_rows   one launch at the exact configs[4] attention shape (bf16, fp8, fp8_fast); rows sampled from
                                         both sequences, first / middle / last heads, first / last query blocks, against the fp32
                                         statement softmax(q k^T / sqrt(128)) v over all 524 352 keys;
  * test_attention_long64_key_coverage   every one of the 8193 tiles counted exactly once with its own weight (V = indicator of the
                                         key's tile; the 8 heads take turns so a channel holds <= 9 tiles and a lost tile moves it by
                                         >= 1/9), as ONE chunk and as the 4 / 8 frame-shard chunks of the multi-GPU walk (one pass over
                                         all chunks, and the two-pass save / resume form from every rank's starting chunk);
  * test_one_layer_model_at_long64       the whole forward (prologue, fused QKV + qk-norm + RoPE over 64 frame positions, the
                                         524 352-key self-attention, cross-attention, MLP, epilogue) of a ONE-layer model of the
                                         configs[4] width at the configs[4] shape, through HipDenoiser, against the oracle evaluated
                                         on sampled rows (oracle.denoiser_forward_rows: everything but the self-attention is per
                                         token, and a row's attention needs only its own Q - seconds of host time).
Stated tolerances are in the tests."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

T64, N64, H64, NSEQ = 64, 8192, 8, 2
L64 = N64 + 1
S64 = T64 * L64                       # 524 352 = 64 x 8193


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from actionmesh_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def long_operands(dev):
    """Unit-variance bf16 q, k, v at the configs[4] attention shape, directly in the kernel layouts (am_head_post's)."""
    from actionmesh_amd import ops
    g = torch.Generator(device=dev).manual_seed(64)
    sq_pad, sk_pad = ops.round_up(S64, 256), ops.round_up(S64, 64)
    assert sk_pad == S64
    Q = torch.zeros((NSEQ, H64, sq_pad, 128), dtype=torch.bfloat16, device=dev)
    K = torch.empty((1, NSEQ, H64, sk_pad, 128), dtype=torch.bfloat16, device=dev)
    v = torch.empty((NSEQ, H64, S64, 128), dtype=torch.bfloat16, device=dev)
    for s in range(NSEQ):                                  # per sequence: keeps the fp32 temporaries of randn at 0.5 GB
        Q[s, :, :S64] = torch.randn((H64, S64, 128), generator=g, device=dev).to(torch.bfloat16)
        K[0, s] = torch.randn((H64, S64, 128), generator=g, device=dev).to(torch.bfloat16)
        v[s] = torch.randn((H64, S64, 128), generator=g, device=dev).to(torch.bfloat16)
    idx = ops.perm16_index(sk_pad, dev)
    Vt = torch.empty((1, NSEQ, H64, 128, sk_pad), dtype=torch.bfloat16, device=dev)
    for s in range(NSEQ):
        Vt[0, s] = v[s][:, idx].transpose(-1, -2)
    assert Q.numel() * 2 > 2 ** 31 and K.numel() * 2 > 2 ** 31 and Vt.numel() * 2 > 2 ** 31
    return Q, K, Vt, v


SAMPLED = [(s, h, r) for s in (0, 1) for h in (0, 3, 7)
           for r in (0, 255, 256, 17 * L64 + 5, S64 // 2 + 1, S64 - 257, S64 - 64, S64 - 17, S64 - 1)]


@pytest.mark.parametrize("form", ["bf16", "fp8", "fp8_fast"])
def test_attention_long64_sampled_rows(dev, long_operands, form):
    """Tolerance: the kernels' own statements (tests/test_kernels_gpu.py::_attn_close: rel-L2 <= 1e-2 and max-abs <= 0.25 x the output
    rms for bf16; tests/test_attention_fp8.py: rel-L2 <= 6e-2 / 6.5e-2 for fp8 / fp8_fast) on the sampled rows, whose reference is the
    fp32 softmax over ALL keys.  Rows of the second sequence and the last heads sit beyond 2^31 bytes in Q, K, V^T and O."""
    from actionmesh_amd import ops
    Q, K, Vt, v = long_operands
    if form == "bf16":
        out = ops.attention(Q, K, Vt, S64, S64)
    else:
        out = ops.attention_fp8(Q, K, Vt, S64, S64, ablate=400 if form == "fp8_fast" else 0)
        ops.attention_fp8.last_quantized = None            # 3 GB of quantised operands: do not keep them alive
    torch.cuda.synchronize()
    assert out.shape == (NSEQ * S64, H64 * 128) and out.numel() * 2 > 2 ** 31
    got, want = [], []
    for s, h, r in SAMPLED:
        q = Q[s, h, r].float()
        sc = (K[0, s, h].float() @ q) * 128 ** -0.5                      # (S64,)
        p = torch.softmax(sc.double(), dim=0)
        want.append((p[None] @ v[s, h].double())[0])
        got.append(out[s * S64 + r, h * 128:(h + 1) * 128].double())
    got, want = torch.stack(got), torch.stack(want)
    assert bool(torch.isfinite(got).all())
    r_all = _rel(got, want)
    worst = max(_rel(g_, w_) for g_, w_ in zip(got, want))
    mx = float((got - want).abs().max() / want.pow(2).mean().sqrt())
    print(f"long64 attention {form}: sampled rows rel-L2 {r_all:.3e}, worst single row {worst:.3e}, max-abs / rms {mx:.3e}")
    tol = {"bf16": 1e-2, "fp8": 6e-2, "fp8_fast": 6.5e-2}[form]
    assert r_all <= tol and worst <= 2 * tol, f"{form}: rel-L2 {r_all:.3e} / worst row {worst:.3e} (tol {tol})"
    if form == "bf16":
        assert mx <= 0.25
    # rows the kernel must not have touched / must have written: spot-check the head-major output stride
    assert bool(torch.isfinite(out[::65537].float()).all())


# ---------------------------------------------------------------------------------------------------------------------------------
def _coverage_long(dev, sq, P, by_tile):
    """tests/test_kernels_gpu.py::_coverage_case at the configs[4] key count, both sequences, 8 heads: head h of sequence s owns the
    tiles whose phase (global tile // 1025, rotated by s) is h - its V is the indicator (channel = global tile mod 128) of THOSE
    tiles and zero elsewhere, so a channel sums <= 9 tiles.  Q = 0 (uniform scores; exact integer arithmetic in bf16 and e4m3) or
    scores that depend on the key's tile only.  Returns the kernel-layout operands and the fp64 expectation (NSEQ, H, 128)."""
    from actionmesh_amd import ops
    assert S64 % P == 0
    skc = S64 // P
    tiles_c = (skc + 63) // 64
    sk_pad = tiles_c * 64
    key = torch.arange(skc, device=dev)
    gt = torch.cat([c * tiles_c + key // 64 for c in range(P)])               # global tile of every key, chunk-major
    n_tiles = P * tiles_c
    per = (n_tiles + H64 - 1) // H64
    sq_pad = ops.round_up(sq, 256)
    Q = torch.zeros((NSEQ, H64, sq_pad, 128), dtype=torch.bfloat16, device=dev)
    K = torch.zeros((P, NSEQ, H64, sk_pad, 128), dtype=torch.bfloat16, device=dev)
    Vt = torch.zeros((P, NSEQ, H64, 128, sk_pad), dtype=torch.bfloat16, device=dev)
    s_tile = torch.zeros(n_tiles, dtype=torch.float64, device=dev)
    if by_tile:
        Q[:, :, :sq, 0] = 8.0
        b = ((torch.arange(n_tiles, device=dev) % 7) - 3).double() * 0.5
        s_tile = 8.0 * b * 128 ** -0.5
        for c in range(P):
            K[c, :, :, :skc, 0] = b[gt[c * skc:(c + 1) * skc]].to(torch.bfloat16)
    w = torch.exp(s_tile - s_tile.max())[gt]                                   # fp64 softmax weight of every key
    pos_of_key = torch.empty(sk_pad, dtype=torch.long, device=dev)
    pos_of_key[ops.perm16_index(sk_pad, dev)] = torch.arange(sk_pad, device=dev)      # V^T position that holds key k
    expect = torch.zeros((NSEQ, H64, 128), dtype=torch.float64, device=dev)
    for s in range(NSEQ):
        phase = (gt // per + s) % H64
        for h in range(H64):
            mine = (phase == h).nonzero()[:, 0]                                # indices into the chunk-major key list
            c_of, k_of = mine // skc, mine % skc
            Vt[c_of, s, h, gt[mine] % 128, pos_of_key[k_of]] = 1.0
            expect[s, h].index_add_(0, gt[mine] % 128, w[mine])
    expect /= w.sum()
    assert K.numel() * 2 > 2 ** 31
    return Q, K, Vt, skc, expect


def _check_coverage(out, expect, sq, what, exact):
    o = out.double().view(NSEQ, sq, H64, 128).permute(0, 2, 1, 3)              # (NSEQ, H, sq, 128)
    e = expect[:, :, None, :]
    nz = (e > 0).expand_as(o)
    if bool((~nz).any()):
        stray = float(o[~nz].abs().max())
        assert (stray == 0.0) if exact else (stray < 1e-6), f"{what}: weight in a channel no key owns"
    err = float(((o - e).abs() / e.clamp_min(1e-300))[nz].max())
    tol = 2.0 ** -7 if exact else 1.5e-2
    assert err <= tol, f"{what}: a key tile is mis-counted (max relative error {err:.3e}, a lost tile is >= 1/9)"
    return err


@pytest.mark.parametrize("form", ["bf16", "fp8", "fp8_fast"])
@pytest.mark.parametrize("P", [1, 4, 8])
def test_attention_long64_key_coverage(dev, P, form):
    """524 352 keys in P chunks (P = 4: 131 088 keys per chunk, P = 8: 65 544 - partial last tiles, the frame shards of a 4 / 8 GPU
    run); sq = 2368 rows = 9 full blocks + the 64-row last block of the real shape.  One pass over all chunks, and for P > 1 the
    two-pass form (local chunk first, state saved, the other chunks resumed in ring order) from ranks 0 and P - 1.
    Uniform scores are EXACT (integer sums, one final rounding: <= 2^-7); tile-dependent scores (bf16 only) <= 1.5e-2."""
    from actionmesh_amd import ops
    sq = 2368
    abl = 400 if form == "fp8_fast" else 0

    def attn(Q, K, Vt, skc, **kw):
        if form == "bf16":
            return ops.attention(Q, K, Vt, sq, skc, **kw)
        return ops.attention_fp8(Q, K, Vt, sq, skc, ablate=abl, **kw)

    for by_tile in ((False, True) if form == "bf16" else (False,)):
        Q, K, Vt, skc, expect = _coverage_long(dev, sq, P, by_tile)
        out = attn(Q, K, Vt, skc, nchunks=P)
        e1 = _check_coverage(out, expect, sq, f"{form} P={P} one pass by_tile={by_tile}", not by_tile)
        print(f"long64 coverage {form} P={P} by_tile={by_tile}: one pass max relative error {e1:.3e}")
        if P > 1:
            quant = ops.attention_fp8.last_quantized if form != "bf16" else None
            kw = dict(quantized=quant) if quant is not None else {}
            state = torch.zeros((NSEQ * H64, Q.shape[2], ops.STATE_LD), device=dev)
            for r in (0, P - 1):
                o2 = torch.zeros((NSEQ * sq, H64 * 128), dtype=torch.bfloat16, device=dev)
                attn(Q, K, Vt, skc, out=o2, nchunks=1, rows=1, state_mode=1, state=state, chunk_first=r, chunk_total=P, **kw)
                attn(Q, K, Vt, skc, out=o2, nchunks=P - 1, rows=1, state_mode=2, state=state, chunk_first=(r + 1) % P, chunk_total=P, **kw)
                attn(Q, K, Vt, skc, out=o2, nchunks=P, rows=2, **kw)
                e2 = _check_coverage(o2, expect, sq, f"{form} P={P} two-pass from rank {r} by_tile={by_tile}", not by_tile)
                print(f"long64 coverage {form} P={P} by_tile={by_tile}: two-pass from rank {r} max relative error {e2:.3e}")
        if form != "bf16":
            ops.attention_fp8.last_quantized = None
        del Q, K, Vt, out
        torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------------------------------------
QK_GAIN = 1.8        # qk-norm gains x 1.8: scores ~ N(0, 3.2^2) - at 524 352 keys unit-variance scores average ~everything
                     # (attention output rms ~2e-3 of V: invisible in the velocity); with this gain the softmax of a row is carried by
                     # a few hundred keys anywhere in the stream and the self-attention moves the residual stream by ~20 %

_ORACLE_ROWS = {}


def _long64_case():
    from oracle import denoiser_oracle as O
    hp = dict(in_channels=64, num_layers=1, num_attention_heads=H64, width=1024, mlp_ratio=4.0, cross_attention_dim=1024,
              inflated_layers=(0,))
    cfg = O.OracleConfig(**hp)
    sd = O.synthetic_state_dict(cfg, seed=64)
    for k in ("blocks.0.s_attn.norm_q.weight", "blocks.0.s_attn.norm_k.weight"):
        sd[k] = sd[k] * QK_GAIN
    g = torch.Generator().manual_seed(640)
    x = torch.randn((1, T64, N64, 64), generator=g)
    ctx = torch.randn((1, T64, 257, 1024), generator=g)
    mask = torch.zeros((1, T64)); mask[0, 0] = 1.0
    fs = torch.arange(T64, dtype=torch.float32)[None]
    rows = torch.stack([torch.randint(0, 2, (48,), generator=g), torch.randint(0, T64, (48,), generator=g),
                        torch.randint(0, N64, (48,), generator=g)], dim=1)
    rows[0] = torch.tensor([1, T64 - 1, N64 - 1]); rows[1] = torch.tensor([0, 0, 0]); rows[2] = torch.tensor([1, 0, 0])
    rows[3] = torch.tensor([0, T64 - 1, N64 - 1])
    return hp, cfg, sd, x, ctx, mask, fs, rows


@pytest.mark.parametrize("dtype", ["bf16", "fp8", "fp8_fast"])
def test_one_layer_model_at_long64(dev, dtype):
    """Stated tolerance on the velocity of the 48 sampled tokens vs the fp32 oracle: bf16 rel-L2 <= 1e-2 (one layer of bf16 rounding:
    the 21-layer forward is 9.8e-3); fp8 / fp8_fast <= 2e-2 - the e4m3 noise of the self-attention (<= 6e-2 of its output at unit
    scores, more at these 3.2-sigma scores) on a branch that carries ~20 % of the residual stream; worst single token <= 3 x.
    MEASURED on MI355X (round 5): bf16 5.2e-3 (worst token 6.4e-3), fp8 8.1e-3 (1.6e-2), fp8_fast 1.2e-2 (3.4e-2)."""
    from actionmesh_amd import ClassifierFreeGuidance, HipDenoiser
    from oracle import denoiser_oracle as O
    hp, cfg, sd, x, ctx, mask, fs, rows = _long64_case()
    cfgd = ClassifierFreeGuidance(True, [[0, 1], [1, 1]], [7.5])
    x_in, c_in, m_in, f_in = cfgd.cfg_at_inference(x, ctx, mask, fs)
    tt = torch.tensor([523.25]).expand(2)
    if "ref" not in _ORACLE_ROWS:
        from conftest import host_threads
        host_threads()
        _ORACLE_ROWS["ref"] = O.denoiser_forward_rows(sd, cfg, x_in, c_in, f_in, tt, m_in, rows, frame_chunk=4)
    ref = _ORACLE_ROWS["ref"]
    model = HipDenoiser(num_tokens_nominal=N64, temporal_context_size=T64, attn_dtype=dtype, **hp)
    model.load_state_dict(sd)
    model.to(dev).eval()
    v, _ = model.forward(x_in.to(dev), c_in.to(dev), f_in.to(dev), tt.to(dev), m_in.to(dev), None)
    torch.cuda.synchronize()
    n8, n16 = model._engine.attention_counters()
    assert (n8 > 0 and n16 == 0) if dtype.startswith("fp8") else (n16 > 0 and n8 == 0), (n8, n16)
    assert v.shape == (2, T64, N64, 64) and bool(torch.isfinite(v.float()).all())
    got = torch.stack([v[b, t, n].float().cpu() for b, t, n in rows.tolist()])
    r = _rel(got, ref)
    worst = max(_rel(g_, w_) for g_, w_ in zip(got, ref))
    print(f"one-layer model at 64 x 8192 tokens, {dtype}: sampled-row velocity rel-L2 vs the fp32 oracle {r:.3e} (worst row {worst:.3e})")
    tol = 1e-2 if dtype == "bf16" else 2e-2
    assert r <= tol and worst <= 3 * tol, f"{dtype}: {r:.3e} / worst row {worst:.3e} (tol {tol})"
    model._engine.close()
