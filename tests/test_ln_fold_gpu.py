"""LayerNorm folded into the linear that consumes it (block.py:138,146,152; VERDICT r03 missing #3 / next #8).

  consumer:  gemm(x, W', bias=d, ln=(stats, colsum)) = rstd (x W'^T - mean colsum) + d   vs   gemm(layernorm(x), W, bias)
  producer:  gemm(..., ln_part=) writes (mean, M2) of every 256-column slice of its output rows; row_stats_finalize merges them
  stand-alone statistics: row_stats(x), layernorm(..., stats_out=)

The un-folded pair rounds the normalised activation to bf16 before the linear; the folded form does not, so the two agree to the
bf16 noise of that rounding and the folded one is the closer of the two to the fp32 result (asserted).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from actionmesh_amd import ops  # noqa: E402

DEV = "cuda:0"


def rnd(*shape, seed=0, dtype=torch.bfloat16, scale=1.0, shift=0.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(shape, device=DEV, generator=g) * scale + shift).to(dtype)


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def ln_ref(x, gamma, beta, eps=1e-5):
    return torch.nn.functional.layer_norm(x.float(), (x.shape[-1],), gamma, beta, eps)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C", [1024, 320, 2048])
def test_row_stats_match_torch(dtype, C):
    x = rnd(777, C, seed=1, dtype=dtype, scale=3.0, shift=5.0)          # rows far from zero: nothing may cancel
    st = ops.row_stats(x)
    xf = x.float()
    mean = xf.mean(-1)
    rstd = (xf.var(-1, unbiased=False) + 1e-5).rsqrt()
    assert torch.allclose(st[:, 0], mean, rtol=2e-6, atol=2e-6)
    assert torch.allclose(st[:, 1], rstd, rtol=5e-6)


def test_layernorm_stats_out_are_the_statistics_of_its_output():
    C = 1024
    x = rnd(515, C, seed=2, scale=2.0, shift=-1.0)
    gamma = torch.rand(C, device=DEV) + 0.5
    beta = torch.randn(C, device=DEV) * 0.3
    st = torch.empty((515, 2), dtype=torch.float32, device=DEV)
    y = ops.layernorm(x, gamma, beta, stats_out=st)
    assert torch.equal(y, ops.layernorm(x, gamma, beta))
    assert torch.equal(st, ops.row_stats(y))                              # same two-pass reduction on the same bf16 values


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_ln_fold_weight(dtype):
    N, K = 384, 1024
    w = rnd(N, K, seed=3, dtype=dtype, scale=0.05)
    gamma = torch.rand(K, device=DEV) + 0.5
    beta = torch.randn(K, device=DEV) * 0.3
    bias = torch.randn(N, device=DEV)
    wf, colsum, d = ops.ln_fold_weight(w, gamma, beta, bias)
    want = (w.float() * gamma).to(dtype)
    bad = wf != want
    if dtype == torch.bfloat16:
        assert not bool(bad.any())
    else:       # half: one rounding each; allow last-place differences in the subnormal range, report how many
        assert (wf.float() - want.float()).abs().max().item() <= 2.0 ** -11 * want.float().abs().max().item(), int(bad.sum())
        assert int(bad.sum()) <= 0.01 * bad.numel(), f"{int(bad.sum())} of {bad.numel()} differ"
    assert torch.allclose(colsum, wf.float().sum(-1), rtol=1e-5, atol=1e-5)
    assert torch.allclose(d, w.float() @ beta + bias, rtol=1e-5, atol=1e-5)
    _, _, d0 = ops.ln_fold_weight(w, gamma, beta, None)
    assert torch.allclose(d0, w.float() @ beta, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,gelu,res", [(2048 + 32, 1024, 1024, False, False), (2304, 512, 1024, True, False),
                                            (1024 + 96, 768, 512, False, True), (300, 136, 256, True, True)])
def test_folded_gemm_against_the_unfolded_pair_and_fp32(dtype, M, N, K, gelu, res):
    x = rnd(M, K, seed=4, dtype=dtype, scale=1.7, shift=0.4)
    w = rnd(N, K, seed=5, dtype=dtype, scale=K ** -0.5)
    gamma = torch.rand(K, device=DEV) + 0.5
    beta = torch.randn(K, device=DEV) * 0.2
    bias = torch.randn(N, device=DEV).to(dtype).float()
    r = rnd(M, N, seed=6, dtype=dtype) if res else None
    wf, colsum, d = ops.ln_fold_weight(w, gamma, beta, bias)
    st = ops.row_stats(x)
    big = M >= 1024
    out = ops.gemm(x, wf, bias=d, residual=r, gelu=gelu, ln=(st, colsum), force_big=big)
    out_small = ops.gemm(x, wf, bias=d, residual=r, gelu=gelu, ln=(st, colsum), force_small=True)
    assert rel(out, out_small) < 2.0 ** -9                                # the two tilings (different MFMA shapes: not bit-identical)
    unfolded = ops.gemm(ops.layernorm(x, gamma, beta), w, bias=bias, residual=r, gelu=gelu, force_big=big)
    ref = ln_ref(x, gamma, beta) @ w.float().T + bias
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    if res:
        ref = ref + r.float()
    e_fold, e_pair = rel(out, ref), rel(unfolded, ref)
    eps16 = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert e_fold < 0.75 * eps16, (e_fold, e_pair)                        # rounding of the output only
    assert e_fold <= e_pair * 1.02, (e_fold, e_pair)                      # never worse than the pair that rounds z on the way
    assert rel(out, unfolded) < 1.5 * eps16


@pytest.mark.parametrize("M,N,K,res", [(2048 + 32, 1024, 512, True), (1300, 1024, 256, False), (2048, 320, 256, True), (200, 512, 128, False),
                                       (2304, 2048, 256, True)])
def test_producer_row_parts(M, N, K, res):
    a = rnd(M, K, seed=7, scale=1.3)
    w = rnd(N, K, seed=8, scale=K ** -0.5)
    bias = torch.randn(N, device=DEV)
    r = rnd(M, N, seed=9, scale=2.0, shift=3.0) if res else None
    nparts = (N + 255) // 256
    part = torch.full((M, nparts, 2), float("nan"), dtype=torch.float32, device=DEV)
    big = M >= 1024
    out = ops.gemm(a, w, bias=bias, residual=r, force_big=big, ln_part=part)
    assert torch.equal(out, ops.gemm(a, w, bias=bias, residual=r, force_big=big))          # the store path is untouched
    assert not torch.isnan(part).any()
    o = out.float()
    for j in range(nparts):
        sl = o[:, j * 256:(j + 1) * 256]
        assert torch.allclose(part[:, j, 0], sl.mean(-1), rtol=1e-5, atol=1e-5)
        m2 = ((sl - sl.mean(-1, keepdim=True)) ** 2).sum(-1)
        assert torch.allclose(part[:, j, 1], m2, rtol=2e-5, atol=1e-5)
    st = ops.row_stats_finalize(part, N)
    direct = ops.row_stats(out)
    assert torch.allclose(st[:, 0], direct[:, 0], rtol=1e-5, atol=1e-5)
    assert torch.allclose(st[:, 1], direct[:, 1], rtol=2e-5)


def test_producer_then_consumer_chain():
    """out-proj + residual -> (its row statistics) -> folded norm + ff1 + GELU: the chain of block.py:137-152 without the LayerNorm
    launch and without the normalised activation in HBM."""
    M, C, F = 4096 + 32, 1024, 2048
    ao = rnd(M, C, seed=10)
    h = rnd(M, C, seed=11, scale=2.0)
    w_o = rnd(C, C, seed=12, scale=C ** -0.5)
    b_o = torch.randn(C, device=DEV)
    w1 = rnd(F, C, seed=13, scale=C ** -0.5)
    b1 = torch.randn(F, device=DEV).bfloat16().float()
    gamma = torch.rand(C, device=DEV) + 0.5
    beta = torch.randn(C, device=DEV) * 0.2
    part = torch.empty((M, C // 256, 2), dtype=torch.float32, device=DEV)
    h2 = ops.gemm(ao, w_o, bias=b_o, residual=h, ln_part=part, force_big=True)
    st = ops.row_stats_finalize(part, C)
    wf, colsum, d = ops.ln_fold_weight(w1, gamma, beta, b1)
    y = ops.gemm(h2, wf, bias=d, gelu=True, ln=(st, colsum), force_big=True)
    y_pair = ops.gemm(ops.layernorm(h2, gamma, beta), w1, bias=b1, gelu=True)
    ref = torch.nn.functional.gelu(ln_ref(h2, gamma, beta) @ w1.float().T + b1)
    assert rel(y, ref) <= rel(y_pair, ref) * 1.02
    assert rel(y, ref) < 3e-3


def test_folded_projection_with_fused_head_split():
    """norm_s_attn folded into the q | k | v projection, through am_gemm_headpost_bf16 (fused epilogue + 128x128 tail + partial head
    split) against the un-fused sequence on the folded linear's own output."""
    heads, L, T, B = 2, 513, 16, 2                      # 16 416 rows: 64 full 256-row tiles (195 workgroups: fused) + 32 tail rows
    rows, C = B * T * L, 256
    N = heads * 3 * 128
    x = rnd(rows, C, seed=14, scale=1.5, shift=0.3)
    w = rnd(N, C, seed=15, scale=C ** -0.5)
    gamma = torch.rand(C, device=DEV) + 0.5
    beta = torch.randn(C, device=DEV) * 0.2
    wq = torch.rand(128, device=DEV) + 0.5
    wk = torch.rand(128, device=DEV) + 0.5
    wf, colsum, d = ops.ln_fold_weight(w, gamma, beta, None)
    st = ops.row_stats(x)
    q, k, vt = ops.gemm_head_post(x, wf, heads, (0, 1, 2), T * L, L, w_q=wq, w_k=wk, bias=d, ln=(st, colsum))
    lin = ops.gemm(x, wf, bias=d, ln=(st, colsum))
    q0, k0, vt0 = ops.head_post(lin, heads, (0, 1, 2), T * L, L, w_q=wq, w_k=wk)
    assert torch.equal(q, q0) and torch.equal(k, k0) and torch.equal(vt, vt0)


def test_canonical_row_statistics_against_the_numpy_restatement():
    """oracle/row_stats_oracle.py restates the canonical definition (groups of 8, balanced tree of equal-count merges, slices left to
    right) in numpy binary32.  Every producer must give ITS bits: the per-slice (mean, M2) a GEMM's store loop writes and the ones the
    read-back pass writes, exactly; the merged mean exactly; rstd to 2 ulp (the device's rsqrt is not the correctly rounded
    1 / sqrt)."""
    import numpy as np
    from oracle import row_stats_oracle as RO
    M, N, K = 2048 + 32, 1024, 256                      # 2048 rows through the store loop, 32 through the read-back pass
    a = rnd(M, K, seed=21, scale=1.3)
    w = rnd(N, K, seed=22, scale=K ** -0.5)
    r = rnd(M, N, seed=23, scale=2.0, shift=1.5)
    part = torch.empty((M, N // 256, 2), dtype=torch.float32, device=DEV)
    out = ops.gemm(a, w, residual=r, force_big=True, ln_part=part)
    mean, rstd, parts = RO.row_stats(out.float().cpu().numpy())
    got = part.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), parts.view(np.uint32)), int((got.view(np.uint32) != parts.view(np.uint32)).sum())
    for st in (ops.row_stats_finalize(part, N), ops.row_stats(out)):
        st = st.cpu().numpy()
        assert np.array_equal(st[:, 0].view(np.uint32), mean.view(np.uint32))
        ulp = np.abs(st[:, 1].view(np.int32).astype(np.int64) - rstd.view(np.int32).astype(np.int64))
        assert int(ulp.max()) <= 2, int(ulp.max())
    gamma = torch.rand(N, device=DEV) + 0.5
    beta = torch.randn(N, device=DEV) * 0.3
    st = torch.empty((M, 2), dtype=torch.float32, device=DEV)
    y = ops.layernorm(out, gamma, beta, stats_out=st)
    m2, _, _ = RO.row_stats(y.float().cpu().numpy())
    assert np.array_equal(st[:, 0].cpu().numpy().view(np.uint32), m2.view(np.uint32))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shift,outlier", [(50.0, 1.0), (500.0, 1.0), (50.0, 300.0), (500.0, 300.0)])
def test_fold_under_large_row_means_and_outlier_channels(dtype, shift, outlier):
    """VERDICT r04 weak #1 / ADVICE r04: the folded form evaluates rstd (x W'^T - mean colsum), which cancels `mean colsum` against
    the accumulated product - trained residual streams carry rows with |mean| / sigma >> 1 and a few channels 10^2..10^3 x the rest.
    Rows ~ 1.7 N(0, 1) + shift (|mean| rstd up to ~300), three channels scaled x outlier, through the whole chain: the PRODUCER's
    statistics (ln_part of a GEMM whose residual carries the shift and the outliers, merged by row_stats_finalize) feeding the folded
    consumer, against the un-folded pair (layernorm kernel + plain linear) and the fp32 statement.
    Stated: the folded result is no further from fp32 than the pair that rounds the normalised activation on the way (x 1.05), and its
    statistics match fp32 ones (mean to 2e-6 relative, rstd to 2e-5)."""
    M, C, N = 2048 + 32, 1024, 768
    eps16 = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    g = torch.Generator(device=DEV).manual_seed(int(shift) + int(outlier))
    base = torch.randn((M, C), device=DEV, generator=g) * 1.7
    ch = torch.tensor([7, 300, 777], device=DEV)
    base[:, ch] *= outlier
    sign = torch.where(torch.arange(M, device=DEV) % 2 == 0, 1.0, -1.0)[:, None]
    r = (base + shift * sign).to(dtype)                                     # the residual operand carries the hard statistics
    a = rnd(M, 256, seed=31, dtype=dtype, scale=0.5)
    w_o = rnd(C, 256, seed=32, dtype=dtype, scale=256 ** -0.5)
    part = torch.empty((M, C // 256, 2), dtype=torch.float32, device=DEV)
    x = ops.gemm(a, w_o, residual=r, ln_part=part, force_big=True)          # the producer: out-proj + residual (block.py:137)
    st = ops.row_stats_finalize(part, C, kind=dtype)
    xf = x.float()
    mean, rstd = xf.double().mean(-1), (xf.double().var(-1, unbiased=False) + 1e-5).rsqrt()
    assert torch.allclose(st[:, 0].double(), mean, rtol=2e-6, atol=1e-6), float((st[:, 0].double() - mean).abs().max())
    assert torch.allclose(st[:, 1].double(), rstd, rtol=2e-5)
    assert torch.equal(st, ops.row_stats(x))                                # the canonical statistics, whichever producer
    w = rnd(N, C, seed=33, dtype=dtype, scale=C ** -0.5)
    gamma = torch.rand(C, device=DEV) + 0.5
    beta = torch.randn(C, device=DEV) * 0.2
    bias = torch.randn(N, device=DEV).to(dtype).float()
    wf, colsum, d = ops.ln_fold_weight(w, gamma, beta, bias)
    folded = ops.gemm(x, wf, bias=d, ln=(st, colsum), force_big=True)
    pair = ops.gemm(ops.layernorm(x, gamma, beta), w, bias=bias, force_big=True)
    ref = (torch.nn.functional.layer_norm(xf.double(), (C,), gamma.double(), beta.double(), 1e-5) @ w.double().T + bias.double()).float()
    e_fold, e_pair = rel(folded, ref), rel(pair, ref)
    ratio = float((mean.abs() * rstd).max())
    print(f"LN fold stress {dtype} shift {shift} outlier x{outlier}: max |mean| rstd {ratio:.1f}; folded vs fp64 {e_fold:.3e}, "
          f"un-folded pair {e_pair:.3e}, folded vs pair {rel(folded, pair):.3e}")
    assert bool(torch.isfinite(folded.float()).all())
    assert e_fold <= 1.05 * e_pair and e_fold < 1.5 * eps16, (e_fold, e_pair)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_norm_out_folded_into_proj_out_through_the_row_map(dtype):
    """Round 6 (temporal_denoiser.py:239-242): norm_out -> drop each frame's time token -> proj_out (C -> 64) as
    ONE narrow linear that reads the un-normalised residual stream through its row map, with the statistics indexed by the MAPPED row.
    Against the un-folded pair (LayerNorm kernel, then the row-mapped linear) and the fp32 statement."""
    frames, N, C, Din = 6, 333, 1024, 64
    L = N + 1
    x = rnd(frames * L, C, seed=21, dtype=dtype, scale=1.9, shift=-0.7)
    w = rnd(Din, C, seed=22, dtype=dtype, scale=C ** -0.5)
    gamma = torch.rand(C, device=DEV) + 0.5
    beta = torch.randn(C, device=DEV) * 0.2
    bias = torch.randn(Din, device=DEV).to(dtype).float()
    wf, colsum, d = ops.ln_fold_weight(w, gamma, beta, bias)
    st = ops.row_stats(x)
    amap = (N, L, 1)
    out = ops.gemm(x, wf, bias=d, ln=(st, colsum), a_map=amap, M=frames * N)
    unfolded = ops.gemm(ops.layernorm(x, gamma, beta), w, bias=bias, a_map=amap, M=frames * N)
    keep = x.view(frames, L, C)[:, 1:].reshape(-1, C)
    ref = ln_ref(keep, gamma, beta) @ w.float().T + bias
    e_fold, e_pair = rel(out, ref), rel(unfolded, ref)
    eps16 = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert e_fold < 0.75 * eps16 and e_fold <= e_pair * 1.02, (e_fold, e_pair)
    # the statistics really are the mapped rows': shifting every time-token row (never read) changes nothing, bit for bit
    x2 = x.clone().view(frames, L, C)
    x2[:, 0] += 100.0
    x2 = x2.view(-1, C)
    assert torch.equal(ops.gemm(x2, wf, bias=d, ln=(ops.row_stats(x2), colsum), a_map=amap, M=frames * N), out)
