"""Test infrastructure (never on the product path): numpy restatement of the CANONICAL LayerNorm row statistics that
actionmesh_amd/csrc/am_common.h defines for the folded LayerNorms - the one definition every producer of a row's
(mean, rstd) must reproduce bit for bit (a GEMM's store loop, the read-back pass, a LayerNorm's own output, the stand-alone pass):

  slice   = 256 consecutive columns = 32 groups of 8 values
  level 0 = (mean, M2) of a group: mean = (((x0+x1)+(x2+x3))+((x4+x5)+(x6+x7))) / 8 ; M2 = fma chain over (x - mean)^2
  1 ... 5 = equal-count merges of neighbouring groups up a balanced binary tree:
            d = b.mean - a.mean ; M2 = fma(d*d, n/2, a.M2 + b.M2) ; mean = 0.5 (a.mean + b.mean)      (n = values in each of the two)
  slices -> row, left to right: tot = n + nj ; d = mj - mean ; w = nj / tot ; mean = fma(d, w, mean) ;
            M2 = fma(d*d, n*w, M2 + qj) ; n = tot ;   rstd = 1 / sqrt(M2 / n + eps)

All arithmetic in IEEE binary32; fma is emulated through x87 extended precision (numpy.longdouble: the product of two binary32 is
exact in 64 bits of mantissa and the sum is rounded once more at 64 bits before the final rounding to 24 - a double rounding that
can differ from a true fma only when the 64-bit sum falls exactly on a binary32 tie, probability ~2^-40 per operation).
"""
import numpy as np

F = np.float32
LD = np.longdouble


def fma(a, b, c):
    return (a.astype(LD) * b.astype(LD) + c.astype(LD)).astype(F)


def group_stats(x):
    """x (..., 8) float32 -> (mean, m2) of each group."""
    x = x.astype(F)
    mean = (((x[..., 0] + x[..., 1]) + (x[..., 2] + x[..., 3])) + ((x[..., 4] + x[..., 5]) + (x[..., 6] + x[..., 7]))) * F(0.125)
    m2 = np.zeros_like(mean)
    for e in range(8):
        d = x[..., e] - mean
        m2 = fma(d, d, m2)
    return mean.astype(F), m2.astype(F)


def slice_stats(x):
    """x (rows, 256) float32 -> (mean, M2) of each row's slice: the balanced tree over its 32 groups."""
    rows = x.shape[0]
    mean, m2 = group_stats(x.reshape(rows, 32, 8))
    half_n = F(4.0)
    while mean.shape[1] > 1:
        a_m, b_m, a_q, b_q = mean[:, 0::2], mean[:, 1::2], m2[:, 0::2], m2[:, 1::2]
        d = (b_m - a_m).astype(F)
        dd = (d * d).astype(F)
        s = (a_q + b_q).astype(F)
        m2 = fma(dd, np.full_like(dd, half_n), s)
        mean = (F(0.5) * (a_m + b_m)).astype(F)
        half_n = F(half_n * 2)
    return mean[:, 0], m2[:, 0]


def row_stats(x, eps=1e-5):
    """x (rows, C) float32 (the exact values of the 16-bit rows), C % 256 == 0 -> (mean, rstd, per-slice (mean, M2))."""
    rows, C = x.shape
    assert C % 256 == 0
    parts = [slice_stats(x[:, j * 256:(j + 1) * 256]) for j in range(C // 256)]
    n = np.zeros(rows, F); mean = np.zeros(rows, F); m2 = np.zeros(rows, F)
    for mj, qj in parts:
        nj = F(256.0)
        tot = (n + nj).astype(F)
        d = (mj - mean).astype(F)
        w = (nj / tot).astype(F)
        mean = fma(d, w, mean)
        dd = (d * d).astype(F)
        k = (n * w).astype(F)
        s = (m2 + qj).astype(F)
        m2 = fma(dd, k, s)
        n = tot
    var = (m2 / n).astype(F)
    rstd = (F(1.0) / np.sqrt((var + F(eps)).astype(F))).astype(F)
    return mean, rstd, np.stack([np.stack(p, -1) for p in parts], 1)
