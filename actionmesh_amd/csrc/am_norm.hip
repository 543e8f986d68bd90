// HBM-bound row kernels: LayerNorm, and the head split + qk-RMSNorm + RoPE +
// attention-operand layout kernel.  16-byte vector accesses throughout.
#include "am_common.h"

namespace {

// ---------------------------------------------------------------------------
// FP32LayerNorm (diffusers) / nn.LayerNorm: block.py:64,83,98,107 and
// temporal_denoiser.py:107.  One wave per row; lane holds NCH chunks of 8.
// ---------------------------------------------------------------------------
// Canonical (am_common.h) row statistics from the LayerNorm kernels' register layout: lane holds the 8-value groups (j * 64 + lane),
// i.e. group lane & 31 of slice 2 j + (lane >> 5).  C % 256 == 0.
template <int NCH>
__device__ __forceinline__ void canonical_row_stats(const float (&v)[NCH][8], int C, float eps, float& mean, float& rstd) {
  const int np = C >> 8;
  float n = 0.f, mu = 0.f, m2 = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    if (2 * j < np) {                       // wave-uniform
      float pm, pq;
      row_part8(v[j], pm, pq);
      float half_n = 4.f;
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) {
        const float om = __shfl_xor(pm, off), oq = __shfl_xor(pq, off);
        row_part_merge_equal(pm, pq, om, oq, half_n);
        half_n *= 2.f;
      }
      const float m_lo = __shfl(pm, 0), q_lo = __shfl(pq, 0), m_hi = __shfl(pm, 32), q_hi = __shfl(pq, 32);
      row_stats_merge(n, mu, m2, 256.f, m_lo, q_lo);
      if (2 * j + 1 < np) row_stats_merge(n, mu, m2, 256.f, m_hi, q_hi);
    }
  }
  mean = mu;
  rstd = row_stats_rstd(n, m2, eps);
}

// `stats` (optional, [rows][2] fp32): (mean, rstd) for a linear that absorbs the NEXT LayerNorm (am_gemm_args.ln_stats) - of
// the rows of x when y == nullptr (statistics only: one read of x, nothing else), of the bf16-rounded OUTPUT rows otherwise.
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        int64_t rows, int C, float eps, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * C;
  float v[NCH][8];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int col = (j * 64 + lane) * 8;
    if (col < C) {
      const u32x4_t u = *reinterpret_cast<const u32x4_t*>(xr + col);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[j][2 * e] = bflo(u[e]);
        v[j][2 * e + 1] = bfhi(u[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[j][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int col = (j * 64 + lane) * 8;
    if (col < C) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[j][e] - mean;
        sq += d * d;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
  const float rstd = rsqrtf(sq / (float)C + eps);
  if (y == nullptr) {                   // statistics only: the canonical form when the row is whole slices, the two-pass values otherwise
    float cm = mean, cr = rstd;
    if ((C & 255) == 0) canonical_row_stats<NCH>(v, C, eps, cm, cr);
    if (lane == 0) *reinterpret_cast<f32x2_t*>(stats + 2 * row) = f32x2_t{cm, cr};
    return;
  }
  bf16_t* yr = y + row * C;
  float osum = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int col = (j * 64 + lane) * 8;
    if (col < C) {
      const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(w + col);
      const f32x4_t w1 = *reinterpret_cast<const f32x4_t*>(w + col + 4);
      const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(b + col);
      const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(b + col + 4);
      float o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (v[j][e] - mean) * rstd * w0[e] + b0[e];
        o[4 + e] = (v[j][4 + e] - mean) * rstd * w1[e] + b1[e];
      }
      u32x4_t u;
#pragma unroll
      for (int e = 0; e < 4; ++e) u[e] = pack_bf2(o[2 * e], o[2 * e + 1]);
      *reinterpret_cast<u32x4_t*>(yr + col) = u;
      if (stats) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[j][2 * e] = bflo(u[e]);
          v[j][2 * e + 1] = bfhi(u[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) osum += v[j][e];
      }
    }
  }
  if (stats) {          // statistics of the rows just written (v now holds the bf16-rounded outputs)
    float om, orstd;
    if ((C & 255) == 0) {
      canonical_row_stats<NCH>(v, C, eps, om, orstd);
    } else {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) osum += __shfl_xor(osum, off);
      om = osum / (float)C;
      float osq = 0.f;
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int col = (j * 64 + lane) * 8;
        if (col < C) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = v[j][e] - om;
            osq += d * d;
          }
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) osq += __shfl_xor(osq, off);
      orstd = rsqrtf(osq / (float)C + eps);
    }
    if (lane == 0) *reinterpret_cast<f32x2_t*>(stats + 2 * row) = f32x2_t{om, orstd};
  }
}

// am_add_bias_rows (am_elementwise.hip: h += to_out bias, the exact result of a cross-attention branch over an all-zero context) in the
// LayerNorm kernels' row layout, leaving the canonical (mean, rstd) of the rows it writes - one pass over h instead of two.
template <int NCH>
__global__ __launch_bounds__(256) void add_bias_rows_stats_kernel(bf16_t* __restrict__ h, const float* __restrict__ bias, int64_t rows, int C,
                                                                  float eps, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  bf16_t* hr = h + row * C;
  float v[NCH][8];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int col = (j * 64 + lane) * 8;
    if (col < C) {
      u32x4_t u = *reinterpret_cast<const u32x4_t*>(hr + col);
      const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(bias + col), b1 = *reinterpret_cast<const f32x4_t*>(bias + col + 4);
      u[0] = pack_bf2(bflo(u[0]) + rbf(b0[0]), bfhi(u[0]) + rbf(b0[1]));
      u[1] = pack_bf2(bflo(u[1]) + rbf(b0[2]), bfhi(u[1]) + rbf(b0[3]));
      u[2] = pack_bf2(bflo(u[2]) + rbf(b1[0]), bfhi(u[2]) + rbf(b1[1]));
      u[3] = pack_bf2(bflo(u[3]) + rbf(b1[2]), bfhi(u[3]) + rbf(b1[3]));
      *reinterpret_cast<u32x4_t*>(hr + col) = u;
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[j][2 * e] = bflo(u[e]); v[j][2 * e + 1] = bfhi(u[e]); }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
    }
  }
  float mean, rstd;
  canonical_row_stats<NCH>(v, C, eps, mean, rstd);
  if (lane == 0) *reinterpret_cast<f32x2_t*>(stats + 2 * row) = f32x2_t{mean, rstd};
}

// (mean, M2) of every 256-column slice of rows [row0, row0 + rows) of X - the rows a producer GEMM's store loop did not cover
// (am_gemm_args.ln_part; emit_row_part in am_gemm.hip computes the same pairs in flight).  One wave per row, 4 columns per lane.
__global__ __launch_bounds__(256) void row_part_kernel(const bf16_t* __restrict__ X, int ldx, int64_t row0, int64_t rows, int N,
                                                       float* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int64_t row = row0 + r;
  const int nparts = (N + 255) >> 8, nfull = N >> 8;
  // whole slices, two per trip (one per half-wave): the canonical groups and tree of am_common.h
  for (int t = 0; 2 * t < nfull; ++t) {
    const int sl = 2 * t + (lane >> 5);
    float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (sl < nfull) {
      const u32x4_t u = *reinterpret_cast<const u32x4_t*>(X + row * ldx + sl * 256 + (lane & 31) * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[2 * e] = bflo(u[e]); x[2 * e + 1] = bfhi(u[e]); }
    }
    float pm, pq;
    row_part8(x, pm, pq);
    float half_n = 4.f;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      const float om = __shfl_xor(pm, off), oq = __shfl_xor(pq, off);
      row_part_merge_equal(pm, pq, om, oq, half_n);
      half_n *= 2.f;
    }
    if ((lane & 31) == 0 && sl < nfull) *reinterpret_cast<f32x2_t*>(part + 2 * (row * nparts + sl)) = f32x2_t{pm, pq};
  }
  if (nfull < nparts) {          // a short last slice (N % 256 != 0: no GEMM store loop writes these): plain two-pass, 4 columns per lane
    const int j = nfull;
    const int col = j * 256 + lane * 4;
    const int cnt = N - j * 256;
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    const bool in = col < N;                 // N % 8 == 0 and col % 4 == 0: a lane's four columns are in or out together
    if (in) {
      const u32x2_t u = *reinterpret_cast<const u32x2_t*>(X + row * ldx + col);
      x[0] = bflo(u[0]); x[1] = bfhi(u[0]); x[2] = bflo(u[1]); x[3] = bfhi(u[1]);
    }
    float sum = (x[0] + x[1]) + (x[2] + x[3]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / (float)cnt;
    float sq = 0.f;
    if (in) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = x[e] - mean; sq += d * d; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    if (lane == 0) *reinterpret_cast<f32x2_t*>(part + 2 * (row * nparts + j)) = f32x2_t{mean, sq};
  }
}

// Merge the slices of a row: running (n, mean, M2) += (n_j, mean_j, M2_j) by Chan et al.'s update.  One thread per row.
__global__ __launch_bounds__(256) void row_stats_finalize_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ stats,
                                                                 int64_t rows, float eps) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const f32x2_t* p = reinterpret_cast<const f32x2_t*>(part) + row * nparts;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int j = 0; j < nparts; ++j) {
    const f32x2_t q = p[j];
    row_stats_merge(n, mean, m2, (float)min(256, C - j * 256), q[0], q[1]);
  }
  *reinterpret_cast<f32x2_t*>(stats + 2 * row) = f32x2_t{mean, row_stats_rstd(n, m2, eps)};
}

// Weight preparation of a linear that absorbs its LayerNorm (am_ln_fold_weight).  One wave per output row n.
__global__ __launch_bounds__(256) void ln_fold_weight_kernel(const bf16_t* __restrict__ W, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ bias,
                                                             bf16_t* __restrict__ Wf, float* __restrict__ colsum, float* __restrict__ dvec,
                                                             int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float cs = 0.f, dd = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float w = bf2f(W[(int64_t)n * K + k]);
    const bf16_t wf = f2bf(w * gamma[k]);
    Wf[(int64_t)n * K + k] = wf;
    cs += bf2f(wf);
    dd = fmaf(w, beta[k], dd);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { cs += __shfl_xor(cs, off); dd += __shfl_xor(dd, off); }
  if (lane == 0) {
    colsum[n] = cs;
    dvec[n] = dd + (bias ? bias[n] : 0.f);
  }
}

// ---------------------------------------------------------------------------
// head_post: attention_processor.py:106-130.  One workgroup = 64 consecutive
// tokens of one sequence x one head x one part (q / k / v).
//   Q/K parts: per-(token, head) RMSNorm over 128 (fp32, affine), optional
//              interleaved-pair RoPE (fp32), rounded to bf16 (SDPA entry cast).
//   V part:    transposed through LDS to [128][positions] with perm16 key order.
// 16 lanes cooperate on one token (8 elements each).
// ---------------------------------------------------------------------------
constexpr int HP_TOK = 64;
constexpr int VT_LD = HP_TOK + 2;  // bf16 per LDS row of the transposed V tile (33 dwords: odd stride)

// Partial launches (am_head_post_partial, used behind the fused QKV GEMM of am_gemm.hip): only the 64-token blocks blk0 .. of every
// sequence are visited, and tokens s < s_min (s_min_last for the last sequence, s_min_other for the others; multiples of 16) are neither
// read nor written - what is left are the rows the GEMM's 128x128 tail kernel produced and the zero fill of the pad rows / columns.
__global__ __launch_bounds__(256) void head_post_kernel(am_headpost_args p, int blocks_per_seq, int blk0, int s_min_last, int s_min_other,
                                                        int nseq) {
  __shared__ bf16_t vt[128 * VT_LD];
  const int tid = threadIdx.x;
  const int nblk = blocks_per_seq - blk0;
  const int sblk = blk0 + blockIdx.x % nblk;
  const int sidx = blockIdx.x / nblk;   // sequence index
  const int s_min = sidx == nseq - 1 ? s_min_last : s_min_other;
  const int head = blockIdx.y;
  const int part = blockIdx.z;
  const int kind = p.kinds[part];
  const int s0 = sblk * HP_TOK;
  const int sub = tid & 15;         // which 8 of the 128 channels
  const int tok_in_pass = tid >> 4; // 16 tokens per pass
  const int col = (head * p.nparts + part) * 128 + sub * 8;

  if (kind != 2) {
    const float* wt = kind == 0 ? p.w_q : p.w_k;
    float wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) wv[e] = wt ? wt[sub * 8 + e] : 1.f;
    bf16_t* out = kind == 0 ? p.out_q : p.out_k;
    const int s_pad = kind == 0 ? p.sq_pad : p.sk_pad;
#pragma unroll
    for (int pass = 0; pass < HP_TOK / 16; ++pass) {
      const int s = s0 + pass * 16 + tok_in_pass;
      if (s < s_min) continue;          // partial launch: this token's row was written by the fused GEMM epilogue
      if (s >= p.seq_len) {             // uniform per 16-lane group
        // K rows of the padded tail of the last key tile must be zero: the attention kernel
        // relies on score(padded key) == 0 instead of masking (am_attention.hip, tail_fix)
        if (kind == 1 && s < s_pad) {
          bf16_t* dst = out + (((int64_t)sidx * p.heads + head) * s_pad + s) * 128 + sub * 8;
          *reinterpret_cast<u32x4_t*>(dst) = u32x4_t{0u, 0u, 0u, 0u};
        }
        continue;
      }
      const int64_t row = (int64_t)sidx * p.seq_len + s;
      const u32x4_t u = *reinterpret_cast<const u32x4_t*>(p.X + row * p.ldx + col);
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] = bflo(u[e]);
        v[2 * e + 1] = bfhi(u[e]);
      }
      if (wt) {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(v[e], v[e], ss);      // one rounding per term, in both kernels that normalise
        // butterfly over the 16 lanes of a token as DPP row rotations: ss is (16 / step)-periodic after each step, so rotating by
        // `step` meets the same partner value as lane ^ step - bit-identical to the __shfl_xor form, no LDS-crossbar round trip
        ss += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss), 0x128, 0xf, 0xf, false));
        ss += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss), 0x124, 0xf, 0xf, false));
        ss += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss), 0x122, 0xf, 0xf, false));
        ss += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss), 0x121, 0xf, 0xf, false));
        const float r = rsqrtf(ss * (1.0f / 128.0f) + p.eps);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] * r) * wv[e];
      }
      if (p.rope_cos) {
        const int64_t frame = row / p.rows_per_frame;
        const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(p.rope_cos + frame * 64 + sub * 4);
        const f32x4_t sn = *reinterpret_cast<const f32x4_t*>(p.rope_sin + frame * 64 + sub * 4);
        // NOTE (round 3): this file is built WITHOUT the SLP vectoriser.  Packed, the rotation below becomes
        // v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[0,0], and on MI355X that instruction returns a wrong low half in lanes 48-63
        // while another PROCESS runs bf16 GEMMs on the device (tools/repro/pk_mul_cross_process.hip) - the same-device divergence
        // of round 2.  tests/test_host_cpu.py audits the built library for that operand-selection form.
#pragma unroll
        for (int e = 0; e < 4; ++e) rope_rotate(v[2 * e], v[2 * e + 1], cs[e], sn[e]);    // rotary_embedding.py:116-122
      }
      u32x4_t w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
      bf16_t* dst = out + (((int64_t)sidx * p.heads + head) * s_pad + s) * 128 + sub * 8;
      *reinterpret_cast<u32x4_t*>(dst) = w;
    }
  } else {
    // V: gather 64 tokens x 128 d, transpose in LDS, write 128 rows of 64 positions.
#pragma unroll
    for (int pass = 0; pass < HP_TOK / 16; ++pass) {
      const int tl = pass * 16 + tok_in_pass;
      const int s = s0 + tl;
      u32x4_t u = {0u, 0u, 0u, 0u};
      if (s < p.seq_len && s >= s_min) {
        const int64_t row = (int64_t)sidx * p.seq_len + s;
        u = *reinterpret_cast<const u32x4_t*>(p.X + row * p.ldx + col);
      }
      const int pos = (tl & ~15) | perm16(tl & 15);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        vt[(sub * 8 + 2 * e) * VT_LD + pos] = (bf16_t)(u[e] & 0xffffu);
        vt[(sub * 8 + 2 * e + 1) * VT_LD + pos] = (bf16_t)(u[e] >> 16);
      }
    }
    __syncthreads();
    // 128 rows x 128 B; 8 threads per row (16 B each), 32 rows per pass
    const int c8 = (tid & 7) * 8;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int d = pass * 32 + (tid >> 3);
      const uint32_t* src = reinterpret_cast<const uint32_t*>(&vt[d * VT_LD + c8]);  // 4-byte aligned (VT_LD even)
      u32x4_t w = {src[0], src[1], src[2], src[3]};
      bf16_t* dst = p.out_vt + (((int64_t)sidx * p.heads + head) * 128 + d) * p.sk_pad + s0 + c8;
      if (s0 + c8 >= s_min) *reinterpret_cast<u32x4_t*>(dst) = w;       // s_min % 16 == 0: a whole 16-key group is in or out
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// fp32 residual stream (round 6).  The reference's Stage II keeps its residual stream in fp32: torch.cat of the bf16 projected
// latents with the fp32 alpha embedding promotes (temporal_autoencoder.py:258), every `h + branch` under autocast then adds a 16-bit
// linear output into fp32, FP32LayerNorm returns fp32 and the next autocast linear rounds its input to 16 bits; the DINOv2 encoder runs
// in fp32 altogether (pipeline.py:665-667).  One pass per branch:  h32 += y16 (the branch's 16-bit linear output; may be NULL), then
// z16 = LayerNorm(h32) rounded to 16 bits for the next linear (z16 / w / b may be NULL: accumulate only).  One wave per row.
template <int NCH>
__global__ __launch_bounds__(256) void add_layernorm_f32_kernel(float* __restrict__ h, const bf16_t* __restrict__ y, bf16_t* __restrict__ z,
                                                                const float* __restrict__ w, const float* __restrict__ b,
                                                                int64_t rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* hr = h + row * C;
  float v[NCH][8];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int col = (j * 64 + lane) * 8;
    if (col < C) {
      const f32x4_t h0 = *reinterpret_cast<const f32x4_t*>(hr + col), h1 = *reinterpret_cast<const f32x4_t*>(hr + col + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[j][e] = h0[e]; v[j][4 + e] = h1[e]; }
      if (y) {
        const u32x4_t u = *reinterpret_cast<const u32x4_t*>(y + row * C + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[j][2 * e] += bflo(u[e]); v[j][2 * e + 1] += bfhi(u[e]); }
        *reinterpret_cast<f32x4_t*>(hr + col) = f32x4_t{v[j][0], v[j][1], v[j][2], v[j][3]};
        *reinterpret_cast<f32x4_t*>(hr + col + 4) = f32x4_t{v[j][4], v[j][5], v[j][6], v[j][7]};
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[j][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
    }
  }
  if (z == nullptr) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int col = (j * 64 + lane) * 8;
    if (col < C) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[j][e] - mean;
        sq += d * d;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
  const float rstd = rsqrtf(sq / (float)C + eps);
  bf16_t* zr = z + row * C;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int col = (j * 64 + lane) * 8;
    if (col < C) {
      const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(w + col), w1 = *reinterpret_cast<const f32x4_t*>(w + col + 4);
      const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(b + col), b1 = *reinterpret_cast<const f32x4_t*>(b + col + 4);
      float o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (v[j][e] - mean) * rstd * w0[e] + b0[e];
        o[4 + e] = (v[j][4 + e] - mean) * rstd * w1[e] + b1[e];
      }
      u32x4_t u;
#pragma unroll
      for (int e = 0; e < 4; ++e) u[e] = pack_bf2(o[2 * e], o[2 * e + 1]);
      *reinterpret_cast<u32x4_t*>(zr + col) = u;
    }
  }
}

int launch_layernorm(const uint16_t* x, uint16_t* y, const float* w, const float* b, int64_t rows, int C, float eps, float* stats,
                     void* stream) {
  const dim3 grid(ceil_div(rows, 4)), block(256);
  const int nch = ceil_div(C, 512);
  hipStream_t s = (hipStream_t)stream;
  if (nch <= 1) hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, x, y, w, b, rows, C, eps, stats);
  else if (nch <= 2) hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, x, y, w, b, rows, C, eps, stats);
  else if (nch <= 4) hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, x, y, w, b, rows, C, eps, stats);
  else hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, s, x, y, w, b, rows, C, eps, stats);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

}  // namespace

extern "C" int am_layernorm_bf16(const uint16_t* x, uint16_t* y, const float* w, const float* b,
                                 int64_t rows, int C, float eps, void* stream) {
  AM_CHECK(x && y && w && b, "am_layernorm_bf16: null operand");
  AM_CHECK(rows > 0 && C > 0 && C % 8 == 0 && C <= 4096, "am_layernorm_bf16: bad shape rows=%lld C=%d", (long long)rows, C);
  AM_CHECK(((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)b) % 16 == 0, "am_layernorm_bf16: operands misaligned");
  return launch_layernorm(x, y, w, b, rows, C, eps, nullptr, stream);
}

extern "C" int am_add_layernorm_f32(float* h, const uint16_t* y, uint16_t* z, const float* w, const float* b, int64_t rows, int C, float eps,
                                    void* stream) {
  AM_CHECK(h, "am_add_layernorm_f32: null residual stream");
  AM_CHECK(z == nullptr || (w && b), "am_add_layernorm_f32: a LayerNorm output needs weight and bias");
  AM_CHECK(y || z, "am_add_layernorm_f32: nothing to do (neither a branch to add nor an output)");
  AM_CHECK(rows > 0 && C > 0 && C % 8 == 0 && C <= 4096, "am_add_layernorm_f32: bad shape rows=%lld C=%d", (long long)rows, C);
  AM_CHECK(((uintptr_t)h | (uintptr_t)y | (uintptr_t)z | (uintptr_t)w | (uintptr_t)b) % 16 == 0, "am_add_layernorm_f32: operands misaligned");
  const dim3 grid(ceil_div(rows, 4)), block(256);
  const int nch = ceil_div(C, 512);
  hipStream_t s = (hipStream_t)stream;
  const bf16_t* yy = reinterpret_cast<const bf16_t*>(y);
  bf16_t* zz = reinterpret_cast<bf16_t*>(z);
  if (nch <= 1) hipLaunchKernelGGL(add_layernorm_f32_kernel<1>, grid, block, 0, s, h, yy, zz, w, b, rows, C, eps);
  else if (nch <= 2) hipLaunchKernelGGL(add_layernorm_f32_kernel<2>, grid, block, 0, s, h, yy, zz, w, b, rows, C, eps);
  else if (nch <= 4) hipLaunchKernelGGL(add_layernorm_f32_kernel<4>, grid, block, 0, s, h, yy, zz, w, b, rows, C, eps);
  else hipLaunchKernelGGL(add_layernorm_f32_kernel<8>, grid, block, 0, s, h, yy, zz, w, b, rows, C, eps);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

extern "C" int am_layernorm_stats_bf16(const uint16_t* x, uint16_t* y, const float* w, const float* b,
                                       int64_t rows, int C, float eps, float* stats_y, void* stream) {
  AM_CHECK(x && y && w && b && stats_y, "am_layernorm_stats_bf16: null operand");
  AM_CHECK(rows > 0 && C > 0 && C % 8 == 0 && C <= 4096, "am_layernorm_stats_bf16: bad shape rows=%lld C=%d", (long long)rows, C);
  AM_CHECK(((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)b) % 16 == 0 && (uintptr_t)stats_y % 8 == 0,
           "am_layernorm_stats_bf16: operands misaligned");
  return launch_layernorm(x, y, w, b, rows, C, eps, stats_y, stream);
}

extern "C" int am_row_stats_bf16(const uint16_t* x, float* stats, int64_t rows, int C, float eps, void* stream) {
  AM_CHECK(x && stats, "am_row_stats_bf16: null operand");
  AM_CHECK(rows > 0 && C > 0 && C % 8 == 0 && C <= 4096, "am_row_stats_bf16: bad shape rows=%lld C=%d", (long long)rows, C);
  AM_CHECK((uintptr_t)x % 16 == 0 && (uintptr_t)stats % 8 == 0, "am_row_stats_bf16: operands misaligned");
  return launch_layernorm(x, nullptr, nullptr, nullptr, rows, C, eps, stats, stream);
}

int am_add_bias_rows(bf16_t* h, const float* bias, int64_t rows, int C, void* stream);     // am_elementwise.hip

// internal (am_model.hip): h += bias over `rows` rows and the rows' (mean, rstd) into stats
int am_add_bias_rows_stats(bf16_t* h, const float* bias, int64_t rows, int C, float eps, float* stats, void* stream) {
  if (rows <= 0) return AM_OK;
  if (C % 256 != 0 || C > 4096) {
    AM_TRY(am_add_bias_rows(h, bias, rows, C, stream));
    return am_row_stats_bf16(h, stats, rows, C, eps, stream);
  }
  const dim3 grid(ceil_div(rows, 4)), block(256);
  const int nch = C / 512;
  hipStream_t s = (hipStream_t)stream;
  if (nch <= 1) hipLaunchKernelGGL(add_bias_rows_stats_kernel<1>, grid, block, 0, s, h, bias, rows, C, eps, stats);
  else if (nch <= 2) hipLaunchKernelGGL(add_bias_rows_stats_kernel<2>, grid, block, 0, s, h, bias, rows, C, eps, stats);
  else if (nch <= 4) hipLaunchKernelGGL(add_bias_rows_stats_kernel<4>, grid, block, 0, s, h, bias, rows, C, eps, stats);
  else hipLaunchKernelGGL(add_bias_rows_stats_kernel<8>, grid, block, 0, s, h, bias, rows, C, eps, stats);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

int am_row_part(const bf16_t* Cmat, int ldc, int64_t row0, int64_t rows, int N, float* part, void* stream) {
  if (rows <= 0) return AM_OK;
  hipLaunchKernelGGL(row_part_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, Cmat, ldc, row0, rows, N, part);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

extern "C" int am_row_stats_finalize(const float* part, int nparts, int C, float* stats, int64_t rows, float eps, void* stream) {
  AM_CHECK(part && stats, "am_row_stats_finalize: null operand");
  AM_CHECK(rows > 0 && C > 0 && nparts == (C + 255) / 256, "am_row_stats_finalize: bad shape rows=%lld C=%d nparts=%d", (long long)rows, C, nparts);
  AM_CHECK(((uintptr_t)part | (uintptr_t)stats) % 8 == 0, "am_row_stats_finalize: operands misaligned");
  hipLaunchKernelGGL(row_stats_finalize_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, (hipStream_t)stream, part, nparts, C, stats, rows, eps);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

extern "C" int am_ln_fold_weight(const uint16_t* W, const float* gamma, const float* beta, const float* bias, uint16_t* Wf, float* colsum,
                                 float* d, int N, int K, void* stream) {
  AM_CHECK(W && gamma && beta && Wf && colsum && d, "am_ln_fold_weight: null operand");
  AM_CHECK(N > 0 && K > 0, "am_ln_fold_weight: bad shape N=%d K=%d", N, K);
  hipLaunchKernelGGL(ln_fold_weight_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const bf16_t*>(W), gamma,
                     beta, bias, reinterpret_cast<bf16_t*>(Wf), colsum, d, N, K);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

int am_head_post_check(const am_headpost_args* a);

extern "C" int am_head_post(const am_headpost_args* a, void* stream) {
  AM_TRY(am_head_post_check(a));
  const int nseq = (int)(a->rows / a->seq_len);
  const int bps = ceil_div(a->seq_len, HP_TOK);
  dim3 grid((unsigned)((int64_t)nseq * bps), a->heads, a->nparts);
  hipLaunchKernelGGL(head_post_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a, bps, 0, 0, 0, nseq);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

int am_head_post_check(const am_headpost_args* a) {
  AM_CHECK(a && a->X, "am_head_post: null args");
  AM_CHECK(a->nparts >= 1 && a->nparts <= 3, "am_head_post: nparts=%d", a->nparts);
  AM_CHECK(a->rows > 0 && a->seq_len > 0 && a->rows % a->seq_len == 0, "am_head_post: rows=%lld not a multiple of seq_len=%d",
           (long long)a->rows, a->seq_len);
  AM_CHECK(a->rows_per_frame > 0, "am_head_post: rows_per_frame");
  AM_CHECK(a->ldx % 8 == 0 && (uintptr_t)a->X % 16 == 0, "am_head_post: X misaligned");
  AM_CHECK((a->rope_cos == nullptr) == (a->rope_sin == nullptr), "am_head_post: rope tables must come in pairs");
  for (int i = 0; i < a->nparts; ++i) {
    const int k = a->kinds[i];
    AM_CHECK(k >= 0 && k <= 2, "am_head_post: kind[%d]=%d", i, k);
    if (k == 0) AM_CHECK(a->out_q && a->sq_pad >= a->seq_len && (uintptr_t)a->out_q % 16 == 0, "am_head_post: out_q");
    if (k == 1) AM_CHECK(a->out_k && a->sk_pad >= a->seq_len && (uintptr_t)a->out_k % 16 == 0, "am_head_post: out_k");
    if (k == 2) AM_CHECK(a->out_vt && a->sk_pad % 64 == 0 && a->sk_pad >= round_up(a->seq_len, 64) && (uintptr_t)a->out_vt % 16 == 0,
                         "am_head_post: out_vt / sk_pad=%d", a->sk_pad);
  }
  AM_CHECK(a->heads <= 65535, "am_head_post: heads");
  return AM_OK;
}

// Internal (am_gemm.hip, fused QKV epilogue): the tokens s >= s_min of every sequence only - see head_post_kernel.
int am_head_post_partial(const am_headpost_args* a, int s_min_last, int s_min_other, void* stream) {
  const int nseq = (int)(a->rows / a->seq_len);
  const int bps = ceil_div(a->seq_len, HP_TOK);
  const int lo = s_min_last < s_min_other ? s_min_last : s_min_other;
  const int blk0 = lo / HP_TOK;
  if (blk0 >= bps) return AM_OK;                         // nothing left: no tail rows and no pad rows / columns
  dim3 grid((unsigned)((int64_t)nseq * (bps - blk0)), a->heads, a->nparts);
  hipLaunchKernelGGL(head_post_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a, bps, blk0, s_min_last, s_min_other, nseq);
  AM_HIP(hipGetLastError());
  return AM_OK;
}
