// HBM-bound row kernels: LayerNorm, and the head split + qk-RMSNorm + RoPE +
// attention-operand layout kernel.  16-byte vector accesses throughout.
#include "am_common.h"

namespace {

// ---------------------------------------------------------------------------
// FP32LayerNorm (diffusers) / nn.LayerNorm: block.py:64,83,98,107 and
// temporal_denoiser.py:107.  One wave per row; lane holds NCH chunks of 8.
// ---------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        int64_t rows, int C, float eps) {
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
  bf16_t* yr = y + row * C;
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
    }
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

}  // namespace

extern "C" int am_layernorm_bf16(const uint16_t* x, uint16_t* y, const float* w, const float* b,
                                 int64_t rows, int C, float eps, void* stream) {
  AM_CHECK(x && y && w && b, "am_layernorm_bf16: null operand");
  AM_CHECK(rows > 0 && C > 0 && C % 8 == 0 && C <= 4096, "am_layernorm_bf16: bad shape rows=%lld C=%d", (long long)rows, C);
  AM_CHECK(((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)b) % 16 == 0, "am_layernorm_bf16: operands misaligned");
  const dim3 grid(ceil_div(rows, 4)), block(256);
  const int nch = ceil_div(C, 512);
  hipStream_t s = (hipStream_t)stream;
  if (nch <= 1) hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, x, y, w, b, rows, C, eps);
  else if (nch <= 2) hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, x, y, w, b, rows, C, eps);
  else if (nch <= 4) hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, x, y, w, b, rows, C, eps);
  else hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, s, x, y, w, b, rows, C, eps);
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
