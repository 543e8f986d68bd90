// Flash attention forward (non-causal, head_dim 128, bf16 MFMA, fp32 online
// softmax) for gfx950.  Replaces F.scaled_dot_product_attention at
// attention_processor.py:133-139 for both the inflated self-attention
// (seq = T*L, 68-99 % of the step's flops) and the per-frame cross-attention.
//
// Work decomposition: one workgroup = 8 waves = 256 query rows of one
// (sequence, head); each wave owns 32 query rows.  Keys/values stream through
// LDS in tiles of 64 keys, double-buffered, staged global -> VGPR -> LDS.
//
// MFMA formulation (v_mfma_f32_32x32x16_bf16, "swapped" so that the softmax
// row is lane-local and P never leaves registers):
//   S^T[key][q] = K[key][:] . Q[q][:]        A-operand = K rows  (LDS, 16 B reads)
//                                            B-operand = Q       (registers, loaded once)
//   lane (q = lane&31, hi = lane>>5) holds S^T[key = (r&3)+8(r>>2)+4hi][q], r=0..15
//   O^T[d][q]  += V^T[d][key] . P^T[key][q]  A-operand = V^T rows (LDS, 16 B reads)
//                                            B-operand = P^T straight from the S registers
// The B-operand k-slot (hi, j) of the P.V MFMA then carries key
// (j&3)+8(j>>2)+4hi of its 16-key group, so V^T is stored (by am_head_post)
// with keys permuted inside each group of 16 (perm16: bit2<->bit3) and the
// matching A-operand is one contiguous 16-byte LDS read.
//
// Schedule (STAGGER = true): a CU holds one workgroup = two waves per SIMD
// (wave i and wave i+4 share SIMD i).  Waves 4-7 run exactly one phase behind
// waves 0-3 (they take one extra s_barrier before the loop, waves 0-3 one
// after it), and every tile is two phases separated by barriers:
//   phase 1: S^T = K Q^T (16 MFMA)          | also: LDS-write K(t+1), issue V(t+1) loads
//   phase 2: softmax (VALU) then P.V (16 MFMA) | also: issue K(t+2) loads, LDS-write V(t+1)
// so on every SIMD one wave's QK^T MFMAs run beside its partner's softmax
// VALU work and are followed by the partner's P.V MFMAs: the matrix pipe is
// never idle during a softmax.  Each half-workgroup stages its own half of
// every K / V^T tile; the hazard analysis (which global phase may write which
// LDS buffer) is in DESIGN.md section "attention".
//
// Multi-GPU: K/V arrive as `nchunks` frame shards ([chunk][seq][head]...);
// softmax is permutation-invariant over keys, so chunks are simply
// concatenated tile streams, each with its own valid-key count.
#include "am_common.h"

namespace {

constexpr int QBLK = 256;          // query rows per workgroup
constexpr int KVBLK = 64;          // keys per tile
constexpr int HD = 128;
constexpr int K_LD = HD + 8;       // padded K row in LDS (272 B): 16 distinct 16-B slots per lane group
constexpr int V_LD = KVBLK + 8;    // padded V^T row in LDS (144 B)
constexpr int K_TILE = KVBLK * K_LD;
constexpr int V_TILE = HD * V_LD;
constexpr int SMEM_BYTES = 2 * (K_TILE + V_TILE) * (int)sizeof(bf16_t);   // 71680

template <int DEFER, bool STAGGER, int ABL>
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel(am_attn_args p, int tiles_per_chunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);       // [2][64][K_LD]
  bf16_t* Vs = Ks + 2 * K_TILE;                       // [2][128][V_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.y;                          // sequence * heads + head
  const int head = bh % p.heads, seq = bh / p.heads;
  const int q0 = blockIdx.x * QBLK + wave * 32;
  const bool late = STAGGER && (__builtin_amdgcn_readfirstlane(tid) >= 256);   // waves 4-7: one phase behind

  // ---- Q fragments (B operand): Q[q0 + l31][ks*16 + hi*8 .. +8] -------------
  bf16x8_t qf[8];
  {
    const bf16_t* qp = p.Q + ((int64_t)bh * p.sq_pad + q0 + l31) * HD + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 16);
  }

  // ---- K / V^T tile staging: 1024 x 16 B each, two per thread.  With STAGGER each
  // half-workgroup (256 threads) owns one half of every tile: c in [half*512, half*512+512).
  int k_src_off[2], v_src_row[2], v_src_col[2], k_lds[2], v_lds[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = STAGGER ? ((tid >> 8) * 512 + (tid & 255) + 256 * i) : (tid + 512 * i);
    const int krow = c >> 4, kcol = (c & 15) * 8;
    k_src_off[i] = krow * HD + kcol;
    k_lds[i] = krow * K_LD + kcol;
    const int vrow = c >> 3, vcol = (c & 7) * 8;
    v_src_row[i] = vrow;
    v_src_col[i] = vcol;
    v_lds[i] = vrow * V_LD + vcol;
  }
  const int64_t k_seq_stride = (int64_t)p.sk_pad * HD;     // per (seq, head)
  const int total_tiles = p.nchunks * tiles_per_chunk;

  u32x4_t kreg[2], vreg[2];
  auto tile_base = [&](int t, const bf16_t*& kb, const bf16_t*& vb) {
    const int chunk = t / tiles_per_chunk;
    const int tt = t - chunk * tiles_per_chunk;
    const int64_t off = (int64_t)chunk * p.chunk_stride + (int64_t)bh * k_seq_stride;
    kb = p.K + off + (int64_t)tt * KVBLK * HD;
    vb = p.Vt + off + (int64_t)tt * KVBLK;
  };
  auto load_k = [&](int t) {
    if (ABL == 5 || ABL == 6) return;
    const bf16_t *kb, *vb;
    tile_base(t, kb, vb);
#pragma unroll
    for (int i = 0; i < 2; ++i) kreg[i] = *reinterpret_cast<const u32x4_t*>(kb + k_src_off[i]);
  };
  auto load_v = [&](int t) {
    if (ABL == 5 || ABL == 6) return;
    const bf16_t *kb, *vb;
    tile_base(t, kb, vb);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      vreg[i] = *reinterpret_cast<const u32x4_t*>(vb + (int64_t)v_src_row[i] * p.sk_pad + v_src_col[i]);
  };
  auto store_k = [&](int buf) {
    if (ABL == 5 || ABL == 6) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4_t*>(&Ks[buf * K_TILE + k_lds[i]]) = kreg[i];
  };
  auto store_v = [&](int buf) {
    if (ABL == 5 || ABL == 6) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4_t*>(&Vs[buf * V_TILE + v_lds[i]]) = vreg[i];
  };

  f32x16_t o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY;   // running max, in log2 units (score * scale * log2 e)
  float l_run = 0.f;         // this half-lane's partial row sum
  const float c = p.scale * 1.4426950408889634f;

  const int k_frag = l31 * K_LD + hi * 8;
  const int v_frag = l31 * V_LD + hi * 8;
  f32x16_t s[2];

  // ---- phase 1: S^T = K Q^T.  The 16 K fragments are read 8 deep ahead of the MFMAs that
  // consume them (LDS latency ~100+ cycles vs 32 cycles per MFMA): 8 reads up front, then every
  // MFMA of the first key block re-fills its fragment slot with the second block's fragment.
  auto qk_phase = [&](int t) {
    const int buf = t & 1;
    const bf16_t* kp0 = Ks + buf * K_TILE + k_frag;
    const bf16_t* kp1 = kp0 + 32 * K_LD;
    bf16x8_t kf[8];
    if (ABL == 3) return;                      // ablation: no MFMA at all
    if (ABL == 4 || ABL == 5) {                // ablation: MFMAs fed from registers (no LDS reads)
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[0][r] = 0.f; s[1][r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[(ks + 1) & 7], qf[ks], s[0], 0, 0, 0);
        s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[(ks + 3) & 7], qf[ks], s[1], 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kf[ks] = *reinterpret_cast<const bf16x8_t*>(kp0 + ks * 16);
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[0][r] = 0.f; s[1][r] = 0.f; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s[0], 0, 0, 0);
      kf[ks] = *reinterpret_cast<const bf16x8_t*>(kp1 + ks * 16);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s[1], 0, 0, 0);
  };

  // ---- phase 2: online softmax (row = this lane's query), then O^T += V^T P^T -------
  auto softmax_pv_phase = [&](int t) {
    const int buf = t & 1;
    // V^T fragments of the first two 16-key steps are fetched now and land under the softmax
    const bf16_t* vp = Vs + buf * V_TILE + v_frag;
    bf16x8_t vf[8];
    if (ABL == 1 || ABL == 5) {                // ablation: no softmax arithmetic (P := S)
      bf16x8_t pq[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        u32x4_t w;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          w[e] = pack_bf2(s[kk >> 1][(kk & 1) * 8 + 2 * e], s[kk >> 1][(kk & 1) * 8 + 2 * e + 1]);
        pq[kk] = __builtin_bit_cast(bf16x8_t, w);
      }
      l_run = 1.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const bf16x8_t a = (ABL == 5) ? qf[(kk + d) & 7]
                                        : *reinterpret_cast<const bf16x8_t*>(vp + kk * 16 + d * 32 * V_LD);
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pq[kk], o[d], 0, 0, 0);
        }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int d = 0; d < 4; ++d)
        vf[kk * 4 + d] = *reinterpret_cast<const bf16x8_t*>(vp + kk * 16 + d * 32 * V_LD);
    __builtin_amdgcn_sched_barrier(0);
    {   // mask the padded keys of a chunk's last tile
      const int chunk = t / tiles_per_chunk;
      const int tt = t - chunk * tiles_per_chunk;
      const int valid = p.sk - tt * KVBLK;      // wave-uniform
      if (valid < KVBLK) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= valid) s[kb][r] = -INFINITY;
          }
      }
    }
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_tile = mx * c;
    bool rescale = true;
    if (DEFER > 0) rescale = !__all(m_tile - m_run <= (float)DEFER);
    if (rescale) {
      const float m_new = fmaxf(m_run, m_tile);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    }
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], c, -m_run));
        s[kb][r] = pv;
        rs += pv;
      }
    l_run += rs;
    // P^T fragments (B operand), straight from the S registers
    bf16x8_t pf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      u32x4_t w;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        w[e] = pack_bf2(s[kk >> 1][(kk & 1) * 8 + 2 * e], s[kk >> 1][(kk & 1) * 8 + 2 * e + 1]);
      pf[kk] = __builtin_bit_cast(bf16x8_t, w);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ABL == 3) {                            // ablation: softmax only, keep P alive
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) asm volatile("" ::"v"(pf[kk]));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(vf[i]));
      return;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kk * 4 + d], pf[kk], o[d], 0, 0, 0);
        vf[kk * 4 + d] = *reinterpret_cast<const bf16x8_t*>(vp + (kk + 2) * 16 + d * 32 * V_LD);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 2; kk < 4; ++kk)
#pragma unroll
      for (int d = 0; d < 4; ++d)
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[(kk - 2) * 4 + d], pf[kk], o[d], 0, 0, 0);
  };

  if (STAGGER) {
    // prologue: tile 0 complete in buffer 0 (each half-workgroup brings its half), K(1) in flight
    load_k(0);
    load_v(0);
    store_k(0);
    store_v(0);
    if (total_tiles > 1) load_k(1);
    if (late) __syncthreads();                  // waves 4-7 start one phase late
    for (int t = 0; t < total_tiles; ++t) {
      __syncthreads();                          // ---- phase 1 of tile t
      if (t + 1 < total_tiles) load_v(t + 1);
      qk_phase(t);
      if (t + 1 < total_tiles) store_k((t + 1) & 1);
      __syncthreads();                          // ---- phase 2 of tile t
      if (t + 2 < total_tiles) load_k(t + 2);
      softmax_pv_phase(t);
      if (t + 1 < total_tiles) store_v((t + 1) & 1);
    }
    if (!late) __syncthreads();                 // balance the barrier count
  } else {
    load_k(0);
    load_v(0);
    store_k(0);
    store_v(0);
    for (int t = 0; t < total_tiles; ++t) {
      __syncthreads();
      if (t + 1 < total_tiles) { load_k(t + 1); load_v(t + 1); }
      qk_phase(t);
      softmax_pv_phase(t);
      if (t + 1 < total_tiles) { store_k((t + 1) & 1); store_v((t + 1) & 1); }
    }
  }

  // ---- normalise and store O[q][head*128 + d] ---------------------------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
  if (q < p.sq) {
    bf16_t* op = p.O + ((int64_t)seq * p.sq + q) * p.ldo + head * HD + 4 * hi;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t w;
        w[0] = pack_bf2(o[d][4 * g] * inv, o[d][4 * g + 1] * inv);
        w[1] = pack_bf2(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
        *reinterpret_cast<u32x2_t*>(op + d * 32 + 8 * g) = w;
      }
  }
}

template <int DEFER, bool STAGGER, int ABL = 0>
int launch(const am_attn_args* a, void* stream) {
  static bool attr_set = false;
  if (!attr_set) {
    AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<DEFER, STAGGER, ABL>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
    attr_set = true;
  }
  const int tiles_per_chunk = ceil_div(a->sk, KVBLK);
  dim3 grid(ceil_div(a->sq, QBLK), a->nseq * a->heads);
  hipLaunchKernelGGL((attn_fwd_kernel<DEFER, STAGGER, ABL>), grid, dim3(512), SMEM_BYTES, (hipStream_t)stream, *a,
                     tiles_per_chunk);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

}  // namespace

// defer_log2: 0 or 8 = deferred-rescale threshold; add 100 to select the un-staggered
// (lockstep) schedule, kept for A/B measurements (tools/kernel_bench.py --variant).
extern "C" int am_attention_bf16(const am_attn_args* a, void* stream) {
  AM_CHECK(a != nullptr, "am_attention_bf16: null args");
  AM_CHECK(a->Q && a->K && a->Vt && a->O, "am_attention_bf16: null operand");
  AM_CHECK(a->nseq > 0 && a->heads > 0 && a->sq > 0 && a->sk > 0 && a->nchunks > 0,
           "am_attention_bf16: empty problem");
  AM_CHECK(a->sq_pad % QBLK == 0 && a->sq_pad >= a->sq, "am_attention_bf16: sq_pad=%d must be a multiple of %d and >= sq=%d",
           a->sq_pad, QBLK, a->sq);
  AM_CHECK(a->sk_pad % KVBLK == 0 && a->sk_pad >= a->sk, "am_attention_bf16: sk_pad=%d must be a multiple of %d and >= sk=%d",
           a->sk_pad, KVBLK, a->sk);
  AM_CHECK(a->ldo % 4 == 0 && a->ldo >= a->heads * HD, "am_attention_bf16: ldo=%d too small / misaligned", a->ldo);
  AM_CHECK(a->nchunks == 1 || a->chunk_stride >= (int64_t)a->nseq * a->heads * a->sk_pad * HD,
           "am_attention_bf16: chunk_stride too small");
  AM_CHECK(((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->Vt) % 16 == 0 && (uintptr_t)a->O % 8 == 0,
           "am_attention_bf16: operands misaligned");
  AM_CHECK((int64_t)a->nseq * a->heads <= 65535, "am_attention_bf16: nseq*heads=%lld exceeds grid.y",
           (long long)a->nseq * a->heads);
  switch (a->defer_log2) {
    case 0: return launch<0, true>(a, stream);
    case 8: return launch<8, true>(a, stream);
    case 100: return launch<0, false>(a, stream);
    case 108: return launch<8, false>(a, stream);
#ifdef AM_ATTN_ABLATIONS   // timing-only variants (wrong results by construction); tools/kernel_bench.py --ablate
    case 1001: return launch<8, true, 1>(a, stream);
    case 1003: return launch<8, true, 3>(a, stream);
    case 1004: return launch<8, true, 4>(a, stream);
    case 1005: return launch<8, true, 5>(a, stream);
    case 1006: return launch<8, true, 6>(a, stream);
    case 1101: return launch<8, false, 1>(a, stream);
    case 1103: return launch<8, false, 3>(a, stream);
    case 1104: return launch<8, false, 4>(a, stream);
    case 1105: return launch<8, false, 5>(a, stream);
    case 1106: return launch<8, false, 6>(a, stream);
#endif
    default: AM_FAIL(AM_ERR_INVALID, "am_attention_bf16: defer_log2 must be 0 or 8 (got %d)", a->defer_log2);
  }
}
