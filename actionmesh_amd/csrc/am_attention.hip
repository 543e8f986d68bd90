// Flash attention forward (non-causal, head_dim 128, bf16 MFMA, fp32 online
// softmax) for gfx950.  Replaces F.scaled_dot_product_attention at
// attention_processor.py:133-139 for both the inflated self-attention
// (seq = T*L, 68-99 % of the step's flops) and the per-frame cross-attention.
//
// This file: the entry point am_attention_bf16 (dispatch, split tail, two-pass row selection) and the 8-wave kernel.
// Long key streams (>= 16 tiles: the inflated self-attention) run their full 256-row query blocks on the 4-wave x
// 64-row kernel of am_attention64.hip, which shares every layout described here; the 8-wave kernel below serves
// short key streams (cross-attention), the short last query block (split over the key range) and A/B comparisons.
//
// Work decomposition: one workgroup = 8 waves = 256 query rows of one
// (sequence, head); each wave owns 32 query rows.  Keys/values stream through
// LDS in super-tiles of 128 keys (two 64-key sub-tiles), double-buffered
// (128 KiB), DMA'd straight from global memory (global_load_lds_dwordx4: no
// VGPR round trip, no ds_write pass), one workgroup barrier per super-tile.
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
// LDS image (DMA writes are lane-linear, so the bank swizzle is applied to the
// per-lane SOURCE address and mirrored on the read side):
//   K sub-tile   [64 keys][16 units of 16 B]: unit c of row r at position c ^ (r & 15)
//   V^T sub-tile [128 d  ][ 8 units of 16 B]: unit c of row r at position c ^ ((r>>1) & 7)
// Both give every 16-lane ds_read_b128 group 16 distinct 16-byte slots
// (SQ_LDS_BANK_CONFLICT = 0, profiles/).
//
// Why the loop looks the way it does (measurements on MI355X: tools/ubench,
// tools/kernel_bench.py --ablate, tools/attn_profile.py):
// time(full) ~= time(MFMA only) + time(everything else) - the two waves sharing a
// SIMD do not hide each other's softmax, and neither staggering the half-
// workgroups nor threading the softmax through the MFMA gaps recovers it (those
// schedules live in am_attention_variants.hip for A/B).  So this kernel minimises
// what is NOT an MFMA:
//   * Q is pre-multiplied by scale*log2(e) once (re-rounded to bf16): scores are
//     born in log2 units, no per-element multiply;
//   * the running max is subtracted two elements at a time (v_pk_add_f32);
//   * no key mask: padded K rows / V^T columns are zero (am_head_post), so a padded
//     key scores exactly 0 (harmless in the running max) and adds nothing to O;
//     the row sum of a chunk's partial last sub-tile is corrected once (tail);
//   * row max with v_max3 through inline asm (fmaxf on MFMA outputs makes hipcc
//     canonicalise every operand with an extra v_max), four independent chains;
//     cross-half exchange with v_permlane32_swap (no LDS round trip);
//   * online softmax with deferred rescale (threshold 2^8): O is rescaled only when
//     a row's max grows by more than 8 in log2 units;
//   * one barrier per 128 keys; LDS addresses are immediates (loop unrolled by 2).
// Per score element and lane: 1 exp + 0.5 sub + 0.5 add + 0.5 cvt + 0.5 max.
//
// The short last query block: seq = T*(N+1) is 256*k + a few rows for every
// reference shape (the +1 time token per frame); a 16-row block would cost a full
// extra round of workgroups (4112 blocks on 256 CUs = 16.06 rounds).  It is
// instead split 16 ways over the key range (SPLIT) and merged by a tiny kernel.
//
// Multi-GPU: K/V arrive as `nchunks` frame shards ([chunk][seq][head]...);
// softmax is permutation-invariant over keys, so chunks are simply
// concatenated tile streams, each with its own valid-key count.
#include <algorithm>

#include "am_common.h"

namespace {

constexpr int KVBLK = 64;          // keys per sub-tile
constexpr int HD = 128;
constexpr int SUB_B = KVBLK * HD * 2;          // one K or V^T sub-tile: 16 KiB
constexpr int PART_LD = HD + 4;                // floats per partial row: O[128], m, l, pad (16-B aligned rows)
constexpr int SPLIT_Z = 16;
// Geometry (template NW = waves per workgroup, NSUB = 64-key sub-tiles per barrier):
//   NW = 8, NSUB = 2: one 8-wave workgroup per CU, 256 query rows, 128 KiB LDS
//   NW = 4, NSUB = 1: two independent 4-wave workgroups per CU (128 query rows, 64 KiB LDS each): the two
//                     waves of a SIMD belong to different workgroups, share no barrier and drift into
//                     complementary phases (one in softmax while the other issues MFMAs)
template <int NW, int NSUB>
struct Geo {
  static constexpr int QBLK = NW * 32;                 // query rows per workgroup
  static constexpr int THREADS = NW * 64;
  static constexpr int UPT = 1024 / THREADS;           // 16-byte DMA units per thread per sub-tile operand
  static constexpr int BUF_B = 2 * NSUB * SUB_B;       // [K x NSUB][V x NSUB]
  static constexpr int SMEM = 2 * BUF_B;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// SPLIT: the workgroup handles only super-tiles [z*ns/Z, (z+1)*ns/Z) (z = blockIdx.z) of query block
// `qblk_base + blockIdx.x` and writes un-normalised fp32 (O, m, l) partials for attn_combine_kernel.
template <int DEFER, bool SPLIT, int NW, int NSUB>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel(am_attn_args p, int tiles_per_chunk, int qblk_base,
                                                             float* part) {
  using G = Geo<NW, NSUB>;
  constexpr int QBLK = G::QBLK, UPT = G::UPT, BUF_B = G::BUF_B;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.y;                          // sequence * heads + head
  const int head = bh % p.heads, seq = bh / p.heads;
  const int q0 = (qblk_base + blockIdx.x) * QBLK + wave * 32;
  const float c = p.scale * 1.4426950408889634f;

  // ---- Q fragments (B operand), pre-scaled to log2 units: Q[q0 + l31][ks*16 + hi*8 .. +8] ----
  bf16x8_t qf[8];
  {
    const bf16_t* qp = p.Q + ((int64_t)bh * p.sq_pad + q0 + l31) * HD + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(qp + ks * 16);
      u32x4_t sc;
#pragma unroll
      for (int e = 0; e < 4; ++e) sc[e] = pack_bf2(bflo(raw[e]) * c, bfhi(raw[e]) * c);
      qf[ks] = __builtin_bit_cast(bf16x8_t, sc);
    }
  }

  // ---- DMA descriptors: unit U = j*THREADS + wave*64 + lane (16 B each) of a 1024-unit sub-tile ----
  const int64_t k_seq_stride = (int64_t)p.sk_pad * HD;       // per (seq, head), K and V^T alike
  const bf16_t* k_lane[UPT];
  const bf16_t* v_lane[UPT];
  int u_byte[UPT];
#pragma unroll
  for (int j = 0; j < UPT; ++j) {
    u_byte[j] = (j * G::THREADS + wave * 64) * 16;           // wave-uniform LDS offset of the instruction
    const int U = j * G::THREADS + wave * 64 + lane;
    const int kr = U >> 4, kc = (U & 15) ^ (kr & 15);
    k_lane[j] = p.K + (int64_t)bh * k_seq_stride + kr * HD + kc * 8;
    const int vr = U >> 3, vc = (U & 7) ^ ((vr >> 1) & 7);
    v_lane[j] = p.Vt + (int64_t)bh * k_seq_stride + (int64_t)vr * p.sk_pad + vc * 8;
  }
  // key stream = nchunks x tiles_per_chunk sub-tiles, consumed as super-tiles of up to two
  // sub-tiles that never straddle a chunk
  const int supers_per_chunk = (tiles_per_chunk + NSUB - 1) / NSUB;
  const int all_supers = p.nchunks * supers_per_chunk;
  const int s_begin = SPLIT ? (int)((int64_t)blockIdx.z * all_supers / gridDim.z) : 0;
  const int s_end = SPLIT ? (int)((int64_t)(blockIdx.z + 1) * all_supers / gridDim.z) : all_supers;
  const int n_supers = s_end - s_begin;
  int d_tt = (s_begin % supers_per_chunk) * NSUB;                            // DMA cursor: sub-tile in chunk
  int64_t d_chunk = (int64_t)(s_begin / supers_per_chunk) * p.chunk_stride;  // element offset of its chunk
  auto dma_sub = [&](unsigned char* kdst, unsigned char* vdst, int tt) __attribute__((always_inline)) {
    const int64_t ko = d_chunk + (int64_t)tt * (KVBLK * HD);
    const int64_t vo = d_chunk + (int64_t)tt * KVBLK;
#pragma unroll
    for (int j = 0; j < UPT; ++j)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(k_lane[j] + ko), (lds_ptr_t)(kdst + u_byte[j]), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < UPT; ++j)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(v_lane[j] + vo), (lds_ptr_t)(vdst + u_byte[j]), 16, 0, 0);
  };
  auto dma_super = [&](int buf) __attribute__((always_inline)) {     // super-tile at the cursor -> LDS buffer `buf`; advance the cursor
    unsigned char* b = smem + buf * BUF_B;
    dma_sub(b, b + NSUB * SUB_B, d_tt);
    if (NSUB == 2 && d_tt + 1 < tiles_per_chunk) dma_sub(b + SUB_B, b + 3 * SUB_B, d_tt + 1);
    d_tt += NSUB;
    if (d_tt >= tiles_per_chunk) { d_tt = 0; d_chunk += p.chunk_stride; }
  };

  f32x16_t o[4], zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    zero16[r] = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d][r] = 0.f;
  }
  float m_run = 0.f;       // running max (log2 units); defined by the first sub-tile
  float l_run = 0.f;       // this half-lane's partial row sum
  bool first = true;

  // fragment read offsets (bytes) inside a sub-tile
  int k_off[8], v_off[4];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_off[ks] = l31 * 256 + (((ks * 2 + hi) ^ (l31 & 15)) << 4);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) v_off[kk] = l31 * 128 + (((kk * 2 + hi) ^ ((l31 >> 1) & 7)) << 4);
  int c_tt = (s_begin % supers_per_chunk) * NSUB;   // compute cursor: sub-tile in chunk

  auto max3 = [](float a, float b, float cc) __attribute__((always_inline)) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(cc));
    return d;
  };

  // ---- one 64-key sub-tile: S = K Q'^T, online softmax, O^T += V^T P^T ------------------------
  auto sub_tile = [&](const unsigned char* kp, const unsigned char* vp, int valid) __attribute__((always_inline)) {
    f32x16_t s[2];
    {   // the 16 K fragments are read 8 deep ahead of the MFMAs that consume them
      bf16x8_t kf[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) kf[ks] = *reinterpret_cast<const bf16x8_t*>(kp + k_off[ks]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        s[0] = AM_MFMA_32x32x16(kf[ks], qf[ks], ks == 0 ? zero16 : s[0]);
        kf[ks] = *reinterpret_cast<const bf16x8_t*>(kp + 32 * 256 + k_off[ks]);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        s[1] = AM_MFMA_32x32x16(kf[ks], qf[ks], ks == 0 ? zero16 : s[1]);
    }
    // V^T fragments of the first two 16-key steps: in flight under the softmax
    bf16x8_t vf[8];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int d = 0; d < 4; ++d)
        vf[kk * 4 + d] = *reinterpret_cast<const bf16x8_t*>(vp + d * 32 * 128 + v_off[kk]);
    // ---- row max (four independent v_max3 chains), relative to the running max ------------------
    // hipcc pads no hazards for inline asm: an MFMA result needs 12 wait states before a VALU read
    asm volatile("s_nop 15" : "+v"(s[0]), "+v"(s[1]));
    float mxa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mxa[i] = max3(s[0][i], s[1][i], s[0][i + 4]);
#pragma unroll
    for (int i = 0; i < 4; ++i) mxa[i] = max3(mxa[i], s[1][i + 4], s[0][i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) mxa[i] = max3(mxa[i], s[1][i + 8], s[0][i + 12]);
#pragma unroll
    for (int i = 0; i < 4; ++i) mxa[i] = max3(mxa[i], s[1][i + 12], mxa[i]);
    float mx = max3(mxa[0], mxa[1], max3(mxa[2], mxa[3], mxa[3]));
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = max3(__uint_as_float(sw[0]), __uint_as_float(sw[1]), __uint_as_float(sw[1]));
    }
    mx -= m_run;                                  // how far this sub-tile's max is above the running max
    if (first || !__all(mx <= (float)DEFER)) {   // rare after the first sub-tile (deferred rescale)
      const float delta = first ? mx : fmaxf(mx, 0.f);
      const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
      first = false;
      m_run += delta;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    }
    // ---- P = exp2(S - m_run), row sums, bf16 B-operand fragments -----------------------------------
    const f32x2_t m2 = {m_run, m_run};
    f32x2_t rsa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rsa[i] = f32x2_t{0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2_t x = f32x2_t{s[kb][r], s[kb][r + 1]} - m2;      // v_pk_add_f32
        const f32x2_t pp = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
        s[kb][r] = pp[0];
        s[kb][r + 1] = pp[1];
        rsa[(r >> 1) & 3] += pp;
      }
    {
      const f32x2_t rs = (rsa[0] + rsa[1]) + (rsa[2] + rsa[3]);
      l_run += rs[0] + rs[1];
    }
    if (valid < KVBLK) {   // partial last sub-tile of a chunk: remove the padded keys' exp2(0 - m_run)
      int cnt = 0;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g) cnt += min(4, max(0, kb * 32 + 8 * g + 4 * hi + 4 - valid));
      l_run -= (float)cnt * __builtin_amdgcn_exp2f(-m_run);
    }
    bf16x8_t pf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      u32x4_t w;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        w[e] = pack_bf2(s[kk >> 1][(kk & 1) * 8 + 2 * e], s[kk >> 1][(kk & 1) * 8 + 2 * e + 1]);
      pf[kk] = __builtin_bit_cast(bf16x8_t, w);
    }
    // ---- O^T += V^T P^T --------------------------------------------------------------------------------
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        o[d] = AM_MFMA_32x32x16(vf[kk * 4 + d], pf[kk], o[d]);
        vf[kk * 4 + d] = *reinterpret_cast<const bf16x8_t*>(vp + d * 32 * 128 + v_off[kk + 2]);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 2; kk < 4; ++kk)
#pragma unroll
      for (int d = 0; d < 4; ++d)
        o[d] = AM_MFMA_32x32x16(vf[(kk - 2) * 4 + d], pf[kk], o[d]);
  };

  auto super_tile = [&](int buf, bool more) __attribute__((always_inline)) {
    dma_drain_barrier();                      // super-tile landed; the other buffer is free
    if (more) dma_super(buf ^ 1);
    const unsigned char* b = smem + buf * BUF_B;
    const int valid = p.sk - c_tt * KVBLK;    // wave-uniform: valid keys from this sub-tile to the chunk end
    sub_tile(b, b + NSUB * SUB_B, valid);
    if (NSUB == 2 && valid > KVBLK) sub_tile(b + SUB_B, b + 3 * SUB_B, valid - KVBLK);
    c_tt += NSUB;
    if (c_tt >= tiles_per_chunk) c_tt = 0;
  };

  dma_super(0);
  for (int t = 0; t < n_supers; t += 2) {
    super_tile(0, t + 1 < n_supers);
    if (t + 1 < n_supers) super_tile(1, t + 2 < n_supers);
  }

  // ---- epilogue ------------------------------------------------------------------------------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const int q = q0 + l31;
  if (SPLIT) {
    if (q < p.sq) {
      float* pp = part + (((int64_t)bh * gridDim.z + blockIdx.z) * QBLK + (q - qblk_base * QBLK)) * PART_LD;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4_t*>(pp + d * 32 + 8 * g + 4 * hi) =
              f32x4_t{o[d][4 * g], o[d][4 * g + 1], o[d][4 * g + 2], o[d][4 * g + 3]};
      if (hi == 0) { pp[HD] = m_run; pp[HD + 1] = l_tot; }
    }
    return;
  }
  const float inv = 1.0f / l_tot;
  if (q < p.sq) {       // O[q][head*128 + d]
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

// ===========================================================================
// Cross-attention with the WHOLE key stream resident in LDS (round 6).  The per-frame cross-attention reads S = 257 context tokens
// (pipeline.py:665-667: DINOv2's class token + 256 patches; attention_processor.py:94-115) for L = N + 1 query rows per (frame, head):
// the tile-streaming kernel above gives every 128-row query block its own prologue (Q fragments, the first K / V^T tile in flight), five
// barriers and a fifth key tile that holds ONE key - at the headline shape 8448 workgroups of ~15 us for 5 us of MFMAs each.  Here ONE
// workgroup per (frame, head) loads K and V^T once - the full 64-key tiles (at most 4) plus a SHORT tail tile of at most 16 keys -
// and then walks all the query blocks of its sequence against LDS that nobody writes any more: no barrier and no DMA wait in the
// loop, the next block's Q rows are fetched while the current block computes, and the tail costs 8 + 4 MFMAs instead of 32 (its QK^T
// only multiplies key block 0, its P.V only the first 16-key step; padded keys score 0 and meet zero V^T columns, as everywhere).
// Same layouts, same LDS images, same per-tile arithmetic as attn_fwd_kernel (its sub_tile, copied: that kernel's tuning is not
// touched); the tail's row-sum correction counts 31 padded keys instead of 63, so results agree to rounding, not bit for bit.
// The output rows leave through a 2 KiB staging slice per wave (one 32-channel block at a time, XOR-swizzled 8-byte units): the MFMA
// layout holds one query row per lane, so direct stores were 8-byte pieces of 32 different rows per instruction - 512 line touches per
// block and wave, the floor of the first form of this kernel; staged, a store instruction writes 64 contiguous bytes of 16 rows.
// LDS: n_full x (16 + 16) KiB + 8 KiB (32 K rows of the tail) + 4 KiB (the tail's V^T, 16 key columns, compact) + 16 KiB (output staging)
// <= 156 KiB: one workgroup of 8 waves per CU.
// ===========================================================================
constexpr int RES_MAX_FULL = 4;
constexpr int RES_TAIL_MAX = 16;
constexpr int RES_VT_TAIL_B = 128 * 16 * 2;      // the tail's V^T, compact
constexpr int RES_STAGE_B = 8 * 2048;            // output staging, 2 KiB per wave
template <int DEFER>
__global__ __launch_bounds__(512, 2) void attn_resident_kernel(am_attn_args p, int n_full, int tail_valid, int n_qblk_total, int qblk_per_wg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;                                    // [n_full][SUB_B]
  unsigned char* Vs = smem + n_full * SUB_B;                   // [n_full][SUB_B]
  unsigned char* Kt = smem + 2 * n_full * SUB_B;               // tail: rows 0..31 of the K tile image (8 KiB)
  unsigned char* Vt_ = Kt + SUB_B / 2;                         // tail: V^T [128 d][16 key positions] compact, 32 B per row (4 KiB)
  unsigned char* Os = Vt_ + RES_VT_TAIL_B;                     // output staging: 2 KiB per wave

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.x;                                   // sequence * heads + head
  const int head = bh % p.heads, seq = bh / p.heads;
  const float c = p.scale * 1.4426950408889634f;
  // few (sequence, head) pairs (a rank's share of a sharded run): the query blocks of a pair are cut over gridDim.y workgroups, each
  // with its own copy of the key stream (loading it is ~3 us, a query block ~5)
  const int qb_begin = blockIdx.y * qblk_per_wg;
  const int qb_end = min(n_qblk_total, qb_begin + qblk_per_wg);
  if (qb_begin >= qb_end) return;

  // ---- the key stream -> LDS, once: unit U = j * 512 + tid of a 1024-unit sub-tile operand (same images as attn_fwd_kernel) ----
  {
    const int64_t k_seq_stride = (int64_t)p.sk_pad * HD;
    const bf16_t* kb = p.K + (int64_t)bh * k_seq_stride;
    const bf16_t* vb = p.Vt + (int64_t)bh * k_seq_stride;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int U = j * 512 + tid;
      const int kr = U >> 4, kc = (U & 15) ^ (kr & 15);
      const int vr = U >> 3, vc = (U & 7) ^ ((vr >> 1) & 7);
      const bf16_t* ks = kb + kr * HD + kc * 8;
      const bf16_t* vs = vb + (int64_t)vr * p.sk_pad + vc * 8;
      const int ub = (j * 512 + wave * 64) * 16;
      for (int t = 0; t < n_full; ++t) {
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ks + (int64_t)t * (KVBLK * HD)), (lds_ptr_t)(Ks + t * SUB_B + ub), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(vs + (int64_t)t * KVBLK), (lds_ptr_t)(Vs + t * SUB_B + ub), 16, 0, 0);
      }
      if (tail_valid > 0 && j == 0) {
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ks + (int64_t)n_full * (KVBLK * HD)), (lds_ptr_t)(Kt + ub), 16, 0, 0);
        if (wave < 4) {        // 256 units: unit tid = (d = tid >> 1, the 8 key positions 8 (tid & 1) .. of the tail's first 16)
          const bf16_t* vt = vb + (int64_t)(tid >> 1) * p.sk_pad + (int64_t)n_full * KVBLK + (tid & 1) * 8;
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)vt, (lds_ptr_t)(Vt_ + wave * 1024), 16, 0, 0);
        }
      }
    }
  }

  // fragment read offsets (bytes) inside a sub-tile
  int k_off[8], v_off[4];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_off[ks] = l31 * 256 + (((ks * 2 + hi) ^ (l31 & 15)) << 4);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) v_off[kk] = l31 * 128 + (((kk * 2 + hi) ^ ((l31 >> 1) & 7)) << 4);
  auto max3 = [](float a, float b, float cc) __attribute__((always_inline)) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(cc));
    return d;
  };
  f32x16_t zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
  // padded keys among the ones this lane holds of the tail's key block 0 (keys (r & 3) + 8 (r >> 2) + 4 hi, r = 0 .. 15)
  int tail_pad = 0;
#pragma unroll
  for (int g = 0; g < 4; ++g) tail_pad += min(4, max(0, 8 * g + 4 * hi + 4 - tail_valid));

  // raw Q rows of the first block (the scaling to log2 units happens at the top of the block's iteration)
  u32x4_t qraw[8];
  auto q_fetch = [&](int qb) __attribute__((always_inline)) {
    const bf16_t* qp = p.Q + ((int64_t)bh * p.sq_pad + qb * 256 + wave * 32 + l31) * HD + hi * 8;      // sq_pad % 256 == 0: always in bounds
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qraw[ks] = *reinterpret_cast<const u32x4_t*>(qp + ks * 16);
  };
  bf16x8_t qf[8];
  auto q_scale = [&]() __attribute__((always_inline)) {        // raw rows -> B-operand fragments in log2 units
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      u32x4_t sc;
#pragma unroll
      for (int e = 0; e < 4; ++e) sc[e] = pack_bf2(bflo(qraw[ks][e]) * c, bfhi(qraw[ks][e]) * c);
      qf[ks] = __builtin_bit_cast(bf16x8_t, sc);
    }
  };
  q_fetch(qb_begin);
  dma_drain_barrier();                                         // the key stream has landed and is visible to every wave (and so have the Q rows)
  q_scale();
  if (qb_begin + 1 < qb_end) q_fetch(qb_begin + 1);

  // Software pipeline over the query blocks: while block qb computes, the raw rows of block qb + 1 are in flight; they are scaled into
  // `qf` behind block qb's last MFMA and the rows of block qb + 2 are requested BEFORE block qb's output stores are issued - so the
  // wait in front of q_scale() covers loads and stores that are a whole block old, never the stores just issued (round 6: with the
  // scaling at the loop top every block waited for its predecessor's 16 stores to complete, ~10 us per block for 0.5 us of MFMAs).
  for (int qb = qb_begin; qb < qb_end; ++qb) {
    f32x16_t o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = 0.f, l_run = 0.f;
    bool first = true;

    // the deferred re-base of the online softmax (attn_fwd_kernel): mx = this tile's row max
    auto rebase = [&](float mx) __attribute__((always_inline)) {
      mx -= m_run;
      if (first || !__all(mx <= (float)DEFER)) {
        const float delta = first ? mx : fmaxf(mx, 0.f);
        const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
        first = false;
        m_run += delta;
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      }
    };
    // ---- full 64-key tiles: attn_fwd_kernel's sub_tile, operands from the resident images ----
    for (int t = 0; t < n_full; ++t) {
      const unsigned char* kp = Ks + t * SUB_B;
      const unsigned char* vp = Vs + t * SUB_B;
      f32x16_t s[2];
      {
        bf16x8_t kf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) kf[ks] = *reinterpret_cast<const bf16x8_t*>(kp + k_off[ks]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          s[0] = AM_MFMA_32x32x16(kf[ks], qf[ks], ks == 0 ? zero16 : s[0]);
          kf[ks] = *reinterpret_cast<const bf16x8_t*>(kp + 32 * 256 + k_off[ks]);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) s[1] = AM_MFMA_32x32x16(kf[ks], qf[ks], ks == 0 ? zero16 : s[1]);
      }
      bf16x8_t vf[8];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int d = 0; d < 4; ++d) vf[kk * 4 + d] = *reinterpret_cast<const bf16x8_t*>(vp + d * 32 * 128 + v_off[kk]);
      asm volatile("s_nop 15" : "+v"(s[0]), "+v"(s[1]));        // MFMA results landed before the VALU reads them
      float mxa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) mxa[i] = max3(s[0][i], s[1][i], s[0][i + 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i) mxa[i] = max3(mxa[i], s[1][i + 4], s[0][i + 8]);
#pragma unroll
      for (int i = 0; i < 4; ++i) mxa[i] = max3(mxa[i], s[1][i + 8], s[0][i + 12]);
#pragma unroll
      for (int i = 0; i < 4; ++i) mxa[i] = max3(mxa[i], s[1][i + 12], mxa[i]);
      float mx = max3(mxa[0], mxa[1], max3(mxa[2], mxa[3], mxa[3]));
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = max3(__uint_as_float(sw[0]), __uint_as_float(sw[1]), __uint_as_float(sw[1]));
      }
      rebase(mx);
      const f32x2_t m2 = {m_run, m_run};
      f32x2_t rsa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) rsa[i] = f32x2_t{0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2_t x = f32x2_t{s[kb][r], s[kb][r + 1]} - m2;
          const f32x2_t pp = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
          s[kb][r] = pp[0];
          s[kb][r + 1] = pp[1];
          rsa[(r >> 1) & 3] += pp;
        }
      {
        const f32x2_t rs = (rsa[0] + rsa[1]) + (rsa[2] + rsa[3]);
        l_run += rs[0] + rs[1];
      }
      bf16x8_t pf[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        u32x4_t w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = pack_bf2(s[kk >> 1][(kk & 1) * 8 + 2 * e], s[kk >> 1][(kk & 1) * 8 + 2 * e + 1]);
        pf[kk] = __builtin_bit_cast(bf16x8_t, w);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          o[d] = AM_MFMA_32x32x16(vf[kk * 4 + d], pf[kk], o[d]);
          vf[kk * 4 + d] = *reinterpret_cast<const bf16x8_t*>(vp + d * 32 * 128 + v_off[kk + 2]);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 2; kk < 4; ++kk)
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = AM_MFMA_32x32x16(vf[(kk - 2) * 4 + d], pf[kk], o[d]);
    }
    // ---- the short tail: key block 0 of its tile only (keys 0 .. 31, of which tail_valid <= 16 are real), one 16-key P.V step ----
    if (tail_valid > 0) {
      f32x16_t s0;
      {
        bf16x8_t kf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) kf[ks] = *reinterpret_cast<const bf16x8_t*>(Kt + k_off[ks]);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) s0 = AM_MFMA_32x32x16(kf[ks], qf[ks], ks == 0 ? zero16 : s0);
      }
      bf16x8_t vf[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) vf[d] = *reinterpret_cast<const bf16x8_t*>(Vt_ + ((d * 32 + l31) * 2 + hi) * 16);
      asm volatile("s_nop 15" : "+v"(s0));
      float mxa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) mxa[i] = max3(s0[i], s0[i + 4], s0[i + 8]);
      float mx = max3(max3(mxa[0], mxa[1], mxa[2]), mxa[3], max3(s0[12], s0[13], max3(s0[14], s0[15], s0[15])));
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = max3(__uint_as_float(sw[0]), __uint_as_float(sw[1]), __uint_as_float(sw[1]));
      }
      rebase(mx);
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(s0[r] - m_run);
        rs += s0[r];
      }
      l_run += rs - (float)tail_pad * __builtin_amdgcn_exp2f(-m_run);      // the padded keys scored exactly 0
      u32x4_t w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = pack_bf2(s0[2 * e], s0[2 * e + 1]);
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, w);                  // k-slots of the first 16-key step (keys 0 .. 15 in perm16 order)
#pragma unroll
      for (int d = 0; d < 4; ++d) o[d] = AM_MFMA_32x32x16(vf[d], pf, o[d]);
    }
    // ---- the next block's Q (its loads are a block old), the request for the one after, then this block's output ----
    if (qb + 1 < qb_end) q_scale();
    if (qb + 2 < qb_end) q_fetch(qb + 2);
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    // staged through this wave's 2 KiB slice, one 32-channel block at a time: row r = 64 B = 8 units of 8 B, unit u at u ^ ((r >> 2) & 7)
    // (write: the lane's row l31, units 2 g + hi - conflict free; read-back: lane = (row lane >> 2, 16-byte chunk lane & 3)); the
    // wave's own LDS operations execute in order, the asm statements keep the compiler from re-ordering them
    unsigned char* st = Os + wave * 2048;
    const int sw_w = (l31 >> 2) & 7;
    const int rr = lane >> 2, rc = lane & 3;                   // read-back: rows rr and rr + 16, chunk rc
    const int64_t row0 = (int64_t)seq * p.sq + qb * 256 + wave * 32;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t w;
        w[0] = pack_bf2(o[d][4 * g] * inv, o[d][4 * g + 1] * inv);
        w[1] = pack_bf2(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
        *reinterpret_cast<u32x2_t*>(st + l31 * 64 + (((2 * g + hi) ^ sw_w) << 3)) = w;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = rr + 16 * h;
        const int k = (r >> 2) & 7;
        u32x4_t v = *reinterpret_cast<const u32x4_t*>(st + r * 64 + ((rc ^ (k >> 1)) << 4));
        if (k & 1) v = u32x4_t{v[2], v[3], v[0], v[1]};
        if (qb * 256 + wave * 32 + r < p.sq)
          *reinterpret_cast<u32x4_t*>(p.O + (row0 + r) * p.ldo + head * HD + d * 32 + rc * 8) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

// ===========================================================================
// Balanced two-phase schedule.  The two half-workgroups (waves 0-3 / 4-7; wave i and wave i+4 share
// SIMD i) run exactly one phase apart - waves 4-7 take one extra barrier before the loop, waves 0-3 one
// after it - and a 64-key tile is cut into two phases of EQUAL weight, separated by barriers:
//   phase 1: S = K Q'^T (16 MFMA), then the first half of the softmax (row max, rescale, exp of key block 0)
//   phase 2: second half of the softmax (exp of key block 1, row sums, bf16 P), then O += V^T P^T (16 MFMA)
// so on every SIMD the wave in phase 1 issues its MFMAs while its partner is in the VALU half of phase 2,
// and vice versa: MFMA and VALU work of the pair are interleaved by construction instead of colliding
// (the lockstep loop leaves the matrix pipe idle while both waves are in their softmax).  An earlier
// staggered variant with unbalanced phases (QK^T | softmax + P.V) did not help: the short phase just waited
// at the barrier.  Each half-workgroup DMAs its own half of every K / V^T tile; with global phase g = the
// interval after barrier #g, waves 0-3 run phase 1/2 of tile t at g = 2t / 2t+1, waves 4-7 at 2t+1 / 2t+2:
//   K(t+1) -> the buffer K(t-1) left at g = 2t-1; issued at the start of phase 1(t), drained before the
//             issuing half's next barrier, first read at g = 2t+2;
//   V(t+1) -> the buffer V(t-1) left at g = 2t;   issued at the start of phase 2(t), first read at g = 2t+3.
// ===========================================================================
template <int DEFER>
__global__ __launch_bounds__(512, 2) void attn_fwd_balanced_kernel(am_attn_args p, int tiles_per_chunk) {
  constexpr int QBLK = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;                   // [2][16 KiB]
  unsigned char* Vs = smem + 2 * SUB_B;       // [2][16 KiB]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.y;
  const int head = bh % p.heads, seq = bh / p.heads;
  const int q0 = blockIdx.x * QBLK + wave * 32;
  const float c = p.scale * 1.4426950408889634f;
  const bool late = wave >= 4;

  bf16x8_t qf[8];
  {
    const bf16_t* qp = p.Q + ((int64_t)bh * p.sq_pad + q0 + l31) * HD + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(qp + ks * 16);
      u32x4_t sc;
#pragma unroll
      for (int e = 0; e < 4; ++e) sc[e] = pack_bf2(bflo(raw[e]) * c, bfhi(raw[e]) * c);
      qf[ks] = __builtin_bit_cast(bf16x8_t, sc);
    }
  }
  // this half-workgroup's half of every tile: units U = half*512 + j*256 + (wave&3)*64 + lane
  const int64_t k_seq_stride = (int64_t)p.sk_pad * HD;
  const bf16_t* k_lane[2];
  const bf16_t* v_lane[2];
  int u_byte[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ub = (wave >> 2) * 512 + j * 256 + (wave & 3) * 64;
    u_byte[j] = ub * 16;
    const int U = ub + lane;
    const int kr = U >> 4, kc = (U & 15) ^ (kr & 15);
    k_lane[j] = p.K + (int64_t)bh * k_seq_stride + kr * HD + kc * 8;
    const int vr = U >> 3, vc = (U & 7) ^ ((vr >> 1) & 7);
    v_lane[j] = p.Vt + (int64_t)bh * k_seq_stride + (int64_t)vr * p.sk_pad + vc * 8;
  }
  const int total_tiles = p.nchunks * tiles_per_chunk;
  int dk_tt = 0, dv_tt = 0;
  int64_t dk_chunk = 0, dv_chunk = 0;
  auto dma_k = [&](int buf) __attribute__((always_inline)) {
    const int64_t ko = dk_chunk + (int64_t)dk_tt * (KVBLK * HD);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(k_lane[j] + ko), (lds_ptr_t)(Ks + buf * SUB_B + u_byte[j]), 16, 0, 0);
    if (++dk_tt == tiles_per_chunk) { dk_tt = 0; dk_chunk += p.chunk_stride; }
  };
  auto dma_v = [&](int buf) __attribute__((always_inline)) {
    const int64_t vo = dv_chunk + (int64_t)dv_tt * KVBLK;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(v_lane[j] + vo), (lds_ptr_t)(Vs + buf * SUB_B + u_byte[j]), 16, 0, 0);
    if (++dv_tt == tiles_per_chunk) { dv_tt = 0; dv_chunk += p.chunk_stride; }
  };

  f32x16_t o[4], zero16, s[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    zero16[r] = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d][r] = 0.f;
  }
  float m_run = 0.f, l_run = 0.f;
  bool first = true;
  int k_off[8], v_off[4];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_off[ks] = l31 * 256 + (((ks * 2 + hi) ^ (l31 & 15)) << 4);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) v_off[kk] = l31 * 128 + (((kk * 2 + hi) ^ ((l31 >> 1) & 7)) << 4);
  int c_tt = 0;
  f32x2_t rsa[4];

  auto max3 = [](float a, float b, float cc) __attribute__((always_inline)) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(cc));
    return d;
  };
  auto exp_block = [&](f32x16_t& sb) __attribute__((always_inline)) {   // P = exp2(S - m_run) for one key block
    const f32x2_t m2 = {m_run, m_run};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2_t x = f32x2_t{sb[r], sb[r + 1]} - m2;
      const f32x2_t pp = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
      sb[r] = pp[0];
      sb[r + 1] = pp[1];
      rsa[(r >> 1) & 3] += pp;
    }
  };

  auto phase1 = [&](int buf, bool more) __attribute__((always_inline)) {
    dma_drain_barrier();
    if (more) dma_k(buf ^ 1);
    const unsigned char* kp = Ks + buf * SUB_B;
    {
      bf16x8_t kf[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) kf[ks] = *reinterpret_cast<const bf16x8_t*>(kp + k_off[ks]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        s[0] = AM_MFMA_32x32x16(kf[ks], qf[ks], ks == 0 ? zero16 : s[0]);
        kf[ks] = *reinterpret_cast<const bf16x8_t*>(kp + 32 * 256 + k_off[ks]);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        s[1] = AM_MFMA_32x32x16(kf[ks], qf[ks], ks == 0 ? zero16 : s[1]);
    }
    // ---- softmax, first half: row max, rescale, exp of key block 0 ------------------------------
    asm volatile("s_nop 15" : "+v"(s[0]), "+v"(s[1]));     // MFMA result -> inline-asm VALU read hazard
    float mxa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mxa[i] = max3(s[0][i], s[1][i], s[0][i + 4]);
#pragma unroll
    for (int i = 0; i < 4; ++i) mxa[i] = max3(mxa[i], s[1][i + 4], s[0][i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) mxa[i] = max3(mxa[i], s[1][i + 8], s[0][i + 12]);
#pragma unroll
    for (int i = 0; i < 4; ++i) mxa[i] = max3(mxa[i], s[1][i + 12], mxa[i]);
    float mx = max3(mxa[0], mxa[1], max3(mxa[2], mxa[3], mxa[3]));
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = max3(__uint_as_float(sw[0]), __uint_as_float(sw[1]), __uint_as_float(sw[1]));
    }
    mx -= m_run;
    if (first || !__all(mx <= (float)DEFER)) {
      const float delta = first ? mx : fmaxf(mx, 0.f);
      const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
      first = false;
      m_run += delta;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) rsa[i] = f32x2_t{0.f, 0.f};
    exp_block(s[0]);
  };

  auto phase2 = [&](int buf, bool more) __attribute__((always_inline)) {
    dma_drain_barrier();
    if (more) dma_v(buf ^ 1);
    const unsigned char* vp = Vs + buf * SUB_B;
    bf16x8_t vf[8];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int d = 0; d < 4; ++d)
        vf[kk * 4 + d] = *reinterpret_cast<const bf16x8_t*>(vp + d * 32 * 128 + v_off[kk]);
    // ---- softmax, second half: exp of key block 1, row sums, bf16 P ---------------------------------
    exp_block(s[1]);
    {
      const f32x2_t rs = (rsa[0] + rsa[1]) + (rsa[2] + rsa[3]);
      l_run += rs[0] + rs[1];
    }
    const int valid = p.sk - c_tt * KVBLK;
    if (++c_tt == tiles_per_chunk) c_tt = 0;
    if (valid < KVBLK) {
      int cnt = 0;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g) cnt += min(4, max(0, kb * 32 + 8 * g + 4 * hi + 4 - valid));
      l_run -= (float)cnt * __builtin_amdgcn_exp2f(-m_run);
    }
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
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        o[d] = AM_MFMA_32x32x16(vf[kk * 4 + d], pf[kk], o[d]);
        vf[kk * 4 + d] = *reinterpret_cast<const bf16x8_t*>(vp + d * 32 * 128 + v_off[kk + 2]);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 2; kk < 4; ++kk)
#pragma unroll
      for (int d = 0; d < 4; ++d)
        o[d] = AM_MFMA_32x32x16(vf[(kk - 2) * 4 + d], pf[kk], o[d]);
  };

  dma_k(0);
  dma_v(0);
  if (late) dma_drain_barrier();                 // waves 4-7 start one phase late
  for (int t = 0; t < total_tiles; t += 2) {
    phase1(0, t + 1 < total_tiles);
    phase2(0, t + 1 < total_tiles);
    if (t + 1 < total_tiles) {
      phase1(1, t + 2 < total_tiles);
      phase2(1, t + 2 < total_tiles);
    }
  }
  if (!late) __syncthreads();                    // balance the barrier count

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

// merge the Z partials of the split query block: O = sum_z 2^(m_z - m) O_z / sum_z 2^(m_z - m) l_z
__global__ __launch_bounds__(128) void attn_combine_kernel(am_attn_args p, const float* __restrict__ part, int Z,
                                                           int qblk_base, int rows, int QBLK) {
  const int bh = blockIdx.y, row = blockIdx.x;          // one 128-thread block per (sequence*head, row)
  if (row >= rows) return;
  const int d = threadIdx.x;
  const float* base = part + ((int64_t)bh * Z * QBLK + row) * PART_LD;
  float m = -INFINITY;
  for (int z = 0; z < Z; ++z) m = fmaxf(m, base[(int64_t)z * QBLK * PART_LD + HD]);
  float acc = 0.f, l = 0.f;
  for (int z = 0; z < Z; ++z) {
    const float* pz = base + (int64_t)z * QBLK * PART_LD;
    const float w = __builtin_amdgcn_exp2f(pz[HD] - m);
    acc += w * pz[d];
    l += w * pz[HD + 1];
  }
  const int head = bh % p.heads, seq = bh / p.heads;
  p.O[((int64_t)seq * p.sq + qblk_base * QBLK + row) * p.ldo + head * HD + d] = f2bf(acc / l);
}

}  // namespace
// merge of the split last block's partials, shared with the fp8 kernel (am_attention_fp8.hip): same partial layout
int am_attention_combine_launch(const am_attn_args* a, const float* part, int Z, int qblk_base, int rows, void* stream) {
  hipLaunchKernelGGL(attn_combine_kernel, dim3(rows, a->nseq * a->heads), dim3(128), 0, (hipStream_t)stream, *a, part, Z, qblk_base, rows, 256);
  AM_HIP(hipGetLastError());
  return AM_OK;
}
int am_attention64_main(const am_attn_args* a, int tiles_per_chunk, int nblk_main, int defer, void* stream);  // am_attention64.hip
namespace {

// MAIN: 0 = this file's 8-wave kernel, 1 = balanced two-phase variant, 2 = the 4x64 kernel of am_attention64.hip,
// 3 = product dispatch: 4x64 for long key streams (>= 16 tiles: its pipeline needs a few tiles to fill and its
// per-workgroup prologue is heavier - cross-attention with 257 keys runs 15 % faster on the 8-wave kernel).
// The split tail always runs the 8-wave kernel (same 256-row query blocks).
template <int DEFER, int NW, int NSUB, int MAIN = 0>
int launch(const am_attn_args* a, void* stream) {
  constexpr bool BALANCED = MAIN == 1;
  const bool use64 = MAIN == 2 || (MAIN == 3 && ceil_div(a->sk, KVBLK) * a->nchunks >= 16);
  using G = Geo<NW, NSUB>;
  // The split tail (a short last query block cut SPLIT_Z ways over the key range) always runs the 4-wave / 128-row geometry: at most
  // G::QBLK / 2 <= 128 rows are valid, so one 128-row block holds them, half as many waves walk keys for padding rows, and two
  // workgroups share a CU (round 4; profiles/r04z_split_tail.txt - before, it ran in the caller's 256-row geometry: 143 + 9 us).
  using GS = Geo<4, 1>;
  constexpr int RATIO = G::QBLK / GS::QBLK;
  AM_ONCE_PER_DEVICE({
    if (BALANCED)
      AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_balanced_kernel<DEFER>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 4 * SUB_B));
    AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<DEFER, false, NW, NSUB>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM));
    AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<DEFER, true, 4, 1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, GS::SMEM));
  });
  const int tiles_per_chunk = ceil_div(a->sk, KVBLK);
  const int all_supers = ceil_div(tiles_per_chunk, NSUB) * a->nchunks;
  const int nblk = ceil_div(a->sq, G::QBLK);
  const int bh = a->nseq * a->heads;
  const int tail_rows = a->sq - (nblk - 1) * G::QBLK;
  // split the short last query block over the key range when it would otherwise add a round
  static float* part = nullptr;       // library-owned scratch, grown on demand
  static size_t part_elems = 0;
  // (32 cuts instead of 16 change nothing - measured: the pass re-reads every K / V^T byte of the launch for 16 query rows, 537 MB at the
  // headline shape = 107 us at 5 TB/s, and sits at 116 us: it is at its HBM floor)
  const int Z = SPLIT_Z;
  const size_t need = (size_t)bh * Z * GS::QBLK * PART_LD;
  // a->rows: 0 = every query block; 1 = only the blocks the 4x64 kernel takes ("main": all but a short last block);
  // 2 = only what rows = 1 leaves out.  The main/rest boundary depends on the query geometry alone, so the calls of a
  // two-pass sequence (different nchunks) agree on it.
  const bool tail_geom = nblk >= 9 && tail_rows <= G::QBLK / 2;
  const bool can_split = (int64_t)tiles_per_chunk * a->nchunks >= 2 * SPLIT_Z && all_supers >= 2 * SPLIT_Z && need * sizeof(float) <= (256u << 20);
  const bool split = tail_geom && can_split;
  if (a->rows == 1) {
    AM_CHECK(use64, "am_attention_bf16: rows = 1 / two-pass needs the 4x64 kernel (>= 16 key tiles in the chunks walked)");
    AM_TRY(am_attention64_main(a, tiles_per_chunk, tail_geom ? nblk - 1 : nblk, DEFER, stream));
    AM_HIP(hipGetLastError());
    return AM_OK;
  }
  if (a->rows == 2 && !tail_geom) return AM_OK;
  if (split && part_elems < need) {
    if (part) AM_HIP(hipFree(part));
    part = nullptr; part_elems = 0;
    ++g_am_scratch_generation;
    AM_HIP(hipMalloc(reinterpret_cast<void**>(&part), need * sizeof(float)));
    part_elems = need;
  }
  hipStream_t st = (hipStream_t)stream;
  if (a->rows == 2) {
    if (!split)          // the short last block alone, un-split (key stream too short to cut 16 ways)
      hipLaunchKernelGGL((attn_fwd_kernel<DEFER, false, NW, NSUB>), dim3(1, bh), dim3(G::THREADS), G::SMEM, st, *a,
                         tiles_per_chunk, nblk - 1, (float*)nullptr);
  } else if (use64)
    AM_TRY(am_attention64_main(a, tiles_per_chunk, split ? nblk - 1 : nblk, DEFER, stream));
  else if (BALANCED)
    hipLaunchKernelGGL((attn_fwd_balanced_kernel<DEFER>), dim3(split ? nblk - 1 : nblk, bh), dim3(512), 4 * SUB_B, st, *a,
                       tiles_per_chunk);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<DEFER, false, NW, NSUB>), dim3(split ? nblk - 1 : nblk, bh), dim3(G::THREADS), G::SMEM,
                       st, *a, tiles_per_chunk, 0, (float*)nullptr);
  if (split) {
    hipLaunchKernelGGL((attn_fwd_kernel<DEFER, true, 4, 1>), dim3(1, bh, Z), dim3(GS::THREADS), GS::SMEM, st, *a,
                       tiles_per_chunk, (nblk - 1) * RATIO, part);
    hipLaunchKernelGGL(attn_combine_kernel, dim3(tail_rows, bh), dim3(128), 0, st, *a, part, Z, (nblk - 1) * RATIO, tail_rows,
                       GS::QBLK);
  }
  AM_HIP(hipGetLastError());
  return AM_OK;
}

}  // namespace

// The resident-key-stream kernel (attn_resident_kernel) takes the launches it was built for: one key chunk of at most 4 full tiles plus a
// tail of at most 16 keys (the cross-attention's 257 context tokens), at least 4 query blocks per (sequence, head) to walk and at least
// two rounds' worth of query blocks in the launch; everything else - the encoders' 257-row sequences, tests with odd shapes - stays on the
// tile-streaming kernel.  ACTIONMESH_AMD_XATTN_RESIDENT=0 turns it off (same-box A/B).
static bool resident_eligible(const am_attn_args* a) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("ACTIONMESH_AMD_XATTN_RESIDENT"); on = (e && e[0] == '0') ? 0 : 1; }
  const int n_full = a->sk / KVBLK, tail = a->sk % KVBLK;
  return on == 1 && a->nchunks == 1 && a->state_mode == 0 && a->chunk_total == 0 && n_full >= 1 && n_full <= RES_MAX_FULL && tail <= RES_TAIL_MAX &&
         a->sk_pad >= (n_full + (tail > 0 ? 1 : 0)) * KVBLK && ceil_div(a->sq, 256) >= 4 && (int64_t)a->nseq * a->heads * ceil_div(a->sq, 256) >= 512 &&
         (int64_t)a->nseq * a->heads <= 65535 && a->ldo % 8 == 0 && (uintptr_t)a->O % 16 == 0;      // 16-byte output stores
}
template <int DEFER>
static int launch_resident(const am_attn_args* a, void* stream) {
  const int n_full = a->sk / KVBLK, tail = a->sk % KVBLK;
  const int smem_bytes = 2 * n_full * SUB_B + SUB_B / 2 + RES_VT_TAIL_B + RES_STAGE_B;
  AM_ONCE_PER_DEVICE({
    AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_resident_kernel<DEFER>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               2 * RES_MAX_FULL * SUB_B + SUB_B / 2 + RES_VT_TAIL_B + RES_STAGE_B));
  });
  // one workgroup per CU (the LDS image): aim at a grid of about one round of the chip - whole (sequence, head) pairs when there are
  // enough of them, otherwise each pair's query blocks cut over `qsplit` workgroups of at least two blocks
  const int bh = a->nseq * a->heads, nblk = ceil_div(a->sq, 256);
  int qsplit = 1;
  if (bh < 192) qsplit = std::max(1, std::min(nblk / 2, 256 / bh));
  const int per = ceil_div(nblk, qsplit);
  hipLaunchKernelGGL((attn_resident_kernel<DEFER>), dim3(bh, ceil_div(nblk, per)), dim3(512), smem_bytes, (hipStream_t)stream, *a, n_full, tail,
                     nblk, per);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

#ifdef AM_ATTN_ABLATIONS
int am_attention_variant(const am_attn_args* a, void* stream);   // am_attention_variants.hip
#endif

// defer_log2: 0 = exact online softmax, re-based whenever a row max grows; 8 = product: deferred re-base (threshold 2^8)
// on the 8-wave kernel, the lazy re-base of am_attention64.hip (+ its exact fallback launch) on the 4x64 kernel.
// Other codes force one kernel for A/B runs and parity tests: +60 the 4x64 kernel, +90 the 8-wave kernel,
// +50 / +70 its geometry / schedule variants; 28 = the 4x64 kernel with the exact deferred re-base on its own.
// Builds with -DAM_ATTN_ABLATIONS also accept the experimental schedules of am_attention_variants.hip
// (defer_log2 >= 100, used by tools/kernel_bench.py for A/B measurements).
extern "C" int am_attention_bf16(const am_attn_args* a, void* stream) {
  AM_CHECK(a != nullptr, "am_attention_bf16: null args");
  AM_CHECK(a->Q && a->K && a->Vt && a->O, "am_attention_bf16: null operand");
  AM_CHECK(a->nseq > 0 && a->heads > 0 && a->sq > 0 && a->sk > 0 && a->nchunks > 0,
           "am_attention_bf16: empty problem");
  AM_CHECK(a->sq_pad % 256 == 0 && a->sq_pad >= a->sq, "am_attention_bf16: sq_pad=%d must be a multiple of 256 and >= sq=%d",
           a->sq_pad, a->sq);
  AM_CHECK(a->sk_pad % KVBLK == 0 && a->sk_pad >= a->sk, "am_attention_bf16: sk_pad=%d must be a multiple of %d and >= sk=%d",
           a->sk_pad, KVBLK, a->sk);
  AM_CHECK(a->ldo % 4 == 0 && a->ldo >= a->heads * HD, "am_attention_bf16: ldo=%d too small / misaligned", a->ldo);
  AM_CHECK(a->nchunks == 1 || a->chunk_stride >= (int64_t)a->nseq * a->heads * a->sk_pad * HD,
           "am_attention_bf16: chunk_stride too small");
  AM_CHECK(((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->Vt) % 16 == 0 && (uintptr_t)a->O % 8 == 0,
           "am_attention_bf16: operands misaligned");
  AM_CHECK((int64_t)a->nseq * a->heads <= 65535, "am_attention_bf16: nseq*heads=%lld exceeds grid.y",
           (long long)a->nseq * a->heads);
  AM_CHECK(a->rows >= 0 && a->rows <= 2 && a->state_mode >= 0 && a->state_mode <= 2, "am_attention_bf16: bad rows / state_mode");
  AM_CHECK(a->state_mode == 0 || (a->rows == 1 && a->state != nullptr && (uintptr_t)a->state % 16 == 0),
           "am_attention_bf16: state_mode needs rows = 1 and a 16-byte aligned state buffer");
  AM_CHECK(a->chunk_total == 0 || (a->rows == 1 && a->chunk_total > 0 && a->chunk_first >= 0 && a->chunk_first < a->chunk_total &&
                                   a->nchunks <= a->chunk_total),
           "am_attention_bf16: chunk_first/chunk_total need rows = 1, 0 <= first < total, nchunks <= total");
  AM_CHECK(a->rows != 1 || a->defer_log2 == 0 || a->defer_log2 == 8 || a->defer_log2 == 60 || a->defer_log2 == 68 || a->defer_log2 == 28,
           "am_attention_bf16: rows = 1 runs on the product dispatch only");
  // Short key streams (< 16 tiles: the cross-attention's 257 context tokens, the encoders' sequences) run the 8-wave kernel's
  // code in its OTHER geometry - two independent 4-wave workgroups per CU, 64 KiB of LDS each: a workgroup that walks 5 key tiles is
  // mostly prologue (Q fragments, the first K / V^T tiles in flight) and epilogue, and the second resident workgroup covers them.
  // Same arithmetic per row: bit-identical output; 15-18 % faster at the cross shapes (profiles/r04y_cross_attn_geometry.txt).
  const bool short_stream = a->rows == 0 && (int64_t)ceil_div(a->sk, KVBLK) * a->nchunks < 16;
  if (short_stream && (a->defer_log2 == 0 || a->defer_log2 == 8) && resident_eligible(a))
    return a->defer_log2 == 0 ? launch_resident<0>(a, stream) : launch_resident<8>(a, stream);
  switch (a->defer_log2) {
    case 0: return short_stream ? launch<0, 4, 1>(a, stream) : launch<0, 8, 2, 3>(a, stream);
    case 8: return short_stream ? launch<8, 4, 1>(a, stream) : launch<8, 8, 2, 3>(a, stream);
    case 90: return launch<0, 8, 2>(a, stream);      // forced 8-wave kernel (A/B, tests)
    case 98: return launch<8, 8, 2>(a, stream);
    case 50: return launch<0, 4, 1>(a, stream);     // geometry A/B: two 4-wave workgroups per CU
    case 58: return launch<8, 4, 1>(a, stream);
    case 60: return launch<0, 8, 2, 2>(a, stream);   // 4 waves x 64 rows, one wave per SIMD (am_attention64.hip)
    case 68: return launch<8, 8, 2, 2>(a, stream);
    case 28: return launch<8, 8, 2, 2>(a, stream);   // same kernel family, exact deferred re-base (no lazy pass)
    case 70: return launch<0, 8, 2, 1>(a, stream);   // balanced two-phase schedule
    case 78: return launch<8, 8, 2, 1>(a, stream);
    default:
#ifdef AM_ATTN_ABLATIONS
      if (a->defer_log2 >= 3000) return launch<8, 8, 2, 2>(a, stream);     // 4x64 timing ablations
      if (a->defer_log2 >= 100) return am_attention_variant(a, stream);
#endif
      AM_FAIL(AM_ERR_INVALID, "am_attention_bf16: defer_log2 must be 0 or 8 (got %d)", a->defer_log2);
  }
}
