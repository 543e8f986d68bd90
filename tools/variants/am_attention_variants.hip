// Experimental attention schedules kept for A/B measurements (tools/kernel_bench.py, tools/attn_profile.py).
// Compiled only with -DAM_ATTN_ABLATIONS; the product kernel is am_attention.hip.  All variants use the same
// operand layouts and MFMA formulation; they differ in how the tile loop is scheduled:
//   +100  staggered: waves 4-7 run one phase behind waves 0-3, two barriers per 64-key tile
//   +200  lockstep : one barrier per tile, explicit -inf key mask, un-scaled Q (the round-1 v3 kernel)
//   +300  pipelined: softmax of tile t threaded through the MFMA gaps of tile t+1 (two score tiles live)
//   +400  lean64   : the product kernel's instruction diet with 64-key tiles (no super-tiles, no split tail)
//   1001.. / 2001..: timing-only ablations (wrong results by construction)
// Measured (MI355X, headline shape): all within 10 % of each other and none faster than the product kernel -
// see DESIGN.md section 4.1 for what that says about MFMA/VALU overlap on gfx950.
#ifdef AM_ATTN_ABLATIONS
#include <type_traits>

#include "am_common.h"

namespace {

constexpr int QBLK = 256;          // query rows per workgroup
constexpr int KVBLK = 64;          // keys per tile
constexpr int HD = 128;
constexpr int K_TILE_B = KVBLK * HD * 2;   // 16 KiB
constexpr int V_TILE_B = HD * KVBLK * 2;   // 16 KiB
constexpr int SMEM_BYTES = 2 * (K_TILE_B + V_TILE_B);   // 64 KiB

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int DEFER, bool STAGGER, int ABL>
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel(am_attn_args p, int tiles_per_chunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;                       // [2][16 KiB]
  unsigned char* Vs = smem + 2 * K_TILE_B;        // [2][16 KiB]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.y;                          // sequence * heads + head
  const int head = bh % p.heads, seq = bh / p.heads;
  const int q0 = blockIdx.x * QBLK + wave * 32;
  const bool late = STAGGER && wave >= 4;             // waves 4-7: one phase behind

  // ---- Q fragments (B operand): Q[q0 + l31][ks*16 + hi*8 .. +8] -------------
  bf16x8_t qf[8];
  {
    const bf16_t* qp = p.Q + ((int64_t)bh * p.sq_pad + q0 + l31) * HD + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 16);
  }

  // ---- DMA descriptors.  Unit U of a tile (16 B each, 1024 per tile) is owned by
  // (j, wave, lane): U = j*512 + wave*64 + lane, j = 0,1 (lockstep) or, with STAGGER,
  // each half-workgroup owns one half: U = half*512 + j*256 + (wave&3)*64 + lane.
  int k_src[2], v_src_row[2], v_src_col[2], u_base[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ub = STAGGER ? ((wave >> 2) * 512 + j * 256 + (wave & 3) * 64) : (j * 512 + wave * 64);
    u_base[j] = ub;                                 // wave-uniform LDS unit base of this instruction
    const int U = ub + lane;
    const int kr = U >> 4, kc = (U & 15) ^ (kr & 15);
    k_src[j] = kr * HD + kc * 8;
    const int vr = U >> 3, vc = (U & 7) ^ ((vr >> 1) & 7);
    v_src_row[j] = vr;
    v_src_col[j] = vc * 8;
  }
  const int64_t k_seq_stride = (int64_t)p.sk_pad * HD;     // per (seq, head)
  const int total_tiles = p.nchunks * tiles_per_chunk;
  const bf16_t* k_base = p.K + (int64_t)bh * k_seq_stride;
  const bf16_t* v_base = p.Vt + (int64_t)bh * k_seq_stride;
  const bf16_t* v_lane[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) v_lane[j] = v_base + (int64_t)v_src_row[j] * p.sk_pad + v_src_col[j];

  // tile t -> (chunk, tile in chunk) without a division per tile: the DMA cursors only move forward
  int dk_chunk = 0, dk_tt = 0, dv_chunk = 0, dv_tt = 0;
  auto dma_k = [&](int buf) {      // DMA the K tile at the K cursor, then advance it
    if (ABL == 5 || ABL == 6) return;
    const bf16_t* kb = k_base + (int64_t)dk_chunk * p.chunk_stride + (int64_t)dk_tt * KVBLK * HD;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(kb + k_src[j]), (lds_ptr_t)(Ks + buf * K_TILE_B + u_base[j] * 16), 16, 0, 0);
    if (++dk_tt == tiles_per_chunk) { dk_tt = 0; ++dk_chunk; }
  };
  auto dma_v = [&](int buf) {
    if (ABL == 5 || ABL == 6) return;
    const int64_t off = (int64_t)dv_chunk * p.chunk_stride + (int64_t)dv_tt * KVBLK;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(v_lane[j] + off), (lds_ptr_t)(Vs + buf * V_TILE_B + u_base[j] * 16), 16, 0, 0);
    if (++dv_tt == tiles_per_chunk) { dv_tt = 0; ++dv_chunk; }
  };

  f32x16_t o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY;   // running max, in log2 units (score * scale * log2 e)
  float l_run = 0.f;         // this half-lane's partial row sum
  const float c = p.scale * 1.4426950408889634f;

  // fragment read offsets (bytes) inside a tile
  int k_off[8], v_off[4];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_off[ks] = l31 * 256 + (((ks * 2 + hi) ^ (l31 & 15)) << 4);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) v_off[kk] = l31 * 128 + (((kk * 2 + hi) ^ ((l31 >> 1) & 7)) << 4);
  f32x16_t s[2];
  int c_tt = 0;              // compute cursor: tile index inside its chunk (for the tail mask)

  // ---- phase 1: S^T = K Q^T.  The 16 K fragments are read 8 deep ahead of the MFMAs that
  // consume them: 8 reads up front, then every MFMA of the first key block re-fills its
  // fragment slot with the second block's fragment.
  auto qk_phase = [&](int buf) {
    const unsigned char* kp0 = Ks + buf * K_TILE_B;
    const unsigned char* kp1 = kp0 + 32 * 256;
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
    for (int ks = 0; ks < 8; ++ks) kf[ks] = *reinterpret_cast<const bf16x8_t*>(kp0 + k_off[ks]);
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[0][r] = 0.f; s[1][r] = 0.f; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s[0], 0, 0, 0);
      kf[ks] = *reinterpret_cast<const bf16x8_t*>(kp1 + k_off[ks]);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s[1], 0, 0, 0);
  };

  // ---- phase 2: online softmax (row = this lane's query), then O^T += V^T P^T -------
  auto softmax_pv_phase = [&](int buf) {
    // V^T fragments of the first two 16-key steps are fetched now and land under the softmax
    const unsigned char* vp = Vs + buf * V_TILE_B;
    bf16x8_t vf[8];
    const int valid = p.sk - c_tt * KVBLK;      // wave-uniform; < 64 only on a chunk's last tile
    if (++c_tt == tiles_per_chunk) c_tt = 0;
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
                                        : *reinterpret_cast<const bf16x8_t*>(vp + d * 32 * 128 + v_off[kk]);
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pq[kk], o[d], 0, 0, 0);
        }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int d = 0; d < 4; ++d)
        vf[kk * 4 + d] = *reinterpret_cast<const bf16x8_t*>(vp + d * 32 * 128 + v_off[kk]);
    __builtin_amdgcn_sched_barrier(0);
    if (valid < KVBLK) {     // mask the padded keys of a chunk's last tile (a real branch: rare)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= valid) s[kb][r] = -INFINITY;
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
    float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], c, -m_run));
        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r + 1], c, -m_run));
        s[kb][r] = p0;
        s[kb][r + 1] = p1;
        rs0 += p0;
        rs1 += p1;
      }
    l_run += rs0 + rs1;
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
        vf[kk * 4 + d] = *reinterpret_cast<const bf16x8_t*>(vp + d * 32 * 128 + v_off[kk + 2]);
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
    // prologue: tile 0 in buffer 0 (each half-workgroup DMAs its half)
    dma_k(0);
    dma_v(0);
    if (late) dma_drain_barrier();                  // waves 4-7 start one phase late
    for (int t = 0; t < total_tiles; ++t) {
      dma_drain_barrier();                          // ---- phase 1 of tile t   (drains + publishes DMAs)
      if (t + 1 < total_tiles) dma_k((t + 1) & 1);
      qk_phase(t & 1);
      dma_drain_barrier();                          // ---- phase 2 of tile t
      if (t + 1 < total_tiles) dma_v((t + 1) & 1);
      softmax_pv_phase(t & 1);
    }
    if (!late) dma_drain_barrier();                 // balance the barrier count
  } else {
    dma_k(0);
    dma_v(0);
    for (int t = 0; t < total_tiles; ++t) {
      dma_drain_barrier();                          // tile t landed; everyone is done with the other buffer
      if (t + 1 < total_tiles) { dma_k((t + 1) & 1); dma_v((t + 1) & 1); }
      qk_phase(t & 1);
      softmax_pv_phase(t & 1);
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


// ===========================================================================
// Software-pipelined schedule (default).  Measurements on MI355X (tools/ubench,
// tools/kernel_bench.py --ablate) show that the two waves sharing a SIMD do not
// hide each other's softmax: time(full) ~= time(MFMA only) + time(everything
// else).  What does hide VALU work is issuing it from the SAME wave inside the
// 32-cycle shadow of its own MFMAs.  So each wave keeps two score tiles live and
// every tile iteration is two MFMA streams with the softmax of the neighbouring
// tile threaded through their gaps:
//   phase A: S(t+1) = K(t+1) Q^T   (16 MFMA)  ||  finish softmax(t): exp2, row sums, bf16 P(t)
//   phase B: O += V(t)^T P(t)      (16 MFMA)  ||  start softmax(t+1): mask, row max, rescale decision
// A pending rescale (rare with the deferred threshold) is applied after the
// P.V MFMAs of phase B, i.e. when everything accumulated so far is at the old
// max (the ordering hazard of deferred rescaling).  One barrier per tile; the
// K DMA runs two tiles ahead, the V^T DMA one tile ahead.
// ===========================================================================
template <int DEFER>
__global__ __launch_bounds__(512, 2) void attn_fwd_pipe_kernel(am_attn_args p, int tiles_per_chunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;                       // [2][16 KiB]
  unsigned char* Vs = smem + 2 * K_TILE_B;        // [2][16 KiB]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.y;
  const int head = bh % p.heads, seq = bh / p.heads;
  const int q0 = blockIdx.x * QBLK + wave * 32;
  const float c = p.scale * 1.4426950408889634f;

  bf16x8_t qf[8];                                 // pre-scaled to log2 units (see the lean kernel)
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

  const int64_t k_seq_stride = (int64_t)p.sk_pad * HD;
  const bf16_t* k_lane[2];
  const bf16_t* v_lane[2];
  int u_byte[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    u_byte[j] = (j * 512 + wave * 64) * 16;
    const int U = j * 512 + wave * 64 + lane;
    const int kr = U >> 4, kc = (U & 15) ^ (kr & 15);
    k_lane[j] = p.K + (int64_t)bh * k_seq_stride + kr * HD + kc * 8;
    const int vr = U >> 3, vc = (U & 7) ^ ((vr >> 1) & 7);
    v_lane[j] = p.Vt + (int64_t)bh * k_seq_stride + (int64_t)vr * p.sk_pad + vc * 8;
  }
  const int total_tiles = p.nchunks * tiles_per_chunk;
  int dk_tt = 0, dv_tt = 0;
  int64_t dk_chunk = 0, dv_chunk = 0;
  auto dma_k = [&](int buf) {
    const int64_t ko = dk_chunk + (int64_t)dk_tt * (KVBLK * HD);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(k_lane[j] + ko), (lds_ptr_t)(Ks + buf * K_TILE_B + u_byte[j]), 16, 0, 0);
    if (++dk_tt == tiles_per_chunk) { dk_tt = 0; dk_chunk += p.chunk_stride; }
  };
  auto dma_v = [&](int buf) {
    const int64_t vo = dv_chunk + (int64_t)dv_tt * KVBLK;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(v_lane[j] + vo), (lds_ptr_t)(Vs + buf * V_TILE_B + u_byte[j]), 16, 0, 0);
    if (++dv_tt == tiles_per_chunk) { dv_tt = 0; dv_chunk += p.chunk_stride; }
  };

  f32x16_t o[4], zero16;
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

  int c_tt = 0;                 // tile-in-chunk cursor of the finish step (row-sum tail correction)
  bool pend = false;            // rescale decided by start_softmax, applied after the next P.V
  float pend_delta = 0.f;

  auto max3 = [](float a, float b, float cc) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(cc));
    return d;
  };
  // ---- start softmax of a raw score tile: row max + rescale decision (straight-line) ----------
  auto start_softmax = [&](f32x16_t (&sx)[2]) {
    asm volatile("s_nop 15" : "+v"(sx[0]), "+v"(sx[1]));   // MFMA result -> inline-asm VALU read hazard
    float mxa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mxa[i] = max3(sx[0][i], sx[1][i], sx[0][i + 4]);
#pragma unroll
    for (int i = 0; i < 4; ++i) mxa[i] = max3(mxa[i], sx[1][i + 4], sx[0][i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) mxa[i] = max3(mxa[i], sx[1][i + 8], sx[0][i + 12]);
#pragma unroll
    for (int i = 0; i < 4; ++i) mxa[i] = max3(mxa[i], sx[1][i + 12], mxa[i]);
    float mx = max3(mxa[0], mxa[1], max3(mxa[2], mxa[3], mxa[3]));
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = max3(__uint_as_float(sw[0]), __uint_as_float(sw[1]), __uint_as_float(sw[1]));
    }
    mx -= m_run;
    pend = first || !__all(mx <= (float)DEFER);
    pend_delta = first ? mx : fmaxf(mx, 0.f);
  };
  auto apply_rescale = [&]() {
    if (pend) {
      const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-pend_delta);
      first = false;
      m_run += pend_delta;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      pend = false;
    }
  };
  // ---- finish softmax: P = exp2(S - m), row sums, bf16 B-operand fragments ----------------------
  auto finish_softmax = [&](f32x16_t (&sx)[2], bf16x8_t (&pf)[4]) {
    const f32x2_t m2 = {m_run, m_run};
    f32x2_t rsa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rsa[i] = f32x2_t{0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2_t x = f32x2_t{sx[kb][r], sx[kb][r + 1]} - m2;
        const f32x2_t pp = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
        sx[kb][r] = pp[0];
        sx[kb][r + 1] = pp[1];
        rsa[(r >> 1) & 3] += pp;
      }
    const f32x2_t rs = (rsa[0] + rsa[1]) + (rsa[2] + rsa[3]);
    l_run += rs[0] + rs[1];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      u32x4_t w;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        w[e] = pack_bf2(sx[kk >> 1][(kk & 1) * 8 + 2 * e], sx[kk >> 1][(kk & 1) * 8 + 2 * e + 1]);
      pf[kk] = __builtin_bit_cast(bf16x8_t, w);
    }
  };
  auto tail_fix = [&]() {       // partial last tile of a chunk (see the lean kernel)
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
  };
  constexpr int PF = 4;         // fragment reads run 4 MFMAs ahead of their consumer
  auto qk_mfma = [&](int buf, f32x16_t (&sx)[2]) {
    const unsigned char* kp0 = Ks + buf * K_TILE_B;
    bf16x8_t kf[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) kf[i] = *reinterpret_cast<const bf16x8_t*>(kp0 + k_off[i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) {          // i = kb*8 + ks
      const int kb = i >> 3, ks = i & 7;
      sx[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i % PF], qf[ks], ks == 0 ? zero16 : sx[kb], 0, 0, 0);
      if (i + PF < 16) {
        const int n = i + PF;
        kf[i % PF] = *reinterpret_cast<const bf16x8_t*>(kp0 + (n >> 3) * 32 * 256 + k_off[n & 7]);
      }
    }
  };
  auto pv_mfma = [&](int buf, const bf16x8_t (&pf)[4]) {
    const unsigned char* vp = Vs + buf * V_TILE_B;
    bf16x8_t vf[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) vf[i] = *reinterpret_cast<const bf16x8_t*>(vp + (i & 3) * 32 * 128 + v_off[i >> 2]);
#pragma unroll
    for (int i = 0; i < 16; ++i) {          // i = kk*4 + d
      const int kk = i >> 2, d = i & 3;
      o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i % PF], pf[kk], o[d], 0, 0, 0);
      if (i + PF < 16) {
        const int n = i + PF;
        vf[i % PF] = *reinterpret_cast<const bf16x8_t*>(vp + (n & 3) * 32 * 128 + v_off[n >> 2]);
      }
    }
  };

  // one tile iteration: sc = scores of tile t (started), sn = scores of tile t+1 (to compute)
  auto iteration = [&](int t, f32x16_t (&sc)[2], f32x16_t (&sn)[2]) {
    dma_drain_barrier();                                  // K(t+1), V(t) landed; old buffers free
    if (t + 2 < total_tiles) dma_k(t & 1);
    if (t + 1 < total_tiles) dma_v((t + 1) & 1);
    bf16x8_t pf[4];
    const bool has_next = t + 1 < total_tiles;
    // ---- phase A: QK^T(t+1) || finish softmax(t) --------------------------------------
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) {
      qk_mfma((t + 1) & 1, sn);
      finish_softmax(sc, pf);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // 1 MFMA
        if (g < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // 1 DS read (4 ahead)
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);             // 3 VALU (pk sub / pk add / cvt)
        __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);             // 2 TRANS (v_exp)
      }
    } else {
      finish_softmax(sc, pf);
    }
    __builtin_amdgcn_sched_barrier(0);
    tail_fix();
    // ---- phase B: P.V(t) || start softmax(t+1) --------------------------------------------
    pv_mfma(t & 1, pf);
    if (has_next) {
      start_softmax(sn);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // 1 MFMA
        if (g < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // 1 DS read (4 ahead)
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);             // 2 VALU
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    apply_rescale();                                  // after ALL of P.V(t): everything is at the old max
  };

  // ---- prologue ------------------------------------------------------------------------------
  f32x16_t sa[2], sb[2];
  dma_k(0);
  dma_v(0);
  if (total_tiles > 1) dma_k(1);
  dma_drain_barrier();
  qk_mfma(0, sa);
  start_softmax(sa);
  apply_rescale();
  for (int t = 0; t < total_tiles; t += 2) {
    iteration(t, sa, sb);
    if (t + 1 < total_tiles) iteration(t + 1, sb, sa);
  }

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

template <int DEFER>
int launch_pipe(const am_attn_args* a, void* stream) {
  static bool attr_set = false;
  if (!attr_set) {
    AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_pipe_kernel<DEFER>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
    attr_set = true;
  }
  const int tiles_per_chunk = ceil_div(a->sk, KVBLK);
  dim3 grid(ceil_div(a->sq, QBLK), a->nseq * a->heads);
  hipLaunchKernelGGL((attn_fwd_pipe_kernel<DEFER>), grid, dim3(512), SMEM_BYTES, (hipStream_t)stream, *a, tiles_per_chunk);
  AM_HIP(hipGetLastError());
  return AM_OK;
}


// ===========================================================================
// "Lean" schedule (default).  Measured on MI355X (tools/ubench, kernel_bench
// --ablate): with MFMAs in flight a SIMD retires roughly one instruction per
// ~6 cycles in total, so the kernel time is ~max(32 cyc x MFMAs, 6 cyc x ALL
// instructions): the attention loop is instruction-issue-bound, and neither
// staggering the waves nor threading the softmax through the MFMA gaps helps
// while there are ~10 non-MFMA instructions per MFMA.  This variant therefore
// minimises the instruction count per tile:
//   * Q is pre-multiplied by scale*log2(e) once (in registers, re-rounded to
//     bf16), so scores are born in log2 units: no per-element multiply;
//   * the running max is subtracted two elements at a time (v_pk_add_f32);
//   * no key mask: padded K rows / V^T columns are zero (am_head_post), the row
//     sum of a partial tile is corrected once per tile (tail_fix);
//   * row sums with packed adds, row max with v_max3, cross-half exchange with
//     v_permlane32_swap (no LDS round trip);
//   * tile loop unrolled by two so LDS addresses are immediates; one DMA cursor.
// Per element and lane: 1 exp + 0.5 sub + 0.5 add + 0.5 cvt + 0.5 max.
// ===========================================================================
// SPLIT: the workgroup handles only the key tiles [z*nt/Z, (z+1)*nt/Z) (z = blockIdx.z) of query block
// `qblk_base + blockIdx.x` and writes un-normalised fp32 (O, m, l) partials for attn_combine_kernel.
// Used for the short last query block of every sequence: seq = T*(N+1) is 256*k + a few rows for every
// reference shape, and a 16-row block would otherwise cost a full extra round of workgroups (6 %).
constexpr int PART_LD = HD + 4;   // floats per partial row: O[128], m, l, pad (keeps rows 16-byte aligned)
template <int DEFER, bool PROF = false, int LA = 0, bool SPLIT = false>   // LA: timing-only ablations
__global__ __launch_bounds__(512, 2) void attn_fwd_lean_kernel(am_attn_args p, int tiles_per_chunk,
                                                              unsigned long long* prof = nullptr,
                                                              int qblk_base = 0, float* part = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // PROF: waves of block (0,0) record s_memtime at 6 points of tiles 64..71 -> prof[wave][tile][6]
  int prof_t = 0;
  auto stamp = [&](int slot) {
    if (PROF) {
      if (blockIdx.x == 0 && blockIdx.y == 0 && prof_t >= 64 && prof_t < 72) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long tk = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) prof[((threadIdx.x >> 6) * 8 + (prof_t - 64)) * 6 + slot] = tk;
      }
    }
  };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.y;
  const int head = bh % p.heads, seq = bh / p.heads;
  const int q0 = (qblk_base + blockIdx.x) * QBLK + wave * 32;
  const float c = p.scale * 1.4426950408889634f;

  // ---- Q fragments, pre-scaled to log2 units ---------------------------------------------
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

  // ---- DMA descriptors (unit U = j*512 + wave*64 + lane of each 1024-unit tile) -------------
  const int64_t k_seq_stride = (int64_t)p.sk_pad * HD;
  const bf16_t* k_lane[2];
  const bf16_t* v_lane[2];
  int u_byte[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    u_byte[j] = (j * 512 + wave * 64) * 16;
    const int U = j * 512 + wave * 64 + lane;
    const int kr = U >> 4, kc = (U & 15) ^ (kr & 15);
    k_lane[j] = p.K + (int64_t)bh * k_seq_stride + kr * HD + kc * 8;
    const int vr = U >> 3, vc = (U & 7) ^ ((vr >> 1) & 7);
    v_lane[j] = p.Vt + (int64_t)bh * k_seq_stride + (int64_t)vr * p.sk_pad + vc * 8;
  }
  const int all_tiles = p.nchunks * tiles_per_chunk;
  const int t_begin = SPLIT ? (int)((int64_t)blockIdx.z * all_tiles / gridDim.z) : 0;
  const int t_end = SPLIT ? (int)((int64_t)(blockIdx.z + 1) * all_tiles / gridDim.z) : all_tiles;
  const int total_tiles = t_end - t_begin;
  int d_tt = t_begin % tiles_per_chunk;                                   // DMA cursor: tile inside its chunk
  int64_t d_chunk = (int64_t)(t_begin / tiles_per_chunk) * p.chunk_stride; // element offset of the cursor's chunk
  auto dma_tile = [&](int buf) {      // K and V^T tile at the cursor -> LDS buffer `buf`; advance
    const int64_t ko = d_chunk + (int64_t)d_tt * (KVBLK * HD);
    const int64_t vo = d_chunk + (int64_t)d_tt * KVBLK;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(k_lane[j] + ko), (lds_ptr_t)(smem + buf * K_TILE_B + u_byte[j]), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(v_lane[j] + vo),
                                       (lds_ptr_t)(smem + 2 * K_TILE_B + buf * V_TILE_B + u_byte[j]), 16, 0, 0);
    if (++d_tt == tiles_per_chunk) { d_tt = 0; d_chunk += p.chunk_stride; }
  };

  f32x16_t o[4], zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    zero16[r] = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d][r] = 0.f;
  }
  float m_run = 0.f;       // running max (log2 units); defined by the first tile
  float l_run = 0.f;       // this half-lane's partial row sum
  bool first = true;

  int k_off[8], v_off[4];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_off[ks] = l31 * 256 + (((ks * 2 + hi) ^ (l31 & 15)) << 4);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) v_off[kk] = l31 * 128 + (((kk * 2 + hi) ^ ((l31 >> 1) & 7)) << 4);
  int c_tt = t_begin % tiles_per_chunk;   // compute cursor (tail correction)

  auto tile = [&](int buf, bool more) {
    stamp(0);                                 // before the barrier
    dma_drain_barrier();                          // tile landed (drains the DMA); other buffer is free
    stamp(1);                                 // barrier passed
    if (more) dma_tile(buf ^ 1);
    const unsigned char* kp = smem + buf * K_TILE_B;
    const unsigned char* vp = smem + 2 * K_TILE_B + buf * V_TILE_B;
    // ---- S = K Q'^T (log2 units) ----------------------------------------------------------------
    f32x16_t s[2];
    {
      bf16x8_t kf[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) kf[ks] = *reinterpret_cast<const bf16x8_t*>(kp + k_off[ks]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], ks == 0 ? zero16 : s[0], 0, 0, 0);
        kf[ks] = *reinterpret_cast<const bf16x8_t*>(kp + 32 * 256 + k_off[ks]);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], ks == 0 ? zero16 : s[1], 0, 0, 0);
    }
    if (PROF) { asm volatile("" ::"v"(s[0]), "v"(s[1])); stamp(2); }   // QK^T issued + drained
    // V^T fragments of the first two 16-key steps: in flight under the softmax
    bf16x8_t vf[8];
    if (LA != 5) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int d = 0; d < 4; ++d)
          vf[kk * 4 + d] = *reinterpret_cast<const bf16x8_t*>(vp + d * 32 * 128 + v_off[kk]);
    }
    // ---- row max of S' (= how far this tile's max is above the running max) ----------------------
    // (four independent v_max3 chains: the softmax section is VALU-latency-bound, not
    // throughput-bound - in-kernel s_memtime stamps, tools/attn_profile.py)
    // v_max3 through inline asm: fmaxf() on MFMA outputs makes hipcc emit a canonicalising
    // v_max x,x,x per operand (32 extra VALU per tile)
    auto max3 = [](float a, float b, float cc) {
      float d;
      asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(cc));
      return d;
    };
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
    if (LA == 2 || LA == 6) mx = s[0][0];         // ablation: no row max
    mx -= m_run;                                  // how far this tile's max is above the running max
    if (first || !__all(mx <= (float)DEFER)) {   // rare after the first tile (deferred rescale)
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
    // ---- P = exp2(S - m_run), row sums, bf16 B-operand fragments ------------------------------------
    const f32x2_t m2 = {m_run, m_run};
    f32x2_t rsa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rsa[i] = f32x2_t{0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2_t x = f32x2_t{s[kb][r], s[kb][r + 1]} - m2;      // v_pk_add_f32
        const f32x2_t pp = (LA == 1 || LA == 6) ? x : f32x2_t{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
        s[kb][r] = pp[0];
        s[kb][r + 1] = pp[1];
        if (LA != 3 && LA != 6) rsa[(r >> 1) & 3] += pp;
      }
    {
      const f32x2_t rs = (rsa[0] + rsa[1]) + (rsa[2] + rsa[3]);
      l_run += rs[0] + rs[1];
    }
    {   // partial last tile of a chunk: remove the padded keys' exp2(0 - m_run) from the row sum
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
    }
    bf16x8_t pf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      u32x4_t w;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        w[e] = (LA == 4 || LA == 6) ? __float_as_uint(s[kk >> 1][(kk & 1) * 8 + 2 * e])
                                    : pack_bf2(s[kk >> 1][(kk & 1) * 8 + 2 * e], s[kk >> 1][(kk & 1) * 8 + 2 * e + 1]);
      pf[kk] = __builtin_bit_cast(bf16x8_t, w);
    }
    if (LA == 5) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int d = 0; d < 4; ++d)
          vf[kk * 4 + d] = *reinterpret_cast<const bf16x8_t*>(vp + d * 32 * 128 + v_off[kk]);
    }
    // ---- O^T += V^T P^T ------------------------------------------------------------------------------
    __builtin_amdgcn_sched_barrier(0);
    if (PROF) { asm volatile("" ::"v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3])); stamp(3); }   // softmax done
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kk * 4 + d], pf[kk], o[d], 0, 0, 0);
        vf[kk * 4 + d] = *reinterpret_cast<const bf16x8_t*>(vp + d * 32 * 128 + v_off[kk + 2]);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 2; kk < 4; ++kk)
#pragma unroll
      for (int d = 0; d < 4; ++d)
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[(kk - 2) * 4 + d], pf[kk], o[d], 0, 0, 0);
    if (PROF) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::"v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));
      stamp(4);                               // P.V drained
      ++prof_t;
    }
  };

  dma_tile(0);
  for (int t = 0; t < total_tiles; t += 2) {
    tile(0, t + 1 < total_tiles);
    if (t + 1 < total_tiles) tile(1, t + 2 < total_tiles);
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const int q = q0 + l31;
  if (SPLIT) {
    if (q < p.sq) {
      float* pp = part + ((((int64_t)bh * gridDim.z + blockIdx.z) * QBLK) + (q - qblk_base * QBLK)) * PART_LD;
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

// merge the Z partial results of the split (tail) query block: O = sum_z 2^(m_z - m) O_z / sum_z 2^(m_z - m) l_z
__global__ __launch_bounds__(128) void attn_combine_kernel(am_attn_args p, const float* __restrict__ part, int Z,
                                                           int qblk_base, int rows) {
  const int bh = blockIdx.y, row = blockIdx.x;          // one 128-thread block per (sequence*head, tail row)
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

template <int DEFER, int LA = 0>
int launch_lean(const am_attn_args* a, void* stream) {
  static bool attr_set = false;
  if (!attr_set) {
    AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_lean_kernel<DEFER, false, LA, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
    AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_lean_kernel<DEFER, false, LA, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
    attr_set = true;
  }
  const int tiles_per_chunk = ceil_div(a->sk, KVBLK);
  const int all_tiles = tiles_per_chunk * a->nchunks;
  const int nblk = ceil_div(a->sq, QBLK);
  const int bh = a->nseq * a->heads;
  const int tail_rows = a->sq - (nblk - 1) * QBLK;
  // split the short last query block over the key range when it would otherwise add a round
  constexpr int Z = 16;
  static float* part = nullptr;
  static size_t part_elems = 0;
  const size_t need = (size_t)bh * Z * QBLK * PART_LD;
  const bool split = LA == 0 && nblk >= 9 && tail_rows <= 128 && all_tiles >= 4 * Z && need * sizeof(float) <= (256u << 20);
  if (split && part_elems < need) {     // library-owned scratch, grown on demand (never on a captured stream)
    if (part) AM_HIP(hipFree(part));
    part = nullptr; part_elems = 0;
    AM_HIP(hipMalloc(reinterpret_cast<void**>(&part), need * sizeof(float)));
    part_elems = need;
  }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(split ? nblk - 1 : nblk, bh);
  hipLaunchKernelGGL((attn_fwd_lean_kernel<DEFER, false, LA, false>), grid, dim3(512), SMEM_BYTES, st, *a,
                     tiles_per_chunk, (unsigned long long*)nullptr, 0, (float*)nullptr);
  if (split) {
    hipLaunchKernelGGL((attn_fwd_lean_kernel<DEFER, false, LA, true>), dim3(1, bh, Z), dim3(512), SMEM_BYTES, st, *a,
                       tiles_per_chunk, (unsigned long long*)nullptr, nblk - 1, part);
    hipLaunchKernelGGL(attn_combine_kernel, dim3(tail_rows, bh), dim3(128), 0, st, *a, part, Z, nblk - 1, tail_rows);
  }
  AM_HIP(hipGetLastError());
  return AM_OK;
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

// per-phase s_memtime stamps of the lean kernel (block (0,0), 8 waves x tiles 64..71 x 6 slots)
extern "C" int am_attention_profile(const am_attn_args* a, unsigned long long* prof_dev, void* stream) {
  AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_lean_kernel<8, true>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
  const int tiles_per_chunk = ceil_div(a->sk, KVBLK);
  dim3 grid(ceil_div(a->sq, QBLK), a->nseq * a->heads);
  hipLaunchKernelGGL((attn_fwd_lean_kernel<8, true>), grid, dim3(512), SMEM_BYTES, (hipStream_t)stream, *a,
                     tiles_per_chunk, prof_dev, 0, (float*)nullptr);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

int am_attention_variant(const am_attn_args* a, void* stream) {
  switch (a->defer_log2) {
    case 400: return launch_lean<0>(a, stream);
    case 408: return launch_lean<8>(a, stream);
    case 300: return launch_pipe<0>(a, stream);
    case 308: return launch_pipe<8>(a, stream);
    case 200: return launch<0, false>(a, stream);
    case 208: return launch<8, false>(a, stream);
    case 100: return launch<0, true>(a, stream);
    case 108: return launch<8, true>(a, stream);
    case 2001: return launch_lean<8, 1>(a, stream);
    case 2002: return launch_lean<8, 2>(a, stream);
    case 2003: return launch_lean<8, 3>(a, stream);
    case 2004: return launch_lean<8, 4>(a, stream);
    case 2005: return launch_lean<8, 5>(a, stream);
    case 2006: return launch_lean<8, 6>(a, stream);
    case 1001: return launch<8, false, 1>(a, stream);
    case 1003: return launch<8, false, 3>(a, stream);
    case 1004: return launch<8, false, 4>(a, stream);
    case 1005: return launch<8, false, 5>(a, stream);
    case 1006: return launch<8, false, 6>(a, stream);
    case 1101: return launch<8, true, 1>(a, stream);
    case 1103: return launch<8, true, 3>(a, stream);
    case 1104: return launch<8, true, 4>(a, stream);
    case 1105: return launch<8, true, 5>(a, stream);
    case 1106: return launch<8, true, 6>(a, stream);
    default: AM_FAIL(AM_ERR_INVALID, "am_attention_variant: unknown variant code %d", a->defer_log2);
  }
}
#endif  // AM_ATTN_ABLATIONS
