// bf16 GEMM with fused epilogue for gfx950:  C = act(A @ W^T + bias) + residual
//
// Replaces nn.Linear (+ diffusers GELU, + residual add, + torch.cat of the skip
// branch) on the reference hot path: block.py:131-152,
// attention_processor.py:92-103,147, temporal_denoiser.py:206,214,242.
//
// Structure (round 1): 128x128x64 block tile, 4 waves (2x2), each wave a 64x64
// sub-tile = 2x2 v_mfma_f32_32x32x16_bf16 accumulators.  Operands are staged
// global -> VGPR -> LDS (issue-early / write-late, one barrier per K-tile,
// double-buffered LDS).  LDS rows are padded to 144 B so every ds_read_b128
// lane group hits 16 distinct 16-byte slots.  The MFMA is issued as
// D[n][m] = W[n][k] * A[m][k] so each lane ends with 4 consecutive output
// columns -> 16-byte LDS writes in the epilogue, which re-reads the tile
// row-major for fully coalesced bias/GELU/residual/store.
#include <stdlib.h>
#include <string.h>

#include "am_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_ROW = BK + 8;               // bf16 elements per padded LDS row (144 B)
constexpr int TILE_ELEMS = BM * LDS_ROW;      // one operand tile, one buffer
constexpr int CS_LD = BN + 4;                 // fp32 epilogue staging row (528 B)
constexpr int SMEM_MAIN = 2 * 2 * TILE_ELEMS * (int)sizeof(bf16_t);   // 73728
constexpr int SMEM_EPI = BM * CS_LD * (int)sizeof(float);             // 67584
constexpr int SMEM_BYTES = SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI;
constexpr int GROUP_M = 8;                    // m-tiles swept per pass over the W panels
constexpr int B2 = 256;                       // tile edge of the 256x256 kernels

__device__ inline int64_t map_row(int r, int G, int gs, int off) {
  if (G <= 0) return r;
  int g = r / G;
  return (int64_t)g * gs + off + (r - g * G);
}

__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(am_gemm_args p, int tiles_m, int tiles_n, int m_base) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Bs = As + 2 * TILE_ELEMS;
  float* Cs = reinterpret_cast<float*>(smem);

  // ---- block id -> output tile: bijective XCD remap, then grouped ordering --
  const int nb = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  int lid;
  {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nb >> 3, r = nb & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  {
    const int group_sz = GROUP_M * tiles_n;
    const int g = lid / group_sz;
    const int first_m = g * GROUP_M;
    const int gm = min(GROUP_M, tiles_m - first_m);
    const int in_g = lid - g * group_sz;
    tm = first_m + in_g % gm;
    tn = in_g / gm;
  }
  const int m0 = m_base + tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- per-thread staging addresses (4 x 16 B of A and of W per K-tile) ----
  const bf16_t* a1p[4];
  const bf16_t* a2p[4];
  const bf16_t* wp[4];
  int lds_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    const int row = c >> 3, col = (c & 7) * 8;
    const int ar = min(m0 + row, p.M - 1);
    const int64_t pr = map_row(ar, p.a_G, p.a_gs, p.a_off);
    a1p[i] = p.A1 + pr * p.lda1 + col;
    a2p[i] = p.A2 ? p.A2 + pr * p.lda2 + col - p.K1 : nullptr;
    const int wr = min(n0 + row, p.N - 1);
    wp[i] = p.W + (int64_t)wr * p.ldw + col;
    lds_off[i] = row * LDS_ROW + col;
  }

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4_t areg[4], wreg[4];
  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
    const bool first = k0 < p.K1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16_t* src = first ? a1p[i] + k0 : a2p[i] + k0;
      areg[i] = *reinterpret_cast<const u32x4_t*>(src);
      wreg[i] = *reinterpret_cast<const u32x4_t*>(wp[i] + k0);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4_t*>(&As[buf * TILE_ELEMS + lds_off[i]]) = areg[i];
      *reinterpret_cast<u32x4_t*>(&Bs[buf * TILE_ELEMS + lds_off[i]]) = wreg[i];
    }
  };

  const int nk = p.K / BK;
  load_tile(0);
  store_tile(0);

  const int a_frag_off = (wm * 64 + l31) * LDS_ROW + hi * 8;
  const int b_frag_off = (wn * 64 + l31) * LDS_ROW + hi * 8;

  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const bf16_t* Ab = As + buf * TILE_ELEMS + a_frag_off;
    const bf16_t* Bb = Bs + buf * TILE_ELEMS + b_frag_off;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8_t af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const bf16x8_t*>(Ab + i * 32 * LDS_ROW + ks * 16);
        bfr[i] = *reinterpret_cast<const bf16x8_t*>(Bb + i * 32 * LDS_ROW + ks * 16);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = AM_MFMA_32x32x16(bfr[j], af[i], acc[i][j]);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
  }

  // ---- epilogue: accumulators -> LDS (fp32) -> coalesced fused store --------
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int m = wm * 64 + i * 32 + l31;
        const int n = wn * 64 + j * 32 + 8 * g + 4 * hi;
        f32x4_t v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        *reinterpret_cast<f32x4_t*>(&Cs[m * CS_LD + n]) = v;
      }
  __syncthreads();

  const int c8 = (tid & 15) * 8;
  const int gn = n0 + c8;
  if (gn < p.N) {
    float bv[8], cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bv[e] = p.bias ? p.bias[gn + e] : 0.f;
      cs[e] = p.ln_stats ? p.ln_colsum[gn + e] : 0.f;
    }
#pragma unroll
    for (int pass = 0; pass < BM / 16; ++pass) {
      const int row = pass * 16 + (tid >> 4);
      const int gmr = m0 + row;
      if (gmr < p.M) {
        const int64_t pr = map_row(gmr, p.c_G, p.c_gs, p.c_off);
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(&Cs[row * CS_LD + c8]);
        const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(&Cs[row * CS_LD + c8 + 4]);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (p.ln_stats) {                                           // folded LayerNorm: same two FMAs as the 256x256 kernel
          // the statistics belong to the A row this output row was computed from (norm_out -> drop the time token -> proj_out reads the
          // residual stream through a row map, temporal_denoiser.py:239-242)
          const f32x2_t st = *reinterpret_cast<const f32x2_t*>(p.ln_stats + 2 * map_row(gmr, p.a_G, p.a_gs, p.a_off));
          const float nm = -st[0] * st[1];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = rbf(fmaf(v[e], st[1], fmaf(nm, cs[e], bv[e])));
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = rbf(v[e] + bv[e]);     // nn.Linear result in bf16
        }
        if ((p.act & 0xff) == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = rbf(gelu_erf(v[e]));   // F.gelu on bf16 -> bf16
        }
        if (p.residual) {
          const u32x4_t rv = *reinterpret_cast<const u32x4_t*>(p.residual + pr * p.ldc + gn);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] += bflo(rv[e]);
            v[2 * e + 1] += bfhi(rv[e]);
          }
        }
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
        *reinterpret_cast<u32x4_t*>(p.C + pr * p.ldc + gn) = o;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The remainder rows of a big GEMM (M = B T (N + 1) is 256 k + 32 at every reference shape: the one time token per frame), round 4.
// They used to go to the 128x128 kernel above: N / 128 workgroups, each walking ALL of K one 64-column tile at a time with ONE tile
// in flight - a chain of K / 64 exposed global-load latencies on a grid of 8-32 workgroups, 20-60 us per launch, six launches per
// layer (0.6 % of the step for 0.02 % of the flops).  This kernel is built for latency instead: a workgroup owns 32 rows x 64
// columns, its 8 waves are 2 column blocks x 4 K-QUARTERS, fragments come straight from global memory in MFMA layout (a lane's 8
// consecutive k of its row: one 16-byte load, no LDS staging), eight k-steps of loads are in flight per wave before their MFMAs, and
// the four K-quarters are summed through LDS.  Identity row maps only (am_gemm_bf16 keeps the 128x128 kernel for the rest).
// Epilogue arithmetic = the other kernels' (bias, folded LayerNorm, GELU, residual, the same rounding points).
constexpr int TAIL_LD = 64 + 4;       // fp32 per staged row
__global__ __launch_bounds__(512) void gemm_tail_kernel(am_gemm_args p, int m_base) {
  __shared__ float red[4][32][TAIL_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wn = wave & 1, wk = wave >> 1;
  const int m0 = m_base + blockIdx.y * 32, nb0 = blockIdx.x * 64;
  const int ar = min(m0 + l31, p.M - 1);
  const int wr = min(nb0 + wn * 32 + l31, p.N - 1);
  const bf16_t* a1 = p.A1 + (int64_t)ar * p.lda1 + hi * 8;
  const bf16_t* a2 = p.A2 ? p.A2 + (int64_t)ar * p.lda2 + hi * 8 - p.K1 : nullptr;
  const bf16_t* wp = p.W + (int64_t)wr * p.ldw + hi * 8;
  const int kq = p.K >> 2;                       // K % 64 == 0: a quarter is a whole number of 16-element k-steps
  const int k_begin = wk * kq, k_end = k_begin + kq;
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = k_begin; k0 < k_end; k0 += 128) {
    u32x4_t af[8], wf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int k = min(k0 + s * 16, k_end - 16);        // a short last group re-reads its last step; its MFMA is skipped below
      af[s] = *reinterpret_cast<const u32x4_t*>((k < p.K1 ? a1 : a2) + k);
      wf[s] = *reinterpret_cast<const u32x4_t*>(wp + k);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
      if (k0 + s * 16 < k_end)
        acc = AM_MFMA_32x32x16(__builtin_bit_cast(bf16x8_t, wf[s]), __builtin_bit_cast(bf16x8_t, af[s]), acc);
  }
  // D = W A^T: lane l31 = output row, register 4 g + e = column 8 g + 4 hi + e of the wave's 32
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<f32x4_t*>(&red[wk][l31][wn * 32 + 8 * g + 4 * hi]) = f32x4_t{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
  __syncthreads();
  const int row = tid >> 4, c4 = (tid & 15) * 4;
  const int gmr = m0 + row, gn = nb0 + c4;
  if (gmr >= p.M || gn >= p.N) return;
  float v[4];
  {
    const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(&red[0][row][c4]), s1 = *reinterpret_cast<const f32x4_t*>(&red[1][row][c4]);
    const f32x4_t s2 = *reinterpret_cast<const f32x4_t*>(&red[2][row][c4]), s3 = *reinterpret_cast<const f32x4_t*>(&red[3][row][c4]);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (s0[e] + s1[e]) + (s2[e] + s3[e]);
  }
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    const f32x4_t b = *reinterpret_cast<const f32x4_t*>(p.bias + gn);
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = b[e];
  }
  if (p.ln_stats) {
    const f32x2_t st = *reinterpret_cast<const f32x2_t*>(p.ln_stats + 2 * (int64_t)gmr);
    const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(p.ln_colsum + gn);
    const float nm = -st[0] * st[1];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = rbf(fmaf(v[e], st[1], fmaf(nm, cs[e], bv[e])));
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = rbf(v[e] + bv[e]);
  }
  if ((p.act & 0xff) == 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = rbf(gelu_erf(v[e]));
  }
  if (p.residual) {
    const u32x2_t rv = *reinterpret_cast<const u32x2_t*>(p.residual + (int64_t)gmr * p.ldc + gn);
    v[0] += bflo(rv[0]); v[1] += bfhi(rv[0]); v[2] += bflo(rv[1]); v[3] += bfhi(rv[1]);
  }
  *reinterpret_cast<u32x2_t*>(p.C + (int64_t)gmr * p.ldc + gn) = u32x2_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
}

// Shared tail of the 256x256 kernels: the bf16 tile staged in LDS (row = 512 B, 8-byte unit u of row m at u ^ (m & 15))
// goes out as row-contiguous 16-byte stores with the residual added on the way.
#ifndef AM_GEMM_NT
#define AM_GEMM_NT 1        // non-temporal residual loads and C stores: both are touched once per GEMM and would only evict the A / W panels the
#endif                      // XCD's other workgroups are about to re-read from L2 (0 = the A/B build, tools/build_gemm_variants.sh)
// The residual operand of a full, identity-mapped tile, fetched BEFORE the accumulators are staged through LDS (round 4): 16 loads of
// 16 bytes per thread in flight at once, their HBM latency under the staging pass and its barrier.  Before, the store loop issued them
// four at a time between the stores - four exposed round trips of loaded-HBM latency per tile, and a tile of the K = 1024 linears is
// only ~20 us of main loop.
struct ResidualPrefetch {
  u32x4_t rv[16];
  bool on;
};
__device__ __forceinline__ bool residual_prefetchable(const am_gemm_args& p, int tid, int m0, int n0) {
  return p.residual != nullptr && p.c_G <= 0 && m0 + B2 <= p.M && n0 + (tid & 31) * 8 < p.N && !(p.act & 0x1800);
}
__device__ __forceinline__ void residual_prefetch(const am_gemm_args& p, int tid, int m0, int n0, ResidualPrefetch& r) {
  r.on = residual_prefetchable(p, tid, m0, n0);
  if (r.on) {
    const int k16 = tid & 31, r16 = tid >> 5;
    const uint32_t lane_off = ((uint32_t)r16 * (uint32_t)p.ldc + (uint32_t)(n0 + k16 * 8)) * 2u;
    const int64_t step = (int64_t)16 * p.ldc;
    const bf16_t* rrow = p.residual + (int64_t)m0 * p.ldc;
#pragma unroll
    for (int pass = 0; pass < 16; ++pass) {
      const u32x4_t* q = reinterpret_cast<const u32x4_t*>(reinterpret_cast<const unsigned char*>(rrow + pass * step) + lane_off);
      r.rv[pass] = AM_GEMM_NT ? __builtin_nontemporal_load(q) : *q;
    }
  }
}

// LayerNorm statistics of the rows this tile writes, for the linear that consumes them (am_gemm_args.ln_part): the store loop's
// thread holds 8 consecutive bf16 results of one row, a row's 256 columns are 32 lanes - exactly the groups and the tree of the
// canonical definition in am_common.h.  The tree runs on DPP: quad_perm xor 1, xor 2, row_half_mirror, row_mirror inside the 16-lane
// row, row_bcast15 into the odd row; lane 31 of each half-wave ends with the slice's (mean, M2).  No sum of squares anywhere:
// nothing cancels however far a row sits from zero.
__device__ __forceinline__ void emit_row_part(const u32x4_t& sv, float* part_row, bool writer) {
  float x[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { x[2 * e] = bflo(sv[e]); x[2 * e + 1] = bfhi(sv[e]); }
  float mean, m2;
  row_part8(x, mean, m2);
#define AM_CHAN_STEP(CTRL, ROWMASK, HALF_N)                                                                                  \
  {                                                                                                                          \
    const float om = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, mean), __builtin_bit_cast(int, mean), CTRL, ROWMASK, 0xf, false)); \
    const float oq = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, m2), __builtin_bit_cast(int, m2), CTRL, ROWMASK, 0xf, false));     \
    row_part_merge_equal(mean, m2, om, oq, HALF_N);                                                                          \
  }
  AM_CHAN_STEP(0xB1, 0xf, 4.f)      // quad_perm [1,0,3,2]
  AM_CHAN_STEP(0x4E, 0xf, 8.f)      // quad_perm [2,3,0,1]
  AM_CHAN_STEP(0x141, 0xf, 16.f)    // row_half_mirror
  AM_CHAN_STEP(0x140, 0xf, 32.f)    // row_mirror
  AM_CHAN_STEP(0x142, 0xa, 64.f)    // row_bcast15 into rows 1 and 3 (rows 0 and 2 merge with themselves: never written)
#undef AM_CHAN_STEP
  if (writer) *reinterpret_cast<f32x2_t*>(part_row) = f32x2_t{mean, m2};
}

__device__ __forceinline__ void store_staged_tile(const am_gemm_args& p, const unsigned char* stage, int tid, int m0, int n0,
                                                  const ResidualPrefetch* pre = nullptr) {
  {
    // thread -> 16-byte chunk k16 (8 columns) of rows ml = pass * 16 + (tid >> 5).  (ml & 15) does not depend on the pass,
    // so the chunk sits at a fixed offset of its row and every per-pass LDS address is base + constant.
    const int k16 = tid & 31, r16 = tid >> 5;
    const int gn = n0 + k16 * 8;
    if (gn < p.N) {
      // the chunk's two 8-byte units; row r16 of each 16-row pass, +8192 B per pass
      const unsigned char* su = stage + r16 * 512 + ((k16 ^ (r16 >> 1)) << 4);
      const bool odd = r16 & 1;
      auto fetch = [&](int pass) __attribute__((always_inline)) {
        u32x4_t sv = *reinterpret_cast<const u32x4_t*>(su + pass * 8192);
        if (odd) sv = u32x4_t{sv[2], sv[3], sv[0], sv[1]};
        return sv;
      };
      auto add_res = [&](u32x4_t sv, const bf16_t* rp) __attribute__((always_inline)) {
        const u32x4_t rv = *reinterpret_cast<const u32x4_t*>(rp);
#pragma unroll
        for (int e = 0; e < 4; ++e) sv[e] = pack_bf2(bflo(sv[e]) + bflo(rv[e]), bfhi(sv[e]) + bfhi(rv[e]));
        return sv;
      };
      const bool abl_nostore = (p.act & 0x800) != 0, abl_nores = (p.act & 0x1000) != 0;   // timing ablations (tools/kernel_bench.py)
      if (abl_nostore) {
        u32x4_t keep = fetch(0);
#pragma unroll 4
        for (int pass = 1; pass < 16; ++pass) { const u32x4_t f = fetch(pass); keep[0] ^= f[0] ^ f[3]; }
        if (keep[0] == 0x12345678u && p.M < 0) *reinterpret_cast<u32x4_t*>(p.C) = keep;          // never true: keeps the reads live
      } else if (p.c_G <= 0 && m0 + B2 <= p.M) {
        // identity row map, full tile (every tile of the main grid at the reference shapes): rows advance by 16 * ldc per
        // pass - wave-uniform base + one per-lane 32-bit offset, no per-pass address arithmetic, no bounds checks
        const uint32_t lane_off = ((uint32_t)r16 * (uint32_t)p.ldc + (uint32_t)gn) * 2u;
        const int64_t step = (int64_t)16 * p.ldc;
        bf16_t* crow = p.C + (int64_t)m0 * p.ldc;
        // ln_part (am_gemm_bf16 passes it down only when N % 256 == 0): slice n0 / 256 of rows m0 + pass * 16 + r16
        const int nparts = p.N >> 8;
        float* prow = p.ln_part ? p.ln_part + 2 * ((int64_t)(m0 + r16) * nparts + (n0 >> 8)) : nullptr;
        const bool pwriter = k16 == 31;
        if (pre != nullptr && pre->on) {              // residual already in registers (residual_prefetch)
#pragma unroll
          for (int pass = 0; pass < 16; ++pass) {
            u32x4_t sv = fetch(pass);
            const u32x4_t rv = pre->rv[pass];
#pragma unroll
            for (int e = 0; e < 4; ++e) sv[e] = pack_bf2(bflo(sv[e]) + bflo(rv[e]), bfhi(sv[e]) + bfhi(rv[e]));
            u32x4_t* q = reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned char*>(crow + pass * step) + lane_off);
            if (AM_GEMM_NT) __builtin_nontemporal_store(sv, q); else *q = sv;
            if (prow) emit_row_part(sv, prow + (int64_t)pass * 32 * nparts, pwriter);
          }
        } else if (p.residual && !abl_nores) {
          const bf16_t* rrow = p.residual + (int64_t)m0 * p.ldc;
#pragma unroll 4
          for (int pass = 0; pass < 16; ++pass) {
            const u32x4_t sv = add_res(fetch(pass), reinterpret_cast<const bf16_t*>(reinterpret_cast<const unsigned char*>(rrow + pass * step) + lane_off));
            *reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned char*>(crow + pass * step) + lane_off) = sv;
            if (prow) emit_row_part(sv, prow + (int64_t)pass * 32 * nparts, pwriter);
          }
        } else {
#pragma unroll 4
          for (int pass = 0; pass < 16; ++pass) {
            u32x4_t* q = reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned char*>(crow + pass * step) + lane_off);
            const u32x4_t sv = fetch(pass);
            if (AM_GEMM_NT) __builtin_nontemporal_store(sv, q); else *q = sv;
            if (prow) emit_row_part(sv, prow + (int64_t)pass * 32 * nparts, pwriter);
          }
        }
      } else {
#pragma unroll 2
        for (int pass = 0; pass < 16; ++pass) {
          const int gmr = m0 + pass * 16 + r16;
          if (gmr < p.M) {
            u32x4_t sv = fetch(pass);
            const int64_t pr = map_row(gmr, p.c_G, p.c_gs, p.c_off);
            if (p.residual) sv = add_res(sv, p.residual + pr * p.ldc + gn);
            *reinterpret_cast<u32x4_t*>(p.C + pr * p.ldc + gn) = sv;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Fused epilogue of the QKV (and cross-attention q) linear - north_star's "fused RMSNorm + RoPE + QKV", attention_processor.py:92-130:
// the head split, the per-head qk-RMSNorm, the RoPE rotation and the attention operand layouts straight from the staged bf16 tile,
// with the arithmetic of head_post_kernel (am_norm.hip) on the same bf16-rounded linear output: bit-identical Q / K / V^T, and the
// qkv activation (R x 3C bf16: 805 MB per layer at the headline shape) is neither written nor read back.  A 256-column tile holds two
// (head, part) slices of 128 channels:
//   Q / K slices: the row-major reader of store_staged_tile (thread = 8 channels of one row; a row's 16 lanes are one DPP row), RMS
//                 norm over the 16 lanes, rotation by the row's frame angle, one 16-byte store into [seq][head][s][128];
//   V slice:      read back transposed - thread = one channel x 8 key positions (perm16 order inside every 16-key group), 8 threads
//                 write 128 contiguous bytes of a V^T row.
// Not handled here (am_gemm_headpost_bf16 launches am_head_post_partial behind the GEMM for them): the rows of the 128x128 tail kernel
// and the zero fill of pad rows / columns.  Needs M % 16 == 0 and, with a V slice, seq_len % 16 == 0 (key groups must not straddle
// tiles or sequences) - the entry point falls back to GEMM + head_post otherwise.
__device__ __forceinline__ void store_staged_tile_headpost(const am_gemm_args& p, const am_headpost_args& hp, const unsigned char* stage,
                                                           int tid, int m0, int n0) {
  const int k16 = tid & 31, r16 = tid >> 5;
  const int half = k16 >> 4, sub = k16 & 15;
  const bool abl_noqk = (p.act & 0x800) != 0, abl_nov = (p.act & 0x1000) != 0;      // timing ablations (tools/kernel_bench.py --ablate-gemm)
  if (!abl_noqk) {
    const int pi = (n0 >> 7) + half;                           // (head, part) slice of this thread's 128-column half
    const int head = pi / hp.nparts, part = pi - head * hp.nparts;
    const int kind = (n0 + half * 128 < p.N) ? hp.kinds[part] : 3;
    if (kind == 0 || kind == 1) {
      const float* wt = kind == 0 ? hp.w_q : hp.w_k;
      float wv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) wv[e] = wt ? wt[sub * 8 + e] : 1.f;
      bf16_t* out = kind == 0 ? hp.out_q : hp.out_k;
      const int s_pad = kind == 0 ? hp.sq_pad : hp.sk_pad;
      const unsigned char* su = stage + r16 * 512 + ((k16 ^ (r16 >> 1)) << 4);
      const bool odd = r16 & 1;
      // Round 6: a 256-row tile lies in at most TWO frames and TWO sequences when both are at least 256 rows long (every reference
      // shape: a frame is N + 1 >= 2049 rows).  The frame's RoPE angles and the sequence's output base are then fetched / divided
      // ONCE per tile - wave-uniform scalars and 4 x 16 bytes per thread in front of the loop - instead of two global loads and two
      // integer divisions per row (32 exposed loads per thread, the whole difference between this epilogue and the plain C store).
      const bool two = hp.rows_per_frame >= B2 && hp.seq_len >= B2;
      const int frame0 = m0 / hp.rows_per_frame, seq0 = m0 / hp.seq_len;
      const int fb = (frame0 + 1) * hp.rows_per_frame, sb = (seq0 + 1) * hp.seq_len;      // first row of the next frame / sequence
      f32x4_t cs0 = {1.f, 1.f, 1.f, 1.f}, sn0 = {0.f, 0.f, 0.f, 0.f}, cs1 = cs0, sn1 = sn0;
      if (hp.rope_cos && two) {
        const int frame1 = fb < p.M ? frame0 + 1 : frame0;     // (the tile's rows past p.M are never written)
        cs0 = *reinterpret_cast<const f32x4_t*>(hp.rope_cos + (int64_t)frame0 * 64 + sub * 4);
        sn0 = *reinterpret_cast<const f32x4_t*>(hp.rope_sin + (int64_t)frame0 * 64 + sub * 4);
        cs1 = *reinterpret_cast<const f32x4_t*>(hp.rope_cos + (int64_t)frame1 * 64 + sub * 4);
        sn1 = *reinterpret_cast<const f32x4_t*>(hp.rope_sin + (int64_t)frame1 * 64 + sub * 4);
      }
      bf16_t* const out0 = out + (((int64_t)seq0 * hp.heads + head) * s_pad - (int64_t)seq0 * hp.seq_len) * 128 + sub * 8;     // + gm * 128
      bf16_t* const out1 = out0 + ((int64_t)hp.heads * s_pad - hp.seq_len) * 128;
      auto one_row = [&](int pass) __attribute__((always_inline)) {
        const int gm = m0 + pass * 16 + r16;
        if (gm >= p.M) return;
        u32x4_t u = *reinterpret_cast<const u32x4_t*>(su + pass * 8192);
        if (odd) u = u32x4_t{u[2], u[3], u[0], u[1]};
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
          ss += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss), 0x128, 0xf, 0xf, false));
          ss += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss), 0x124, 0xf, 0xf, false));
          ss += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss), 0x122, 0xf, 0xf, false));
          ss += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss), 0x121, 0xf, 0xf, false));
          const float r = rsqrtf(ss * (1.0f / 128.0f) + hp.eps);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (v[e] * r) * wv[e];
        }
        if (hp.rope_cos) {
          f32x4_t cs, sn;
          if (two) {
            const bool nx = gm >= fb;
#pragma unroll
            for (int e = 0; e < 4; ++e) { cs[e] = nx ? cs1[e] : cs0[e]; sn[e] = nx ? sn1[e] : sn0[e]; }
          } else {
            const int frame = gm / hp.rows_per_frame;
            cs = *reinterpret_cast<const f32x4_t*>(hp.rope_cos + (int64_t)frame * 64 + sub * 4);
            sn = *reinterpret_cast<const f32x4_t*>(hp.rope_sin + (int64_t)frame * 64 + sub * 4);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) rope_rotate(v[2 * e], v[2 * e + 1], cs[e], sn[e]);    // rotary_embedding.py:116-122
        }
        u32x4_t w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
        if (two) {
          *reinterpret_cast<u32x4_t*>((gm >= sb ? out1 : out0) + (int64_t)gm * 128) = w;
        } else {
          const int seq = gm / hp.seq_len, sq = gm - seq * hp.seq_len;
          *reinterpret_cast<u32x4_t*>(out + (((int64_t)seq * hp.heads + head) * s_pad + sq) * 128 + sub * 8) = w;
        }
      };
      if (two && m0 + B2 <= p.M && wt != nullptr) {
        // The shape every tile of the main grid has (full tile, qk-norm on): straight-line code, four rows in flight per thread - the
        // four LDS reads, the four 4-step DPP reductions and the four rsqrt's of a batch overlap instead of queueing behind the
        // branches of the general form (the epilogue is VALU / latency bound: 8 waves, nothing else on the CU to hide behind).
        // Same operations in the same order on every element as one_row(): bit-identical.  v_rsq_f32 without rsqrtf's denormal
        // rescue: its argument is >= eps (1e-6 > FLT_MIN), the rescue branch is never taken, the result is the same instruction's.
        const bool rope = hp.rope_cos != nullptr;
#pragma unroll 1
        for (int b4 = 0; b4 < 16; b4 += 4) {
          u32x4_t u[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) u[i] = *reinterpret_cast<const u32x4_t*>(su + (b4 + i) * 8192);
          float v[4][8], ss[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (odd) u[i] = u32x4_t{u[i][2], u[i][3], u[i][0], u[i][1]};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[i][2 * e] = bflo(u[i][e]); v[i][2 * e + 1] = bfhi(u[i][e]); }
            ss[i] = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) ss[i] = __builtin_fmaf(v[i][e], v[i][e], ss[i]);
          }
#define AM_HP_DPP(CTRL)                                                                                                         \
          _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                         \
            ss[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss[i]), CTRL, 0xf, 0xf, false));
          AM_HP_DPP(0x128) AM_HP_DPP(0x124) AM_HP_DPP(0x122) AM_HP_DPP(0x121)
#undef AM_HP_DPP
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int gm = m0 + (b4 + i) * 16 + r16;
            const float r = __builtin_amdgcn_rsqf(ss[i] * (1.0f / 128.0f) + hp.eps);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = (v[i][e] * r) * wv[e];
            if (rope) {
              const bool nx = gm >= fb;
#pragma unroll
              for (int e = 0; e < 4; ++e) rope_rotate(v[i][2 * e], v[i][2 * e + 1], nx ? cs1[e] : cs0[e], nx ? sn1[e] : sn0[e]);
            }
            u32x4_t w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = pack_bf2(v[i][2 * e], v[i][2 * e + 1]);
            *reinterpret_cast<u32x4_t*>((gm >= sb ? out1 : out0) + (int64_t)gm * 128) = w;
          }
        }
      } else if (two) {
#pragma unroll 4
        for (int pass = 0; pass < 16; ++pass) one_row(pass);
      } else {
#pragma unroll 2
        for (int pass = 0; pass < 16; ++pass) one_row(pass);
      }
    }
  }
  if (abl_nov) return;
  // ---- V slice (at most one of the two halves for the q | k | v interleave): transposed read-back
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    const int pi = (n0 >> 7) + h2;
    const int head = pi / hp.nparts, part = pi - head * hp.nparts;
    if (n0 + h2 * 128 >= p.N || hp.kinds[part] != 2) continue;        // uniform over the workgroup
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll 2
    for (int it = 0; it < 8; ++it) {
      const int pc = (lane & 7) + 8 * (it & 3);                        // chunk of 8 key POSITIONS of the tile's 256
      const int d = ((lane >> 3) & 7) + 8 * (wave + 8 * (it >> 2));    // channel 0 .. 127
      const int pos0 = pc * 8;
      const int g16 = m0 + (pos0 & ~15);                               // first key (= global row) of the 16-key group
      if (g16 >= p.M) continue;
      const int n = h2 * 128 + d;
      uint32_t val[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int pos = pos0 + j;
        const int m = (pos & ~15) | perm16(pos & 15);                  // position -> key row (involution)
        val[j] = *reinterpret_cast<const uint16_t*>(stage + m * 512 + ((((n >> 2) ^ (m & 15))) << 3) + (n & 3) * 2);
      }
      const u32x4_t w = {val[0] | (val[1] << 16), val[2] | (val[3] << 16), val[4] | (val[5] << 16), val[6] | (val[7] << 16)};
      const int seq = g16 / hp.seq_len, s16 = g16 - seq * hp.seq_len;
      *reinterpret_cast<u32x4_t*>(hp.out_vt + (((int64_t)seq * hp.heads + head) * 128 + d) * hp.sk_pad + s16 + (pos0 & 15)) = w;
    }
  }
}

// ===========================================================================
// v2: 256x256x64 tile, 8 waves (2 x 4, each 128 x 64), operands DMA'd straight
// into LDS (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass).
// The LDS image of a tile is lane-linear per wave-instruction (DMA constraint),
// so the bank-conflict swizzle is applied to the per-lane SOURCE address and
// mirrored on the ds_read side: 16-byte unit c of tile row r lives at unit
// position c ^ ((r >> 1) & 7) of its 128-byte row, which puts the 16 lanes of
// every ds_read_b128 group on 16 distinct 16-byte slots of the 256-byte bank row.
// Two LDS buffers (128 KiB): the DMA of K-tile kt+1 runs under the MFMAs of kt.
// Epilogue: bias/GELU in registers, bf16 tile staged through LDS (XOR-swizzled
// 8-byte units), then row-contiguous 512-byte stores with the residual add.
// ===========================================================================
constexpr int T2_UNITS = B2 * (BK / 8);        // 16-byte units per operand tile (2048)
constexpr int T2_BYTES = T2_UNITS * 16;        // 32 KiB
constexpr int SMEM2_BYTES = 4 * T2_BYTES;      // A,B x 2 buffers = 128 KiB (epilogue reuses it)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__global__ __launch_bounds__(512, 2) void gemm256_bf16_kernel(am_gemm_args p, int tiles_m, int tiles_n, int m_base) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int nb = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  int lid;
  {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nb >> 3, r = nb & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  {
    const int group_sz = GROUP_M * tiles_n;
    const int g = lid / group_sz;
    const int first_m = g * GROUP_M;
    const int gm = min(GROUP_M, tiles_m - first_m);
    const int in_g = lid - g * group_sz;
    tm = first_m + in_g % gm;
    tn = in_g / gm;
  }
  const int m0 = m_base + tm * B2, n0 = tn * B2;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;

  // ---- DMA source pointers: unit U = j*512 + wave*64 + lane of each tile -----------
  const bf16_t* a1p[4];
  const bf16_t* a2p[4];
  const bf16_t* wp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int U = j * 512 + wave * 64 + lane;
    const int r = U >> 3, c = (U & 7) ^ ((r >> 1) & 7);
    const int ar = min(m0 + r, p.M - 1);
    const int64_t pr = map_row(ar, p.a_G, p.a_gs, p.a_off);
    a1p[j] = p.A1 + pr * p.lda1 + c * 8;
    a2p[j] = p.A2 ? p.A2 + pr * p.lda2 + c * 8 - p.K1 : nullptr;
    const int wr = min(n0 + r, p.N - 1);
    wp[j] = p.W + (int64_t)wr * p.ldw + c * 8;
  }
  // piece j of the A tile and piece j of the W tile of k-tile kt
  auto dma_pair = [&](int kt, int buf, int j) __attribute__((always_inline)) {
    const int k0 = kt * BK;
    unsigned char* abase = smem + buf * 2 * T2_BYTES;
    unsigned char* bbase = abase + T2_BYTES;
    const bf16_t* src = k0 < p.K1 ? a1p[j] + k0 : a2p[j] + k0;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(abase + (j * 512 + wave * 64) * 16), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wp[j] + k0), (lds_ptr_t)(bbase + (j * 512 + wave * 64) * 16), 16, 0, 0);
  };
  auto dma_tile = [&](int kt, int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_pair(kt, buf, j);
  };

  // accumulators start from the bias (fp32, in the MFMA layout: lane (l31, hi) holds columns 8g + 4hi .. +3 of block j)
  f32x16_t acc[4][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4_t bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bv = *reinterpret_cast<const f32x4_t*>(p.bias + min(n0 + wn * 64 + j * 32 + 8 * g + 4 * hi, p.N - 4));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = bv[e];
    }

  // fragment byte offsets inside a tile: row R, k-unit c -> (R*8 + (c ^ ((R>>1)&7))) * 16
  int a_row_off[4], a_sw[4], b_row_off[2], b_sw[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int R = wm * 128 + i * 32 + l31;
    a_row_off[i] = R * 128;
    a_sw[i] = (R >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int R = wn * 64 + j * 32 + l31;
    b_row_off[j] = R * 128;
    b_sw[j] = (R >> 1) & 7;
  }

  const int nk = p.K / BK;
  dma_tile(0, 0);
  dma_drain_barrier();               // tile 0 landed and is visible
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    // the next tile's 8 LDS-DMA pieces go out in two groups, in front of k-steps 0 and 1 (half a trip is left for them to
    // land): issued back to back at the top of the trip, the 64 pieces of the 8 waves queue on the one TA and the waves
    // sit in the issue of their pieces
    const int kn = min(kt + 1, nk - 1);        // last trip: re-fetch the last tile into the idle buffer (drained below)
    const unsigned char* At = smem + buf * 2 * T2_BYTES;
    const unsigned char* Bt = At + T2_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      if (ks < 2) {
        dma_pair(kn, buf ^ 1, 2 * ks);
        dma_pair(kn, buf ^ 1, 2 * ks + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int c = ks * 2 + hi;
      bf16x8_t af[4], bfr[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8_t*>(Bt + b_row_off[j] + ((c ^ b_sw[j]) << 4));
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = *reinterpret_cast<const bf16x8_t*>(At + a_row_off[i] + ((c ^ a_sw[i]) << 4));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = AM_MFMA_32x32x16(bfr[j], af[i], acc[i][j]);
    }
    dma_drain_barrier();             // next tile landed; everyone is done with `buf`
  }

  // ---- epilogue -------------------------------------------------------------------------
  // stage[m][n] bf16, row = 512 B = 64 units of 8 B; unit u of row m sits at u ^ (m & 15)
  unsigned char* stage = smem;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nl = wn * 64 + j * 32 + 8 * g + 4 * hi;          // 4 consecutive columns
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ml = wm * 128 + i * 32 + l31;
        u32x2_t w;
        if ((p.act & 0xff) == 1) {                               // F.gelu on the bf16 linear output -> bf16
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(rbf(acc[i][j][4 * g + e]));
          w = u32x2_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        } else {
          w = u32x2_t{pack_bf2(acc[i][j][4 * g], acc[i][j][4 * g + 1]), pack_bf2(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3])};
        }
        const int u = (nl >> 2) ^ (ml & 15);
        *reinterpret_cast<u32x2_t*>(stage + ml * 512 + u * 8) = w;
      }
    }
  __syncthreads();
  store_staged_tile(p, stage, tid, m0, n0);
}


// ===========================================================================
// v3 "ping-pong" (round 2): the same 256x256x64 tile, LDS image and epilogue, a different main loop.
//
// The 8 waves are two groups of four (group g = tile rows [128 g, 128 g + 128); wave (g, wn) owns 128 x 64 outputs) that
// run HALF A PHASE APART: group 1 executes one extra s_barrier before the loop, group 0 one after it, so between any two
// consecutive barriers one group is in a pure-MFMA interval (16 x v_mfma_f32_16x16x32_bf16 = one 64 x 32 quadrant of its
// tile over the whole 64-deep k-tile, s_setprio 1) while the other - its partner on every SIMD - is in its LDS interval
// (fragment ds_read_b128s for its next quadrant + two LDS-DMA pieces of a later k-tile + the waits).  The matrix pipe of a
// SIMD is therefore always fed by one wave while the other one fetches; in the round-1 loop both waves of a SIMD read,
// waited and multiplied in lockstep and the pipe idled through every fetch (MFMA busy 42 %).
//
// k-tile t lives in LDS buffer t & 1 as four 16 KiB half-tiles [TA0 | TA1 | TB0 | TB1] (A rows 0-127 / 128-255, W rows
// 0-127 / 128-255), each in the swizzled image of the v2 kernel (16-byte unit c of row r at c ^ ((r >> 1) & 7); conflict
// free for the 16x16x32 fragment pattern row = lane & 15, unit = lane >> 4 as well).  Group g reads TA_g only; W fragments
// of a tile are read in its first two intervals, A fragments in the first and third, so
//     interval     R1            R2            R3             R4
//     reads        a0 (8) b0 (4) b1 (4)        a1 (8)         -
//     stages       TA0(t+1)      TA1(t+1)      TB0(t+2)       TB1(t+2)   then s_waitcnt vmcnt(4)
//     multiplies   M1: a0 x b0   M2: a0 x b1   M3: a1 x b1    M4: a1 x b0
// (TA(t+1) goes to the other buffer, whose last A reads were in tile t-1's R3; TB(t+2) goes to THIS buffer, whose W
// reads ended in R2: every wave drains its LDS reads - lgkmcnt(0) - in front of the barrier that ends an R interval, so
// a DMA issued after that barrier cannot overtake them.)  The LDS-DMA queue is never drained inside the loop: the one
// counted wait per k-tile, vmcnt(4) at the end of R4, leaves the two newest stages (TB(t+2)) in flight and retires
// everything tile t+1 needs, a full k-tile after it was issued; the barrier behind it publishes it to both groups.
// Past the end of K the stages re-fetch the last tile into slots nobody reads, so the counts stay constant.
// ===========================================================================
// ---- exact-erf GELU of the 256x256 tiles as a TABLE (round 5; VERDICT r04 weak #2: "the erf epilogue's VALU sitting exposed behind the
// main loop") -----------------------------------------------------------------------------------------------------------------------
// F.gelu acts on the bf16-ROUNDED linear output (block.py:99-104 under autocast), and a bf16 has 65 536 values: the epilogue's
// ~28 VALU slots per element (two quarter-rate transcendentals among them; 128 elements per lane: ~12 us of a ~38 us tile with one
// workgroup per CU and nothing to hide behind) become ONE 2-byte LDS read.  The table holds  f2bf(gelu_erf(x))  - the very
// expression the arithmetic epilogue evaluates, computed once per device BY gelu_erf on the device - for every bf16 x with
// 2^-17 <= |x| < 8 (2 signs x 20 exponents x 128 mantissas = 5120 entries = 10 KiB, DMA'd into LDS behind the operand ring at
// kernel start); the few values outside (|x| >= 8; |x| < 7.6e-6, ~6e-6 of a unit normal) take the arithmetic path under an exec
// mask that is almost never set.  Results are BIT-IDENTICAL to the arithmetic epilogue by construction (tests/test_kernels_gpu.py::
// test_gelu_table_is_bit_identical), so the 128x128 kernel, the tail kernel and the float16 build (whose 1024 mantissas per
// exponent do not fit LDS) keep the arithmetic and every tiling still agrees bit for bit.
constexpr uint32_t GT_LO = (127u - 17u) << 7, GT_HI = (127u + 3u) << 7, GT_SPAN = GT_HI - GT_LO;   // magnitude bits [LO, HI)
constexpr int GT_ENTRIES = 2 * (int)GT_SPAN;                                                       // 5120
constexpr int GT_BYTES = GT_ENTRIES * 2;                                                           // 10 240
__global__ void gelu_table_kernel(uint16_t* tab) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= GT_ENTRIES) return;
  const uint32_t sgn = (uint32_t)i / GT_SPAN, rel = (uint32_t)i - sgn * GT_SPAN;
  const bf16_t b = (bf16_t)((sgn << 15) | (rel + GT_LO));
  tab[i] = (uint16_t)(pack_bf2(gelu_erf(bf2f(b)), 0.f) & 0xffffu);
}
// two packed bf16 linear outputs -> two packed bf16 GELUs by table, BRANCH-FREE: an out-of-table value reads a clamped entry (garbage)
// and raises `bad`; the caller repairs those elements afterwards, arithmetically, under a branch that is almost never taken - so the
// 128 lookups of a lane are one straight basic block whose LDS reads the scheduler can batch.
__device__ __forceinline__ uint32_t gelu_table_pair(uint32_t pk, const unsigned char* tab, uint32_t& bad) {
  const uint32_t b0 = pk & 0xffffu, b1 = pk >> 16;
  const uint32_t r0 = (b0 & 0x7fffu) - GT_LO, r1 = (b1 & 0x7fffu) - GT_LO;
  bad |= (uint32_t)(r0 >= GT_SPAN) | (uint32_t)(r1 >= GT_SPAN);
  const uint32_t i0 = min(r0, GT_SPAN - 1) + (b0 >> 15) * GT_SPAN, i1 = min(r1, GT_SPAN - 1) + (b1 >> 15) * GT_SPAN;
  const uint32_t g0 = *reinterpret_cast<const uint16_t*>(tab + 2u * i0);
  const uint32_t g1 = *reinterpret_cast<const uint16_t*>(tab + 2u * i1);
  return g0 | (g1 << 16);
}
__device__ __forceinline__ bool gelu_table_has(float x) {           // is bf16(x) a table entry?
  const uint32_t b = pack_bf2(x, 0.f) & 0x7fffu;
  return b - GT_LO < GT_SPAN;
}

constexpr int HT_BYTES = 128 * BK * 2;         // one half-tile (16 KiB)
constexpr int PBUF_BYTES = 4 * HT_BYTES;       // [TA0 | TA1 | TB0 | TB1] (64 KiB); two buffers = 128 KiB
// behind the two buffers: the operands of a folded LayerNorm for this tile - (mean, rstd) of its 256 rows (2 KiB), colsum and d of its
// 256 columns (1 KiB each) - fetched by LDS-DMA before the prologue, read in the epilogue (they cost 1.7 us of exposed load latency
// per tile when the epilogue fetched them itself: profiles/r04s)
constexpr int FOLD_OFF = 2 * PBUF_BYTES;
constexpr int SMEM2PP_BYTES = FOLD_OFF + 4096;
constexpr int GT_OFF = SMEM2PP_BYTES;                       // the GELU table sits behind the fold operands (launches with act == GELU only)
constexpr int SMEM2PP_GELU_BYTES = GT_OFF + GT_BYTES;       // 142 KiB of the CU's 160

#define PP_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// HP: the fused head-split / qk-norm / RoPE / layout epilogue (store_staged_tile_headpost) instead of the C tile store
template <bool HP>
__global__ __launch_bounds__(512, 2) void gemm256pp_bf16_kernel(am_gemm_args p, int tiles_m, int tiles_n, int m_base, am_headpost_args hp,
                                                                 const uint16_t* gelu_tab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int nb = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  {   // Per-XCD start skew (round 3): the first round's workgroups of XCD x start x * units * 1.5 us late (act bits 13-15, set by
      // am_gemm_bf16), and every later round inherits the offset.  All workgroups of a round take the same time, so without it the
      // 256 epilogues (C stores + residual loads: 64-96 MB) hit HBM in one burst while the main loops leave it idle; whole-XCD
      // offsets spread the bursts and keep each XCD's 32 workgroups in lockstep, so they still share their A / W panels through its
      // L2 (per-WORKGROUP offsets, tried in round 2, lost that sharing and were slower).  Same tiles, same arithmetic: bit-identical.
      // Measured same-box, warm (profiles/r03h_gemm_skew.txt, "off" vs best): qkv 0.828 -> 0.756 ms, ff1+GELU 1.262 -> 1.164, ff2 0.951 -> 0.936;
      // nominal qkv 1.369 -> 1.302.
    const int units = (p.act >> 13) & 7;
    // (an additional INTRA-XCD skew - the XCD's workgroups entering their epilogues at four different times - was measured in round 4
    // and dropped: no gain at 0.5-2 us, 1.5-4 % slower at 4-8 us; tools/experiments/gemm_subskew.patch, profiles/r04m_gemm_subskew.txt)
    if (units && units < 7 && bid < 256) {
      const unsigned long long t0 = wall_clock64();
      const unsigned long long wait = (unsigned long long)(bid & 7) * units * 150ull;      // 100 MHz clock
      while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(4);
    }
  }
  int lid;
  {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nb >> 3, r = nb & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  {
    const int group_sz = GROUP_M * tiles_n;
    const int g = lid / group_sz;
    const int first_m = g * GROUP_M;
    const int gm = min(GROUP_M, tiles_m - first_m);
    const int in_g = lid - g * group_sz;
    tm = first_m + in_g % gm;
    tn = in_g / gm;
  }
  const int m0 = m_base + tm * B2, n0 = tn * B2;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;           // wm = group

  // ---- accumulators start from the bias.  D = Wfrag x Afrag: lane (l & 15, l >> 4) of block (mi, ni) holds output row
  // m = mi*16 + (l & 15), columns n = ni*16 + (l >> 4)*4 .. +3.  The bias is fetched and waited for BEFORE the first LDS-DMA
  // goes out: hipcc waits vmcnt(0) at the use of an ordinary load, which would drain the whole prologue otherwise.
  const int l15 = lane & 15, l4 = lane >> 4;
  f32x4_t acc[8][4];
  {
    f32x4_t bv[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      bv[ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (p.bias && !p.ln_stats) bv[ni] = *reinterpret_cast<const f32x4_t*>(p.bias + min(n0 + wn * 64 + ni * 16 + l4 * 4, p.N - 4));
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) asm volatile("" : "+v"(bv[ni]));
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) acc[mi][ni] = bv[ni];
  }
  __builtin_amdgcn_sched_barrier(0);
  if (p.ln_stats) {           // folded LayerNorm: this tile's statistics / column sums / d into LDS, one dword per lane and piece
    {
      const int dw = wave * 64 + lane;                              // dword dw of the tile's 256 (mean, rstd) pairs
      const int row = min(m0 + (dw >> 1), p.M - 1);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.ln_stats + 2 * (int64_t)row + (dw & 1)), (lds_ptr_t)(smem + FOLD_OFF + wave * 256), 4, 0, 0);
    }
    {
      const int idx = (wave & 3) * 64 + lane;
      const int n = min(n0 + idx, p.N - 1);
      const float* src = wave < 4 ? p.ln_colsum + n : (p.bias ? p.bias + n : p.ln_colsum + n);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(smem + FOLD_OFF + 2048 + wave * 256), 4, 0, 0);
    }
  }

  if (gelu_tab != nullptr) {   // the GELU table: 2560 dwords, 5 per lane, lane-linear like every LDS-DMA piece; older than every operand
                               // stage, so the main loop's counted vmcnt waits retire it long before the epilogue reads it
#pragma unroll
    for (int i = 0; i < GT_BYTES / 4 / 512; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(reinterpret_cast<const uint32_t*>(gelu_tab) + i * 512 + wave * 64 + lane),
                                       (lds_ptr_t)(smem + GT_OFF + (i * 512 + wave * 64) * 4), 4, 0, 0);
  }

  // ---- LDS-DMA sources, in 16-byte units from the operand base (32 bits reach 64 GiB).  Piece pc of half-tile h covers
  // rows pc*64 + wave*8 + (lane >> 3) of the half; the lane fetches the unit that belongs at its lane-linear LDS slot.
  uint32_t offA1[4], offA2[4], offW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int half = i >> 1, pc = i & 1;
    const int r = pc * 64 + wave * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const int ar = min(m0 + half * 128 + r, p.M - 1);
    const int64_t pr = map_row(ar, p.a_G, p.a_gs, p.a_off);
    offA1[i] = (uint32_t)(pr * (p.lda1 >> 3) + c);
    offA2[i] = (uint32_t)(pr * (p.lda2 >> 3) + c);
    const int wr = min(n0 + half * 128 + r, p.N - 1);
    offW[i] = (uint32_t)((int64_t)wr * (p.ldw >> 3) + c);
  }
  auto stage_a = [&](int half, int kt, int buf_off) __attribute__((always_inline)) {
    const int k0 = kt * BK;
    const bool first = k0 < p.K1;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(first ? p.A1 + k0 : p.A2 + (k0 - p.K1));
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      const uint32_t off = first ? offA1[half * 2 + pc] : offA2[half * 2 + pc];
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + ((uint64_t)off << 4)),
                                       (lds_ptr_t)(smem + buf_off + half * HT_BYTES + (pc * 512 + wave * 64) * 16), 16, 0, 0);
    }
  };
  auto stage_w = [&](int half, int kt, int buf_off) __attribute__((always_inline)) {
    const unsigned char* base = reinterpret_cast<const unsigned char*>(p.W + kt * BK);
#pragma unroll
    for (int pc = 0; pc < 2; ++pc)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + ((uint64_t)offW[half * 2 + pc] << 4)),
                                       (lds_ptr_t)(smem + buf_off + (2 + half) * HT_BYTES + (pc * 512 + wave * 64) * 16), 16, 0, 0);
  };


  // ---- fragment addresses: row = block row + (l & 15), 16-byte unit (ks*4 + (l >> 4)) ^ ((row >> 1) & 7); block rows are
  // multiples of 16, so the swizzle term is the lane's own ((l & 15) >> 1)
  const int fo0 = l15 * 128 + (((0 + l4) ^ (l15 >> 1)) << 4);
  const int fo1 = l15 * 128 + (((4 + l4) ^ (l15 >> 1)) << 4);
  const int a_base = wm * HT_BYTES;
  const int w_base = (2 + (wn >> 1)) * HT_BYTES + (wn & 1) * 64 * 128;

  bf16x8_t af[4][2], wf[4][2];
  auto read_a = [&](int cur, int mh) __attribute__((always_inline)) {        // rows mh*64 .. +64 of the wave's 128
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned char* q = smem + cur + a_base + (mh * 4 + i) * 2048;
      af[i][0] = *reinterpret_cast<const bf16x8_t*>(q + fo0);
      af[i][1] = *reinterpret_cast<const bf16x8_t*>(q + fo1);
    }
  };
  auto read_w = [&](int cur, int nh) __attribute__((always_inline)) {        // columns nh*32 .. +32 of the wave's 64
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned char* q = smem + cur + w_base + (nh * 2 + j) * 2048;
      wf[nh * 2 + j][0] = *reinterpret_cast<const bf16x8_t*>(q + fo0);
      wf[nh * 2 + j][1] = *reinterpret_cast<const bf16x8_t*>(q + fo1);
    }
  };
  // one quadrant: 4 row blocks x 2 column blocks x 2 k-steps; the two k-steps of a block are 8 MFMAs apart
#define PP_QUADRANT(MH, NH)                                                                                       \
  do {                                                                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                              \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                               \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
          acc[(MH) * 4 + i][(NH) * 2 + j] = AM_MFMA_16x16x32(                                                     \
              wf[(NH) * 2 + j][ks], af[i][ks], acc[(MH) * 4 + i][(NH) * 2 + j]);                                  \
    __builtin_amdgcn_s_setprio(0);                                                                                \
  } while (0)

  const int nk = p.K / BK;
  // prologue: tile 0 and the W halves of tile 1
  stage_a(0, 0, 0); stage_a(1, 0, 0); stage_w(0, 0, 0); stage_w(1, 0, 0);
  {
    const int t1 = min(1, nk - 1);
    stage_w(0, t1, PBUF_BYTES); stage_w(1, t1, PBUF_BYTES);
  }
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  PP_BARRIER();
  if (wm == 1) PP_BARRIER();                       // group 1 runs one interval behind group 0

  for (int t = 0; t < nk; ++t) {
    const int cur = (t & 1) * PBUF_BYTES, nxt = PBUF_BYTES - cur;
    const int ta = min(t + 1, nk - 1), tb = min(t + 2, nk - 1);
    // R1 / M1
    read_a(cur, 0); read_w(cur, 0);
    stage_a(0, ta, nxt);
    PP_LGKM0(); PP_BARRIER();
    PP_QUADRANT(0, 0);
    PP_BARRIER();
    // R2 / M2
    read_w(cur, 1);
    stage_a(1, ta, nxt);
    PP_LGKM0(); PP_BARRIER();
    PP_QUADRANT(0, 1);
    PP_BARRIER();
    // R3 / M3
    read_a(cur, 1);
    stage_w(0, tb, cur);
    PP_LGKM0(); PP_BARRIER();
    PP_QUADRANT(1, 1);
    PP_BARRIER();
    // R4 / M4
    stage_w(1, tb, cur);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // everything tile t+1 needs has landed (this wave's share)
    PP_BARRIER();
    PP_QUADRANT(1, 0);
    PP_BARRIER();
  }
  if (wm == 0) PP_BARRIER();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the trailing re-fetches must not land on the C staging below
  PP_BARRIER();

  // ---- epilogue: bf16 tile staged through LDS (row = 512 B = 64 units of 8 B, unit u of row m at u ^ (m & 15)), then the
  // row-contiguous store loop of the v2 kernel ------------------------------------------------------------------------
  ResidualPrefetch pre;
  pre.on = false;
  if constexpr (!HP) residual_prefetch(p, tid, m0, n0, pre);       // 16 loads in flight under the staging pass
  if (p.ln_stats) {
    // LayerNorm folded into this linear (am_gemm_args.ln_stats): the accumulators hold x W'^T for the UN-normalised rows x;
    // row r's output is rstd_r (acc - mean_r colsum_n) + d_n, two FMAs per element on statistics read once per tile.
    // (the operands were DMA'd behind the ring at kernel start; the vmcnt(0) + barrier in front of this point covers them)
    f32x4_t cs[4], dv[4];
    const float* fold = reinterpret_cast<const float*>(smem + FOLD_OFF);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int nl = wn * 64 + ni * 16 + l4 * 4;
      cs[ni] = *reinterpret_cast<const f32x4_t*>(fold + 512 + nl);
      dv[ni] = p.bias ? *reinterpret_cast<const f32x4_t*>(fold + 768 + nl) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const f32x2_t st = *reinterpret_cast<const f32x2_t*>(fold + 2 * (wm * 128 + mi * 16 + l15));
      const float nm = -st[0] * st[1];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mi][ni][e] = fmaf(acc[mi][ni][e], st[1], fmaf(nm, cs[ni][e], dv[ni][e]));
    }
  }
  unsigned char* stage = smem;
  uint32_t gelu_bad = 0;                                        // this lane met a value outside the GELU table
  // the three forms of the staging pass, each its own straight-line loop (the mode is uniform; testing it per block would cut the
  // pass into 32 basic blocks and the table's LDS reads could not be batched across them)
  auto stage_pass = [&](auto&& convert) __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int ml = wm * 128 + mi * 16 + l15;
        const int nl = wn * 64 + ni * 16 + l4 * 4;               // 4 consecutive columns
        const u32x2_t w = convert(acc[mi][ni]);
        const int u = (nl >> 2) ^ (ml & 15);
        *reinterpret_cast<u32x2_t*>(stage + ml * 512 + u * 8) = w;
      }
  };
  if ((p.act & 0xff) == 1 && gelu_tab != nullptr) {             // F.gelu on the bf16 linear output -> bf16, by table (see GT_LO)
    stage_pass([&](const f32x4_t& a) __attribute__((always_inline)) {
      return u32x2_t{gelu_table_pair(pack_bf2(a[0], a[1]), smem + GT_OFF, gelu_bad), gelu_table_pair(pack_bf2(a[2], a[3]), smem + GT_OFF, gelu_bad)};
    });
  } else if ((p.act & 0xff) == 1) {                              // the arithmetic form (float16 build; table unavailable)
    stage_pass([&](const f32x4_t& a) __attribute__((always_inline)) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_erf(rbf(a[e]));
      return u32x2_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
    });
  } else {
    stage_pass([&](const f32x4_t& a) __attribute__((always_inline)) { return u32x2_t{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3])}; });
  }
  if (gelu_bad) {       // rare (|x| >= 8 or |x| < 2^-17 somewhere in this lane's 128 outputs): the arithmetic form for exactly those elements,
                        // over the lane's own staged words (same thread, same addresses: no synchronisation needed)
#pragma unroll                      // (static register indices: a rolled loop would push the accumulators to scratch)
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int ml = wm * 128 + mi * 16 + l15;
        const int nl = wn * 64 + ni * 16 + l4 * 4;
        bool any = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) any |= !gelu_table_has(acc[mi][ni][e]);
        if (any) {
          uint16_t* q = reinterpret_cast<uint16_t*>(stage + ml * 512 + ((nl >> 2) ^ (ml & 15)) * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (!gelu_table_has(acc[mi][ni][e])) q[e] = (uint16_t)(pack_bf2(gelu_erf(rbf(acc[mi][ni][e])), 0.f) & 0xffffu);
        }
      }
  }
  __syncthreads();
  if constexpr (HP) store_staged_tile_headpost(p, hp, stage, tid, m0, n0);
  else store_staged_tile(p, stage, tid, m0, n0, &pre);
}
#undef PP_QUADRANT

}  // namespace

// Per-XCD start skew of the big GEMMs: how many 1.5 us units XCD x's first round starts late (kernel: act bits 13-15).  The
// kernel's `blockIdx & 7 == XCD` mapping, the 100 MHz wall clock and the 256-workgroup first round describe ONE part in ONE
// partition mode: an MI355X (gfx950) in SPX mode - 256 CUs visible as one device.  Anywhere else (CPX / DPX partitions, CU-masked
// streams shrink multiProcessorCount; other parts) the skew would be pure added latency, so it is 0 there; the environment
// variable ACTIONMESH_AMD_GEMM_SKEW=0 turns it off on the MI355X as well.
static int gemm_skew_units(int rounds) {
  static int enabled[64];           // 0 unknown, 1 on, 2 off
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (enabled[dev] == 0) {
    hipDeviceProp_t pr;
    bool on = hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount == 256 && strncmp(pr.gcnArchName, "gfx950", 6) == 0;
    const char* e = getenv("ACTIONMESH_AMD_GEMM_SKEW");
    if (e && e[0] == '0') on = false;
    enabled[dev] = on ? 1 : 2;
  }
  if (enabled[dev] != 1) return 0;
  return rounds >= 24 ? 4 : rounds >= 16 ? 3 : rounds >= 12 ? 2 : rounds >= 4 ? 1 : 0;
}

// The device's GELU table (gelu_table_kernel), built on first use and kept for the life of the process.  Returns nullptr - the kernel
// then evaluates the arithmetic form, bit-identical - in the float16 build, with ACTIONMESH_AMD_GELU_TABLE=0 (A/B), and when the first
// use happens inside a stream capture (no allocation / synchronisation may run there; the next eager call builds it).
static const uint16_t* gelu_table(hipStream_t st) {
#ifdef AM_F16
  return nullptr;
#else
  static uint16_t* tab[64];
  static int state[64];              // 0 unknown, 1 ready, 2 off
  static std::mutex mu;
  // the table belongs to the device that OWNS the stream, which need not be the calling thread's current device; the
  // null stream has no owner to ask: the current device then
  int dev = 0, cur = 0;
  if (hipGetDevice(&cur) != hipSuccess) return nullptr;
  dev = cur;
  if (st != nullptr && hipStreamGetDevice(st, &dev) != hipSuccess) { (void)hipGetLastError(); dev = cur; }
  if (dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (state[dev] == 0) {
    const char* e = getenv("ACTIONMESH_AMD_GELU_TABLE");
    if (e && e[0] == '0') { state[dev] = 2; return nullptr; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    // a failure here (allocation, launch) is NOT latched: this call runs the arithmetic epilogue - same bits - and the next one tries
    // again; only the environment switch turns the table off for the life of the process
    if (dev != cur && hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    uint16_t* t = nullptr;
    bool ok = hipMalloc(reinterpret_cast<void**>(&t), GT_BYTES) == hipSuccess;
    if (ok) {
      hipLaunchKernelGGL(gelu_table_kernel, dim3(ceil_div(GT_ENTRIES, 256)), dim3(256), 0, st, t);
      ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
      if (!ok) (void)hipFree(t);
    }
    if (dev != cur) (void)hipSetDevice(cur);
    if (!ok) { (void)hipGetLastError(); return nullptr; }
    tab[dev] = t;                    // synchronised: every stream of the device may read it from here on
    state[dev] = 1;
  }
  return state[dev] == 1 ? tab[dev] : nullptr;
#endif
}

// am_norm.hip: (mean, M2) of the 256-column slices of rows [row0, row0 + rows) of C, into part [row][ceil(N / 256)][2]
int am_row_part(const bf16_t* Cmat, int ldc, int64_t row0, int64_t rows, int N, float* part, void* stream);

// rows [m_main, M) of a big GEMM: the latency-built tail kernel when the row maps are the identity (every reference shape), the
// 128x128 kernel otherwise (or with ACTIONMESH_AMD_GEMM_TAIL=128, for A/B)
static void launch_tail(const am_gemm_args& args, int m_main, hipStream_t st) {
  static int use128 = -1;
  if (use128 < 0) { const char* e = getenv("ACTIONMESH_AMD_GEMM_TAIL"); use128 = (e && strcmp(e, "128") == 0) ? 1 : 0; }
  const int rows = args.M - m_main;
  if (!use128 && args.a_G == 0 && args.c_G == 0 && rows <= 128 && args.N % 4 == 0 && ((uintptr_t)args.bias % 16 == 0)) {
    hipLaunchKernelGGL(gemm_tail_kernel, dim3(ceil_div(args.N, 64), ceil_div(rows, 32)), dim3(512), 0, st, args, m_main);
  } else {
    const int tn = ceil_div(args.N, BN);
    hipLaunchKernelGGL(gemm_bf16_kernel, dim3(tn), dim3(256), SMEM_BYTES, st, args, 1, tn, m_main);
  }
}

extern "C" int am_gemm_bf16(const am_gemm_args* a, void* stream) {
  AM_CHECK(a != nullptr, "am_gemm_bf16: null args");
  AM_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "am_gemm_bf16: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
  AM_CHECK(a->K % BK == 0, "am_gemm_bf16: K=%d must be a multiple of %d", a->K, BK);
  AM_CHECK(a->N % 8 == 0, "am_gemm_bf16: N=%d must be a multiple of 8", a->N);
  AM_CHECK(a->A1 && a->W && a->C, "am_gemm_bf16: null operand");
  AM_CHECK(a->K1 > 0 && a->K1 <= a->K && a->K1 % BK == 0, "am_gemm_bf16: K1=%d invalid for K=%d", a->K1, a->K);
  AM_CHECK(a->K1 == a->K || a->A2 != nullptr, "am_gemm_bf16: K1 < K requires A2");
  AM_CHECK(a->lda1 % 8 == 0 && a->ldw % 8 == 0 && a->ldc % 8 == 0 && (a->A2 == nullptr || a->lda2 % 8 == 0),
           "am_gemm_bf16: leading dimensions must be multiples of 8 elements (16 B)");
  AM_CHECK(((uintptr_t)a->A1 | (uintptr_t)a->W | (uintptr_t)a->C | (uintptr_t)a->A2 | (uintptr_t)a->residual) % 16 == 0,
           "am_gemm_bf16: operands must be 16-byte aligned");
  if (a->ln_stats) {
    AM_CHECK(a->ln_colsum != nullptr, "am_gemm_bf16: ln_stats without ln_colsum");
    AM_CHECK(a->A2 == nullptr, "am_gemm_bf16: a folded LayerNorm needs one A operand");
    // a row map on A is honoured by the 128x128 kernel only (ln_stats is indexed by the mapped A row): narrow linears such as proj_out
    AM_CHECK(a->a_G == 0 || a->N < 256 || (a->act & 0x100), "am_gemm_bf16: a folded LayerNorm over a row-mapped A needs N < 256 (the 128x128 kernel)");
    AM_CHECK(!(a->act & 0x200), "am_gemm_bf16: the round-1 lockstep kernel has no folded-LayerNorm epilogue");
    AM_CHECK(((uintptr_t)a->ln_stats % 8 == 0) && ((uintptr_t)a->ln_colsum % 16 == 0) && (a->bias == nullptr || (uintptr_t)a->bias % 16 == 0) &&
             a->N % 4 == 0, "am_gemm_bf16: ln_stats / ln_colsum / bias misaligned");
  }
  if (a->ln_part) {
    AM_CHECK(a->c_G == 0, "am_gemm_bf16: ln_part needs the identity C row map");
    AM_CHECK((uintptr_t)a->ln_part % 8 == 0, "am_gemm_bf16: ln_part misaligned");
  }
  AM_ONCE_PER_DEVICE({
    AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
    AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_bf16_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES));
    AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256pp_bf16_kernel<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2PP_GELU_BYTES));
    AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256pp_bf16_kernel<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2PP_BYTES));
  });
  // act bit 8 (0x100) forces the 128x128 register-staged kernel (tests compare the two tilings); bit 9 (0x200) the round-1
  // lockstep main loop of the 256x256 tile (same-box A/B against the ping-pong loop)
  am_gemm_args args = *a;
  const bool force_small = (args.act & 0x100) != 0;
  const bool legacy = (args.act & 0x200) != 0;
  const bool force_big = (args.act & 0x400) != 0;       // tests: the 256x256 tile whatever the grid size
  const int abl = args.act & 0xF800;     // bits 11 / 12: timing ablations; bits 13-15: per-XCD start skew (experiment)
  const bool no_gelu_table = (args.act & 0x10000) != 0;   // tests / A/B: the arithmetic GELU epilogue in the 256x256 tile as well
  args.act &= 0xff;
  AM_CHECK(args.act == 0 || args.act == 1, "am_gemm_bf16: unknown activation %d", args.act);
  // the 256x256 tiles need a grid that fills the 256 CUs; mid-sized problems (the context encoder's 16 x 257 rows)
  // get four times as many 128x128 workgroups instead
  const bool big = !force_small && args.N >= 8 && args.M >= 1 &&
                   (force_big || (args.N >= 256 && args.M >= 1024 && (int64_t)ceil_div(args.M, B2) * ceil_div(args.N, B2) >= 192));
  args.act |= abl;          // kernels test `act & 0xff`; bits 11 / 12 are the store / residual timing ablations
  // ln_part: full 256-row tiles of the ping-pong kernel write their slices from the store loop (N % 256 == 0); every other row -
  // edge tiles, the 128x128 kernel's rows, the lockstep kernel - gets them from am_row_part below, reading C back
  float* const ln_part = args.ln_part;
  int64_t part_done = 0;                                       // rows [0, part_done) are covered by the fused epilogue
  args.ln_part = nullptr;
  if (big && (abl >> 13) == 0) {
    // start skew per XCD in units of 1.5 us: the more rounds of workgroups a GEMM has, the better the 7-unit tail amortises
    // (24-32 rounds: 6 us per XCD; 8 rounds: 1.5 us); act bits 13-15 = 7 turn it off (A/B runs)
    const int rounds = (int)(((int64_t)ceil_div(args.M, B2) * ceil_div(args.N, B2)) / 256);
    args.act |= gemm_skew_units(rounds) << 13;
  }
  if (big) {
    // M = B*T*(N+1) is 256*k + a small remainder for every reference shape (the +1 time token per
    // frame): a last 256-row tile holding a few rows would cost a whole extra round of workgroups.
    // The remainder rows go to the 128x128 kernel in a second, tiny launch instead.
    const int rem = args.M % B2;
    const int m_main = (rem != 0 && rem <= 128 && args.M > 8 * B2) ? args.M - rem : args.M;
    const int tiles_m = ceil_div(m_main, B2), tiles_n = ceil_div(args.N, B2);
    if (legacy) {
      hipLaunchKernelGGL(gemm256_bf16_kernel, dim3(tiles_m * tiles_n), dim3(512), SMEM2_BYTES,
                         (hipStream_t)stream, args, tiles_m, tiles_n, 0);
    } else {
      am_gemm_args main_args = args;
      if (ln_part && args.N % B2 == 0 && !(args.act & 0x1800)) {
        main_args.ln_part = ln_part;
        part_done = (int64_t)(args.M / B2) * B2;               // the kernel emits for tiles with m0 + 256 <= M only
        if (part_done > m_main) part_done = m_main;
      }
      const uint16_t* gt = ((main_args.act & 0xff) == 1 && !no_gelu_table) ? gelu_table((hipStream_t)stream) : nullptr;
      hipLaunchKernelGGL(gemm256pp_bf16_kernel<false>, dim3(tiles_m * tiles_n), dim3(512), gt ? SMEM2PP_GELU_BYTES : SMEM2PP_BYTES,
                         (hipStream_t)stream, main_args, tiles_m, tiles_n, 0, am_headpost_args{}, gt);
    }
    if (m_main < args.M) launch_tail(args, m_main, (hipStream_t)stream);
  } else {
    const int tiles_m = ceil_div(args.M, BM), tiles_n = ceil_div(args.N, BN);
    hipLaunchKernelGGL(gemm_bf16_kernel, dim3(tiles_m * tiles_n), dim3(256), SMEM_BYTES,
                       (hipStream_t)stream, args, tiles_m, tiles_n, 0);
  }
  AM_HIP(hipGetLastError());
  if (ln_part && part_done < args.M)
    AM_TRY(am_row_part(args.C, args.ldc, part_done, args.M - part_done, args.N, ln_part, stream));
  return AM_OK;
}

int am_head_post_partial(const am_headpost_args* a, int s_min_last, int s_min_other, void* stream);   // am_norm.hip

// nn.Linear (bias-free q | k | v projection, or the cross-attention to_q) + am_head_post in ONE launch: attention_processor.py:92-130.
// `g` describes the linear exactly as for am_gemm_bf16 with C = hp->X / ldc = hp->ldx (the buffer the un-fused pair would go through;
// only the rows of the 128x128 tail kernel are written to it), `hp` the head split as for am_head_post.  Falls back to the two calls
// - same results, bit for bit - when the shape does not qualify (small grids, row maps, an activation, key groups that would straddle).
extern "C" int am_gemm_headpost_bf16(const am_gemm_args* g, const am_headpost_args* hp, void* stream) {
  AM_CHECK(g && hp, "am_gemm_headpost_bf16: null argument");
  AM_CHECK(g->C == hp->X && g->ldc == hp->ldx && (int64_t)g->M == hp->rows && g->N == hp->heads * hp->nparts * 128,
           "am_gemm_headpost_bf16: the GEMM's output (C, ldc, M, N) must be the head split's input (X, ldx, rows, heads * nparts * 128)");
  bool has_v = false;
  for (int i = 0; i < hp->nparts; ++i) has_v |= hp->kinds[i] == 2;
  const int rem = g->M % B2;
  const bool tail_split = rem != 0 && rem <= 128 && g->M > 8 * B2;
  AM_CHECK(g->ln_part == nullptr, "am_gemm_headpost_bf16: no row statistics of a head-split output");
  const bool fuse = (g->act & 0xff) == 0 && !g->residual && (!g->bias || g->ln_stats) && !g->A2 && g->a_G == 0 && g->c_G == 0 && g->N % 256 == 0 && g->M % 16 == 0 &&
                    g->N >= 256 && g->M >= 1024 && (int64_t)ceil_div(g->M, B2) * ceil_div(g->N, B2) >= 192 &&
                    (!has_v || hp->seq_len % 16 == 0) && (!tail_split || rem <= hp->seq_len) && hp->rows % hp->seq_len == 0 &&
                    !(g->act & 0x700) && getenv("ACTIONMESH_AMD_NO_FUSED_QKV") == nullptr;
  if (!fuse) {
    AM_TRY(am_gemm_bf16(g, stream));
    return am_head_post(hp, stream);
  }
  AM_CHECK(g->K % BK == 0 && g->K1 == g->K && g->lda1 % 8 == 0 && g->ldw % 8 == 0, "am_gemm_headpost_bf16: bad GEMM operands");
  AM_CHECK(((uintptr_t)g->A1 | (uintptr_t)g->W | (uintptr_t)g->C) % 16 == 0, "am_gemm_headpost_bf16: operands must be 16-byte aligned");
  if (g->ln_stats)
    AM_CHECK(g->ln_colsum && (uintptr_t)g->ln_stats % 8 == 0 && (uintptr_t)g->ln_colsum % 16 == 0 && (g->bias == nullptr || (uintptr_t)g->bias % 16 == 0),
             "am_gemm_headpost_bf16: ln_stats / ln_colsum / bias missing or misaligned");
  AM_TRY(am_head_post_check(hp));
  AM_ONCE_PER_DEVICE({
    AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256pp_bf16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2PP_BYTES));
    AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
  });
  am_gemm_args args = *g;
  args.act = g->act & 0x1800;                 // the epilogue's timing ablations (tools/kernel_bench.py); 0 on the product path
  const int m_main = tail_split ? args.M - rem : args.M;
  const int tiles_m = ceil_div(m_main, B2), tiles_n = ceil_div(args.N, B2);
  {
    const int rounds = (int)(((int64_t)tiles_m * tiles_n) / 256);
    args.act |= gemm_skew_units(rounds) << 13;
  }
  am_gemm_args main_args = args;
  main_args.M = m_main;                      // the fused epilogue bounds its rows by M: the main grid owns [0, m_main)
  hipLaunchKernelGGL(gemm256pp_bf16_kernel<true>, dim3(tiles_m * tiles_n), dim3(512), SMEM2PP_BYTES, (hipStream_t)stream, main_args, tiles_m,
                     tiles_n, 0, *hp, (const uint16_t*)nullptr);
  if (m_main < args.M) launch_tail(args, m_main, (hipStream_t)stream);   // the remainder rows: plain linear into X, then the head split of exactly those rows
  AM_HIP(hipGetLastError());
  // tokens the fused epilogue did not produce: the tail rows (all in the last sequence) and the pad rows / columns of every sequence
  const int s_min_last = hp->seq_len - (args.M - m_main);
  return am_head_post_partial(hp, s_min_last, hp->seq_len, stream);
}
