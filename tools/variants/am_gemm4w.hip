// bf16 GEMM main-loop kernel, "4 x 128" structure: one workgroup = 4 waves = ONE wave per SIMD, 256 x 256 output
// tile, each wave a 128 x 128 sub-tile whose 256 accumulators live in AccVGPRs a[0:255], named literally in inline
// asm (am_gemm4w_asm.inc, tools/gen_gemm4w_asm.py).  Same maths, operand conventions and epilogue as
// gemm256_bf16_kernel in am_gemm.hip (read its header first); what changes is how the main loop is issued - by hand,
// like the 4x64 attention kernel (am_attention64.hip), whose P.V phase is this loop plus a softmax:
//   * every fragment read from LDS feeds FOUR MFMAs (the 8-wave kernel: two or four per read, 6 reads per 8 MFMAs;
//     here 8 reads per 16 MFMAs), AGPR-form MFMAs leave the arch file to fragments and addresses;
//   * 32-wide k-tiles in a 4-deep LDS ring (4 x 32 KiB): tile kt + 3 is fetched (LDS-DMA, global_load_lds_dwordx4) while
//     tile kt is multiplied; one barrier per k-tile with a counted s_waitcnt vmcnt(8) = "everything but the last tile's
//     pieces has landed"; the 8 pieces a wave issues per k-tile go out one per 4 MFMAs (back to back they queue on the
//     one TA and the waves sit in their issue);
//   * two fragment register sets: the MFMAs of k-step 0 read set 0 while set 1 is loaded (k-step 1 of the same tile),
//     the MFMAs of k-step 1 read set 1 while set 0 is loaded from the NEXT tile (landed: the counted wait is one tile
//     ahead of the multiply);
//   * LDS image of an operand tile: row r (64 B = 4 units of 16 B), unit c at position c ^ ((r >> 2) & 3): the 8 lanes
//     a ds_read_b128 serves per clock hit 8 distinct 16-byte slots of the 256-byte bank row; LDS-DMA writes are
//     lane-linear, so the swizzle is applied to the per-lane SOURCE address and mirrored on the read.
// hipcc must never touch an AccVGPR in this kernel (tests/test_host_cpu.py::test_gemm4w_register_audit).
#include "am_common.h"

namespace {

constexpr int G4_B = 256;                     // BM = BN
constexpr int G4_BK = 32;
constexpr int G4_TILE_B = G4_B * G4_BK * 2;   // one operand tile: 16 KiB
constexpr int G4_STAGE_B = 2 * G4_TILE_B;     // [A tile][W tile]
constexpr int G4_NST = 4;                     // ring depth: 128 KiB (the epilogue's staging area re-uses it)
constexpr int G4_GROUP_M = 8;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

#include "am_gemm4w_asm.inc"

#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define HOLD4(a, b, c, d) asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d))

__device__ inline int64_t g4_map_row(int r, int G, int gs, int off) {
  if (G <= 0) return r;
  const int g = r / G;
  return (int64_t)g * gs + off + (r - g * G);
}

__global__ __launch_bounds__(256, 1) void gemm4w_bf16_kernel(am_gemm_args p, int tiles_m, int tiles_n, int m_base) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // ---- tile of this workgroup: bijective XCD remap + grouped order (as gemm256_bf16_kernel) ----
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
    const int group_sz = G4_GROUP_M * tiles_n;
    const int g = lid / group_sz;
    const int first_m = g * G4_GROUP_M;
    const int gm = min(G4_GROUP_M, tiles_m - first_m);
    const int in_g = lid - g * group_sz;
    tm = first_m + in_g % gm;
    tn = in_g / gm;
  }
  const int m0 = m_base + tm * G4_B, n0 = tn * G4_B;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- LDS-DMA source pointers: 16-byte unit U = i*256 + wave*64 + lane of a 1024-unit operand tile, i = 0..3 ----
  const bf16_t* a1p[4];
  const bf16_t* a2p[4];
  const bf16_t* wp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int U = i * 256 + wave * 64 + lane;
    const int r = U >> 2, c = (U & 3) ^ ((r >> 2) & 3);
    const int ar = min(m0 + r, p.M - 1);
    const int64_t pr = g4_map_row(ar, p.a_G, p.a_gs, p.a_off);
    a1p[i] = p.A1 + pr * p.lda1 + c * 8;
    a2p[i] = p.A2 ? p.A2 + pr * p.lda2 + c * 8 - p.K1 : nullptr;
    const int wr = min(n0 + r, p.N - 1);
    wp[i] = p.W + (int64_t)wr * p.ldw + c * 8;
  }
  const int nk = p.K / G4_BK;
  // piece `pc` (0..7: A pieces 0..3, W pieces 0..3) of k-tile kt (clamped: past the end the last tile is re-fetched into
  // a stage nobody reads, so that the counted wait sees a constant number of pieces per trip)
  auto dma_piece = [&](int kt, int pc) __attribute__((always_inline)) {
    const int kc = min(kt, nk - 1);
    const int k0 = kc * G4_BK;
    unsigned char* dst = smem + (kt & (G4_NST - 1)) * G4_STAGE_B + (pc >> 2) * G4_TILE_B + ((pc & 3) * 256 + wave * 64) * 16;
    const bf16_t* src = pc < 4 ? (k0 < p.K1 ? a1p[pc & 3] + k0 : a2p[pc & 3] + k0) : wp[pc & 3] + k0;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
  };

  // ---- accumulators start from the bias (fp32; lane (l31, hi) of block (i, j) holds W rows 8g + 4hi .. +3 of block i) ----
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4_t bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bv = *reinterpret_cast<const f32x4_t*>(p.bias + min(n0 + wn * 128 + i * 32 + 8 * g + 4 * hi, p.N - 4));
#pragma unroll
      for (int j = 0; j < 4; ++j) g4_acc_write4(i * 4 + j, g, bv);
    }

  // ---- fragment read offsets inside a stage: row R = block * 32 + l31, unit (ks * 2 + hi) ^ ((R >> 2) & 3) ----
  const int sw = (l31 >> 2) & 3;          // block bases are multiples of 32 rows: the swizzle depends on l31 only
  int a_off[2], w_off[2];                 // per k-step; + block * 2048 (immediate), + stage base
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int u = ((ks * 2 + hi) ^ sw) << 4;
    a_off[ks] = (wm * 128 + l31) * 64 + u;
    w_off[ks] = G4_TILE_B + (wn * 128 + l31) * 64 + u;
  }
  bf16x8_t af[2][4], wf[2][4];            // [set = k-step][block]
  auto a_frag = [&](const unsigned char* st, int ks, int j) __attribute__((always_inline)) {
    return *reinterpret_cast<const bf16x8_t*>(st + a_off[ks] + j * 2048);
  };
  auto w_frag = [&](const unsigned char* st, int ks, int i) __attribute__((always_inline)) {
    return *reinterpret_cast<const bf16x8_t*>(st + w_off[ks] + i * 2048);
  };

  // ---- prologue: tiles 0..2 in flight; set 0 of tile 0 ----
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) dma_piece(t, pc);
  asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");        // tile 0 landed and is visible
#pragma unroll
  for (int b = 0; b < 4; ++b) { af[0][b] = a_frag(smem, 0, b); wf[0][b] = w_frag(smem, 0, b); }

  // ---- main loop: one 32-wide k-tile per trip ----
  // k-step s of a trip: 16 MFMAs on set s (block order i-major: W block i against the 4 A blocks); behind the first 8
  // the 8 reads that fill the other set (in the order the next k-step uses them), behind the last 8 four DMA pieces.
  for (int kt = 0; kt < nk; ++kt) {
    // all but the previous trip's 8 pieces have landed (tiles <= kt + 1); every wave is done with the stage this trip refills
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned char* st = smem + (kt & (G4_NST - 1)) * G4_STAGE_B;
    const unsigned char* stn = smem + ((kt + 1) & (G4_NST - 1)) * G4_STAGE_B;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const unsigned char* src = s == 0 ? st : stn;       // set 1 <- this tile's k-step 1, set 0 <- next tile's k-step 0
      const int o = s ^ 1;                                 // the set being filled; it holds k-step (s ^ 1) of `src`
      FENCE();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          g4_mfma(i * 4 + j, wf[s][i], af[s][j]);
          const int n = i * 4 + j;                         // MFMA index in the k-step
          if (n < 8) {                                     // 8 reads behind the first 8 MFMAs: a0 w0 a1 a2 a3 w1 w2 w3; the
            const int rd = n;                              // compiler waits lgkmcnt(0) in front of the next k-step, by
            if (rd == 0) af[o][0] = a_frag(src, o, 0);     // which time the last of them is 8 MFMAs old
            else if (rd == 1) wf[o][0] = w_frag(src, o, 0);
            else if (rd < 5) af[o][rd - 1] = a_frag(src, o, rd - 1);
            else wf[o][rd - 4] = w_frag(src, o, rd - 4);
          } else if ((n & 1) == 0) {
            dma_piece(kt + 3, s * 4 + ((n - 8) >> 1));     // 4 pieces behind MFMAs 8, 10, 12, 14
          }
          FENCE();
        }
      }
      HOLD4(af[s][0], af[s][1], af[s][2], af[s][3]);
      HOLD4(wf[s][0], wf[s][1], wf[s][2], wf[s][3]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // no piece may land in the staging area below
  g4_read_fence();

  // ---- epilogue: activation in registers, bf16 tile staged through LDS, row-contiguous stores + residual ----
  // stage[m][n] bf16, row = 512 B = 64 units of 8 B; unit u of row m sits at u ^ (m & 15)
  unsigned char* stage = smem;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x16_t acc = g4_acc_read16(i * 4 + j);
      const int ml = wm * 128 + j * 32 + l31;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = wn * 128 + i * 32 + 8 * g + 4 * hi;       // 4 consecutive columns
        u32x2_t w;
        if (p.act == 1) {                                        // F.gelu on the bf16 linear output -> bf16
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(rbf(acc[4 * g + e]));
          w = u32x2_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        } else {
          w = u32x2_t{pack_bf2(acc[4 * g], acc[4 * g + 1]), pack_bf2(acc[4 * g + 2], acc[4 * g + 3])};
        }
        const int u = (nl >> 2) ^ (ml & 15);
        *reinterpret_cast<u32x2_t*>(stage + ml * 512 + u * 8) = w;
      }
    }
  __syncthreads();
  {
    // thread -> 16-byte chunk k16 (8 columns) of rows ml = pass * 8 + (tid >> 5)
    const int k16 = tid & 31, r8 = tid >> 5;
    const int gn = n0 + k16 * 8;
    if (gn < p.N) {
      auto fetch = [&](int ml) __attribute__((always_inline)) {
        u32x4_t sv = *reinterpret_cast<const u32x4_t*>(stage + ml * 512 + ((k16 ^ ((ml & 15) >> 1)) << 4));
        if (ml & 1) sv = u32x4_t{sv[2], sv[3], sv[0], sv[1]};
        return sv;
      };
      auto add_res = [&](u32x4_t sv, const bf16_t* rp) __attribute__((always_inline)) {
        const u32x4_t rv = *reinterpret_cast<const u32x4_t*>(rp);
#pragma unroll
        for (int e = 0; e < 4; ++e) sv[e] = pack_bf2(bflo(sv[e]) + bflo(rv[e]), bfhi(sv[e]) + bfhi(rv[e]));
        return sv;
      };
      if (p.c_G <= 0 && m0 + G4_B <= p.M) {
        const uint32_t lane_off = ((uint32_t)r8 * (uint32_t)p.ldc + (uint32_t)gn) * 2u;
        const int64_t step = (int64_t)8 * p.ldc;
        bf16_t* crow = p.C + (int64_t)m0 * p.ldc;
        if (p.residual) {
          const bf16_t* rrow = p.residual + (int64_t)m0 * p.ldc;
#pragma unroll 4
          for (int pass = 0; pass < 32; ++pass) {
            const u32x4_t sv = add_res(fetch(pass * 8 + r8), reinterpret_cast<const bf16_t*>(reinterpret_cast<const unsigned char*>(rrow + pass * step) + lane_off));
            *reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned char*>(crow + pass * step) + lane_off) = sv;
          }
        } else {
#pragma unroll 4
          for (int pass = 0; pass < 32; ++pass)
            *reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned char*>(crow + pass * step) + lane_off) = fetch(pass * 8 + r8);
        }
      } else {
#pragma unroll 2
        for (int pass = 0; pass < 32; ++pass) {
          const int ml = pass * 8 + r8;
          const int gmr = m0 + ml;
          if (gmr < p.M) {
            u32x4_t sv = fetch(ml);
            const int64_t pr = g4_map_row(gmr, p.c_G, p.c_gs, p.c_off);
            if (p.residual) sv = add_res(sv, p.residual + pr * p.ldc + gn);
            *reinterpret_cast<u32x4_t*>(p.C + pr * p.ldc + gn) = sv;
          }
        }
      }
    }
  }
}

}  // namespace

// Main grid of the 4-wave kernel (called by am_gemm_bf16 in am_gemm.hip for K % 32 == 0 problems that fill the chip).
int am_gemm4w_launch(const am_gemm_args* a, int tiles_m, int tiles_n, void* stream) {
  static bool attr_set = false;
  if (!attr_set) {
    AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm4w_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               G4_NST * G4_STAGE_B));
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm4w_bf16_kernel, dim3(tiles_m * tiles_n), dim3(256), G4_NST * G4_STAGE_B, (hipStream_t)stream, *a, tiles_m,
                     tiles_n, 0);
  return AM_OK;
}
