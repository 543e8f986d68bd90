// fp8 (OCP e4m3) flash attention for gfx950: QK^T and P.V on v_mfma_scale_f32_32x32x64_f8f6f4 (the MX-scaled K = 64
// form, the only fp8 MFMA that runs at twice the bf16 rate; block scales fixed at 2^0), fp32 online softmax.
//
// The variant BASELINE.json configs[4] / north_star ask for ("64f x 8192tok, fp8 MFMA") of
// F.scaled_dot_product_attention (attention_processor.py:133-139).  The bf16 kernels (am_attention64.hip,
// am_attention.hip) stay the default dtype; this one is selected per handle (am_config.attn_fp8) or called directly.
//
// Operands (am_attention_quantize_fp8 writes them from the bf16 operand layouts of am_head_post):
//   Q8   [nseq][H][sq_pad][128]            q * scale * log2(e), so scores are born in log2 units
//   K8   [chunks][nseq][H][sk_pad][128]
//   V8T  [chunks][nseq][H][128][sk_pad]    inside every 64-key tile, position pos = 32 h + j holds key
//                                          kperm(pos) = 32 (j >> 4) + (j & 3) + 8 ((j & 15) >> 2) + 4 h
// The scaled MFMA contracts over 64 k-slots; lane (row, h = lane >> 5) carries 32 consecutive bytes of its row.  Which
// logical k a (h, byte) slot is does not matter as long as both operands agree (the A and B register layouts are
// symmetric), so: QK^T contracts channels with slot (h, j) = channel 64 s + 32 h + j, and P.V contracts the keys of a tile
// with slot (h, j) = the key the S^T = K Q^T accumulator register j of lane-half h holds (32x32 C/D layout: register r
// of block kb is key 32 kb + (r & 3) + 8 (r >> 2) + 4 h): the probabilities go from the score registers straight into the
// P.V B operand (v_cvt_pk_fp8_f32, four per VGPR), and V^T is stored in that key order.
//
// Structure: 8 waves x 32 query rows, two groups of four waves HALF A TILE APART (the ping-pong of the round-2 GEMM):
// between two barriers one group runs its matrix interval - O += V^T(t) P^T(t) and S(t+1) = K(t+1) Q^T, 8 MFMAs of 64
// cycles - at s_setprio 1 while its partner on every SIMD runs its softmax interval (32 exp2 / row max / row sum / fp8
// packing per lane, plus the fragment ds_reads and LDS-DMA issue for its next matrix interval).  K8 / V8T tiles (8 + 8 KiB
// per 64 keys) arrive by LDS-DMA into a 4-deep ring, three tiles ahead, retired by one counted vmcnt(2) per tile.
// Online softmax with a deferred re-base (threshold 2^3): p = 2^(s - m_run + 5), so p <= 2^8 fits e4m3 (max 448) and
// probabilities down to 2^-14 of the row maximum survive the 2^-9 flush.  The VALU interval is what bounds this kernel, so
// it carries only what has to be there - 32 exp2, 16 packs, 11 max3 per lane and tile: the scores are born relative to the
// running max (the QK^T accumulators start from a splat of 5 - m_run: no subtraction in front of the exponentials), and keys
// past a chunk's end are a wave-uniform branch taken once per chunk; fragment reads and LDS-DMA issue sit in the matrix
// interval, between MFMAs, where the wave only waits for the pipe.  (Row sums on the matrix pipe - one more MFMA against a
// block of ones - were written and dropped: 24 more registers, and at 256 the kernel spills.)
#include "am_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

#ifndef AM_F8_PKSUM
#define AM_F8_PKSUM 1      // row sums as packed adds: 17.55 -> 17.15 ms at the headline launch (profiles/r03k_ab.txt)
#endif
constexpr int HD8 = 128;
constexpr int KT = 64;                 // keys per tile
constexpr int STAGE_BYTES = 16384;     // K8 tile [64][128] + V8T tile [128][64]
constexpr int NSTAGE = 8;             // ring slots (tiles are staged five ahead)
constexpr float P_SHIFT = 5.f;         // probabilities are carried as 2^5 p (row sums too: the factor cancels in O / l)
constexpr float DEFER_T = 3.f;         // deferred re-base threshold (log2 units)
constexpr int SCALE_ONE = 0x7f7f7f7f;  // E8M0 block scale 2^0

__device__ inline float clamp_e4m3(float x) { return fminf(fmaxf(x, -448.f), 448.f); }

// key held by k-slot `pos` of a 64-key tile (see header)
__host__ __device__ inline int kperm(int pos) {
  const int h = pos >> 5, j = pos & 31;
  return 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * h;
}

// Value of the same register in lane ^ 32 (v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second).
// hipcc (ROCm 7.2) FOLDS arithmetic on the two results when both operands are one SSA value - fmaxf(sw[0], sw[1]) becomes
// sw[0] and sw[0] + sw[1] becomes 2 sw[0], although the results differ in every lane (measured: row sums of the softmax off
// by the ratio of the two half-row sums).  Making one operand opaque stops the fold; the partner's value is then selected
// by lane half and combined with the lane's own value explicitly.
__device__ inline float other_half(float x, int hi) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  asm volatile("" : "+v"(b));
  const auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  return __builtin_bit_cast(float, hi ? sw[0] : sw[1]);
}

// The MFMAs of the main loop are inline asm: a builtin MFMA is a pure value to the IR passes, which sink it below the
// fragment reads that are meant to REUSE its operand registers (both fragment sets live at once: spills at 256 VGPRs and a
// vmcnt(0) in front of every reload).  asm volatile statements keep their order among themselves and, with the "memory"
// clobber, against the LDS reads / LDS-DMA around them.  hipcc pads nothing for an asm statement: every consumer of these
// results sits behind at least four further MFMAs or a barrier plus tens of instructions (16-pass MFMA: 19 wait states).
__device__ inline void mfma_f8_acc(f32x16_t& d, const i32x8_t& a, const i32x8_t& b, int one) {          // d += a x b
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(d) : "v"(a), "v"(b), "v"(one) : "memory");
}
__device__ inline void mfma_f8_init(f32x16_t& d, const i32x8_t& a, const i32x8_t& b, const f32x16_t& c, int one) {   // d = a x b + c
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %4 op_sel_hi:[0,0,0]" : "=&v"(d) : "v"(a), "v"(b), "v"(c), "v"(one) : "memory");
}

#define F8_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// ------------------------------------------------------------------------------------------------------------------
// quantisation of the bf16 operand layouts
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quant_rows_kernel(const bf16_t* __restrict__ src, uint8_t* __restrict__ dst, int64_t n16,
                                                         float mul) {
  // 16 elements per thread: 32 B in, 16 B out
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
    const u32x4_t a = *reinterpret_cast<const u32x4_t*>(src + i * 16);
    const u32x4_t b = *reinterpret_cast<const u32x4_t*>(src + i * 16 + 8);
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      int w0 = 0, w1 = 0;
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(bflo(a[2 * e]) * mul), clamp_e4m3(bfhi(a[2 * e]) * mul), w0, false);
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(bflo(a[2 * e + 1]) * mul), clamp_e4m3(bfhi(a[2 * e + 1]) * mul), w0, true);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(bflo(b[2 * e]) * mul), clamp_e4m3(bfhi(b[2 * e]) * mul), w1, false);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(bflo(b[2 * e + 1]) * mul), clamp_e4m3(bfhi(b[2 * e + 1]) * mul), w1, true);
      o[e] = (uint32_t)w0;
      o[2 + e] = (uint32_t)w1;
    }
    *reinterpret_cast<u32x4_t*>(dst + i * 16) = o;
  }
}

// V^T: one thread per (channel row, 64-key tile): 128 B of bf16 in (perm16 key order), 64 B of fp8 out (kperm order)
__global__ __launch_bounds__(256) void quant_vt_kernel(const bf16_t* __restrict__ src, uint8_t* __restrict__ dst, int64_t rows,
                                                       int tiles, int sk_pad) {
  const int64_t total = rows * tiles;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / tiles;
    const int tile = (int)(i - row * tiles);
    const bf16_t* s = src + row * sk_pad + tile * KT;
    uint32_t w[32];       // 64 bf16
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const u32x4_t v = *reinterpret_cast<const u32x4_t*>(s + u * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) w[u * 4 + e] = v[e];
    }
    auto at = [&](int key) {   // V^T position of `key` inside the tile: perm16 within its group of 16
      const int p = (key & ~15) | perm16(key & 15);
      const uint32_t x = w[p >> 1];
      return clamp_e4m3((p & 1) ? bfhi(x) : bflo(x));
    };
    uint32_t o[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      int x = 0;
      x = __builtin_amdgcn_cvt_pk_fp8_f32(at(kperm(4 * q)), at(kperm(4 * q + 1)), x, false);
      x = __builtin_amdgcn_cvt_pk_fp8_f32(at(kperm(4 * q + 2)), at(kperm(4 * q + 3)), x, true);
      o[q] = (uint32_t)x;
    }
    uint8_t* d = dst + row * sk_pad + tile * KT;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      *reinterpret_cast<u32x4_t*>(d + u * 16) = u32x4_t{o[4 * u], o[4 * u + 1], o[4 * u + 2], o[4 * u + 3]};
  }
}

// ------------------------------------------------------------------------------------------------------------------
// the attention kernel
// ------------------------------------------------------------------------------------------------------------------
struct f8_args {
  const uint8_t* Q; const uint8_t* K; const uint8_t* Vt; bf16_t* O;
  int heads, sq, sq_pad, sk, sk_pad, nchunks, tiles_per_chunk, ldo;
  int64_t chunk_stride;
  int chunk_first, chunk_total;    // the chunks walked are (chunk_first + i) % chunk_total (chunk_total = 0: 0 .. nchunks-1)
  int qblk_base;                   // query block of blockIdx.x = 0
  float* state;                    // MODE 1 / 2: un-normalised (O, m, l) per row, [seq*heads][sq_pad][F8_STATE_LD]
  float* part;                     // MODE 3: partials of the split last block, [seq*heads][Z][256][F8_STATE_LD]
};
constexpr int F8_STATE_LD = 132;   // floats per saved row: O[128], m, l, pad - the layout of the bf16 kernels (am_attention64.hip)
constexpr int F8_SPLIT_Z = 16;

// ABL: timing ablations (numerically meaningless; tools/kernel_bench.py --ablate-fp8): 1 no exponentials, 2 no LDS-DMA in the
// loop, 4 no fragment reads in the loop, 8 no MFMAs in the loop, 16 no row max.
// MODE (round 3, the forms the bf16 kernels already had): 0 = one pass; 1 = stop after the chunks walked and save the un-normalised
// (O, m, l) of every row to p.state (multi-GPU: the local K/V shard while the others are in flight); 2 = resume from p.state, finish,
// write O; 3 = the short last query block split over the key range: workgroup z = blockIdx.z takes tiles [z n / Z, (z+1) n / Z) and
// writes a partial for attn_combine (am_attention.hip) instead of costing a 17th round of workgroups.
// O and l carry the factor 2^P_SHIFT in every mode; it cancels in O / l, in the resume and in the merge of the partials.
template <int ABL, int MODE = 0>
__global__ __launch_bounds__(512, 2) void attn_fp8_kernel(f8_args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int l31 = lane & 31, hi = lane >> 5;
  const int qblk = p.qblk_base + blockIdx.x, sh = blockIdx.y;          // sh = seq * heads + head
  const int seq = sh / p.heads, head = sh - seq * p.heads;

  // ---- Q fragments: lane (row l31, half hi), k-step s: channels 64 s + 32 hi .. + 31
  const int qrow = qblk * 256 + wave * 32 + l31;
  i32x8_t qf[2];
  {
    const uint8_t* qp = p.Q + ((int64_t)sh * p.sq_pad + qrow) * HD8 + hi * 32;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const u32x4_t a = *reinterpret_cast<const u32x4_t*>(qp + s * 64);
      const u32x4_t b = *reinterpret_cast<const u32x4_t*>(qp + s * 64 + 16);
      qf[s] = i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) asm volatile("" : "+v"(qf[s]));     // waited for here, before any LDS-DMA is in flight
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- LDS-DMA: per tile every wave moves one 1 KiB piece of K8 (8 key rows) and one of V8T (16 channel rows);
  // lane-linear destination, swizzle on the source unit (K: unit ^ ((row >> 1) & 7); V^T: unit ^ ((row >> 2) & 3))
  const int all_tiles = p.nchunks * p.tiles_per_chunk;
  const int t_begin = MODE == 3 ? (int)((int64_t)blockIdx.z * all_tiles / gridDim.z) : 0;
  const int t_end = MODE == 3 ? (int)((int64_t)(blockIdx.z + 1) * all_tiles / gridDim.z) : all_tiles;
  const int total_tiles = t_end - t_begin;                 // tiles this workgroup walks (local numbering 0 .. total_tiles-1)
  const int kr = wave * 8 + (lane >> 3);
  const uint32_t k_lane_off = (uint32_t)kr * HD8 + (uint32_t)(((lane & 7) ^ ((kr >> 1) & 7)) << 4);
  const int vr = wave * 16 + (lane >> 2);
  const uint32_t v_lane_off = (uint32_t)vr * (uint32_t)p.sk_pad + (uint32_t)(((lane & 3) ^ ((vr >> 2) & 3)) << 4);
  // running source of the next tile to stage (wave-uniform): advances one tile per call, jumps at chunk ends, and stays
  // on the last tile past the end of the stream (the re-fetch lands in a ring slot nobody reads)
  const uint8_t* base_k = p.K + (int64_t)sh * p.sk_pad * HD8;
  const uint8_t* base_v = p.Vt + (int64_t)sh * p.sk_pad * HD8;
  int st_n = 0, st_ti = t_begin % p.tiles_per_chunk, st_pos = t_begin / p.tiles_per_chunk;     // st_pos: chunk position in the walk
  auto chunk_off = [&](int pos) __attribute__((always_inline)) {          // physical chunk of walk position `pos`
    int c = p.chunk_first + pos;
    if (p.chunk_total > 0 && c >= p.chunk_total) c -= p.chunk_total;
    return (int64_t)c * p.chunk_stride;
  };
  const uint8_t* st_k = base_k + chunk_off(st_pos) + (int64_t)st_ti * KT * HD8;
  const uint8_t* st_v = base_v + chunk_off(st_pos) + (int64_t)st_ti * KT;
  auto stage = [&]() __attribute__((always_inline)) {
    unsigned char* slot = smem + (st_n & (NSTAGE - 1)) * STAGE_BYTES;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(st_k + k_lane_off), (lds_ptr_t)(slot + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(st_v + v_lane_off), (lds_ptr_t)(slot + 8192 + wave * 1024), 16, 0, 0);
    ++st_n;
    if (st_n < total_tiles) {
      if (st_ti == p.tiles_per_chunk - 1) {
        st_ti = 0;
        ++st_pos;
        st_k = base_k + chunk_off(st_pos);
        st_v = base_v + chunk_off(st_pos);
      } else {
        ++st_ti;
        st_k += KT * HD8;
        st_v += KT;
      }
    }
  };

  // ---- fragment addresses
  // K8 tile: row r (key), 8 units of 16 B; lane reads units 4 s + 2 hi + {0, 1} of key 32 kb + l31
  const int ksw = (l31 >> 1) & 7;                          // (row >> 1) & 7 with row = 32 kb + l31
  int k_off[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 2; ++e) k_off[s][e] = l31 * HD8 + (((4 * s + 2 * hi + e) ^ ksw) << 4);
  // V8T tile: row c (channel), 4 units; lane reads units 2 hi + {0, 1} of channel 32 cb + l31
  const int vsw = (l31 >> 2) & 3;
  int v_off[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) v_off[e] = 8192 + l31 * 64 + (((2 * hi + e) ^ vsw) << 4);

  i32x8_t kf[2][2], vf[4];
  auto read_k = [&](int tt) __attribute__((always_inline)) {
    const unsigned char* slot = smem + (tt & (NSTAGE - 1)) * STAGE_BYTES;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const u32x4_t a = *reinterpret_cast<const u32x4_t*>(slot + kb * 32 * HD8 + k_off[s][0]);
        const u32x4_t b = *reinterpret_cast<const u32x4_t*>(slot + kb * 32 * HD8 + k_off[s][1]);
        kf[kb][s] = i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
      }
  };
  auto read_v1 = [&](int tt, int cb) __attribute__((always_inline)) {          // one V^T fragment (2 ds_read_b128)
    const unsigned char* slot = smem + (tt & (NSTAGE - 1)) * STAGE_BYTES;
    const u32x4_t a = *reinterpret_cast<const u32x4_t*>(slot + cb * 32 * 64 + v_off[0]);
    const u32x4_t b = *reinterpret_cast<const u32x4_t*>(slot + cb * 32 * 64 + v_off[1]);
    vf[cb] = i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
  };
  auto read_k1 = [&](int tt, int kb, int s2) __attribute__((always_inline)) {   // one K fragment (2 ds_read_b128)
    const unsigned char* slot = smem + (tt & (NSTAGE - 1)) * STAGE_BYTES;
    const u32x4_t a = *reinterpret_cast<const u32x4_t*>(slot + kb * 32 * HD8 + k_off[s2][0]);
    const u32x4_t b = *reinterpret_cast<const u32x4_t*>(slot + kb * 32 * HD8 + k_off[s2][1]);
    kf[kb][s2] = i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
  };
  auto read_v = [&](int tt) __attribute__((always_inline)) {
    const unsigned char* slot = smem + (tt & (NSTAGE - 1)) * STAGE_BYTES;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const u32x4_t a = *reinterpret_cast<const u32x4_t*>(slot + cb * 32 * 64 + v_off[0]);
      const u32x4_t b = *reinterpret_cast<const u32x4_t*>(slot + cb * 32 * 64 + v_off[1]);
      vf[cb] = i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
    }
  };

  f32x16_t o[4], sc[2];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[cb][r] = 0.f;
  // Scores are born RELATIVE: the QK^T accumulators start from a splat of binit = P_SHIFT - m_run (column q = this lane's
  // row), so  sc = s - m_run + P_SHIFT  needs no per-element subtraction before the exponential; a re-base (rare) shifts
  // the scores already in registers.  m_run starts at P_SHIFT (binit = 0) and the first tile always re-bases.
  float m_run = P_SHIFT, l_run = 0.f;
  if (MODE == 2) {                                           // resume: (O, m, l) of this lane's row as the first pass left them
    const float* sp = p.state + ((int64_t)sh * p.sq_pad + qrow) * F8_STATE_LD;
    m_run = sp[HD8];
    l_run = hi == 0 ? sp[HD8 + 1] : 0.f;                     // the two half-lanes' sums are joined at the end
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t t4 = *reinterpret_cast<const f32x4_t*>(sp + cb * 32 + 8 * g + 4 * hi);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[cb][4 * g + i] = t4[i];
      }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) asm volatile("" : "+v"(o[cb]));      // landed before the first LDS-DMA piece is counted
    asm volatile("" : "+v"(m_run), "+v"(l_run));
  }
  int one = SCALE_ONE;                                       // E8M0 block scales 2^0 (a VGPR operand of the scaled MFMA)
  asm volatile("" : "+v"(one));
  f32x16_t bsplat;                                           // P_SHIFT - m_run in 16 registers: SrcC of the first QK^T MFMAs,
#pragma unroll                                               // rewritten on a re-base only
  for (int r = 0; r < 16; ++r) bsplat[r] = P_SHIFT - m_run;  // 0 unless resuming
  asm volatile("" : "+v"(bsplat));
  auto qk = [&]() __attribute__((always_inline)) {          // S^T = K Q^T + splat for the tile whose fragments are in kf
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) mfma_f8_init(sc[kb], kf[kb][0], qf[0], bsplat, one);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) mfma_f8_acc(sc[kb], kf[kb][1], qf[1], one);
  };

  i32x8_t pf = {0, 0, 0, 0, 0, 0, 0, 0};                     // P^T B operand: byte j = 16 kb + r
  const int tail_valid = p.sk - (p.tiles_per_chunk - 1) * KT;   // valid keys in a chunk's last tile (1 .. 64)
  int tic = t_begin % p.tiles_per_chunk;                     // tile-in-chunk counter of the softmax tile

  auto softmax = [&](bool first) __attribute__((always_inline)) {
    const bool last_of_chunk = tic == p.tiles_per_chunk - 1;
    tic = last_of_chunk ? 0 : tic + 1;
    if (last_of_chunk && tail_valid < KT) {                  // wave-uniform, once per chunk: keys past the chunk's end
      asm volatile("" ::: "memory");                         // (keeps this a branch: if-converted it costs 32 selects per tile)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hi >= tail_valid) sc[kb][r] = -INFINITY;
    }
    // row max: four independent chains (v_max3 under -fno-honor-nans), then the other half of the row
    float mxa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mxa[i] = fmaxf(fmaxf(sc[0][i], sc[1][i]), sc[0][i + 4]);
#pragma unroll
    for (int i = 0; i < 4; ++i) mxa[i] = fmaxf(fmaxf(mxa[i], sc[1][i + 4]), sc[0][i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) mxa[i] = fmaxf(fmaxf(mxa[i], sc[1][i + 8]), sc[0][i + 12]);
#pragma unroll
    for (int i = 0; i < 4; ++i) mxa[i] = fmaxf(mxa[i], sc[1][i + 12]);
    float mx = fmaxf(fmaxf(mxa[0], mxa[1]), fmaxf(mxa[2], mxa[3]));
    if (ABL & 16) mx = sc[0][0];
    mx = fmaxf(mx, other_half(mx, hi));
    // relative to the running max the row max is mx - P_SHIFT; re-base when it is above 2^DEFER_T (or on the first tile)
    if (first || __builtin_amdgcn_ballot_w64(mx > P_SHIFT + DEFER_T) != 0) {
      const float delta = first ? mx - P_SHIFT : fmaxf(mx - P_SHIFT, 0.f);
      const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[cb][r] *= alpha;
      l_run *= alpha;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[kb][r] -= delta;
      m_run += delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) bsplat[r] = P_SHIFT - m_run;
    }
#if AM_F8_PKSUM       // the 32 row-sum adds as 16 packed adds (v_pk_add_f32, default operand selection - not the form of DESIGN.md section 9)
    f32x2_t ps2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#else
    float ps[4] = {0.f, 0.f, 0.f, 0.f};                      // four independent row-sum chains (fp32, before the rounding)
#endif
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        float e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          e[i] = (ABL & 1) ? sc[kb][4 * q4 + i] : __builtin_amdgcn_exp2f(sc[kb][4 * q4 + i]);
#if !AM_F8_PKSUM
          ps[i] += e[i];
#endif
        }
#if AM_F8_PKSUM
        ps2[0] += f32x2_t{e[0], e[1]};
        ps2[1] += f32x2_t{e[2], e[3]};
#endif
        int w = pf[4 * kb + q4];                               // both halves are overwritten: no zero-initialising move
        w = __builtin_amdgcn_cvt_pk_fp8_f32(e[0], e[1], w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(e[2], e[3], w, true);
        pf[4 * kb + q4] = w;
      }
#if AM_F8_PKSUM
    { const f32x2_t t = ps2[0] + ps2[1]; l_run += t[0] + t[1]; }
#else
    l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
#endif
  };

  // ---- schedule.  Per tile t a wave runs a softmax interval V(t) - VALU only - and a matrix interval M(t):
  //   V(t):  LDS-DMA of tile t+5; softmax of S(t) -> P(t)              | s_waitcnt vmcnt(4) (tile t+3 landed), lgkmcnt(0), barrier
  //   M(t):  O += V^T(t) P(t), S(t+1) = K(t+1) Q^T; in the gaps between the MFMAs (one MFMA time of free issue each): the
  //          fragment reads of V(t+1) and K(t+2) into the registers the MFMAs have just consumed               | barrier
  // Group 1 runs one interval behind group 0, so on every SIMD one wave multiplies while the other exponentiates.
  // A tile is read by the other group up to one interval after this wave's wait for it: tile t+3 is retired at the end of
  // V(t) and first read in M(t+1).
  stage(); stage(); stage(); stage(); stage();            // tiles 0 .. 4
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  F8_BARRIER();
  read_k(0);
  qk();                                                   // S(0) (every lane's m_run is still P_SHIFT: the accumulators start from 0)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // asm MFMA results: nothing is padded for the first softmax
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // tiles 1, 2
  F8_BARRIER();
  read_k(1);
  read_v(0);
  if (grp == 1) F8_BARRIER();

  for (int t = 0; t < total_tiles; ++t) {
    if (!(ABL & 2)) stage();                               // tile t + 5
    softmax(MODE != 2 && t == 0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // tile t + 3 (tiles t + 4, t + 5 stay in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ABL & 64) __builtin_amdgcn_s_setprio(0);
    F8_BARRIER();
    if (!(ABL & 96)) __builtin_amdgcn_s_setprio(1);
    // Between two MFMAs the wave has one MFMA time (64 cycles) of free issue: the fragment reads (into registers whose MFMA
    // has been issued) and the accumulator-initialising moves are spread over the gaps, two reads / eight moves at most each.
    constexpr bool MM = !(ABL & 8), RD = !(ABL & 4);
    if (MM) mfma_f8_acc(o[0], vf[0], pf, one);             // O += V^T(t) P(t)
    if (MM) mfma_f8_acc(o[1], vf[1], pf, one);
    if (RD) read_v1(t + 1, 0);
    if (MM) mfma_f8_acc(o[2], vf[2], pf, one);
    if (RD) read_v1(t + 1, 1);
    if (MM) mfma_f8_acc(o[3], vf[3], pf, one);
    if (RD) read_v1(t + 1, 2);
    if (MM) mfma_f8_init(sc[0], kf[0][0], qf[0], bsplat, one);      // S(t+1) = K(t+1) Q^T + (P_SHIFT - m_run)
    if (RD) read_v1(t + 1, 3);
    if (MM) mfma_f8_init(sc[1], kf[1][0], qf[0], bsplat, one);
    if (RD) read_k1(t + 2, 0, 0);
    if (MM) mfma_f8_acc(sc[0], kf[0][1], qf[1], one);
    if (RD) read_k1(t + 2, 1, 0);
    if (MM) mfma_f8_acc(sc[1], kf[1][1], qf[1], one);
    if (RD) { read_k1(t + 2, 0, 1); read_k1(t + 2, 1, 1); }
    if (!(ABL & 96)) __builtin_amdgcn_s_setprio(0);
    F8_BARRIER();
    if (ABL & 64) __builtin_amdgcn_s_setprio(1);           // priority to the softmax interval instead
  }
  if (grp == 0) F8_BARRIER();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- normalise and store: lane (row l31, half hi) holds channels 32 cb + 8 g + 4 hi .. + 3 in registers 4 g .. 4 g + 3
  l_run += other_half(l_run, hi);
  if (MODE == 1 || MODE == 3) {          // un-normalised (O, m, l): the state of the first pass / a partial of the split last block
    float* sp = MODE == 1 ? p.state + ((int64_t)sh * p.sq_pad + qrow) * F8_STATE_LD
                          : p.part + (((int64_t)sh * gridDim.z + blockIdx.z) * 256 + (qrow - p.qblk_base * 256)) * F8_STATE_LD;
    if (MODE == 1 || qrow < p.sq) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4_t*>(sp + cb * 32 + 8 * g + 4 * hi) = f32x4_t{o[cb][4 * g], o[cb][4 * g + 1], o[cb][4 * g + 2], o[cb][4 * g + 3]};
      if (hi == 0) { sp[HD8] = m_run; sp[HD8 + 1] = l_run; }
    }
    return;
  }
  if (qrow < p.sq) {
    const float inv = 1.f / l_run;
    bf16_t* op = p.O + ((int64_t)seq * p.sq + qrow) * p.ldo + head * HD8;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const u32x2_t w = {pack_bf2(o[cb][4 * g] * inv, o[cb][4 * g + 1] * inv), pack_bf2(o[cb][4 * g + 2] * inv, o[cb][4 * g + 3] * inv)};
        *reinterpret_cast<u32x2_t*>(op + cb * 32 + 8 * g + 4 * hi) = w;
      }
  }
}

}  // namespace

int am_attention_combine_launch(const am_attn_args* a, const float* part, int Z, int qblk_base, int rows, void* stream);   // am_attention.hip

static int check_args(const am_attn_args* a, const char* who) {
  AM_CHECK(a != nullptr, "%s: null args", who);
  AM_CHECK(a->nseq > 0 && a->heads > 0 && a->sq > 0 && a->sk > 0 && a->nchunks > 0, "%s: empty problem", who);
  AM_CHECK(a->sq_pad % 256 == 0 && a->sq_pad >= a->sq, "%s: sq_pad=%d must be a multiple of 256 and >= sq=%d", who, a->sq_pad, a->sq);
  AM_CHECK(a->sk_pad % KT == 0 && a->sk_pad >= a->sk, "%s: sk_pad=%d must be a multiple of 64 and >= sk=%d", who, a->sk_pad, a->sk);
  AM_CHECK((a->nchunks == 1 && a->chunk_total == 0) || a->chunk_stride >= (int64_t)a->nseq * a->heads * a->sk_pad * HD8,
           "%s: chunk_stride too small", who);
  AM_CHECK(a->rows >= 0 && a->rows <= 2 && a->state_mode >= 0 && a->state_mode <= 2, "%s: bad rows / state_mode", who);
  AM_CHECK(a->state_mode == 0 || (a->rows == 1 && a->state != nullptr && (uintptr_t)a->state % 16 == 0),
           "%s: state_mode needs rows = 1 and a 16-byte aligned state buffer", who);
  AM_CHECK(a->chunk_total == 0 || (a->chunk_total > 0 && a->chunk_first >= 0 && a->chunk_first < a->chunk_total && a->nchunks <= a->chunk_total),
           "%s: chunk_first / chunk_total need 0 <= first < total, nchunks <= total", who);
  AM_CHECK((int64_t)a->nseq * a->heads <= 65535, "%s: nseq*heads exceeds grid.y", who);
  return AM_OK;
}

// chunk_first / chunk_total select the chunks that are quantised (the ones an attention call with the same arguments walks); Q is
// quantised unless rows == 2 (the caller has done it with the rows = 1 call of the same layer).
extern "C" int am_attention_quantize_fp8(const am_attn_args* a, uint8_t* q8, uint8_t* k8, uint8_t* vt8, void* stream) {
  AM_TRY(check_args(a, "am_attention_quantize_fp8"));
  AM_CHECK(a->K && a->Vt && k8 && vt8 && (a->rows == 2 || (a->Q && q8)), "am_attention_quantize_fp8: null operand");
  AM_CHECK(((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->Vt | (uintptr_t)q8 | (uintptr_t)k8 | (uintptr_t)vt8) % 16 == 0 &&
               a->chunk_stride % 16 == 0, "am_attention_quantize_fp8: operands misaligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t per_head_q = (int64_t)a->sq_pad * HD8, per_chunk = (int64_t)a->nseq * a->heads * a->sk_pad * HD8;
  const int64_t nq16 = (int64_t)a->nseq * a->heads * per_head_q / 16;
  const float qmul = a->scale * 1.44269504088896340736f;
  auto grid = [](int64_t n) { const int64_t b = (n + 255) / 256; return dim3((unsigned)(b < 65536 ? b : 65536)); };
  if (a->rows != 2) hipLaunchKernelGGL(quant_rows_kernel, grid(nq16), dim3(256), 0, st, a->Q, q8, nq16, qmul);
  for (int i = 0; i < a->nchunks; ++i) {
    int c = a->chunk_first + i;
    if (a->chunk_total > 0 && c >= a->chunk_total) c -= a->chunk_total;
    const int64_t off = (int64_t)c * ((a->nchunks > 1 || a->chunk_total > 0) ? a->chunk_stride : 0);
    hipLaunchKernelGGL(quant_rows_kernel, grid(per_chunk / 16), dim3(256), 0, st, a->K + off, k8 + off, per_chunk / 16, 1.0f);
    const int64_t rows = (int64_t)a->nseq * a->heads * HD8;
    const int tiles = a->sk_pad / KT;
    hipLaunchKernelGGL(quant_vt_kernel, grid(rows * tiles), dim3(256), 0, st, a->Vt + off, vt8 + off, rows, tiles, a->sk_pad);
  }
  AM_HIP(hipGetLastError());
  return AM_OK;
}

extern "C" int am_attention_fp8(const am_attn_args* a, const uint8_t* q8, const uint8_t* k8, const uint8_t* vt8, void* stream) {
  AM_TRY(check_args(a, "am_attention_fp8"));
  AM_CHECK(q8 && k8 && vt8 && a->O, "am_attention_fp8: null operand");
  AM_CHECK(((uintptr_t)q8 | (uintptr_t)k8 | (uintptr_t)vt8) % 16 == 0 && (uintptr_t)a->O % 8 == 0 && a->chunk_stride % 16 == 0,
           "am_attention_fp8: operands misaligned");
  AM_CHECK(a->ldo % 4 == 0 && a->ldo >= a->heads * HD8, "am_attention_fp8: ldo=%d too small / misaligned", a->ldo);
  AM_CHECK((int64_t)HD8 * a->sk_pad * 1 < (1ll << 31), "am_attention_fp8: sk_pad too large for 32-bit lane offsets");
  const int abl = a->defer_log2 >= 5000 ? a->defer_log2 - 5000 : 0;     // 5000 + ABL: timing ablations (one-pass form only)
  AM_CHECK(abl == 0 || (a->rows == 0 && a->state_mode == 0), "am_attention_fp8: ablation codes run the one-pass form only");
#define F8_ATTR(...) AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fp8_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE_BYTES))
  AM_ONCE_PER_DEVICE({ F8_ATTR(0, 0); F8_ATTR(1, 0); F8_ATTR(2, 0); F8_ATTR(4, 0); F8_ATTR(8, 0); F8_ATTR(16, 0); F8_ATTR(17, 0); F8_ATTR(6, 0);
                       F8_ATTR(32, 0); F8_ATTR(64, 0); F8_ATTR(40, 0); F8_ATTR(0, 1); F8_ATTR(0, 2); F8_ATTR(0, 3); });
#undef F8_ATTR
  f8_args p;
  p.Q = q8; p.K = k8; p.Vt = vt8; p.O = a->O;
  p.heads = a->heads; p.sq = a->sq; p.sq_pad = a->sq_pad; p.sk = a->sk; p.sk_pad = a->sk_pad;
  p.nchunks = a->nchunks; p.tiles_per_chunk = (a->sk + KT - 1) / KT; p.ldo = a->ldo;
  p.chunk_stride = (a->nchunks > 1 || a->chunk_total > 0) ? a->chunk_stride : 0;
  p.chunk_first = a->chunk_total > 0 ? a->chunk_first : 0; p.chunk_total = a->chunk_total;
  p.qblk_base = 0; p.state = a->state; p.part = nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int bh = a->nseq * a->heads;
  // The same main / rest boundary as the bf16 kernels (query geometry alone): a short last block (<= 128 of 256 rows, >= 9 blocks)
  // is "rest"; it is split 16 ways over the key range when the stream is long enough, and merged by attn_combine.
  const int nblk = (a->sq + 255) / 256, tail_rows = a->sq - (nblk - 1) * 256;
  const bool tail_geom = nblk >= 9 && tail_rows <= 128;
  const int all_tiles = p.nchunks * p.tiles_per_chunk;
  static float* part = nullptr;            // library-owned scratch, grown on demand (launches on a device come from one thread)
  static size_t part_elems = 0;
  const size_t need = (size_t)bh * F8_SPLIT_Z * 256 * F8_STATE_LD;
  const bool split = tail_geom && all_tiles >= 4 * F8_SPLIT_Z && need * sizeof(float) <= (256u << 20);
  const int nblk_main = tail_geom ? nblk - 1 : nblk;
#define F8_LAUNCH(A, M, GRID) hipLaunchKernelGGL((attn_fp8_kernel<A, M>), GRID, dim3(512), NSTAGE * STAGE_BYTES, st, p)
  if (a->rows != 2) {                       // the main grid
    const dim3 grid(nblk_main, bh);
    if (a->state_mode == 1) F8_LAUNCH(0, 1, grid);
    else if (a->state_mode == 2) F8_LAUNCH(0, 2, grid);
    else switch (abl) {
      case 0: F8_LAUNCH(0, 0, grid); break;
      case 1: F8_LAUNCH(1, 0, grid); break;
      case 2: F8_LAUNCH(2, 0, grid); break;
      case 4: F8_LAUNCH(4, 0, grid); break;
      case 6: F8_LAUNCH(6, 0, grid); break;
      case 8: F8_LAUNCH(8, 0, grid); break;
      case 16: F8_LAUNCH(16, 0, grid); break;
      case 17: F8_LAUNCH(17, 0, grid); break;
      case 32: F8_LAUNCH(32, 0, grid); break;
      case 64: F8_LAUNCH(64, 0, grid); break;
      case 40: F8_LAUNCH(40, 0, grid); break;
      default: AM_FAIL(AM_ERR_INVALID, "am_attention_fp8: unknown ablation code %d", a->defer_log2);
    }
  }
  if (a->rows != 1 && tail_geom) {          // the short last block
    p.qblk_base = nblk - 1;
    p.state = nullptr;
    if (split) {
      if (part_elems < need) {
        if (part) AM_HIP(hipFree(part));
        part = nullptr; part_elems = 0;
        ++g_am_scratch_generation;
        AM_HIP(hipMalloc(reinterpret_cast<void**>(&part), need * sizeof(float)));
        part_elems = need;
      }
      p.part = part;
      F8_LAUNCH(0, 3, dim3(1, bh, F8_SPLIT_Z));
      AM_TRY(am_attention_combine_launch(a, part, F8_SPLIT_Z, nblk - 1, tail_rows, stream));
    } else {
      F8_LAUNCH(0, 0, dim3(1, bh));
    }
  }
#undef F8_LAUNCH
  AM_HIP(hipGetLastError());
  return AM_OK;
}
