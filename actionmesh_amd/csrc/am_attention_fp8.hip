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

// ==================================================================================================================
// Round 4: the 4 x 64 form of the fp8 attention (VERDICT r03 weak #2 / next #1b).
//
// Why: in the 8-wave kernel above the two waves of a SIMD alternate a softmax interval (V, pure VALU: 32 exp2, 32 row-sum adds,
// 16 packs, the row max) with a matrix interval (M: 8 MFMAs of 64 cycles), a barrier apart.  An interval lasts max(V, M); V alone
// is ~840 cycles for 512 cycles of MFMA time (a wave issues one instruction per 4 cycles, v_exp_f32 holds the port for two
// slots), and beside its partner's matrix interval it stretches to ~1110: the matrix pipe is 46 % busy, "the intervals add"
// (profiles/r03f_*_pmc.csv).  The e4m3 MFMA halved the matrix time per score; the per-score VALU work did not
// shrink, so the kernel is bound by ONE wave's VALU issue while the partner's VALU sits behind a barrier.
// Here a workgroup is 4 waves = ONE wave per SIMD, each wave owns 64 query rows (blocks j = 0, 1 of 32) and the whole
// 512-register file, and every MFMA is threaded through the wave's OWN softmax stream (the structure of am_attention64.hip):
//     phase 1(t):  O_j += V^T(t-1) P_j(t-1), j = 0, 1        (8 MFMA)  ||  softmax of block 0 of tile t
//     phase 2(t):  S_j(t+1) = K(t+1) Q_j^T + (5 - m_j)        (8 MFMA)  ||  softmax of block 1 of tile t
// No interval ever waits for a partner: the VALU stream runs continuously and the MFMAs are issued inside it, 1 per ~18 issue
// slots; every K8 / V8T fragment read from LDS feeds two MFMAs (one per block): half the ds_reads, LDS-DMA pieces and barriers
// per MFMA.  O (128 registers) and the Q fragments (32) live in AccVGPRs ("a" asm operands: P.V is an AGPR-form MFMA, QK^T
// reads its B operand from the accumulator file), the arch file holds scores (block 0: one set, block 1: a ping-pong pair -
// its softmax runs under its own next QK^T), the packed P (block 0 doubled), the two 5 - m_run splats and FOUR rotating
// fragment sets shared by V^T and K (set i holds V^T channel block i in phase 1 and K fragment i in phase 2; it is refilled
// for the next phase one MFMA pair after its last use).
// Same operands, same numerics and same modes (one pass / save state / resume) as attn_fp8_kernel; the split last block (MODE 3)
// stays on the 8-wave kernel.  One barrier per tile with a counted vmcnt(8): tile t+1 has landed, tiles t+2, t+3 stay in
// flight; tile t+4 is staged in iteration t (2 K8 + 2 V8T pieces per wave, in the light row-max gaps).
// Exact online softmax (running max, deferred re-base at 2^3) - the lazy re-base of the bf16 4x64 kernel needs bf16's exponent
// range for P; e4m3 holds 2^-9 .. 448.
// ==================================================================================================================
#define X_FENCE() __builtin_amdgcn_sched_barrier(0)
#define X_PIN(x) asm volatile("" : "+v"(x))
constexpr int X_AHEAD = 4;                 // tiles staged ahead of the one whose K is being multiplied

__device__ __forceinline__ void x_pv(f32x16_t& o, const i32x8_t& v, const i32x8_t& p, int one) {            // o (AGPR) += v x p
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(o) : "v"(v), "v"(p), "v"(one) : "memory");
}
__device__ __forceinline__ void x_qk_first(f32x16_t& d, const i32x8_t& k, const i32x8_t& q, const f32x16_t& c, int one) {   // d = k x q + c
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %4 op_sel_hi:[0,0,0]" : "=&v"(d) : "v"(k), "a"(q), "v"(c), "v"(one) : "memory");
}
__device__ __forceinline__ void x_qk_acc(f32x16_t& d, const i32x8_t& k, const i32x8_t& q, int one) {       // d += k x q
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(d) : "v"(k), "a"(q), "v"(one) : "memory");
}

// row max of one 32-query block over the tile's 64 keys: 20 single-instruction steps (four v_max3 chains, the other half-lane)
struct XRowMax {
  float a[4];
  float mx;
  __device__ __forceinline__ void step(int n, const f32x16_t& sa, const f32x16_t& sb, int hi) {
    if (n < 4) { a[n] = fmaxf(fmaxf(sa[n], sb[n]), sa[n + 4]); X_PIN(a[n]); }
    else if (n < 8) { const int i = n - 4; a[i] = fmaxf(fmaxf(a[i], sb[i + 4]), sa[i + 8]); X_PIN(a[i]); }
    else if (n < 12) { const int i = n - 8; a[i] = fmaxf(fmaxf(a[i], sb[i + 8]), sa[i + 12]); X_PIN(a[i]); }
    else if (n < 16) { const int i = n - 12; a[i] = fmaxf(a[i], sb[i + 12]); X_PIN(a[i]); }
    else if (n == 16) { a[2] = fmaxf(a[2], a[3]); X_PIN(a[2]); }
    else if (n == 17) { mx = fmaxf(fmaxf(a[0], a[1]), a[2]); X_PIN(mx); }
    else if (n == 18) { a[0] = other_half(mx, hi); X_PIN(a[0]); }
    else { mx = fmaxf(mx, a[0]); X_PIN(mx); }
  }
};
// exp2 / row sum / e4m3 pack of one block as 80 single-instruction steps (32 v_exp_f32, 32 v_add_f32, 16 v_cvt_pk_fp8_f32), each
// consumer a round behind the exponentials it reads; element e = 16 kb + r of the block's two score registers sets
enum XEsKind : int { XES_EX = 0, XES_AD = 1, XES_PK = 2 };
struct XEsSeq { int n; int cost; int kind[80]; int arg[80]; };
__device__ __host__ constexpr XEsSeq x_es_seq() {
  XEsSeq q{};
  int n = 0;
  auto put = [&](int k, int a) { q.kind[n] = k; q.arg[n] = a; ++n; };
  for (int r = 0; r < 16; ++r) {
    put(XES_EX, 2 * r);
    put(XES_EX, 2 * r + 1);
    if (r >= 1) { put(XES_AD, 2 * r - 2); put(XES_AD, 2 * r - 1); put(XES_PK, r - 1); }
  }
  put(XES_AD, 30); put(XES_AD, 31); put(XES_PK, 15);
  q.n = n;
  for (int i = 0; i < n; ++i) q.cost += q.kind[i] == XES_EX ? 2 : 1;      // issue slots: v_exp_f32 holds the port for two
  return q;
}
constexpr int X_ES_GAPS = 6;               // the 80 steps go behind MFMAs 2 .. 7 of a phase
struct XEsTab { int lo[X_ES_GAPS + 1]; };
__device__ __host__ constexpr XEsTab x_es_tab(const XEsSeq& q) {
  XEsTab t{};
  for (int i = 0; i <= X_ES_GAPS; ++i) {
    int n = 0, c = 0;
    while (n < q.n && c * X_ES_GAPS < i * q.cost) { c += q.kind[n] == XES_EX ? 2 : 1; ++n; }
    t.lo[i] = n;
  }
  t.lo[X_ES_GAPS] = q.n;
  return t;
}
template <int ABL>
struct XExpSumPack {
  static constexpr XEsSeq SEQ = x_es_seq();
  float rs[4];
  __device__ __forceinline__ void init() { rs[0] = rs[1] = rs[2] = rs[3] = 0.f; }
  __device__ __forceinline__ static float get(const f32x16_t& sa, const f32x16_t& sb, int e) { return e < 16 ? sa[e] : sb[e - 16]; }
  __device__ __forceinline__ void step(int n, f32x16_t& sa, f32x16_t& sb, i32x8_t& w) {
    const int k = SEQ.kind[n], a = SEQ.arg[n];
    if (k == XES_EX) {
      float v = (ABL & 1) ? get(sa, sb, a) : __builtin_amdgcn_exp2f(get(sa, sb, a));
      X_PIN(v);
      if (a < 16) sa[a] = v; else sb[a - 16] = v;
    } else if (k == XES_AD) {
      rs[a & 3] += get(sa, sb, a);
      X_PIN(rs[a & 3]);
    } else {                              // pair a = elements 2a, 2a+1 -> half (a & 1) of word a >> 1 (byte j = 16 kb + r of the B operand)
      int x = w[a >> 1];
      x = (a & 1) ? __builtin_amdgcn_cvt_pk_fp8_f32(get(sa, sb, 2 * a), get(sa, sb, 2 * a + 1), x, true)
                  : __builtin_amdgcn_cvt_pk_fp8_f32(get(sa, sb, 2 * a), get(sa, sb, 2 * a + 1), x, false);
      X_PIN(x);
      w[a >> 1] = x;
    }
  }
  __device__ __forceinline__ float total() const { return (rs[0] + rs[1]) + (rs[2] + rs[3]); }
};

// ABL: 1 no exponentials, 2 no LDS-DMA in the loop, 8 no softmax steps at all (timing ablations, numerically meaningless).
// MODE: 0 one pass, 1 save the un-normalised (O, m, l), 2 resume from it (attn_fp8_kernel above).  PROF: s_memtime stamps.
template <int ABL, int MODE, bool PROF = false>
__global__ __launch_bounds__(256, 1) void attn_fp8x64_kernel(f8_args p, unsigned long long* prof) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int qblk = p.qblk_base + blockIdx.x, sh = blockIdx.y;          // sh = seq * heads + head
  const int seq = sh / p.heads, head = sh - seq * p.heads;
  const int q0 = qblk * 256 + wave * 64;                               // block j: rows q0 + 32 j + l31
  auto stamp = [&](int t, int slot) __attribute__((always_inline)) {
    if (PROF && blockIdx.x == 0 && sh == 0 && t >= 64 && t < 72) {
      unsigned long long c;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c));
      if (lane == 0) prof[(wave * 8 + (t - 64)) * 8 + slot] = c;
    }
  };

  // ---- Q fragments of both blocks (B operand of QK^T), parked in AccVGPRs: lane (row l31, half hi), k-step s: channels 64 s + 32 hi ..
  i32x8_t qf[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const uint8_t* qp = p.Q + ((int64_t)sh * p.sq_pad + q0 + 32 * j + l31) * HD8 + hi * 32;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const u32x4_t a = *reinterpret_cast<const u32x4_t*>(qp + s * 64);
      const u32x4_t b = *reinterpret_cast<const u32x4_t*>(qp + s * 64 + 16);
      qf[j][s] = i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int s = 0; s < 2; ++s) asm volatile("" : "+a"(qf[j][s]));       // landed (and in the accumulator file) before any LDS-DMA is counted

  // ---- accumulators and softmax state
  f32x16_t o[2][4];
  float m_run[2] = {P_SHIFT, P_SHIFT}, l_run[2] = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[j][cb][r] = 0.f;
  if (MODE == 2) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float* sp = p.state + ((int64_t)sh * p.sq_pad + q0 + 32 * j + l31) * F8_STATE_LD;
      m_run[j] = sp[HD8];
      l_run[j] = hi == 0 ? sp[HD8 + 1] : 0.f;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4_t t4 = *reinterpret_cast<const f32x4_t*>(sp + cb * 32 + 8 * g + 4 * hi);
#pragma unroll
          for (int i = 0; i < 4; ++i) o[j][cb][4 * g + i] = t4[i];
        }
        asm volatile("" : "+a"(o[j][cb]));
      }
      asm volatile("" : "+v"(m_run[j]), "+v"(l_run[j]));
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) asm volatile("" : "+a"(o[j][cb]));
  int one = SCALE_ONE;
  asm volatile("" : "+v"(one));
  f32x16_t bs[2];                                           // P_SHIFT - m_run splats: SrcC of the first QK^T MFMAs, rewritten on a re-base only
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) bs[j][r] = P_SHIFT - m_run[j];
    X_PIN(bs[j]);
  }

  // ---- LDS-DMA: a tile = 8 pieces of K8 (8 key rows each) + 8 of V8T (16 channel rows each), 1 KiB per wave-instruction; wave w
  // moves pieces w and w + 4 of each.  Lane-linear destination, swizzle on the source unit (K: unit ^ ((row >> 1) & 7), V^T:
  // unit ^ ((row >> 2) & 3)) - both are the same for pieces w and w + 4, so one lane offset serves both.
  const int all_tiles = p.nchunks * p.tiles_per_chunk;
  const int kr = wave * 8 + (lane >> 3);
  const uint32_t k_lane_off = (uint32_t)kr * HD8 + (uint32_t)(((lane & 7) ^ ((kr >> 1) & 7)) << 4);
  const int vr = wave * 16 + (lane >> 2);
  const uint32_t v_lane_off = (uint32_t)vr * (uint32_t)p.sk_pad + (uint32_t)(((lane & 3) ^ ((vr >> 2) & 3)) << 4);
  const uint8_t* base_k = p.K + (int64_t)sh * p.sk_pad * HD8;
  const uint8_t* base_v = p.Vt + (int64_t)sh * p.sk_pad * HD8;
  int st_n = 0, st_ti = 0, st_pos = 0;
  auto chunk_off = [&](int pos) __attribute__((always_inline)) {
    int c = p.chunk_first + pos;
    if (p.chunk_total > 0 && c >= p.chunk_total) c -= p.chunk_total;
    return (int64_t)c * p.chunk_stride;
  };
  const uint8_t* st_k = base_k + chunk_off(0);
  const uint8_t* st_v = base_v + chunk_off(0);
  const unsigned smem_lds = (unsigned)reinterpret_cast<uintptr_t>((lds_ptr_t)smem);
  auto uniform = [](const uint8_t* q) __attribute__((always_inline)) {
    const uint64_t u = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hh = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<const uint8_t*>(((uint64_t)hh << 32) | lo);
  };
  // one piece: scalar base + 32-bit lane offset -> LDS at (M0) + lane * 16; asm: hipcc neither tracks it nor pads it (the counted
  // vmcnt in front of the tile barrier is the only wait there is)
  auto dma_piece = [&](const uint8_t* src, uint32_t lane_off, unsigned lds_off) __attribute__((always_inline)) {
    const unsigned lds = __builtin_amdgcn_readfirstlane(smem_lds + lds_off);
    const uint8_t* s = uniform(src);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(s), "s"(lds) : "memory");
  };
  auto stage_k = [&](int i) __attribute__((always_inline)) {       // piece wave + 4 i of the K8 tile under the cursor
    dma_piece(st_k + i * (32 * HD8), k_lane_off, (unsigned)((st_n & (NSTAGE - 1)) * STAGE_BYTES + (wave + 4 * i) * 1024));
  };
  auto stage_v = [&](int i) __attribute__((always_inline)) {
    dma_piece(st_v + (int64_t)i * 64 * p.sk_pad, v_lane_off, (unsigned)((st_n & (NSTAGE - 1)) * STAGE_BYTES + 8192 + (wave + 4 * i) * 1024));
  };
  auto stage_advance = [&]() __attribute__((always_inline)) {      // past the end of the stream the cursor stays on the last tile
    ++st_n;
    if (st_n < all_tiles) {
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

  // ---- fragment addresses (as in attn_fp8_kernel): K8 fragment n = (kb = n & 1, s = n >> 1): key 32 kb + l31, units 4 s + 2 hi + {0, 1};
  // V8T fragment cb: channel 32 cb + l31, units 2 hi + {0, 1}
  const int ksw = (l31 >> 1) & 7, vsw = (l31 >> 2) & 3;
  int k_off[2][2], v_off[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 2; ++e) k_off[s][e] = l31 * HD8 + (((4 * s + 2 * hi + e) ^ ksw) << 4);
#pragma unroll
  for (int e = 0; e < 2; ++e) v_off[e] = 8192 + l31 * 64 + (((2 * hi + e) ^ vsw) << 4);
  auto k_frag = [&](int tt, int n) __attribute__((always_inline)) {
    const unsigned char* slot = smem + (tt & (NSTAGE - 1)) * STAGE_BYTES + (n & 1) * 32 * HD8;
    const u32x4_t a = *reinterpret_cast<const u32x4_t*>(slot + k_off[n >> 1][0]);
    const u32x4_t b = *reinterpret_cast<const u32x4_t*>(slot + k_off[n >> 1][1]);
    return i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
  };
  auto v_frag = [&](int tt, int cb) __attribute__((always_inline)) {
    const unsigned char* slot = smem + (tt & (NSTAGE - 1)) * STAGE_BYTES + cb * 32 * 64;
    const u32x4_t a = *reinterpret_cast<const u32x4_t*>(slot + v_off[0]);
    const u32x4_t b = *reinterpret_cast<const u32x4_t*>(slot + v_off[1]);
    return i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
  };

  // ---- pipeline registers (arch file)
  f32x16_t s0[2];              // block 0: S(t) on entry of an iteration, S(t+1) on exit
  f32x16_t s1[2][2];           // block 1: ping-pong, S(t) in s1[cur], S(t+1) born in s1[cur ^ 1]
  i32x8_t p0[2], p1;           // packed P: block 0 doubled (P(t) is born while P(t-1) is multiplied), block 1 single
  i32x8_t fr[4];               // rotating fragment sets
  const i32x8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  p0[0] = p0[1] = p1 = zero8;
  X_PIN(p0[0]); X_PIN(p0[1]); X_PIN(p1);
  const int tail_valid = p.sk - (p.tiles_per_chunk - 1) * KT;
  int tic = 0;

  // ---- prologue: tiles 0 .. 3 in flight; V8T of ring slot 7 zeroed (iteration 0 reads "V8T(-1)" fragment 3 from it: P(-1) = 0, but
  // 0 x NaN byte patterns would poison O); S(0) of both blocks
#pragma unroll
  for (int a = 0; a < X_AHEAD; ++a) {
    stage_k(0); stage_k(1); stage_v(0); stage_v(1);
    stage_advance();
  }
  {
    unsigned char* z = smem + 7 * STAGE_BYTES + 8192 + tid * 32;
    *reinterpret_cast<u32x4_t*>(z) = u32x4_t{0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4_t*>(z + 16) = u32x4_t{0u, 0u, 0u, 0u};
  }
  asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");       // tile 0 (tiles 1 .. 3 = 12 pieces stay in flight)
#pragma unroll
  for (int n = 0; n < 4; ++n) fr[n] = k_frag(0, n);
  x_qk_first(s0[0], fr[0], qf[0][0], bs[0], one);
  x_qk_first(s0[1], fr[1], qf[0][0], bs[0], one);
  x_qk_first(s1[0][0], fr[0], qf[1][0], bs[1], one);
  x_qk_first(s1[0][1], fr[1], qf[1][0], bs[1], one);
  x_qk_acc(s0[0], fr[2], qf[0][1], one);
  x_qk_acc(s0[1], fr[3], qf[0][1], one);
  x_qk_acc(s1[0][0], fr[2], qf[1][1], one);
  x_qk_acc(s1[0][1], fr[3], qf[1][1], one);
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]));   // operands fetched
#pragma unroll
  for (int n = 0; n < 4; ++n) { fr[n] = zero8; X_PIN(fr[n]); }       // "V8T(-1)" = 0

  // rare: keys past the end of a chunk (its last tile), once per chunk
  auto mask_tail = [&](f32x16_t& sa, f32x16_t& sb) __attribute__((always_inline)) {
    int tv = tail_valid - 4 * hi;
    asm volatile("" : "+v"(tv));          // the 32 lane masks are computed here, in the rare branch, not held in 64 SGPRs across the loop
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if ((r & 3) + 8 * (r >> 2) >= tv) sa[r] = -INFINITY;
      if (32 + (r & 3) + 8 * (r >> 2) >= tv) sb[r] = -INFINITY;
    }
  };
  auto o_settle = [&](int j) __attribute__((always_inline)) {          // in-flight P.V results of block j have landed (asm MFMAs: nothing is padded)
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+a"(o[j][0]), "+a"(o[j][1]), "+a"(o[j][2]), "+a"(o[j][3]));
  };
  auto o_scale = [&](int j, float alpha) __attribute__((always_inline)) {
    o_settle(j);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[j][cb][r] *= alpha;
      asm volatile("" : "+a"(o[j][cb]));
      X_FENCE();                          // the rare path runs with every register of the loop live: 16 temporaries at a time
    }
  };

  // ---- one tile
  auto iteration = [&](const int t, f32x16_t (&s1c)[2], f32x16_t (&s1n)[2], i32x8_t& p0c, i32x8_t& p0n) __attribute__((always_inline)) {
    stamp(t, 0);
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");      // tile t+1 has landed everywhere (tiles t+2, t+3 in flight)
    stamp(t, 1);
    const bool first = MODE != 2 && t == 0;
    const bool last_of_chunk = tic == p.tiles_per_chunk - 1;
    tic = last_of_chunk ? 0 : tic + 1;
    const bool masked = last_of_chunk && tail_valid < KT;
    XRowMax rm;
    XExpSumPack<ABL> es;
    constexpr XEsTab ES = x_es_tab(es.SEQ);
    auto es_gap = [&](int gap, f32x16_t& sa, f32x16_t& sb, i32x8_t& w) __attribute__((always_inline)) {
      if (!(ABL & 8))
#pragma unroll
        for (int n = ES.lo[gap]; n < ES.lo[gap + 1]; ++n) es.step(n, sa, sb, w);
    };
    auto rm_gap = [&](int half, const f32x16_t& sa, const f32x16_t& sb) __attribute__((always_inline)) {
#pragma unroll
      for (int n = 10 * half; n < 10 * half + 10; ++n) rm.step(n, sa, sb, hi);
    };

    // ===== phase 1: O_j += V^T(t-1) P_j(t-1) || softmax of block 0; K8(t+4) staged; K8(t+1) fragments fetched =====
    if (masked) { asm volatile("" ::: "memory"); mask_tail(s0[0], s0[1]); }
    X_FENCE();
    x_pv(o[0][0], fr[0], p0c, one);
    if (!(ABL & 2)) stage_k(0);
    rm_gap(0, s0[0], s0[1]);
    X_FENCE();
    x_pv(o[1][0], fr[0], p1, one);
    if (!(ABL & 2)) stage_k(1);
    fr[3] = v_frag(t + 7, 3);                       // V8T(t-1), the set K8(t) fragment 3 left one pair ago   ((t - 1) & 7 == (t + 7) & 7)
    rm_gap(1, s0[0], s0[1]);
    X_FENCE();
    stamp(t, 2);
    bool flag0 = false;
    float alpha0 = 1.f;
    if (first || __builtin_amdgcn_ballot_w64(rm.mx > P_SHIFT + DEFER_T) != 0) {     // rare: re-base block 0
      const float delta = first ? rm.mx - P_SHIFT : fmaxf(rm.mx - P_SHIFT, 0.f);
      alpha0 = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
      l_run[0] *= alpha0;
      m_run[0] += delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[0][r] -= delta; s0[1][r] -= delta; bs[0][r] = P_SHIFT - m_run[0]; }
      flag0 = true;                                  // O_0 is scaled at the end of the phase: its P.V of tile t-1 is being issued now
    }
    es.init();
    X_FENCE();
    x_pv(o[0][1], fr[1], p0c, one);
    es_gap(0, s0[0], s0[1], p0n);
    X_FENCE();
    x_pv(o[1][1], fr[1], p1, one);
    fr[0] = k_frag(t + 1, 0);
    es_gap(1, s0[0], s0[1], p0n);
    X_FENCE();
    x_pv(o[0][2], fr[2], p0c, one);
    es_gap(2, s0[0], s0[1], p0n);
    X_FENCE();
    x_pv(o[1][2], fr[2], p1, one);
    fr[1] = k_frag(t + 1, 1);
    es_gap(3, s0[0], s0[1], p0n);
    X_FENCE();
    x_pv(o[0][3], fr[3], p0c, one);
    es_gap(4, s0[0], s0[1], p0n);
    X_FENCE();
    x_pv(o[1][3], fr[3], p1, one);
    fr[2] = k_frag(t + 1, 2);
    es_gap(5, s0[0], s0[1], p0n);
    X_FENCE();
    stamp(t, 3);
    l_run[0] += es.total();
    if (flag0) o_scale(0, alpha0);

    // ===== phase 2: S_j(t+1) = K8(t+1) Q_j^T || softmax of block 1; V8T(t+4) staged; V8T(t) fragments fetched =====
    if (masked) { asm volatile("" ::: "memory"); mask_tail(s1c[0], s1c[1]); }
    X_FENCE();
    x_qk_first(s0[0], fr[0], qf[0][0], bs[0], one);
    if (!(ABL & 2)) stage_v(0);
    rm_gap(0, s1c[0], s1c[1]);
    X_FENCE();
    x_qk_first(s0[1], fr[1], qf[0][0], bs[0], one);
    if (!(ABL & 2)) stage_v(1);
    fr[3] = k_frag(t + 1, 3);
    rm_gap(1, s1c[0], s1c[1]);
    X_FENCE();
    stamp(t, 4);
    if (first || __builtin_amdgcn_ballot_w64(rm.mx > P_SHIFT + DEFER_T) != 0) {     // rare: re-base block 1 (its QK^T MFMAs come behind)
      const float delta = first ? rm.mx - P_SHIFT : fmaxf(rm.mx - P_SHIFT, 0.f);
      const float a1 = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
      l_run[1] *= a1;
      m_run[1] += delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s1c[0][r] -= delta; s1c[1][r] -= delta; bs[1][r] = P_SHIFT - m_run[1]; }
      o_scale(1, a1);                                // O_1 is complete through tile t-1
    }
    es.init();
    X_FENCE();
    x_qk_first(s1n[0], fr[0], qf[1][0], bs[1], one);
    es_gap(0, s1c[0], s1c[1], p1);
    X_FENCE();
    x_qk_first(s1n[1], fr[1], qf[1][0], bs[1], one);
    es_gap(1, s1c[0], s1c[1], p1);
    X_FENCE();
    x_qk_acc(s0[0], fr[2], qf[0][1], one);
    fr[0] = v_frag(t, 0);
    es_gap(2, s1c[0], s1c[1], p1);
    X_FENCE();
    x_qk_acc(s1n[0], fr[2], qf[1][1], one);
    fr[1] = v_frag(t, 1);
    es_gap(3, s1c[0], s1c[1], p1);
    X_FENCE();
    x_qk_acc(s0[1], fr[3], qf[0][1], one);
    es_gap(4, s1c[0], s1c[1], p1);
    X_FENCE();
    x_qk_acc(s1n[1], fr[3], qf[1][1], one);
    fr[2] = v_frag(t, 2);
    es_gap(5, s1c[0], s1c[1], p1);
    X_FENCE();
    stamp(t, 5);
    l_run[1] += es.total();
    if (!(ABL & 2)) stage_advance();
  };

  int t = 0;
  for (; t + 1 < all_tiles; t += 2) {
    iteration(t, s1[0], s1[1], p0[0], p0[1]);
    iteration(t + 1, s1[1], s1[0], p0[1], p0[0]);
  }
  if (t < all_tiles) iteration(t, s1[0], s1[1], p0[0], p0[1]);
  const bool odd = (all_tiles & 1) != 0;
  // ---- drain: the scores of the tile past the end are still in flight (their registers are held until they land); O += V^T(n-1) P(n-1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA piece may outlive the workgroup's LDS allocation
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(s0[0]), "+v"(s0[1]), "+v"(s1[0][0]), "+v"(s1[0][1]), "+v"(s1[1][0]), "+v"(s1[1][1]));
  fr[3] = v_frag(all_tiles - 1, 3);
  {
    const i32x8_t& p0last = odd ? p0[1] : p0[0];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      x_pv(o[0][cb], fr[cb], p0last, one);
      x_pv(o[1][cb], fr[cb], p1, one);
    }
  }
  o_settle(0);
  o_settle(1);

  // ---- normalise and store: lane (row l31, half hi) holds channels 32 cb + 8 g + 4 hi .. + 3 in registers 4 g .. 4 g + 3
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int qrow = q0 + 32 * j + l31;
    float l = l_run[j];
    l += other_half(l, hi);
    if (MODE == 1) {
      float* sp = p.state + ((int64_t)sh * p.sq_pad + qrow) * F8_STATE_LD;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4_t*>(sp + cb * 32 + 8 * g + 4 * hi) = f32x4_t{o[j][cb][4 * g], o[j][cb][4 * g + 1], o[j][cb][4 * g + 2], o[j][cb][4 * g + 3]};
      if (hi == 0) { sp[HD8] = m_run[j]; sp[HD8 + 1] = l; }
      continue;
    }
    if (qrow < p.sq) {
      const float inv = 1.f / l;
      bf16_t* op = p.O + ((int64_t)seq * p.sq + qrow) * p.ldo + head * HD8;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const u32x2_t w = {pack_bf2(o[j][cb][4 * g] * inv, o[j][cb][4 * g + 1] * inv), pack_bf2(o[j][cb][4 * g + 2] * inv, o[j][cb][4 * g + 3] * inv)};
          *reinterpret_cast<u32x2_t*>(op + cb * 32 + 8 * g + 4 * hi) = w;
        }
    }
  }
}

// ==================================================================================================================
// The product form (round 4): 8 waves x 32 rows again - TWO waves per SIMD - but FREE-RUNNING: each wave threads its 8 MFMAs through
// its own softmax stream as the 4 x 64 form above does (4 P.V MFMAs of tile t-1 and 4 QK^T MFMAs of tile t+1 inside the softmax of
// tile t), and there is ONE barrier per tile (LDS-DMA visibility) instead of the two that forced the ping-pong.  What round 4's
// measurements said (profiles/r04b_*): the fp8 attention is bound by the VALU PIPE, not by issue slots - v_exp_f32 is a quarter-rate
// instruction (16 cycles per wave64; 64 of them per 64 x 64 score tile = 1024 cycles = the tile's whole MFMA time) and the row sums,
// packs and row max add ~550 more: ~1570 VALU cycles per SIMD and tile against 1024 MFMA cycles.  The ping-pong kernel leaves that
// pipe idle whenever its softmax wave stalls (its partner is in its matrix interval: 8 MFMAs, 16 ds_reads, no VALU to offer) - 2226
// cycles per tile; the 4 x 64 form has nobody to cover ANY stall of its single wave (~150 cycles per LDS-DMA piece, the barrier, the
// scalar tail: 3400 cycles per tile, 12 % slower).  Two free-running waves per SIMD cover each other's stalls with VALU work.
// Registers: 256 per wave; O (64) and Q (16) in AccVGPRs, scores ping-pong (S(t+1) is born while S(t) is exponentiated), P doubled,
// four rotating fragment sets (V^T channel block i at MFMA i, K fragment i at MFMA 4 + i).
// ==================================================================================================================
// all-arch-VGPR forms of the MFMA wrappers for the two-waves-per-SIMD kernel below: as soon as a kernel names an AccVGPR hipcc splits a
// 256-register budget 128 / 128 and spills the arch side into the accumulator side; without one it hands out all 256 as arch VGPRs
__device__ __forceinline__ void p_pv(f32x16_t& o, const i32x8_t& v, const i32x8_t& p, int one) {
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(o) : "v"(v), "v"(p), "v"(one) : "memory");
}
__device__ __forceinline__ void p_qk_first(f32x16_t& d, const i32x8_t& k, const i32x8_t& q, const f32x16_t& c, int one) {
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %4 op_sel_hi:[0,0,0]" : "=&v"(d) : "v"(k), "v"(q), "v"(c), "v"(one) : "memory");
}
__device__ __forceinline__ void p_qk_acc(f32x16_t& d, const i32x8_t& k, const i32x8_t& q, int one) {
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(d) : "v"(k), "v"(q), "v"(one) : "memory");
}
// exp2 + e4m3 pack of one 16-score register set as single-instruction steps (the row sums are NOT here: they run on the matrix pipe,
// l += ones x P^T, from the ROUNDED probabilities).  FAST = 0: 16 v_exp_f32 + 8 v_cvt_pk_fp8_f32, each pack a round behind its
// exponentials.  FAST = 1 (attn_dtype "fp8_fast"): the scores are born as z = 8 (s - m_run + 5) + 56 and ONE v_cvt_pk_u8_f32 per score
// (round to nearest even, saturating at 0) writes the e4m3 BYTE whose exponent field is floor(z / 8) and whose mantissa is z mod 8:
// p = 2^n (1 + f) instead of 2^(n + f) - the exponential interpolated linearly between powers of two (exact at integer scores, at most
// 6.1 % high in between, the same function in the numerator and in the row sum).  No transcendental instruction at all.
template <int ABL, int FAST>
struct PHalf {
  static constexpr int N = FAST ? 16 : 24;
  // Step n as a TEMPLATE parameter: every register / word index below is a constant at the first IR pass (with a run-time n the packed-P
  // words were addressed through memory and stayed in scratch even after the loops had been unrolled).
  // exact: steps 0, 1 = EX 0, EX 1; then for r = 1 .. 7: EX 2r, EX 2r+1, PK r-1; last: PK 7.   fast: step e = CV e.
  template <int n>
  __device__ __forceinline__ static void step(f32x16_t& s, i32x8_t& w, int half) {
    if constexpr (FAST != 0) {
      constexpr int word = n >> 2;
      int x = half ? w[4 + word] : w[word];
      if constexpr ((n & 3) == 0) asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(x) : "v"(s[n]));
      else if constexpr ((n & 3) == 1) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(x) : "v"(s[n]));
      else if constexpr ((n & 3) == 2) asm volatile("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(x) : "v"(s[n]));
      else asm volatile("v_cvt_pk_u8_f32 %0, %1, 3, %0" : "+v"(x) : "v"(s[n]));
      if (half) w[4 + word] = x; else w[word] = x;
    } else {
      constexpr bool is_pk = n == 23 || (n >= 2 && (n - 2) % 3 == 2);
      if constexpr (!is_pk) {
        constexpr int e = n < 2 ? n : 2 * ((n - 2) / 3 + 1) + (n - 2) % 3;
        float v = (ABL & 1) ? s[e] : __builtin_amdgcn_exp2f(s[e]);
        X_PIN(v);
        s[e] = v;
      } else {                                     // pair pr = elements 2 pr, 2 pr + 1 -> half (pr & 1) of word 4 half + (pr >> 1)
        constexpr int pr = n == 23 ? 7 : (n - 2) / 3;
        constexpr int word = pr >> 1;
        int x = half ? w[4 + word] : w[word];
        if constexpr ((pr & 1) != 0) x = __builtin_amdgcn_cvt_pk_fp8_f32(s[2 * pr], s[2 * pr + 1], x, true);
        else x = __builtin_amdgcn_cvt_pk_fp8_f32(s[2 * pr], s[2 * pr + 1], x, false);
        X_PIN(x);
        if (half) w[4 + word] = x; else w[word] = x;
      }
    }
  }
  template <int LO, int HI>
  __device__ __forceinline__ static void run(f32x16_t& s, i32x8_t& w, int half) {
    if constexpr (LO < HI) {
      step<LO>(s, w, half);
      run<LO + 1, HI>(s, w, half);
    }
  }
};

constexpr int P_LDS_ONES = NSTAGE * STAGE_BYTES;       // 2 KiB of e4m3 1.0 behind the ring: the A operand of the row-sum MFMA
constexpr int P_LDS_BYTES = P_LDS_ONES + 2048;

template <int ABL, int MODE, int FAST = 0, bool PROF = false>
__global__ __launch_bounds__(512, 2) void attn_fp8p_kernel(f8_args p, unsigned long long* prof) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int qblk = p.qblk_base + blockIdx.x, sh = blockIdx.y;
  const int seq = sh / p.heads, head = sh - seq * p.heads;
  const int qrow = qblk * 256 + wave * 32 + l31;
  auto stamp = [&](int t, int slot) __attribute__((always_inline)) {
    if (PROF && blockIdx.x == 0 && sh == 0 && t >= 64 && t < 72) {
      unsigned long long c;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c));
      if (lane == 0) prof[(wave * 8 + (t - 64)) * 8 + slot] = c;
    }
  };
  // score units: u = SC (s - m_run + P_SHIFT) + OFF  (exact: log2 units; fast: eighths of an octave, biased so that u IS the e4m3 byte)
  constexpr float SC = FAST ? 8.f : 1.f, OFF = FAST ? 56.f : 0.f;
  constexpr float TOP = SC * P_SHIFT + OFF, THR = SC * (P_SHIFT + DEFER_T) + OFF;

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
    for (int s = 0; s < 2; ++s) asm volatile("" : "+v"(qf[s]));
  }
  f32x16_t o[4], lacc;                                       // lacc: every register = the row sum of this lane's query (ones x P^T)
  float m_run = P_SHIFT;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[cb][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
  if (MODE == 2) {
    const float* sp = p.state + ((int64_t)sh * p.sq_pad + qrow) * F8_STATE_LD;
    m_run = sp[HD8];
    const float l0 = sp[HD8 + 1];
#pragma unroll
    for (int r = 0; r < 16; ++r) lacc[r] = l0;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t t4 = *reinterpret_cast<const f32x4_t*>(sp + cb * 32 + 8 * g + 4 * hi);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[cb][4 * g + i] = t4[i];
      }
    asm volatile("" : "+v"(m_run));
  }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) asm volatile("" : "+v"(o[cb]));
  asm volatile("" : "+v"(lacc));
  int one = SCALE_ONE;
  int qk_scale = FAST ? 0x82828282 : SCALE_ONE;              // E8M0 2^3: the fast form's scores are born in eighths of an octave
  asm volatile("" : "+v"(one), "+v"(qk_scale));
  f32x16_t bs;
#pragma unroll
  for (int r = 0; r < 16; ++r) bs[r] = SC * (P_SHIFT - m_run) + OFF;
  X_PIN(bs);

  // ---- LDS-DMA: per tile every wave moves one 1 KiB piece of K8 (8 key rows) and one of V8T (16 channel rows)
  const int all_tiles = p.nchunks * p.tiles_per_chunk;
  const int kr = wave * 8 + (lane >> 3);
  const uint32_t k_lane_off = (uint32_t)kr * HD8 + (uint32_t)(((lane & 7) ^ ((kr >> 1) & 7)) << 4);
  const int vr = wave * 16 + (lane >> 2);
  const uint32_t v_lane_off = (uint32_t)vr * (uint32_t)p.sk_pad + (uint32_t)(((lane & 3) ^ ((vr >> 2) & 3)) << 4);
  const uint8_t* base_k = p.K + (int64_t)sh * p.sk_pad * HD8;
  const uint8_t* base_v = p.Vt + (int64_t)sh * p.sk_pad * HD8;
  int st_n = 0, st_ti = 0, st_pos = 0;
  auto chunk_off = [&](int pos) __attribute__((always_inline)) {
    int c = p.chunk_first + pos;
    if (p.chunk_total > 0 && c >= p.chunk_total) c -= p.chunk_total;
    return (int64_t)c * p.chunk_stride;
  };
  const uint8_t* st_k = base_k + chunk_off(0);
  const uint8_t* st_v = base_v + chunk_off(0);
  const unsigned smem_lds = (unsigned)reinterpret_cast<uintptr_t>((lds_ptr_t)smem);
  auto uniform = [](const uint8_t* q) __attribute__((always_inline)) {
    const uint64_t u = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hh = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<const uint8_t*>(((uint64_t)hh << 32) | lo);
  };
  auto dma_piece = [&](const uint8_t* src, uint32_t lane_off, unsigned lds_off) __attribute__((always_inline)) {
    const unsigned lds = __builtin_amdgcn_readfirstlane(smem_lds + lds_off);
    const uint8_t* s = uniform(src);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(s), "s"(lds) : "memory");
  };
  auto stage_k = [&]() __attribute__((always_inline)) { dma_piece(st_k, k_lane_off, (unsigned)((st_n & (NSTAGE - 1)) * STAGE_BYTES + wave * 1024)); };
  auto stage_v = [&]() __attribute__((always_inline)) { dma_piece(st_v, v_lane_off, (unsigned)((st_n & (NSTAGE - 1)) * STAGE_BYTES + 8192 + wave * 1024)); };
  auto stage_advance = [&]() __attribute__((always_inline)) {
    ++st_n;
    if (st_n < all_tiles) {
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

  const int ksw = (l31 >> 1) & 7, vsw = (l31 >> 2) & 3;
  int k_off[2][2], v_off[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 2; ++e) k_off[s][e] = l31 * HD8 + (((4 * s + 2 * hi + e) ^ ksw) << 4);
#pragma unroll
  for (int e = 0; e < 2; ++e) v_off[e] = 8192 + l31 * 64 + (((2 * hi + e) ^ vsw) << 4);
  bool in_loop = false;
  i32x8_t stale = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};
  X_PIN(stale);
  auto k_frag = [&](int tt, int kb, int s2) __attribute__((always_inline)) {
    if ((ABL & 32) && in_loop) return stale;
    const unsigned char* slot = smem + (tt & (NSTAGE - 1)) * STAGE_BYTES + kb * 32 * HD8;
    const u32x4_t a = *reinterpret_cast<const u32x4_t*>(slot + k_off[s2][0]);
    const u32x4_t b = *reinterpret_cast<const u32x4_t*>(slot + k_off[s2][1]);
    return i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
  };
  auto v_frag = [&](int tt, int cb) __attribute__((always_inline)) {
    if ((ABL & 32) && in_loop) return stale;
    const unsigned char* slot = smem + (tt & (NSTAGE - 1)) * STAGE_BYTES + cb * 32 * 64;
    const u32x4_t a = *reinterpret_cast<const u32x4_t*>(slot + v_off[0]);
    const u32x4_t b = *reinterpret_cast<const u32x4_t*>(slot + v_off[1]);
    return i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
  };
  auto ones_frag = [&]() __attribute__((always_inline)) {
    if ((ABL & 32) && in_loop) return stale;
    // every byte of the block is e4m3 1.0, so every lane reads the SAME 32 bytes: a broadcast, which the LDS serves without a bank
    // conflict.  Round 4 read lane * 32 - a 32-byte lane stride, lanes i and i + 8 of every 16-lane group on the same banks: two of
    // the 18 fragment reads of a tile two-way conflicted = the 10 % SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE the r04 PMC pass
    // showed where the bf16 kernel has ~0
    const unsigned char* q = smem + P_LDS_ONES;
    const u32x4_t a = *reinterpret_cast<const u32x4_t*>(q);
    const u32x4_t b = *reinterpret_cast<const u32x4_t*>(q + 16);
    return i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
  };

  // Scores: the kb = 0 half (keys 0-31 of the tile) lives in ONE register set - its next tile's QK^T MFMAs are issued behind the last
  // step that reads it - and only the kb = 1 half is a ping-pong pair: 48 registers instead of 64, which is what pays for lacc.
  f32x16_t sa, sb[2];
  i32x8_t pf[2];               // P(t-1) in pf[cur] (being multiplied), P(t) born in pf[cur ^ 1]
  i32x8_t fa, fb, fc;          // THREE rotating fragment sets for the 9 MFMAs of a tile: MFMA j reads set j % 3, refilled right behind it
  const i32x8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  pf[0] = pf[1] = zero8;
  X_PIN(pf[0]); X_PIN(pf[1]);
  const int tail_valid = p.sk - (p.tiles_per_chunk - 1) * KT;
  int tic = 0;

  // ---- prologue: tiles 0 .. 3 in flight; V8T of ring slot 7 zeroed ("V8T(-1)": P(-1) = 0, but stale LDS bytes may be NaN patterns);
  // the block of ones
#pragma unroll
  for (int a = 0; a < X_AHEAD; ++a) { stage_k(); stage_v(); stage_advance(); }
  *reinterpret_cast<u32x4_t*>(smem + 7 * STAGE_BYTES + 8192 + tid * 16) = u32x4_t{0u, 0u, 0u, 0u};
  *reinterpret_cast<uint32_t*>(smem + P_LDS_ONES + tid * 4) = 0x38383838u;
  asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");        // tile 0 (tiles 1 .. 3 = 6 pieces stay in flight)
  fa = k_frag(0, 0, 0); fb = k_frag(0, 1, 0); fc = k_frag(0, 0, 1);
  {
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel_hi:[0,0,0]" : "=&v"(sa) : "v"(fa), "v"(qf[0]), "v"(bs), "v"(qk_scale), "v"(one) : "memory");
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel_hi:[0,0,0]" : "=&v"(sb[0]) : "v"(fb), "v"(qf[0]), "v"(bs), "v"(qk_scale), "v"(one) : "memory");
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(sa) : "v"(fc), "v"(qf[1]), "v"(qk_scale), "v"(one) : "memory");
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(fa), "+v"(fb), "+v"(fc));
    fa = k_frag(0, 1, 1);
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(sb[0]) : "v"(fa), "v"(qf[1]), "v"(qk_scale), "v"(one) : "memory");
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(fa));
  }
  fa = zero8; fb = zero8; fc = zero8;                      // "V8T(-1)" channel blocks 0 .. 2
  X_PIN(fa); X_PIN(fb); X_PIN(fc);

  auto qk_first = [&](f32x16_t& d, const i32x8_t& k) __attribute__((always_inline)) {
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel_hi:[0,0,0]" : "=&v"(d) : "v"(k), "v"(qf[0]), "v"(bs), "v"(qk_scale), "v"(one) : "memory");
  };
  auto qk_acc = [&](f32x16_t& d, const i32x8_t& k) __attribute__((always_inline)) {
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(d) : "v"(k), "v"(qf[1]), "v"(qk_scale), "v"(one) : "memory");
  };
  auto o_settle = [&]() __attribute__((always_inline)) {
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(lacc));
  };
  using Half = PHalf<ABL, FAST>;
  constexpr int HN = Half::N;

  auto iteration = [&](const int t, f32x16_t& sbc, f32x16_t& sbn, i32x8_t& pc, i32x8_t& pn) __attribute__((always_inline)) {
    stamp(t, 0);
    if (!(ABL & 4)) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");      // tile t+1 has landed everywhere (tiles t+2, t+3 in flight)
    stamp(t, 1);
    const bool first = MODE != 2 && t == 0;
    const bool last_of_chunk = tic == p.tiles_per_chunk - 1;
    tic = last_of_chunk ? 0 : tic + 1;
    XRowMax rm;
    rm.mx = 0.f;
#define STEPS_A(G) do { if (!(ABL & 8)) Half::template run<HN * (G) / 4, HN * ((G) + 1) / 4>(sa, pn, 0); } while (0)      /* share G of 4 of the kb = 0 half */
#define STEPS_B(G) do { if (!(ABL & 8)) Half::template run<HN * (G) / 3, HN * ((G) + 1) / 3>(sbc, pn, 1); } while (0)     /* share G of 3 of the kb = 1 half */
    if (last_of_chunk && tail_valid < KT) {                  // rare, wave-uniform: keys past the chunk's end
      int tv = tail_valid - 4 * hi;
      asm volatile("" : "+v"(tv));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if ((r & 3) + 8 * (r >> 2) >= tv) sa[r] = -INFINITY;
        if (32 + (r & 3) + 8 * (r >> 2) >= tv) sbc[r] = -INFINITY;
      }
    }
    bool flag = false;
    float alpha = 1.f;
    if constexpr (FAST != 0) {
      // fp8_fast: the conversions are SPECULATIVE - a saturating v_cvt_pk_u8_f32 cannot trap, and nothing reads P(t) before the next
      // tile - so the row max is no longer in front of them: its dependent chain (20 steps, the half-lane swap, the ballot) runs
      // BESIDE the conversions of the keys-0-31 half and is only checked before the MFMA that overwrites those scores (j6).  The
      // rare path (a row max above the deferred threshold, or the first tile) shifts the scores and the splat, repeats the 16
      // conversions of the first half, and shifts the keys-32-63 scores of tile t+1 that j5 has already started on the old splat.
      X_FENCE();
      p_pv(o[0], fa, pc, one);                      // j0
      if (!(ABL & 2)) stage_k();
      fa = v_frag(t + 7, 3);
      X_FENCE();
      p_pv(o[1], fb, pc, one);                      // j1
      if (!(ABL & 2)) stage_v();
      fb = ones_frag();
      X_FENCE();
      p_pv(o[2], fc, pc, one);                      // j2
      fc = k_frag(t + 1, 1, 0);
#pragma unroll
      for (int n = 0; n < 5; ++n) if (!(ABL & 16)) rm.step(n, sa, sbc, hi);
      STEPS_A(0);
      X_FENCE();
      p_pv(o[3], fa, pc, one);                      // j3
      fa = k_frag(t + 1, 0, 0);
#pragma unroll
      for (int n = 5; n < 10; ++n) if (!(ABL & 16)) rm.step(n, sa, sbc, hi);
      STEPS_A(1);
      X_FENCE();
      p_pv(lacc, fb, pc, one);                      // j4
      fb = k_frag(t + 1, 1, 1);
#pragma unroll
      for (int n = 10; n < 15; ++n) if (!(ABL & 16)) rm.step(n, sa, sbc, hi);
      STEPS_A(2);
      X_FENCE();
      qk_first(sbn, fc);                            // j5
      fc = k_frag(t + 1, 0, 1);
#pragma unroll
      for (int n = 15; n < 20; ++n) if (!(ABL & 16)) rm.step(n, sa, sbc, hi);
      STEPS_A(3);
      X_FENCE();
      stamp(t, 2);
      if (first || __builtin_amdgcn_ballot_w64(rm.mx > THR) != 0) {     // rare: re-base, then redo the first half
        const float du = first ? rm.mx - TOP : fmaxf(rm.mx - TOP, 0.f);
        const float delta = du * (1.f / SC);
        alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
        m_run += delta;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] -= du; sbc[r] -= du; bs[r] = SC * (P_SHIFT - m_run) + OFF; }
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(sbn));       // j5 has landed
#pragma unroll
        for (int r = 0; r < 16; ++r) sbn[r] -= du;
        X_PIN(sbn);
        flag = true;
        Half::template run<0, HN>(sa, pn, 0);
      }
    } else {
    // The four P.V MFMAs need nothing of this tile's softmax: they are issued first, so that they EXECUTE under the two LDS-DMA pieces
    // (~150 cycles of issue stall each) and under the dependent row-max chain - the wave's own stalls are covered by its own MFMAs.
    X_FENCE();
    p_pv(o[0], fa, pc, one);                      // j0
    if (!(ABL & 2)) stage_k();
    fa = v_frag(t + 7, 3);                        // for j3: V8T(t-1) channel block 3  ((t - 1) & 7 == (t + 7) & 7)
    X_FENCE();
    p_pv(o[1], fb, pc, one);                      // j1
    if (!(ABL & 2)) stage_v();
    fb = ones_frag();                             // for j4
    X_FENCE();
    p_pv(o[2], fc, pc, one);                      // j2
    fc = k_frag(t + 1, 1, 0);                     // for j5
#pragma unroll
    for (int n = 0; n < 10; ++n) if (!(ABL & 16)) rm.step(n, sa, sbc, hi);
    X_FENCE();
    p_pv(o[3], fa, pc, one);                      // j3
    fa = k_frag(t + 1, 0, 0);                     // for j6
#pragma unroll
    for (int n = 10; n < 20; ++n) if (!(ABL & 16)) rm.step(n, sa, sbc, hi);
    X_FENCE();
    stamp(t, 2);
    if (first || __builtin_amdgcn_ballot_w64(rm.mx > THR) != 0) {     // rare: re-base
      const float du = first ? rm.mx - TOP : fmaxf(rm.mx - TOP, 0.f);
      const float delta = du * (1.f / SC);
      alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
      m_run += delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sa[r] -= du; sbc[r] -= du; bs[r] = SC * (P_SHIFT - m_run) + OFF; }
      flag = true;                                   // O and l are scaled at the end of the tile: their MFMAs of tile t-1 are in flight
    }
    X_FENCE();
    p_pv(lacc, fb, pc, one);                      // j4: l += ones x P^T(t-1)
    fb = k_frag(t + 1, 1, 1);                     // for j7
    STEPS_A(0);
    X_FENCE();
    qk_first(sbn, fc);                            // j5: keys 32-63 of tile t+1
    fc = k_frag(t + 1, 0, 1);                     // for j8
    STEPS_A(1);
    STEPS_A(2);
    STEPS_A(3);                                   // the kb = 0 half of S(t) is consumed
    }
    X_FENCE();
    qk_first(sa, fa);                             // j6: keys 0-31 of tile t+1, into the registers S(t) has just left
    fa = v_frag(t, 0);                            // for the next tile's j0
    STEPS_B(0);
    X_FENCE();
    qk_acc(sbn, fb);                              // j7
    fb = v_frag(t, 1);
    STEPS_B(1);
    X_FENCE();
    qk_acc(sa, fc);                               // j8
    fc = v_frag(t, 2);
    STEPS_B(2);
    X_FENCE();
    stamp(t, 3);
    if (flag) {
      o_settle();
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[cb][r] *= alpha;
        asm volatile("" : "+v"(o[cb]));
        X_FENCE();
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) lacc[r] *= alpha;
      asm volatile("" : "+v"(lacc));
      X_FENCE();
    }
    if (!(ABL & 2)) stage_advance();
#undef STEPS_A
#undef STEPS_B
  };

  int t = 0;
  in_loop = true;
  for (; t + 1 < all_tiles; t += 2) {
    iteration(t, sb[0], sb[1], pf[0], pf[1]);
    iteration(t + 1, sb[1], sb[0], pf[1], pf[0]);
  }
  if (t < all_tiles) iteration(t, sb[0], sb[1], pf[0], pf[1]);
  in_loop = false;
  const bool odd = (all_tiles & 1) != 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(sa), "+v"(sb[0]), "+v"(sb[1]));
  {
    i32x8_t plast;
#pragma unroll
    for (int e = 0; e < 8; ++e) plast[e] = odd ? pf[1][e] : pf[0][e];       // by value: a select between two ADDRESSES keeps pf in memory
    p_pv(o[0], fa, plast, one);
    p_pv(o[1], fb, plast, one);
    p_pv(o[2], fc, plast, one);
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(fa), "+v"(fb));
    fa = v_frag(all_tiles - 1, 3);
    fb = ones_frag();
    p_pv(o[3], fa, plast, one);
    p_pv(lacc, fb, plast, one);
  }
  o_settle();

  const float l = lacc[0];
  if (MODE == 1) {
    float* sp = p.state + ((int64_t)sh * p.sq_pad + qrow) * F8_STATE_LD;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4_t*>(sp + cb * 32 + 8 * g + 4 * hi) = f32x4_t{o[cb][4 * g], o[cb][4 * g + 1], o[cb][4 * g + 2], o[cb][4 * g + 3]};
    if (hi == 0) { sp[HD8] = m_run; sp[HD8 + 1] = l; }
    return;
  }
  if (qrow < p.sq) {
    const float inv = 1.f / l;
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
  AM_CHECK(abl == 0 || abl == 200 || abl == 100 || abl == 400 || (a->rows == 0 && a->state_mode == 0), "am_attention_fp8: ablation codes run the one-pass form only");
#define F8_ATTR(...) AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fp8_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE_BYTES))
#define X64_ATTR(...) AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fp8x64_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE_BYTES))
  AM_ONCE_PER_DEVICE({ X64_ATTR(0, 0); X64_ATTR(1, 0); X64_ATTR(2, 0); X64_ATTR(8, 0); X64_ATTR(0, 1); X64_ATTR(0, 2); });
#undef X64_ATTR
#define P_ATTR(...) AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fp8p_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES))
#ifdef AM_ATTN_ABLATIONS
  AM_ONCE_PER_DEVICE({ P_ATTR(12, 0, 0); P_ATTR(24, 0, 0); P_ATTR(40, 0, 0); P_ATTR(10, 0, 0); P_ATTR(62, 0, 0); P_ATTR(4, 0, 0); P_ATTR(16, 0, 0); P_ATTR(32, 0, 0); });
#endif
  AM_ONCE_PER_DEVICE({ P_ATTR(0, 0, 0); P_ATTR(1, 0, 0); P_ATTR(2, 0, 0); P_ATTR(8, 0, 0); P_ATTR(0, 1, 0); P_ATTR(0, 2, 0); P_ATTR(0, 0, 1); P_ATTR(0, 1, 1); P_ATTR(0, 2, 1); });
#undef P_ATTR
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
#define X64_LAUNCH(A, M, GRID) hipLaunchKernelGGL((attn_fp8x64_kernel<A, M>), GRID, dim3(256), NSTAGE * STAGE_BYTES, st, p, (unsigned long long*)nullptr)
  // Product kernel of the main grid: the 4 x 64 form (round 4) for key streams of at least 8 tiles; defer_log2 = 5100 + ABL selects its
  // timing ablations, 5200 forces the 8-wave kernel (same-box A/B, tests), 5000 + ABL are the 8-wave kernel's ablations as before.
  const bool want8 = a->defer_log2 == 5200 || (abl != 0 && abl < 100);
  const bool want64 = abl >= 100 && abl < 200;              // 5100 + ABL: the 4 x 64 form (measured 12 % slower: kept for the A/B)
  const int xabl = want64 ? abl - 100 : 0;
  const int pabl = abl >= 300 && abl < 400 ? abl - 300 : 0; // 5300 + ABL: timing ablations of the product (free-running) kernel
  const bool fast = abl == 400;                             // 5400: the exponent-field form of the probabilities (attn_dtype "fp8_fast")
  const bool long_stream = all_tiles >= 8 && getenv("ACTIONMESH_AMD_FP8_8WAVE") == nullptr;
  const bool use_x64 = want64 && long_stream;
  const bool use_p = !want8 && !want64 && long_stream;
#define P_LAUNCH(A, M, F, GRID) hipLaunchKernelGGL((attn_fp8p_kernel<A, M, F>), GRID, dim3(512), P_LDS_BYTES, st, p, (unsigned long long*)nullptr)
  if (a->rows != 2 && use_p) {
    const dim3 grid(nblk_main, bh);
    if (fast) {
      if (a->state_mode == 1) P_LAUNCH(0, 1, 1, grid);
      else if (a->state_mode == 2) P_LAUNCH(0, 2, 1, grid);
      else P_LAUNCH(0, 0, 1, grid);
    } else if (a->state_mode == 1) P_LAUNCH(0, 1, 0, grid);
    else if (a->state_mode == 2) P_LAUNCH(0, 2, 0, grid);
    else switch (pabl) {
      case 0: P_LAUNCH(0, 0, 0, grid); break;
      case 1: P_LAUNCH(1, 0, 0, grid); break;
      case 2: P_LAUNCH(2, 0, 0, grid); break;
      case 8: P_LAUNCH(8, 0, 0, grid); break;
#ifdef AM_ATTN_ABLATIONS
      case 12: P_LAUNCH(12, 0, 0, grid); break;
      case 24: P_LAUNCH(24, 0, 0, grid); break;
      case 40: P_LAUNCH(40, 0, 0, grid); break;
      case 10: P_LAUNCH(10, 0, 0, grid); break;
      case 62: P_LAUNCH(62, 0, 0, grid); break;
      case 4: P_LAUNCH(4, 0, 0, grid); break;
      case 16: P_LAUNCH(16, 0, 0, grid); break;
      case 32: P_LAUNCH(32, 0, 0, grid); break;
#endif
      default: AM_FAIL(AM_ERR_INVALID, "am_attention_fp8: unknown ablation code %d", a->defer_log2);
    }
  } else
#undef P_LAUNCH
  if (a->rows != 2 && use_x64) {
    const dim3 grid(nblk_main, bh);
    if (a->state_mode == 1) X64_LAUNCH(0, 1, grid);
    else if (a->state_mode == 2) X64_LAUNCH(0, 2, grid);
    else switch (xabl) {
      case 0: X64_LAUNCH(0, 0, grid); break;
      case 1: X64_LAUNCH(1, 0, grid); break;
      case 2: X64_LAUNCH(2, 0, grid); break;
      case 8: X64_LAUNCH(8, 0, grid); break;
      default: AM_FAIL(AM_ERR_INVALID, "am_attention_fp8: unknown 4x64 ablation code %d", a->defer_log2);
    }
  } else if (a->rows != 2) {                       // the main grid
    const dim3 grid(nblk_main, bh);
    if (a->state_mode == 1) F8_LAUNCH(0, 1, grid);
    else if (a->state_mode == 2) F8_LAUNCH(0, 2, grid);
    else switch (abl == 200 || abl == 400 ? 0 : abl) {   // fp8_fast (400) on a stream too short for the product kernel: the exact-exp2 form
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
#undef X64_LAUNCH
  AM_HIP(hipGetLastError());
  return AM_OK;
}

#ifdef AM_ATTN_ABLATIONS
extern "C" int am_attention_fp8p_profile(const am_attn_args* a, const uint8_t* q8, const uint8_t* k8, const uint8_t* vt8,
                                         unsigned long long* prof_dev, void* stream) {
  AM_TRY(check_args(a, "am_attention_fp8p_profile"));
  AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fp8p_kernel<0, 0, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             P_LDS_BYTES));
  f8_args p;
  p.Q = q8; p.K = k8; p.Vt = vt8; p.O = a->O;
  p.heads = a->heads; p.sq = a->sq; p.sq_pad = a->sq_pad; p.sk = a->sk; p.sk_pad = a->sk_pad;
  p.nchunks = a->nchunks; p.tiles_per_chunk = (a->sk + KT - 1) / KT; p.ldo = a->ldo;
  p.chunk_stride = 0; p.chunk_first = 0; p.chunk_total = 0; p.qblk_base = 0; p.state = nullptr; p.part = nullptr;
  hipLaunchKernelGGL((attn_fp8p_kernel<0, 0, 0, true>), dim3((a->sq + 255) / 256, a->nseq * a->heads), dim3(512), P_LDS_BYTES,
                     (hipStream_t)stream, p, prof_dev);
  AM_HIP(hipGetLastError());
  return AM_OK;
}
// per-phase s_memtime stamps of workgroup (0,0) of the 4 x 64 kernel: prof[4 waves][8 tiles (64..71)][8 slots]  (tools/attn_profile.py --fp8x64)
extern "C" int am_attention_fp8x64_profile(const am_attn_args* a, const uint8_t* q8, const uint8_t* k8, const uint8_t* vt8,
                                           unsigned long long* prof_dev, void* stream) {
  AM_TRY(check_args(a, "am_attention_fp8x64_profile"));
  AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fp8x64_kernel<0, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             NSTAGE * STAGE_BYTES));
  f8_args p;
  p.Q = q8; p.K = k8; p.Vt = vt8; p.O = a->O;
  p.heads = a->heads; p.sq = a->sq; p.sq_pad = a->sq_pad; p.sk = a->sk; p.sk_pad = a->sk_pad;
  p.nchunks = a->nchunks; p.tiles_per_chunk = (a->sk + KT - 1) / KT; p.ldo = a->ldo;
  p.chunk_stride = 0; p.chunk_first = 0; p.chunk_total = 0; p.qblk_base = 0; p.state = nullptr; p.part = nullptr;
  hipLaunchKernelGGL((attn_fp8x64_kernel<0, 0, true>), dim3((a->sq + 255) / 256, a->nseq * a->heads), dim3(256), NSTAGE * STAGE_BYTES,
                     (hipStream_t)stream, p, prof_dev);
  AM_HIP(hipGetLastError());
  return AM_OK;
}
#endif
