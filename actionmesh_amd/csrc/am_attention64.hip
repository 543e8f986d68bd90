// Flash attention forward, "4 x 64" structure: one workgroup = 4 waves = ONE wave per SIMD, each wave owns
// 64 query rows (two 32-row blocks j = 0, 1) and the whole 512-register file.  Same maths, operand layouts
// and LDS images as am_attention.hip (read its header first); what changes is the work per wave:
//   * every K / V^T fragment read from LDS feeds TWO MFMAs (one per query block): half the LDS reads,
//     DMA instructions and barriers per MFMA of the 8-wave kernel;
//   * the softmax of tile g runs in the issue gaps of the MFMAs of its neighbours, inside one wave, so
//     the overlap does not depend on how the SIMD arbitrates between two waves:
//         phase 1(g):  O += V^T(g-1) P^T(g-1)   (32 MFMA)  ||  softmax of block 0 of tile g
//         phase 2(g):  S(g+1) = K(g+1) Q^T      (32 MFMA)  ||  softmax of block 1 of tile g
//   * -m_run is folded into the MFMA: the score accumulators start from a 16-register splat of -m_run per block
//     (rewritten only on a re-base), so scores are born as S - m_run and the per-element subtraction disappears;
//   * two forms of the online softmax (template parameter LAZY):
//       exact  each phase = [8 MFMA || row max] -> rare re-base branch (deferred: threshold 2^8) -> [24 MFMA || exp,
//              row sum, bf16 pack].  Block 1's accumulators are initialised while its previous softmax may still
//              move m_run; that offset is carried as a pending correction (`pend1`), applied in the tile after;
//       lazy   (product) no row max: m_run starts at 0 and moves up by whole octaves when a tile's row sums say it
//              has fallen behind (> 2^12) - O, l and the pending P are multiplied by an exact power of two; each
//              phase = 32 MFMA || the whole exp / row sum / pack of one block.  What it cannot represent marks the
//              workgroup, and the exact kernel launched behind it over the same grid redoes the marked ones
//              (comment at LAZY_T below);
//   * K/V^T tiles (64 keys) live in a 4-deep LDS ring (128 KiB): in iteration g the MFMAs read V^T(g-1) and
//     K(g+1); K is fetched three tiles ahead and V^T two (LDS-DMA pieces issued one per MFMA pair inside the two
//     light phases, source = scalar base + 32-bit lane offset), one barrier per tile with a counted
//     s_waitcnt vmcnt(8); fragments sit in explicit rotating register sets with HOLD() keep-alives;
//   * the padded keys of a chunk's partial last tile score exactly 0 and their exp2(-m) is removed from the
//     row sums once, after the loop (rescales multiply it like every other term);
//   * O (128) and the pre-scaled Q (64) live in AccVGPRs a[64:255], named literally in the inline asm of
//     am_attention64_asm.inc (P.V runs as AGPR-form MFMAs, QK^T takes its B operand from the accumulator file);
//   * two-pass form for the multi-GPU overlap (STATE): save (O, m, l) after the local key chunk, resume over
//     the remote chunks in ring order (am_attn_args.state_mode / chunk_first / chunk_total).
// Compiled with -fno-slp-vectorize (v_pk_*_f32 beside MFMAs costs more than two scalar ops on gfx950) and
// IEEE mode off for this file, so fmaxf chains become v_max3_f32 without canonicalising v_max's; hipcc owns the
// MFMA -> VALU hazards of its own instructions, the asm statements carry theirs (DESIGN.md 4.1 lists the three
// that bit: SrcC write-after-read, in-flight results at loop exit, operand registers re-used too early).
#include "am_common.h"

namespace {

constexpr int KVBLK = 64;
constexpr int HD = 128;
constexpr int SUB_B = KVBLK * HD * 2;       // one K or V^T tile: 16 KiB
constexpr int STAGE_B = 2 * SUB_B;          // [K tile][V^T tile]
constexpr int NSTAGE = 4;
constexpr int QBLK = 256;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

#include "am_attention64_asm.inc"

__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
#define FENCE() __builtin_amdgcn_sched_barrier(0)
// A sched_barrier region is [MFMA asm, fragment read / DMA piece, softmax steps]; nothing orders the un-pinned VALU steps
// behind the region's MFMA, and hipcc hoisted them above it in the QK^T phase (tools/attn64_gaps.py: every other gap empty,
// the next one twice as full).  POST_PV() / POST_QK() close the region right behind the MFMA so the steps stay in their gap.
// AM_A64_POSTFENCE: bit 0 = QK^T phase, bit 1 = P.V phase.
#ifndef AM_A64_POSTFENCE
#define AM_A64_POSTFENCE 3
#endif
#ifndef AM_A64_SADDR
#define AM_A64_SADDR 1
#endif
#ifndef AM_A64_SNAKE
#define AM_A64_SNAKE 0       // operand-order probe of the lazy iteration (see phase 1); bit-identical either way
#endif
#ifndef AM_A64_PREP
#define AM_A64_PREP 1
#endif
// the counted wait in front of the per-tile barrier (8 = everything but the previous iteration's pieces); 0 = full drain (A/B builds)
#ifndef AM_A64_VMCNT
#define AM_A64_VMCNT 8
#endif
#define AM_STR2(x) #x
#define AM_STR(x) AM_STR2(x)
#define A64_TILE_BARRIER() asm volatile("s_waitcnt vmcnt(" AM_STR(AM_A64_VMCNT) ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define POST_QK() do { if (AM_A64_POSTFENCE & 1) FENCE(); } while (0)
#define POST_PV() do { if (AM_A64_POSTFENCE & 2) FENCE(); } while (0)
// Pins a value to this point of the instruction stream: the (empty) volatile asm is ordered with the MFMA asm
// statements, so the producer of x cannot be sunk or hoisted to another phase by the optimiser.
#define PIN(x) asm volatile("" : "+v"(x))
// Keeps MFMA source registers allocated up to this point.  hipcc treats an MFMA asm statement like any other
// instruction and hands its A/B registers to the next ds_read / VALU result as soon as the statement has issued;
// the hardware then holds that writer until the (queued) MFMA has fetched its operands - measured at ~28 cycles
// per LDS fragment read, 7 ms of a 29 ms kernel.
#define HOLD2(a, b) asm volatile("" :: "v"(a), "v"(b))
#define HOLD4(a, b, c, d) asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d))

// ---- softmax of one 32-query block, cut into single-instruction steps that the phases thread through their
//      MFMA gaps.  sa/sb = the block's scores for keys 0-31 / 32-63 of the tile (already relative to m_run). ----
// The 20 row-max steps go into the first 7 of a phase-a's 8 MFMA gaps: the dependent tail (swap, final max) must
// be out of the way when the last MFMA issues, so that only the compare + branch separate it from phase b's first.
__device__ __host__ constexpr int rm_lo(int gap) { return gap >= 7 ? 20 : (20 * gap + 6) / 7; }
struct RowMax {            // 20 steps: four v_max3 chains, cross-half swap
  float a[4];
  float mx;
  __device__ __forceinline__ void step(int n, const f32x16_t& sa, const f32x16_t& sb) {
    if (n < 4) { a[n] = max3(sa[n], sb[n], sa[n + 4]); PIN(a[n]); }
    else if (n < 8) { a[n - 4] = max3(a[n - 4], sb[n], sa[n + 4]); PIN(a[n - 4]); }
    else if (n < 12) { a[n - 8] = max3(a[n - 8], sb[n], sa[n + 4]); PIN(a[n - 8]); }
    else if (n < 16) { a[n - 12] = __builtin_fmaxf(a[n - 12], sb[n]); PIN(a[n - 12]); }
    else if (n == 16) { a[2] = __builtin_fmaxf(a[2], a[3]); PIN(a[2]); }
    else if (n == 17) { mx = max3(a[0], a[1], a[2]); PIN(mx); }
    else if (n == 18) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      a[0] = __uint_as_float(sw[0]);
      a[1] = __uint_as_float(sw[1]);
      PIN(a[0]);
    } else { mx = __builtin_fmaxf(a[0], a[1]); PIN(mx); }
  }
};
// ---- exp / row sum / bf16 pack of one 32-query block as a sequence of 80 single-instruction steps: 32 v_exp_f32,
// 32 v_add_f32 (row sum of the unrounded probabilities), 16 v_cvt_pk_bf16_f32; each op trails the exponentials it
// depends on by a round.  (Tried: v_dot2c_f32_bf16 on the packed pairs instead of the adds - 64 steps - no faster.)
// AM_A64_DOTSUM (round-6 probe of the lazy kernel, OFF: product ISA unchanged): the row sum over the bf16-ROUNDED probabilities - the
// values the P.V MFMAs multiply - as ONE v_dot2c_f32_bf16 per packed pair (acc += p.lo * 1 + p.hi * 1) instead of two v_add_f32 on the
// unrounded exponentials: 64 softmax steps per block instead of 80.  Measured on one box, three interleaved rounds
// (profiles/r06m_dotsum_ab.txt): 26.92 ms against 25.60 (+5 %), 37.1 J per launch against 34.7, matrix pipe 0.63 busy against 0.75 - the
// dot occupies the VALU port longer than the two adds it replaces, as v_pk_add_f32 did in round 5.  Fewer instructions is not fewer cycles.
#ifndef AM_A64_DOTSUM
#define AM_A64_DOTSUM 0
#endif
enum EsKind : int { ES_EX = 0, ES_AD = 1, ES_PK = 2, ES_DT = 3 };
struct EsSeq { int n; int cost; int kind[80]; int arg[80]; };
__device__ __host__ constexpr EsSeq es_make_seq(bool dot = false) {
  EsSeq q{};
  int n = 0;
  auto put = [&](int k, int a) { q.kind[n] = k; q.arg[n] = a; ++n; };
  for (int r = 0; r < 16; ++r) {
    put(ES_EX, 2 * r);
    put(ES_EX, 2 * r + 1);
    if (dot) {                       // the pack trails its exponentials by a round, the dot its pack by another
      if (r >= 1) put(ES_PK, r - 1);
      if (r >= 2) put(ES_DT, r - 2);
    } else if (r >= 1) {
      put(ES_AD, 2 * r - 2);
      put(ES_AD, 2 * r - 1);
      put(ES_PK, r - 1);
    }
  }
  if (dot) {
    put(ES_PK, 15);
    put(ES_DT, 14);
    put(ES_DT, 15);
  } else {
    put(ES_AD, 30);
    put(ES_AD, 31);
    put(ES_PK, 15);
  }
  q.n = n;
  // Issue-slot cost model for spreading the steps over a phase's MFMA gaps: a wave issues one instruction per 4 cycles
  // and v_exp_f32 holds the port for two slots (tools/ubench/mfma_fillers.hip), so a step costs 2 (exp) or 1.
  for (int i = 0; i < n; ++i) q.cost += q.kind[i] == ES_EX ? 2 : 1;
  return q;
}
__device__ __host__ constexpr int es_cost_before(const EsSeq& q, int n) { int c = 0; for (int i = 0; i < n; ++i) c += q.kind[i] == ES_EX ? 2 : 1; return c; }
// exact kernel: 24 gaps (phase b), gap i takes the steps whose cumulative cost starts in [cost i / 24, cost (i+1) / 24)
struct EsTab { int lo[25]; };        // steps [lo[i], lo[i+1]) go into gap i
__device__ __host__ constexpr EsTab es_make_tab(const EsSeq& q) {
  EsTab t{};
  for (int i = 0; i <= 24; ++i) {
    int n = 0;
    while (n < q.n && es_cost_before(q, n) * 24 / q.cost < i) ++n;
    t.lo[i] = n;
  }
  return t;
}
// LAZY kernels have no row-max pass in front of the exponentials, so a phase spreads its steps over all 32 MFMA gaps.
// Gap capacities (issue slots left beside what else the gap holds): a gap with an LDS-DMA piece / a fragment read gets
// fewer steps.  pv = true: the P.V phase (pieces in the even gaps 0..6, reads in gaps 0..3 of every k-step);
// false: the QK^T phase (pieces in even gaps 0..6, a read in every odd gap).
#ifndef AM_ES_CAP_FULL
#define AM_ES_CAP_FULL 8
#endif
#ifndef AM_ES_CAP_DMA
#define AM_ES_CAP_DMA 0
#endif
#ifndef AM_ES_CAP_READ
#define AM_ES_CAP_READ 6
#endif
struct EsTab32 { int lo[33]; };
__device__ __host__ constexpr int es_cap32(bool pv, int gap) {
  const bool dma = gap < 8 && (gap & 1) == 0;
  const bool rd = pv ? (gap & 7) < 4 : (gap & 1) == 1;
  return dma ? AM_ES_CAP_DMA : rd ? AM_ES_CAP_READ : AM_ES_CAP_FULL;
}
__device__ __host__ constexpr EsTab32 es_make_tab32(const EsSeq& q, bool pv) {
  int cum[33] = {};
  for (int i = 0; i < 32; ++i) cum[i + 1] = cum[i] + es_cap32(pv, i);
  EsTab32 t{};
  int n = 0;
  for (int i = 0; i <= 32; ++i) {        // steps whose cumulative cost starts below cum[i] / cum[32] of the total go before gap i
    while (n < q.n && es_cost_before(q, n) * cum[32] < cum[i] * q.cost) ++n;
    t.lo[i] = n;
  }
  t.lo[32] = q.n;
  return t;
}

template <int ABL, bool DOT = false>
struct ExpSumPackT {
  static constexpr EsSeq SEQ = es_make_seq(DOT);
  float rs[4];
  __device__ __forceinline__ void init() { rs[0] = rs[1] = rs[2] = rs[3] = 0.f; }
  __device__ __forceinline__ static float get(const f32x16_t& sa, const f32x16_t& sb, int e) { return e < 16 ? sa[e] : sb[e - 16]; }
  __device__ __forceinline__ void ex(f32x16_t& sa, f32x16_t& sb, int e) {
    float v = (ABL & 1) ? get(sa, sb, e) * 0.5f : __builtin_amdgcn_exp2f(get(sa, sb, e));
    PIN(v);
    if (e < 16) sa[e] = v;
    else sb[e - 16] = v;
  }
  __device__ __forceinline__ void ad(f32x16_t& sa, f32x16_t& sb, int e) { rs[e & 3] += get(sa, sb, e); PIN(rs[e & 3]); }
  __device__ __forceinline__ void pk(f32x16_t& sa, f32x16_t& sb, int pr, u32x4_t (&w)[4]) {
    uint32_t v = pack_bf2(get(sa, sb, 2 * pr), get(sa, sb, 2 * pr + 1));
    PIN(v);
    w[pr >> 2][pr & 3] = v;
  }
  // rs[pr & 3] += lo + hi of packed pair pr.  Inline asm (this hipcc cannot select the builtin), so its hazards are ours: a dot's result may
  // feed the SrcC of the next dot of the same opcode back to back, but any OTHER VALU reading it needs 3 wait states - total() below.
  __device__ __forceinline__ void dt(int pr, u32x4_t (&w)[4]) {
#ifdef AM_F16
    asm volatile("v_dot2c_f32_f16 %0, 0x3c003c00, %1" : "+v"(rs[pr & 3]) : "v"(w[pr >> 2][pr & 3]));
#else
    asm volatile("v_dot2c_f32_bf16 %0, 0x3f803f80, %1" : "+v"(rs[pr & 3]) : "v"(w[pr >> 2][pr & 3]));
#endif
  }
  __device__ __forceinline__ void step(int n, f32x16_t& sa, f32x16_t& sb, u32x4_t (&w)[4]) {
    const int k = SEQ.kind[n], a = SEQ.arg[n];
    if (k == ES_EX) ex(sa, sb, a);
    else if (k == ES_AD) ad(sa, sb, a);
    else if (k == ES_DT) dt(a, w);
    else pk(sa, sb, a, w);
  }
  __device__ __forceinline__ float total() {
    if (DOT) asm volatile("s_nop 2" : "+v"(rs[0]), "+v"(rs[1]), "+v"(rs[2]), "+v"(rs[3]));
    return (rs[0] + rs[1]) + (rs[2] + rs[3]);
  }
};

// ABL (timing ablations, numerically meaningless, AM_ATTN_ABLATIONS builds only): 1 = no exp, 2 = no row max,
// 4 = no barrier / DMA drain, 8 = no exp/sum/pack at all, 16 = no LDS fragment reads (stale registers)
// STATE (two-pass attention, am_attn_args.state_mode): 0 = one pass; 1 = save the un-normalised (O, m, l) of every row
// to p.state instead of writing O; 2 = resume from p.state, finish, write O.
constexpr int STATE_LD = 132;      // floats per saved row: O[128], m, l, pad (16-byte aligned rows)
// LAZY: no row max in the loop.  m_run starts at tile 0's row max rounded up to a whole octave; the scores of a tile are exponentiated relative to the
// m_run they were born with; the tile's row sums (computed anyway) tell afterwards whether m_run has fallen behind
// (sum > 2^12), and the rare re-base multiplies O, l and the not yet consumed P by an exact power of two and moves m_run up
// by that many octaves.  fp32 and bf16 share an 8-bit exponent, so a lag of up to 2^60 loses nothing.  What this cannot
// represent - a single-tile jump beyond 2^60, a non-finite sum, a row whose scores all sit more than ~100 octaves below
// m_run (final sum under 2^-100) - marks the workgroup in `flags`, and the exact kernel (LAZY = false, same grid, launched
// behind it) recomputes the marked workgroups and clears the marks.
constexpr float LAZY_T = 4096.f;
__device__ __forceinline__ bool tid_is_zero() { return __builtin_amdgcn_workitem_id_x() == 0; }
template <int DEFER, int ABL = 0, bool PROF = false, int STATE = 0, bool LAZY = false>
__global__ __launch_bounds__(256, 1) void attn_fwd64_kernel(am_attn_args p, int tiles_per_chunk, unsigned long long* prof,
                                                            unsigned* flags, int flag_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int flag_idx = __builtin_amdgcn_workgroup_id_y() * flag_stride + __builtin_amdgcn_workgroup_id_x();
  if (!LAZY && flags != nullptr) {                                    // fallback launch: only the marked workgroups run
    if (flags[flag_idx] == 0) return;
    if (tid_is_zero()) atomicAdd(flags - 4, 1u);                      // diagnostic: am_attention_fallback_count
  }
  const int tid = __builtin_amdgcn_workitem_id_x();
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int bh = __builtin_amdgcn_workgroup_id_y();
  const int head = bh % p.heads, seq = bh / p.heads;
  const int q0 = __builtin_amdgcn_workgroup_id_x() * QBLK + wave * 64;
  const float c = p.scale * 1.4426950408889634f;
  // PROF: s_memtime stamps of workgroup (0,0), tiles 64..71: prof[wave][tile - 64][slot]
  auto stamp = [&](int g, int slot) __attribute__((always_inline)) {
    if (PROF && __builtin_amdgcn_workgroup_id_x() == 0 && bh == 0 && g >= 64 && g < 72) {
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
      if (lane == 0) prof[(wave * 8 + (g - 64)) * 8 + slot] = t;
    }
  };

  // ---- Q fragments (B operand) of both query blocks, pre-scaled to log2 units, parked in a[192:255] ----
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const bf16_t* qp = p.Q + ((int64_t)bh * p.sq_pad + q0 + 32 * j + l31) * HD + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(qp + ks * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) q_write((j * 8 + ks) * 4 + e, pack_bf2(bflo(raw[e]) * c, bfhi(raw[e]) * c));
    }
  }
  if (STATE != 2) o_zero();

  // ---- LDS-DMA: 16-byte unit U = i*256 + wave*64 + lane of a 1024-unit tile operand, i = 0..3 ----
  // Source address = wave-uniform base (scalar registers: operand base + tile + piece) + a 32-bit per-lane byte
  // offset, so hipcc selects the saddr form of global_load_lds and a piece costs no VALU address arithmetic.
  const int64_t k_seq_stride = (int64_t)p.sk_pad * HD;
  const int x = wave * 64 + lane;
  const int kr = x >> 4, kc = (x & 15) ^ (kr & 15);
  const unsigned k_lane_off = (unsigned)(kr * HD + kc * 8) * 2u;                       // bytes; + i * 16 rows (uniform)
  const int vr = x >> 3, vc = (x & 7) ^ ((vr >> 1) & 7);
  const unsigned v_lane_off = ((unsigned)vr * (unsigned)p.sk_pad + (unsigned)vc * 8u) * 2u;   // bytes; + i * 32 rows (uniform)
  const char* k_base = reinterpret_cast<const char*>(p.K + (int64_t)bh * k_seq_stride);
  const char* v_base = reinterpret_cast<const char*>(p.Vt + (int64_t)bh * k_seq_stride);
  const int64_t v_step = (int64_t)32 * p.sk_pad * 2;                                          // bytes
  const int n_tiles = p.nchunks * tiles_per_chunk;
  // pieces issued so far / 4 (= ring slot), source tile, tile in chunk, physical chunk, element offset of that chunk
  struct Cursor { int j, t, tt, ci; int64_t off; };
  const int ci0 = p.chunk_total > 0 ? p.chunk_first : 0;      // the chunks walked are (chunk_first + i) % chunk_total
  Cursor kcur{0, 0, 0, ci0, (int64_t)ci0 * p.chunk_stride}, vcur = kcur;
  auto advance = [&](Cursor& cu) __attribute__((always_inline)) {
    ++cu.j;
    if (cu.t + 1 < n_tiles) {
      ++cu.t;
      if (++cu.tt == tiles_per_chunk) {
        cu.tt = 0;
        if (++cu.ci == p.chunk_total) cu.ci = 0;      // chunk_total = 0 (no wrap) is never reached
        cu.off = (int64_t)cu.ci * p.chunk_stride;
      }
    }
  };
  auto uniform = [](const char* q) __attribute__((always_inline)) {      // pin a wave-uniform pointer to scalar registers
    const uint64_t u = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  // One LDS-DMA piece: 64 lanes x 16 bytes from (wave-uniform base + 32-bit lane offset) to LDS at dst + lane * 16.
  // AM_A64_SADDR: the scalar-base form of the instruction, written as asm (hipcc selects the 64-bit vector-address form for the
  // builtin and spends a v_lshl_add_u64 per piece on it); M0 = the wave's LDS base, one wait state before its use.
  const unsigned smem_lds = (unsigned)reinterpret_cast<uintptr_t>((lds_ptr_t)smem);
  auto dma_piece = [&](const char* src, unsigned lane_off, unsigned char* dst) __attribute__((always_inline)) {
#if AM_A64_SADDR
    const unsigned lds = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)(dst - smem));      // integer arithmetic: no null-checked address-space cast per piece
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(src), "s"(lds) : "memory");
#else
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + lane_off), (lds_ptr_t)dst, 16, 0, 0);
#endif
  };
  auto dma_k = [&](int i) __attribute__((always_inline)) {      // piece i of 4 of the next K tile -> ring slot j & 3
    unsigned char* dst = smem + (kcur.j & 3) * STAGE_B + wave * 1024 + i * 4096;
    dma_piece(uniform(k_base + (kcur.off + (int64_t)kcur.tt * (KVBLK * HD) + i * 16 * HD) * 2), k_lane_off, dst);
  };
  auto dma_v = [&](int i) __attribute__((always_inline)) {
    unsigned char* dst = smem + (vcur.j & 3) * STAGE_B + SUB_B + wave * 1024 + i * 4096;
    dma_piece(uniform(v_base + (vcur.off + (int64_t)vcur.tt * KVBLK) * 2 + i * v_step), v_lane_off, dst);
  };

  // AM_A64_PREP (lazy kernel): a piece's scalar address arithmetic runs one gap ahead of its issue, so the DMA gaps - the ones
  // over their issue budget - hold nothing but the M0 write and the instruction itself.
  const char* k_src_n = k_base;
  const char* v_src_n = v_base;
  unsigned k_lds_n = 0, v_lds_n = 0;
  auto k_prep = [&](int i) __attribute__((always_inline)) {
    k_lds_n = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)((kcur.j & 3) * STAGE_B + wave * 1024 + i * 4096));
    k_src_n = uniform(k_base + (kcur.off + (int64_t)kcur.tt * (KVBLK * HD) + i * 16 * HD) * 2);
    asm volatile("" : "+s"(k_lds_n), "+s"(k_src_n));
  };
  auto v_prep = [&](int i) __attribute__((always_inline)) {
    v_lds_n = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)((vcur.j & 3) * STAGE_B + SUB_B + wave * 1024 + i * 4096));
    v_src_n = uniform(v_base + (vcur.off + (int64_t)vcur.tt * KVBLK) * 2 + i * v_step);
    asm volatile("" : "+s"(v_lds_n), "+s"(v_src_n));
  };
  auto k_issue = [&]() __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(k_lane_off), "s"(k_src_n), "s"(k_lds_n) : "memory");
  };
  auto v_issue = [&]() __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(v_lane_off), "s"(v_src_n), "s"(v_lds_n) : "memory");
  };

  // fragment read offsets (bytes) inside a ring stage
  int k_off[8], v_off[4];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_off[ks] = l31 * 256 + (((ks * 2 + hi) ^ (l31 & 15)) << 4);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) v_off[kk] = SUB_B + l31 * 128 + (((kk * 2 + hi) ^ ((l31 >> 1) & 7)) << 4);
  bf16x8_t stale = __builtin_bit_cast(bf16x8_t, u32x4_t{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u});
  PIN(stale);
  auto k_frag = [&](const unsigned char* st, int kb, int ks) __attribute__((always_inline)) {
    if (ABL & 128) return stale;
    if (ABL & 16) st = smem;
    return *reinterpret_cast<const bf16x8_t*>(st + kb * 32 * 256 + k_off[ks]);
  };
  auto v_frag = [&](const unsigned char* st, int d, int kk) __attribute__((always_inline)) {
    if (ABL & 128) return stale;
    if (ABL & 16) st = smem;
    return *reinterpret_cast<const bf16x8_t*>(st + d * 32 * 128 + v_off[kk]);
  };

  // ---- state (arch VGPRs) ----
  f32x16_t s0[2];            // block 0: S(g) on entry of an iteration, S(g+1) on exit (same registers)
  f32x16_t s1[2][2];         // block 1: ping-pong sets, S(g) in s1[cur], S(g+1) born in s1[cur^1]
  u32x4_t p0[2][4];          // block 0 P (bf16 pairs): P(g-1) in p0[cur], P(g) born in p0[cur^1]
  u32x4_t p1[4];             // block 1 P: P(g-1) consumed in phase 1, P(g) born in phase 2 (same registers)
  f32x16_t negm[2];          // -m_run splat: C operand of the first k-step
  bf16x8_t vf[2][4];         // V^T fragments, two sets: k-step kk of P.V reads vf[kk & 1] while vf[~kk & 1] is being loaded
  bf16x8_t kf[3][2];         // K fragments, three sets: k-step ks of QK^T reads kf[ks % 3], loads go two steps ahead
  float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
  float off1 = 0.f;
  bool pend1 = false;
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[0][r] = negm[1][r] = 0.f;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) p0[0][kk] = p0[1][kk] = p1[kk] = u32x4_t{0u, 0u, 0u, 0u};
  // Opaque to the optimiser: these are real, loop-carried registers from here on.  (As known constants, the
  // C operand of a first-k-step MFMA was a dying temporary that hipcc re-used for a ds_read straight after the
  // asm statement - an MFMA reads SrcC over its passes, and hipcc pads SrcC write-after-read only for MFMAs
  // it knows about.)
  if (STATE == 2) {          // resume: O, m_run, l_run of both blocks from the state the first pass saved
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float* sp = p.state + ((int64_t)bh * p.sq_pad + q0 + 32 * j + l31) * STATE_LD;
      m_run[j] = sp[HD];
      l_run[j] = hi == 0 ? sp[HD + 1] : 0.f;       // the two half-lanes' partial sums are added at the end
#pragma unroll
      for (int d = 0; d < 4; ++d) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) o_write4(j * 4 + d, gq, *reinterpret_cast<const f32x4_t*>(sp + d * 32 + 8 * gq + 4 * hi));
        asm volatile("" ::: "memory");         // 16 registers of loads in flight at a time, not 128
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) negm[j][r] = -m_run[j];
    }
  }
  PIN(negm[0]); PIN(negm[1]);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) { PIN(p0[0][kk]); PIN(p0[1][kk]); PIN(p1[kk]); }

  // ---- prologue: tiles 0 and 1 in flight; V^T of ring stage 3 zeroed (iteration 0 multiplies it by P = 0) ----
#pragma unroll
  for (int t = 0; t < 3; ++t) {                 // K(0..2), V^T(0..1)
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_k(i);
    advance(kcur);
    if (t < 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_v(i);
      advance(vcur);
    }
  }
  {
    unsigned char* z = smem + 3 * STAGE_B + SUB_B + tid * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4_t*>(z + i * 16) = u32x4_t{0u, 0u, 0u, 0u};
  }
  dma_drain_barrier();
  auto first_scores = [&](f32x16_t (&s1c)[2]) __attribute__((always_inline)) {   // S(0) = K(0) Q^T
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const bf16x8_t kfr = k_frag(smem, kb, ks);
        if (ks == 0) { s0[kb] = qk_mfma_first(ks, kfr, negm[0]); s1c[kb] = qk_mfma_first(8 + ks, kfr, negm[1]); }
        else { qk_mfma_acc(ks, kfr, s0[kb]); qk_mfma_acc(8 + ks, kfr, s1c[kb]); }
      }
#pragma unroll
    for (int d = 0; d < 4; ++d) vf[0][d] = v_frag(smem + 3 * STAGE_B, d, 0);
    asm volatile("" :: "v"(negm[0]), "v"(negm[1]));     // SrcC of the first k-step stays allocated until here
    if (LAZY && STATE != 2) {
      // m_run starts at the exact row max of tile 0 rounded up to a whole octave (when resuming it comes from the saved
      // state): un-normalised q.k (Stage II has no qk-norm) may sit tens of octaves away from 0, and a start that far off
      // would send every workgroup through the exact fallback.
      asm volatile("s_nop 15\n\ts_nop 15" : "+v"(s0[0]), "+v"(s0[1]), "+v"(s1c[0]), "+v"(s1c[1]));   // MFMA results landed
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16_t& sa = j == 0 ? s0[0] : s1c[0];
        f32x16_t& sb = j == 0 ? s0[1] : s1c[1];
        RowMax rm;
#pragma unroll
        for (int n = 0; n < 20; ++n) rm.step(n, sa, sb);
        const float m0 = __builtin_ceilf(rm.mx);
        m_run[j] = m0;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] -= m0; sb[r] -= m0; negm[j][r] = -m0; }
        FENCE();
      }
      PIN(negm[0]); PIN(negm[1]);
    }
  };
  bool poison = false;
  uint64_t ok1_prev = ~0ull;
  float part1_prev = 0.f;
  // rare (LAZY): m_run of block j has fallen behind by more than 2^12 - move it up by a whole number of octaves.
  // w = a P tile that has been computed but not consumed yet (nullptr: none), snext = the block's next scores
  // (accumulated, or still accumulating - their MFMAs may be in flight -, not yet exponentiated).
  auto lazy_rebase = [&](int j, float part, u32x4_t (*w)[4], f32x16_t* snext) __attribute__((always_inline)) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(part), __float_as_uint(part), false, false);
    const float rs = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);      // the row's sum over the tile, same in both half-lanes
    const bool bad = !(rs < 0x1p60f);
    poison |= bad;
    const int e = bad ? 0 : max(__builtin_amdgcn_frexp_expf(rs), 0);
    const float delta = (float)e;
    const float alpha = __builtin_amdgcn_exp2f(-delta);                    // exact
    m_run[j] += delta;
    l_run[j] *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[j][r] = -m_run[j];
    FENCE();                            // the rare path runs with every register of the loop live: keep its temporaries few
    o_scale(j, alpha);
    if (w != nullptr) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) (*w)[kk][i] = pack_bf2(bflo((*w)[kk][i]) * alpha, bfhi((*w)[kk][i]) * alpha);
        FENCE();
      }
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(snext[0]), "+v"(snext[1]));
#pragma unroll
    for (int r = 0; r < 16; ++r) { snext[0][r] -= delta; snext[1][r] -= delta; }
    FENCE();
  };

  // ---- one tile, LAZY form: phase 1 = 32 P.V MFMAs || the whole softmax of block 0, phase 2 = 32 QK^T MFMAs || the
  //      whole softmax of block 1; fragment reads, DMA pieces and register sets exactly as in the exact form below ----
  auto iteration_lazy = [&](const int g, f32x16_t (&s1c)[2], f32x16_t (&s1n)[2], u32x4_t (&p0c)[4],
                            u32x4_t (&p0n)[4]) __attribute__((always_inline)) {
    stamp(g, 0);
    if (g > 0) { advance(kcur); advance(vcur); }
    if (AM_A64_PREP) { k_prep(0); v_prep(0); }
    if (!(ABL & 4)) A64_TILE_BARRIER();
    stamp(g, 1);
    const unsigned char* v_st = smem + ((g + 3) & 3) * STAGE_B;   // V^T(g-1)
    const unsigned char* k_st = smem + ((g + 1) & 3) * STAGE_B;   // K(g+1)
    const unsigned char* vn_st = smem + (g & 3) * STAGE_B;        // V^T(g), for the next iteration's first step
    ExpSumPackT<ABL, AM_A64_DOTSUM != 0> es;
    constexpr EsTab32 ES1 = es_make_tab32(es.SEQ, true), ES2 = es_make_tab32(es.SEQ, false);
    auto bf = [](const u32x4_t& w) __attribute__((always_inline)) { return __builtin_bit_cast(bf16x8_t, w); };
    // ===== phase 1: O += V^T(g-1) P^T(g-1) || softmax of block 0; K(g+3) DMA; K(g+1) prefetch =====
    stamp(g, 2);
    es.init();
    FENCE();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int gap = (kk * 4 + d) * 2;
        // AM_A64_SNAKE (round 6 probe): the two MFMAs of a pair share their A operand (the V^T / K fragment); in the plain order the
        // step to the next pair changes BOTH operands (A_d, P0) (A_d, P1) (A_d+1, P0) ..., in the snake order - odd pairs run block 1
        // first - every step changes ONE: (A_d, P0) (A_d, P1) (A_d+1, P1) (A_d+1, P0).  Same per-block accumulation order: same bits.
        const bool sw1 = AM_A64_SNAKE && (d & 1);
        if (sw1) pv_mfma(4 + d, vf[kk & 1][d], bf(p1[kk])); else pv_mfma(d, vf[kk & 1][d], bf(p0c[kk]));
        POST_PV();
        if (kk == 0) { if (AM_A64_PREP) k_issue(); else dma_k(d); }      // one LDS-DMA piece per MFMA pair
        if (d < 2) {                                            // next fragments in gaps 0..3 of the step: landed by its end
          if (kk < 3) vf[(kk + 1) & 1][2 * d] = v_frag(v_st, 2 * d, kk + 1);
          else kf[d][0] = k_frag(k_st, 0, d);
        }
        if (!(ABL & (8 | 32)))
#pragma unroll
          for (int n = ES1.lo[gap]; n < ES1.lo[gap + 1]; ++n) es.step(n, s0[0], s0[1], p0n);
        FENCE();
        if (sw1) pv_mfma(d, vf[kk & 1][d], bf(p0c[kk])); else pv_mfma(4 + d, vf[kk & 1][d], bf(p1[kk]));
        POST_PV();
        if (AM_A64_PREP && kk == 0 && d < 3) k_prep(d + 1);
        if (d < 2) {
          if (kk < 3) vf[(kk + 1) & 1][2 * d + 1] = v_frag(v_st, 2 * d + 1, kk + 1);
          else kf[d][1] = k_frag(k_st, 1, d);
        }
        if (!(ABL & (8 | 32)))
#pragma unroll
          for (int n = ES1.lo[gap + 1]; n < ES1.lo[gap + 2]; ++n) es.step(n, s0[0], s0[1], p0n);
        FENCE();
      }
      HOLD4(vf[kk & 1][0], vf[kk & 1][1], vf[kk & 1][2], vf[kk & 1][3]);
      if (kk == 0) stamp(g, 3);
    }
    stamp(g, 4);
    const float part0 = es.total();
    l_run[0] += part0;
    uint64_t ok0 = __builtin_amdgcn_ballot_w64(part0 <= LAZY_T);    // evaluated here, branched on a phase later:
    asm volatile("" : "+s"(ok0));                                   // no VALU -> branch latency in the loop
    // block 1, previous tile: its P has just been consumed (O and l agree), S(g) is not exponentiated yet
    if (ok1_prev != ~0ull) lazy_rebase(1, part1_prev, nullptr, s1c);
    // ===== phase 2: S(g+1) = K(g+1) Q^T || softmax of block 1; V^T(g+2) DMA; V^T(g) prefetch =====
    es.init();
    FENCE();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int gap = (ks * 2 + kb) * 2;
        const bool sw2 = AM_A64_SNAKE && (kb & 1);
        auto qk_j0 = [&]() __attribute__((always_inline)) {
          if (ks == 0) s0[kb] = qk_mfma_first(ks, kf[ks % 3][kb], negm[0]);
          else qk_mfma_acc(ks, kf[ks % 3][kb], s0[kb]);
        };
        auto qk_j1 = [&]() __attribute__((always_inline)) {
          if (ks == 0) s1n[kb] = qk_mfma_first(8 + ks, kf[ks % 3][kb], negm[1]);
          else qk_mfma_acc(8 + ks, kf[ks % 3][kb], s1n[kb]);
        };
        if (sw2) qk_j1(); else qk_j0();
        POST_QK();
        if (ks < 2) { if (AM_A64_PREP) v_issue(); else dma_v(ks * 2 + kb); }
        if (!(ABL & (8 | 64)))
#pragma unroll
          for (int n = ES2.lo[gap]; n < ES2.lo[gap + 1]; ++n) es.step(n, s1c[0], s1c[1], p1);
        FENCE();
        if (sw2) qk_j0(); else qk_j1();
        POST_QK();
        if (AM_A64_PREP && ks < 2 && ks * 2 + kb < 3) v_prep(ks * 2 + kb + 1);
        if (ks < 6) kf[(ks + 2) % 3][kb] = k_frag(k_st, kb, ks + 2);      // into the set the previous k-step has finished with
        else vf[0][(ks - 6) * 2 + kb] = v_frag(vn_st, (ks - 6) * 2 + kb, 0);
        if (!(ABL & (8 | 64)))
#pragma unroll
          for (int n = ES2.lo[gap + 1]; n < ES2.lo[gap + 2]; ++n) es.step(n, s1c[0], s1c[1], p1);
        FENCE();
      }
      HOLD2(kf[ks % 3][0], kf[ks % 3][1]);
      if (ks == 1) stamp(g, 5);
    }
    stamp(g, 6);
    const float part1 = es.total();
    l_run[1] += part1;
    // block 0, this tile: P(g) is computed but not consumed, S(g+1) is accumulated
    if (ok0 != ~0ull) lazy_rebase(0, part0, &p0n, s0);
    ok1_prev = __builtin_amdgcn_ballot_w64(part1 <= LAZY_T);
    part1_prev = part1;
    asm volatile("" : "+s"(ok1_prev));
  };

  // ---- one tile.  cur = ping-pong set holding S(g) of block 1 and P(g-1) of block 0 ----
  auto iteration = [&](const int g, f32x16_t (&s1c)[2], f32x16_t (&s1n)[2], u32x4_t (&p0c)[4],
                       u32x4_t (&p0n)[4]) __attribute__((always_inline)) {
    stamp(g, 0);
    if (g > 0) { advance(kcur); advance(vcur); }   // past the pieces issued in the previous iteration (scalar work, parked at the barrier)
    if (!(ABL & 4)) {                          // all but the previous iteration's 8 pieces have landed (K(g+1), V^T(g));
      // every wave is done with the slots this iteration refills.  (Not __syncthreads(): its fence drains vmcnt.)
      A64_TILE_BARRIER();
    }
    stamp(g, 1);
    const unsigned char* v_st = smem + ((g + 3) & 3) * STAGE_B;   // V^T(g-1)
    const unsigned char* k_st = smem + ((g + 1) & 3) * STAGE_B;   // K(g+1)
    const unsigned char* vn_st = smem + (g & 3) * STAGE_B;        // V^T(g), for the next iteration's first step
    const bool first = g == 0;          // tile 0 always takes the re-base paths (when resuming they only re-base, `init` false)
    const bool init = first && STATE != 2;
    RowMax rm;
    ExpSumPackT<ABL> es;

    constexpr EsTab ES = es_make_tab(es.SEQ);
    auto es_gap = [&](int gap, f32x16_t& sa, f32x16_t& sb, u32x4_t (&w)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int n = ES.lo[gap]; n < ES.lo[gap + 1]; ++n) es.step(n, sa, sb, w);
    };
    auto bf = [](const u32x4_t& w) __attribute__((always_inline)) { return __builtin_bit_cast(bf16x8_t, w); };
    // ===== phase 1a: first 8 P.V MFMAs (k-step 0) || row max of block 0; K(g+3) DMA =====
    stamp(g, 2);
    FENCE();
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      pv_mfma(d, vf[0][d], bf(p0c[0]));
      dma_k(d);                                               // one LDS-DMA piece per MFMA pair
      if (d < 2) vf[1][2 * d] = v_frag(v_st, 2 * d, 1);       // the next k-step's fragments: issued in gaps 0..3
#pragma unroll
      for (int n = rm_lo(2 * d); n < rm_lo(2 * d + 1); ++n) if (!(ABL & 2)) rm.step(n, s0[0], s0[1]);
      FENCE();
      pv_mfma(4 + d, vf[0][d], bf(p1[0]));
      if (d < 2) vf[1][2 * d + 1] = v_frag(v_st, 2 * d + 1, 1);
#pragma unroll
      for (int n = rm_lo(2 * d + 1); n < rm_lo(2 * d + 2); ++n) if (!(ABL & 2)) rm.step(n, s0[0], s0[1]);
      FENCE();
    }
    HOLD4(vf[0][0], vf[0][1], vf[0][2], vf[0][3]);
    stamp(g, 3);
    bool flag0 = false;
    float alpha0 = 1.f;
    if (ABL & 2) rm.mx = 0.f;
    if (first || !__all(rm.mx <= (float)DEFER)) {      // rare: move m_run of block 0, re-base its scores
      const float delta = init ? rm.mx : __builtin_fmaxf(rm.mx, 0.f);
      alpha0 = init ? 0.f : __builtin_amdgcn_exp2f(-delta);
      m_run[0] += delta;
      l_run[0] *= alpha0;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[0][r] -= delta; s0[1][r] -= delta; negm[0][r] = -m_run[0]; }
      flag0 = true;
    }
    // ===== phase 1b: 24 P.V MFMAs (k-steps 1..3) || exp / row sum / bf16 pack of block 0; K(g+1) prefetch =====
    es.init();
    FENCE();
#pragma unroll
    for (int kk = 1; kk < 4; ++kk) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int gap = ((kk - 1) * 4 + d) * 2;
        pv_mfma(d, vf[kk & 1][d], bf(p0c[kk]));
        if (d < 2) {                            // next fragments in gaps 0..3 of the step: landed by its end
          if (kk < 3) vf[(kk + 1) & 1][2 * d] = v_frag(v_st, 2 * d, kk + 1);
          else kf[d][0] = k_frag(k_st, 0, d);
        }
        if (!(ABL & (8 | 32))) es_gap(gap, s0[0], s0[1], p0n);
        FENCE();
        pv_mfma(4 + d, vf[kk & 1][d], bf(p1[kk]));
        if (d < 2) {
          if (kk < 3) vf[(kk + 1) & 1][2 * d + 1] = v_frag(v_st, 2 * d + 1, kk + 1);
          else kf[d][1] = k_frag(k_st, 1, d);
        }
        if (!(ABL & (8 | 32))) es_gap(gap + 1, s0[0], s0[1], p0n);
        FENCE();
      }
      HOLD4(vf[kk & 1][0], vf[kk & 1][1], vf[kk & 1][2], vf[kk & 1][3]);
    }
    stamp(g, 4);
    l_run[0] += es.total();
    if (flag0) o_scale(0, alpha0);             // rare: O of block 0 is complete through tile g-1 only now
    if (pend1) {                               // rare: S(g) of block 1 was born before block 1's last re-base
#pragma unroll
      for (int r = 0; r < 16; ++r) { s1c[0][r] -= off1; s1c[1][r] -= off1; }
      pend1 = false;
    }

    // ===== phase 2a: first 8 QK^T MFMAs (k-steps 0, 1) || row max of block 1; V^T(g+2) DMA =====
    FENCE();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int gap = (ks * 2 + kb) * 2;
        if (ks == 0) s0[kb] = qk_mfma_first(ks, kf[ks][kb], negm[0]);
        else qk_mfma_acc(ks, kf[ks][kb], s0[kb]);
        dma_v(ks * 2 + kb);
#pragma unroll
        for (int n = rm_lo(gap); n < rm_lo(gap + 1); ++n) if (!(ABL & 2)) rm.step(n, s1c[0], s1c[1]);
        FENCE();
        if (ks == 0) s1n[kb] = qk_mfma_first(8 + ks, kf[ks][kb], negm[1]);
        else qk_mfma_acc(8 + ks, kf[ks][kb], s1n[kb]);
        kf[(ks + 2) % 3][kb] = k_frag(k_st, kb, ks + 2);       // into the set the previous k-step has finished with
#pragma unroll
        for (int n = rm_lo(gap + 1); n < rm_lo(gap + 2); ++n) if (!(ABL & 2)) rm.step(n, s1c[0], s1c[1]);
        FENCE();
      }
      HOLD2(kf[ks][0], kf[ks][1]);
    }
    stamp(g, 5);
    if (ABL & 2) rm.mx = 0.f;
    if (first || !__all(rm.mx <= (float)DEFER)) {      // rare: move m_run of block 1
      const float delta = init ? rm.mx : __builtin_fmaxf(rm.mx, 0.f);
      const float a1 = init ? 0.f : __builtin_amdgcn_exp2f(-delta);
      m_run[1] += delta;
      l_run[1] *= a1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s1c[0][r] -= delta; s1c[1][r] -= delta; negm[1][r] = -m_run[1]; }
      off1 = delta;                            // S(g+1) of block 1 is already accumulating on the old -m_run
      pend1 = true;
      o_scale(1, a1);
    }
    // ===== phase 2b: 24 QK^T MFMAs (k-steps 2..7) || exp / row sum / bf16 pack of block 1; V^T(g) prefetch =====
    es.init();
    FENCE();
#pragma unroll
    for (int ks = 2; ks < 8; ++ks) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int gap = ((ks - 2) * 2 + kb) * 2;
        qk_mfma_acc(ks, kf[ks % 3][kb], s0[kb]);
        if (!(ABL & (8 | 64))) es_gap(gap, s1c[0], s1c[1], p1);
        FENCE();
        qk_mfma_acc(8 + ks, kf[ks % 3][kb], s1n[kb]);
        if (ks < 6) kf[(ks + 2) % 3][kb] = k_frag(k_st, kb, ks + 2);
        else vf[0][(ks - 6) * 2 + kb] = v_frag(vn_st, (ks - 6) * 2 + kb, 0);
        if (!(ABL & (8 | 64))) es_gap(gap + 1, s1c[0], s1c[1], p1);
        FENCE();
      }
      HOLD2(kf[ks % 3][0], kf[ks % 3][1]);
    }
    stamp(g, 6);
    l_run[1] += es.total();
  };

  // ---- main loop, two tiles per trip so the ping-pong sets are compile-time names ----
  auto tile = [&](const int g, f32x16_t (&s1c)[2], f32x16_t (&s1n)[2], u32x4_t (&p0c)[4], u32x4_t (&p0n)[4])
      __attribute__((always_inline)) {
    if constexpr (LAZY) iteration_lazy(g, s1c, s1n, p0c, p0n);
    else iteration(g, s1c, s1n, p0c, p0n);
  };
  // Odd tile counts take their extra tile after the loop (peeled in front of it, the two entry paths meet with every
  // register of the pipeline live in different roles, and hipcc parks dozens of values in AccVGPRs to reconcile them).
  auto finish = [&](u32x4_t (&p0last)[4]) __attribute__((always_inline)) {
    // The last iteration's QK^T MFMAs (scores of a tile that does not exist) are still in flight and hipcc does
    // not know they are MFMAs: hold their destination registers until the results have landed, or the epilogue's
    // temporaries allocated there are overwritten (MFMA D -> any writer: 12 wait states).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // no LDS-DMA piece may outlive the workgroup's LDS allocation
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(s0[0]), "+v"(s0[1]), "+v"(s1[0][0]), "+v"(s1[0][1]), "+v"(s1[1][0]), "+v"(s1[1][1]));
    // drain: O += V^T(n-1) P^T(n-1)
    const unsigned char* v_st = smem + ((n_tiles + 3) & 3) * STAGE_B;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const bf16x8_t vfr = kk == 0 ? vf[0][d] : v_frag(v_st, d, kk);
        pv_mfma(d, vfr, __builtin_bit_cast(bf16x8_t, p0last[kk]));
        pv_mfma(4 + d, vfr, __builtin_bit_cast(bf16x8_t, p1[kk]));
      }
  };
  first_scores(s1[0]);
  int g = 0;
  for (; g + 1 < n_tiles; g += 2) {
    tile(g, s1[0], s1[1], p0[0], p0[1]);
    tile(g + 1, s1[1], s1[0], p0[1], p0[0]);
  }
  if (g < n_tiles) {
    tile(g, s1[0], s1[1], p0[0], p0[1]);
    finish(p0[1]);
  } else {
    finish(p0[0]);
  }
  o_read_fence();

  // ---- epilogue: remove the padded keys' exp2(0 - m) (one partial tile per chunk), normalise, store ----
  const int pad_valid = p.sk - (tiles_per_chunk - 1) * KVBLK;      // valid keys in a chunk's last tile
  int cnt = 0;
  if (pad_valid < KVBLK) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) cnt += min(4, max(0, kb * 32 + 8 * gq + 4 * hi + 4 - pad_valid));
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    // (lazy kernel with AM_A64_DOTSUM: the row sums hold ROUNDED probabilities, a padded key's among them; its m_run is a whole number
    // of octaves unless a resumed state says otherwise, so the rounding is the identity almost always)
    const float p_pad = __builtin_amdgcn_exp2f(-m_run[j]);
    float l = l_run[j] - (float)(cnt * p.nchunks) * ((LAZY && AM_A64_DOTSUM) ? rbf(p_pad) : p_pad);
    l += __shfl_xor(l, 32);
    // LAZY: m_run starts at tile 0's max (or at the saved state's) and only ever moves up by whole octaves when a row sum says so.
    // A row whose scores all sit far below it has lost its sum to underflow: let the exact kernel redo the workgroup.
    if (LAZY) poison |= !(l >= 0x1p-100f && l < 0x1p100f);
    const int q = q0 + 32 * j + l31;
    if (STATE == 1) {          // first pass of two: save (O, m, l), every row of the padded block
      float* sp = p.state + ((int64_t)bh * p.sq_pad + q) * STATE_LD;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<f32x4_t*>(sp + d * 32 + 8 * gq + 4 * hi) = o_read4(j * 4 + d, gq);
      if (hi == 0) { sp[HD] = m_run[j]; sp[HD + 1] = l; }
      continue;
    }
    const float inv = 1.0f / l;
    if (q < p.sq) {
      bf16_t* op = p.O + ((int64_t)seq * p.sq + q) * p.ldo + head * HD + 4 * hi;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const f32x4_t ov = o_read4(j * 4 + d, gq);
          u32x2_t w;
          w[0] = pack_bf2(ov[0] * inv, ov[1] * inv);
          w[1] = pack_bf2(ov[2] * inv, ov[3] * inv);
          *reinterpret_cast<u32x2_t*>(op + d * 32 + 8 * gq) = w;
        }
    }
  }
  if (LAZY) {                 // a jump the lazy re-base cannot represent: have the exact kernel redo this workgroup
    if (__any(poison) && lane == 0) flags[flag_idx] = 1u;
  } else if (flags != nullptr && tid == 0) {
    flags[flag_idx] = 0u;     // every wave read the mark before its first barrier
  }
}

}  // namespace

// Per-device mark buffers of the LAZY kernel: G_REGIONS regions of [4 header words | one mark word per workgroup], all marks
// zero between launches (the exact fallback clears what the lazy pass set).  Consecutive launches (a lazy kernel + its
// fallback) rotate through the regions, so launches that overlap on different streams of one device do not share marks.
// Header word 0 of a region counts the workgroups its fallbacks have recomputed.  Grow-only; launches on one device are
// issued from one thread (C-ABI contract).
constexpr int G_REGIONS = 8;
static unsigned* g_flag_buf[64] = {};
static int64_t g_flag_cap[64] = {};        // mark words per region
static uint64_t g_flag_launch[64] = {};
static uint64_t g_flag_carry[64] = {};     // counts carried over a re-allocation
static int lazy_flag_total(int dev, uint64_t* total) {
  *total = g_flag_carry[dev];
  if (g_flag_buf[dev]) {
    AM_HIP(hipDeviceSynchronize());
    for (int r = 0; r < G_REGIONS; ++r) {
      unsigned c = 0;
      AM_HIP(hipMemcpy(&c, g_flag_buf[dev] + (size_t)r * (g_flag_cap[dev] + 4), sizeof(c), hipMemcpyDeviceToHost));
      *total += c;
    }
  }
  return AM_OK;
}
static int lazy_flags(int64_t n, unsigned** out) {
  int dev = 0;
  AM_HIP(hipGetDevice(&dev));
  AM_CHECK(dev >= 0 && dev < 64, "am_attention64: device index %d", dev);
  if (g_flag_cap[dev] < n) {
    uint64_t total = 0;
    AM_TRY(lazy_flag_total(dev, &total));
    if (g_flag_buf[dev]) AM_HIP(hipFree(g_flag_buf[dev]));
    ++g_am_scratch_generation;
    g_flag_carry[dev] = total;
    g_flag_cap[dev] = n > (1 << 17) ? n : (1 << 17);
    const size_t words = (size_t)G_REGIONS * (g_flag_cap[dev] + 4);
    AM_HIP(hipMalloc(&g_flag_buf[dev], words * sizeof(unsigned)));
    AM_HIP(hipMemset(g_flag_buf[dev], 0, words * sizeof(unsigned)));
  }
  const int region = (int)(g_flag_launch[dev]++ % G_REGIONS);
  *out = g_flag_buf[dev] + (size_t)region * (g_flag_cap[dev] + 4) + 4;
  return AM_OK;
}
// Diagnostic (tests): number of workgroups of the current device the exact fallback kernel has recomputed since the
// library was loaded.  Synchronises the device.
extern "C" int am_attention_fallback_count(uint64_t* count) {
  AM_CHECK(count != nullptr, "am_attention_fallback_count: null argument");
  int dev = 0;
  AM_HIP(hipGetDevice(&dev));
  AM_CHECK(dev >= 0 && dev < 64, "am_attention_fallback_count: device index %d", dev);
  return lazy_flag_total(dev, count);
}

// Main (non-split) grid of the 4x64 kernel: query blocks [0, nblk_main) of every (sequence, head).
// LAZY: the lazy kernel, then the exact kernel over the same grid for the workgroups the lazy one marked (normally none:
// its workgroups read one word and exit).
template <int DEFER, int ABL, int STATE, bool LAZY = false>
static int launch64(const am_attn_args* a, int tiles_per_chunk, int nblk_main, void* stream) {
  AM_ONCE_PER_DEVICE({
    AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd64_kernel<DEFER, ABL, false, STATE, LAZY>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE_B));
  });
  unsigned* flags = nullptr;
  if (LAZY) AM_TRY(lazy_flags((int64_t)nblk_main * a->nseq * a->heads, &flags));
  hipLaunchKernelGGL((attn_fwd64_kernel<DEFER, ABL, false, STATE, LAZY>), dim3(nblk_main, a->nseq * a->heads), dim3(256),
                     NSTAGE * STAGE_B, (hipStream_t)stream, *a, tiles_per_chunk, (unsigned long long*)nullptr, flags, nblk_main);
  if (LAZY && ABL == 0) {
    AM_ONCE_PER_DEVICE({
      AM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd64_kernel<8, 0, false, STATE, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE_B));
    });
    AM_HIP(hipGetLastError());
    hipLaunchKernelGGL((attn_fwd64_kernel<8, 0, false, STATE, false>), dim3(nblk_main, a->nseq * a->heads), dim3(256),
                       NSTAGE * STAGE_B, (hipStream_t)stream, *a, tiles_per_chunk, (unsigned long long*)nullptr, flags, nblk_main);
  }
  return AM_OK;
}
int am_attention64_main(const am_attn_args* a, int tiles_per_chunk, int nblk_main, int defer, void* stream) {
#ifdef AM_ATTN_ABLATIONS
  switch (a->defer_log2) {     // 3000 + ABL: timing ablations (tools/kernel_bench.py --ablate64)
    case 3001: return launch64<8, 1, 0, true>(a, tiles_per_chunk, nblk_main, stream);
    case 3002: return launch64<8, 2, 0, true>(a, tiles_per_chunk, nblk_main, stream);
    case 3004: return launch64<8, 4, 0, true>(a, tiles_per_chunk, nblk_main, stream);
    case 3008: return launch64<8, 8, 0, true>(a, tiles_per_chunk, nblk_main, stream);
    case 3010: return launch64<8, 10, 0, true>(a, tiles_per_chunk, nblk_main, stream);
    case 3014: return launch64<8, 14, 0, true>(a, tiles_per_chunk, nblk_main, stream);
    case 3016: return launch64<8, 16, 0, true>(a, tiles_per_chunk, nblk_main, stream);
    case 3030: return launch64<8, 30, 0, true>(a, tiles_per_chunk, nblk_main, stream);
    case 3032: return launch64<8, 32, 0, true>(a, tiles_per_chunk, nblk_main, stream);
    case 3064: return launch64<8, 64, 0, true>(a, tiles_per_chunk, nblk_main, stream);
    case 3128: return launch64<8, 128, 0, true>(a, tiles_per_chunk, nblk_main, stream);
    case 3132: return launch64<8, 132, 0, true>(a, tiles_per_chunk, nblk_main, stream);
    case 3142: return launch64<8, 142, 0, true>(a, tiles_per_chunk, nblk_main, stream);
    default: break;
  }
#endif
  // defer > 0: the lazy kernel (+ exact fallback); defer == 0: the exact kernel with an immediate re-base;
  // a->defer_log2 == 28: the exact kernel with the deferred re-base on its own (A/B, tests)
#ifdef AM_F16
  const bool exact = true;                 // float16 build: P in IEEE half cannot carry the lazy re-base's 2^+-60 lags - running row max only
#else
  const bool exact = a->defer_log2 == 28;
#endif
  if (a->state_mode == 1)
    return defer == 0 ? launch64<0, 0, 1>(a, tiles_per_chunk, nblk_main, stream)
           : exact    ? launch64<8, 0, 1>(a, tiles_per_chunk, nblk_main, stream)
                      : launch64<8, 0, 1, true>(a, tiles_per_chunk, nblk_main, stream);
  if (a->state_mode == 2)
    return defer == 0 ? launch64<0, 0, 2>(a, tiles_per_chunk, nblk_main, stream)
           : exact    ? launch64<8, 0, 2>(a, tiles_per_chunk, nblk_main, stream)
                      : launch64<8, 0, 2, true>(a, tiles_per_chunk, nblk_main, stream);
  return defer == 0 ? launch64<0, 0, 0>(a, tiles_per_chunk, nblk_main, stream)
         : exact    ? launch64<8, 0, 0>(a, tiles_per_chunk, nblk_main, stream)
                    : launch64<8, 0, 0, true>(a, tiles_per_chunk, nblk_main, stream);
}

#ifdef AM_ATTN_ABLATIONS
// per-phase s_memtime stamps of workgroup (0,0): prof[4 waves][8 tiles (64..71)][8 slots]  (tools/attn_profile.py --k64)
extern "C" int am_attention64_profile(const am_attn_args* a, unsigned long long* prof_dev, void* stream) {
  const bool lazy = a->defer_log2 != 28;
  const void* fn = lazy ? reinterpret_cast<const void*>(attn_fwd64_kernel<8, 0, true, 0, true>)
                        : reinterpret_cast<const void*>(attn_fwd64_kernel<8, 0, true>);
  AM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE_B));
  const int tiles_per_chunk = ceil_div(a->sk, KVBLK);
  const int nblk = ceil_div(a->sq, QBLK);
  unsigned* flags = nullptr;
  AM_TRY(lazy_flags((int64_t)nblk * a->nseq * a->heads, &flags));
  if (lazy)
    hipLaunchKernelGGL((attn_fwd64_kernel<8, 0, true, 0, true>), dim3(nblk, a->nseq * a->heads), dim3(256), NSTAGE * STAGE_B,
                       (hipStream_t)stream, *a, tiles_per_chunk, prof_dev, flags, nblk);
  else
    hipLaunchKernelGGL((attn_fwd64_kernel<8, 0, true>), dim3(nblk, a->nseq * a->heads), dim3(256), NSTAGE * STAGE_B,
                       (hipStream_t)stream, *a, tiles_per_chunk, prof_dev, (unsigned*)nullptr, 0);
  AM_HIP(hipGetLastError());
  return AM_OK;
}
#endif
