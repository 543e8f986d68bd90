// Shared device/host helpers for libactionmesh_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/actionmesh_amd.h"

// The library's 16-bit storage / MFMA element type.  Default: bfloat16 (the reference's default autocast dtype).  The SAME sources built
// with -DAM_F16 are libactionmesh_amd_f16.so, the reference CLI's `--dtype float16` (inference/video_to_animated_mesh.py:153,222;
// pipeline.py:671): every "bf16" name below then means IEEE half - storage, conversions (round to nearest even), the rounding points that
// mirror autocast, and the MFMA element type (v_mfma_f32_32x32x16_f16 / 16x16x32_f16: same shapes, same rate, same register layout).
// The 4x64 attention runs its EXACT form there (running row max): the lazy re-base of the bf16 build needs bf16's 8-bit exponent for P.
#ifdef AM_F16
typedef _Float16 am_h16;
#define AM_H16_NAME "f16"
#define AM_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define AM_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#else
typedef __bf16 am_h16;
#define AM_H16_NAME "bf16"
#define AM_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define AM_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
typedef uint16_t bf16_t;  // raw 16-bit payload (bf16; IEEE half in the AM_F16 build)
typedef __attribute__((ext_vector_type(8))) am_h16 bf16x8_t;   // MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;  // 16-byte vector load/store unit
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

// ---- error plumbing -------------------------------------------------------
void am_set_error(const char* fmt, ...);
#define AM_FAIL(code, ...)        \
  do {                            \
    am_set_error(__VA_ARGS__);    \
    return (code);                \
  } while (0)
#define AM_HIP(call)                                                                    \
  do {                                                                                  \
    hipError_t e__ = (call);                                                            \
    if (e__ != hipSuccess)                                                              \
      AM_FAIL(AM_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)
#define AM_CHECK(cond, ...)                        \
  do {                                             \
    if (!(cond)) AM_FAIL(AM_ERR_INVALID, __VA_ARGS__); \
  } while (0)
#define AM_TRY(call)              \
  do {                            \
    int s__ = (call);             \
    if (s__ != AM_OK) return s__; \
  } while (0)

// ---- 16-bit <-> f32 (round-to-nearest-even, NaN preserved) -------------------
#ifdef AM_F16
__host__ __device__ inline float bf2f(bf16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
__host__ __device__ inline bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
#else
__host__ __device__ inline float bf2f(bf16_t b) {
  union { uint32_t u; float f; } c;
  c.u = (uint32_t)b << 16;
  return c.f;
}
__host__ __device__ inline bf16_t f2bf(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  uint32_t u = c.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x0040u);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
#endif
// Device-side conversions use the gfx950 hardware converters (v_cvt_pk_bf16_f32 / v_cvt_f16_f32, RNE).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) am_h16 bf16x2_t;
// round an fp32 value to the 16-bit type and back (the "autocast result" rounding)
__device__ inline float rbf(float f) { return (float)(am_h16)f; }
__device__ inline uint32_t pack_bf2(float lo, float hi) {
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}
__device__ inline float bflo(uint32_t w) { return bf2f((bf16_t)(w & 0xffffu)); }
__device__ inline float bfhi(uint32_t w) { return bf2f((bf16_t)(w >> 16)); }

// RoPE rotation of one channel pair (rotary_embedding.py:116-122: x cos + rotate(x) sin in fp32), with the roundings spelled out so
// that every kernel that rotates (head_post_kernel, the fused QKV epilogue of am_gemm.hip) produces the same bits whatever the
// compiler would have contracted: x0' = fma(x0, c, -(x1 s)), x1' = fma(x0, s, x1 c).
__device__ inline void rope_rotate(float& x0, float& x1, float c, float s) {
  const float a = x0, b = x1;
  const float bs = b * s, bc = b * c;
  x0 = __builtin_fmaf(a, c, -bs);
  x1 = __builtin_fmaf(a, s, bc);
}

// ---- canonical LayerNorm row statistics (the folded LayerNorms of am_model.hip) ------------------------------------------------
// A row's (mean, rstd) must be the same BITS whichever kernel wrote the row - a GEMM's store loop (emit_row_part, am_gemm.hip), the
// read-back pass behind a GEMM (row_part_kernel), a LayerNorm's own output, or the stand-alone statistics pass - because the exact
// shortcuts of am_set_branch_hints and the graph replays are tested bit for bit against the full computation.  One definition:
//   slice   = 256 consecutive columns = 32 groups of 8 values;
//   level 0 = (mean, M2) of a group: the balanced sum / 8, then M2 by an fma chain over (x - mean);
//   1 ... 5 = equal-count merges  mean = (a + b) / 2,  M2 = M2a + M2b + (a - b)^2 n / 2  (Chan et al.) up a balanced binary tree of
//             neighbouring groups (lane xor 1, 2, 4, 8, 16; both partners of a merge compute identical values, so any lane of the
//             other half is a valid partner - DPP mirrors in the GEMM, shuffles in the row kernels);
//   slices  -> row: merged left to right by row_stats_merge.
// Every operation is spelled as the instruction it must become (explicit fmaf; products and sums that feed an fma only through
// its multiplicand / a named temporary), so that the compiler's contraction choices cannot differ between the kernels.
__device__ __forceinline__ void row_part8(const float (&x)[8], float& mean, float& m2) {
  mean = (((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]))) * 0.125f;
  m2 = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float d = x[e] - mean;
    m2 = __builtin_fmaf(d, d, m2);
  }
}
__device__ __forceinline__ void row_part_merge_equal(float& mean, float& m2, float om, float oq, float half_n) {
  const float d = om - mean;
  const float dd = d * d;
  const float s = m2 + oq;
  m2 = __builtin_fmaf(dd, half_n, s);
  mean = 0.5f * (mean + om);
}
__device__ __forceinline__ void row_stats_merge(float& n, float& mean, float& m2, float nj, float mj, float qj) {
  const float tot = n + nj;
  const float d = mj - mean;
  const float w = nj / tot;
  mean = __builtin_fmaf(d, w, mean);
  const float dd = d * d;
  const float k = n * w;
  const float s = m2 + qj;
  m2 = __builtin_fmaf(dd, k, s);
  n = tot;
}
__device__ __forceinline__ float row_stats_rstd(float n, float m2, float eps) {
  const float var = m2 / n;
  return rsqrtf(var + eps);
}

// erf-GELU (F.gelu(approximate="none")).  erfc(|x|/sqrt2) by Abramowitz-Stegun 7.1.26 on the hardware
// rcp / exp2: |error| <= 3.3e-7 absolute, <= 1.7e-4 relative for |gelu| > 1e-3 - an order of magnitude
// below the bf16 rounding applied to the result - at ~14 instructions instead of ~50 for libm erff
// (the GELU epilogue was 40 % of the ff1 GEMM's time).
__device__ inline float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float p = 1.061405429f;
  p = __builtin_fmaf(p, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float half_erfc = 0.5f * p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  return x >= 0.f ? __builtin_fmaf(-x, half_erfc, x) : x * half_erfc;
}

// Barrier that publishes LDS-DMA (global_load_lds) data: the DMA is a pending VMEM op of the issuing
// wave, and hipcc does not reliably emit the vmcnt wait in front of s_barrier on every loop path
// (observed: missing on a loop back-edge when a DMA sits under a wave-uniform branch), so the wait is
// explicit.  Every wave drains its own DMAs, then the barrier makes all of them visible.
__device__ inline void dma_drain_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// hipFuncSetAttribute (dynamic LDS above 64 KiB) is per device: a once-flag per (call site, device), safe from any thread.
// Usage:  AM_ONCE_PER_DEVICE({ AM_HIP(hipFuncSetAttribute(...)); ... });
#include <atomic>
#include <mutex>
#define AM_ONCE_PER_DEVICE(BODY)                                                             \
  do {                                                                                       \
    static std::atomic<bool> done__[64];                                                     \
    static std::mutex mu__;                                                                  \
    int dev__ = 0;                                                                           \
    AM_HIP(hipGetDevice(&dev__));                                                            \
    AM_CHECK(dev__ >= 0 && dev__ < 64, "device index %d out of range", dev__);               \
    if (!done__[dev__].load(std::memory_order_acquire)) {                                    \
      std::lock_guard<std::mutex> lk__(mu__);                                                \
      if (!done__[dev__].load(std::memory_order_relaxed)) {                                  \
        BODY                                                                                 \
        done__[dev__].store(true, std::memory_order_release);                                \
      }                                                                                      \
    }                                                                                        \
  } while (0)

// Library-owned device scratch that is grown on demand (the split-tail partials, the lazy kernel's mark buffers) is freed when it
// grows: everything that bakes such a pointer into a HIP graph keys the graph on this generation (am_model.hip).
extern std::atomic<uint64_t> g_am_scratch_generation;

// Diagnostic trace (am_debug_trace_begin / _end, am_elementwise.hip): while a trace is open, am_trace() appends a position-weighted
// integer checksum of a device buffer to the caller's device log, in stream order (one tiny kernel; integer adds, so the value does
// not depend on the order the words are visited in).  A no-op otherwise.  tag = stage * 100 + layer (am_debug_trace_stage_name).
void am_trace(int tag, const void* dev_ptr, size_t bytes, void* stream);
bool am_trace_on();

int am_head_post_check(const am_headpost_args* a);      // am_norm.hip: argument validation shared with the fused QKV GEMM

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// Within every aligned group of 16 keys, V^T stores key k at position
// perm16(k) = k with bits 2 and 3 swapped (an involution).  This makes the 8
// keys a lane holds after the S^T = K Q^T MFMA ((r&3) + 8*(r>>2) + 4*hi, the
// 32x32 C/D layout) contiguous in V^T, so the P.V MFMA A-operand is one
// 16-byte LDS read.  See am_attention.hip.
__host__ __device__ inline int perm16(int k) {
  return (k & ~0xC) | ((k & 4) << 1) | ((k & 8) >> 1);
}
