// Micro-benchmark of a GEMM main loop's instruction mix in isolation, ONE wave per SIMD (4 waves per workgroup, one workgroup
// per CU): per "k-step" 16 AGPR-form v_mfma_f32_32x32x16_bf16 (or 32 v_mfma_f32_16x16x32_bf16: same flops) with R
// ds_read_b128 and D LDS-DMA pieces (global_load_lds_dwordx4) threaded between them, a counted-vmcnt barrier every two
// k-steps.  The DMA source is a 64 KiB window per workgroup (L2-resident after the first pass) or the whole buffer
// streamed once (HBM/MALL).  Answers: is ~50 % MFMA utilisation a property of the instruction mix, or of operand delivery?
// Build: hipcc --offload-arch=gfx950 -O3 -o gemm_stream gemm_stream.hip ; run: ./gemm_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define S(x) #x
#define MF32(lo, hi) asm volatile("v_mfma_f32_32x32x16_bf16 a[" S(lo) ":" S(hi) "], %0, %1, a[" S(lo) ":" S(hi) "]" :: "v"(a), "v"(b))
#define MF16(lo, hi) asm volatile("v_mfma_f32_16x16x32_bf16 a[" S(lo) ":" S(hi) "], %0, %1, a[" S(lo) ":" S(hi) "]" :: "v"(a), "v"(b))

template <int N> struct I {};
__device__ __forceinline__ void mf32(int i, bf16x8_t a, bf16x8_t b) {
  switch (i & 15) {
    case 0: MF32(0, 15); break; case 1: MF32(16, 31); break; case 2: MF32(32, 47); break; case 3: MF32(48, 63); break;
    case 4: MF32(64, 79); break; case 5: MF32(80, 95); break; case 6: MF32(96, 111); break; case 7: MF32(112, 127); break;
    case 8: MF32(128, 143); break; case 9: MF32(144, 159); break; case 10: MF32(160, 175); break; case 11: MF32(176, 191); break;
    case 12: MF32(192, 207); break; case 13: MF32(208, 223); break; case 14: MF32(224, 239); break; default: MF32(240, 255); break;
  }
}
__device__ __forceinline__ void mf16(int i, bf16x8_t a, bf16x8_t b) {      // 32 distinct 4-register accumulators
  switch (i & 31) {
    case 0: MF16(0, 3); break; case 1: MF16(4, 7); break; case 2: MF16(8, 11); break; case 3: MF16(12, 15); break;
    case 4: MF16(16, 19); break; case 5: MF16(20, 23); break; case 6: MF16(24, 27); break; case 7: MF16(28, 31); break;
    case 8: MF16(32, 35); break; case 9: MF16(36, 39); break; case 10: MF16(40, 43); break; case 11: MF16(44, 47); break;
    case 12: MF16(48, 51); break; case 13: MF16(52, 55); break; case 14: MF16(56, 59); break; case 15: MF16(60, 63); break;
    case 16: MF16(64, 67); break; case 17: MF16(68, 71); break; case 18: MF16(72, 75); break; case 19: MF16(76, 79); break;
    case 20: MF16(80, 83); break; case 21: MF16(84, 87); break; case 22: MF16(88, 91); break; case 23: MF16(92, 95); break;
    case 24: MF16(96, 99); break; case 25: MF16(100, 103); break; case 26: MF16(104, 107); break; case 27: MF16(108, 111); break;
    case 28: MF16(112, 115); break; case 29: MF16(116, 119); break; case 30: MF16(120, 123); break; default: MF16(124, 127); break;
  }
}

// READS: ds_read_b128 per k-step (0 or 8); DMA: LDS-DMA pieces per k-step (0 or 4); M16: 16x16x32 MFMAs; STREAM: DMA source walks
// the whole buffer instead of a 64 KiB window
template <int READS, int DMA, bool M16, bool STREAM, int WAITN = 8, int SKEW = 0>
__global__ __launch_bounds__(256, 1) void k(const unsigned char* __restrict__ src, size_t src_bytes, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  asm volatile("v_accvgpr_write_b32 a255, 0" ::: "a255");
  bf16x8_t fr[2][8];
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < 8; ++i)
      for (int e = 0; e < 8; ++e) fr[s][i][e] = (__bf16)(0.001f * (tid + i + e));
  for (int i = tid; i < 131072 / 16; i += 256) *reinterpret_cast<u32x4_t*>(smem + i * 16) = u32x4_t{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  __syncthreads();
  const unsigned char* lp = smem + lane * 16 + wave * 1024;
  const size_t window = STREAM ? src_bytes : 65536;
  size_t pos = STREAM ? ((size_t)blockIdx.x * 65536) % src_bytes : (size_t)blockIdx.x * 65536 % src_bytes;
  int piece = 0;
  u32x4_t stg[2][4] = {};
  for (int it = 0; it < iters; ++it) {
    if (DMA == 104) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (DMA) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(DMA == 2 || DMA == 3 ? WAITN / 2 : DMA == 16 ? 32 : WAITN) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int o = s ^ 1;
      FENCE();
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        if (M16) { mf16(2 * n, fr[s][n & 7], fr[s][(n >> 2) & 7]); mf16(2 * n + 1, fr[s][n & 7], fr[s][(n >> 2) & 7]); }
        else mf32(n, fr[s][n & 3], fr[s][4 + (n >> 2)]);
        if (READS && n < 8) fr[o][n] = *reinterpret_cast<const bf16x8_t*>(lp + (((it * 2 + s) * 8 + n) & 31) * 4096);
        if (DMA == 104 && n >= 8) {          // register-staged: 4 global loads (VGPR) + 4 ds_write_b128 per k-step
          const int i = (n - 8) >> 1;
          if ((n & 1) == 0) {
            const size_t o2 = STREAM ? (pos + (size_t)(piece & 63) * 1024 + (size_t)wave * 16384) % src_bytes
                                     : ((size_t)blockIdx.x * 65536 + ((size_t)(piece & 15) * 1024 + (size_t)wave * 16384)) % src_bytes;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stg[s][i]) : "v"(src + o2 + lane * 16) : "memory");
            ++piece;
          } else {
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(stg[o][i]));
            *reinterpret_cast<u32x4_t*>(smem + ((piece + i) & 31) * 1024 + wave * 32768 + lane * 16) = stg[o][i];
          }
        }
        if (DMA == 16) {                      // same bytes as DMA = 4, moved 4 B per lane: one 256-byte piece per MFMA
          const size_t o3 = ((size_t)blockIdx.x * 65536 + ((size_t)(piece & 63) * 256 + (size_t)wave * 16384)) % (src_bytes - 65536);
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + o3 + lane * 4), (lds_ptr_t)(smem + (piece & 127) * 256 + wave * 32768), 4, 0, 0);
          ++piece;
        }
        if (DMA == 3 ? (n & 7) == 2 * wave : DMA && DMA < 16 && n >= 8 && (DMA == 2 ? (n & 3) == 0 : (n & 1) == 0)) {   // 3: 2 pieces, waves staggered
          const size_t off = (pos + (size_t)(piece & 63) * 1024 + (size_t)wave * 16384) % (STREAM ? src_bytes : (size_t)src_bytes);
          // SKEW: every workgroup's window starts at a different offset inside the 4 KiB / 64 KiB address-interleave periods
          const size_t o2 = STREAM ? off : ((size_t)blockIdx.x * (65536 + SKEW) + ((size_t)((piece + (SKEW ? blockIdx.x * 5 : 0)) & 15) * 1024 + (size_t)wave * 16384)) % (src_bytes - 65536);
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + o2 + lane * 16), (lds_ptr_t)(smem + (piece & 31) * 1024 + wave * 32768), 16, 0, 0);
          ++piece;
        }
        FENCE();
      }
      asm volatile("" :: "v"(fr[s][0]), "v"(fr[s][1]), "v"(fr[s][2]), "v"(fr[s][3]), "v"(fr[s][4]), "v"(fr[s][5]), "v"(fr[s][6]), "v"(fr[s][7]));
    }
    if (STREAM) pos = (pos + (size_t)gridDim.x * 65536) % src_bytes;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float v;
  asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a0" : "=v"(v));
  if (v == 123456.f) out[tid] = v;
}

template <int READS, int DMA, bool M16, bool STREAM, int WAITN = 8, int SKEW = 0>
static void run(const char* name, const unsigned char* src, size_t bytes, float* out) {
  const int iters = 4000, grid = 256 * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<READS, DMA, M16, STREAM, WAITN, SKEW>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<READS, DMA, M16, STREAM, WAITN, SKEW>), dim3(grid), dim3(256), 131072, 0, src, bytes, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<READS, DMA, M16, STREAM, WAITN, SKEW>), dim3(grid), dim3(256), 131072, 0, src, bytes, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * 32 * 32768.0;       // 32 MFMA-equivalents (32x32x16) per wave per trip
  printf("%-58s %8.3f ms  %7.1f TFLOP/s  (%s)\n", name, ms, flops / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  unsigned char* src; float* out;
  hipMalloc(&src, bytes); hipMemset(src, 0x3c, bytes); hipMalloc(&out, 4096);
  run<0, 0, false, false>("32x32x16  MFMA only", src, bytes, out);
  run<8, 0, false, false>("32x32x16  + 8 ds_read_b128 / 16 MFMA", src, bytes, out);
  run<0, 4, false, false>("32x32x16  + 4 DMA pieces / 16 MFMA (L2 window)", src, bytes, out);
  run<8, 4, false, false>("32x32x16  + 8 reads + 4 DMA (L2 window)", src, bytes, out);
  run<8, 4, false, true>("32x32x16  + 8 reads + 4 DMA (streamed from HBM)", src, bytes, out);
  run<8, 2, false, false>("32x32x16  + 8 reads + 2 DMA pieces / 16 MFMA (the attention kernel's ratio, L2)", src, bytes, out);
  run<8, 3, false, false>("32x32x16  + 8 reads + 2 DMA pieces / 16 MFMA, the four waves staggered by 2 MFMAs (L2)", src, bytes, out);
  run<8, 2, false, true>("32x32x16  + 8 reads + 2 DMA pieces / 16 MFMA (HBM stream)", src, bytes, out);
  run<8, 16, false, false>("32x32x16  + 8 reads + 16 DMA pieces of 4 B/lane / 16 MFMA (same bytes, L2)", src, bytes, out);
  run<8, 4, false, false, 0>("32x32x16  + 8 reads + 4 DMA (L2), <= 8 pieces in flight per wave", src, bytes, out);
  run<8, 4, false, false, 16>("32x32x16  + 8 reads + 4 DMA (L2), <= 24 in flight", src, bytes, out);
  run<8, 4, false, false, 32>("32x32x16  + 8 reads + 4 DMA (L2), <= 40 in flight", src, bytes, out);
  run<8, 4, false, false, 56>("32x32x16  + 8 reads + 4 DMA (L2), <= 64 in flight", src, bytes, out);
  run<8, 4, false, true, 56>("32x32x16  + 8 reads + 4 DMA (HBM stream), <= 64 in flight", src, bytes, out);
  run<8, 4, false, false, 8, 4352>("32x32x16  + 8 reads + 4 DMA (L2), windows skewed by 4352 B, rotated", src, bytes, out);
  run<8, 4, false, false, 8, 256>("32x32x16  + 8 reads + 4 DMA (L2), windows skewed by 256 B, rotated", src, bytes, out);
  run<8, 104, false, false>("32x32x16  + 8 reads + 4 (global_load + ds_write) (L2 window)", src, bytes, out);
  run<8, 104, false, true>("32x32x16  + 8 reads + 4 (global_load + ds_write) (HBM stream)", src, bytes, out);
  run<0, 0, true, false>("16x16x32  MFMA only", src, bytes, out);
  run<8, 4, true, false>("16x16x32  + 8 reads + 4 DMA (L2 window)", src, bytes, out);
  run<8, 4, true, true>("16x16x32  + 8 reads + 4 DMA (streamed from HBM)", src, bytes, out);
  return 0;
}
