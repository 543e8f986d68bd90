// Micro-benchmark: do MFMA and VALU work from the two waves that share a SIMD overlap on gfx950?
// 1 workgroup of 8 waves per CU (waves w and w+4 share a SIMD).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ void mfma_block(f32x16_t (&acc)[4], bf16x8_t a, bf16x8_t b) {
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
}
__device__ __forceinline__ void valu_block(float (&v)[16], float c) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], c, 0.5f);
}
__device__ __forceinline__ void exp_block(float (&v)[16]) {
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
}

// MODE 0: all waves MFMA; 1: all waves VALU(fma); 2: waves 0-3 MFMA / 4-7 VALU; 3: each wave alternates
// 16 MFMA + 64 VALU per iteration; 4: all waves exp; 5: waves 0-3 MFMA / 4-7 exp; 6: MODE 3 with s_setprio
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, float c) {
  const int tid = threadIdx.x;
  const bool late = __builtin_amdgcn_readfirstlane(tid) >= 256;
  f32x16_t acc[4];
  float v[16];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int i = 0; i < 16; ++i) v[i] = tid * 0.001f + i;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(tid * 0.01f + i); b[i] = (__bf16)(1.0f - i * 0.1f); }
  if ((MODE == 12 || MODE == 13) && late) __syncthreads();     // stagger waves 4-7 by one phase
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) mfma_block(acc, a, b);
    else if (MODE == 1) valu_block(v, c);
    else if (MODE == 2) { if (late) valu_block(v, c); else mfma_block(acc, a, b); }
    else if (MODE == 3) { mfma_block(acc, a, b); valu_block(v, c); }
    else if (MODE == 4) exp_block(v);
    else if (MODE == 5) { if (late) exp_block(v); else mfma_block(acc, a, b); }
    else if (MODE == 6) {
      __builtin_amdgcn_s_setprio(1); mfma_block(acc, a, b); __builtin_amdgcn_s_setprio(0); valu_block(v, c);
    }
    else if (MODE == 7) { if (!late) mfma_block(acc, a, b); }
    else if (MODE == 8) { if (late) valu_block(v, c); else { __builtin_amdgcn_s_setprio(3); mfma_block(acc, a, b); } }
    else if (MODE == 9) { if (late) { __builtin_amdgcn_s_setprio(3); valu_block(v, c); } else mfma_block(acc, a, b); }
    else if (MODE == 10 || MODE == 11 || MODE == 12 || MODE == 13) {
      // attention-like iteration: 16 MFMA | ~100 VALU (64 fma + 32 exp) | 16 MFMA
      if (MODE == 11) __syncthreads();
      if (MODE == 12 || MODE == 13) __syncthreads();
      if (MODE == 13) __builtin_amdgcn_s_setprio(1);
      mfma_block(acc, a, b);
      if (MODE == 13) __builtin_amdgcn_s_setprio(0);
      if (MODE == 12 || MODE == 13) __syncthreads();
      valu_block(v, c); exp_block(v);
      if (MODE == 13) __builtin_amdgcn_s_setprio(1);
      mfma_block(acc, a, b);
      if (MODE == 13) __builtin_amdgcn_s_setprio(0);
    }
  }
  if ((MODE == 12 || MODE == 13) && !late) __syncthreads();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
float run(float* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, 10, 0.999f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, 0.999f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  const char* names[] = {"all waves: 16 MFMA/iter", "all waves: 64 v_fma/iter", "waves0-3 MFMA | waves4-7 v_fma",
                         "every wave: 16 MFMA then 64 v_fma", "all waves: 32 v_exp/iter", "waves0-3 MFMA | waves4-7 v_exp",
                         "every wave: setprio(1) 16 MFMA setprio(0) 64 v_fma",
                         "waves0-3 MFMA | waves4-7 idle", "waves0-3 MFMA prio3 | waves4-7 v_fma", "waves0-3 MFMA | waves4-7 v_fma prio3",
                         "attn-like 16MFMA|64fma+32exp|16MFMA free-running", "attn-like + 1 barrier/iter (lockstep)",
                         "attn-like, halves staggered, 2 barriers/iter", "attn-like staggered + setprio on MFMA"};
  float t[14] = {run<0>(out, iters), run<1>(out, iters), run<2>(out, iters), run<3>(out, iters),
                run<4>(out, iters), run<5>(out, iters), run<6>(out, iters), run<7>(out, iters), run<8>(out, iters),
                run<9>(out, iters), run<10>(out, iters), run<11>(out, iters), run<12>(out, iters), run<13>(out, iters)};
  for (int m = 0; m < 14; ++m) {
    double cyc = t[m] * 1e-3 * 2.4e9 / iters;   // cycles per iteration at 2.4 GHz nominal
    printf("mode %d %-52s %8.3f ms  %7.1f cyc/iter@2.4GHz\n", m, names[m], t[m], cyc);
  }
  double tf = 2.0 * 32 * 32 * 16 * 16.0 * iters * 8 * 256 / (t[0] * 1e-3) / 1e12;
  printf("mode 0 MFMA rate: %.0f TFLOP/s\n", tf);
  return 0;
}
