// Micro-benchmark: ONE wave per SIMD (256-thread workgroups, 1 per CU).  How many independent VALU instructions
// fit in the shadow of a v_mfma_f32_32x32x16_bf16 issued by the same wave?  Stream = [MFMA, N fillers] repeated.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_fillers mfma_fillers.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// KIND 0: v_fma_f32 fillers, 1: v_exp_f32, 2: v_add_f32 chain-free, 3: v_cvt_pk_bf16_f32, 4: v_max3_f32
// ACC 0: MFMA accumulators in VGPRs (builtin), 1: literal AGPRs via asm
template <int N, int KIND, int ACC>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float c) {
  const int tid = threadIdx.x;
  f32x16_t acc[4];
  float v[16];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int i = 0; i < 16; ++i) v[i] = tid * 0.001f + i;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(tid * 0.01f + i); b[i] = (__bf16)(1.0f - i * 0.1f); }
  if (ACC == 1) {
    asm volatile("v_accvgpr_write_b32 a0, 0" ::: "a0");
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (ACC == 0) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
      else if ((m & 3) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" :: "v"(a), "v"(b) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15");
      else if ((m & 3) == 1) asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" :: "v"(a), "v"(b) : "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
      else if ((m & 3) == 2) asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" :: "v"(a), "v"(b) : "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47");
      else asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" :: "v"(a), "v"(b) : "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63");
#pragma unroll
      for (int f = 0; f < N; ++f) {
        const int i = (m * N + f) & 15;
        if (KIND == 0) v[i] = __builtin_fmaf(v[i], c, 0.5f);
        else if (KIND == 1) v[i] = __builtin_amdgcn_exp2f(v[i]);
        else if (KIND == 2) v[i] = v[i] + c;
        else if (KIND == 3) { auto w = __builtin_convertvector((__attribute__((ext_vector_type(2))) float){v[i], v[(i + 1) & 15]}, __attribute__((ext_vector_type(2))) __bf16); v[i] = __builtin_bit_cast(float, w); }
        else v[i] = __builtin_fmaxf(__builtin_fmaxf(v[i], v[(i + 5) & 15]), c);
        asm volatile("" : "+v"(v[i]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  if (ACC == 0) for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 256 + tid] = s;
}

template <int N, int KIND, int ACC>
void run(float* out, const char* name) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<N, KIND, ACC>), dim3(256), dim3(256), 0, 0, out, 10, 0.999f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<N, KIND, ACC>), dim3(256), dim3(256), 0, 0, out, iters, 0.999f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ns_per_mfma = ms * 1e6 / (iters * 16.0);
  printf("%-10s acc=%s fillers/MFMA=%d: %7.2f ns/MFMA (= %5.1f cyc @2.0 GHz)\n", name, ACC ? "agpr" : "vgpr", N, ns_per_mfma, ns_per_mfma * 2.0);
}

int main() {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  run<0, 0, 0>(out, "none"); run<0, 0, 1>(out, "none");
  run<2, 0, 0>(out, "v_fma"); run<4, 0, 0>(out, "v_fma"); run<6, 0, 0>(out, "v_fma"); run<8, 0, 0>(out, "v_fma");
  run<2, 0, 1>(out, "v_fma"); run<4, 0, 1>(out, "v_fma"); run<6, 0, 1>(out, "v_fma"); run<8, 0, 1>(out, "v_fma");
  run<2, 1, 1>(out, "v_exp"); run<4, 1, 1>(out, "v_exp"); run<6, 1, 1>(out, "v_exp");
  run<4, 2, 1>(out, "v_add"); run<4, 3, 1>(out, "v_cvt_pk"); run<4, 4, 1>(out, "v_max3");
  run<4, 1, 0>(out, "v_exp"); run<4, 4, 0>(out, "v_max3");
  return 0;
}
