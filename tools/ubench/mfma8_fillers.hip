// Round 4 micro-benchmark: how much VALU work hides in the shadow of the e4m3 MX-scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4,
// 16 passes = 64 cycles) - the question behind "the fp8 attention's softmax interval and matrix interval add instead of overlapping".
// Stream per wave = [MFMA, N independent fillers] x 16 x iters; WAVES = 4 (one wave per SIMD) or 8 (two); the accumulators in arch
// VGPRs or in AccVGPRs.  Cycles per MFMA from s_memtime (shader clock, independent of DVFS).  If the fillers hide, cycles per MFMA
// stay at 64 until N x (filler issue cost) exceeds 64; if they do not, it is 64 + N x cost.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma8_fillers tools/ubench/mfma8_fillers.hip && /tmp/mfma8_fillers
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
// KIND 0 v_add_f32, 1 v_exp_f32, 2 v_cvt_pk_fp8_f32, 3 v_cvt_pk_u8_f32, 4 v_max3_f32, 5 v_fma_f32
template <int N, int KIND, int ACC, int THREADS>
__global__ __launch_bounds__(THREADS) void k(unsigned long long* out, int iters, float c) {
  const int tid = threadIdx.x;
  f32x16_t acc[4];
  float v[16];
  int w[4] = {1, 2, 3, 4};
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int i = 0; i < 16; ++i) v[i] = tid * 0.001f + i * 0.125f;
  i32x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + tid; b[i] = 0x30303030 + i; }
  int one = 0x7f7f7f7f;
  asm volatile("" : "+v"(one), "+v"(a), "+v"(b));
  if (ACC == 1) for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(acc[i]));
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (ACC == 0) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc[m & 3]) : "v"(a), "v"(b), "v"(one));
      else asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(acc[m & 3]) : "v"(a), "v"(b), "v"(one));
#pragma unroll
      for (int f = 0; f < N; ++f) {
        const int i = (m * N + f) & 15;
        if (KIND == 0) { v[i] = v[i] + c; asm volatile("" : "+v"(v[i])); }
        else if (KIND == 1) { v[i] = __builtin_amdgcn_exp2f(v[i]); asm volatile("" : "+v"(v[i])); }
        else if (KIND == 2) { w[i & 3] = __builtin_amdgcn_cvt_pk_fp8_f32(v[i], v[(i + 1) & 15], w[i & 3], false); asm volatile("" : "+v"(w[i & 3])); }
        else if (KIND == 3) { asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(w[i & 3]) : "v"(v[i])); }
        else if (KIND == 4) { v[i] = __builtin_fmaxf(__builtin_fmaxf(v[i], v[(i + 5) & 15]), c); asm volatile("" : "+v"(v[i])); }
        else { v[i] = __builtin_fmaf(v[i], c, 0.5f); asm volatile("" : "+v"(v[i])); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += w[i];
  if (s == 1234.5f) out[4096] = 1;
  if ((tid & 63) == 0 && blockIdx.x == 0) out[tid >> 6] = t1 - t0;
}
template <int N, int KIND, int ACC, int THREADS>
void run(unsigned long long* out, const char* name) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<N, KIND, ACC, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, iters, 0.999f);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  double lo = 1e30, hi = 0;
  for (int i = 0; i < THREADS / 64; ++i) { double c = h[i] / (iters * 16.0); lo = c < lo ? c : lo; hi = c > hi ? c : hi; }
  printf("%d waves/SIMD  acc=%s  %-16s x%2d per MFMA: %6.1f .. %6.1f cycles per MFMA per wave  (per SIMD: %6.1f cycles per MFMA)\n", THREADS / 256, ACC ? "agpr" : "vgpr",
         name, N, lo, hi, hi / (THREADS / 256));
}
int main() {
  unsigned long long* out; hipMalloc(&out, 65536);
#define ROW(KIND, NAME, ACC, T) run<0, KIND, ACC, T>(out, NAME); run<2, KIND, ACC, T>(out, NAME); run<4, KIND, ACC, T>(out, NAME); run<8, KIND, ACC, T>(out, NAME); run<12, KIND, ACC, T>(out, NAME); run<16, KIND, ACC, T>(out, NAME);
  ROW(0, "v_add_f32", 0, 256) ROW(0, "v_add_f32", 1, 256) ROW(1, "v_exp_f32", 0, 256) ROW(1, "v_exp_f32", 1, 256)
  ROW(2, "v_cvt_pk_fp8_f32", 1, 256) ROW(3, "v_cvt_pk_u8_f32", 1, 256) ROW(4, "v_max3_f32", 1, 256)
  ROW(0, "v_add_f32", 0, 512) ROW(1, "v_exp_f32", 0, 512) ROW(2, "v_cvt_pk_fp8_f32", 0, 512) ROW(3, "v_cvt_pk_u8_f32", 0, 512)
  return 0;
}
