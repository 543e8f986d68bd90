// Round 6 probe: v_exp_legacy_f32 (assembles for gfx950) against v_exp_f32 - cycles per wave64 instruction with one wave per SIMD
// (s_memtime around 256 independent issues, as tools/ubench/valu_rates.hip), beside an MFMA stream, and its accuracy against exp2
// in double over the range the attention softmax feeds it (x in [-40, 8]) plus the special cases.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/exp_legacy tools/ubench/exp_legacy.hip && /tmp/exp_legacy
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
template <int OP>
__global__ __launch_bounds__(256) void rate(unsigned long long* out, float seed) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    if (OP == 0) asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
    if (OP == 1) asm volatile("v_exp_legacy_f32 %0, %0\n v_exp_legacy_f32 %1, %1\n v_exp_legacy_f32 %2, %2\n v_exp_legacy_f32 %3, %3\n v_exp_legacy_f32 %4, %4\n v_exp_legacy_f32 %5, %5\n v_exp_legacy_f32 %6, %6\n v_exp_legacy_f32 %7, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
    if (OP == 2) asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %4\n v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_add_f32 %6, %6, %7\n v_add_f32 %7, %7, %0" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
    if (OP == 3) asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
    if (OP == 4) asm volatile("v_ldexp_f32 %0, %0, %1\n v_ldexp_f32 %1, %1, %2\n v_ldexp_f32 %2, %2, %3\n v_ldexp_f32 %3, %3, %4\n v_ldexp_f32 %4, %4, %5\n v_ldexp_f32 %5, %5, %6\n v_ldexp_f32 %6, %6, %7\n v_ldexp_f32 %7, %7, %0" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.678f) out[100] = 1;
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
// the filler price beside an MFMA stream (one wave per SIMD): 64 x [MFMA 32x32x16 + N fillers]
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;
template <int OP, int N>
__global__ __launch_bounds__(256) void beside(unsigned long long* out, float seed) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
  bf8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(seed + i); y[i] = (__bf16)(seed - i); }
  f16v acc0 = {}, acc1 = {};
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    if (r & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(x), "v"(y));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(x), "v"(y));
#pragma unroll
    for (int n = 0; n < N; ++n) {
      if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[n & 7]));
      if (OP == 1) asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(a[n & 7]));
      if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[n & 7]) : "v"(a[(n + 1) & 7]));
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  if (s == 12345.678f) out[100] = 1;
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
__global__ void acc(const float* x, float* y0, float* y1, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float a, b;
    asm volatile("v_exp_f32 %0, %2\n\tv_exp_legacy_f32 %1, %2\n\ts_nop 4" : "=&v"(a), "=&v"(b) : "v"(x[i]));
    y0[i] = a; y1[i] = b;
  }
}
int main() {
  unsigned long long* dt; hipMalloc(&dt, 2048); unsigned long long ht[16];
  const char* names[] = {"v_exp_f32", "v_exp_legacy_f32", "v_add_f32", "v_exp_f16", "v_ldexp_f32"};
#define RUN(OP) rate<OP><<<1, 256>>>(dt, 1.0f); hipDeviceSynchronize(); hipMemcpy(ht, dt, 64, hipMemcpyDeviceToHost); printf("%-18s cycles per instruction per wave (one wave per SIMD): %.2f\n", names[OP], ht[0] / 256.0);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
#define RUNB(OP, N) beside<OP, N><<<1, 256>>>(dt, 1.0f); hipDeviceSynchronize(); hipMemcpy(ht, dt, 64, hipMemcpyDeviceToHost); printf("64 x [MFMA 32x32x16 + %d x %-18s]: %.1f cycles per MFMA\n", N, names[OP], ht[0] / 64.0);
  RUNB(2, 0) RUNB(0, 2) RUNB(1, 2) RUNB(2, 2) RUNB(0, 3) RUNB(1, 3) RUNB(2, 3) RUNB(0, 4) RUNB(1, 4) RUNB(2, 4)
  const int n = 1 << 20;
  float* hx = new float[n]; float *dx, *d0, *d1;
  for (int i = 0; i < n; ++i) hx[i] = -40.f + 48.f * (float)i / n;
  float sp[] = {0.f, -0.f, 1.f, -1.f, 127.f, 128.f, 129.f, -126.f, -127.f, -140.f, -149.f, -150.f, -200.f, INFINITY, -INFINITY, NAN, 0.5f, -0.5f};
  const int ns = sizeof(sp) / 4;
  for (int i = 0; i < ns; ++i) hx[i] = sp[i];
  hipMalloc(&dx, n * 4); hipMalloc(&d0, n * 4); hipMalloc(&d1, n * 4);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  acc<<<n / 256, 256>>>(dx, d0, d1, n);
  float* h0 = new float[n]; float* h1 = new float[n];
  hipMemcpy(h0, d0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < ns; ++i) printf("x = %-8g  v_exp_f32 = %-14g v_exp_legacy_f32 = %-14g\n", sp[i], h0[i], h1[i]);
  double w0 = 0, w1 = 0; long differ = 0;
  for (int i = ns; i < n; ++i) {
    const double r = exp2((double)hx[i]);
    w0 = fmax(w0, fabs(h0[i] - r) / r); w1 = fmax(w1, fabs(h1[i] - r) / r);
    differ += h0[i] != h1[i];
  }
  printf("x in [-40, 8], %d points: max relative error v_exp_f32 %.3e (%.2f ulp), v_exp_legacy_f32 %.3e (%.2f ulp); results differ at %ld points\n", n - ns, w0, w0 / 5.96e-8, w1, w1 / 5.96e-8, differ);
  return 0;
}
