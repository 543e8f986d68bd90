// Micro-benchmark of one "phase 1b" of the 4x64 attention kernel in isolation, ONE wave per SIMD:
// 24 AGPR-accumulating MFMAs with the exp / row-sum / bf16-pack steps of 32 scores threaded through the gaps.
// Variants isolate what keeps the fillers from hiding.  Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
#define PIN(x) asm volatile("" : "+v"(x))
#define FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void mf(int i, bf16x8_t a, bf16x8_t b) {
  switch (i & 3) {
    case 0: asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" :: "v"(a), "v"(b) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15"); break;
    case 1: asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" :: "v"(a), "v"(b) : "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31"); break;
    case 2: asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" :: "v"(a), "v"(b) : "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47"); break;
    default: asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" :: "v"(a), "v"(b) : "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63"); break;
  }
}

// MODE bit0: dependent softmax steps (else independent fma fillers of the same count)
//      bit1: one ds_read_b128 per 2 MFMAs feeding the A operand
//      bit2: evenly spread (3,3,4 per gap by cost) instead of the kernel's order
//      bit3: no exp (mul instead)
//      bit4: drop the FENCE()s
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float c) {
  extern __shared__ unsigned char smem[];
  const int tid = threadIdx.x;
  f32x16_t sa, sb;
  for (int r = 0; r < 16; ++r) { sa[r] = -0.01f * (tid + r); sb[r] = -0.02f * (tid + r); }
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(tid * 0.01f + i); b[i] = (__bf16)(1.0f - i * 0.1f); }
  for (int i = tid; i < 16384 / 16; i += 256) *reinterpret_cast<u32x4_t*>(smem + i * 16) = u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  const unsigned char* lp = smem + (tid & 63) * 16;
  float rs[4] = {0.f, 0.f, 0.f, 0.f};
  unsigned w[16];
  for (int i = 0; i < 16; ++i) w[i] = 0;
  bf16x8_t fr[4] = {a, a, a, a};
  bf16x8_t fr2[2][4] = {{a, a, a, a}, {a, a, a, a}};
  auto get = [&](int e) { return e < 16 ? sa[e] : sb[e - 16]; };
  auto ex = [&](int e) { float v = (MODE & 8) ? get(e) * 0.999f : __builtin_amdgcn_exp2f(get(e)); PIN(v); if (e < 16) sa[e] = v; else sb[e - 16] = v; };
  auto ad = [&](int e) { rs[e & 3] += get(e); PIN(rs[e & 3]); };
  auto pk = [&](int pr) { f32x2_t f = {get(2 * pr), get(2 * pr + 1)}; unsigned v = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t)); PIN(v); w[pr] = v; };
  constexpr int SKEW = 1 + ((MODE >> 5) & 3);
  // generic skewed order: round r (0..15+SKEW): exp of pair r (if r < 16), adds + pack of pair r - SKEW (if >= 0)
  auto step_skew = [&](int n) {
    int idx = 0;
#pragma unroll
    for (int r = 0; r < 16 + SKEW; ++r) {
      if (r < 16) { if (idx == n) ex(2 * r); ++idx; if (idx == n) ex(2 * r + 1); ++idx; }
      if (r >= SKEW) { const int q = r - SKEW; if (idx == n) ad(2 * q); ++idx; if (idx == n) ad(2 * q + 1); ++idx; if (idx == n) pk(q); ++idx; }
    }
  };
  auto step = [&](int n) {
    if (MODE & 96) { step_skew(n); return; }
    if (!(MODE & 1)) { const int e = n & 31; float v = __builtin_fmaf(get(e), c, 0.5f); PIN(v); if (e < 16) sa[e] = v; else sb[e - 16] = v; return; }
    if (n < 2) { ex(n); return; }
    if (n >= 77) { if (n == 77) ad(30); else if (n == 78) ad(31); else pk(15); return; }
    const int r = (n - 2) / 5 + 1, kk = (n - 2) % 5;
    if (kk == 0) ex(2 * r); else if (kk == 1) ex(2 * r + 1); else if (kk == 2) ad(2 * r - 2); else if (kk == 3) ad(2 * r - 1); else pk(r - 1);
  };
  // even order: per pair p: exp a, exp b | add a, add b, pk  -> gaps alternate (2 exp + 1 light) / (2 light) ...
  auto step_even = [&](int n) {      // 80 steps re-ordered so every 10 consecutive steps hold 4 exp + 6 light
    step(n);
  };
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      if (MODE & 256) mf(m, fr2[((m >> 3) & 1)][(m >> 1) & 3], b);
      else if (MODE & 128) mf(m, a, b);
      else mf(m, fr[m & 3], b);
      if ((MODE & 2) && (m & 1)) {
        if (MODE & 256) { if ((m & 7) < 8) fr2[((m >> 3) & 1) ^ 1][(m >> 1) & 3] = *reinterpret_cast<const bf16x8_t*>(lp + ((m * 1024) & 8191)); }
        else fr[(m >> 1) & 3] = *reinterpret_cast<const bf16x8_t*>(lp + ((m * 1024) & 8191));
      }
      if (MODE & 4) {
        // cost-balanced: gap m gets steps so that exp count per gap <= 2
        const int lo = (m * 10) / 3, hi = ((m + 1) * 10) / 3;
#pragma unroll
        for (int n = lo; n < hi; ++n) step_even(n);
      } else {
#pragma unroll
        for (int n = (80 * m + 23) / 24; n < (80 * (m + 1) + 23) / 24; ++n) step(n);
      }
      if (!(MODE & 16)) FENCE();
    }
    for (int r = 0; r < 16; r += 8) { sa[r] = sa[r] * -0.5f - 0.25f; sb[r] = sb[r] * -0.5f - 0.125f; }
  }
  float s = rs[0] + rs[1] + rs[2] + rs[3];
  for (int r = 0; r < 16; ++r) s += sa[r] + sb[r] + __builtin_bit_cast(float, w[r]);
  for (int i = 0; i < 4; ++i) s += (float)fr2[0][i][0] + (float)fr2[1][i][0];
  out[blockIdx.x * 256 + tid] = s + (float)fr[0][0] + (float)fr[1][0] + (float)fr[2][0] + (float)fr[3][0];
}

template <int MODE>
void run(float* out, const char* name) {
  const int iters = 3000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 16384, 0, out, 10, 0.999f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 16384, 0, out, iters, 0.999f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / (iters * 24.0);
  printf("mode %2d %-58s %7.2f ns/MFMA (= %5.1f cyc @2.0 GHz)\n", MODE, name, ns, ns * 2.0);
}

int main() {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  run<0>(out, "80 independent v_fma per 24 MFMA");
  run<1>(out, "softmax steps (32 exp, 32 add, 16 cvt), kernel order");
  run<9>(out, "softmax steps, exp replaced by mul");
  run<3>(out, "softmax steps + 12 ds_read_b128");
  run<2>(out, "independent v_fma + 12 ds_read_b128");
  run<17>(out, "softmax steps, no sched_barrier fences");
  run<19>(out, "softmax steps + ds_read, no fences");
  run<131>(out, "softmax steps + 12 ds_read never consumed by MFMA");
  run<130>(out, "independent v_fma + 12 ds_read never consumed");
  run<259>(out, "softmax steps + 12 ds_read consumed 8 MFMAs later");
  run<258>(out, "independent v_fma + ds_read consumed 8 MFMAs later");
  return 0;
}
