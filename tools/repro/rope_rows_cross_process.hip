// Stand-alone form of the r03 same-device divergence (no libactionmesh_amd linked; the kernel below is the round-2 Q/K path of
// head_post_kernel, plain C++ as hipcc compiles it - no inline asm, no LDS-DMA).  Per token (16 lanes x 8 channels): RMS-normalise
// one 128-channel head, rotate channel pairs by the token's frame angle (cos / sin rows fetched INSIDE the pass, two
// global_load_dwordx4 whose results are consumed straight behind the s_waitcnt), round to bf16.  The host repeats the launch
// and counts output words that differ from the first launch, by lane quarter and channel-within-lane.
//   hipcc --offload-arch=gfx950 -O3 -o rope_rows_cross_process rope_rows_cross_process.hip      (variants: -fno-slp-vectorize, -DDELAY)
//   ./rope_rows_cross_process 5                                  # alone: 0 differing words
//   python -c "import torch,time; a=torch.randn(4096,256,device='cuda').bfloat16(); b=a.t().contiguous(); t=time.time()
//   while time.time()-t<25:
//       [a@b for _ in range(200)]; torch.cuda.synchronize()" &  sleep 8; ./rope_rows_cross_process 8
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <vector>
typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ inline float bflo(uint32_t w) { union { uint32_t u; float f; } c; c.u = w << 16; return c.f; }
__device__ inline float bfhi(uint32_t w) { union { uint32_t u; float f; } c; c.u = w & 0xffff0000u; return c.f; }
__device__ inline uint32_t pack_bf2(float lo, float hi) { const f32x2_t f = {lo, hi}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t)); }
// X [rows][ldx] bf16, head h part p at columns (h * nparts + p) * 128; out [seq][heads][seq_len][128]
__global__ __launch_bounds__(256) void rope_rows(const bf16_t* X, int ldx, int seq_len, int rows_per_frame, int heads, int nparts, const float* w,
                                                 float eps, const float* rope_cos, const float* rope_sin, bf16_t* out, int blocks_per_seq) {
  const int tid = threadIdx.x, sblk = blockIdx.x % blocks_per_seq, sidx = blockIdx.x / blocks_per_seq, head = blockIdx.y, part = blockIdx.z;
  const int s0 = sblk * 64, sub = tid & 15, tok_in_pass = tid >> 4;
  const int col = (head * nparts + part) * 128 + sub * 8;
  float wv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) wv[e] = w ? w[sub * 8 + e] : 1.f;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int s = s0 + pass * 16 + tok_in_pass;
    if (s >= seq_len) continue;
    const int64_t row = (int64_t)sidx * seq_len + s;
    const u32x4_t u = *reinterpret_cast<const u32x4_t*>(X + row * ldx + col);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = bflo(u[e]); v[2 * e + 1] = bfhi(u[e]); }
    if (w) {
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
      const float r = rsqrtf(ss * (1.0f / 128.0f) + eps);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * r * wv[e];
    }
    if (rope_cos) {          // (a pointer test, as in the library: it keeps the compiler from hoisting the table loads above the norm)
      const int64_t frame = row / rows_per_frame;
      f32x4_t cs = *reinterpret_cast<const f32x4_t*>(rope_cos + frame * 64 + sub * 4);      // the 16 lanes of a token read 256 B;
      f32x4_t sn = *reinterpret_cast<const f32x4_t*>(rope_sin + frame * 64 + sub * 4);      // the wave's 4 tokens the SAME 256 B
#ifdef DELAY      // -DDELAY: both loads waited for, then ~64 idle cycles before the first use - the mismatches disappear
      asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(cs), "+v"(sn));
#endif
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = v[2 * e], bb = v[2 * e + 1];
        v[2 * e] = a * cs[e] + (-bb) * sn[e];
        v[2 * e + 1] = bb * cs[e] + a * sn[e];
      }
    }
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
    *reinterpret_cast<u32x4_t*>(out + ((((int64_t)sidx * heads + head) * nparts + part) * seq_len + s) * 128 + sub * 8) = o;
  }
}
// bad[quarter * 8 + channel-within-lane]: words of `out` that differ from `ref`
__global__ void compare(const bf16_t* out, const bf16_t* ref, size_t n, unsigned* bad) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (out[i] != ref[i]) {
      const int token = (int)((i / 128) % 2048), ch = (int)(i % 128);
      atomicAdd(&bad[((token & 3)) * 8 + (ch & 7)], 1u);          // token & 3 = the 16-lane quarter of the wave that wrote it
    }
}
int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 6.0;
  const int B = 2, T = 4, L = 512, H = 2, NP = 3, seq_len = T * L, rows = B * seq_len, ldx = H * NP * 128;
  std::vector<uint16_t> hx((size_t)rows * ldx);
  srand(5);
  for (auto& x : hx) { union { float f; uint32_t u; } c; c.f = (rand() / (float)RAND_MAX - 0.5f) * 4.f; x = (uint16_t)(c.u >> 16); }
  std::vector<float> hc(B * T * 64), hs(B * T * 64), hw(128, 1.0f);
  for (int f = 0; f < B * T; ++f) for (int i = 0; i < 64; ++i) { const float a = (f % T) * powf(10000.f, -2.f * i / 128.f); hc[f * 64 + i] = cosf(a); hs[f * 64 + i] = sinf(a); }
  bf16_t *X, *out, *ref; float *c, *s, *w; unsigned* bad;
  const size_t n_out = (size_t)B * H * NP * seq_len * 128;
  hipMalloc(&X, hx.size() * 2); hipMalloc(&out, n_out * 2); hipMalloc(&ref, n_out * 2); hipMalloc(&c, hc.size() * 4); hipMalloc(&s, hs.size() * 4);
  hipMalloc(&w, 512); hipMalloc(&bad, 128); hipMemset(bad, 0, 128);
  hipMemcpy(X, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(c, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(s, hs.data(), hs.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), 512, hipMemcpyHostToDevice);
  const dim3 grid(B * (seq_len / 64), H, NP);
  hipLaunchKernelGGL(rope_rows, grid, dim3(256), 0, 0, X, ldx, seq_len, L, H, NP, w, 1e-6f, c, s, ref, seq_len / 64);
  hipDeviceSynchronize();
  const time_t t0 = time(nullptr); long launches = 0;
  while (difftime(time(nullptr), t0) < secs) {
    for (int k = 0; k < 50; ++k, ++launches) {
      hipLaunchKernelGGL(rope_rows, grid, dim3(256), 0, 0, X, ldx, seq_len, L, H, NP, w, 1e-6f, c, s, out, seq_len / 64);
      hipLaunchKernelGGL(compare, dim3(512), dim3(256), 0, 0, out, ref, n_out, bad);
    }
    hipDeviceSynchronize();
  }
  unsigned h[32]; hipMemcpy(h, bad, 128, hipMemcpyDeviceToHost);
  unsigned tot = 0; for (unsigned v : h) tot += v;
  printf("[rope_rows] %ld launches, %u differing output words; by lane quarter (channel-within-lane 0..7):\n", launches, tot);
  for (int q = 0; q < 4; ++q) { printf("   lanes %2d-%2d:", q * 16, q * 16 + 15); for (int e = 0; e < 8; ++e) printf(" %u", h[q * 8 + e]); printf("\n"); }
  return 0;
}
