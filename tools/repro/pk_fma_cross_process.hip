// Stand-alone reproducer (nothing of libactionmesh_amd in it): on MI355X (gfx950, ROCm 7.2) a packed-FP32 VALU instruction
// (v_pk_fma_f32 with operand modifiers, exactly what hipcc's SLP vectoriser emits for a 2-D rotation) returns a WRONG LOW HALF in
// lanes 48-63 while a wave of ANOTHER PROCESS issues bf16 MFMAs on the same SIMD.  Same process, two streams: never.
//   hipcc --offload-arch=gfx950 -O2 -o pk_fma_cross_process pk_fma_cross_process.hip
//   ./pk_fma_cross_process victim                                  # alone: 0 mismatches
//   ./pk_fma_cross_process aggressor & ./pk_fma_cross_process victim; wait      # two processes: mismatches, lanes 48-63, low half
//   ./pk_fma_cross_process both                                    # one process, two streams: 0 mismatches
// The victim checks v_pk_fma_f32 against two scalar v_fma_f32 on the same registers (an FMA is exact: bit-equal or broken).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16 __attribute__((ext_vector_type(16)));
// FORM: 0 = the form hipcc emitted for the rotation (op_sel_hi + neg on src2); 1 = neg only; 2 = op_sel_hi only; 3 = plain;
//       4 = v_pk_mul_f32 op_sel_hi:[0,1]; 5 = plain v_pk_add_f32
template <int FORM>
__global__ void victim(unsigned* bad, int iters) {            // bad[quarter * 2 + half]
  const int lane = threadIdx.x & 63;
  float a = 0.37f + 0.01f * lane, b = 1.3f - 0.02f * lane;
  for (int it = 0; it < iters; ++it) {
    f2 x = {a, b}, y = {b * 0.5f, a}, z = {a * b, b - a}, r;
    float e0, e1;
    if (FORM == 0) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
      asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(e0) : "v"(x.x), "v"(y.x), "v"(z.x));      // low  = x.lo * y.lo - z.lo
      asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(e1) : "v"(x.x), "v"(y.y), "v"(z.y));      // high = x.lo * y.hi - z.hi
    } else if (FORM == 1) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
      asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(e0) : "v"(x.x), "v"(y.x), "v"(z.x));
      asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(e1) : "v"(x.y), "v"(y.y), "v"(z.y));
    } else if (FORM == 2) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(x.x), "v"(y.x), "v"(z.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(x.x), "v"(y.y), "v"(z.y));
    } else if (FORM == 3) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(x.x), "v"(y.x), "v"(z.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(x.y), "v"(y.y), "v"(z.y));
    } else if (FORM == 4) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(x.x), "v"(y.x));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(x.x), "v"(y.y));
    } else {
      asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(x.x), "v"(y.x));
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(x.y), "v"(y.y));
    }
    if (__float_as_uint(r.x) != __float_as_uint(e0)) atomicAdd(&bad[(lane >> 4) * 2], 1u);
    if (__float_as_uint(r.y) != __float_as_uint(e1)) atomicAdd(&bad[(lane >> 4) * 2 + 1], 1u);
    a += 1e-3f; b -= 1e-3f;
  }
}
__global__ void aggressor(float* sink, int iters) {
  bf8 p, q;
  for (int i = 0; i < 8; ++i) { p[i] = (__bf16)(0.01f * (threadIdx.x + i)); q[i] = (__bf16)(0.5f - 0.001f * (threadIdx.x + i)); }
  f16 acc[4] = {};
  for (int it = 0; it < iters; ++it)
    for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, acc[k], 0, 0, 0);
  if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.f) sink[0] = 1.f;
}
int main(int argc, char** argv) {
  const char* role = argc > 1 ? argv[1] : "victim";
  const double secs = argc > 2 ? atof(argv[2]) : 6.0;
  unsigned* bad; float* sink; unsigned h[8];
  hipMalloc(&bad, 32); hipMemset(bad, 0, 32); hipMalloc(&sink, 4);
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  const bool v = strcmp(role, "aggressor") != 0, g = strcmp(role, "victim") != 0;
  const char* names[6] = {"v_pk_fma_f32 op_sel_hi:[0,1,1] neg_lo/hi:[0,0,1]", "v_pk_fma_f32 neg_lo/hi:[0,0,1]", "v_pk_fma_f32 op_sel_hi:[0,1,1]",
                          "v_pk_fma_f32", "v_pk_mul_f32 op_sel_hi:[0,1]", "v_pk_add_f32"};
  for (int form = 0; form < (v ? 6 : 1); ++form) {
    hipMemset(bad, 0, 32);
    const time_t t0 = time(nullptr); long launches = 0;
    while (difftime(time(nullptr), t0) < (v ? secs : 6 * secs + 2)) {
      for (int k = 0; k < 20; ++k) {
        if (g) hipLaunchKernelGGL(aggressor, dim3(1024), dim3(256), 0, s2, sink, 3000);
        if (v) switch (form) {
          case 0: hipLaunchKernelGGL(victim<0>, dim3(2048), dim3(256), 0, s1, bad, 500); break;
          case 1: hipLaunchKernelGGL(victim<1>, dim3(2048), dim3(256), 0, s1, bad, 500); break;
          case 2: hipLaunchKernelGGL(victim<2>, dim3(2048), dim3(256), 0, s1, bad, 500); break;
          case 3: hipLaunchKernelGGL(victim<3>, dim3(2048), dim3(256), 0, s1, bad, 500); break;
          case 4: hipLaunchKernelGGL(victim<4>, dim3(2048), dim3(256), 0, s1, bad, 500); break;
          default: hipLaunchKernelGGL(victim<5>, dim3(2048), dim3(256), 0, s1, bad, 500); break;
        }
        ++launches;
      }
      hipDeviceSynchronize();
    }
    hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost);
    if (v) printf("[pk_fma] role %s, %-50s %6ld launches: mismatches low / high half by lane quarter  0-15: %u / %u  16-31: %u / %u  32-47: %u / %u  "
                  "48-63: %u / %u\n", role, names[form], launches, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    else printf("[pk_fma] aggressor done (%ld launches)\n", launches);
  }
  return 0;
}
