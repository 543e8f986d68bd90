// Minimal form (ALU only - no loads in the loop, no LDS, nothing of libactionmesh_amd): a packed-FP32 multiply with a "swapped"
// operand selection, exactly as hipcc's SLP vectoriser emits it for a 2-D rotation, checked against two scalar v_mul_f32 on the same
// registers.  On MI355X (gfx950, ROCm 7.2) the LOW half of its result is wrong in lanes 48-63 while ANOTHER PROCESS runs bf16 GEMMs
// on the device (tools/repro/run.sh); never alone.
//   hipcc --offload-arch=gfx950 -O2 -o pk_mul pk_mul_cross_process.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
typedef float f2 __attribute__((ext_vector_type(2)));
// FORM 0: v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,0]   lo = a.lo * b.hi, hi = a.lo * b.lo     (the form that fails in head_post)
// FORM 1: v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[1,0]   lo = a.hi * b.hi, hi = a.hi * b.lo
// FORM 2: v_pk_mul_f32 (default)                      lo = a.lo * b.lo, hi = a.hi * b.hi
template <int FORM>
__global__ void victim(unsigned* bad, int iters) {            // bad[quarter * 2 + half]
  const int lane = threadIdx.x & 63;
  float a0 = 0.37f + 0.01f * lane, a1 = 1.3f - 0.02f * lane;
  for (int it = 0; it < iters; ++it) {
    f2 a = {a0, a1}, b = {a1 * 0.5f, a0 + 0.25f}, r;
    float e0, e1;
    if (FORM == 0) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(a.x), "v"(b.y));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(a.x), "v"(b.x));
    } else if (FORM == 1) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(a.y), "v"(b.y));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(a.y), "v"(b.x));
    } else {
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(a.x), "v"(b.x));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(a.y), "v"(b.y));
    }
    if (__float_as_uint(r.x) != __float_as_uint(e0)) atomicAdd(&bad[(lane >> 4) * 2], 1u);
    if (__float_as_uint(r.y) != __float_as_uint(e1)) atomicAdd(&bad[(lane >> 4) * 2 + 1], 1u);
    a0 += 1e-3f; a1 -= 1e-3f;
  }
}
int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 5.0;
  unsigned* bad; unsigned h[8];
  hipMalloc(&bad, 32);
  const char* names[3] = {"op_sel:[0,1] op_sel_hi:[0,0]", "op_sel:[1,1] op_sel_hi:[1,0]", "default operand selection"};
  for (int form = 0; form < 3; ++form) {
    hipMemset(bad, 0, 32);
    const time_t t0 = time(nullptr); long launches = 0;
    while (difftime(time(nullptr), t0) < secs) {
      for (int k = 0; k < 50; ++k, ++launches) {
        if (form == 0) hipLaunchKernelGGL(victim<0>, dim3(2048), dim3(256), 0, 0, bad, 500);
        else if (form == 1) hipLaunchKernelGGL(victim<1>, dim3(2048), dim3(256), 0, 0, bad, 500);
        else hipLaunchKernelGGL(victim<2>, dim3(2048), dim3(256), 0, 0, bad, 500);
      }
      hipDeviceSynchronize();
    }
    hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost);
    printf("[pk_mul] v_pk_mul_f32 %-30s %6ld launches: wrong low / high halves by lane quarter  0-15: %u / %u  16-31: %u / %u  32-47: %u / %u  48-63: %u / %u\n",
           names[form], launches, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  }
  return 0;
}
