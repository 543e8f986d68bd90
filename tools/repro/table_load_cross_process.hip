// Stand-alone reproducer, second form (nothing of libactionmesh_amd in it): every 16-lane group of a wave loads the SAME 256 bytes of a
// small, hot table with one global_load_dwordx4 per lane (what a per-frame cos / sin lookup looks like: 4 tokens of a frame per wave)
// and uses the result at once.  The kernel checks every loaded word against the table's known pattern.  On MI355X / ROCm 7.2 the
// check fails - lanes 48-63, dwords 0 and 2 of the four - while ANOTHER PROCESS runs bf16 GEMMs on the same GPU; never alone, never
// with the same GEMMs on a second stream of the same process.
//   hipcc --offload-arch=gfx950 -O2 -o table_load_cross_process table_load_cross_process.hip
//   ./table_load_cross_process [seconds]            # alone
//   python -c "import torch,time; a=torch.randn(4096,256,device='cuda').bfloat16(); b=a.t().contiguous(); t=time.time();
//   exec('while time.time()-t<20:\n for _ in range(200): a@b\n torch.cuda.synchronize()')" & sleep 6; ./table_load_cross_process
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
typedef float f4 __attribute__((ext_vector_type(4)));
// table[row][64] floats, row = "frame"; word value = row * 1000 + column.  bad[quarter * 4 + dword]
__global__ void victim(const float* table, int rows, const float* x, float* out, unsigned* bad) {
  const int lane = threadIdx.x & 63, sub = threadIdx.x & 15;
  const int token = blockIdx.x * 16 + (threadIdx.x >> 4);           // 16 tokens per 256-thread block, 4 per wave
  for (int pass = 0; pass < 4; ++pass) {
    const int row = ((token + pass * 16 * gridDim.x) / 37) % rows;  // a "frame" = 37 tokens: groups of a wave sometimes straddle two rows
    const f4 xv = *reinterpret_cast<const f4*>(x + ((size_t)token * 4 + pass) * 64 + sub * 4);     // streaming operand first
    const f4 t = *reinterpret_cast<const f4*>(table + row * 64 + sub * 4);                           // hot table, duplicate addresses
    f4 r = {t[0] * xv[1], t[1] * xv[0], t[2] * xv[3], t[3] * xv[2]};                                   // used at once
    for (int d = 0; d < 4; ++d)
      if (t[d] != (float)(row * 1000 + sub * 4 + d)) atomicAdd(&bad[(lane >> 4) * 4 + d], 1u);
    *reinterpret_cast<f4*>(out + ((size_t)token * 4 + pass) * 64 + sub * 4) = r;
  }
}
int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 6.0;
  const int rows = 8, blocks = 1024;
  float *table, *x, *out; unsigned* bad; unsigned h[16];
  hipMalloc(&table, rows * 64 * 4); hipMalloc(&x, (size_t)blocks * 16 * 4 * 64 * 4); hipMalloc(&out, (size_t)blocks * 16 * 4 * 64 * 4);
  hipMalloc(&bad, 64); hipMemset(bad, 0, 64); hipMemset(x, 0, (size_t)blocks * 16 * 4 * 64 * 4);
  float ht[8 * 64];
  for (int r = 0; r < rows; ++r) for (int c = 0; c < 64; ++c) ht[r * 64 + c] = (float)(r * 1000 + c);
  hipMemcpy(table, ht, sizeof(ht), hipMemcpyHostToDevice);
  const time_t t0 = time(nullptr); long launches = 0;
  while (difftime(time(nullptr), t0) < secs) {
    for (int k = 0; k < 50; ++k, ++launches) hipLaunchKernelGGL(victim, dim3(blocks), dim3(256), 0, 0, table, rows, x, out, bad);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost);
  printf("[table_load] %ld launches; wrong table words by lane quarter (dword 0 / 1 / 2 / 3):", launches);
  for (int q = 0; q < 4; ++q) printf("  %d-%d: %u / %u / %u / %u", q * 16, q * 16 + 15, h[q * 4], h[q * 4 + 1], h[q * 4 + 2], h[q * 4 + 3]);
  printf("\n");
  return 0;
}
