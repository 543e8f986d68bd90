// Reproducer (no code of this library involved): two PROCESSES sharing one MI355X, the same VIRTUAL address in both, different
// contents.  Each process's kernel re-reads its own small table through the vector L1 (plain global_load_dword) and counts words
// that are not its own pattern.  Build: hipcc --offload-arch=gfx950 -O2 -o vmid_l1_alias vmid_l1_alias.hip
// Run:   ./vmid_l1_alias 0 & ./vmid_l1_alias 1 & wait          (same VA in both: the collision case)
//        ./vmid_l1_alias 0 & ./vmid_l1_alias 1 shift & wait    (rank 1 allocates 64 MiB first: different VA, control)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void probe(const unsigned* table, int words, unsigned tag, int iters, unsigned* bad, unsigned* sample) {
  unsigned mism = 0, last = 0;
  int i = threadIdx.x % words;
  for (int it = 0; it < iters; ++it) {
    unsigned v;
    const unsigned* p = table + i;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");   // cacheable load
    if (v != (tag | (unsigned)i)) { ++mism; last = v; }
    i = (i + 97) % words;
  }
  if (mism) { atomicAdd(bad, mism); *sample = last; }
}
int main(int argc, char** argv) {
  const int rank = argc > 1 ? atoi(argv[1]) : 0;
  void* dummy = nullptr;
  if (argc > 2 && !strcmp(argv[2], "shift") && rank == 1) CK(hipMalloc(&dummy, 64 << 20));
  const int words = 512;                                   // 2 KiB: the size of a per-frame cos / sin table
  unsigned *table, *bad, *sample, host[512];
  CK(hipMalloc(&table, words * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&sample, 4));
  const unsigned tag = 0xA0000000u | ((unsigned)rank << 20);
  for (int i = 0; i < words; ++i) host[i] = tag | i;
  CK(hipMemcpy(table, host, sizeof(host), hipMemcpyHostToDevice));
  CK(hipMemset(bad, 0, 4)); CK(hipMemset(sample, 0, 4));
  while (time(nullptr) % 4 != 0) {}                        // both processes start their launches in the same second
  for (int k = 0; k < 4000; ++k) hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, 0, table, words, tag, 2000, bad, sample);
  CK(hipDeviceSynchronize());
  unsigned hb = 0, hs = 0;
  CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hs, sample, 4, hipMemcpyDeviceToHost));
  printf("[vmid_l1_alias] rank %d table VA %p: %u foreign words read (a sample: 0x%08x; own tag 0x%08x)\n", rank, (void*)table, hb, hs, tag);
  return 0;
}
