// Copy-engine exchange of the [K | V^T] shards between the ranks of one node (one process per GPU).
//
// The RCCL all-gather of sharding.py runs as kernels that need CUs; the inflated self-attention launches one workgroup per CU
// (128 KiB of LDS, 512 registers per lane), so the "overlapped" collective may simply queue behind the attention grid
// (VERDICT r01 weak #6).  An all-gather on a full xGMI mesh is P-1 independent pushes: each rank copies its shard into every
// peer's gather buffer with the SDMA engines (hipMemcpyAsync between IPC-mapped device buffers on a side stream: no CU), then
// raises a sequence flag in the peer's memory; the peer's compute stream waits for the flags of the shards it is about to
// read.  The flag write / wait are one-lane kernels (system-scope release store / polled system-scope loads + acquire): the
// only CU work of the exchange.  Buffers come from hipMalloc directly (IPC handles cannot be taken on sub-allocations of a
// caching allocator).  Protocol and stream wiring: actionmesh_amd/sharding.py PeerExchange.
#include <stdlib.h>
#include <string.h>

#include "am_common.h"

namespace {

__global__ void peer_signal_kernel(uint32_t* flag, uint32_t value) {
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Bounded spin (MI355X_MICROARCH.md "bound every spin"): gives up after ~20 s of wall clock and reports through fault_word
// instead of hanging the box; relaxed polls, ONE acquire at the end.
__global__ void peer_wait_kernel(const uint32_t* flag, uint32_t value, uint32_t* fault_word) {
  // a wait that gave up has already invalidated the forward: every later wait returns at once instead of spinning its own 20 s
  // (7 peers x 21 layers x the steps of a run would otherwise hold the device for hours behind one dead peer)
  if (fault_word && __hip_atomic_load(fault_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
  const uint64_t t0 = wall_clock64();
  while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
    __builtin_amdgcn_s_sleep(8);
    if (wall_clock64() - t0 > 2000000000ull) {          // 100 MHz constant clock: 20 s
      if (fault_word) __hip_atomic_store(fault_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

}  // namespace

extern "C" int am_peer_alloc(size_t bytes, void** out) {
  AM_CHECK(out != nullptr && bytes > 0, "am_peer_alloc: bad argument");
  AM_HIP(hipMalloc(out, bytes));
  AM_HIP(hipMemset(*out, 0, bytes));
  AM_HIP(hipDeviceSynchronize());
  return AM_OK;
}
// The FLAG block (arrived[P] | consumed[P] | fault word) is polled by a running kernel of the owning device while ANOTHER device writes it
// over xGMI.  System-scope atomics are only specified to be visible across agents on fine-grained allocations: in a coarse-grained
// hipMalloc block the poll may be served from the owner's L2 and never see the peer's store (same-device runs cannot expose this;
// VERDICT r04 weak #5a).  So the flags come from hipExtMallocWithFlags(hipDeviceMallocFinegrained) - device-local, cache-coherent
// across agents - while the shards stay coarse-grained (they are only read behind a flag's acquire, at kernel boundaries).
// *fine_grained reports what was obtained: 1, or 0 when the runtime refused and the block is an ordinary hipMalloc (the caller
// decides whether that is acceptable; bench.py prints it).  ACTIONMESH_AMD_PEER_COARSE_FLAGS=1 forces the old behaviour (A/B).
extern "C" int am_peer_alloc_flags(size_t bytes, void** out, int* fine_grained) {
  AM_CHECK(out != nullptr && bytes > 0, "am_peer_alloc_flags: bad argument");
  const char* coarse = getenv("ACTIONMESH_AMD_PEER_COARSE_FLAGS");
  int fine = 0;
  *out = nullptr;
  if (!(coarse && coarse[0] == '1')) {
    if (hipExtMallocWithFlags(out, bytes, hipDeviceMallocFinegrained) == hipSuccess && *out) fine = 1;
    else { (void)hipGetLastError(); *out = nullptr; }
  }
  if (!fine) AM_HIP(hipMalloc(out, bytes));
  AM_HIP(hipMemset(*out, 0, bytes));
  AM_HIP(hipDeviceSynchronize());
  if (fine_grained) *fine_grained = fine;
  return AM_OK;
}
extern "C" int am_peer_free(void* p) {
  if (p) AM_HIP(hipFree(p));
  return AM_OK;
}
extern "C" int am_peer_export(void* p, uint8_t* handle64) {
  AM_CHECK(p && handle64, "am_peer_export: null argument");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  hipIpcMemHandle_t h;
  AM_HIP(hipIpcGetMemHandle(&h, p));
  memcpy(handle64, &h, 64);
  return AM_OK;
}
extern "C" int am_peer_open(const uint8_t* handle64, void** out) {
  AM_CHECK(handle64 && out, "am_peer_open: null argument");
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  AM_HIP(hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess));
  return AM_OK;
}
extern "C" int am_peer_close(void* p) {
  if (p) AM_HIP(hipIpcCloseMemHandle(p));
  return AM_OK;
}
extern "C" int am_peer_copy(void* dst, const void* src, size_t bytes, void* stream) {
  AM_CHECK(dst && src && bytes > 0, "am_peer_copy: bad argument");
  AM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, (hipStream_t)stream));
  return AM_OK;
}
extern "C" int am_peer_signal(uint32_t* flag, uint32_t value, void* stream) {
  AM_CHECK(flag != nullptr, "am_peer_signal: null flag");
  hipLaunchKernelGGL(peer_signal_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, value);
  AM_HIP(hipGetLastError());
  return AM_OK;
}
extern "C" int am_peer_wait(const uint32_t* flag, uint32_t value, uint32_t* fault_word, void* stream) {
  AM_CHECK(flag != nullptr, "am_peer_wait: null flag");
  hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, value, fault_word);
  AM_HIP(hipGetLastError());
  return AM_OK;
}
