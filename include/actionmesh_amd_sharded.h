/* actionmesh_amd: the frame-sharded forward's per-layer phase loop in C, for the copy-engine exchange back-end.
 *
 * include/actionmesh_amd.h gives a caller the phases (am_forward_begin, am_layer_pre_attn, am_layer_attn_local, am_layer_post_attn,
 * am_forward_end) and the exchange primitives (am_peer_copy / am_peer_signal / am_peer_wait); actionmesh_amd/sharding.py strings them
 * together from Python.  This entry point is that loop as ONE C call: what a C / C++ host of the reference's sampler
 * (scheduler.py:139-170 -> ActionMeshDenoiser.forward, temporal_denoiser.py:219-236 with the K/V exchange of SURVEY K15 behind
 * every inflated layer) binds when the ranks exchange their K / V^T shards through the copy engines.  Host-only code: it launches
 * no kernel of its own (actionmesh_amd/csrc/host/am_phase_loop.hip).
 */
#ifndef ACTIONMESH_AMD_SHARDED_H
#define ACTIONMESH_AMD_SHARDED_H
#include "actionmesh_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

#define AM_PEER_MAX_RANKS 16

/* One rank's view of the exchange ring (sharding.PeerExchange): its own gather buffer [world][chunk_bytes] and flag block
 * (uint32: arrived[world] | consumed[world] | fault), the peers' buffers and flag blocks as opened with am_peer_open (entry `rank`
 * unused), a side stream the pushes run on, and the sequence number of the last exchange (in / out: it keeps turning across calls
 * and is shared with a Python-driven exchange on the same ring). */
typedef struct {
  int32_t world, rank;
  uint64_t chunk_bytes;
  void* kv;
  void* flags;
  void* peer_kv[AM_PEER_MAX_RANKS];
  void* peer_flags[AM_PEER_MAX_RANKS];
  void* side_stream;
  uint32_t seq;
  /* owned by the ring: two hipEvent_t (fork: compute -> side stream; pushed: side -> compute), created by the first
   * am_forward_sharded_peer on the ring's device, destroyed by am_peer_ring_destroy.  Zero-initialise. */
  void* ev_fork;
  void* ev_pushed;
} am_peer_ring;

/* The whole sharded forward of this rank: begin; per layer pre-attention, then for an inflated layer the exchange - pushes of the
 * local shard to every peer on ring->side_stream (each behind the peer's `consumed` flag of the previous exchange), the attention
 * against the LOCAL shard meanwhile, the wait for every peer's `arrived` flag, the rest of the layer, the `consumed` signals - and
 * the epilogue into v_out (B, T_local, N, Din) bf16.  Same launches in the same order as sharding.sharded_forward(exchange=...):
 * bit-identical results.  `inflated[i] != 0`: layer i attends over all frames (am_config.inflated_mask).  The handle must have been
 * created with world_size = ring->world > 1 and bound to ring->kv (am_bind_kv_buffers / am_bind_kv8_buffers). */
int am_forward_sharded_peer(am_handle h, const float* x_dev, const float* t_bt_host, int B, int T_local, int N, uint16_t* v_out,
                            am_peer_ring* ring, const uint8_t* inflated, int num_layers, void* stream);

/* Destroys the events the ring owns (before its buffers are freed / the struct is dropped).  The ring may be used again afterwards. */
int am_peer_ring_destroy(am_peer_ring* ring);

#ifdef __cplusplus
}
#endif
#endif
