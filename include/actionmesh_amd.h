/*
 * actionmesh_amd.h - C ABI of libactionmesh_amd.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for ONE hot path of facebookresearch/actionmesh: the Stage-I
 * temporal-3D flow-matching denoise loop.  The reference has no FFI (it is pure
 * Python); its plug-in seams for this path are:
 *
 *   S1  actionmesh/scheduler/scheduler.py:252-295   SchedulerFlow.denoise
 *   S2  actionmesh/model/temporal_denoiser.py:151-249 ActionMeshDenoiser.forward
 *   S3  actionmesh/model/utils/attention_processor.py:36-168 AttentionProcessor.__call__
 *
 * Each entry point below names the reference interface it replaces.  The
 * reference-side binding (a ctypes stub) is shown in INTEGRATION.md; the
 * in-tree binding is actionmesh_amd/_lib.py.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.
 *   - every `*_dev` pointer is device memory valid on `stream` (a hipStream_t
 *     passed as void*; NULL = the null stream).  Pointers are borrowed for the
 *     duration of the call; nothing synchronises the device unless stated.
 *   - bf16 tensors are raw uint16 payloads, row-major.
 *   - return 0 on success, negative am_status otherwise; am_last_error() gives
 *     the message of the calling thread's last failure.
 *   - one handle per device per rank; a handle is not thread-safe (the
 *     reference drives this path from a single Python thread).
 */
#ifndef ACTIONMESH_AMD_H
#define ACTIONMESH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  AM_OK = 0,
  AM_ERR_INVALID = -1,   /* bad argument / shape */
  AM_ERR_HIP = -2,       /* a HIP runtime call failed */
  AM_ERR_STATE = -3,     /* call sequence violated (e.g. forward before weights) */
  AM_ERR_NOTFOUND = -4   /* unknown weight name */
} am_status;

const char* am_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int am_abi_version(void);

/* ------------------------------------------------------------------------ */
/* Model-level API                                                            */
/* ------------------------------------------------------------------------ */

/* Hyper-parameters of ActionMeshDenoiser (temporal_denoiser.py:29-48) plus the
 * static problem bounds the workspace is sized for. head_dim must be 128. */
typedef struct {
  int32_t in_channels;          /* Din (64) */
  int32_t num_layers;           /* NL (21) */
  int32_t num_heads;            /* H */
  int32_t width;                /* C = H*128 */
  int32_t ff_inner;             /* F = int(C*mlp_ratio) */
  int32_t cross_dim;            /* Dc (1024) */
  uint32_t inflated_mask_lo;    /* bit i set => layer i uses inflated (joint T*L) self-attention */
  uint32_t inflated_mask_hi;    /* layers 32..63 */
  int32_t max_batch;            /* CFG batch B (2) */
  int32_t max_frames_local;     /* frames owned by this rank (T / world) */
  int32_t max_tokens;           /* N latent tokens per frame */
  int32_t max_ctx_tokens;       /* S context tokens per frame */
  int32_t world_size;           /* frame-shard degree P (1 = single GPU) */
  int32_t rank;                 /* this rank's shard index */
  int32_t attn_defer_log2;      /* online-softmax deferred-rescale threshold (log2 units); 0 = always rescale */
  int32_t attn_fp8;             /* 1 = inflated self-attention on the fp8 kernel (am_attention_fp8); 0 = bf16 (default); 2 = fp8 with the
                                   exponent-field form of the probabilities (no v_exp: p = 2^n (1 + f), am_attention_fp8 defer_log2 = 5400) */
  int32_t reserved[6];
} am_config;

typedef struct am_model* am_handle;

int am_create(const am_config* cfg, am_handle* out);
int am_destroy(am_handle h);

/* Replaces PyTorchModelHubMixin.from_pretrained / load_state_dict for
 * ActionMeshDenoiser (pipeline.py:180-184).  `name` is a reference state-dict
 * key; `host_f32` is the fp32 tensor, row-major, `numel`
 * elements.  The library converts / fuses (q|k|v concatenated) and owns its copy. */
int am_load_weight(am_handle h, const char* name, const float* host_f32, size_t numel);
/* Number of reference state-dict keys still missing (0 = ready). */
int am_weights_missing(am_handle h);

/* Step-invariant conditioning for one window (replaces the per-step
 * to_k/to_v(context) of attention_processor.py:102-103 with a cache; and
 * precompute_freqs_rot, temporal_denoiser.py:114-149).
 *   ctx_dev       fp32 (B, T_local, S, Dc) device
 *   rope_cos/sin  fp32 host (B*T_local, 64): cos/sin of position*inv_freq per frame
 */
int am_set_context(am_handle h, const float* ctx_dev, int B, int T_local, int S,
                   const float* rope_cos_host, const float* rope_sin_host, void* stream);

/* Optional, after am_set_context: two EXACT shortcuts of the CFG batch (bit-identical results; the algorithmic flop count
 * of am_step_flops is unchanged and is what the bench reports against).  Cleared by the next am_set_context.
 *   ctx_is_zero_host[b] != 0: the context of batch row b is identically zero (the unconditional guidance branch,
 *       guidance.py:38-93).  to_k / to_v have no bias, so K = V = 0 and the cross-attention output of that row is exactly
 *       to_out[0].bias: the row's cross-attention branch (norm_x_attn, to_q, head split, SDPA, to_out) is
 *       replaced by h += bias, and its K/V cache is not built.  NULL = no row is zero.
 *   shared_prefix != 0: the caller asserts that every batch row carries the same hidden_states and the same t_bt (the sampler
 *       expands ONE latent tensor over the guidance branches, scheduler.py:215-217).  The rows then differ only from the first
 *       cross-attention on, so layer 0's skip-free prefix (norm_s_attn, QKV, qk-norm, RoPE, self-attention, to_out + residual)
 *       is computed for row 0 and copied to the other rows.  Ignored when world_size > 1. */
int am_set_branch_hints(am_handle h, const uint8_t* ctx_is_zero_host, int shared_prefix);

/* ActionMeshDenoiser.forward (temporal_denoiser.py:151-249), CFG-batched.
 *   x_dev     fp32 (B, T_local, N, Din)
 *   t_bt_host fp32 (B*T_local): per-(b,t) diffusion time AFTER the mask
 *             (temporal_denoiser.py:209-212)
 *   v_out_dev bf16 (B, T_local, N, Din)
 * Single-rank convenience = begin + for each layer {pre, post} + end. */
int am_denoise_forward(am_handle h, const float* x_dev, const float* t_bt_host,
                       int B, int T_local, int N, uint16_t* v_out_dev, void* stream);

/* am_denoise_forward through a HIP graph: the forward's ~450 launches are captured once per (x_dev, v_out_dev, shape, stream,
 * bound window) and replayed; the per-frame times are uploaded in front of every launch.  First call with a new key runs eagerly,
 * the second captures, later ones replay; a failed capture turns the path off for the handle (eager from then on).  The caller
 * keeps x_dev / v_out_dev at fixed addresses to benefit; `stream` must not be the null stream (falls back to eager).
 * am_graph_stats: counts4 = {replays, captures, eager forwards, capture failed}. */
int am_denoise_forward_graph(am_handle h, const float* x_dev, const float* t_bt_host,
                             int B, int T_local, int N, uint16_t* v_out_dev, void* stream);
int am_graph_stats(am_handle h, uint64_t* counts4);

/* The same forward, split at the temporal-attention boundary so the host can
 * run the K/V all-gather (RCCL) between `pre` and `post` of an inflated layer. */
int am_forward_begin(am_handle h, const float* x_dev, const float* t_bt_host,
                     int B, int T_local, int N, void* stream);
int am_layer_pre_attn(am_handle h, int layer, void* stream);   /* skip+LN+QKV+qk-norm+RoPE -> local K/V shard */
int am_layer_post_attn(am_handle h, int layer, void* stream);  /* self-attn .. FFN */
int am_forward_end(am_handle h, uint16_t* v_out_dev, void* stream);
/* Optional, multi-GPU: between pre_attn and post_attn of a layer, while the K/V all-gather is in flight, attend to
 * the LOCAL shard only (two-pass self-attention, am_attn_args.state_mode); post_attn then resumes over the remote
 * shards.  A no-op when the layer or the shapes do not qualify. */
int am_layer_attn_local(am_handle h, int layer, void* stream);

/* K / V^T gather buffers for inflated self-attention, laid out
 * [world][B][H][sk_pad][128] (K) and [world][B][H][128][sk_pad] (V^T); this
 * rank writes chunk `rank`.  By default the library owns them; a multi-GPU
 * host binds its own (torch-allocated) buffers so it can all-gather in place.
 * `chunk_stride_elems` = distance between consecutive ranks' chunks (0 = elems_per_chunk, i.e. two dense
 * arrays); 2 * elems_per_chunk with vt_dev = k_dev + elems_per_chunk interleaves [rank][K | V^T] so that ONE
 * all-gather per layer moves both operands. */
int am_kv_chunk_elems(am_handle h, size_t* elems_per_chunk);
int am_bind_kv_buffers(am_handle h, uint16_t* k_dev, uint16_t* vt_dev, size_t chunk_stride_elems);

/* fp8 handles (am_config.attn_fp8) with world_size > 1 exchange the QUANTISED shards: am_layer_pre_attn quantises this rank's K / V^T
 * shard into chunk `rank` of these buffers ([world][B][H][sk_pad][128] / [world][B][H][128][sk_pad] bytes; one byte per element,
 * chunk stride in bytes, 0 = elems_per_chunk), the host all-gathers them (half the bytes of the bf16 exchange), and the fp8
 * two-pass attention of am_layer_attn_local / am_layer_post_attn reads them.  Must be bound before the first forward; binding the
 * bf16 buffers on such a handle is an error.  With world_size = 1 the library owns its fp8 copies. */
int am_bind_kv8_buffers(am_handle h, uint8_t* k8_dev, uint8_t* vt8_dev, size_t chunk_stride_bytes);
/* counts2 = {fp8, bf16} inflated self-attention launches of the handle so far (a two-pass layer counts once). */
int am_attention_counters(am_handle h, uint64_t* counts2);

/* ClassifierFreeGuidance.aggregate_cfg (guidance.py:95-118) + the Euler flow
 * step and masked write of SchedulerFlow._flow_sample (scheduler.py:238-248):
 *   v = v_0 + sum_i scale_i (v_{i+1} - v_i)   (bf16 arithmetic, as the reference)
 *   latents[f] += sign * bf16(dt * v[f])  for frames with unobserved[f] != 0
 *   latents_dev fp32 (T_local, N, Din); v_dev bf16 (n_branches, T_local, N, Din) */
int am_flow_step(const uint16_t* v_dev, float* latents_dev, int n_branches,
                 const float* scales_host, float dt, int is_additive,
                 const uint8_t* unobserved_host, int T_local, int N, int Din, void* stream);

/* algorithmic flops of one forward (SURVEY.md 8(d) formula) for the bound shape */
double am_step_flops(am_handle h, int B, int T_total, int N, int S);

/* ------------------------------------------------------------------------ */
/* Kernel-level API (stateless).  These are what seam S3 binds, and what the   */
/* parity tests call one by one.                                               */
/* ------------------------------------------------------------------------ */

/* C[M,N] = act(A[M,K] @ W[N,K]^T + bias) + residual, bf16 in/out, fp32 accumulate.
 * Replaces nn.Linear (+ GELU of diffusers FeedForward, + the block's residual add)
 * at block.py:131-152 / attention_processor.py:92-103,147.
 *   A may be split in two column blocks (A1: first K1 cols, A2: rest) = the
 *   torch.cat([skip, h]) of block.py:131 without materialising it.
 *   Row maps: physical_row(r) = (r / G) * gs + off + r % G  (G = 0 => identity).
 *   residual uses C's row map and leading dimension; may alias C.            */
typedef struct {
  const uint16_t* A1; int32_t lda1; int32_t K1;
  const uint16_t* A2; int32_t lda2;
  const uint16_t* W;  int32_t ldw;
  const float* bias;
  const uint16_t* residual;
  uint16_t* C; int32_t ldc;
  int32_t M, N, K;
  int32_t act;                 /* 0 none, 1 exact-erf GELU */
  int32_t a_G, a_gs, a_off;
  int32_t c_G, c_gs, c_off;
  /* LayerNorm folded into the linear that consumes it (block.py:138,146,152: norm -> to_q|k|v / to_q / ff.net.0): with
   * ln_stats != NULL, A holds the UN-normalised rows x, W holds bf16(W (.) gamma), `bias` holds d = W beta + b,
   * ln_colsum[n] = sum_k W'[n][k], ln_stats[r] = (mean_r, rstd_r) (am_row_stats_bf16 / am_row_stats_finalize), and
   *   C = act(rstd_r (x W'^T - mean_r colsum) + d) + residual
   * - the normalised activation is never written.  Identity A row map, no A2.  am_ln_fold_weight prepares W', colsum, d. */
  const float* ln_stats;
  const float* ln_colsum;
  /* LayerNorm statistics of the OUTPUT rows for the next consumer (written, not read): ln_part [M][ceil(N / 256)] pairs
   * (mean, M2) of each row's 256-column slice of the bf16-rounded C (after act + residual); am_row_stats_finalize merges them. */
  float* ln_part;
} am_gemm_args;
int am_gemm_bf16(const am_gemm_args* args, void* stream);

/* (mean, rstd) of every row of x [rows][C] bf16, fp32 two-pass statistics (the ones am_layernorm_bf16 uses). */
int am_row_stats_bf16(const uint16_t* x, float* stats, int64_t rows, int C, float eps, void* stream);
/* Merge the per-slice (mean, M2) pairs a producer GEMM wrote (am_gemm_args.ln_part, `nparts` = ceil(C / 256) slices of 256
 * columns, the last one shorter when C % 256 != 0) into (mean, rstd) per row - Chan's pairwise update, no E[x^2] - mean^2 cancellation. */
int am_row_stats_finalize(const float* part, int nparts, int C, float* stats, int64_t rows, float eps, void* stream);
/* am_layernorm_bf16 that also writes (mean, rstd) of its bf16-rounded OUTPUT rows to stats_y [rows][2] (block.py:133 -> :138:
 * norm_skip's output is the input of the next, folded, LayerNorm). */
int am_layernorm_stats_bf16(const uint16_t* x, uint16_t* y, const float* w, const float* b,
                            int64_t rows, int C, float eps, float* stats_y, void* stream);
/* One-time weight preparation of a folded linear: Wf = bf16(W (.) gamma) [N][K], colsum[n] = sum_k Wf[n][k],
 * d[n] = sum_k W[n][k] beta[k] + bias[n] (bias may be NULL).  W is the bf16 weight the un-folded linear uses. */
int am_ln_fold_weight(const uint16_t* W, const float* gamma, const float* beta, const float* bias, uint16_t* Wf, float* colsum,
                      float* d, int N, int K, void* stream);

/* FP32LayerNorm / nn.LayerNorm (block.py:64,83,98,107; temporal_denoiser.py:107):
 * y = (x-mean)/sqrt(var+eps)*w+b, fp32 statistics, bf16 in/out. C % 8 == 0, C <= 4096. */
int am_layernorm_bf16(const uint16_t* x, uint16_t* y, const float* w, const float* b,
                      int64_t rows, int C, float eps, void* stream);

/* fp32 residual stream (Stage II, temporal_autoencoder.py:258: the reference's torch.cat promotes its residual stream to fp32, every
 * `h + branch` under autocast adds a 16-bit linear output into it, FP32LayerNorm reads it in fp32 and the next autocast linear rounds its
 * input to 16 bits; the DINOv2 encoder, image_encoder.py:38-55 / pipeline.py:665-667, runs in fp32).  One pass:
 *   h32 [rows][C] += y16 [rows][C]   (the branch's 16-bit output; NULL: nothing to add),
 *   z16 [rows][C]  = LayerNorm(h32) * w + b, rounded to 16 bits   (NULL: accumulate only).   C % 8 == 0, C <= 4096. */
int am_add_layernorm_f32(float* h32, const uint16_t* y16, uint16_t* z16, const float* w, const float* b, int64_t rows, int C, float eps,
                         void* stream);

/* Head split + qk RMSNorm + RoPE + attention operand layout
 * (attention_processor.py:106-130).
 *   X [rows][ldx] bf16; head h, part p at columns (h*nparts + p)*128.
 *   rows are grouped in sequences of `seq_len` rows; each sequence is
 *   `seq_len / rows_per_frame` frames; the RoPE row of a token is its global
 *   frame index  (row / rows_per_frame).
 *   part kinds: 0 = Q-like (RMSNorm w0, optional RoPE) -> out0 [nseq][H][s0_pad][128]
 *               1 = K-like (RMSNorm w1, optional RoPE) -> out1 [nseq][H][s1_pad][128]
 *               2 = V      (copy)                      -> out2 [nseq][H][128][s1_pad], keys
 *                   permuted inside each group of 16 (bit2<->bit3) for the PV MFMA.  */
typedef struct {
  const uint16_t* X; int32_t ldx;
  int64_t rows; int32_t seq_len; int32_t rows_per_frame;
  int32_t heads; int32_t nparts; int32_t kinds[3];
  const float* w_q; const float* w_k; float eps;
  const float* rope_cos; const float* rope_sin;   /* [frames][64] or NULL */
  uint16_t* out_q; int32_t sq_pad;
  uint16_t* out_k; uint16_t* out_vt; int32_t sk_pad;
} am_headpost_args;
int am_head_post(const am_headpost_args* args, void* stream);

/* The bias-free q | k | v projection (or the cross-attention to_q) AND am_head_post in one launch (north_star's "fused RMSNorm + RoPE +
 * QKV"; attention_processor.py:92-130): the head split, qk-RMSNorm, RoPE and the operand layouts run in the GEMM's epilogue on the
 * bf16-rounded linear output - bit-identical to am_gemm_bf16 followed by am_head_post, without the (rows x N) activation round trip.
 * `gemm` as for am_gemm_bf16 with C = headpost->X, ldc = headpost->ldx, M = headpost->rows, N = heads * nparts * 128 (X is still needed:
 * the remainder rows of the tile grid go through it).  Shapes that do not qualify run the two calls instead - same results. */
int am_gemm_headpost_bf16(const am_gemm_args* gemm, const am_headpost_args* headpost, void* stream);

/* F.scaled_dot_product_attention, non-causal, head_dim 128
 * (attention_processor.py:133-139).  Online-softmax flash kernel on bf16 MFMA.
 *   Q  [nseq][H][sq_pad][128]; K [chunks][nseq][H][sk_pad][128];
 *   Vt [chunks][nseq][H][128][sk_pad] (permuted as am_head_post writes it);
 *   O  [nseq*sq][ldo] at column h*128 (heads concatenated, token-major).
 *   Each chunk holds `sk` valid keys.  sq_pad % 256 == 0, sk_pad % 64 == 0. */
typedef struct {
  const uint16_t* Q; const uint16_t* K; const uint16_t* Vt; uint16_t* O;
  int32_t nseq, heads, sq, sq_pad, sk, sk_pad, nchunks;
  int64_t chunk_stride;        /* elements between consecutive chunks of K (and of Vt) */
  int32_t ldo; float scale; int32_t defer_log2;
  /* Two-pass attention over a key stream that arrives in pieces (multi-GPU: the local K/V shard is there before
   * the all-gather of the others completes).  All zero = one pass over chunks 0 .. nchunks-1.
   *   chunk_total > 0: the `nchunks` chunks walked are (chunk_first + i) % chunk_total, i = 0 .. nchunks-1;
   *   rows: 0 = all query blocks, 1 = the full 256-row blocks a long key stream runs on the 4x64 kernel ("main"),
   *         2 = whatever `rows = 1` leaves out (the short last block);
   *   state_mode (rows = 1 only): 1 = stop after these chunks and save the un-normalised (O, m, l) of every row
   *         to `state` ([nseq*heads][sq_pad][132] floats), writing no output; 2 = start from `state`, finish, write O. */
  int32_t chunk_first, chunk_total, rows, state_mode;
  float* state;
} am_attn_args;
int am_attention_bf16(const am_attn_args* args, void* stream);
/* Diagnostic: the long-key-stream kernel keeps no running row max in its loop (it re-bases lazily from the row sums);
 * a workgroup that meets a single-tile jump of more than 2^60 is recomputed by an exact kernel launched behind it.
 * Returns how many workgroups that fallback has recomputed on the current device so far (synchronises the device). */
int am_attention_fallback_count(uint64_t* count);

/* fp8 (OCP e4m3) variant of the same attention (BASELINE.json configs[4]: "fp8 MFMA"): QK^T and P.V on the MX-scaled
 * K = 64 MFMA (block scales 2^0), fp32 online softmax, bf16 output.  Two calls: quantise the bf16 operand layouts that
 * am_head_post writes (Q pre-multiplied by scale * log2 e; V^T re-ordered to the key order of the fp8 P.V operand), then
 * attend.  q8 / k8 / vt8 have the element counts and strides of Q / K / Vt (one byte per element; chunk_stride counts
 * bytes).  The forms of am_attention_bf16 (round 3): chunk_first / chunk_total ring walks (the quantiser converts exactly the
 * chunks an attention call with the same arguments walks; rows = 2 skips Q), rows = 1 / 2, state_mode 1 / 2 with the same
 * [seq*heads][sq_pad][132] state layout, and the short last query block split over the key range.  Stated tolerance vs fp32
 * SDPA: tests/test_attention_fp8.py. */
int am_attention_quantize_fp8(const am_attn_args* args, uint8_t* q8, uint8_t* k8, uint8_t* vt8, void* stream);
int am_attention_fp8(const am_attn_args* args, const uint8_t* q8, const uint8_t* k8, const uint8_t* vt8, void* stream);

/* Copy-engine exchange of the K / V^T shards between ranks (multi-GPU, one process per GPU; the alternative to the RCCL
 * all-gather of seam S2's phase API, sharding.py PeerExchange): IPC-shareable device buffers, SDMA copies between them, and
 * sequence flags written / awaited in stream order by one-lane kernels.
 *   am_peer_alloc / free      hipMalloc'd (zeroed) buffer whose IPC handle can be exported
 *   am_peer_export / open / close   64-byte HIP IPC handle <-> mapped pointer in another process of the node
 *   am_peer_copy              asynchronous device-to-device copy on `stream` (copy engines, no CU)
 *   am_peer_signal            *flag = value (system-scope release) after everything earlier on `stream`
 *   am_peer_wait              holds `stream` until (int32)(*flag - value) >= 0; gives up after ~20 s and sets *fault_word */
int am_peer_alloc(size_t bytes, void** dev_ptr);
/* The flag block of an exchange (polled by a kernel of the owning device while a PEER device writes it): device-local FINE-GRAINED
 * memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained)), zeroed, IPC-exportable like am_peer_alloc's; *fine_grained = 1, or 0
 * when the runtime refused and an ordinary allocation was returned instead.  Free with am_peer_free. */
int am_peer_alloc_flags(size_t bytes, void** dev_ptr, int* fine_grained);
int am_peer_free(void* dev_ptr);
int am_peer_export(void* dev_ptr, uint8_t* handle64);
int am_peer_open(const uint8_t* handle64, void** dev_ptr);
int am_peer_close(void* dev_ptr);
int am_peer_copy(void* dst_dev, const void* src_dev, size_t bytes, void* stream);
int am_peer_signal(uint32_t* flag_dev, uint32_t value, void* stream);
int am_peer_wait(const uint32_t* flag_dev, uint32_t value, uint32_t* fault_word_dev, void* stream);

/* Diagnostic: while a trace is open, every kernel of the phase API (am_forward_begin .. am_forward_end) is followed by a
 * position-weighted integer checksum of its output, appended in stream order to `log_dev` (capacity uint64 words, zeroed by the
 * caller).  am_debug_trace_end closes the trace and returns the entries' tags (stage * 100 + layer; am_debug_trace_stage_name).
 * Process-wide, single-threaded like the rest of the API; used by tools/peer_selftest.py --ktrace to find which kernel's output
 * moves between two forwards on bit-identical inputs. */
int am_debug_trace_begin(uint64_t* log_dev, int capacity);
int am_debug_trace_end(int32_t* tags_host, int capacity, int* n_entries);
const char* am_debug_trace_stage_name(int stage);

/* small fused elementwise ops */
int am_f32_to_bf16(const float* x, uint16_t* y, size_t n, void* stream);
int am_bf16_to_f32(const uint16_t* x, float* y, size_t n, void* stream);
/* diffusers Timesteps(flip_sin_to_cos=False, shift=0): [sin(t f) | cos(t f)] -> bf16 (rows, C) */
int am_timestep_sinusoid(const float* t_dev, uint16_t* out, int rows, int C, void* stream);

/* Stage II (ActionMeshAutoencoder, temporal_autoencoder.py) featurisation around the kernels above:
 *   am_point_embed: FrequencyPositionalEmbedding (embeddings.py:14-52) of query (rows, ld_in) fp32 = [xyz | extra] ->
 *                   bf16 (rows, ld_out) = [x | sin(x f) | cos(x f) | extra | 0 pad], f_j = 2^j (* pi if include_pi)
 *   am_displacement: out (rows, out_dim) fp32 = 2 sigmoid(-logits) - 1   (temporal_autoencoder.py:156-157, 267) */
int am_point_embed(const float* q_dev, int ld_in, int64_t rows, int in_channels, int extra_channels, int num_freqs,
                   int include_pi, uint16_t* out, int ld_out, void* stream);
int am_displacement(const uint16_t* logits, int ld, int64_t rows, int out_dim, float* out, void* stream);

/* Context encoder (image_encoder.py:38-55 -> transformers Dinov2PatchEmbeddings): im2col of the kernel = stride patch
 * convolution, pixels (frames, channels, height, width) fp32 -> bf16 (frames * (height/patch) * (width/patch), ld_out),
 * columns in flattened-Conv2d-weight order (c, ky, kx), zero padded to ld_out; the projection itself is am_gemm_bf16. */
int am_patchify(const float* pixels, int frames, int channels, int height, int width, int patch, uint16_t* out,
                int ld_out, void* stream);

/* ActionBench quality gate (actionbench/chamfer.py:13-86, actionbench/icp.py:94): exact nearest-neighbour
 * search, the arithmetic of scipy.spatial.KDTree(points).query(queries) and of pytorch3d's chamfer_distance.
 *   points  (batch, n_points, 3) fp32, batch stride points_bstride ELEMENTS (0 = one cloud shared by every batch entry);
 *   queries (batch, n_queries, 3) fp32 likewise;
 *   out_index (batch, n_queries) int32 = argmin_p |q - p|^2, ties -> lowest index;
 *   out_d2    (batch, n_queries)       = that minimum SQUARED distance, double when precise != 0 (fp64 arithmetic,
 *             ((dx*dx) + (dy*dy)) + dz*dz without contraction: bit-identical to the reference's KD-tree), else float.
 * am_nn_workspace_bytes() bytes of device scratch (may be 0) must be passed for the same (n_points, n_queries, batch, precise). */
typedef struct {
  const float* points;
  int64_t n_points;
  int64_t points_bstride;
  const float* queries;
  int64_t n_queries;
  int64_t queries_bstride;
  int32_t batch;
  int32_t precise;
  int32_t* out_index;
  void* out_d2;
} am_nn_args;
size_t am_nn_workspace_bytes(int64_t n_points, int64_t n_queries, int batch, int precise);
int am_nn_search(const am_nn_args* args, void* workspace_dev, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACTIONMESH_AMD_H */
