// Model-level orchestration of the Stage-I denoiser forward on one MI355X:
// weights resident in HBM (bf16 matrices, fp32 norm/bias vectors), every
// activation buffer pre-allocated for the bound problem size, and the forward
// issued as a fixed sequence of the kernels in this library on the caller's
// stream.  Mirrors ActionMeshDenoiser.forward (temporal_denoiser.py:151-249)
// and FlowMatchingBlock.forward (block.py:110-154); the cross-attention K/V of
// the step-invariant context are cached per window (am_set_context).
#include <map>
#include <set>
#include <string>
#include <vector>

#include <string.h>

#include "am_common.h"

int am_add_bias_rows(bf16_t* h, const float* bias, int64_t rows, int C, void* stream);    // am_elementwise.hip
int am_add_bias_rows_stats(bf16_t* h, const float* bias, int64_t rows, int C, float eps, float* stats, void* stream);   // am_norm.hip

// diagnostic trace point (am_common.h am_trace): a checksum of `bytes` at `ptr` behind the kernel that has just been enqueued
#define TR(stage, layer, ptr, bytes) do { if (am_trace_on()) am_trace((stage) * 100 + (layer), (ptr), (size_t)(bytes), (void*)st); } while (0)

namespace {
constexpr int HD = 128;
inline int pad_to(int64_t x, int m) { return (int)round_up(x, m); }
}  // namespace

struct am_layer {
  bf16_t *w_qkv = nullptr, *w_so = nullptr, *w_xq = nullptr, *w_xkv = nullptr, *w_xo = nullptr;
  bf16_t *w_ff1 = nullptr, *w_ff2 = nullptr, *w_skip = nullptr;
  float *b_so = nullptr, *b_xo = nullptr, *b_ff1 = nullptr, *b_ff2 = nullptr, *b_skip = nullptr;
  float *ln_s_w = nullptr, *ln_s_b = nullptr, *ln_x_w = nullptr, *ln_x_b = nullptr;
  float *ln_f_w = nullptr, *ln_f_b = nullptr, *ln_k_w = nullptr, *ln_k_b = nullptr;
  float *s_nq = nullptr, *s_nk = nullptr, *x_nq = nullptr, *x_nk = nullptr;
  bf16_t *kx = nullptr, *vtx = nullptr;   // cross-attention K / V^T cache
  // the three linears that sit behind a LayerNorm (block.py:138 norm_s_attn -> to_q|k|v, :146 norm_x_attn -> to_q, :152 norm_ff ->
  // ff.net.0) with the norm folded in (am_ln_fold_weight): W (.) gamma, its column sums, and W beta + bias
  bf16_t *wf_qkv = nullptr, *wf_xq = nullptr, *wf_ff1 = nullptr;
  float *cs_qkv = nullptr, *d_qkv = nullptr, *cs_xq = nullptr, *d_xq = nullptr, *cs_ff1 = nullptr, *d_ff1 = nullptr;
};

struct am_model {
  am_config cfg;
  int C, H, F, Dc, Din, NL, P, rank;
  std::vector<am_layer> layers;
  bf16_t *w_t1 = nullptr, *w_t2 = nullptr, *w_in = nullptr, *w_out = nullptr;
  float *b_t1 = nullptr, *b_t2 = nullptr, *b_in = nullptr, *b_out = nullptr, *ln_o_w = nullptr, *ln_o_b = nullptr;
  bf16_t* wf_out = nullptr;                     // proj_out with norm_out folded in (round 6; temporal_denoiser.py:239-242)
  float *cs_out = nullptr, *d_out = nullptr;
  std::set<std::string> expected, loaded;
  std::vector<void*> allocs;
  float* stage_f32 = nullptr;
  size_t stage_elems = 0;
  // Host arguments are BORROWED for the call only (header conventions), but an asynchronous copy from pageable memory may run
  // after the call has returned - the caller's array can be gone by then (a ctypes temporary is freed at once; seen as one
  // garbage per-frame diffusion time in a forward when the device was busy).  Host data is therefore copied into a pinned
  // ring slot owned by the handle before the asynchronous copy is queued; a slot is re-used only after its copy has executed.
  // am_denoise_forward_graph: one captured forward per (operands, shape, stream, window) key
  struct GraphCache {
    const float* x = nullptr; uint16_t* v = nullptr; int B = 0, T = 0, N = 0; hipStream_t st = nullptr; uint64_t ctx_gen = 0, scratch_gen = 0;
    bool warm = false, disabled = false;
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    uint64_t replays = 0, captures = 0, eager = 0;
  } gc;
  uint64_t ctx_gen = 0;        // bumped by everything a captured forward bakes in (context, hints, K/V buffers)
  struct HostStage { void* pinned = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; };
  HostStage hstage[8];
  unsigned hstage_next = 0;

  // workspace
  int maxB, maxT, maxN, maxS, maxL;
  int64_t maxR;
  bf16_t *hwork = nullptr, *z = nullptr, *ao = nullptr, *qkv = nullptr, *ffh = nullptr;
  std::vector<bf16_t*> skip;
  bf16_t *Qb = nullptr, *Kg = nullptr, *Vtg = nullptr;
  uint8_t *Q8 = nullptr, *K8 = nullptr, *Vt8 = nullptr;   // fp8 operand copies (cfg.attn_fp8); K8 / Vt8 = the gather buffers when P > 1
  bool kv8_external = false;
  size_t chunk_stride8 = 0;    // bytes between consecutive ranks' K8 (or Vt8) chunks
  uint64_t n_attn_fp8 = 0, n_attn_bf16 = 0;   // inflated self-attention launches by arithmetic type (am_attention_counters)
  bool kv_external = false;
  size_t chunk_elems = 0;
  size_t chunk_stride = 0;     // elements between consecutive ranks' K (or V^T) chunks
  float* attn_state = nullptr; // two-pass self-attention: (O, m, l) of every query row after the local K/V shard
  bool local_done = false;     // am_layer_attn_local ran for the layer in flight
  bf16_t *xb = nullptr, *te0 = nullptr, *te1 = nullptr, *ctxb = nullptr, *kvtmp = nullptr;
  float *tdev = nullptr, *rope_cos = nullptr, *rope_sin = nullptr;
  // Folded LayerNorms: norm_s_attn / norm_x_attn / norm_ff are never launched and their outputs never exist in HBM.
  // INVARIANT while ln_fold is on: between the kernels of a forward, ln_stats[r] = (mean, rstd) of row r of the current residual
  // stream buffer (h->hsrc / h->hwork) - written by whoever wrote the rows: the producer GEMM's store loop through ln_part +
  // am_row_stats_finalize, the norm_skip LayerNorm, or am_row_stats_bf16 for rows no GEMM produced.
  bool ln_fold = true, folds_ready = false, ln_recompute = false;
  float *ln_stats = nullptr, *ln_part = nullptr;

  // per-window / per-forward state
  std::vector<uint8_t> ctx_zero;   // am_set_branch_hints: batch rows whose context is identically zero
  bool shared_prefix = false;      // am_set_branch_hints: all batch rows share hidden_states and t_bt
  bool ctx_set = false;
  int ctxB = 0, ctxT = 0, ctxS = 0;
  bool in_forward = false;
  int B = 0, T = 0, N = 0, L = 0;
  int64_t R = 0;
  bf16_t* hsrc = nullptr;
  int skip_top = 0;
  int next_layer = 0;   // enforces pre/post call order
  bool pre_done = false;

  bool inflated(int i) const {
    return i < 32 ? ((cfg.inflated_mask_lo >> i) & 1u) : ((cfg.inflated_mask_hi >> (i - 32)) & 1u);
  }
  bool has_skip(int i) const { return i > NL / 2; }   // temporal_denoiser.py:92
};

namespace {

void graph_drop(am_model* h) {
  if (h->gc.exec) (void)hipGraphExecDestroy(h->gc.exec);
  if (h->gc.graph) (void)hipGraphDestroy(h->gc.graph);
  h->gc.exec = nullptr; h->gc.graph = nullptr;
}

// dst_dev <- host bytes, asynchronously on `st`, safe for host memory that dies when the calling entry point returns
int stage_h2d(am_model* m, void* dst_dev, const void* src_host, size_t bytes, hipStream_t st) {
  am_model::HostStage& s = m->hstage[m->hstage_next++ % 8];
  if (s.used) AM_HIP(hipEventSynchronize(s.ev));          // the copy queued from this slot 8 uploads ago has executed
  if (s.cap < bytes) {
    if (s.pinned) AM_HIP(hipHostFree(s.pinned));
    s.pinned = nullptr;
    s.cap = bytes < 4096 ? 4096 : bytes;
    AM_HIP(hipHostMalloc(&s.pinned, s.cap, hipHostMallocDefault));
  }
  if (!s.ev) AM_HIP(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
  memcpy(s.pinned, src_host, bytes);
  AM_HIP(hipMemcpyAsync(dst_dev, s.pinned, bytes, hipMemcpyHostToDevice, st));
  AM_HIP(hipEventRecord(s.ev, st));
  s.used = true;
  return AM_OK;
}

int dev_alloc(am_model* m, void** p, size_t bytes, bool zero = true) {
  AM_HIP(hipMalloc(p, bytes ? bytes : 16));
  m->allocs.push_back(*p);
  if (zero) AM_HIP(hipMemset(*p, 0, bytes ? bytes : 16));
  return AM_OK;
}
template <class T>
int dev_alloc_t(am_model* m, T** p, size_t elems, bool zero = true) {
  return dev_alloc(m, reinterpret_cast<void**>(p), elems * sizeof(T), zero);
}

enum WKind { W_MAT, W_BIAS, W_NORM };
struct WSlot { void* dst; WKind kind; size_t numel; size_t dst_off; };

// Reference state-dict key -> destination in our fused layout.
bool resolve(am_model* m, const std::string& name, WSlot* s) {
  const size_t C = m->C, F = m->F, Dc = m->Dc, Din = m->Din;
  auto mat = [&](bf16_t* d, size_t n, size_t off = 0) { *s = {d, W_MAT, n, off}; return true; };
  auto bias = [&](float* d, size_t n) { *s = {d, W_BIAS, n, 0}; return true; };
  auto norm = [&](float* d, size_t n) { *s = {d, W_NORM, n, 0}; return true; };
  if (name == "time_proj.linear_1.weight") return mat(m->w_t1, 4 * C * C);
  if (name == "time_proj.linear_1.bias") return bias(m->b_t1, 4 * C);
  if (name == "time_proj.linear_2.weight") return mat(m->w_t2, 4 * C * C);
  if (name == "time_proj.linear_2.bias") return bias(m->b_t2, C);
  if (name == "proj_in.weight") return mat(m->w_in, C * Din);
  if (name == "proj_in.bias") return bias(m->b_in, C);
  if (name == "norm_out.weight") return norm(m->ln_o_w, C);
  if (name == "norm_out.bias") return norm(m->ln_o_b, C);
  if (name == "proj_out.weight") return mat(m->w_out, Din * C);
  if (name == "proj_out.bias") return bias(m->b_out, Din);
  if (name.rfind("blocks.", 0) != 0) return false;
  const size_t dot = name.find('.', 7);
  if (dot == std::string::npos) return false;
  int li = -1;
  try { li = std::stoi(name.substr(7, dot - 7)); } catch (...) { return false; }
  if (li < 0 || li >= m->NL) return false;
  am_layer& l = m->layers[li];
  const std::string r = name.substr(dot + 1);
  if (r == "norm_s_attn.weight") return norm(l.ln_s_w, C);
  if (r == "norm_s_attn.bias") return norm(l.ln_s_b, C);
  if (r == "s_attn.norm_q.weight") return norm(l.s_nq, HD);
  if (r == "s_attn.norm_k.weight") return norm(l.s_nk, HD);
  // attention_processor.py:106-110: head h reads columns [384h, 384h+384) of cat(q,k,v)
  // => the fused weight is the plain row-concatenation [Wq; Wk; Wv].
  if (r == "s_attn.to_q.weight") return mat(l.w_qkv, C * C, 0);
  if (r == "s_attn.to_k.weight") return mat(l.w_qkv, C * C, C * C);
  if (r == "s_attn.to_v.weight") return mat(l.w_qkv, C * C, 2 * C * C);
  if (r == "s_attn.to_out.0.weight") return mat(l.w_so, C * C);
  if (r == "s_attn.to_out.0.bias") return bias(l.b_so, C);
  if (r == "norm_x_attn.weight") return norm(l.ln_x_w, C);
  if (r == "norm_x_attn.bias") return norm(l.ln_x_b, C);
  if (r == "x_attn.norm_q.weight") return norm(l.x_nq, HD);
  if (r == "x_attn.norm_k.weight") return norm(l.x_nk, HD);
  if (r == "x_attn.to_q.weight") return mat(l.w_xq, C * C);
  if (r == "x_attn.to_k.weight") return mat(l.w_xkv, C * Dc, 0);          // :111-115 cat(k, v)
  if (r == "x_attn.to_v.weight") return mat(l.w_xkv, C * Dc, C * Dc);
  if (r == "x_attn.to_out.0.weight") return mat(l.w_xo, C * C);
  if (r == "x_attn.to_out.0.bias") return bias(l.b_xo, C);
  if (r == "norm_ff.weight") return norm(l.ln_f_w, C);
  if (r == "norm_ff.bias") return norm(l.ln_f_b, C);
  if (r == "ff.net.0.proj.weight") return mat(l.w_ff1, F * C);
  if (r == "ff.net.0.proj.bias") return bias(l.b_ff1, F);
  if (r == "ff.net.2.weight") return mat(l.w_ff2, C * F);
  if (r == "ff.net.2.bias") return bias(l.b_ff2, C);
  if (m->has_skip(li)) {
    if (r == "norm_skip.weight") return norm(l.ln_k_w, C);
    if (r == "norm_skip.bias") return norm(l.ln_k_b, C);
    if (r == "linear_skip.weight") return mat(l.w_skip, 2 * C * C);
    if (r == "linear_skip.bias") return bias(l.b_skip, C);
  }
  return false;
}

void build_expected(am_model* m) {
  auto& e = m->expected;
  for (const char* n : {"time_proj.linear_1.weight", "time_proj.linear_1.bias", "time_proj.linear_2.weight",
                        "time_proj.linear_2.bias", "proj_in.weight", "proj_in.bias", "norm_out.weight",
                        "norm_out.bias", "proj_out.weight", "proj_out.bias"})
    e.insert(n);
  for (int i = 0; i < m->NL; ++i) {
    const std::string p = "blocks." + std::to_string(i) + ".";
    for (const char* n :
         {"norm_s_attn.weight", "norm_s_attn.bias", "s_attn.norm_q.weight", "s_attn.norm_k.weight",
          "s_attn.to_q.weight", "s_attn.to_k.weight", "s_attn.to_v.weight", "s_attn.to_out.0.weight",
          "s_attn.to_out.0.bias", "norm_x_attn.weight", "norm_x_attn.bias", "x_attn.norm_q.weight",
          "x_attn.norm_k.weight", "x_attn.to_q.weight", "x_attn.to_k.weight", "x_attn.to_v.weight",
          "x_attn.to_out.0.weight", "x_attn.to_out.0.bias", "norm_ff.weight", "norm_ff.bias",
          "ff.net.0.proj.weight", "ff.net.0.proj.bias", "ff.net.2.weight", "ff.net.2.bias"})
      e.insert(p + n);
    if (m->has_skip(i))
      for (const char* n : {"norm_skip.weight", "norm_skip.bias", "linear_skip.weight", "linear_skip.bias"})
        e.insert(p + n);
  }
}

int gemm(hipStream_t st, const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias, const bf16_t* res,
         bf16_t* Cp, int ldc, int64_t M, int N, int K, int act, const bf16_t* A2 = nullptr, int lda2 = 0, int K1 = 0,
         int aG = 0, int ags = 0, int aoff = 0, int cG = 0, int cgs = 0, int coff = 0) {
  am_gemm_args g = {};
  g.A1 = A; g.lda1 = lda; g.K1 = A2 ? K1 : K;
  g.A2 = A2; g.lda2 = lda2;
  g.W = W; g.ldw = ldw; g.bias = bias; g.residual = res; g.C = Cp; g.ldc = ldc;
  g.M = (int)M; g.N = N; g.K = K; g.act = act;
  g.a_G = aG; g.a_gs = ags; g.a_off = aoff;
  g.c_G = cG; g.c_gs = cgs; g.c_off = coff;
  return am_gemm_bf16(&g, st);
}

int prepare_folds(am_model* m, hipStream_t st) {
  const int C = m->C, F = m->F;
  {   // the synchronisation below is illegal inside a stream capture (it would invalidate it): a caller that captures the split API
      // must run one eager forward after am_load_weights first (am_denoise_forward_graph does) - say so instead
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
      AM_FAIL(AM_ERR_STATE, "%s", "the folded LayerNorm weights are built (and synchronised) by the first forward after am_load_weights: "
                                  "run one forward eagerly before capturing this stream");
    (void)hipGetLastError();
  }
  for (am_layer& l : m->layers) {
    AM_TRY(am_ln_fold_weight(l.w_qkv, l.ln_s_w, l.ln_s_b, nullptr, l.wf_qkv, l.cs_qkv, l.d_qkv, 3 * C, C, st));
    AM_TRY(am_ln_fold_weight(l.w_xq, l.ln_x_w, l.ln_x_b, nullptr, l.wf_xq, l.cs_xq, l.d_xq, C, C, st));
    AM_TRY(am_ln_fold_weight(l.w_ff1, l.ln_f_w, l.ln_f_b, l.b_ff1, l.wf_ff1, l.cs_ff1, l.d_ff1, F, C, st));
  }
  AM_TRY(am_ln_fold_weight(m->w_out, m->ln_o_w, m->ln_o_b, m->b_out, m->wf_out, m->cs_out, m->d_out, m->Din, C, st));
  // once per weight load, never inside a capture: the folded weights are read by every later forward on WHATEVER stream it is
  // launched on (an eager forward on another torch stream, the graph stream after an eager call) - make them visible to all of
  // them before the flag says so (they used to be ordered only with the stream of the first forward)
  AM_HIP(hipStreamSynchronize(st));
  m->folds_ready = true;
  return AM_OK;
}

// (mean, rstd) of rows [r0, r0 + nr) of the residual stream from the slices a producer GEMM left in ln_part
// (x = the buffer the rows live in; ACTIONMESH_AMD_LN_STATS=recompute reads them back instead - same bits by construction
// (canonical statistics, am_common.h), which tests/test_denoiser_gpu.py asserts on whole forwards)
int finalize_stats(am_model* h, const bf16_t* x, int64_t r0, int64_t nr, hipStream_t st) {
  if (h->ln_recompute) return am_row_stats_bf16(x + r0 * h->C, h->ln_stats + 2 * r0, nr, h->C, 1e-5f, st);
  const int np = ceil_div(h->C, 256);
  return am_row_stats_finalize(h->ln_part + 2 * r0 * np, np, h->C, h->ln_stats + 2 * r0, nr, 1e-5f, st);
}

}  // namespace

// ---------------------------------------------------------------------------
extern "C" int am_create(const am_config* cfg, am_handle* out) {
  AM_CHECK(cfg && out, "am_create: null argument");
  AM_CHECK(cfg->num_heads > 0 && cfg->width == cfg->num_heads * HD, "am_create: width=%d must equal heads=%d * 128",
           cfg->width, cfg->num_heads);
  AM_CHECK(cfg->num_layers > 0 && cfg->num_layers <= 64, "am_create: num_layers=%d", cfg->num_layers);
  AM_CHECK(cfg->in_channels % 64 == 0 && cfg->in_channels > 0, "am_create: in_channels=%d must be a multiple of 64", cfg->in_channels);
  AM_CHECK(cfg->cross_dim % 64 == 0 && cfg->cross_dim > 0, "am_create: cross_dim=%d must be a multiple of 64", cfg->cross_dim);
  AM_CHECK(cfg->ff_inner % 64 == 0 && cfg->ff_inner > 0, "am_create: ff_inner=%d must be a multiple of 64", cfg->ff_inner);
  AM_CHECK(cfg->max_batch > 0 && cfg->max_frames_local > 0 && cfg->max_tokens > 0 && cfg->max_ctx_tokens > 0,
           "am_create: workspace bounds must be positive");
  AM_CHECK(cfg->world_size >= 1 && cfg->rank >= 0 && cfg->rank < cfg->world_size, "am_create: rank %d / world %d",
           cfg->rank, cfg->world_size);
  AM_CHECK(cfg->attn_defer_log2 == 0 || cfg->attn_defer_log2 == 8, "am_create: attn_defer_log2 must be 0 or 8");
  AM_CHECK(cfg->attn_fp8 >= 0 && cfg->attn_fp8 <= 2, "am_create: attn_fp8 must be 0 (bf16), 1 (fp8) or 2 (fp8, exponent-field probabilities)");
  int ndev = 0;
  AM_HIP(hipGetDeviceCount(&ndev));
  AM_CHECK(ndev > 0, "am_create: no HIP device visible (this library has no CPU path)");

  am_model* m = new am_model();
  m->cfg = *cfg;
  m->C = cfg->width; m->H = cfg->num_heads; m->F = cfg->ff_inner; m->Dc = cfg->cross_dim;
  m->Din = cfg->in_channels; m->NL = cfg->num_layers; m->P = cfg->world_size; m->rank = cfg->rank;
  m->layers.resize(m->NL);
  build_expected(m);
  const size_t C = m->C, F = m->F, Dc = m->Dc, Din = m->Din;
  int st = AM_OK;
#define A_(call) do { if (st == AM_OK) st = (call); } while (0)
  A_(dev_alloc_t(m, &m->w_t1, 4 * C * C)); A_(dev_alloc_t(m, &m->b_t1, 4 * C));
  A_(dev_alloc_t(m, &m->w_t2, 4 * C * C)); A_(dev_alloc_t(m, &m->b_t2, C));
  A_(dev_alloc_t(m, &m->w_in, C * Din)); A_(dev_alloc_t(m, &m->b_in, C));
  A_(dev_alloc_t(m, &m->w_out, Din * C)); A_(dev_alloc_t(m, &m->b_out, Din));
  A_(dev_alloc_t(m, &m->ln_o_w, C)); A_(dev_alloc_t(m, &m->ln_o_b, C));
  A_(dev_alloc_t(m, &m->wf_out, Din * C)); A_(dev_alloc_t(m, &m->cs_out, Din)); A_(dev_alloc_t(m, &m->d_out, Din));

  // workspace bounds
  m->maxB = cfg->max_batch; m->maxT = cfg->max_frames_local; m->maxN = cfg->max_tokens; m->maxS = cfg->max_ctx_tokens;
  m->maxL = m->maxN + 1;
  m->maxR = (int64_t)m->maxB * m->maxT * m->maxL;
  const int64_t BT = (int64_t)m->maxB * m->maxT;
  const int Spad = pad_to(m->maxS, 64);
  for (int i = 0; i < m->NL && st == AM_OK; ++i) {
    am_layer& l = m->layers[i];
    A_(dev_alloc_t(m, &l.w_qkv, 3 * C * C)); A_(dev_alloc_t(m, &l.w_so, C * C)); A_(dev_alloc_t(m, &l.b_so, C));
    A_(dev_alloc_t(m, &l.w_xq, C * C)); A_(dev_alloc_t(m, &l.w_xkv, 2 * C * Dc));
    A_(dev_alloc_t(m, &l.w_xo, C * C)); A_(dev_alloc_t(m, &l.b_xo, C));
    A_(dev_alloc_t(m, &l.w_ff1, F * C)); A_(dev_alloc_t(m, &l.b_ff1, F));
    A_(dev_alloc_t(m, &l.w_ff2, C * F)); A_(dev_alloc_t(m, &l.b_ff2, C));
    A_(dev_alloc_t(m, &l.ln_s_w, C)); A_(dev_alloc_t(m, &l.ln_s_b, C));
    A_(dev_alloc_t(m, &l.ln_x_w, C)); A_(dev_alloc_t(m, &l.ln_x_b, C));
    A_(dev_alloc_t(m, &l.ln_f_w, C)); A_(dev_alloc_t(m, &l.ln_f_b, C));
    A_(dev_alloc_t(m, &l.s_nq, (size_t)HD)); A_(dev_alloc_t(m, &l.s_nk, (size_t)HD));
    A_(dev_alloc_t(m, &l.x_nq, (size_t)HD)); A_(dev_alloc_t(m, &l.x_nk, (size_t)HD));
    if (m->has_skip(i)) {
      A_(dev_alloc_t(m, &l.w_skip, 2 * C * C)); A_(dev_alloc_t(m, &l.b_skip, C));
      A_(dev_alloc_t(m, &l.ln_k_w, C)); A_(dev_alloc_t(m, &l.ln_k_b, C));
    }
    A_(dev_alloc_t(m, &l.wf_qkv, 3 * C * C)); A_(dev_alloc_t(m, &l.cs_qkv, 3 * C)); A_(dev_alloc_t(m, &l.d_qkv, 3 * C));
    A_(dev_alloc_t(m, &l.wf_xq, C * C)); A_(dev_alloc_t(m, &l.cs_xq, C)); A_(dev_alloc_t(m, &l.d_xq, C));
    A_(dev_alloc_t(m, &l.wf_ff1, F * C)); A_(dev_alloc_t(m, &l.cs_ff1, F)); A_(dev_alloc_t(m, &l.d_ff1, F));
    A_(dev_alloc_t(m, &l.kx, (size_t)BT * m->H * Spad * HD));
    A_(dev_alloc_t(m, &l.vtx, (size_t)BT * m->H * HD * Spad));
  }
  const size_t R = (size_t)m->maxR;
  A_(dev_alloc_t(m, &m->hwork, R * C)); A_(dev_alloc_t(m, &m->z, R * C)); A_(dev_alloc_t(m, &m->ao, R * C));
  A_(dev_alloc_t(m, &m->qkv, R * 3 * C)); A_(dev_alloc_t(m, &m->ffh, R * F));
  m->skip.resize(m->NL / 2, nullptr);
  for (int i = 0; i < m->NL / 2; ++i) A_(dev_alloc_t(m, &m->skip[i], R * C));
  {
    const size_t q_inf = (size_t)m->maxB * m->H * pad_to((int64_t)m->maxT * m->maxL, 256) * HD;
    const size_t q_frm = (size_t)BT * m->H * pad_to(m->maxL, 256) * HD;
    A_(dev_alloc_t(m, &m->Qb, q_inf > q_frm ? q_inf : q_frm));
    const size_t k_inf = (size_t)m->maxB * m->H * pad_to((int64_t)m->maxT * m->maxL, 64) * HD;
    const size_t k_frm = (size_t)BT * m->H * pad_to(m->maxL, 64) * HD;
    m->chunk_elems = k_inf > k_frm ? k_inf : k_frm;
    m->chunk_stride = m->chunk_elems;
  }
  A_(dev_alloc_t(m, &m->xb, (size_t)BT * m->maxN * Din));
  A_(dev_alloc_t(m, &m->te0, (size_t)BT * C)); A_(dev_alloc_t(m, &m->te1, (size_t)BT * 4 * C));
  A_(dev_alloc_t(m, &m->tdev, (size_t)BT));
  A_(dev_alloc_t(m, &m->ctxb, (size_t)BT * m->maxS * Dc)); A_(dev_alloc_t(m, &m->kvtmp, (size_t)BT * m->maxS * 2 * C));
  A_(dev_alloc_t(m, &m->rope_cos, (size_t)BT * 64)); A_(dev_alloc_t(m, &m->rope_sin, (size_t)BT * 64));
  A_(dev_alloc_t(m, &m->ln_stats, R * 2)); A_(dev_alloc_t(m, &m->ln_part, R * 2 * (size_t)ceil_div((int)C, 256)));
  {
    const char* e = getenv("ACTIONMESH_AMD_LN_FOLD");       // 0: the round-3 sequence (LayerNorm kernel + plain linear), for A/B
    m->ln_fold = !(e && e[0] == '0');
    const char* r = getenv("ACTIONMESH_AMD_LN_STATS");
    m->ln_recompute = r && strcmp(r, "recompute") == 0;
  }
#undef A_
  if (st != AM_OK) {
    am_destroy(m);
    return st;
  }
  *out = m;
  return AM_OK;
}

extern "C" int am_destroy(am_handle h) {
  if (!h) return AM_OK;
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->stage_f32) (void)hipFree(h->stage_f32);
  graph_drop(h);
  for (auto& s : h->hstage) {
    if (s.used) (void)hipEventSynchronize(s.ev);
    if (s.ev) (void)hipEventDestroy(s.ev);
    if (s.pinned) (void)hipHostFree(s.pinned);
  }
  delete h;
  return AM_OK;
}

extern "C" int am_load_weight(am_handle h, const char* name, const float* host, size_t numel) {
  AM_CHECK(h && name && host, "am_load_weight: null argument");
  WSlot s;
  if (!resolve(h, name, &s)) AM_FAIL(AM_ERR_NOTFOUND, "am_load_weight: unknown state-dict key '%s'", name);
  AM_CHECK(numel == s.numel, "am_load_weight: '%s' has %zu elements, expected %zu", name, numel, s.numel);
  if (s.kind == W_MAT) {
    if (h->stage_elems < numel) {
      if (h->stage_f32) AM_HIP(hipFree(h->stage_f32));
      h->stage_f32 = nullptr; h->stage_elems = 0;
      AM_HIP(hipMalloc(reinterpret_cast<void**>(&h->stage_f32), numel * sizeof(float)));
      h->stage_elems = numel;
    }
    AM_HIP(hipMemcpy(h->stage_f32, host, numel * sizeof(float), hipMemcpyHostToDevice));
    AM_TRY(am_f32_to_bf16(h->stage_f32, reinterpret_cast<bf16_t*>(s.dst) + s.dst_off, numel, nullptr));
    AM_HIP(hipStreamSynchronize(nullptr));
  } else {
    std::vector<float> tmp(host, host + numel);
    if (s.kind == W_BIAS)   // autocast casts the bias to bf16 with the weight
      for (auto& v : tmp) v = bf2f(f2bf(v));
    AM_HIP(hipMemcpy(s.dst, tmp.data(), numel * sizeof(float), hipMemcpyHostToDevice));
  }
  h->loaded.insert(name);
  h->folds_ready = false;         // the folded linears (am_layer.wf_*) are rebuilt by the next forward's entry point, never inside a capture
  return AM_OK;
}

extern "C" int am_weights_missing(am_handle h) {
  if (!h) return -1;
  int missing = 0;
  for (const auto& n : h->expected)
    if (!h->loaded.count(n)) ++missing;
  return missing;
}

extern "C" int am_kv_chunk_elems(am_handle h, size_t* elems) {
  AM_CHECK(h && elems, "am_kv_chunk_elems: null argument");
  *elems = h->chunk_elems;
  return AM_OK;
}

extern "C" int am_bind_kv_buffers(am_handle h, uint16_t* k_dev, uint16_t* vt_dev, size_t chunk_stride_elems) {
  AM_CHECK(h && k_dev && vt_dev, "am_bind_kv_buffers: null argument");
  if (h->cfg.attn_fp8 && h->P > 1)
    AM_FAIL(AM_ERR_STATE, "am_bind_kv_buffers: an fp8 handle exchanges the QUANTISED shards - bind them with am_bind_kv8_buffers");
  AM_CHECK(((uintptr_t)k_dev | (uintptr_t)vt_dev) % 16 == 0 && chunk_stride_elems % 8 == 0, "am_bind_kv_buffers: misaligned");
  AM_CHECK(chunk_stride_elems == 0 || chunk_stride_elems >= h->chunk_elems, "am_bind_kv_buffers: chunk stride %zu < chunk size %zu",
           chunk_stride_elems, h->chunk_elems);
  h->Kg = k_dev; h->Vtg = vt_dev; h->kv_external = true;
  ++h->ctx_gen;
  h->chunk_stride = chunk_stride_elems ? chunk_stride_elems : h->chunk_elems;
  return AM_OK;
}

extern "C" int am_bind_kv8_buffers(am_handle h, uint8_t* k8_dev, uint8_t* vt8_dev, size_t chunk_stride_bytes) {
  AM_CHECK(h && k8_dev && vt8_dev, "am_bind_kv8_buffers: null argument");
  if (!h->cfg.attn_fp8) AM_FAIL(AM_ERR_STATE, "am_bind_kv8_buffers: the handle was not created with attn_fp8");
  AM_CHECK(((uintptr_t)k8_dev | (uintptr_t)vt8_dev) % 16 == 0 && chunk_stride_bytes % 16 == 0, "am_bind_kv8_buffers: misaligned");
  AM_CHECK(chunk_stride_bytes == 0 || chunk_stride_bytes >= h->chunk_elems, "am_bind_kv8_buffers: chunk stride %zu < chunk size %zu",
           chunk_stride_bytes, h->chunk_elems);
  if (h->K8 && !h->kv8_external) AM_FAIL(AM_ERR_STATE, "am_bind_kv8_buffers: bind before the first forward");
  h->K8 = k8_dev; h->Vt8 = vt8_dev; h->kv8_external = true;
  h->chunk_stride8 = chunk_stride_bytes ? chunk_stride_bytes : h->chunk_elems;
  ++h->ctx_gen;
  return AM_OK;
}

static int ensure_kv(am_model* m) {
  const bool fp8_sharded = m->cfg.attn_fp8 && m->P > 1;
  if (!m->Kg) {          // fp8 + sharding: the bf16 K / V^T are a private staging copy of the LOCAL shard only (the fp8 shards travel)
    const size_t chunks = fp8_sharded ? 1 : (size_t)m->P;
    AM_TRY(dev_alloc_t(m, &m->Kg, m->chunk_elems * chunks));
    AM_TRY(dev_alloc_t(m, &m->Vtg, m->chunk_elems * chunks));
  }
  if (m->cfg.attn_fp8) {
    if (!m->Q8) {
      const size_t q_inf = (size_t)m->maxB * m->H * pad_to((int64_t)m->maxT * m->maxL, 256) * HD;
      AM_TRY(dev_alloc_t(m, &m->Q8, q_inf));
    }
    if (!m->K8) {
      if (fp8_sharded) AM_FAIL(AM_ERR_STATE, "fp8 attention with world_size=%d: bind the fp8 gather buffers first (am_bind_kv8_buffers)", m->P);
      AM_TRY(dev_alloc_t(m, &m->K8, m->chunk_elems));
      AM_TRY(dev_alloc_t(m, &m->Vt8, m->chunk_elems));
      m->chunk_stride8 = m->chunk_elems;
    }
  }
  return AM_OK;
}

extern "C" int am_set_context(am_handle h, const float* ctx_dev, int B, int T, int S, const float* cos_host,
                              const float* sin_host, void* stream) {
  AM_CHECK(h && ctx_dev && cos_host && sin_host, "am_set_context: null argument");
  if (am_weights_missing(h) != 0) AM_FAIL(AM_ERR_STATE, "am_set_context: %d weights not loaded", am_weights_missing(h));
  AM_CHECK(B > 0 && B <= h->maxB && T > 0 && T <= h->maxT && S > 0 && S <= h->maxS,
           "am_set_context: (B=%d,T=%d,S=%d) exceeds workspace (%d,%d,%d)", B, T, S, h->maxB, h->maxT, h->maxS);
  hipStream_t st = (hipStream_t)stream;
  const int64_t BT = (int64_t)B * T;
  ++h->ctx_gen;
  AM_TRY(stage_h2d(h, h->rope_cos, cos_host, BT * 64 * sizeof(float), st));
  AM_TRY(stage_h2d(h, h->rope_sin, sin_host, BT * 64 * sizeof(float), st));
  AM_TRY(am_f32_to_bf16(ctx_dev, h->ctxb, (size_t)BT * S * h->Dc, st));
  const int Spad = pad_to(S, 64);
  for (int i = 0; i < h->NL; ++i) {
    am_layer& l = h->layers[i];
    AM_TRY(gemm(st, h->ctxb, h->Dc, l.w_xkv, h->Dc, nullptr, nullptr, h->kvtmp, 2 * h->C, BT * S, 2 * h->C, h->Dc, 0));
    am_headpost_args hp = {};
    hp.X = h->kvtmp; hp.ldx = 2 * h->C; hp.rows = BT * S; hp.seq_len = S; hp.rows_per_frame = S;
    hp.heads = h->H; hp.nparts = 2; hp.kinds[0] = 1; hp.kinds[1] = 2;
    hp.w_q = nullptr; hp.w_k = l.x_nk; hp.eps = 1e-6f;
    hp.out_k = l.kx; hp.out_vt = l.vtx; hp.sk_pad = Spad;
    AM_TRY(am_head_post(&hp, st));
  }
  h->ctx_set = true; h->ctxB = B; h->ctxT = T; h->ctxS = S;
  h->ctx_zero.assign((size_t)B, 0);
  h->shared_prefix = false;
  return AM_OK;
}

extern "C" int am_set_branch_hints(am_handle h, const uint8_t* ctx_is_zero_host, int shared_prefix) {
  AM_CHECK(h, "am_set_branch_hints: null handle");
  if (!h->ctx_set) AM_FAIL(AM_ERR_STATE, "am_set_branch_hints: am_set_context has not been called");
  h->ctx_zero.assign((size_t)h->ctxB, 0);
  if (ctx_is_zero_host)
    for (int b = 0; b < h->ctxB; ++b) h->ctx_zero[b] = ctx_is_zero_host[b] ? 1 : 0;
  h->shared_prefix = shared_prefix != 0 && h->P == 1 && h->ctxB > 1;
  ++h->ctx_gen;
  return AM_OK;
}

static int forward_check(am_model* h, const float* x_dev, const float* t_bt_host, int B, int T, int N) {
  AM_CHECK(h && x_dev && t_bt_host, "am_forward_begin: null argument");
  if (!h->ctx_set) AM_FAIL(AM_ERR_STATE, "am_forward_begin: am_set_context has not been called");
  AM_CHECK(B == h->ctxB && T == h->ctxT, "am_forward_begin: (B=%d,T=%d) differs from the bound context (%d,%d)", B, T, h->ctxB, h->ctxT);
  AM_CHECK(N > 0 && N <= h->maxN, "am_forward_begin: N=%d exceeds workspace %d", N, h->maxN);
  return AM_OK;
}
// everything of am_forward_begin behind the upload of the per-frame times (h->tdev): pure device work, capturable
static int forward_begin_body(am_model* h, const float* x_dev, int B, int T, int N, hipStream_t st);

extern "C" int am_forward_begin(am_handle h, const float* x_dev, const float* t_bt_host, int B, int T, int N, void* stream) {
  AM_TRY(forward_check(h, x_dev, t_bt_host, B, T, N));
  AM_TRY(ensure_kv(h));
  hipStream_t st = (hipStream_t)stream;
  if (h->ln_fold && !h->folds_ready) AM_TRY(prepare_folds(h, st));     // weights were (re)loaded since the last forward
  AM_TRY(stage_h2d(h, h->tdev, t_bt_host, (size_t)B * T * sizeof(float), st));
  return forward_begin_body(h, x_dev, B, T, N, st);
}

static int forward_begin_body(am_model* h, const float* x_dev, int B, int T, int N, hipStream_t st) {
  const int C = h->C, Din = h->Din;
  h->B = B; h->T = T; h->N = N; h->L = N + 1; h->R = (int64_t)B * T * h->L;
  const int64_t BT = (int64_t)B * T;
  // proj_in (temporal_denoiser.py:205-206) written behind each frame's time token (:217)
  AM_TRY(am_f32_to_bf16(x_dev, h->xb, (size_t)BT * N * Din, st));
  AM_TRY(gemm(st, h->xb, Din, h->w_in, Din, h->b_in, nullptr, h->hwork, C, BT * N, C, Din, 0, nullptr, 0, 0, 0, 0, 0,
              /*cG*/ N, /*cgs*/ h->L, /*coff*/ 1));
  // time token (:213-217): sinusoid -> Linear -> GELU -> Linear -> row 0 of each frame
  AM_TRY(am_timestep_sinusoid(h->tdev, h->te0, (int)BT, C, st));
  AM_TRY(gemm(st, h->te0, C, h->w_t1, C, h->b_t1, nullptr, h->te1, 4 * C, BT, 4 * C, C, 1));
  AM_TRY(gemm(st, h->te1, 4 * C, h->w_t2, 4 * C, h->b_t2, nullptr, h->hwork, C, BT, C, 4 * C, 0, nullptr, 0, 0, 0, 0, 0,
              /*cG*/ 1, /*cgs*/ h->L, /*coff*/ 0));
  TR(23, 0, h->hwork, (size_t)h->R * C * 2);
  if (h->ln_fold) AM_TRY(am_row_stats_bf16(h->hwork, h->ln_stats, h->R, C, 1e-5f, st));
  h->hsrc = h->hwork;
  h->skip_top = 0;
  h->next_layer = 0;
  h->pre_done = false;
  h->in_forward = true;
  return AM_OK;
}

static void self_attn_args(am_model* h, am_attn_args* at);

extern "C" int am_layer_pre_attn(am_handle h, int i, void* stream) {
  AM_CHECK(h, "am_layer_pre_attn: null handle");
  if (!h->in_forward || i != h->next_layer || h->pre_done)
    AM_FAIL(AM_ERR_STATE, "am_layer_pre_attn: layer %d out of order (next=%d)", i, h->next_layer);
  hipStream_t st = (hipStream_t)stream;
  am_layer& l = h->layers[i];
  const int C = h->C, L = h->L;
  const int64_t R = h->R;
  if (h->has_skip(i)) {   // block.py:131-133
    AM_CHECK(h->skip_top > 0, "am_layer_pre_attn: skip stack empty at layer %d", i);
    const bf16_t* sk = h->skip[--h->skip_top];
    AM_TRY(gemm(st, sk, C, l.w_skip, 2 * C, l.b_skip, nullptr, h->z, C, R, C, 2 * C, 0, h->hsrc, C, C));
    TR(1, i, h->z, (size_t)R * C * 2);
    if (h->ln_fold) AM_TRY(am_layernorm_stats_bf16(h->z, h->hwork, l.ln_k_w, l.ln_k_b, R, C, 1e-5f, h->ln_stats, st));
    else AM_TRY(am_layernorm_bf16(h->z, h->hwork, l.ln_k_w, l.ln_k_b, R, C, 1e-5f, st));
    TR(2, i, h->hwork, (size_t)R * C * 2);
    h->hsrc = h->hwork;
  }
  // exact shortcut (am_set_branch_hints): until the first cross-attention every batch row is the same tensor - layer 0's
  // self-attention branch runs on row 0 only and am_layer_post_attn copies its result to the other rows
  const bool shared = i == 0 && h->shared_prefix && !h->has_skip(0);
  const int64_t Rs = shared ? R / h->B : R;
  am_gemm_args gq = {};      // the q | k | v projection (:92-103), fused with the head split below (am_gemm_headpost_bf16)
  if (h->ln_fold) {          // norm_s_attn (block.py:138) inside the projection: rows of h, their statistics, W (.) gamma
    gq.A1 = h->hsrc; gq.W = l.wf_qkv; gq.bias = l.d_qkv; gq.ln_stats = h->ln_stats; gq.ln_colsum = l.cs_qkv;
  } else {
    AM_TRY(am_layernorm_bf16(h->hsrc, h->z, l.ln_s_w, l.ln_s_b, Rs, C, 1e-5f, st));      // block.py:138
    TR(3, i, h->z, (size_t)Rs * C * 2);
    gq.A1 = h->z; gq.W = l.w_qkv;
  }
  gq.lda1 = C; gq.K1 = C; gq.ldw = C; gq.C = h->qkv; gq.ldc = 3 * C;
  gq.M = (int)Rs; gq.N = 3 * C; gq.K = C;
  am_headpost_args hp = {};
  hp.X = h->qkv; hp.ldx = 3 * C; hp.rows = Rs; hp.rows_per_frame = L;
  hp.heads = h->H; hp.nparts = 3; hp.kinds[0] = 0; hp.kinds[1] = 1; hp.kinds[2] = 2;
  hp.w_q = l.s_nq; hp.w_k = l.s_nk; hp.eps = 1e-6f;
  hp.rope_cos = h->rope_cos; hp.rope_sin = h->rope_sin;
  hp.out_q = h->Qb;
  const bool fp8_layer = h->cfg.attn_fp8 && h->inflated(i);
  const bool fp8_sharded = h->cfg.attn_fp8 && h->P > 1;
  if (h->inflated(i)) {
    hp.seq_len = h->T * L;
    hp.sq_pad = pad_to(hp.seq_len, 256); hp.sk_pad = pad_to(hp.seq_len, 64);
    hp.out_k = fp8_sharded ? h->Kg : h->Kg + (size_t)h->rank * h->chunk_stride;
    hp.out_vt = fp8_sharded ? h->Vtg : h->Vtg + (size_t)h->rank * h->chunk_stride;
  } else {
    hp.seq_len = L;
    hp.sq_pad = pad_to(L, 256); hp.sk_pad = pad_to(L, 64);
    hp.out_k = h->Kg; hp.out_vt = h->Vtg;
  }
  if (am_trace_on()) {       // trace runs keep the two launches apart so that the linear's output can be checksummed
    AM_TRY(am_gemm_bf16(&gq, st));
    TR(4, i, h->qkv, (size_t)Rs * 3 * C * 2);
    AM_TRY(am_head_post(&hp, st));
  } else {
    AM_TRY(am_gemm_headpost_bf16(&gq, &hp, st));
  }
  if (am_trace_on()) {
    const size_t nseq = h->inflated(i) ? (size_t)h->B : (size_t)h->B * h->T;
    TR(5, i, h->Qb, nseq * h->H * hp.sq_pad * HD * 2);
    TR(6, i, hp.out_k, h->chunk_elems * 2);
    TR(6, i, hp.out_vt, h->chunk_elems * 2);
  }
  if (fp8_layer) {
    // fp8 variant: Q and THIS rank's K / V^T shard are quantised here, so that with world_size > 1 the shards that travel are the
    // fp8 ones (half the bytes on the links, and every rank quantises 1 / P of the keys instead of all of them after the gather)
    am_attn_args qa = {};
    self_attn_args(h, &qa);
    if (shared) qa.nseq /= h->B;
    qa.K = hp.out_k; qa.Vt = hp.out_vt; qa.nchunks = 1; qa.chunk_stride = 0;
    AM_TRY(am_attention_quantize_fp8(&qa, h->Q8, h->K8 + (size_t)h->rank * h->chunk_stride8, h->Vt8 + (size_t)h->rank * h->chunk_stride8, st));
  }
  h->pre_done = true;
  return AM_OK;
}

// Shared by the one-pass and the two-pass self-attention of an inflated layer.
static void self_attn_args(am_model* h, am_attn_args* at) {
  const int L = h->L;
  at->Q = h->Qb; at->K = h->Kg; at->Vt = h->Vtg; at->O = h->ao;
  at->heads = h->H; at->ldo = h->C; at->scale = 0.08838834764831845f; at->defer_log2 = h->cfg.attn_defer_log2;
  at->nseq = h->B; at->sq = h->T * L; at->sq_pad = pad_to(at->sq, 256);
  at->sk = h->T * L; at->sk_pad = pad_to(at->sk, 64);
  at->nchunks = h->P; at->chunk_stride = (int64_t)h->chunk_stride;
}

// Multi-GPU overlap: between am_layer_pre_attn and am_layer_post_attn of an inflated layer, while the all-gather of
// the other ranks' K / V^T shards is in flight, run the self-attention of the full query blocks against the LOCAL
// shard only and park the un-normalised (O, m, l).  am_layer_post_attn then resumes over the remote shards.
// A no-op (AM_OK) when the layer / shapes do not qualify: post_attn falls back to the one-pass attention.
extern "C" int am_layer_attn_local(am_handle h, int i, void* stream) {
  AM_CHECK(h, "am_layer_attn_local: null handle");
  if (!h->in_forward || i != h->next_layer || !h->pre_done)
    AM_FAIL(AM_ERR_STATE, "am_layer_attn_local: layer %d out of order (next=%d)", i, h->next_layer);
  h->local_done = false;
  const int tiles = (h->T * h->L + 63) / 64;
  const bool product = h->cfg.attn_defer_log2 == 0 || h->cfg.attn_defer_log2 == 8;
  if (h->P < 2 || !h->inflated(i) || tiles < 16 || !product) return AM_OK;
  am_attn_args at = {};
  self_attn_args(h, &at);
  if (!h->attn_state)
    AM_TRY(dev_alloc_t(h, &h->attn_state, (size_t)h->maxB * h->H * pad_to((int64_t)h->maxT * h->maxL, 256) * 132));
  at.rows = 1; at.state_mode = 1; at.state = h->attn_state;
  at.nchunks = 1; at.chunk_first = h->rank; at.chunk_total = h->P;
  if (h->cfg.attn_fp8) {                       // never a silent bf16 pass on an fp8 handle
    at.chunk_stride = (int64_t)h->chunk_stride8;
    if (h->cfg.attn_fp8 == 2) at.defer_log2 = 5400;      /* the exponent-field form of the probabilities (am_attention_fp8) */
    AM_TRY(am_attention_fp8(&at, h->Q8, h->K8, h->Vt8, stream));
    ++h->n_attn_fp8;
  } else {
    AM_TRY(am_attention_bf16(&at, stream));
    ++h->n_attn_bf16;
  }
  {
    hipStream_t st = (hipStream_t)stream;
    TR(7, i, h->attn_state, (size_t)at.nseq * at.heads * at.sq_pad * 132 * 4);
  }
  h->local_done = true;
  return AM_OK;
}

extern "C" int am_layer_post_attn(am_handle h, int i, void* stream) {
  AM_CHECK(h, "am_layer_post_attn: null handle");
  if (!h->in_forward || i != h->next_layer || !h->pre_done)
    AM_FAIL(AM_ERR_STATE, "am_layer_post_attn: layer %d out of order (next=%d)", i, h->next_layer);
  hipStream_t st = (hipStream_t)stream;
  am_layer& l = h->layers[i];
  const int C = h->C, L = h->L, F = h->F;
  const int64_t R = h->R;
  const float scale = 0.08838834764831845f;   // 1/sqrt(128)
  const bool shared = i == 0 && h->shared_prefix && !h->has_skip(0);
  const int64_t Rs = shared ? R / h->B : R;
  // ---- self-attention (attention_processor.py:133-166) ------------------------
  am_attn_args at = {};
  at.Q = h->Qb; at.K = h->Kg; at.Vt = h->Vtg; at.O = h->ao;
  at.heads = h->H; at.ldo = C; at.scale = scale; at.defer_log2 = h->cfg.attn_defer_log2;
  if (h->inflated(i) && h->local_done) {
    // second pass: the full blocks resume from the saved state over the P-1 remote shards (rank+1 .. rank-1, wrapping);
    // the short last block runs one pass over all shards
    self_attn_args(h, &at);
    am_attn_args rest = at;
    at.rows = 1; at.state_mode = 2; at.state = h->attn_state;
    at.nchunks = h->P - 1; at.chunk_first = (h->rank + 1) % h->P; at.chunk_total = h->P;
    if (am_trace_on()) {
      // the shards the attention really reads: an fp8 handle with P > 1 gathers QUANTISED shards in K8 / Vt8 (Kg / Vtg then hold one
      // chunk only - ensure_kv - so walking P chunks of them would read out of bounds)
      const bool fp8_sharded = h->cfg.attn_fp8 && h->P > 1;
      for (int c = 0; c < h->P; ++c) {
        if (fp8_sharded) {
          TR(8, i, h->K8 + (size_t)c * h->chunk_stride8, h->chunk_elems);
          TR(8, i, h->Vt8 + (size_t)c * h->chunk_stride8, h->chunk_elems);
        } else {
          TR(8, i, h->Kg + (size_t)c * h->chunk_stride, h->chunk_elems * 2);
          TR(8, i, h->Vtg + (size_t)c * h->chunk_stride, h->chunk_elems * 2);
        }
      }
      TR(9, i, h->Qb, (size_t)at.nseq * at.heads * at.sq_pad * HD * 2);
      TR(26, i, h->attn_state, (size_t)at.nseq * at.heads * at.sq_pad * 132 * 4);
    }
    if (h->cfg.attn_fp8) {
      at.chunk_stride = rest.chunk_stride = (int64_t)h->chunk_stride8;
      if (h->cfg.attn_fp8 == 2) at.defer_log2 = 5400;      /* the exponent-field form of the probabilities (am_attention_fp8) */
      AM_TRY(am_attention_fp8(&at, h->Q8, h->K8, h->Vt8, st));
      TR(10, i, h->ao, (size_t)Rs * C * 2);
      rest.rows = 2;
      if (h->cfg.attn_fp8 == 2) rest.defer_log2 = 5400;      /* the exponent-field form of the probabilities (am_attention_fp8) */
      AM_TRY(am_attention_fp8(&rest, h->Q8, h->K8, h->Vt8, st));
    } else {
      AM_TRY(am_attention_bf16(&at, st));
      TR(10, i, h->ao, (size_t)Rs * C * 2);
      rest.rows = 2;
      AM_TRY(am_attention_bf16(&rest, st));
    }
    TR(11, i, h->ao, (size_t)Rs * C * 2);
    h->local_done = false;
  } else {
    if (h->inflated(i)) {
      self_attn_args(h, &at);
    } else {
      at.nseq = h->B * h->T; at.sq = L; at.sq_pad = pad_to(L, 256);
      at.sk = L; at.sk_pad = pad_to(L, 64); at.nchunks = 1; at.chunk_stride = 0;
    }
    if (shared) at.nseq /= h->B;                           // row 0's sequences only (they come first in every layout)
    if (h->cfg.attn_fp8 && h->inflated(i)) {       // fp8 variant of the long-key-stream attention (configs[4]); operands quantised in pre
      at.chunk_stride = (int64_t)h->chunk_stride8;
      if (h->cfg.attn_fp8 == 2) at.defer_log2 = 5400;      /* the exponent-field form of the probabilities (am_attention_fp8) */
      AM_TRY(am_attention_fp8(&at, h->Q8, h->K8, h->Vt8, st));
      ++h->n_attn_fp8;
    } else {
      AM_TRY(am_attention_bf16(&at, st));
      if (h->inflated(i)) ++h->n_attn_bf16;
    }
    TR(12, i, h->ao, (size_t)Rs * C * 2);
  }
  {   // to_out + residual (block.py:137); with the norms folded, its store loop leaves the row statistics of the new h
    am_gemm_args go = {};
    go.A1 = h->ao; go.lda1 = C; go.K1 = C; go.W = l.w_so; go.ldw = C; go.bias = l.b_so; go.residual = h->hsrc;
    go.C = h->hwork; go.ldc = C; go.M = (int)Rs; go.N = C; go.K = C;
    if (h->ln_fold) go.ln_part = h->ln_part;
    AM_TRY(am_gemm_bf16(&go, st));
    if (h->ln_fold) AM_TRY(finalize_stats(h, h->hwork, 0, Rs, st));
  }
  TR(13, i, h->hwork, (size_t)Rs * C * 2);
  if (shared)
    for (int b = 1; b < h->B; ++b) {
      AM_HIP(hipMemcpyAsync(h->hwork + (size_t)b * Rs * C, h->hwork, (size_t)Rs * C * sizeof(bf16_t), hipMemcpyDeviceToDevice, st));
      if (h->ln_fold)
        AM_HIP(hipMemcpyAsync(h->ln_stats + (size_t)b * Rs * 2, h->ln_stats, (size_t)Rs * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
  h->hsrc = h->hwork;
  // ---- cross-attention to the frame's own context tokens (block.py:146-149) ----
  // Batch rows whose context is identically zero (am_set_branch_hints: the unconditional guidance branch) get the exact
  // result of the branch - h += to_out bias - instead of the branch; the others run it, one contiguous run of rows at a time.
  {
    const int64_t R1 = R / h->B;                       // rows of one batch row
    for (int b0 = 0; b0 < h->B;) {
      const bool zero = !h->ctx_zero.empty() && h->ctx_zero[b0];
      int b1 = b0 + 1;
      while (b1 < h->B && (!h->ctx_zero.empty() && h->ctx_zero[b1]) == zero) ++b1;
      const int64_t r0 = (int64_t)b0 * R1, nr = (int64_t)(b1 - b0) * R1;
      bf16_t* hrun = h->hwork + (size_t)r0 * C;
      if (zero) {
        if (h->ln_fold) AM_TRY(am_add_bias_rows_stats(hrun, l.b_xo, nr, C, 1e-5f, h->ln_stats + 2 * r0, st));
        else AM_TRY(am_add_bias_rows(hrun, l.b_xo, nr, C, st));
        TR(19, i, hrun, (size_t)nr * C * 2);
      } else {
        am_gemm_args gx = {};      // cross-attention to_q, fused with its head split
        if (h->ln_fold) {          // norm_x_attn (block.py:146) inside to_q
          gx.A1 = hrun; gx.W = l.wf_xq; gx.bias = l.d_xq; gx.ln_stats = h->ln_stats + 2 * r0; gx.ln_colsum = l.cs_xq;
        } else {
          AM_TRY(am_layernorm_bf16(hrun, h->z, l.ln_x_w, l.ln_x_b, nr, C, 1e-5f, st));
          TR(14, i, h->z, (size_t)nr * C * 2);
          gx.A1 = h->z; gx.W = l.w_xq;
        }
        gx.lda1 = C; gx.K1 = C; gx.ldw = C; gx.C = h->qkv; gx.ldc = C;
        gx.M = (int)nr; gx.N = C; gx.K = C;
        am_headpost_args hp = {};
        hp.X = h->qkv; hp.ldx = C; hp.rows = nr; hp.seq_len = L; hp.rows_per_frame = L;
        hp.heads = h->H; hp.nparts = 1; hp.kinds[0] = 0;
        hp.w_q = l.x_nq; hp.eps = 1e-6f;
        hp.out_q = h->Qb; hp.sq_pad = pad_to(L, 256);
        if (am_trace_on()) {
          AM_TRY(am_gemm_bf16(&gx, st));
          TR(15, i, h->qkv, (size_t)nr * C * 2);
          AM_TRY(am_head_post(&hp, st));
        } else {
          AM_TRY(am_gemm_headpost_bf16(&gx, &hp, st));
        }
        TR(16, i, h->Qb, (size_t)(b1 - b0) * h->T * h->H * hp.sq_pad * HD * 2);
        const int Spad = pad_to(h->ctxS, 64);
        am_attn_args ax = {};
        ax.Q = h->Qb;
        ax.K = l.kx + (size_t)b0 * h->T * h->H * Spad * HD;
        ax.Vt = l.vtx + (size_t)b0 * h->T * h->H * HD * Spad;
        ax.O = h->ao;
        ax.nseq = (b1 - b0) * h->T; ax.heads = h->H; ax.sq = L; ax.sq_pad = pad_to(L, 256);
        ax.sk = h->ctxS; ax.sk_pad = Spad; ax.nchunks = 1; ax.chunk_stride = 0;
        ax.ldo = C; ax.scale = scale; ax.defer_log2 = h->cfg.attn_defer_log2;
        AM_TRY(am_attention_bf16(&ax, st));
        TR(17, i, h->ao, (size_t)nr * C * 2);
        {
          am_gemm_args go = {};
          go.A1 = h->ao; go.lda1 = C; go.K1 = C; go.W = l.w_xo; go.ldw = C; go.bias = l.b_xo; go.residual = hrun;
          go.C = hrun; go.ldc = C; go.M = (int)nr; go.N = C; go.K = C;
          if (h->ln_fold) go.ln_part = h->ln_part + 2 * r0 * ceil_div(C, 256);
          AM_TRY(am_gemm_bf16(&go, st));
          if (h->ln_fold) AM_TRY(finalize_stats(h, h->hwork, r0, nr, st));
        }
        TR(18, i, hrun, (size_t)nr * C * 2);
      }
      b0 = b1;
    }
  }
  // ---- feed-forward (block.py:152; diffusers FeedForward "gelu") ------------------
  if (h->ln_fold) {          // norm_ff (block.py:152) inside ff.net.0
    am_gemm_args g1 = {};
    g1.A1 = h->hwork; g1.lda1 = C; g1.K1 = C; g1.W = l.wf_ff1; g1.ldw = C; g1.bias = l.d_ff1; g1.C = h->ffh; g1.ldc = F;
    g1.M = (int)R; g1.N = F; g1.K = C; g1.act = 1; g1.ln_stats = h->ln_stats; g1.ln_colsum = l.cs_ff1;
    AM_TRY(am_gemm_bf16(&g1, st));
  } else {
    AM_TRY(am_layernorm_bf16(h->hwork, h->z, l.ln_f_w, l.ln_f_b, R, C, 1e-5f, st));
    TR(20, i, h->z, (size_t)R * C * 2);
    AM_TRY(gemm(st, h->z, C, l.w_ff1, C, l.b_ff1, nullptr, h->ffh, F, R, F, C, 1));
  }
  TR(21, i, h->ffh, (size_t)R * F * 2);
  bf16_t* dst = h->hwork;
  if (i < h->NL / 2) dst = h->skip[h->skip_top++];     // temporal_denoiser.py:231-232 (kept, not copied)
  {
    // the next reader of the row statistics is the next block's norm_s_attn - unless that block starts with its skip linear + norm_skip
    // (which writes its own); behind the last block the reader is norm_out, folded into proj_out (am_forward_end, round 6)
    const bool want = h->ln_fold && (i + 1 >= h->NL || !h->has_skip(i + 1));
    am_gemm_args g2 = {};
    g2.A1 = h->ffh; g2.lda1 = F; g2.K1 = F; g2.W = l.w_ff2; g2.ldw = F; g2.bias = l.b_ff2; g2.residual = h->hwork;
    g2.C = dst; g2.ldc = C; g2.M = (int)R; g2.N = C; g2.K = F;
    if (want) g2.ln_part = h->ln_part;
    AM_TRY(am_gemm_bf16(&g2, st));
    if (want) AM_TRY(finalize_stats(h, dst, 0, R, st));
  }
  TR(22, i, dst, (size_t)R * C * 2);
  h->hsrc = dst;
  h->next_layer = i + 1;
  h->pre_done = false;
  return AM_OK;
}

extern "C" int am_forward_end(am_handle h, uint16_t* v_out, void* stream) {
  AM_CHECK(h && v_out, "am_forward_end: null argument");
  if (!h->in_forward || h->next_layer != h->NL) AM_FAIL(AM_ERR_STATE, "am_forward_end: %d of %d layers run", h->next_layer, h->NL);
  hipStream_t st = (hipStream_t)stream;
  const int C = h->C;
  // norm_out -> drop the time token -> proj_out (temporal_denoiser.py:239-242)
  if (h->ln_fold && h->NL > 0) {
    // round 6: norm_out inside proj_out like the three LayerNorms of a block - the last ff.net.2 left the row statistics of the
    // residual stream (am_layer_post_attn: `want`), proj_out reads the un-normalised rows through its row map (time token dropped)
    // and evaluates rstd (x W'^T - mean colsum) + d in its epilogue; the LayerNorm kernel and its R x C round trip are gone
    am_gemm_args g = {};
    g.A1 = h->hsrc; g.lda1 = C; g.K1 = C; g.W = h->wf_out; g.ldw = C; g.bias = h->d_out; g.C = v_out; g.ldc = h->Din;
    g.M = (int)((int64_t)h->B * h->T * h->N); g.N = h->Din; g.K = C;
    g.a_G = h->N; g.a_gs = h->L; g.a_off = 1;
    g.ln_stats = h->ln_stats; g.ln_colsum = h->cs_out;
    AM_TRY(am_gemm_bf16(&g, st));
  } else {
    AM_TRY(am_layernorm_bf16(h->hsrc, h->z, h->ln_o_w, h->ln_o_b, h->R, C, 1e-5f, st));
    TR(24, 0, h->z, (size_t)h->R * C * 2);
    AM_TRY(gemm(st, h->z, C, h->w_out, C, h->b_out, nullptr, v_out, h->Din, (int64_t)h->B * h->T * h->N, h->Din, C, 0,
                nullptr, 0, 0, /*aG*/ h->N, /*ags*/ h->L, /*aoff*/ 1));
  }
  TR(25, 0, v_out, (size_t)h->B * h->T * h->N * h->Din * 2);
  h->in_forward = false;
  return AM_OK;
}

extern "C" int am_denoise_forward(am_handle h, const float* x_dev, const float* t_bt_host, int B, int T, int N,
                                  uint16_t* v_out, void* stream) {
  AM_CHECK(h, "am_denoise_forward: null handle");
  if (h->P != 1) AM_FAIL(AM_ERR_STATE, "am_denoise_forward: world_size=%d needs the split API with a K/V all-gather", h->P);
  AM_TRY(am_forward_begin(h, x_dev, t_bt_host, B, T, N, stream));
  for (int i = 0; i < h->NL; ++i) {
    AM_TRY(am_layer_pre_attn(h, i, stream));
    AM_TRY(am_layer_post_attn(h, i, stream));
  }
  return am_forward_end(h, v_out, stream);
}

// am_denoise_forward through a HIP graph: the ~450 launches of a forward are captured once per (operands, shape, stream, window)
// and replayed; only the upload of the per-frame diffusion times stays outside the graph.  First call with a new key: eager (it
// also performs every lazy allocation / attribute call a capture must not contain); second: capture + instantiate + launch;
// later: launch.  A capture that fails disables the graph path for the handle (eager from then on, counted in am_graph_stats).
static int forward_body(am_model* h, const float* x_dev, int B, int T, int N, uint16_t* v_out, hipStream_t st) {
  AM_TRY(forward_begin_body(h, x_dev, B, T, N, st));
  for (int i = 0; i < h->NL; ++i) {
    AM_TRY(am_layer_pre_attn(h, i, st));
    AM_TRY(am_layer_post_attn(h, i, st));
  }
  return am_forward_end(h, v_out, st);
}
extern "C" int am_denoise_forward_graph(am_handle h, const float* x_dev, const float* t_bt_host, int B, int T, int N,
                                        uint16_t* v_out, void* stream) {
  AM_CHECK(h && v_out, "am_denoise_forward_graph: null argument");
  if (h->P != 1) AM_FAIL(AM_ERR_STATE, "am_denoise_forward_graph: world_size=%d needs the split API", h->P);
  am_model::GraphCache& g = h->gc;
  hipStream_t st = (hipStream_t)stream;
  if (g.disabled || st == nullptr) {          // the legacy null stream cannot be captured
    ++g.eager;
    return am_denoise_forward(h, x_dev, t_bt_host, B, T, N, v_out, stream);
  }
  AM_TRY(forward_check(h, x_dev, t_bt_host, B, T, N));
  if (h->ln_fold && !h->folds_ready) AM_TRY(prepare_folds(h, st));     // in front of a replay or a capture, never inside one
  const bool same = g.x == x_dev && g.v == v_out && g.B == B && g.T == T && g.N == N && g.st == st && g.ctx_gen == h->ctx_gen &&
                    g.scratch_gen == g_am_scratch_generation.load();
  if (!same) {
    graph_drop(h);
    g.x = x_dev; g.v = v_out; g.B = B; g.T = T; g.N = N; g.st = st; g.ctx_gen = h->ctx_gen; g.warm = true;
    ++g.eager;
    const int rc0 = am_denoise_forward(h, x_dev, t_bt_host, B, T, N, v_out, stream);
    g.scratch_gen = g_am_scratch_generation.load();      // after the eager run: it may have grown the scratch
    return rc0;
  }
  AM_TRY(stage_h2d(h, h->tdev, t_bt_host, (size_t)B * T * sizeof(float), st));
  if (!g.exec) {
    AM_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = forward_body(h, x_dev, B, T, N, v_out, st);
    hipGraph_t gr = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &gr);
    hipError_t e2 = hipSuccess;
    if (rc == AM_OK && e == hipSuccess && gr) e2 = hipGraphInstantiate(&g.exec, gr, nullptr, nullptr, 0);
    if (rc != AM_OK || e != hipSuccess || !gr || e2 != hipSuccess || !g.exec) {
      (void)hipGetLastError();
      if (gr) (void)hipGraphDestroy(gr);
      g.exec = nullptr; g.disabled = true; h->in_forward = false;
      ++g.eager;
      return am_denoise_forward(h, x_dev, t_bt_host, B, T, N, v_out, stream);
    }
    g.graph = gr;
    ++g.captures;
  } else {
    ++g.replays;
  }
  AM_HIP(hipGraphLaunch(g.exec, st));
  return AM_OK;
}
// counts[0] = graph launches of an existing executable, [1] = captures, [2] = eager forwards (first use of a key, null stream,
// or after a failed capture), [3] = 1 when a capture has failed and the graph path is off
extern "C" int am_graph_stats(am_handle h, uint64_t* counts4) {
  AM_CHECK(h && counts4, "am_graph_stats: null argument");
  counts4[0] = h->gc.replays; counts4[1] = h->gc.captures; counts4[2] = h->gc.eager; counts4[3] = h->gc.disabled ? 1 : 0;
  return AM_OK;
}

// counts2 = {fp8, bf16} inflated self-attention launches of this handle so far (a two-pass layer counts once): what a run really
// computed in, whatever it was configured for - bench.py derives its `dtype` and `roofline.peak` from this.
extern "C" int am_attention_counters(am_handle h, uint64_t* counts2) {
  AM_CHECK(h && counts2, "am_attention_counters: null argument");
  counts2[0] = h->n_attn_fp8; counts2[1] = h->n_attn_bf16;
  return AM_OK;
}

extern "C" double am_step_flops(am_handle h, int B, int T, int N, int S) {
  if (!h) return 0.0;
  const double C = h->C, F = h->F, Dc = h->Dc, Din = h->Din;
  const double TL = (double)T * (N + 1);
  double tot = 0.0;
  for (int i = 0; i < h->NL; ++i) {
    const double attn = h->inflated(i) ? 4.0 * TL * TL * C : 4.0 * T * (double)(N + 1) * (N + 1) * C;
    double per = 6.0 * TL * C * C + attn + 2.0 * TL * C * C;
    per += 2.0 * TL * C * C + 4.0 * T * S * Dc * C + 4.0 * TL * S * C + 2.0 * TL * C * C;
    per += 4.0 * TL * C * F;
    if (h->has_skip(i)) per += 4.0 * TL * C * C;
    tot += per;
  }
  tot += 4.0 * T * N * Din * C + 16.0 * T * C * C;
  return B * tot;
}
