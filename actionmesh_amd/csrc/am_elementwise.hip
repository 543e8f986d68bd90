// Small fused elementwise kernels of the denoise loop.
#include <stdarg.h>

#include "am_common.h"

// ---- error string (thread local) ---------------------------------------------
static thread_local char g_err[512] = "";
std::atomic<uint64_t> g_am_scratch_generation{0};

void am_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* am_last_error(void) { return g_err; }
extern "C" int am_abi_version(void) { return 2; }

namespace {

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x * 8;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      const f32x4_t a = *reinterpret_cast<const f32x4_t*>(x + i);
      const f32x4_t b = *reinterpret_cast<const f32x4_t*>(x + i + 4);
      u32x4_t o = {pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])};
      *reinterpret_cast<u32x4_t*>(y + i) = o;
    } else {
      for (size_t j = i; j < n; ++j) y[j] = f2bf(x[j]);
    }
  }
}

__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = bf2f(x[i]);
}

// diffusers Timesteps(num_channels=C, flip_sin_to_cos=False, downscale_freq_shift=0)
// as built at temporal_denoiser.py:57-61, followed by .to(bf16) (:213).
__global__ __launch_bounds__(256) void timestep_sinusoid_kernel(const float* __restrict__ t, bf16_t* __restrict__ out,
                                                                int rows, int C) {
  const int half = C / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * half) return;
  const int r = idx / half, i = idx - r * half;
  const float exponent = -9.210340371976184f * (float)i / (float)half;   // -ln(10000) * i / half
  const float arg = t[r] * expf(exponent);
  out[(int64_t)r * C + i] = f2bf(sinf(arg));
  out[(int64_t)r * C + half + i] = f2bf(cosf(arg));
}

// h[r][c] = bf16(h[r][c] + bias[c]) over `rows` rows: what `h + to_out(attention)` is when the attention output is exactly 0
// (the GEMM epilogue rounds acc = bias to bf16 - the bias is stored already rounded - and adds the bf16 residual).
__global__ __launch_bounds__(256) void add_bias_rows_kernel(bf16_t* __restrict__ h, const float* __restrict__ bias, int64_t n8, int C8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8) * 8;
    u32x4_t v = *reinterpret_cast<const u32x4_t*>(h + i * 8);
    const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(bias + c), b1 = *reinterpret_cast<const f32x4_t*>(bias + c + 4);
    v[0] = pack_bf2(bflo(v[0]) + rbf(b0[0]), bfhi(v[0]) + rbf(b0[1]));
    v[1] = pack_bf2(bflo(v[1]) + rbf(b0[2]), bfhi(v[1]) + rbf(b0[3]));
    v[2] = pack_bf2(bflo(v[2]) + rbf(b1[0]), bfhi(v[2]) + rbf(b1[1]));
    v[3] = pack_bf2(bflo(v[3]) + rbf(b1[2]), bfhi(v[3]) + rbf(b1[3]));
    *reinterpret_cast<u32x4_t*>(h + i * 8) = v;
  }
}

// aggregate_cfg (guidance.py:95-118) in bf16 + Euler step + masked write
// (scheduler.py:238-248).  dtype flow per SURVEY.md App. C: every bf16 op result
// is rounded to bf16; dt * v is rounded to bf16 before the fp32 latent add.
struct FlowArgs {
  const bf16_t* v; float* lat; int nb; float scales[3]; float dt; float sign;
  uint8_t unobs[256]; int T; int per_frame;
};
__global__ __launch_bounds__(256) void flow_step_kernel(FlowArgs a) {
  const int64_t total = (int64_t)a.T * a.per_frame;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int f = (int)(i / a.per_frame);
  if (!a.unobs[f]) return;
  float out = bf2f(a.v[i]);
  float prev = out;
  for (int b = 1; b < a.nb; ++b) {
    const float cur = bf2f(a.v[(int64_t)b * total + i]);
    out = rbf(out + rbf(a.scales[b - 1] * rbf(cur - prev)));
    prev = cur;
  }
  a.lat[i] += a.sign * rbf(a.dt * out);
}

}  // namespace

int am_add_bias_rows(bf16_t* h, const float* bias, int64_t rows, int C, void* stream) {      // internal (am_model.hip)
  const int64_t n8 = rows * (C / 8);
  const int64_t blocks = (n8 + 255) / 256;
  hipLaunchKernelGGL(add_bias_rows_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, h, bias, n8, C / 8);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

extern "C" int am_f32_to_bf16(const float* x, uint16_t* y, size_t n, void* stream) {
  AM_CHECK(x && y && n > 0, "am_f32_to_bf16: bad args");
  AM_CHECK(((uintptr_t)x | (uintptr_t)y) % 16 == 0, "am_f32_to_bf16: misaligned");
  const int blocks = (int)std::min<size_t>((n / 8 + 255) / 256 + 1, 4096);
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, n);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

extern "C" int am_bf16_to_f32(const uint16_t* x, float* y, size_t n, void* stream) {
  AM_CHECK(x && y && n > 0, "am_bf16_to_f32: bad args");
  const int blocks = (int)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, n);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

extern "C" int am_timestep_sinusoid(const float* t_dev, uint16_t* out, int rows, int C, void* stream) {
  AM_CHECK(t_dev && out && rows > 0 && C > 0 && C % 2 == 0, "am_timestep_sinusoid: bad args");
  const int n = rows * (C / 2);
  hipLaunchKernelGGL(timestep_sinusoid_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, t_dev, out, rows, C);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

// ---- Stage II (temporal_autoencoder.py) input / output featurisation -------------------------------------------
// FrequencyPositionalEmbedding (embeddings.py:14-52, logspace, include_input) + the extra channels (normals):
//   out[row] = [x (3) | sin(x_c f_j), c-major (3F) | cos(x_c f_j) (3F) | extra | 0-pad], f_j = 2^j (* pi)
__global__ void point_embed_kernel(const float* __restrict__ q, int ld_in, int64_t rows, int in_ch, int extra, int nfreq,
                                   float fscale, bf16_t* __restrict__ out, int ld_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ld_out) return;
  const int64_t row = i / ld_out;
  const int c = (int)(i - row * ld_out);
  const float* x = q + row * ld_in;
  const int nf = in_ch * nfreq;
  float v = 0.f;
  if (c < in_ch) v = x[c];
  else if (c < in_ch + 2 * nf) {
    const int e = (c - in_ch) % nf;
    const float a = x[e / nfreq] * (fscale * (float)(1 << (e % nfreq)));
    v = c < in_ch + nf ? sinf(a) : cosf(a);
  } else if (c < in_ch + 2 * nf + extra) v = x[in_ch + (c - in_ch - 2 * nf)];
  out[i] = f2bf(v);
}
// temporal_autoencoder.py:156-157, 267: logits * -1, then 2 * sigmoid - 1
__global__ void displacement_kernel(const bf16_t* __restrict__ logits, int ld, int64_t rows, int out_dim, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * out_dim) return;
  const int64_t row = i / out_dim;
  const float x = -bf2f(logits[row * ld + (i - row * out_dim)]);
  out[i] = 2.0f / (1.0f + __expf(-x)) - 1.0f;
}

// ---- context encoder (transformers Dinov2PatchEmbeddings: Conv2d with kernel = stride = patch) ------------------------
// im2col of the stride = kernel convolution: out[t * nh * nw + py * nw + px][c * p * p + ky * p + kx] = pix[t][c][py p + ky][px p + kx]
// (the column order of a flattened Conv2d weight), bf16, zero padded to ld_out so the patch projection is one GEMM.
__global__ void patchify_kernel(const float* __restrict__ pix, int ch, int H, int W, int patch, int nh, int nw, int64_t rows,
                                bf16_t* __restrict__ out, int ld_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ld_out) return;
  const int64_t row = i / ld_out;
  const int c = (int)(i - row * ld_out);
  float v = 0.f;
  if (c < ch * patch * patch) {
    const int64_t t = row / (nh * nw);
    const int pr = (int)(row - t * nh * nw);
    const int py = pr / nw, px = pr - py * nw;
    const int cc = c / (patch * patch), k = c - cc * patch * patch;
    const int ky = k / patch, kx = k - ky * patch;
    v = pix[((t * ch + cc) * H + py * patch + ky) * (int64_t)W + px * patch + kx];
  }
  out[i] = f2bf(v);
}

extern "C" int am_patchify(const float* pixels, int frames, int channels, int height, int width, int patch, uint16_t* out,
                           int ld_out, void* stream) {
  AM_CHECK(pixels && out && frames > 0 && channels > 0 && patch > 0, "am_patchify: bad args");
  AM_CHECK(height >= patch && width >= patch, "am_patchify: image %dx%d smaller than one patch (%d)", height, width, patch);
  AM_CHECK(ld_out >= channels * patch * patch, "am_patchify: ld_out=%d too small", ld_out);
  const int nh = height / patch, nw = width / patch;
  const int64_t rows = (int64_t)frames * nh * nw;
  const int64_t n = rows * ld_out;
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)ceil_div(n, (int64_t)256)), dim3(256), 0, (hipStream_t)stream, pixels,
                     channels, height, width, patch, nh, nw, rows, out, ld_out);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

extern "C" int am_point_embed(const float* q_dev, int ld_in, int64_t rows, int in_channels, int extra_channels, int num_freqs,
                              int include_pi, uint16_t* out, int ld_out, void* stream) {
  AM_CHECK(q_dev && out && rows > 0, "am_point_embed: bad args");
  AM_CHECK(in_channels > 0 && extra_channels >= 0 && num_freqs > 0 && num_freqs < 24 && ld_in >= in_channels + extra_channels,
           "am_point_embed: bad channel counts");
  AM_CHECK(ld_out >= in_channels * (2 * num_freqs + 1) + extra_channels, "am_point_embed: ld_out=%d too small", ld_out);
  const int64_t n = rows * ld_out;
  hipLaunchKernelGGL(point_embed_kernel, dim3((unsigned)ceil_div(n, (int64_t)256)), dim3(256), 0, (hipStream_t)stream, q_dev, ld_in,
                     rows, in_channels, extra_channels, num_freqs, include_pi ? 3.14159265358979323846f : 1.0f, out, ld_out);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

extern "C" int am_displacement(const uint16_t* logits, int ld, int64_t rows, int out_dim, float* out, void* stream) {
  AM_CHECK(logits && out && rows > 0 && out_dim > 0 && ld >= out_dim, "am_displacement: bad args");
  const int64_t n = rows * out_dim;
  hipLaunchKernelGGL(displacement_kernel, dim3((unsigned)ceil_div(n, (int64_t)256)), dim3(256), 0, (hipStream_t)stream, logits, ld,
                     rows, out_dim, out);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

extern "C" int am_flow_step(const uint16_t* v_dev, float* latents_dev, int n_branches, const float* scales_host,
                            float dt, int is_additive, const uint8_t* unobserved_host, int T_local, int N, int Din,
                            void* stream) {
  AM_CHECK(v_dev && latents_dev, "am_flow_step: null operand");
  AM_CHECK(n_branches >= 1 && n_branches <= 4, "am_flow_step: n_branches=%d", n_branches);
  AM_CHECK(n_branches == 1 || scales_host, "am_flow_step: scales required");
  AM_CHECK(T_local > 0 && T_local <= 256 && N > 0 && Din > 0, "am_flow_step: bad shape");
  FlowArgs a;
  a.v = v_dev;
  a.lat = latents_dev;
  a.nb = n_branches;
  for (int i = 0; i < 3; ++i) a.scales[i] = (i < n_branches - 1) ? scales_host[i] : 0.f;
  a.dt = dt;
  a.sign = is_additive ? 1.f : -1.f;
  for (int f = 0; f < 256; ++f) a.unobs[f] = (f < T_local) ? (unobserved_host ? unobserved_host[f] : 1) : 0;
  a.T = T_local;
  a.per_frame = N * Din;
  const int64_t total = (int64_t)T_local * N * Din;
  hipLaunchKernelGGL(flow_step_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
  AM_HIP(hipGetLastError());
  return AM_OK;
}

// ---- diagnostic trace: a checksum after every kernel of the forward (which kernel's output moves between two runs?) -------------
namespace {
__global__ __launch_bounds__(256) void checksum_kernel(const uint32_t* __restrict__ p, size_t nwords, unsigned long long* out) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x)
    acc += (unsigned long long)p[i] * (2ull * i + 1ull);          // position-weighted: a swapped pair of words changes the sum
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);               // integer: order-free
}
unsigned long long* g_trace_log = nullptr;
int g_trace_cap = 0, g_trace_n = 0;
int g_trace_tags[4096];
}  // namespace
bool am_trace_on() { return g_trace_log != nullptr; }
void am_trace(int tag, const void* dev_ptr, size_t bytes, void* stream) {
  if (!g_trace_log || g_trace_n >= g_trace_cap || !dev_ptr) return;
  const size_t nwords = bytes / 4;
  const int blocks = (int)((nwords + 256 * 16 - 1) / (256 * 16));
  g_trace_tags[g_trace_n] = tag;
  hipLaunchKernelGGL(checksum_kernel, dim3(blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const uint32_t*>(dev_ptr), nwords, g_trace_log + g_trace_n);
  ++g_trace_n;
}
extern "C" int am_debug_trace_begin(uint64_t* log_dev, int capacity) {
  AM_CHECK(log_dev != nullptr && capacity > 0, "am_debug_trace_begin: bad argument");
  g_trace_log = reinterpret_cast<unsigned long long*>(log_dev);
  g_trace_cap = capacity < 4096 ? capacity : 4096;
  g_trace_n = 0;
  return AM_OK;
}
extern "C" int am_debug_trace_end(int32_t* tags_host, int capacity, int* n_entries) {
  AM_CHECK(n_entries != nullptr, "am_debug_trace_end: null argument");
  *n_entries = g_trace_n;
  for (int i = 0; tags_host && i < g_trace_n && i < capacity; ++i) tags_host[i] = g_trace_tags[i];
  g_trace_log = nullptr; g_trace_cap = 0; g_trace_n = 0;
  return AM_OK;
}
extern "C" const char* am_debug_trace_stage_name(int stage) {
  static const char* names[] = {"?", "skip linear (z)", "skip LayerNorm (h)", "norm_s_attn (z)", "qkv GEMM", "head_post Q", "head_post local K|V^T chunk",
                                "attn local pass: state", "K|V^T all chunks before attention", "Q before attention", "self-attn O after resume pass",
                                "self-attn O after last-block pass", "self-attn O (one pass)", "to_out + residual (h)", "norm_x_attn (z)", "cross to_q GEMM",
                                "cross head_post Q", "cross-attn O", "cross to_out + residual (h)", "bias-only cross branch (h)", "norm_ff (z)",
                                "ff1 + GELU", "ff2 + residual (h)", "proj_in + time token (h)", "norm_out (z)", "proj_out (v)", "attn state before resume"};
  return stage >= 0 && stage < (int)(sizeof(names) / sizeof(names[0])) ? names[stage] : "?";
}
