// Exact nearest-neighbour search between point clouds: the arithmetic behind the ActionBench Chamfer metrics
// (reference actionbench/chamfer.py:13-86 - scipy.spatial.KDTree(points).query(queries), an exact Euclidean search in
// fp64 - and the pytorch3d chamfer_distance inside actionbench/icp.py:94).  Brute force: a KD-tree trades arithmetic for
// pointer chasing, which is the wrong trade on 256 CUs - 100 000 x 100 000 pairs are 1.2e11 fp64 lane-operations, a few ms.
//
//   grid = (query blocks, point splits, batch); block = 256 threads x QPT queries each, held in registers.
//   A block walks its split of the points in tiles of TILE, staged once into LDS as {x, y, z, 0} in the compute type
//   (the float -> double conversion is paid per tile, not per pair); all lanes read the same point (LDS broadcast).
//   d2 = ((dx*dx) + (dy*dy)) + dz*dz with contraction OFF: the summation of cKDTree (and of the numpy oracle), so the
//   fp64 result is bit-identical to the reference's; ties keep the lowest point index.
//   With more than one split the partial (d2, index) pairs go to a workspace and nn_reduce_kernel takes the minimum
//   (lowest split first, so the tie rule survives).
// precise = 1: fp64 arithmetic, d2 out as double (metrics).  precise = 0: fp32 (the ICP inner loop, 4800 searches).
#include "am_common.h"

namespace {

constexpr int NN_THREADS = 256;
constexpr int NN_TILE = 512;

template <typename T> struct Vec4 { T x, y, z, w; };

template <typename T, int QPT>
__global__ __launch_bounds__(NN_THREADS) void nn_search_kernel(const float* __restrict__ pts, int64_t P, int64_t p_bstride,
                                                               const float* __restrict__ qry, int64_t Q, int64_t q_bstride,
                                                               int64_t chunk, int nsplit, T* __restrict__ out_d2,
                                                               int32_t* __restrict__ out_idx) {
#pragma clang fp contract(off)
  __shared__ Vec4<T> tile[NN_TILE];
  const int b = blockIdx.z, split = blockIdx.y;
  pts += (int64_t)b * p_bstride;
  qry += (int64_t)b * q_bstride;
  const int64_t q0 = ((int64_t)blockIdx.x * NN_THREADS + threadIdx.x) * QPT;
  T qx[QPT], qy[QPT], qz[QPT], best[QPT];
  int32_t bi[QPT];
#pragma unroll
  for (int j = 0; j < QPT; ++j) {
    const int64_t q = q0 + j < Q ? q0 + j : Q - 1;      // clamp: the extra lanes redo the last query and do not store
    qx[j] = (T)qry[q * 3 + 0];
    qy[j] = (T)qry[q * 3 + 1];
    qz[j] = (T)qry[q * 3 + 2];
    best[j] = (T)INFINITY;
    bi[j] = -1;
  }
  const int64_t p_lo = (int64_t)split * chunk;
  const int64_t p_hi = p_lo + chunk < P ? p_lo + chunk : P;
  for (int64_t base = p_lo; base < p_hi; base += NN_TILE) {
    const int n = (int)(p_hi - base < NN_TILE ? p_hi - base : NN_TILE);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += NN_THREADS) {
      const float* s = pts + (base + i) * 3;
      tile[i] = Vec4<T>{(T)s[0], (T)s[1], (T)s[2], (T)0};
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
      const Vec4<T> p = tile[i];
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        const T dx = qx[j] - p.x, dy = qy[j] - p.y, dz = qz[j] - p.z;
        const T d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 < best[j]) {          // strict: the first (lowest-index) minimum wins
          best[j] = d2;
          bi[j] = (int32_t)(base + i);
        }
      }
    }
  }
  const int64_t o = ((int64_t)b * nsplit + split) * Q;
#pragma unroll
  for (int j = 0; j < QPT; ++j)
    if (q0 + j < Q) {
      out_d2[o + q0 + j] = best[j];
      out_idx[o + q0 + j] = bi[j];
    }
}

template <typename T>
__global__ void nn_reduce_kernel(const T* __restrict__ part_d2, const int32_t* __restrict__ part_idx, int64_t Q, int nsplit,
                                 T* __restrict__ out_d2, int32_t* __restrict__ out_idx) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (q >= Q) return;
  T best = (T)INFINITY;
  int32_t bi = -1;
  for (int s = 0; s < nsplit; ++s) {
    const T d = part_d2[((int64_t)b * nsplit + s) * Q + q];
    if (d < best) {
      best = d;
      bi = part_idx[((int64_t)b * nsplit + s) * Q + q];
    }
  }
  out_d2[(int64_t)b * Q + q] = best;
  out_idx[(int64_t)b * Q + q] = bi;
}

struct NnPlan { int qpt, qblocks, nsplit; int64_t chunk; };
NnPlan nn_plan(int64_t P, int64_t Q, int batch) {
  NnPlan pl;
  pl.qpt = Q * batch >= (int64_t)4 * NN_THREADS * 512 ? 4 : 1;       // 4 queries per thread once that still fills 512 blocks
  pl.qblocks = ceil_div(Q, (int64_t)NN_THREADS * pl.qpt);
  const int64_t blocks = (int64_t)pl.qblocks * batch;
  int want = blocks >= 512 ? 1 : (int)((512 + blocks - 1) / blocks);   // >= 2 workgroups per CU
  const int max_split = ceil_div(P, 4 * NN_TILE);                      // a split is worth at least four tiles
  if (want > max_split) want = max_split;
  if (want < 1) want = 1;
  pl.chunk = round_up(ceil_div(P, want), NN_TILE);
  pl.nsplit = ceil_div(P, pl.chunk);
  return pl;
}

template <typename T>
int nn_launch(const am_nn_args* a, const NnPlan& pl, void* ws, hipStream_t st) {
  T* out_d2 = reinterpret_cast<T*>(a->out_d2);
  T* part_d2 = out_d2;
  int32_t* part_idx = a->out_index;
  if (pl.nsplit > 1) {
    part_d2 = reinterpret_cast<T*>(ws);
    part_idx = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + sizeof(T) * (size_t)a->batch * pl.nsplit * a->n_queries);
  }
  const dim3 grid(pl.qblocks, pl.nsplit, a->batch);
  if (pl.qpt == 4)
    hipLaunchKernelGGL((nn_search_kernel<T, 4>), grid, dim3(NN_THREADS), 0, st, a->points, a->n_points, a->points_bstride,
                       a->queries, a->n_queries, a->queries_bstride, pl.chunk, pl.nsplit, part_d2, part_idx);
  else
    hipLaunchKernelGGL((nn_search_kernel<T, 1>), grid, dim3(NN_THREADS), 0, st, a->points, a->n_points, a->points_bstride,
                       a->queries, a->n_queries, a->queries_bstride, pl.chunk, pl.nsplit, part_d2, part_idx);
  AM_HIP(hipGetLastError());
  if (pl.nsplit > 1) {
    hipLaunchKernelGGL((nn_reduce_kernel<T>), dim3(ceil_div(a->n_queries, 256), a->batch), dim3(256), 0, st, part_d2, part_idx,
                       a->n_queries, pl.nsplit, out_d2, a->out_index);
    AM_HIP(hipGetLastError());
  }
  return AM_OK;
}

int nn_check(const am_nn_args* a) {
  AM_CHECK(a != nullptr, "am_nn_search: null arguments");
  AM_CHECK(a->batch >= 1 && a->n_points >= 1 && a->n_queries >= 1, "am_nn_search: empty problem (batch %d, %lld points, %lld queries)",
           a->batch, (long long)a->n_points, (long long)a->n_queries);
  AM_CHECK(a->n_points < (int64_t)1 << 31, "am_nn_search: %lld points do not fit a 32-bit index", (long long)a->n_points);
  AM_CHECK(a->batch <= 65535, "am_nn_search: batch %d exceeds the grid's z extent", a->batch);
  AM_CHECK(a->points && a->queries && a->out_d2 && a->out_index, "am_nn_search: null pointer");
  return AM_OK;
}

}  // namespace

extern "C" size_t am_nn_workspace_bytes(int64_t n_points, int64_t n_queries, int batch, int precise) {
  if (n_points < 1 || n_queries < 1 || batch < 1) return 0;
  const NnPlan pl = nn_plan(n_points, n_queries, batch);
  if (pl.nsplit <= 1) return 0;
  return ((precise ? sizeof(double) : sizeof(float)) + sizeof(int32_t)) * (size_t)batch * pl.nsplit * n_queries;
}

extern "C" int am_nn_search(const am_nn_args* a, void* workspace_dev, size_t workspace_bytes, void* stream) {
  AM_TRY(nn_check(a));
  const NnPlan pl = nn_plan(a->n_points, a->n_queries, a->batch);
  const size_t need = am_nn_workspace_bytes(a->n_points, a->n_queries, a->batch, a->precise);
  AM_CHECK(need == 0 || (workspace_dev != nullptr && workspace_bytes >= need), "am_nn_search: workspace of %zu bytes needed, %zu given",
           need, workspace_bytes);
  return a->precise ? nn_launch<double>(a, pl, workspace_dev, (hipStream_t)stream)
                    : nn_launch<float>(a, pl, workspace_dev, (hipStream_t)stream);
}
