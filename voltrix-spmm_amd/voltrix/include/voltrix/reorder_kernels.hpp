// Voltrix-SpMM for MI355X (gfx950) -- Cuthill-McKee row order on the device (the "reorder"
// half; profiles/HISTORY.md section 3.4).
//
// Integer work on the CSR, once per graph; HBM-bound (every level reads the rows of its frontier in A and in A^T once),
// no MFMA.  No reference counterpart: the reference reads externally reordered <name>.reorder.npz files
// (bench/graph_gen.py:42-45, bench/bench_all.py:120-129).  The order is a FUNCTION of the CSR (no race decides anything):
//   * graph: rows u, v < n are neighbours when A[u, v] or A[v, u] is stored (columns >= n of a rectangular A are not nodes);
//     the kernels walk row u of A and row u of A^T, so A + A^T is never materialised (and never de-duplicated);
//   * deg(u) = entries of row u of A + entries of row u of A^T;  tie(u) = position of u in the stable sort by deg;
//   * a component is searched breadth first from a start node; level(v) = distance from the start;
//   * inside level d the nodes are ordered by (rank of the earliest-ranked neighbour in level d - 1, tie) -- Cuthill-McKee;
//     rank = position in the component's order.
// Two phases per search, both with a single-workgroup form for runs of small levels (a banded graph has thousands of
// levels of a few hundred nodes: one launch walks them all, workgroup barriers between levels) and a whole-chip form for
// big levels (a social graph has three or four levels holding everything):
//   levels  push: frontier node u marks its unvisited neighbours (atomicCAS on level) and appends them to the queue
//           (atomicAdd on the tail) -- the queue's order INSIDE a level is arbitrary, the level sets are not;
//   ranks   pull: node v of level d takes min rank over its neighbours of level d - 1, key = (min rank, tie(v)), the
//           level's queue segment is sorted by key (LDS bitonic / rocPRIM radix sort, library plumbing) and ranked.
// oracle/oracle_np.py::cm_order restates the specification; tests compare permutations element for element.
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "voltrix/csr_preprocess.hpp"

namespace voltrix {

// ctrl int32[8] (device)
enum BfsCtrl {
  kBfsHead = 0,       // queue[head, tail) = the current frontier (level `depth`)
  kBfsTail = 1,
  kBfsNextTail = 2,   // appended so far
  kBfsDepth = 3,
  kBfsDone = 4,       // the frontier after the last level was empty; levels = depth + 1, nodes = tail
  kBfsCtrlInts = 8
};

constexpr int kBfsNarrowThreads = 1024;
constexpr int kBfsNarrowMax = 2048;     // frontiers up to here stay in the single-workgroup kernel
constexpr int kCmSmallLevel = 1024;     // levels up to here are ranked by the single-workgroup kernel (LDS bitonic)

struct BfsGraph {
  const int* indptr;      // [n + 1]
  const int* indices;
  const int* t_indptr;    // [t_rows + 1]  CSR of A^T (the search does not care about the order inside a row)
  const int* t_indices;
  int n;
  int t_rows;             // rows of A^T = columns of A
};

__device__ __forceinline__ int load_agent(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one wave: mark and append the unvisited neighbours of u
__device__ __forceinline__ void bfs_expand_node(const BfsGraph& g, int u, int depth, int* level, int* queue, int* ctrl,
                                                int lane) {
#pragma unroll 1
  for (int side = 0; side < 2; ++side) {
    if (side == 1 && u >= g.t_rows) break;
    const int* ptr = side ? g.t_indptr : g.indptr;
    const int* idx = side ? g.t_indices : g.indices;
    const int beg = ptr[u], end = ptr[u + 1];
    for (int e = beg + lane; e < end; e += 64) {
      const int v = idx[e];
      if (static_cast<unsigned>(v) >= static_cast<unsigned>(g.n)) continue;
      if (load_agent(level + v) != -1) continue;
      if (atomicCAS(level + v, -1, depth + 1) == -1) queue[atomicAdd(ctrl + kBfsNextTail, 1)] = v;
    }
  }
}

__device__ __forceinline__ void bfs_advance(int* ctrl, int* level_off) {
  const int tail = load_agent(ctrl + kBfsTail), next = load_agent(ctrl + kBfsNextTail);
  if (next == tail) {
    ctrl[kBfsDone] = 1;
  } else {
    const int depth = load_agent(ctrl + kBfsDepth) + 1;
    ctrl[kBfsDepth] = depth;
    ctrl[kBfsHead] = tail;
    ctrl[kBfsTail] = next;
    level_off[depth + 1] = next;
  }
}

__global__ void bfs_seed_kernel(int start, int* level, int* queue, int* ctrl, int* level_off) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    level[start] = 0;
    queue[0] = start;
    ctrl[kBfsHead] = 0;
    ctrl[kBfsTail] = 1;
    ctrl[kBfsNextTail] = 1;
    ctrl[kBfsDepth] = 0;
    ctrl[kBfsDone] = 0;
    level_off[0] = 0;
    level_off[1] = 1;
  }
}

// whole chip, ONE level per launch
__global__ __launch_bounds__(256) void bfs_wide_kernel(BfsGraph g, int* level, int* queue, int* ctrl) {
  if (ctrl[kBfsDone]) return;
  const int head = ctrl[kBfsHead], tail = ctrl[kBfsTail], depth = ctrl[kBfsDepth];
  const int lane = threadIdx.x & 63;
  const int waves = gridDim.x * 4;
  for (int i = head + blockIdx.x * 4 + (threadIdx.x >> 6); i < tail; i += waves)
    bfs_expand_node(g, queue[i], depth, level, queue, ctrl, lane);
}

__global__ void bfs_advance_kernel(int* ctrl, int* level_off) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && !ctrl[kBfsDone]) bfs_advance(ctrl, level_off);
}

// one workgroup, as many levels as stay narrow
__global__ __launch_bounds__(kBfsNarrowThreads) void bfs_narrow_kernel(BfsGraph g, int* level, int* queue, int* ctrl,
                                                                       int* level_off, int max_frontier) {
  __shared__ int s_head, s_tail, s_depth, s_done;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int iter = 0; iter <= g.n; ++iter) {           // every pass ends the search or adds a level: at most n passes
    if (threadIdx.x == 0) {
      s_head = load_agent(ctrl + kBfsHead);
      s_tail = load_agent(ctrl + kBfsTail);
      s_depth = load_agent(ctrl + kBfsDepth);
      s_done = load_agent(ctrl + kBfsDone);
    }
    __syncthreads();
    const int head = s_head, tail = s_tail, depth = s_depth;
    if (s_done || tail - head > max_frontier) break;
    for (int i = head + wave; i < tail; i += kBfsNarrowThreads / 64)
      bfs_expand_node(g, load_agent(queue + i), depth, level, queue, ctrl, lane);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      bfs_advance(ctrl, level_off);
      __threadfence();
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------ ranks
// one wave: key of node v of level d = (min rank over its neighbours of level d - 1) << 32 | tie(v)
__device__ __forceinline__ unsigned long long cm_key(const BfsGraph& g, int v, int d, const int* level, const int* rank,
                                                     const int* tie, int lane) {
  int best = 0x7fffffff;
#pragma unroll 1
  for (int side = 0; side < 2; ++side) {
    if (side == 1 && v >= g.t_rows) break;
    const int* ptr = side ? g.t_indptr : g.indptr;
    const int* idx = side ? g.t_indices : g.indices;
    const int beg = ptr[v], end = ptr[v + 1];
    for (int e = beg + lane; e < end; e += 64) {
      const int u = idx[e];
      if (static_cast<unsigned>(u) >= static_cast<unsigned>(g.n)) continue;
      if (level[u] == d - 1) best = min(best, load_agent(rank + u));
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) best = min(best, __shfl_xor(best, m, 64));
  return (static_cast<unsigned long long>(static_cast<unsigned>(best)) << 32) | static_cast<unsigned>(tie[v]);
}

// big level: keys of queue[off, off + m)
__global__ __launch_bounds__(256) void cm_keys_kernel(BfsGraph g, const int* level, const int* rank, const int* tie,
                                                      const int* queue, int off, int m, int d, unsigned long long* keys) {
  const int lane = threadIdx.x & 63;
  const int waves = gridDim.x * 4;
  for (int j = blockIdx.x * 4 + (threadIdx.x >> 6); j < m; j += waves) {
    const unsigned long long k = cm_key(g, queue[off + j], d, level, rank, tie, lane);
    if (lane == 0) keys[j] = k;
  }
}

// big level: the sorted segment back into the queue, ranks = base + position
__global__ void cm_apply_kernel(const int* sorted, int off, int m, int base, int* queue, int* rank) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) {
    const int v = sorted[j];
    queue[off + j] = v;
    rank[v] = base + off + j;
  }
}

// levels [d0, d1), every one of at most kCmSmallLevel nodes: one workgroup walks them in order
__global__ __launch_bounds__(1024) void cm_small_levels_kernel(BfsGraph g, const int* level, int* rank, const int* tie,
                                                               int* queue, const int* level_off, int d0, int d1, int base) {
  __shared__ unsigned long long s_key[kCmSmallLevel];
  __shared__ int s_val[kCmSmallLevel];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
  for (int d = d0; d < d1; ++d) {
    const int off = level_off[d], m = level_off[d + 1] - off;
    if (m > kCmSmallLevel) return;                     // host contract; never taken
    int p2 = 1;
    while (p2 < m) p2 <<= 1;
    for (int j = wave; j < m; j += 16) {
      const int v = load_agent(queue + off + j);
      const unsigned long long k = cm_key(g, v, d, level, rank, tie, lane);
      if (lane == 0) {
        s_key[j] = k;
        s_val[j] = v;
      }
    }
    if (tid >= m && tid < p2) s_key[tid] = ~0ull;
    __syncthreads();
    for (int k = 2; k <= p2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        const int partner = tid ^ j;
        if (tid < p2 && partner > tid) {
          const bool up = (tid & k) == 0;
          const unsigned long long a = s_key[tid], b = s_key[partner];
          if ((a > b) == up) {
            s_key[tid] = b;
            s_key[partner] = a;
            const int t = s_val[tid];
            s_val[tid] = s_val[partner];
            s_val[partner] = t;
          }
        }
        __syncthreads();
      }
    }
    if (tid < m) {
      const int v = s_val[tid];
      queue[off + tid] = v;
      __hip_atomic_store(rank + v, base + off + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------- host launchers
inline int bfs_grid(int n) {
  const long long want = (static_cast<long long>(n) + 3) / 4;
  return static_cast<int>(want < 1 ? 1 : (want > 4096 ? 4096 : want));
}

inline int bfs_seed(int start, int n, int* level, int* queue, int* ctrl, int* level_off, hipStream_t stream) {
  if (start < 0 || start >= n) return 1;
  hipLaunchKernelGGL(bfs_seed_kernel, dim3(1), dim3(64), 0, stream, start, level, queue, ctrl, level_off);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// one narrow launch (all the levels that stay <= kBfsNarrowMax) followed by `wide_levels` whole-chip levels; the caller
// reads ctrl afterwards (its one sync per call) and calls again until ctrl[kBfsDone]
inline int bfs_levels(const BfsGraph& g, int* level, int* queue, int* ctrl, int* level_off, int wide_levels,
                      hipStream_t stream) {
  if (g.n <= 0 || g.t_rows < 0 || wide_levels < 0) return 1;
  hipLaunchKernelGGL(bfs_narrow_kernel, dim3(1), dim3(kBfsNarrowThreads), 0, stream, g, level, queue, ctrl, level_off,
                     kBfsNarrowMax);
  const int grid = bfs_grid(g.n);
  for (int i = 0; i < wide_levels; ++i) {
    hipLaunchKernelGGL(bfs_wide_kernel, dim3(grid), dim3(256), 0, stream, g, level, queue, ctrl);
    hipLaunchKernelGGL(bfs_advance_kernel, dim3(1), dim3(64), 0, stream, ctrl, level_off);
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

struct CmWorkspace {
  unsigned long long* keys_in;
  unsigned long long* keys_out;
  int* vals_out;
  void* sort_temp;
  size_t sort_temp_bytes;
  long long bytes;
};
inline CmWorkspace cm_workspace(void* base, long long max_level) {
  CmWorkspace ws{};
  char* p = static_cast<char*>(base);
  ws.keys_in = reinterpret_cast<unsigned long long*>(p);
  p += align16(8 * max_level);
  ws.keys_out = reinterpret_cast<unsigned long long*>(p);
  p += align16(8 * max_level);
  ws.vals_out = reinterpret_cast<int*>(p);
  p += align16(4 * max_level);
  size_t temp = 0;
  if (max_level > 0)
    (void)rocprim::radix_sort_pairs(nullptr, temp, static_cast<unsigned long long*>(nullptr),
                                    static_cast<unsigned long long*>(nullptr), static_cast<int*>(nullptr),
                                    static_cast<int*>(nullptr), static_cast<size_t>(max_level), 0, 64, hipStream_t(0));
  ws.sort_temp = p;
  ws.sort_temp_bytes = temp;
  p += align16(static_cast<long long>(temp));
  ws.bytes = p - static_cast<char*>(base);
  return ws;
}
inline long long cm_rank_workspace_bytes(long long max_level) { return cm_workspace(nullptr, max_level).bytes; }

// ranks of one searched component: queue[0, level_off[num_levels]) holds its nodes level by level (host copy of the
// offsets in `level_off_host`, device copy in `level_off`); afterwards the queue is the component's order and
// rank[v] = base + position.  `workspace` sized for the largest level above kCmSmallLevel (0 bytes if there is none).
inline int cm_rank(const BfsGraph& g, const int* level, int* rank, const int* tie, int* queue, const int* level_off,
                   const int* level_off_host, int num_levels, int base, void* workspace, hipStream_t stream) {
  if (g.n <= 0 || num_levels < 1 || level_off_host[0] != 0 || level_off_host[1] != 1) return 1;
  long long max_level = 0;
  for (int d = 1; d < num_levels; ++d) {
    const long long m = level_off_host[d + 1] - level_off_host[d];
    if (m < 1) return 1;
    if (m > kCmSmallLevel && m > max_level) max_level = m;
  }
  CmWorkspace ws = cm_workspace(workspace, max_level);
  // level 0: the start node
  hipLaunchKernelGGL(cm_apply_kernel, dim3(1), dim3(64), 0, stream, queue, 0, 1, base, queue, rank);
  int d = 1;
  while (d < num_levels) {
    const int m = level_off_host[d + 1] - level_off_host[d];
    if (m <= kCmSmallLevel) {
      int e = d + 1;
      while (e < num_levels && level_off_host[e + 1] - level_off_host[e] <= kCmSmallLevel) ++e;
      hipLaunchKernelGGL(cm_small_levels_kernel, dim3(1), dim3(1024), 0, stream, g, level, rank, tie, queue, level_off, d, e,
                         base);
      d = e;
    } else {
      const int off = level_off_host[d];
      hipLaunchKernelGGL(cm_keys_kernel, dim3(bfs_grid(m)), dim3(256), 0, stream, g, level, rank, tie, queue, off, m, d,
                         ws.keys_in);
      size_t temp = ws.sort_temp_bytes;
      if (rocprim::radix_sort_pairs(ws.sort_temp, temp, ws.keys_in, ws.keys_out, queue + off, ws.vals_out,
                                    static_cast<size_t>(m), 0, 64, stream) != hipSuccess)
        return 2;
      hipLaunchKernelGGL(cm_apply_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, ws.vals_out, off, m, base, queue,
                         rank);
      ++d;
    }
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// ---------------------------------------------------------------------------------------------------------- transpose
// CSR of A^T, rows sorted (entries of a column in ascending row order, duplicates kept): row ids expanded per entry, one
// stable radix sort of (column, row) pairs by column (rocPRIM, library plumbing), row pointers by binary search in the
// sorted columns -- no atomics, the result is a function of the input.  Entries whose column id lies outside [0, num_cols)
// sort behind every valid one (unsigned keys) and are left out of t_indptr.
__global__ __launch_bounds__(256) void csr_expand_rows_kernel(const int* indptr, int num_rows, int* rows) {
  const int lane = threadIdx.x & 63;
  const int waves = gridDim.x * 4;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < num_rows; r += waves) {
    const int beg = indptr[r], end = indptr[r + 1];
    for (int e = beg + lane; e < end; e += 64) rows[e] = r;
  }
}

__global__ void csr_column_bounds_kernel(const unsigned* sorted_cols, long long nnz, int num_cols, int* t_indptr) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > num_cols) return;
  long long lo = 0, hi = nnz;                      // first position whose column is >= c
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (sorted_cols[mid] < static_cast<unsigned>(c)) lo = mid + 1; else hi = mid;
  }
  t_indptr[c] = static_cast<int>(lo);
}

struct TransposeWorkspace {
  int* rows;
  unsigned* cols_sorted;
  void* sort_temp;
  size_t sort_temp_bytes;
  long long bytes;
};
inline TransposeWorkspace transpose_workspace(void* base, long long nnz) {
  TransposeWorkspace ws{};
  char* p = static_cast<char*>(base);
  ws.rows = reinterpret_cast<int*>(p);
  p += align16(4 * nnz);
  ws.cols_sorted = reinterpret_cast<unsigned*>(p);
  p += align16(4 * nnz);
  size_t temp = 0;
  if (nnz > 0)
    (void)rocprim::radix_sort_pairs(nullptr, temp, static_cast<unsigned*>(nullptr), static_cast<unsigned*>(nullptr),
                                    static_cast<int*>(nullptr), static_cast<int*>(nullptr), static_cast<size_t>(nnz), 0, 32,
                                    hipStream_t(0));
  ws.sort_temp = p;
  ws.sort_temp_bytes = temp;
  p += align16(static_cast<long long>(temp));
  ws.bytes = p - static_cast<char*>(base);
  return ws;
}
inline long long csr_transpose_workspace_bytes(long long nnz) { return transpose_workspace(nullptr, nnz).bytes; }

inline int csr_transpose(const int* indptr, const int* indices, int num_rows, int num_cols, long long nnz, void* workspace,
                         int* t_indptr, int* t_indices, hipStream_t stream) {
  if (num_rows < 0 || num_cols < 0 || nnz < 0 || nnz > 0x7fffffffll) return 1;
  TransposeWorkspace ws = transpose_workspace(workspace, nnz);
  if (nnz > 0) {
    hipLaunchKernelGGL(csr_expand_rows_kernel, dim3(bfs_grid(num_rows)), dim3(256), 0, stream, indptr, num_rows, ws.rows);
    size_t temp = ws.sort_temp_bytes;
    if (rocprim::radix_sort_pairs(ws.sort_temp, temp, reinterpret_cast<const unsigned*>(indices), ws.cols_sorted, ws.rows,
                                  t_indices, static_cast<size_t>(nnz), 0, 32, stream) != hipSuccess)
      return 2;
  }
  hipLaunchKernelGGL(csr_column_bounds_kernel, dim3((num_cols + 256) / 256), dim3(256), 0, stream, ws.cols_sorted, nnz,
                     num_cols, t_indptr);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// ----------------------------------------------------------------------------------------------------------------------
// Cholesky QR of the spectral order's subspace iteration (voltrix/reorder.py::spectral_permutation), the small factor on the
// device (round 4): gram = X^T X (k x k, k <= 64, fp32) -> out = inv(chol(gram + eps trace(gram) I))^T (fp32), so that
// X <- X @ out has orthonormal columns.  Rounds 2-3 took the factor on the host with numpy: two device -> host syncs per
// iteration step.  One workgroup of 64 threads, arithmetic in double in LDS (k = 32: 16 KiB), column-by-column Cholesky
// (thread j owns row j of the trailing update) and one forward substitution per thread for the inverse.  A pivot that is
// not positive (rank-deficient subspace) is replaced by eps trace: the column is then a small multiple of noise, which the
// next deflation / iteration step repairs -- the host form raised instead.
constexpr int kCholMax = 64;

static __global__ __launch_bounds__(64) void chol_inv_transposed_kernel(const float* __restrict__ gram, const int k,
                                                                      const double eps, float* __restrict__ out) {
  __shared__ double a[kCholMax][kCholMax + 1];
  __shared__ double inv[kCholMax][kCholMax + 1];
  const int t = threadIdx.x;
  double trace = 0.0;
  for (int i = 0; i < k; ++i) trace += (double)gram[i * k + i];
  if (t < k)
    for (int c = 0; c < k; ++c) a[t][c] = 0.5 * ((double)gram[t * k + c] + (double)gram[c * k + t]) + (t == c ? eps * trace : 0.0);
  __syncthreads();
  for (int j = 0; j < k; ++j) {           // right-looking: after step j column j of `a` holds column j of L
    const double d = a[j][j] > 0.0 ? a[j][j] : (eps * trace > 0.0 ? eps * trace : 1e-300);
    const double piv = __builtin_sqrt(d);
    __syncthreads();
    if (t == j) a[j][j] = piv;
    if (t > j && t < k) a[t][j] = a[t][j] / piv;
    __syncthreads();
    if (t > j && t < k)
      for (int c = j + 1; c <= t; ++c) a[t][c] -= a[t][j] * a[c][j];
    __syncthreads();
  }
  // inv(L): thread c solves L x = e_c (x lower triangular column c)
  if (t < k) {
    for (int r = 0; r < k; ++r) {
      double v = (r == t) ? 1.0 : 0.0;
      for (int m = t; m < r; ++m) v -= a[r][m] * inv[m][t];
      inv[r][t] = r < t ? 0.0 : v / a[r][r];
    }
  }
  __syncthreads();
  if (t < k)
    for (int c = 0; c < k; ++c) out[t * k + c] = (float)inv[c][t];     // transposed
}

inline int chol_inv_transposed(const float* gram, int k, double eps, float* out, hipStream_t stream) {
  if (k < 1 || k > kCholMax || gram == nullptr || out == nullptr) return kErrBadShape;
  hipLaunchKernelGGL(chol_inv_transposed_kernel, dim3(1), dim3(64), 0, stream, gram, k, eps, out);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace voltrix
