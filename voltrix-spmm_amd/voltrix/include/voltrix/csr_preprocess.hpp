// Voltrix-SpMM for MI355X (gfx950) -- fused GPU preprocess: CSR (device) -> (pointer1, hspa_packed, hind).
//
// Produces, bit for bit, what the reference pipeline
//     voltrix::preprocess (CPU, bmat_kernels.cuh:264-320) -> hmat_cuda (:195-212) -> hmat_packed_swizzle_cuda (:228-242)
// produces, but entirely on the GPU and without the reference's three costs:
//   * the single-threaded std::map condensing on the host,
//   * the O(TC blocks x window edges) rescan in hmat_cuda_kernel (:66,:94),
//   * the transient fp32 `hspa` (512 bytes per TC block).
//
// Algorithm (one workgroup per 16-row window, grid-strided):
//   1. csr_window_sort_kernel   key = (column << 4) | (row & 15) for every edge of the window; bitonic sort
//                               (all-ascending network, virtual +inf padding) in LDS, or in the global workspace when
//                               the window has more than kSortLdsKeys edges; sorted keys -> workspace; distinct
//                               columns counted with wave ballots -> block_partition[w] = ceil(U_w / 8) (0 -> 1).
//   2. scan_* kernels           pointer1 = exclusive prefix sum (wave shuffles + LDS, three small launches).
//   3. csr_handle_zero_kernel   zero hspa_packed / hind (T read on the device from pointer1[W]).
//      csr_window_fill_kernel   per sorted key: "first of its column" flags -> ballot/popcount prefix sum = condensed
//                               column rank; hind[8*pointer1[w] + rank] = column; bit (row, rank) OR-ed into the
//                               reference's swizzled word/bit position.
// Limits: num_nodes <= 2^28 (the packed key keeps the column in 28 bits), num_edges + W <= INT32_MAX.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "voltrix/traits.hpp"

namespace voltrix {

constexpr int kSortThreads = 256;
constexpr int kSortLdsKeys = 8192;   // 32 KiB of LDS per workgroup -> 4 workgroups per CU
constexpr int kScanChunk = 2048;     // elements per scan workgroup (256 threads x 8)

// All-ascending bitonic network over keys[0..n): every comparator moves the smaller key to the lower index, so
// indices >= n can be treated as +inf and skipped (no power-of-two padding is materialised).
template <class Ptr>
__device__ __forceinline__ void bitonic_sort_ascending(Ptr keys, const int n, const int tid) {
  int P = 1;
  while (P < n) P <<= 1;
  const int half = P >> 1;
  for (int k = 2; k <= P; k <<= 1) {
    const int hk = k >> 1;
    for (int i = tid; i < half; i += kSortThreads) {  // "flip" step: partner mirrored inside the k-block
      const int base = (i / hk) * k, off = i % hk;
      const int lo = base + off, hi = base + k - 1 - off;
      if (hi < n) {
        const uint32_t x = keys[lo], y = keys[hi];
        if (x > y) {
          keys[lo] = y;
          keys[hi] = x;
        }
      }
    }
    __syncthreads();
    for (int j = k >> 2; j > 0; j >>= 1) {  // "shear" steps: partner at distance j
      for (int i = tid; i < half; i += kSortThreads) {
        const int lo = (i / j) * 2 * j + (i % j), hi = lo + j;
        if (hi < n) {
          const uint32_t x = keys[lo], y = keys[hi];
          if (x > y) {
            keys[lo] = y;
            keys[hi] = x;
          }
        }
      }
      __syncthreads();
    }
  }
}

static __global__ __launch_bounds__(kSortThreads) void csr_window_sort_kernel(const int* __restrict__ indptr,
                                                                       const int* __restrict__ indices,
                                                                       const int num_nodes, const int num_windows,
                                                                       uint32_t* __restrict__ keys_ws,
                                                                       int* __restrict__ block_partition) {
  __shared__ uint32_t lkeys[kSortLdsKeys];
  __shared__ int rowptr[kBlkH + 1];
  __shared__ int wave_cnt[kSortThreads / kWave];
  const int tid = threadIdx.x;
  for (int w = blockIdx.x; w < num_windows; w += gridDim.x) {
    if (tid <= kBlkH) {
      const long long r = (long long)w * kBlkH + tid;
      rowptr[tid] = indptr[r < num_nodes ? r : num_nodes];
    }
    __syncthreads();
    const int lo = rowptr[0], n = rowptr[kBlkH] - lo;
    uint32_t* const dst = keys_ws + lo;
    const bool in_lds = n <= kSortLdsKeys;  // workgroup-uniform
    for (int i = tid; i < n; i += kSortThreads) {
      const int e = lo + i;
      int rl = 0;
#pragma unroll
      for (int k = 1; k < kBlkH; ++k) rl += (rowptr[k] <= e) ? 1 : 0;  // local row of edge e
      const uint32_t key = ((uint32_t)indices[e] << 4) | (uint32_t)rl;
      if (in_lds) lkeys[i] = key; else dst[i] = key;
    }
    __syncthreads();
    if (in_lds) bitonic_sort_ascending(lkeys, n, tid); else bitonic_sort_ascending(dst, n, tid);

    int cnt = 0;  // distinct columns seen by this thread
    for (int i = tid; i < n; i += kSortThreads) {
      const uint32_t k = in_lds ? lkeys[i] : dst[i];
      const uint32_t prev = i ? (in_lds ? lkeys[i - 1] : dst[i - 1]) : 0u;
      cnt += (i == 0 || (k >> 4) != (prev >> 4)) ? 1 : 0;
      if (in_lds) dst[i] = k;
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, kWave);
    if ((tid & (kWave - 1)) == 0) wave_cnt[tid / kWave] = cnt;
    __syncthreads();
    if (tid == 0) {
      int u = 0;
#pragma unroll
      for (int i = 0; i < kSortThreads / kWave; ++i) u += wave_cnt[i];
      block_partition[w] = n == 0 ? 1 : (u + kBlkW - 1) / kBlkW;  // empty window -> 1 (reference quirk, :252)
    }
    __syncthreads();
  }
}

// ---- exclusive prefix sum of block_partition[W] -> pointer1[W+1] -------------------------------------------------
__device__ __forceinline__ int wave_inclusive_scan(int v) {
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int t = __shfl_up(v, off, kWave);
    if ((int)(threadIdx.x & (kWave - 1)) >= off) v += t;
  }
  return v;
}

static __global__ __launch_bounds__(256) void scan_chunk_sums_kernel(const int* __restrict__ in, const int n,
                                                              int* __restrict__ chunk_sums) {
  __shared__ int wsum[4];
  const long long base = (long long)blockIdx.x * kScanChunk;
  int s = 0;
  for (int i = threadIdx.x; i < kScanChunk; i += 256) s += (base + i < n) ? in[base + i] : 0;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) s += __shfl_down(s, off, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = s;
  __syncthreads();
  if (threadIdx.x == 0) chunk_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

static __global__ __launch_bounds__(256) void scan_chunk_offsets_kernel(int* __restrict__ chunk_sums, const int nchunks) {
  __shared__ int wsum[4];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nchunks; base += 256) {
    const int i = base + threadIdx.x;
    const int v = i < nchunks ? chunk_sums[i] : 0;
    const int inc = wave_inclusive_scan(v);
    if ((threadIdx.x & (kWave - 1)) == kWave - 1) wsum[threadIdx.x / kWave] = inc;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < (int)(threadIdx.x / kWave); ++k) woff += wsum[k];
    const int carry = carry_s;
    if (i < nchunks) chunk_sums[i] = carry + woff + inc - v;  // exclusive
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + woff + inc;
    __syncthreads();
  }
}

static __global__ __launch_bounds__(256) void scan_apply_kernel(const int* __restrict__ in, const int n,
                                                         const int* __restrict__ chunk_offsets,
                                                         int* __restrict__ out /* [n+1] */) {
  __shared__ int wsum[4];
  const long long base = (long long)blockIdx.x * kScanChunk;
  int carry = chunk_offsets[blockIdx.x];
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
  for (int it = 0; it < kScanChunk / 256; ++it) {
    const long long i = base + it * 256 + threadIdx.x;
    const int v = i < n ? in[i] : 0;
    const int inc = wave_inclusive_scan(v);
    if ((threadIdx.x & (kWave - 1)) == kWave - 1) wsum[threadIdx.x / kWave] = inc;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < (int)(threadIdx.x / kWave); ++k) woff += wsum[k];
    if (i < n) out[i + 1] = carry + woff + inc;
    carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
}

// ---- fill ----------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void csr_handle_zero_kernel(const int* __restrict__ pointer1, const int num_windows,
                                                              uint32_t* __restrict__ hspa_packed,
                                                              int* __restrict__ hind) {
  const long long total = pointer1[num_windows];
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  uint4* p4 = reinterpret_cast<uint4*>(hspa_packed);
  for (long long i = gid; i < total; i += stride) p4[i] = make_uint4(0u, 0u, 0u, 0u);
  int4* h4 = reinterpret_cast<int4*>(hind);
  for (long long i = gid; i < total * 2; i += stride) h4[i] = make_int4(0, 0, 0, 0);
}

static __global__ __launch_bounds__(kSortThreads) void csr_window_fill_kernel(const int* __restrict__ indptr,
                                                                       const int num_nodes, const int num_windows,
                                                                       const uint32_t* __restrict__ keys_ws,
                                                                       const int* __restrict__ pointer1,
                                                                       uint32_t* __restrict__ hspa_packed,
                                                                       int* __restrict__ hind) {
  __shared__ int wave_tot[kSortThreads / kWave];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  for (int w = blockIdx.x; w < num_windows; w += gridDim.x) {
    const long long r0 = (long long)w * kBlkH, r1 = r0 + kBlkH;
    const int lo = indptr[r0 < num_nodes ? r0 : num_nodes];
    const int n = indptr[r1 < num_nodes ? r1 : num_nodes] - lo;
    const long long p0 = pointer1[w];
    int carry = 0;  // distinct columns in the keys before `base` (workgroup-uniform)
    for (int base = 0; base < n; base += kSortThreads) {
      const int i = base + tid;
      const bool valid = i < n;
      const uint32_t key = valid ? keys_ws[lo + i] : 0u;
      const uint32_t prev = (valid && i > 0) ? keys_ws[lo + i - 1] : 0u;
      const bool first = valid && (i == 0 || (key >> 4) != (prev >> 4));  // first key of its column
      const unsigned long long m = __ballot(first);
      const int incl = __popcll(m & (~0ull >> (kWave - 1 - lane)));       // flags at lanes <= lane
      if (lane == 0) wave_tot[wv] = __popcll(m);
      __syncthreads();
      int woff = 0, tot = 0;
#pragma unroll
      for (int k = 0; k < kSortThreads / kWave; ++k) {
        woff += k < wv ? wave_tot[k] : 0;
        tot += wave_tot[k];
      }
      if (valid) {
        const int rank = carry + woff + incl - 1;  // condensed column of this edge (reference edgeToColumn)
        const int r = key & 15, c = rank & 7;
        const long long b = p0 + (rank >> 3);
        if (first) hind[8 * b + c] = (int)(key >> 4);
        // reference bit order (bmat_kernels.cuh:180-188): word t = (r>>3) + 2*(c>>2), bit 4*(r&7) + (c&3)
        atomicOr(&hspa_packed[4 * b + (r >> 3) + 2 * (c >> 2)], 1u << (4 * (r & 7) + (c & 3)));
      }
      carry += tot;
      __syncthreads();
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------
inline long long align16(long long x) { return (x + 15) & ~15ll; }

inline long long csr_preprocess_workspace_bytes(int num_nodes, long long num_edges) {
  const long long W = ((long long)num_nodes + kBlkH - 1) / kBlkH;
  const long long nchunks = (W + kScanChunk - 1) / kScanChunk + 1;
  return align16(num_edges * 4) + align16(nchunks * 4) + 16;
}

inline int csr_check(int num_nodes, long long num_edges) {
  if (num_nodes < 0 || num_edges < 0) return kErrBadShape;
  if (num_nodes > (1 << 28)) return kErrBadShape;  // packed sort key keeps the column in 28 bits
  const long long W = ((long long)num_nodes + kBlkH - 1) / kBlkH;
  if (num_edges + W > 0x7FFFFFFFll) return kErrOverflow;  // T <= E + W must fit the int32 handle
  return kOk;
}

inline int csr_window_count(const int* indptr, const int* indices, int num_nodes, long long num_edges, void* workspace,
                            int* block_partition, int* pointer1, hipStream_t stream) {
  if (int rc = csr_check(num_nodes, num_edges)) return rc;
  if ((uintptr_t)workspace & 15) return kErrBadShape;
  const int W = (num_nodes + kBlkH - 1) / kBlkH;
  if (W == 0) {
    return hipMemsetAsync(pointer1, 0, sizeof(int), stream) == hipSuccess ? kOk : kErrLaunch;
  }
  uint32_t* keys = reinterpret_cast<uint32_t*>(workspace);
  int* chunk_sums = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + align16(num_edges * 4));
  const int nchunks = (W + kScanChunk - 1) / kScanChunk;
  const int grid = W < 256 * 8 ? W : 256 * 8;
  hipLaunchKernelGGL(csr_window_sort_kernel, dim3(grid), dim3(kSortThreads), 0, stream, indptr, indices, num_nodes, W,
                     keys, block_partition);
  hipLaunchKernelGGL(scan_chunk_sums_kernel, dim3(nchunks), dim3(256), 0, stream, block_partition, W, chunk_sums);
  hipLaunchKernelGGL(scan_chunk_offsets_kernel, dim3(1), dim3(256), 0, stream, chunk_sums, nchunks);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nchunks), dim3(256), 0, stream, block_partition, W, chunk_sums, pointer1);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

inline int csr_fill(const int* indptr, const int* indices, int num_nodes, long long num_edges, void* workspace,
                    const int* pointer1, uint32_t* hspa_packed, int* hind, hipStream_t stream) {
  (void)indices;
  if (int rc = csr_check(num_nodes, num_edges)) return rc;
  if (((uintptr_t)workspace & 15) || ((uintptr_t)hspa_packed & 15) || ((uintptr_t)hind & 15)) return kErrBadShape;
  const int W = (num_nodes + kBlkH - 1) / kBlkH;
  if (W == 0) return kOk;
  const uint32_t* keys = reinterpret_cast<const uint32_t*>(workspace);
  hipLaunchKernelGGL(csr_handle_zero_kernel, dim3(256 * 8), dim3(256), 0, stream, pointer1, W, hspa_packed, hind);
  const int grid = W < 256 * 8 ? W : 256 * 8;
  hipLaunchKernelGGL(csr_window_fill_kernel, dim3(grid), dim3(kSortThreads), 0, stream, indptr, num_nodes, W, keys,
                     pointer1, hspa_packed, hind);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace voltrix
