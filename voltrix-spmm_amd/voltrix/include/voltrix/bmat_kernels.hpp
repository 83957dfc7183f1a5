// Voltrix-SpMM for MI355X (gfx950) -- CSR -> row-window / TC-block format ("bmat") kernels.
//
// Same *results* as the reference's voltrix/include/voltrix/bmat_kernels.cuh, different
// algorithms:
//   preprocess()               ref :264-320  one thread, std::map per window
//                              here: windows spread over host threads, sort + unique + binary search
//   hmat_hip()                 ref :21-111,195-212  per TC block re-scan of all window edges, O(TCb*E)
//                              here: zero-fill kernel + one scatter per edge, O(E)
//   hmat_packed_swizzle_hip()  ref :151-193,228-242  4 active threads / TC block, 32 strided reads each
//                              here: one wave64 per TC block, 2 coalesced loads + 2 ballots + nibble compress
// The byte layout of every output (pointer1, blockPartition, edgeToColumn, edgeToRow, hspa, hind,
// hspa_packed) is the reference's, bit for bit; tests check that against
// oracle/.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "voltrix/traits.hpp"

namespace voltrix {

// ----------------------------------------------------------------------------------------------
// Host: row-window condensing (reference bmat_kernels.cuh:264-320).
//   blockPartition[w] = ceil(U_w / 8)   (U_w = distinct neighbour ids of rows 16w..16w+15; U_w = 0 -> 1,
//                                        the reference's quirk, :252,:298-299 -- kept, goldens depend on it)
//   edgeToColumn[e]   = rank of edgeList[e] among the window's sorted distinct ids (:304-307)
//   edgeToRow[e]      = row of edge e (:273-276)
//   Pointer1          = exclusive prefix sum of blockPartition (:312-319)
// Returns a ReturnCode; never prints (the reference printf's TC_Blocks/Exp_Edges, :309-310).
inline int preprocess(const int32_t* edgeList, const int32_t* nodePointer, int num_nodes, int blockSize_h,
                      int blockSize_w, int32_t* blockPartition, int32_t* edgeToColumn, int32_t* edgeToRow,
                      int32_t* Pointer1) {
  if (num_nodes < 0 || blockSize_h != kBlkH || blockSize_w != kBlkW) return kErrBadShape;
  const int64_t num_windows = ((int64_t)num_nodes + kBlkH - 1) / kBlkH;
  unsigned hw = std::thread::hardware_concurrency();
  const int64_t num_edges = num_nodes > 0 ? nodePointer[num_nodes] : 0;
  int nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(hw ? hw : 1, std::max<int64_t>(1, num_edges / 65536)));

  std::atomic<int64_t> next{0};
  constexpr int64_t kChunk = 64;  // windows claimed per grab (dynamic balance for skewed graphs)
  auto worker = [&]() {
    std::vector<uint32_t> nb;
    for (;;) {
      const int64_t w0 = next.fetch_add(kChunk);
      if (w0 >= num_windows) break;
      const int64_t w1 = std::min(num_windows, w0 + kChunk);
      for (int64_t w = w0; w < w1; ++w) {
        const int64_t row0 = w * kBlkH;
        const int64_t row1 = std::min<int64_t>(row0 + kBlkH, num_nodes);
        for (int64_t r = row0; r < row1; ++r)
          for (int32_t e = nodePointer[r]; e < nodePointer[r + 1]; ++e) edgeToRow[e] = (int32_t)r;
        const int32_t lo = nodePointer[row0], hi = nodePointer[row1];
        if (hi == lo) {
          blockPartition[w] = 1;
          continue;
        }
        nb.assign(reinterpret_cast<const uint32_t*>(edgeList) + lo, reinterpret_cast<const uint32_t*>(edgeList) + hi);
        std::sort(nb.begin(), nb.end());
        const size_t u = std::unique(nb.begin(), nb.end()) - nb.begin();
        blockPartition[w] = (int32_t)((u + kBlkW - 1) / kBlkW);
        for (int32_t e = lo; e < hi; ++e)
          edgeToColumn[e] =
              (int32_t)(std::lower_bound(nb.begin(), nb.begin() + u, (uint32_t)edgeList[e]) - nb.begin());
      }
    }
  };
  if (nthreads <= 1) {
    worker();
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
  }

  int64_t acc = 0;
  Pointer1[0] = 0;
  for (int64_t w = 0; w < num_windows; ++w) {
    acc += blockPartition[w];
    if (acc > INT32_MAX) return kErrOverflow;  // the handle stores int32 block offsets
    Pointer1[w + 1] = (int32_t)acc;
  }
  return kOk;
}

// ----------------------------------------------------------------------------------------------
// Device: dense 0/1 tiles + per-tile column map (reference hmat_cuda_kernel, :21-111).
static __global__ __launch_bounds__(256) void hmat_scatter_kernel(const int* __restrict__ edgeList,
                                                           const int* __restrict__ edgeToColumn,
                                                           const int* __restrict__ edgeToRow,
                                                           const int* __restrict__ Pointer1, const int64_t numEdges,
                                                           float* __restrict__ hspa, int* __restrict__ hind) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < numEdges; e += stride) {
    const int row = edgeToRow[e];
    const int col = edgeToColumn[e];
    const int64_t blk = (int64_t)Pointer1[row / kBlkH] + col / kBlkW;  // 64-bit: ref overflows at :52
    const int cl = col % kBlkW;
    hspa[blk * (kBlkH * kBlkW) + (row % kBlkH) * kBlkW + cl] = 1.0f;  // ref :100-102
    hind[blk * kBlkW + cl] = edgeList[e];                              // ref :103-105
  }
}

// Zero-fill of hspa / hind (reference :71-79 does it inside hmat_cuda_kernel).  The block count is read from
// Pointer1[num_row_windows] ON THE DEVICE so that the launcher needs no host copy (stream-capturable).
static __global__ __launch_bounds__(256) void hmat_zero_kernel(const int* __restrict__ Pointer1, const int num_row_windows,
                                                        float* __restrict__ hspa, int* __restrict__ hind) {
  const int64_t total_blocks = Pointer1[num_row_windows];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float4* hspa4 = reinterpret_cast<float4*>(hspa);  // torch allocations are >= 256-B aligned
  for (int64_t i = gid; i < total_blocks * (kBlkH * kBlkW / 4); i += stride) hspa4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  int4* hind4 = reinterpret_cast<int4*>(hind);
  for (int64_t i = gid; i < total_blocks * (kBlkW / 4); i += stride) hind4[i] = make_int4(0, 0, 0, 0);
}

inline int hmat_hip(const int32_t* nodePointer, const int32_t* edgeList, const int32_t* blockPartition,
                    const int32_t* edgeToColumn, const int32_t* edgeToRow, const int32_t* Pointer1,
                    int32_t num_row_windows, int num_nodes, int num_edges, float* hspa, int* hind,
                    hipStream_t stream) {
  (void)nodePointer;
  (void)blockPartition;
  if (num_row_windows < 0 || num_nodes < 0 || num_edges < 0) return kErrBadShape;
  if (num_row_windows == 0) return kOk;
  if (((uintptr_t)hspa & 15) || ((uintptr_t)hind & 15)) return kErrBadShape;
  hipLaunchKernelGGL(hmat_zero_kernel, dim3(256 * 8), dim3(256), 0, stream, Pointer1, (int)num_row_windows, hspa, hind);
  if (num_edges > 0) {
    const int threads = 256;
    const int64_t want = ((int64_t)num_edges + threads - 1) / threads;
    const int blocks = (int)std::min<int64_t>(want, 256 * 32);
    hipLaunchKernelGGL(hmat_scatter_kernel, dim3(blocks), dim3(threads), 0, stream, edgeList, edgeToColumn, edgeToRow,
                       Pointer1, (int64_t)num_edges, hspa, hind);
  }
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ----------------------------------------------------------------------------------------------
// Device: 128 floats -> 4 x uint32 in the reference's "swizzled" (mma.m16n8k8 A-fragment) bit order
// (reference hmat_convert_uint32_swizzle_cuda_kernel, :151-193):
//   word t, bit b  <=  hspa[(b>>2) + 8*(t&1)][(b&3) + 4*(t>>1)] != 0
__device__ __forceinline__ uint32_t compress_low_nibbles(uint64_t x) {
  x &= 0x0F0F0F0F0F0F0F0FULL;
  x = (x | (x >> 4)) & 0x00FF00FF00FF00FFULL;
  x = (x | (x >> 8)) & 0x0000FFFF0000FFFFULL;
  x = (x | (x >> 16)) & 0x00000000FFFFFFFFULL;
  return (uint32_t)x;
}

static __global__ __launch_bounds__(256) void hmat_pack_swizzle_kernel(const int* __restrict__ Pointer1,
                                                                const int num_row_windows,
                                                                const float* __restrict__ hspa,
                                                                uint32_t* __restrict__ hspa_packed) {
  const int64_t total_blocks = Pointer1[num_row_windows];
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / kWave;
  for (int64_t b = wave; b < total_blocks; b += nwaves) {
    const float* tile = hspa + b * (kBlkH * kBlkW);
    const float lo = tile[lane];        // rows 0-7 : element 8r+c at lane 8r+c
    const float hi = tile[64 + lane];   // rows 8-15
    const uint64_t mlo = __ballot(fabsf(lo - 0.0f) > 1e-5f);  // ref :186
    const uint64_t mhi = __ballot(fabsf(hi - 0.0f) > 1e-5f);
    if (lane == 0) {
      uint4 w;
      w.x = compress_low_nibbles(mlo);       // t=0: rows 0-7,  cols 0-3
      w.y = compress_low_nibbles(mhi);       // t=1: rows 8-15, cols 0-3
      w.z = compress_low_nibbles(mlo >> 4);  // t=2: rows 0-7,  cols 4-7
      w.w = compress_low_nibbles(mhi >> 4);  // t=3: rows 8-15, cols 4-7
      *reinterpret_cast<uint4*>(hspa_packed + b * 4) = w;
    }
  }
}

inline int hmat_packed_swizzle_hip(int32_t num_row_windows, const int32_t* Pointer1, const float* hspa,
                                   uint32_t* hspa_packed, hipStream_t stream) {
  if (num_row_windows < 0) return kErrBadShape;
  if (num_row_windows == 0) return kOk;
  if ((uintptr_t)hspa_packed & 15) return kErrBadShape;
  hipLaunchKernelGGL(hmat_pack_swizzle_kernel, dim3(256 * 16), dim3(256), 0, stream, Pointer1, (int)num_row_windows,
                     hspa, hspa_packed);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace voltrix
