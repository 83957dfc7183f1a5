// Voltrix-SpMM for MI355X (gfx950) -- builder of the one-launch kernel's residual stage records (integer / byte work).
//
// Block-format handle of the RESIDUAL matrix (blk_offsets, hspa_packed, hind) -> (wave_ptr, records), bit-identical to the
// plain-loop definition in oracle/oracle_np.py::fused_records (layout: spmm_fused_kernels.hpp, include/voltrix_capi.h).
// Two phases around the one host sync the caller needs anyway (the record count sizes the output):
//   count  one thread per (panel, wave): stages of its kFusedRowBlocks (8) windows (a window whose only TC block is all zero -- the
//          reference's empty-window quirk -- has none), then an exclusive scan -> wave_ptr
//   fill   one wave64 per (panel, wave): its windows' stages merged by first column (ties: lower row block), one 256-byte
//          record per step written by the 64 lanes: 32 rows of B (columns nobody references and blocks past the window's end
//          repeat the window's first column), 16 bitmap words, the row block.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "voltrix/csr_preprocess.hpp"
#include "voltrix/spmm_fused_kernels.hpp"

namespace voltrix {

__device__ __forceinline__ int fused_window_stages(const int* __restrict__ blk_offsets,
                                                   const uint32_t* __restrict__ hspa_packed, const int w) {
  const int kb0 = blk_offsets[w], nblk = blk_offsets[w + 1] - kb0;
  if (nblk == 1) {
    const uint4 z = *reinterpret_cast<const uint4*>(hspa_packed + 4ll * kb0);
    if ((z.x | z.y | z.z | z.w) == 0u) return 0;
  }
  return (nblk + kTcbPerStage - 1) / kTcbPerStage;
}

static __global__ __launch_bounds__(256) void fused_records_count_kernel(const int* __restrict__ blk_offsets,
                                                                        const uint32_t* __restrict__ hspa_packed,
                                                                        const int num_windows, const int num_waves,
                                                                        int* __restrict__ counts) {
  const int gw = blockIdx.x * 256 + threadIdx.x;
  if (gw >= num_waves) return;
  int c = 0;
  for (int j = 0; j < kFusedRowBlocks; ++j) {
    const int w = kFusedRowBlocks * gw + j;
    if (w < num_windows) c += fused_window_stages(blk_offsets, hspa_packed, w);
  }
  counts[gw] = c;
}

static __global__ __launch_bounds__(256) void fused_records_fill_kernel(const int* __restrict__ blk_offsets,
                                                                       const uint32_t* __restrict__ hspa_packed,
                                                                       const int* __restrict__ hind, const int num_windows,
                                                                       const int num_waves, const int* __restrict__ wave_ptr,
                                                                       uint32_t* __restrict__ records) {
  const int lane = threadIdx.x & (kWave - 1);
  const int gw = blockIdx.x * (256 / kWave) + (int)(threadIdx.x / kWave);
  if (gw >= num_waves) return;   // wave-uniform
  int kb0[kFusedRowBlocks], kb1[kFusedRowBlocks], next[kFusedRowBlocks], left[kFusedRowBlocks], safe[kFusedRowBlocks];
#pragma unroll
  for (int j = 0; j < kFusedRowBlocks; ++j) {
    const int w = kFusedRowBlocks * gw + j;
    kb0[j] = kb1[j] = next[j] = left[j] = safe[j] = 0;
    if (w < num_windows) {
      kb0[j] = blk_offsets[w];
      kb1[j] = blk_offsets[w + 1];
      left[j] = fused_window_stages(blk_offsets, hspa_packed, w);
      next[j] = kb0[j];
      safe[j] = hind[8ll * kb0[j]];
    }
  }
  long long rec = wave_ptr[gw];
  const long long end = wave_ptr[gw + 1];
  for (; rec < end; ++rec) {
    // the window whose next stage starts at the smallest column (ties: the lower row block)
    int best = -1, best_col = 0;
#pragma unroll
    for (int j = 0; j < kFusedRowBlocks; ++j) {
      if (left[j] > 0) {
        const int col = hind[8ll * next[j]];
        if (best < 0 || col < best_col) {
          best = j;
          best_col = col;
        }
      }
    }
    int sb = 0, wkb1 = 0, wsafe = 0;
#pragma unroll
    for (int j = 0; j < kFusedRowBlocks; ++j) {
      if (j == best) {
        sb = next[j];
        wkb1 = kb1[j];
        wsafe = safe[j];
        next[j] += kTcbPerStage;
        left[j] -= 1;
      }
    }
    uint32_t word = 0u;
    if (lane < 32) {
      const int blk = sb + (lane >> 3), c = lane & 7;
      bool used = false;
      if (blk < wkb1) {
        const uint32_t mask = 0x11111111u << (c & 3);
        used = ((hspa_packed[4ll * blk + 2 * (c >> 2)] | hspa_packed[4ll * blk + 2 * (c >> 2) + 1]) & mask) != 0u;
      }
      word = (uint32_t)(used ? hind[8ll * blk + c] : wsafe);
    } else if (lane < 48) {
      const int t = lane - 32, blk = sb + (t >> 2);
      word = blk < wkb1 ? hspa_packed[4ll * blk + (t & 3)] : 0u;
    } else if (lane == kRecordBlockWord) {
      word = (uint32_t)best;
    }
    records[rec * kRecordWords + lane] = word;
  }
}

inline long long fused_records_workspace_bytes(int num_nodes) {
  if (num_nodes < 0) return 0;
  const long long waves = (long long)kFusedWaves * ((num_nodes + kFusedPanelRows - 1) / kFusedPanelRows);
  return align16(4 * (waves + 1)) + align16(4 * ((waves + kScanChunk - 1) / kScanChunk + 1));
}

// Phase 1: wave_ptr int32 [8 NP + 1]; the caller reads R = wave_ptr[8 NP] and allocates records uint32 [(R + 1) * 64].
inline int fused_records_count(const int* blk_offsets, const uint32_t* hspa_packed, int num_nodes, void* workspace,
                               int* wave_ptr, hipStream_t stream) {
  if (num_nodes < 0 || ((uintptr_t)workspace & 15) || ((uintptr_t)hspa_packed & 15)) return kErrBadShape;
  const int num_windows = (num_nodes + kBlkH - 1) / kBlkH;
  const int num_waves = kFusedWaves * ((num_nodes + kFusedPanelRows - 1) / kFusedPanelRows);
  if (hipMemsetAsync(wave_ptr, 0, sizeof(int), stream) != hipSuccess) return kErrLaunch;
  if (num_waves == 0) return kOk;
  int* counts = static_cast<int*>(workspace);
  int* chunk_sums = reinterpret_cast<int*>(static_cast<char*>(workspace) + align16(4ll * (num_waves + 1)));
  hipLaunchKernelGGL(fused_records_count_kernel, dim3((num_waves + 255) / 256), dim3(256), 0, stream, blk_offsets,
                     hspa_packed, num_windows, num_waves, counts);
  const int nchunks = (num_waves + kScanChunk - 1) / kScanChunk;
  hipLaunchKernelGGL(scan_chunk_sums_kernel, dim3(nchunks), dim3(256), 0, stream, counts, num_waves, chunk_sums);
  hipLaunchKernelGGL(scan_chunk_offsets_kernel, dim3(1), dim3(256), 0, stream, chunk_sums, nchunks);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nchunks), dim3(256), 0, stream, counts, num_waves, chunk_sums, wave_ptr);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// Phase 2: every word of records [(R + 1) * 64] is written (the padding record is zero).
inline int fused_records_fill(const int* blk_offsets, const uint32_t* hspa_packed, const int* hind, int num_nodes,
                              const int* wave_ptr, long long num_records, uint32_t* records, hipStream_t stream) {
  if (num_nodes < 0 || num_records < 0 || ((uintptr_t)records & 15) || ((uintptr_t)hspa_packed & 15)) return kErrBadShape;
  if (hipMemsetAsync(records + num_records * kRecordWords, 0, kRecordBytes, stream) != hipSuccess) return kErrLaunch;
  const int num_windows = (num_nodes + kBlkH - 1) / kBlkH;
  const int num_waves = kFusedWaves * ((num_nodes + kFusedPanelRows - 1) / kFusedPanelRows);
  if (num_waves == 0 || num_records == 0) return kOk;
  hipLaunchKernelGGL(fused_records_fill_kernel, dim3((num_waves + 3) / 4), dim3(256), 0, stream, blk_offsets, hspa_packed,
                     hind, num_windows, num_waves, wave_ptr, records);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace voltrix
