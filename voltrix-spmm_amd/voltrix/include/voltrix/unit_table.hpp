// Voltrix-SpMM for MI355X (gfx950) -- builder of the window kernel's unit table (SpmmArgs::units, spmm_kernels.hpp).
//
// Integer work on the handle's blk_offsets, once per handle.  No reference counterpart (the reference's equal-work
// scheduler, spmm_kernels.cuh:499-540, is dead code); the table layout and its semantics are profiles/HISTORY.md section 3.2:
//   * a window of nst stages (a stage = 4 TC blocks = one MFMA K step) longer than L stages is cut into
//     k = ceil(nst / L) interleaved units, unit j = stages j, j + k, j + 2k, ...  (length ceil((nst - j) / k));
//   * L = max(8, floor(1.5 x the lower median of nst)) unless the caller gives one; handles of FEWER THAN 1024 WINDOWS
//     (round 5): at most max(8, ceil(S / 1024)), S = all stages -- a window is one wave's serial stream, and a few hundred
//     long windows of one length (ddi-like: 267 windows of ~115 stages, nothing above 1.5 x the median) leave three SIMDs of
//     four without a wave; cut to about one unit per SIMD of the chip the product runs 1.7-2.0 x faster
//     (profiles/r05/experiment_few_windows.log);
//   * units int32[U][4] = {window, j, k, slot}: the units of XCD x's window range, back to back, LONGEST FIRST (ties: window,
//     then j -- a stable order, so the table is a function of the handle).  The ranges: [x wpx, (x + 1) wpx), wpx =
//     ceil(W / 8), or -- round 4 -- the caller's xcd_ptr int32[9] (first window of every range; xcd_ptr[8] = W): ranges of
//     equal WORK instead of equal window counts (graphs with community structure: the k-steps / stages per row vary by
//     community, an equal-rows split left one XCD with 1.57 x the mean work on the reddit-size block model);
//     slot = index of the unit's partial tile (cut windows: consecutive slots in unit order) or -1;
//   * unit_ptr int32[9] = first unit of every XCD's range; cuts int32[C][4] = {window, first slot, k, 0} per cut window.
// Two phases around the one host sync the caller needs anyway (U and C size the outputs):
//   count  histogram of nst -> median -> L; k per window, totals, prefix sums (first unit / first slot / cut index)
//   fill   one sort key per unit (XCD, top - length), stable radix sort (rocPRIM, library plumbing), gather.
// voltrix/schedule.py::unit_table_torch is the torch-tensor restatement the tests compare this against, bit for bit.
#pragma once

#include <cstring>

#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>

#include <cstdint>

#include "voltrix/csr_preprocess.hpp"
#include "voltrix/traits.hpp"

namespace voltrix {

constexpr int kUtHistBins = 65536;  // nst values above are counted in the last bin (a median up there: L = 1.5 x 65536)
constexpr int kUtMinStages = 8;
constexpr int kUtFewWindows = 1024;  // handles with fewer windows are cut into about this many units (one per SIMD)

// header int32[8] (device; the caller reads it between the phases)
enum UnitTableHeader {
  kUtNumUnits = 0,
  kUtNumCuts = 1,
  kUtNumSlots = 2,
  kUtMaxUnitsPerXcd = 3,
  kUtMaxStages = 4,
  kUtTop = 5,        // longest unit, in stages
  kUtHeaderInts = 8
};

struct UtWorkspace {      // count-phase workspace (kept, untouched, for the fill phase)
  int* hist;              // [kUtHistBins + 1]
  int* stats;             // [16]: 0 max_stages, 1 top, 2 num_units, 3 num_cuts, 4 num_slots, 8..15 units per XCD
  int* k;                 // [W]
  int* kcut;              // [W]   k where k > 1, else 0
  int* cutflag;           // [W]
  int* first;             // [W + 1] exclusive prefix of k
  int* slot_first;        // [W + 1] exclusive prefix of kcut
  int* cut_pos;           // [W + 1] exclusive prefix of cutflag
  int* chunk_sums;        // scan scratch
  long long bytes;
};
inline UtWorkspace ut_workspace(void* base, int num_nodes) {
  const long long W = ((long long)num_nodes + kBlkH - 1) / kBlkH;
  const long long nchunks = (W + kScanChunk - 1) / kScanChunk + 1;
  char* p = static_cast<char*>(base);
  UtWorkspace ws;
  auto take = [&](long long ints) {
    int* q = reinterpret_cast<int*>(p);
    p += align16(4 * ints);
    return q;
  };
  ws.hist = take(kUtHistBins + 1);
  ws.stats = take(16);
  ws.k = take(W);
  ws.kcut = take(W);
  ws.cutflag = take(W);
  ws.first = take(W + 1);
  ws.slot_first = take(W + 1);
  ws.cut_pos = take(W + 1);
  ws.chunk_sums = take(nchunks);
  ws.bytes = p - static_cast<char*>(base);
  return ws;
}
inline long long unit_table_workspace_bytes(int num_nodes) { return ut_workspace(nullptr, num_nodes).bytes; }

struct UtFillWorkspace {  // fill-phase workspace, sized by the number of units
  int* unit_window;       // [U]
  uint32_t* keys_in;      // [U]
  uint32_t* keys_out;     // [U]
  uint32_t* vals_in;      // [U]
  uint32_t* vals_out;     // [U]
  void* sort_temp;
  long long sort_temp_bytes;
  long long bytes;
};
inline UtFillWorkspace ut_fill_workspace(void* base, long long num_units) {
  char* p = static_cast<char*>(base);
  UtFillWorkspace ws;
  auto take = [&](long long ints) {
    void* q = p;
    p += align16(4 * ints);
    return q;
  };
  ws.unit_window = static_cast<int*>(take(num_units));
  ws.keys_in = static_cast<uint32_t*>(take(num_units));
  ws.keys_out = static_cast<uint32_t*>(take(num_units));
  ws.vals_in = static_cast<uint32_t*>(take(num_units));
  ws.vals_out = static_cast<uint32_t*>(take(num_units));
  ws.sort_temp = p;
  // rocPRIM's temporary storage: alternate key / value buffers + digit histograms; its exact size needs a device to ask
  // for, so the workspace reserves a bound and the fill launch checks the real figure against it
  ws.sort_temp_bytes = align16(8 * num_units) + (8ll << 20);
  p += ws.sort_temp_bytes;
  ws.bytes = p - static_cast<char*>(base);
  return ws;
}
inline long long unit_table_fill_workspace_bytes(long long num_units) {
  return ut_fill_workspace(nullptr, num_units < 0 ? 0 : num_units).bytes;
}

__device__ __forceinline__ int ut_stages(const int* __restrict__ blk_offsets, const int w) {
  return (blk_offsets[w + 1] - blk_offsets[w] + kTcbPerStage - 1) / kTcbPerStage;
}
// XCD range of window w: the caller's boundaries (xcd_ptr[x] <= w < xcd_ptr[x + 1]) or the equal split
__device__ __forceinline__ int ut_xcd_of(const int* __restrict__ xcd_ptr, const int windows_per_xcd, const int w) {
  if (xcd_ptr == nullptr) return w / windows_per_xcd;
  int x = 0;
#pragma unroll
  for (int i = 1; i < kNumXcd; ++i) x += w >= xcd_ptr[i] ? 1 : 0;
  return x;
}

static __global__ __launch_bounds__(256) void ut_hist_kernel(const int* __restrict__ blk_offsets, const int num_windows,
                                                             int* __restrict__ hist) {
  for (int w = blockIdx.x * 256 + threadIdx.x; w < num_windows; w += gridDim.x * 256) {
    const int nst = ut_stages(blk_offsets, w);
    atomicAdd(&hist[nst < kUtHistBins ? nst : kUtHistBins], 1);
  }
}

// stats[0] = L: the caller's max_stages, or max(8, floor(1.5 x lower median of nst)) (torch.median = element (W-1)/2 of
// the sorted values), capped at max(8, ceil(S / kUtFewWindows)) on handles of fewer than kUtFewWindows windows (S = sum of
// min(nst, kUtHistBins)).  One workgroup walks the histogram.
static __global__ __launch_bounds__(1024) void ut_median_kernel(const int* __restrict__ hist, const int num_windows,
                                                                const int max_stages_arg, int* __restrict__ stats) {
  __shared__ int wsum[16];
  __shared__ int carry_s, median_s;
  __shared__ unsigned long long stages_s;
  if (threadIdx.x == 0) {
    carry_s = 0;
    median_s = 0;
    stages_s = 0ull;
  }
  __syncthreads();
  if (max_stages_arg <= 0) {
    const int target = (num_windows - 1) / 2;  // 0-based rank of the lower median
    unsigned long long mine = 0ull;            // this thread's share of S (fewer than 2^10 windows of at most 2^16 stages)
    for (int base = 0; base <= kUtHistBins; base += 1024) {
      const int i = base + threadIdx.x;
      const int v = i <= kUtHistBins ? hist[i] : 0;
      mine += (unsigned long long)v * (unsigned long long)i;
      const int inc = wave_inclusive_scan(v);
      if ((threadIdx.x & (kWave - 1)) == kWave - 1) wsum[threadIdx.x / kWave] = inc;
      __syncthreads();
      int woff = 0, tot = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        woff += q < (int)(threadIdx.x / kWave) ? wsum[q] : 0;
        tot += wsum[q];
      }
      const int before = carry_s + woff + inc - v;  // values below bin i
      if (v > 0 && before <= target && target < before + v) median_s = i;
      __syncthreads();
      if (threadIdx.x == 0) carry_s += tot;
      __syncthreads();
    }
    if (num_windows < kUtFewWindows && mine) atomicAdd(&stages_s, mine);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int L = max_stages_arg;
    if (L <= 0) {
      const long long m = median_s;
      L = (int)((3 * m) / 2);
      if (L < kUtMinStages) L = kUtMinStages;
      if (num_windows < kUtFewWindows) {
        long long few = ((long long)stages_s + kUtFewWindows - 1) / kUtFewWindows;
        if (few < kUtMinStages) few = kUtMinStages;
        if (few < L) L = (int)few;
      }
    }
    stats[0] = L;
  }
}

static __global__ __launch_bounds__(256) void ut_cut_kernel(const int* __restrict__ blk_offsets, const int num_windows,
                                                            const int windows_per_xcd, const int* __restrict__ xcd_ptr,
                                                            int* __restrict__ stats,
                                                            int* __restrict__ k_out, int* __restrict__ kcut_out,
                                                            int* __restrict__ cutflag_out) {
  const int L = stats[0];
  for (int w = blockIdx.x * 256 + threadIdx.x; w < num_windows; w += gridDim.x * 256) {
    const int nst = ut_stages(blk_offsets, w);
    int k = (int)(((long long)nst + L - 1) / L);
    if (k < 1) k = 1;
    k_out[w] = k;
    kcut_out[w] = k > 1 ? k : 0;
    cutflag_out[w] = k > 1 ? 1 : 0;
    atomicMax(&stats[1], (nst + k - 1) / k);   // top: the longest unit (unit 0 of its window)
    atomicAdd(&stats[8 + ut_xcd_of(xcd_ptr, windows_per_xcd, w)], k);
  }
}

static __global__ void ut_header_kernel(const int* __restrict__ stats, const int* __restrict__ first,
                                        const int* __restrict__ slot_first, const int* __restrict__ cut_pos,
                                        const int num_windows, int* __restrict__ header) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int mx = 0;
    for (int x = 0; x < kNumXcd; ++x) mx = stats[8 + x] > mx ? stats[8 + x] : mx;
    header[kUtNumUnits] = first[num_windows];
    header[kUtNumCuts] = cut_pos[num_windows];
    header[kUtNumSlots] = slot_first[num_windows];
    header[kUtMaxUnitsPerXcd] = mx;
    header[kUtMaxStages] = stats[0];
    header[kUtTop] = stats[1];
    header[6] = 0;
    header[7] = 0;
  }
}

inline int ut_exclusive_scan(const int* in, int n, int* chunk_sums, int* out, hipStream_t stream) {
  const int nchunks = (n + kScanChunk - 1) / kScanChunk;
  hipLaunchKernelGGL(scan_chunk_sums_kernel, dim3(nchunks), dim3(256), 0, stream, in, n, chunk_sums);
  hipLaunchKernelGGL(scan_chunk_offsets_kernel, dim3(1), dim3(256), 0, stream, chunk_sums, nchunks);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nchunks), dim3(256), 0, stream, in, n, chunk_sums, out);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

inline int unit_table_check(int num_nodes) {
  if (num_nodes < 0 || num_nodes > (1 << 28)) return kErrBadShape;
  return kOk;
}

// Phase 1: header[8] <- {U, C, slots, max units per XCD, L, top, 0, 0}.  max_stages <= 0: the default bound.
inline int unit_table_count(const int* blk_offsets, int num_nodes, int max_stages, void* workspace, int* header,
                            hipStream_t stream, const int* xcd_ptr = nullptr /* device int32[9] or NULL: equal split */) {
  if (int rc = unit_table_check(num_nodes)) return rc;
  if (((uintptr_t)workspace & 15) || header == nullptr) return kErrBadShape;
  if (hipMemsetAsync(header, 0, kUtHeaderInts * sizeof(int), stream) != hipSuccess) return kErrLaunch;
  const int W = (num_nodes + kBlkH - 1) / kBlkH;
  if (W == 0) return kOk;
  const UtWorkspace ws = ut_workspace(workspace, num_nodes);
  if (hipMemsetAsync(ws.hist, 0, (size_t)((char*)ws.k - (char*)ws.hist), stream) != hipSuccess) return kErrLaunch;  // hist + stats
  const int grid = (W + 255) / 256 < 4096 ? (W + 255) / 256 : 4096;
  if (max_stages <= 0) hipLaunchKernelGGL(ut_hist_kernel, dim3(grid), dim3(256), 0, stream, blk_offsets, W, ws.hist);
  hipLaunchKernelGGL(ut_median_kernel, dim3(1), dim3(1024), 0, stream, ws.hist, W, max_stages, ws.stats);
  const int wpx = (W + kNumXcd - 1) / kNumXcd;
  hipLaunchKernelGGL(ut_cut_kernel, dim3(grid), dim3(256), 0, stream, blk_offsets, W, wpx, xcd_ptr, ws.stats, ws.k, ws.kcut,
                     ws.cutflag);
  if (int rc = ut_exclusive_scan(ws.k, W, ws.chunk_sums, ws.first, stream)) return rc;
  if (int rc = ut_exclusive_scan(ws.kcut, W, ws.chunk_sums, ws.slot_first, stream)) return rc;
  if (int rc = ut_exclusive_scan(ws.cutflag, W, ws.chunk_sums, ws.cut_pos, stream)) return rc;
  hipLaunchKernelGGL(ut_header_kernel, dim3(1), dim3(64), 0, stream, ws.stats, ws.first, ws.slot_first, ws.cut_pos, W,
                     header);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// one thread per window: its units' sort keys, and the cut record of a cut window
static __global__ __launch_bounds__(256) void ut_emit_kernel(const int* __restrict__ blk_offsets, const int num_windows,
                                                             const int windows_per_xcd, const int* __restrict__ xcd_ptr,
                                                             const int* __restrict__ stats,
                                                             const int* __restrict__ k_in, const int* __restrict__ first,
                                                             const int* __restrict__ slot_first,
                                                             const int* __restrict__ cut_pos,
                                                             int* __restrict__ unit_window, uint32_t* __restrict__ keys,
                                                             uint32_t* __restrict__ vals, int4* __restrict__ cuts) {
  const int top = stats[1];
  for (int w = blockIdx.x * 256 + threadIdx.x; w < num_windows; w += gridDim.x * 256) {
    const int nst = ut_stages(blk_offsets, w);
    const int k = k_in[w];
    const int u0 = first[w];
    const uint32_t group = (uint32_t)ut_xcd_of(xcd_ptr, windows_per_xcd, w) * (uint32_t)(top + 1);
    for (int j = 0; j < k; ++j) {
      const int length = (nst - j + k - 1) / k;
      unit_window[u0 + j] = w;
      keys[u0 + j] = group + (uint32_t)(top - length);
      vals[u0 + j] = (uint32_t)(u0 + j);
    }
    if (k > 1) cuts[cut_pos[w]] = make_int4(w, slot_first[w], k, 0);
  }
}

static __global__ __launch_bounds__(256) void ut_gather_kernel(const int num_units, const uint32_t* __restrict__ order,
                                                               const int* __restrict__ unit_window,
                                                               const int* __restrict__ k_in, const int* __restrict__ first,
                                                               const int* __restrict__ slot_first,
                                                               int4* __restrict__ units) {
  for (int p = blockIdx.x * 256 + threadIdx.x; p < num_units; p += gridDim.x * 256) {
    const int u = (int)order[p];
    const int w = unit_window[u];
    const int j = u - first[w];
    const int k = k_in[w];
    units[p] = make_int4(w, j, k, k > 1 ? slot_first[w] + j : -1);
  }
}

static __global__ void ut_ptr_kernel(const int* __restrict__ stats, int* __restrict__ unit_ptr) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int run = 0;
    unit_ptr[0] = 0;
    for (int x = 0; x < kNumXcd; ++x) {
      run += stats[8 + x];
      unit_ptr[x + 1] = run;
    }
  }
}

// Phase 2 (same workspace, untouched since phase 1; num_units / num_cuts / top = the header the caller read).
// units int32[U][4], unit_ptr int32[9], cuts int32[C][4]: every element is written.
inline int unit_table_fill(const int* blk_offsets, int num_nodes, void* workspace, void* fill_workspace, int num_units,
                           int num_cuts, int top, int* units, int* unit_ptr, int* cuts, hipStream_t stream,
                           const int* xcd_ptr = nullptr /* the same boundaries as in phase 1 */) {
  if (int rc = unit_table_check(num_nodes)) return rc;
  if (((uintptr_t)workspace & 15) || ((uintptr_t)fill_workspace & 15) || num_units < 0 || num_cuts < 0 || top < 0)
    return kErrBadShape;
  if (((uintptr_t)units & 15) || ((uintptr_t)cuts & 15)) return kErrBadShape;
  const int W = (num_nodes + kBlkH - 1) / kBlkH;
  if (W == 0 || num_units == 0)
    return hipMemsetAsync(unit_ptr, 0, (kNumXcd + 1) * sizeof(int), stream) == hipSuccess ? kOk : kErrLaunch;
  const UtWorkspace ws = ut_workspace(workspace, num_nodes);
  const UtFillWorkspace fw = ut_fill_workspace(fill_workspace, num_units);
  const int wpx = (W + kNumXcd - 1) / kNumXcd;
  const int grid = (W + 255) / 256 < 4096 ? (W + 255) / 256 : 4096;
  hipLaunchKernelGGL(ut_emit_kernel, dim3(grid), dim3(256), 0, stream, blk_offsets, W, wpx, xcd_ptr, ws.stats, ws.k, ws.first,
                     ws.slot_first, ws.cut_pos, fw.unit_window, fw.keys_in, fw.vals_in, reinterpret_cast<int4*>(cuts));
  // stable sort by (XCD, top - length): 3 bits of XCD above the bits of top
  int bits = 1;
  while ((1ll << bits) < (long long)kNumXcd * ((long long)top + 1)) ++bits;
  if (bits > 32) return kErrOverflow;
  size_t temp_bytes = 0;
  if (rocprim::radix_sort_pairs(nullptr, temp_bytes, fw.keys_in, fw.keys_out, fw.vals_in, fw.vals_out,
                                (unsigned)num_units, 0u, (unsigned)bits, stream) != hipSuccess)
    return kErrLaunch;
  if ((long long)temp_bytes > fw.sort_temp_bytes) return kErrBadConfig;
  if (rocprim::radix_sort_pairs(fw.sort_temp, temp_bytes, fw.keys_in, fw.keys_out, fw.vals_in, fw.vals_out,
                                (unsigned)num_units, 0u, (unsigned)bits, stream) != hipSuccess)
    return kErrLaunch;
  const int ugrid = (num_units + 255) / 256 < 4096 ? (num_units + 255) / 256 : 4096;
  hipLaunchKernelGGL(ut_gather_kernel, dim3(ugrid), dim3(256), 0, stream, num_units, fw.vals_out, fw.unit_window, ws.k,
                     ws.first, ws.slot_first, reinterpret_cast<int4*>(units));
  hipLaunchKernelGGL(ut_ptr_kernel, dim3(1), dim3(64), 0, stream, ws.stats, unit_ptr);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace voltrix
