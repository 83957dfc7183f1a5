// Voltrix-SpMM for MI355X (gfx950) -- schedules as data (round 4): the XCD ranges of equal work and the panel kernel's piece
// table, built on the device (integer work, once per handle; the entry points a C host binds:
// voltrix_launch_xcd_ranges_* / voltrix_launch_panel_parts_*).  voltrix/schedule.py::split_equal_work and
// voltrix/hybrid.py::panel_parts are the torch-tensor restatements the tests compare these with, element by element.
//
//   XCD ranges    xcd_ptr int32[9]: b[0] = 0 <= b[1] <= ... <= b[8] = n, the eight ranges of `work` with about equal sums:
//                 b[x] = the index whose prefix sum is NEAREST to ceil(total * x / 8) (the item that crosses the target goes to
//                 the side that leaves the smaller error), rounded up to `align`, made monotone.  total = 0: ranges of
//                 ceil(n / 8) items.  Work of a window = its stages (4 TC blocks each); work of a panel = kstep_cost_x10 / 10 x
//                 its k-steps + the stages of its windows (a k-step of the panel kernel costs a CU about 6.6 residual stages).
//   piece table   panels of more than `cap` k-steps cut into ceil(k-steps / cap) contiguous pieces of nearly equal length;
//                 parts int32[P][4] = {panel, first k-step inside the panel, k-steps, slot}, per XCD range longest first (ties:
//                 panel, then piece index); slot = -1 for whole panels, else consecutive per cut panel in k-step order;
//                 cuts int32[C][4] = {panel, first slot, pieces, 0}; part_xcd_ptr int32[9] over the positions of `parts`.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "voltrix/spmm_kernels.hpp"

namespace voltrix {

constexpr int kSplitThreads = 1024;
constexpr int kSplitUnroll = 8;      // 64-item steps a wave loads before it sums them (memory-level parallelism of the walk)

// work of item i: a functor over the raw arrays, so that windows and panels need no materialised work array
struct WorkOfArray {
  const int* work;
  __device__ long long operator()(int i) const { return work[i]; }
};
struct WorkOfWindows {   // stages of window i
  const int* blk_offsets;
  __device__ long long operator()(int i) const { return (blk_offsets[i + 1] - blk_offsets[i] + 3) / 4; }
};
struct WorkOfPanels {    // round(kstep_cost x k-steps + stages of the panel's windows), cost given in tenths
  const int* panel_ptr;
  const int* blk_offsets;
  int windows_per_panel, num_windows, kstep_cost_x10;
  __device__ long long operator()(int p) const {
    const int w0 = p * windows_per_panel;
    int w1 = w0 + windows_per_panel;
    w1 = w1 < num_windows ? w1 : num_windows;
    long long stages = 0;
    for (int w = w0; w < w1; ++w) stages += (blk_offsets[w + 1] - blk_offsets[w] + 3) / 4;
    const long long k = panel_ptr[p + 1] - panel_ptr[p];
    // round-half-to-even of k * cost / 10 + stages, as torch.round on the float64 value (exact: the product is an integer
    // number of tenths)
    const long long tenths = k * kstep_cost_x10;
    long long q = tenths / 10;
    const long long r = tenths % 10;
    if (r > 5 || (r == 5 && (q & 1))) ++q;
    return q + stages;
  }
};

// One workgroup of 16 waves; every WAVE owns a contiguous sixteenth of the items and walks it 64 items per step, lane l
// taking item base + l (coalesced 256-byte loads; a thread-contiguous walk would keep 1024 cache lines alive at once).
// Pass 1: per-wave sums -> exclusive prefix over the waves (thread 0: 16 adds) -> total and the seven targets.
// Pass 2: the wave whose range holds the first index whose prefix reaches a target walks it again with a wave-wide inclusive
// scan per step; the lane at the crossing writes the boundary.  Thread 0 rounds, clamps and makes the list monotone.
// Measured (one CU, latency-bound by design -- the table is built once per handle): 14.6 k windows 0.02 ms, 1 M 0.38 ms,
// 6.95 M (the papers-like graph) 4.7 ms.
__device__ __forceinline__ long long wave_inclusive_scan(long long v, const int lane) {
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const long long o = __shfl_up(v, off, kWave);
    if (lane >= off) v += o;
  }
  return v;
}

template <class Work>
static __global__ __launch_bounds__(kSplitThreads) void split_equal_work_kernel(const Work work, const int n, const int align,
                                                                               int* __restrict__ xcd_ptr) {
  constexpr int kWaves = kSplitThreads / kWave;
  __shared__ long long wave_sum[kWaves + 1];
  __shared__ long long target[kNumXcd];
  __shared__ int cut[kNumXcd + 1];
  const int t = threadIdx.x, wave = t / kWave, lane = t % kWave;
  const long long per = ((long long)(n + kWaves - 1) / kWaves + kWave - 1) / kWave * kWave;   // a multiple of 64
  const int i0 = (int)(wave * per < n ? wave * per : n);
  const int i1 = (int)(i0 + per < n ? i0 + per : n);
  long long s = 0;
  for (int base = i0; base < i1; base += kWave * kSplitUnroll) {     // kSplitUnroll independent loads in flight per lane
    long long v[kSplitUnroll];
#pragma unroll
    for (int u = 0; u < kSplitUnroll; ++u) {
      const int i = base + u * kWave + lane;
      v[u] = i < i1 ? work(i) : 0;
    }
#pragma unroll
    for (int u = 0; u < kSplitUnroll; ++u) s += v[u];
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, kWave);
  if (lane == 0) wave_sum[wave] = s;
  if (t < kNumXcd + 1) cut[t] = t == kNumXcd ? n : 0;
  __syncthreads();
  if (t == 0) {
    long long run = 0;
    for (int k = 0; k < kWaves; ++k) {
      const long long v = wave_sum[k];
      wave_sum[k] = run;
      run += v;
    }
    wave_sum[kWaves] = run;
    for (int x = 1; x < kNumXcd; ++x) target[x] = (run * x + kNumXcd - 1) / kNumXcd;
  }
  __syncthreads();
  const long long total = wave_sum[kWaves];
  if (total > 0) {
    const long long before = wave_sum[wave], after = wave_sum[wave + 1];
    for (int x = 1; x < kNumXcd; ++x) {           // wave-uniform
      const long long tg = target[x];
      if (!(before < tg && tg <= after)) continue;   // the first index whose prefix reaches tg is not in this wave's range
      long long run = before;
      bool found = false;
      for (int base = i0; base < i1 && !found; base += kWave * kSplitUnroll) {   // wave-uniform
        long long v[kSplitUnroll], group = 0;
#pragma unroll
        for (int u = 0; u < kSplitUnroll; ++u) {
          const int i = base + u * kWave + lane;
          v[u] = i < i1 ? work(i) : 0;
          group += v[u];
        }
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) group += __shfl_xor(group, off, kWave);
        if (run + group < tg) {          // the crossing is not in these kSplitUnroll x 64 items
          run += group;
          continue;
        }
#pragma unroll
        for (int u = 0; u < kSplitUnroll; ++u) {
          if (found) continue;
          const int i = base + u * kWave + lane;
          const long long cur = run + wave_inclusive_scan(v[u], lane);
          const unsigned long long hits = __ballot(i < i1 && cur >= tg);
          if (hits) {
            if (lane == __ffsll((long long)hits) - 1) {
              const long long under = tg - (cur - v[u]), over = cur - tg;
              cut[x] = under < over ? i : i + 1;         // items before the boundary
            }
            found = true;
          }
          run = __shfl(cur, kWave - 1, kWave);
        }
      }
    }
  }
  __syncthreads();
  if (t == 0) {
    const int per_xcd = (n + kNumXcd - 1) / kNumXcd;
    int top = 0;
    xcd_ptr[0] = 0;
    for (int x = 1; x < kNumXcd; ++x) {
      long long c = total > 0 ? cut[x] : (long long)x * per_xcd;
      if (total > 0) c = (c + align - 1) / align * align;
      c = c < n ? c : n;
      top = (int)c > top ? (int)c : top;
      xcd_ptr[x] = top;
    }
    xcd_ptr[kNumXcd] = n;
  }
}

template <class Work>
inline int split_equal_work(const Work work, int n, int align, int* xcd_ptr, hipStream_t stream) {
  if (n < 0 || align < 1 || xcd_ptr == nullptr) return kErrBadShape;
  hipLaunchKernelGGL(split_equal_work_kernel<Work>, dim3(1), dim3(kSplitThreads), 0, stream, work, n, align, xcd_ptr);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

inline int xcd_ranges_of_work(const int* work, int n, int align, int* xcd_ptr, hipStream_t stream) {
  if (n > 0 && work == nullptr) return kErrBadShape;
  return split_equal_work(WorkOfArray{work}, n, align, xcd_ptr, stream);
}

inline int xcd_ranges_of_windows(const int* blk_offsets, int num_nodes, int align, int* xcd_ptr, hipStream_t stream) {
  if (num_nodes < 0 || (num_nodes > 0 && blk_offsets == nullptr)) return kErrBadShape;
  return split_equal_work(WorkOfWindows{blk_offsets}, (num_nodes + 15) / 16, align, xcd_ptr, stream);
}

// panel ranges (xcd_ptr) and the same ranges in windows (window_xcd_ptr = x windows_per_panel, clipped), in one call
static __global__ void window_ranges_kernel(const int* __restrict__ xcd_ptr, const int windows_per_panel, const int num_windows,
                                            int* __restrict__ window_xcd_ptr) {
  if (threadIdx.x <= kNumXcd) {
    const long long w = (long long)xcd_ptr[threadIdx.x] * windows_per_panel;
    window_xcd_ptr[threadIdx.x] = (int)(w < num_windows ? w : num_windows);
  }
}

inline int xcd_ranges_of_panels(const int* panel_ptr, const int* resid_blk_offsets, int num_nodes, int panel_rows,
                                int kstep_cost_x10, int* xcd_ptr, int* window_xcd_ptr, hipStream_t stream) {
  if (num_nodes < 0 || panel_rows < 16 || panel_rows % 16 != 0 || kstep_cost_x10 < 0) return kErrBadShape;
  if (num_nodes > 0 && (panel_ptr == nullptr || resid_blk_offsets == nullptr)) return kErrBadShape;
  const int num_panels = (num_nodes + panel_rows - 1) / panel_rows;
  const int num_windows = (num_nodes + 15) / 16;
  const int rc = split_equal_work(WorkOfPanels{panel_ptr, resid_blk_offsets, panel_rows / 16, num_windows, kstep_cost_x10},
                                  num_panels, 1, xcd_ptr, stream);
  if (rc != kOk || window_xcd_ptr == nullptr) return rc;
  hipLaunchKernelGGL(window_ranges_kernel, dim3(1), dim3(64), 0, stream, xcd_ptr, panel_rows / 16, num_windows, window_xcd_ptr);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- piece table ---------------------------------------------------------------------------------------------------------
// workspace: first int32[NP + 1] (first piece of panel p in natural order), slot_first int32[NP + 1], cut_index int32[NP + 1],
// part_xcd int32[9]
struct PartsWorkspace {
  int* first;
  int* slot_first;
  int* cut_index;
  int* part_xcd;
  long long bytes;
};

inline PartsWorkspace parts_workspace(void* base, int num_panels) {
  PartsWorkspace ws;
  char* p = static_cast<char*>(base);
  const long long one = (4ll * (num_panels + 1) + 15) / 16 * 16;
  ws.first = reinterpret_cast<int*>(p);
  p += one;
  ws.slot_first = reinterpret_cast<int*>(p);
  p += one;
  ws.cut_index = reinterpret_cast<int*>(p);
  p += one;
  ws.part_xcd = reinterpret_cast<int*>(p);
  p += 64;
  ws.bytes = p - static_cast<char*>(base);
  return ws;
}

inline long long panel_parts_workspace_bytes(int num_panels) {
  return parts_workspace(nullptr, num_panels < 0 ? 0 : num_panels).bytes;
}

__device__ __forceinline__ int parts_xcd_of(const int* __restrict__ panel_xcd_ptr, const int per_xcd, const int p) {
  if (panel_xcd_ptr == nullptr) return p / per_xcd;
  int x = 0;
  for (int i = 1; i < kNumXcd; ++i) x += p >= panel_xcd_ptr[i] ? 1 : 0;
  return x;
}

__device__ __forceinline__ int parts_pieces_of(const int nks, const int cap) {
  const int k = (nks + cap - 1) / cap;
  return k < 1 ? 1 : k;
}

// Phase 1 (one workgroup): pieces per panel -> three exclusive scans over the panels in natural order (first piece, first
// slot, index among the cut panels), pieces per XCD range.  header int32[8] = {pieces, cut panels, slots, longest range of
// pieces, cap, 0, 0, 0}.
static __global__ __launch_bounds__(kSplitThreads) void panel_parts_count_kernel(const int* __restrict__ panel_ptr,
                                                                                const int num_panels, const int cap,
                                                                                const int* __restrict__ panel_xcd_ptr,
                                                                                const PartsWorkspace ws, int* __restrict__ header) {
  __shared__ int sums[3][kSplitThreads + 1];
  __shared__ int per_range[kNumXcd];
  const int t = threadIdx.x;
  const int per = (num_panels + kSplitThreads - 1) / kSplitThreads;
  const int p0 = t * per < num_panels ? t * per : num_panels;
  const int p1 = p0 + per < num_panels ? p0 + per : num_panels;
  const int per_xcd = (num_panels + kNumXcd - 1) / kNumXcd > 0 ? (num_panels + kNumXcd - 1) / kNumXcd : 1;
  if (t < kNumXcd) per_range[t] = 0;
  __syncthreads();
  int pieces = 0, slots = 0, cuts = 0;
  for (int p = p0; p < p1; ++p) {
    const int k = parts_pieces_of(panel_ptr[p + 1] - panel_ptr[p], cap);
    pieces += k;
    slots += k > 1 ? k : 0;
    cuts += k > 1 ? 1 : 0;
    atomicAdd(&per_range[parts_xcd_of(panel_xcd_ptr, per_xcd, p)], k);
  }
  sums[0][t] = pieces;
  sums[1][t] = slots;
  sums[2][t] = cuts;
  __syncthreads();
  if (t < 3) {
    int run = 0;
    for (int k = 0; k < kSplitThreads; ++k) {
      const int v = sums[t][k];
      sums[t][k] = run;
      run += v;
    }
    sums[t][kSplitThreads] = run;
  }
  __syncthreads();
  int f = sums[0][t], s = sums[1][t], c = sums[2][t];
  for (int p = p0; p < p1; ++p) {
    const int k = parts_pieces_of(panel_ptr[p + 1] - panel_ptr[p], cap);
    ws.first[p] = f;
    ws.slot_first[p] = s;
    ws.cut_index[p] = c;
    f += k;
    s += k > 1 ? k : 0;
    c += k > 1 ? 1 : 0;
  }
  if (t == 0) {
    ws.first[num_panels] = sums[0][kSplitThreads];
    ws.slot_first[num_panels] = sums[1][kSplitThreads];
    ws.cut_index[num_panels] = sums[2][kSplitThreads];
    int run = 0, longest = 0;
    ws.part_xcd[0] = 0;
    for (int x = 0; x < kNumXcd; ++x) {
      longest = per_range[x] > longest ? per_range[x] : longest;
      run += per_range[x];
      ws.part_xcd[x + 1] = run;
    }
    header[0] = sums[0][kSplitThreads];
    header[1] = sums[2][kSplitThreads];
    header[2] = sums[1][kSplitThreads];
    header[3] = longest;
    header[4] = cap;
    header[5] = header[6] = header[7] = 0;
  }
}

__device__ __forceinline__ int parts_length(const int nks, const int k, const int j) { return nks / k + (j < nks % k ? 1 : 0); }

// Phase 2: one thread per (panel, piece).  Position inside the XCD range = pieces of the range that are longer, or as long
// and earlier in (panel, piece) order -- counted over the range's panels (NP / 8 of them, a handful of pieces each).
static __global__ __launch_bounds__(256) void panel_parts_fill_kernel(const int* __restrict__ panel_ptr, const int num_panels,
                                                                      const int cap, const int* __restrict__ panel_xcd_ptr,
                                                                      const PartsWorkspace ws, int* __restrict__ parts,
                                                                      int* __restrict__ part_xcd_ptr, int* __restrict__ cuts) {
  const int per_xcd = (num_panels + kNumXcd - 1) / kNumXcd > 0 ? (num_panels + kNumXcd - 1) / kNumXcd : 1;
  if (blockIdx.x == 0 && threadIdx.x <= kNumXcd) part_xcd_ptr[threadIdx.x] = ws.part_xcd[threadIdx.x];
  for (int p = blockIdx.x; p < num_panels; p += gridDim.x) {      // workgroup-uniform
    const int nks = panel_ptr[p + 1] - panel_ptr[p];
    const int k = parts_pieces_of(nks, cap);
    const int x = parts_xcd_of(panel_xcd_ptr, per_xcd, p);
    int lo = x * per_xcd, hi = lo + per_xcd < num_panels ? lo + per_xcd : num_panels;
    if (panel_xcd_ptr != nullptr) {
      lo = panel_xcd_ptr[x];
      hi = panel_xcd_ptr[x + 1];
    }
    if (threadIdx.x == 0 && k > 1) {
      int* c = cuts + 4 * (long long)ws.cut_index[p];
      c[0] = p;
      c[1] = ws.slot_first[p];
      c[2] = k;
      c[3] = 0;
    }
    for (int j = threadIdx.x; j < k; j += blockDim.x) {
      const int mine = parts_length(nks, k, j);
      int before = 0;
      for (int q = lo; q < hi; ++q) {
        const int qn = panel_ptr[q + 1] - panel_ptr[q];
        const int qk = parts_pieces_of(qn, cap);
        const int longer = qn % qk, base = qn / qk;       // pieces 0 .. longer-1 have base + 1 k-steps, the rest base
        // pieces of q ahead of (p, j): strictly longer ones, and equally long ones that come first in (panel, piece) order
        int ahead = 0;
        if (base + 1 > mine) ahead += longer;
        if (base > mine) ahead += qk - longer;
        if (base + 1 == mine && longer > 0) ahead += q < p ? longer : (q == p ? (j < longer ? j : longer) : 0);
        if (base == mine) ahead += q < p ? qk - longer : (q == p ? (j > longer ? j - longer : 0) : 0);
        before += ahead;
      }
      int* out = parts + 4 * (long long)(ws.part_xcd[x] + before);
      out[0] = p;
      out[1] = j * (nks / k) + (j < nks % k ? j : nks % k);
      out[2] = mine;
      out[3] = k > 1 ? ws.slot_first[p] + j : -1;
    }
  }
}

inline int panel_parts_check(int num_panels, int cap, const void* workspace) {
  if (num_panels < 0 || cap < 1 || ((uintptr_t)workspace & 15)) return kErrBadShape;
  return kOk;
}

inline int panel_parts_count(const int* panel_ptr, int num_panels, int cap, const int* panel_xcd_ptr, void* workspace,
                             int* header, hipStream_t stream) {
  if (const int rc = panel_parts_check(num_panels, cap, workspace)) return rc;
  if (header == nullptr || workspace == nullptr || (num_panels > 0 && panel_ptr == nullptr)) return kErrBadShape;
  hipLaunchKernelGGL(panel_parts_count_kernel, dim3(1), dim3(kSplitThreads), 0, stream, panel_ptr, num_panels, cap,
                     panel_xcd_ptr, parts_workspace(workspace, num_panels), header);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// Same panel_ptr / cap / panel_xcd_ptr / workspace as the count call (untouched since); parts int32[header[0]][4],
// part_xcd_ptr int32[9], cuts int32[max(1, header[1])][4].
inline int panel_parts_fill(const int* panel_ptr, int num_panels, int cap, const int* panel_xcd_ptr, void* workspace,
                            int* parts, int* part_xcd_ptr, int* cuts, hipStream_t stream) {
  if (const int rc = panel_parts_check(num_panels, cap, workspace)) return rc;
  if (workspace == nullptr || part_xcd_ptr == nullptr || (num_panels > 0 && (panel_ptr == nullptr || parts == nullptr)))
    return kErrBadShape;
  const int blocks = num_panels < 1 ? 1 : (num_panels < 4096 ? num_panels : 4096);
  hipLaunchKernelGGL(panel_parts_fill_kernel, dim3(blocks), dim3(256), 0, stream, panel_ptr, num_panels, cap, panel_xcd_ptr,
                     parts_workspace(workspace, num_panels), parts, part_xcd_ptr, cuts);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace voltrix
