// Voltrix-SpMM for MI355X (gfx950 / CDNA4) -- the stream kernel's schedule tables, built on the device by the library.
//
// spmm_stream_kernel (spmm_stream_kernels.hpp) walks RUNS of consecutive units of the window format.  This is the builder of
// its tables from the handle's three tensors (no reference counterpart: the reference maps one CTA to one row window,
// voltrix/include/voltrix/spmm_kernels.cuh:1476-1490); voltrix/schedule.py::stream_tables is its torch-tensor restatement, and
// tests/test_gpu_schedule.py compares the two element by element.  Integer work, a few launches, once per handle.
//
//   unit   = a whole window, or -- windows of more than `cut_stages` stages -- one of its k = ceil(stages / cut_stages)
//            interleaved pieces (unit j runs the stages j, j + k, ...; partial tiles summed in unit order by
//            combine_partials_kernel, as in unit_table.hpp).  Units stay in WINDOW order.
//            int32 [U][8] = {first TC block, end TC block of the window, 4 k, window, partial-tile slot or -1, stages,
//            columns of the window's last TC block that carry an edge (0: a window without edges), 0}
//   cost   = stages + 1 per unit (the store of its 16 rows); the eight XCD ranges of units hold equal cost (the boundary goes
//            where the prefix is nearest to x / 8 of the total: schedule.py::split_equal_work)
//   run    = the units of one XCD range whose cost prefix (from the range's start) falls into the same bucket of `run_cost`:
//            int32 [R][4] = {first unit, units (<= 64 for run_cost <= 128), stages, 0};  run_ptr int32 [9]
//   defaults (argument <= 0): run_cost = clamp((stages + windows) / (1536 wave slots x 6 runs), 6, 48); cut_stages =
//            max(run_cost, max(8, 1.5 x the lower median of the windows' stages))
//
// Two phases around the host reads that size the outputs:
//   count  header int32 [8] <- {U, cut windows C, partial-tile slots, bound on R, run_cost, cut_stages, 0, 0}
//   fill   units, cuts [C][4] = {window, first slot, units, 0}, runs, run_ptr;  header2 int32 [4] <- {R, max runs per XCD, 0, 0}
#pragma once

#include "voltrix/unit_table.hpp"

namespace voltrix {

constexpr int kStWaveSlots = 256 * 6;   // waves the chip holds at the stream kernel's default ring depth
constexpr int kStRunsPerSlot = 6;

struct StWorkspace {
  int* hist;        // [kUtHistBins + 1]
  int* stats;       // [16]: 0 = 1.5 x median (unit table's default), 2 = run_cost, 3 = cut_stages, 4..5 = total stages (64 bit)
  int* k;           // [W + 1] units per window
  int* kcut;        // [W + 1]
  int* cutflag;     // [W + 1]
  int* cost;        // [W + 1] stages + units
  int* nst;         // [W + 1]
  int* ncl;         // [W]
  int* first;       // [W + 1] exclusive scans
  int* slot_first;  // [W + 1]
  int* cut_pos;     // [W + 1]
  int* cost_before; // [W + 1]
  int* stage_before;// [W + 1]
  int* chunk_sums;
  long long bytes;
};
inline StWorkspace st_workspace(void* base, int num_nodes) {
  const long long W = ((long long)num_nodes + kBlkH - 1) / kBlkH;
  const long long nchunks = (W + 1 + kScanChunk - 1) / kScanChunk + 1;
  char* p = static_cast<char*>(base);
  StWorkspace ws;
  auto take = [&](long long ints) {
    int* q = reinterpret_cast<int*>(p);
    p += align16(4 * ints);
    return q;
  };
  ws.hist = take(kUtHistBins + 1);
  ws.stats = take(16);
  ws.k = take(W + 1);
  ws.kcut = take(W + 1);
  ws.cutflag = take(W + 1);
  ws.cost = take(W + 1);
  ws.nst = take(W + 1);
  ws.ncl = take(W);
  ws.first = take(W + 2);
  ws.slot_first = take(W + 2);
  ws.cut_pos = take(W + 2);
  ws.cost_before = take(W + 2);
  ws.stage_before = take(W + 2);
  ws.chunk_sums = take(nchunks);
  ws.bytes = p - static_cast<char*>(base);
  return ws;
}
inline long long stream_table_workspace_bytes(int num_nodes) { return st_workspace(nullptr, num_nodes).bytes; }

struct StFillWorkspace {
  int* unit_cost_before;   // [U + 1]
  int* unit_stage_before;  // [U + 1]
  int* flags;              // [U + 1]
  int* run_id;             // [U + 2] (exclusive scan of flags: run of unit u = run_id[u + 1] - 1)
  int* ub;                 // [16] unit boundaries of the XCD ranges
  int* chunk_sums;
  long long bytes;
};
inline StFillWorkspace st_fill_workspace(void* base, long long num_units) {
  const long long nchunks = (num_units + 2 + kScanChunk - 1) / kScanChunk + 1;
  char* p = static_cast<char*>(base);
  StFillWorkspace ws;
  auto take = [&](long long ints) {
    int* q = reinterpret_cast<int*>(p);
    p += align16(4 * ints);
    return q;
  };
  ws.unit_cost_before = take(num_units + 1);
  ws.unit_stage_before = take(num_units + 1);
  ws.flags = take(num_units + 1);
  ws.run_id = take(num_units + 2);
  ws.ub = take(16);
  ws.chunk_sums = take(nchunks);
  ws.bytes = p - static_cast<char*>(base);
  return ws;
}
inline long long stream_table_fill_workspace_bytes(long long num_units) {
  return st_fill_workspace(nullptr, num_units < 0 ? 0 : num_units).bytes;
}

// histogram of the windows' stages (median) + their total
static __global__ __launch_bounds__(256) void st_hist_kernel(const int* __restrict__ blk_offsets, const int num_windows,
                                                             int* __restrict__ hist, unsigned long long* __restrict__ total) {
  unsigned long long mine = 0;
  for (int w = blockIdx.x * 256 + threadIdx.x; w < num_windows; w += gridDim.x * 256) {
    const int nst = ut_stages(blk_offsets, w);
    atomicAdd(&hist[nst < kUtHistBins ? nst : kUtHistBins], 1);
    mine += (unsigned long long)(nst < 1 ? 1 : nst);
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) mine += __shfl_xor(mine, off, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0 && mine) atomicAdd(total, mine);
}

static __global__ void st_params_kernel(int* __restrict__ stats, const int num_windows, const int run_cost_arg,
                                        const int cut_stages_arg) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const unsigned long long stages = *reinterpret_cast<const unsigned long long*>(stats + 4);
  int run_cost = run_cost_arg;
  if (run_cost <= 0) {
    const unsigned long long per = (stages + (unsigned long long)num_windows) / (unsigned long long)(kStWaveSlots * kStRunsPerSlot);
    run_cost = per < 6 ? 6 : (per > 48 ? 48 : (int)per);
  }
  int cut = cut_stages_arg;
  if (cut <= 0) cut = run_cost > stats[0] ? run_cost : stats[0];   // stats[0] = max(8, 1.5 x lower median): ut_median_kernel
  stats[2] = run_cost;
  stats[3] = cut;
}

// per window: units, cost, the last TC block's edge-carrying columns
static __global__ __launch_bounds__(256) void st_window_kernel(const int* __restrict__ blk_offsets,
                                                               const uint32_t* __restrict__ hspa_packed,
                                                               const int* __restrict__ hind, const int num_windows,
                                                               const int* __restrict__ stats, int* __restrict__ k_out,
                                                               int* __restrict__ kcut_out, int* __restrict__ cutflag_out,
                                                               int* __restrict__ cost_out, int* __restrict__ nst_out,
                                                               int* __restrict__ ncl_out) {
  const int cut = stats[3];
  for (int w = blockIdx.x * 256 + threadIdx.x; w <= num_windows; w += gridDim.x * 256) {
    if (w == num_windows) {   // the scans' trailing zero: out[W] = total
      k_out[w] = kcut_out[w] = cutflag_out[w] = cost_out[w] = nst_out[w] = 0;
      continue;
    }
    int nst = ut_stages(blk_offsets, w);
    if (nst < 1) nst = 1;
    int k = (int)(((long long)nst + cut - 1) / cut);
    if (k < 1) k = 1;
    k_out[w] = k;
    kcut_out[w] = k > 1 ? k : 0;
    cutflag_out[w] = k > 1 ? 1 : 0;
    cost_out[w] = nst + k;
    nst_out[w] = nst;
    const long long last = (long long)blk_offsets[w + 1] - 1;
    const int4 lo = *reinterpret_cast<const int4*>(hind + 8 * last);
    const int4 hi = *reinterpret_cast<const int4*>(hind + 8 * last + 4);
    int ncl = 1 + (lo.y > 0) + (lo.z > 0) + (lo.w > 0) + (hi.x > 0) + (hi.y > 0) + (hi.z > 0) + (hi.w > 0);
    const uint4 bits = *reinterpret_cast<const uint4*>(hspa_packed + 4 * last);
    if ((bits.x | bits.y | bits.z | bits.w) == 0u) ncl = 0;   // a window without edges: one all-zero TC block
    ncl_out[w] = ncl;
  }
}

static __global__ void st_header_kernel(const int* __restrict__ stats, const int* __restrict__ first,
                                        const int* __restrict__ slot_first, const int* __restrict__ cut_pos,
                                        const int* __restrict__ cost_before, const int num_windows, int* __restrict__ header) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    header[0] = first[num_windows];
    header[1] = cut_pos[num_windows];
    header[2] = slot_first[num_windows];
    header[3] = cost_before[num_windows] / stats[2] + 16;   // bound on the number of runs
    header[4] = stats[2];
    header[5] = stats[3];
    header[6] = 0;
    header[7] = 0;
  }
}

inline int stream_table_count(const int* blk_offsets, const uint32_t* hspa_packed, const int* hind, int num_nodes, int run_cost,
                              int cut_stages, void* workspace, int* header, hipStream_t stream) {
  if (int rc = unit_table_check(num_nodes)) return rc;
  if (((uintptr_t)workspace & 15) || header == nullptr || run_cost > 128 || ((uintptr_t)hind & 15) || ((uintptr_t)hspa_packed & 15))
    return kErrBadShape;
  if (hipMemsetAsync(header, 0, 8 * sizeof(int), stream) != hipSuccess) return kErrLaunch;
  const int W = (num_nodes + kBlkH - 1) / kBlkH;
  if (W == 0) return kOk;
  const StWorkspace ws = st_workspace(workspace, num_nodes);
  if (hipMemsetAsync(ws.hist, 0, (size_t)((char*)ws.k - (char*)ws.hist), stream) != hipSuccess) return kErrLaunch;  // hist + stats
  const int grid = (W + 256) / 256 < 4096 ? (W + 256) / 256 : 4096;
  hipLaunchKernelGGL(st_hist_kernel, dim3(grid), dim3(256), 0, stream, blk_offsets, W, ws.hist,
                     reinterpret_cast<unsigned long long*>(ws.stats + 4));
  hipLaunchKernelGGL(ut_median_kernel, dim3(1), dim3(1024), 0, stream, ws.hist, W, 0, ws.stats);
  hipLaunchKernelGGL(st_params_kernel, dim3(1), dim3(64), 0, stream, ws.stats, W, run_cost, cut_stages);
  hipLaunchKernelGGL(st_window_kernel, dim3(grid), dim3(256), 0, stream, blk_offsets, hspa_packed, hind, W, ws.stats, ws.k,
                     ws.kcut, ws.cutflag, ws.cost, ws.nst, ws.ncl);
  if (int rc = ut_exclusive_scan(ws.k, W, ws.chunk_sums, ws.first, stream)) return rc;
  if (int rc = ut_exclusive_scan(ws.kcut, W, ws.chunk_sums, ws.slot_first, stream)) return rc;
  if (int rc = ut_exclusive_scan(ws.cutflag, W, ws.chunk_sums, ws.cut_pos, stream)) return rc;
  if (int rc = ut_exclusive_scan(ws.cost, W, ws.chunk_sums, ws.cost_before, stream)) return rc;
  if (int rc = ut_exclusive_scan(ws.nst, W, ws.chunk_sums, ws.stage_before, stream)) return rc;
  hipLaunchKernelGGL(st_header_kernel, dim3(1), dim3(64), 0, stream, ws.stats, ws.first, ws.slot_first, ws.cut_pos,
                     ws.cost_before, W, header);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// one thread per unit: the record, the prefixes of cost and stages in front of it, and the cut record of a cut window
static __global__ __launch_bounds__(256) void st_units_kernel(const int* __restrict__ blk_offsets, const int num_windows,
                                                              const int num_units, const int* __restrict__ k_in,
                                                              const int* __restrict__ nst_in, const int* __restrict__ ncl_in,
                                                              const int* __restrict__ first,
                                                              const int* __restrict__ slot_first,
                                                              const int* __restrict__ cut_pos,
                                                              const int* __restrict__ cost_before,
                                                              const int* __restrict__ stage_before, int* __restrict__ units,
                                                              int* __restrict__ cuts, int* __restrict__ unit_cost_before,
                                                              int* __restrict__ unit_stage_before) {
  for (int u = blockIdx.x * 256 + threadIdx.x; u <= num_units; u += gridDim.x * 256) {
    if (u == num_units) {
      unit_cost_before[u] = cost_before[num_windows];
      unit_stage_before[u] = stage_before[num_windows];
      continue;
    }
    int lo = 0, hi = num_windows;   // the window whose units hold u: first[w] <= u < first[w + 1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (first[mid] <= u) lo = mid; else hi = mid;
    }
    const int w = lo, j = u - first[w], k = k_in[w], nst = nst_in[w];
    const int q = nst / k, r = nst % k;                       // units j < r run q + 1 stages, the others q
    const int len = q + (j < r ? 1 : 0);
    const int stages_in_front = j <= r ? j * (q + 1) : r * (q + 1) + (j - r) * q;
    unit_cost_before[u] = cost_before[w] + stages_in_front + j;
    unit_stage_before[u] = stage_before[w] + stages_in_front;
    int4* const rec = reinterpret_cast<int4*>(units + 8ll * u);
    rec[0] = int4{blk_offsets[w] + 4 * j, blk_offsets[w + 1], 4 * k, w};
    rec[1] = int4{k > 1 ? slot_first[w] + j : -1, len, ncl_in[w], 0};
    if (j == 0 && k > 1) *reinterpret_cast<int4*>(cuts + 4ll * cut_pos[w]) = int4{w, slot_first[w], k, 0};
  }
}

// unit boundaries of the eight XCD ranges: where the cost prefix is nearest to x / 8 of the total (split_equal_work)
static __global__ void st_xcd_kernel(const int* __restrict__ unit_cost_before, const int num_units, int* __restrict__ ub) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const long long total = unit_cost_before[num_units];
  ub[0] = 0;
  ub[kNumXcd] = num_units;
  int prev = 0;
  for (int x = 1; x < kNumXcd; ++x) {
    int cut;
    if (total > 0) {
      const long long target = (total * x + kNumXcd - 1) / kNumXcd;
      int lo = 0, hi = num_units;   // smallest i in [1, U] with prefix(i items) = unit_cost_before[i] >= target
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (unit_cost_before[mid + 1] >= target) hi = mid; else lo = mid + 1;
      }
      int items = lo + 1;           // the crossing item included
      if (items > num_units) items = num_units;
      const long long over = (long long)unit_cost_before[items] - target;
      const long long under = target - (items >= 2 ? (long long)unit_cost_before[items - 1] : 0ll);
      cut = under < over ? items - 1 : items;
    } else {
      const int per = (num_units + kNumXcd - 1) / kNumXcd;
      cut = x * per < num_units ? x * per : num_units;
    }
    if (cut < prev) cut = prev;
    ub[x] = cut;
    prev = cut;
  }
}

__device__ __forceinline__ long long st_run_key(const int* __restrict__ unit_cost_before, const int* __restrict__ ub,
                                                const int run_cost, const long long buckets, const int u) {
  int x = 0;
#pragma unroll
  for (int i = 1; i < kNumXcd; ++i) x += u >= ub[i] ? 1 : 0;
  return x * buckets + (unit_cost_before[u] - unit_cost_before[ub[x]]) / run_cost;
}

static __global__ __launch_bounds__(256) void st_flags_kernel(const int* __restrict__ unit_cost_before, const int* __restrict__ ub,
                                                              const int num_units, const int run_cost, int* __restrict__ flags) {
  const long long buckets = (long long)unit_cost_before[num_units] / run_cost + 2;
  for (int u = blockIdx.x * 256 + threadIdx.x; u <= num_units; u += gridDim.x * 256) {
    if (u == num_units) {
      flags[u] = 0;
      continue;
    }
    flags[u] = (u == 0 || st_run_key(unit_cost_before, ub, run_cost, buckets, u) !=
                              st_run_key(unit_cost_before, ub, run_cost, buckets, u - 1)) ? 1 : 0;
  }
}

// run_id = exclusive scan of flags: a unit with a flag opens run run_id[u]; run_id[U] = R
static __global__ __launch_bounds__(256) void st_run_first_kernel(const int* __restrict__ flags, const int* __restrict__ run_id,
                                                                  const int num_units, int* __restrict__ runs) {
  for (int u = blockIdx.x * 256 + threadIdx.x; u < num_units; u += gridDim.x * 256)
    if (flags[u]) runs[4ll * run_id[u]] = u;
}

static __global__ __launch_bounds__(256) void st_run_fill_kernel(const int* __restrict__ run_id, const int* __restrict__ ub,
                                                                 const int* __restrict__ unit_stage_before, const int num_units,
                                                                 const int run_bound, int* __restrict__ runs,
                                                                 int* __restrict__ run_ptr, int* __restrict__ header2) {
  const int R = run_id[num_units];
  for (int r = blockIdx.x * 256 + threadIdx.x; r < run_bound; r += gridDim.x * 256) {
    if (r >= R) {
      *reinterpret_cast<int4*>(runs + 4ll * r) = int4{0, 0, 0, 0};
      continue;
    }
    const int a = runs[4ll * r];
    const int b = r + 1 < R ? runs[4ll * (r + 1)] : num_units;   // the next run's first unit was written by st_run_first_kernel
    runs[4ll * r + 1] = b - a;
    if (b - a > 64) atomicOr(header2 + 2, 1);   // never with run_cost <= 128 (a unit costs >= 2): the launcher refuses such a table
    runs[4ll * r + 2] = unit_stage_before[b] - unit_stage_before[a];
    runs[4ll * r + 3] = 0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int mx = 0, prev = 0;
    for (int x = 0; x <= kNumXcd; ++x) {
      const int p = x == kNumXcd ? R : (ub[x] < num_units ? run_id[ub[x]] : R);
      run_ptr[x] = p;
      if (x > 0) mx = p - prev > mx ? p - prev : mx;
      prev = p;
    }
    header2[0] = R;
    header2[1] = mx;    // header2[2]: bit 0 set by a run of more than 64 units (zeroed by the launcher's memset); [3] = 0
  }
}

// Phase 2.  units int32 [U][8], cuts int32 [C][4], runs int32 [run_bound][4] (the first R are the runs), run_ptr int32 [9],
// header2 int32 [4] = {R, max runs per XCD, 1 if a run holds more than 64 units (invalid for the kernel), 0}.
inline int stream_table_fill(const int* blk_offsets, int num_nodes, void* workspace, void* fill_workspace, int num_units,
                             int num_cuts, int run_bound, int run_cost, int* units, int* cuts, int* runs, int* run_ptr,
                             int* header2, hipStream_t stream) {
  if (int rc = unit_table_check(num_nodes)) return rc;
  if (((uintptr_t)workspace & 15) || ((uintptr_t)fill_workspace & 15) || ((uintptr_t)units & 15) || ((uintptr_t)runs & 15) ||
      ((uintptr_t)cuts & 15) || run_ptr == nullptr || header2 == nullptr || num_units < 0 || run_bound < 0 || run_cost < 2 ||
      run_cost > 128)   // > 128: runs of more than 64 units -- the kernel keeps a run's unit table one unit per lane (as the count phase)
    return kErrBadShape;
  const int W = (num_nodes + kBlkH - 1) / kBlkH;
  if (hipMemsetAsync(run_ptr, 0, 9 * sizeof(int), stream) != hipSuccess) return kErrLaunch;
  if (hipMemsetAsync(header2, 0, 4 * sizeof(int), stream) != hipSuccess) return kErrLaunch;
  if (W == 0 || num_units == 0) return kOk;
  const StWorkspace ws = st_workspace(workspace, num_nodes);
  const StFillWorkspace fw = st_fill_workspace(fill_workspace, num_units);
  const int grid = (num_units + 256) / 256 < 8192 ? (num_units + 256) / 256 : 8192;
  hipLaunchKernelGGL(st_units_kernel, dim3(grid), dim3(256), 0, stream, blk_offsets, W, num_units, ws.k, ws.nst, ws.ncl, ws.first,
                     ws.slot_first, ws.cut_pos, ws.cost_before, ws.stage_before, units, cuts, fw.unit_cost_before,
                     fw.unit_stage_before);
  hipLaunchKernelGGL(st_xcd_kernel, dim3(1), dim3(64), 0, stream, fw.unit_cost_before, num_units, fw.ub);
  hipLaunchKernelGGL(st_flags_kernel, dim3(grid), dim3(256), 0, stream, fw.unit_cost_before, fw.ub, num_units, run_cost, fw.flags);
  if (int rc = ut_exclusive_scan(fw.flags, num_units, fw.chunk_sums, fw.run_id, stream)) return rc;
  hipLaunchKernelGGL(st_run_first_kernel, dim3(grid), dim3(256), 0, stream, fw.flags, fw.run_id, num_units, runs);
  const int rgrid = (run_bound + 255) / 256 > 0 ? (run_bound + 255) / 256 : 1;
  hipLaunchKernelGGL(st_run_fill_kernel, dim3(rgrid), dim3(256), 0, stream, fw.run_id, fw.ub, fw.unit_stage_before, num_units,
                     run_bound, runs, run_ptr, header2);
  (void)num_cuts;
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace voltrix
