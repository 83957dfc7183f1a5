// Voltrix-SpMM for MI355X (gfx950) -- builder of the two-level format's panel plan (integer / byte work, LDS-bound).
//
// CSR on the device -> residual CSR + (panel_ptr, panel_cols, panel_bits), bit-identical to the plain-loop definition
// in oracle/oracle_np.py::panel_plan (layout: spmm_panel_kernels.hpp).  No reference counterpart.
//
// A column is "shared" in a panel when at least tau of the panel's rows reference it.  One 512-thread workgroup counts a
// panel's references per column in 16-bit LDS counters (a panel has at most 512 rows, so they cannot overflow), one range
// of 2^16 columns at a time (128 KiB of counters; a 233 k-column universe takes 4 ranges, each a pass over the panel's
// edge list out of L2).  Two phases around the one host sync the caller needs anyway (the outputs are data-sized):
//   count  grid = panels x ranges: shared columns per panel, residual edges per row (+ input checks), then prefix sums
//   fill   grid = panels, ranges in order: ranks of the shared columns by popcount prefix over a flag bitmap (no sort),
//          panel_cols in rank order, every row's edges either appended to the residual CSR (order kept) or OR-ed into
//          the adjacency word the panel kernel's lane will read.
// Rows must be sorted and duplicate-free (checked on the device: status[0] counts violations and out-of-universe ids).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "voltrix/csr_preprocess.hpp"
#include "voltrix/traits.hpp"

namespace voltrix {

constexpr int kPlanThreads = 1024;               // 16 waves: the row loops are round-trip-bound, LDS allows one workgroup per CU anyway
constexpr int kPlanWaves = kPlanThreads / kWave;
constexpr int kPlanRange = 1 << 16;                 // columns per counter range
constexpr int kPlanFlagWords = kPlanRange / 32;     // 2048
constexpr int kPlanMaxPanelRows = 512;
constexpr int kPlanMaxRanges = 64;                  // universes up to 2^22 columns
constexpr int kPlanBatch = 8;                       // independent global loads in flight per thread
constexpr int kPlanVecBatch = 8;                    // 16-byte loads in flight per thread in the panel's edge sweep
constexpr int kPlanRowBatch = 8;                    // ... per lane in the per-row loops (8 x 64 edges of a row per round trip)

struct PlanLds {
  unsigned counters[kPlanRange / 2];   // two 16-bit counters per word
  unsigned flags[kPlanFlagWords];      // bit i of word w: column 32 w + i of the range is shared
  int prefix[kPlanFlagWords];          // shared columns of the range below word w
  int rpos[kPlanMaxPanelRows];         // fill: next free slot of every row in the residual CSR
  int wsum[kPlanWaves];
  int total;
  int first_col;
};

// counters[c - c0] = number of edges of [lo, hi) with column c, for the columns of [c0, c0 + 2^16)
__device__ __forceinline__ void plan_count_range(PlanLds& s, const int* __restrict__ indices, const long long lo,
                                                 const long long hi, const int c0) {
  for (int i = threadIdx.x; i < kPlanRange / 2; i += kPlanThreads) s.counters[i] = 0u;
  __syncthreads();
  // the panel's edge list is one contiguous stream: scalar head up to the first 16-byte boundary, then 16 bytes per lane
  // and load (kPlanVecBatch loads = 64 KiB per workgroup in flight; 4-byte loads left this pass latency-bound: 61 round
  // trips for a 250 k-edge panel), scalar tail
  auto bump = [&](const int c) {
    if ((unsigned)c < (unsigned)kPlanRange) atomicAdd(&s.counters[c >> 1], 1u << (16 * (c & 1)));
  };
  const long long head_end = lo + ((4 - (int)(((uintptr_t)(indices + lo) & 15) >> 2)) & 3);
  const long long vec_lo = head_end < hi ? head_end : hi;
  const long long vec_hi = vec_lo + ((hi - vec_lo) & ~3ll);
  for (long long e = lo + threadIdx.x; e < vec_lo; e += kPlanThreads) bump(indices[e] - c0);
  for (long long e = vec_hi + threadIdx.x; e < hi; e += kPlanThreads) bump(indices[e] - c0);
  const int4* const vec = reinterpret_cast<const int4*>(indices + vec_lo);
  const long long nvec = (vec_hi - vec_lo) >> 2;
  for (long long base = threadIdx.x; base < nvec; base += (long long)kPlanThreads * kPlanVecBatch) {
    int4 c[kPlanVecBatch];
#pragma unroll
    for (int b = 0; b < kPlanVecBatch; ++b) {
      const long long i = base + (long long)b * kPlanThreads;
      c[b] = i < nvec ? vec[i] : make_int4(-1, -1, -1, -1);
    }
#pragma unroll
    for (int b = 0; b < kPlanVecBatch; ++b) {
      if (base + (long long)b * kPlanThreads < nvec) {
        bump(c[b].x - c0);
        bump(c[b].y - c0);
        bump(c[b].z - c0);
        bump(c[b].w - c0);
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ unsigned plan_counter(const PlanLds& s, const int c) {
  return (s.counters[c >> 1] >> (16 * (c & 1))) & 0xFFFFu;
}

// workgroup-wide sum (every thread gets it); uses s.wsum / s.total
__device__ __forceinline__ int plan_block_sum(PlanLds& s, int v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) s.wsum[threadIdx.x / kWave] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int k = 0; k < kPlanWaves; ++k) t += s.wsum[k];
  return t;
}

static __global__ __launch_bounds__(kPlanThreads) void panel_plan_count_kernel(
    const int* __restrict__ indptr, const int* __restrict__ indices, const int num_nodes, const int num_cols,
    const int panel_rows, const unsigned tau, const int nranges, int* __restrict__ shared_count /* [NP], zeroed */,
    int* __restrict__ resid_count /* [N], zeroed */, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  PlanLds& s = *reinterpret_cast<PlanLds*>(smem);
  const int p = blockIdx.x / nranges, rho = blockIdx.x % nranges;
  const int r0 = p * panel_rows;
  const int r1 = r0 + panel_rows < num_nodes ? r0 + panel_rows : num_nodes;
  const int c0 = rho * kPlanRange;
  const long long lo = indptr[r0], hi = indptr[r1];
  plan_count_range(s, indices, lo, hi, c0);

  int mine = 0;
  for (int i = threadIdx.x; i < kPlanRange / 2; i += kPlanThreads) {
    const unsigned w = s.counters[i];
    mine += ((w & 0xFFFFu) >= tau) + ((w >> 16) >= tau);
  }
  const int total = plan_block_sum(s, mine);
  if (threadIdx.x == 0 && total) atomicAdd(&shared_count[p], total);

  // residual edges per row: one wave per row; range 0 also checks the input (sorted, duplicate-free, ids in the universe)
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  // row bounds of this wave's rows (row r0 + wave + 16 i in lane i), one round trip for all of them
  static_assert(kPlanMaxPanelRows <= kPlanWaves * kWave, "one lane per row of the wave");
  const int my_row = r0 + wave + kPlanWaves * lane;
  const int my_a = my_row < r1 ? indptr[my_row] : 0, my_b = my_row < r1 ? indptr[my_row + 1] : 0;
  for (int row = r0 + wave, ri = 0; row < r1; row += kPlanWaves, ++ri) {
    const long long a = __shfl(my_a, ri, kWave), b = __shfl(my_b, ri, kWave);
    int cnt = 0, bad = 0, prev_last = -1;
    // kPlanRowBatch x 64 edges of the row per round trip (one load per iteration left the wave waiting on memory for the
    // whole row loop: 64 rows x 8 iterations x ~1.5 us per workgroup and range)
    for (long long e0 = a; e0 < b; e0 += kWave * kPlanRowBatch) {
      int cs[kPlanRowBatch];
#pragma unroll
      for (int k = 0; k < kPlanRowBatch; ++k) {
        const long long e = e0 + k * kWave + lane;
        cs[k] = e < b ? indices[e] : 0x7FFFFFFF;
      }
#pragma unroll
      for (int k = 0; k < kPlanRowBatch; ++k) {
        if (e0 + k * kWave >= b) break;  // wave-uniform
        const long long e = e0 + k * kWave + lane;
        const int c = cs[k];
        if (rho == 0) {
          int prev = __shfl_up(c, 1, kWave);
          if (lane == 0) prev = prev_last;
          bad += (e < b && (c <= prev || c < 0 || c >= num_cols)) ? 1 : 0;
          prev_last = __shfl(c, kWave - 1, kWave);
        }
        const int cr = c - c0;
        const bool resid = e < b && (unsigned)cr < (unsigned)kPlanRange && plan_counter(s, cr) < tau;
        cnt += __popcll(__ballot(resid));
      }
    }
    if (lane == 0 && cnt) atomicAdd(&resid_count[row], cnt);
    if (rho == 0) {
#pragma unroll
      for (int off = kWave / 2; off > 0; off >>= 1) bad += __shfl_xor(bad, off, kWave);
      if (lane == 0 && bad) atomicAdd(status, bad);
    }
  }
}

// panel_ptr[p + 1] = panel_ptr[p] + ceil(shared_count[p] / 32)   (one workgroup; NP is small)
static __global__ __launch_bounds__(256) void panel_ptr_kernel(const int* __restrict__ shared_count, const int num_panels,
                                                               int* __restrict__ panel_ptr) {
  __shared__ int wsum[4];
  __shared__ int carry_s;
  if (threadIdx.x == 0) {
    carry_s = 0;
    panel_ptr[0] = 0;
  }
  __syncthreads();
  for (int base = 0; base < num_panels; base += 256) {
    const int i = base + threadIdx.x;
    const int v = i < num_panels ? (shared_count[i] + kStageK - 1) / kStageK : 0;
    const int inc = wave_inclusive_scan(v);
    if ((threadIdx.x & (kWave - 1)) == kWave - 1) wsum[threadIdx.x / kWave] = inc;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < (int)(threadIdx.x / kWave); ++k) woff += wsum[k];
    const int carry = carry_s;
    if (i < num_panels) panel_ptr[i + 1] = carry + woff + inc;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + woff + inc;
    __syncthreads();
  }
}

static __global__ __launch_bounds__(kPlanThreads) void panel_plan_fill_kernel(
    const int* __restrict__ indptr, const int* __restrict__ indices, const int num_nodes, const int panel_rows,
    const int waves, const int row_blocks, const unsigned tau, const int nranges, const int* __restrict__ panel_ptr,
    const int* __restrict__ shared_count, const int* __restrict__ resid_indptr, int* __restrict__ resid_indices,
    int* __restrict__ panel_cols, unsigned* __restrict__ panel_bits /* zeroed */) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  PlanLds& s = *reinterpret_cast<PlanLds*>(smem);
  const int p = blockIdx.x;
  const int r0 = p * panel_rows;
  const int r1 = r0 + panel_rows < num_nodes ? r0 + panel_rows : num_nodes;
  const long long lo = indptr[r0], hi = indptr[r1];
  const long long ks0 = panel_ptr[p];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  for (int i = threadIdx.x; i < r1 - r0; i += kPlanThreads) s.rpos[i] = resid_indptr[r0 + i];
  if (threadIdx.x == 0) s.first_col = 0;
  int rank_base = 0;  // shared columns of the panel in the ranges before this one (workgroup-uniform)
  // row bounds of this wave's rows (row r0 + wave + 16 i in lane i), loaded once for all ranges
  const int my_row = r0 + wave + kPlanWaves * lane;
  const int my_a = my_row < r1 ? indptr[my_row] : 0, my_b = my_row < r1 ? indptr[my_row + 1] : 0;

  for (int rho = 0; rho < nranges; ++rho) {
    const int c0 = rho * kPlanRange;
    plan_count_range(s, indices, lo, hi, c0);

    // flags + their exclusive popcount prefix: thread t owns words 4t .. 4t+3
    constexpr int WPT = kPlanFlagWords / kPlanThreads;
    unsigned fl[WPT];
    int pc = 0;
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
      const int w = threadIdx.x * WPT + k;
      unsigned f = 0u;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const unsigned cw = s.counters[w * 16 + i];
        f |= ((cw & 0xFFFFu) >= tau ? 1u : 0u) << (2 * i);
        f |= ((cw >> 16) >= tau ? 1u : 0u) << (2 * i + 1);
      }
      fl[k] = f;
      pc += __popc(f);
    }
    const int inc = wave_inclusive_scan(pc);
    if (lane == kWave - 1) s.wsum[wave] = inc;
    __syncthreads();
    int woff = 0, range_total = 0;
#pragma unroll
    for (int k = 0; k < kPlanWaves; ++k) {
      woff += k < wave ? s.wsum[k] : 0;
      range_total += s.wsum[k];
    }
    int run = woff + inc - pc;
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
      const int w = threadIdx.x * WPT + k;
      s.flags[w] = fl[k];
      s.prefix[w] = run;
      // panel_cols in rank order
      unsigned f = fl[k];
      int rank = rank_base + run;
      while (f) {
        const int i = __ffs(f) - 1;
        f &= f - 1u;
        const int col = c0 + 32 * w + i;
        panel_cols[ks0 * kStageK + rank] = col;
        if (rank == 0) s.first_col = col;
        ++rank;
      }
      run += __popc(fl[k]);
    }
    __syncthreads();

    // rows: residual edges keep their order in the residual CSR, shared edges become adjacency bits
    for (int row = r0 + wave, ri = 0; row < r1; row += kPlanWaves, ++ri) {
      const int rp = row - r0;
      const int v = rp / (16 * row_blocks), j = (rp % (16 * row_blocks)) / 16, r16 = rp % 16;
      const long long a = __shfl(my_a, ri, kWave), b = __shfl(my_b, ri, kWave);
      int pos = s.rpos[rp];
      for (long long e0 = a; e0 < b; e0 += kWave * kPlanRowBatch) {   // batched as in the count kernel
        int cs[kPlanRowBatch];
#pragma unroll
        for (int q = 0; q < kPlanRowBatch; ++q) {
          const long long e = e0 + q * kWave + lane;
          cs[q] = e < b ? indices[e] : 0x7FFFFFFF;
        }
#pragma unroll
        for (int q = 0; q < kPlanRowBatch; ++q) {
          if (e0 + q * kWave >= b) break;  // wave-uniform
          const long long e = e0 + q * kWave + lane;
          const int c = cs[q];
          const int cr = c - c0;
          const bool in_range = e < b && (unsigned)cr < (unsigned)kPlanRange;
          const bool shared = in_range && ((s.flags[cr >> 5] >> (cr & 31)) & 1u);
          const bool resid = in_range && !shared;
          const unsigned long long rmask = __ballot(resid);
          if (resid) resid_indices[pos + __popcll(rmask & ((1ull << lane) - 1ull))] = c;
          pos += __popcll(rmask);
          if (shared) {
            const int rank = rank_base + s.prefix[cr >> 5] + __popc(s.flags[cr >> 5] & ((1u << (cr & 31)) - 1u));
            const long long ks = ks0 + (rank >> 5);
            const int k = rank & 31;
            atomicOr(&panel_bits[(ks * waves + v) * kWave + (k >> 3) * 16 + r16],
                     1u << (16 * (k & 1) + 4 * j + ((k & 7) >> 1)));
          }
        }
      }
      if (lane == 0) s.rpos[rp] = pos;
    }
    rank_base += range_total;
    __syncthreads();  // the counters / flags are rewritten by the next range
  }

  // unused slots of the panel's last k-step repeat its first shared column (finite data, zero adjacency bits)
  const int cnt = shared_count[p];
  const int padded = (cnt + kStageK - 1) / kStageK * kStageK;
  if ((int)threadIdx.x < padded - cnt) panel_cols[ks0 * kStageK + cnt + threadIdx.x] = s.first_col;
}

// Launch order of the panel kernel (spmm_panel_kernels.hpp, PanelArgs::panel_order): order_out[position] = panel.  Inside
// every XCD's contiguous range of positions (ceil(NP / 8) each) the panels are taken in GROUPS of `group` consecutive
// panels -- neighbours share most of their band columns, and launched side by side they share the gathered rows through
// L2 -- the groups with the most k-steps first (ties by index: a stable order, so a plan always gets the same one), the
// panels of a group in their natural order.  group = 1: plain longest-first.  Rank by counting inside the range (NP is
// small: N / 512).  Measured on the reddit-like pair with two units per wave on the residual: groups of 4 1.282 ms,
// plain longest-first 1.323 ms, natural order 1.294 ms (profiles/r02/experiment_panel_groups.log) -- but no gain through
// the operator (profiles/r02/bench_ab_panel_group.txt), so hosts pass group = 1.
// xcd_ptr (round 4): int32[9], first position of every XCD's range (xcd_ptr[8] = NP), or NULL = ranges of ceil(NP / 8)
// positions: ranges of equal WORK for graphs whose k-steps per panel vary by community (voltrix/hybrid.py::xcd_partition).
static __global__ __launch_bounds__(256) void panel_order_kernel(const int* __restrict__ panel_ptr, const int num_panels,
                                                                 const int per_xcd, const int group,
                                                                 const int* __restrict__ xcd_ptr,
                                                                 int* __restrict__ order_out) {
  for (int p = blockIdx.x * 256 + threadIdx.x; p < num_panels; p += gridDim.x * 256) {
    int lo = (p / per_xcd) * per_xcd;
    int hi = lo + per_xcd < num_panels ? lo + per_xcd : num_panels;
    if (xcd_ptr != nullptr) {
      int x = 0;
      for (int i = 1; i < kNumXcd; ++i) x += p >= xcd_ptr[i] ? 1 : 0;
      lo = xcd_ptr[x];
      hi = xcd_ptr[x + 1];
    }
    const int my_group = (p - lo) / group;
    const int g0 = lo + my_group * group, g1 = g0 + group < hi ? g0 + group : hi;
    const int mine = panel_ptr[g1] - panel_ptr[g0];      // k-steps of my group
    int before = 0;                                       // panels of the groups launched before mine
    for (int q0 = lo, gi = 0; q0 < hi; q0 += group, ++gi) {
      const int q1 = q0 + group < hi ? q0 + group : hi;
      const int other = panel_ptr[q1] - panel_ptr[q0];
      if (other > mine || (other == mine && gi < my_group)) before += q1 - q0;
    }
    order_out[lo + before + (p - g0)] = p;
  }
}

inline int panel_order(const int* panel_ptr, int num_panels, int group, int* order_out, hipStream_t stream,
                       const int* xcd_ptr = nullptr) {
  if (num_panels < 0 || group < 1) return kErrBadShape;
  if (num_panels == 0) return kOk;
  const int per_xcd = (num_panels + kNumXcd - 1) / kNumXcd;
  hipLaunchKernelGGL(panel_order_kernel, dim3((num_panels + 255) / 256), dim3(256), 0, stream, panel_ptr, num_panels,
                     per_xcd, group, xcd_ptr, order_out);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

struct PlanWorkspace {
  int* shared_count;  // [NP]
  int* resid_count;   // [N]
  int* chunk_sums;    // [ceil(N / kScanChunk)]
  long long bytes;
};

inline PlanWorkspace plan_workspace(void* base, int num_nodes, int panel_rows) {
  const int np = (num_nodes + panel_rows - 1) / panel_rows;
  PlanWorkspace ws;
  char* p = static_cast<char*>(base);
  ws.shared_count = reinterpret_cast<int*>(p);
  p += align16(4ll * (np + 1));
  ws.resid_count = reinterpret_cast<int*>(p);
  p += align16(4ll * (num_nodes + 1));
  ws.chunk_sums = reinterpret_cast<int*>(p);
  p += align16(4ll * ((num_nodes + kScanChunk - 1) / kScanChunk + 1));
  ws.bytes = p - static_cast<char*>(base);
  return ws;
}

inline int plan_check(int num_nodes, int num_cols, long long num_edges, int waves, int row_blocks, int tau) {
  if (num_nodes < 0 || num_edges < 0 || num_edges > 0x7FFFFFFFll || num_cols < 0) return kErrBadShape;
  if (!(waves == 4 || waves == 8) || !(row_blocks == 2 || row_blocks == 4) || tau < 1 || tau > 65535) return kErrBadShape;
  if ((num_cols + kPlanRange - 1) / kPlanRange > kPlanMaxRanges) return kErrBadConfig;  // universe above 2^22 columns
  return kOk;
}

// Phase 1.  panel_ptr[NP+1], resid_indptr[N+1], status[1] are written; the caller reads S = panel_ptr[NP],
// E_r = resid_indptr[N] and status[0] (must be 0) and allocates the outputs of phase 2.
inline int panel_plan_count(const int* indptr, const int* indices, int num_nodes, int num_cols, long long num_edges,
                            int waves, int row_blocks, int tau, void* workspace, int* panel_ptr, int* resid_indptr,
                            int* status, hipStream_t stream) {
  if (int rc = plan_check(num_nodes, num_cols, num_edges, waves, row_blocks, tau)) return rc;
  if (((uintptr_t)workspace & 15) || status == nullptr) return kErrBadShape;
  const int panel_rows = waves * row_blocks * 16;
  const int np = (num_nodes + panel_rows - 1) / panel_rows;
  if (hipMemsetAsync(status, 0, sizeof(int), stream) != hipSuccess) return kErrLaunch;
  if (hipMemsetAsync(panel_ptr, 0, sizeof(int), stream) != hipSuccess) return kErrLaunch;
  if (hipMemsetAsync(resid_indptr, 0, sizeof(int), stream) != hipSuccess) return kErrLaunch;
  if (num_nodes == 0) return kOk;
  const PlanWorkspace ws = plan_workspace(workspace, num_nodes, panel_rows);
  if (hipMemsetAsync(workspace, 0, (size_t)ws.bytes, stream) != hipSuccess) return kErrLaunch;
  const int nranges = num_cols > 0 ? (num_cols + kPlanRange - 1) / kPlanRange : 1;
  if ((long long)np * nranges > 0x7FFFFFFFll) return kErrBadShape;
  if (int rc = bm_set_lds(panel_plan_count_kernel, sizeof(PlanLds))) return rc;
  hipLaunchKernelGGL(panel_plan_count_kernel, dim3(np * nranges), dim3(kPlanThreads), sizeof(PlanLds), stream, indptr,
                     indices, num_nodes, num_cols, panel_rows, (unsigned)tau, nranges, ws.shared_count, ws.resid_count,
                     status);
  hipLaunchKernelGGL(panel_ptr_kernel, dim3(1), dim3(256), 0, stream, ws.shared_count, np, panel_ptr);
  const int nchunks = (num_nodes + kScanChunk - 1) / kScanChunk;
  hipLaunchKernelGGL(scan_chunk_sums_kernel, dim3(nchunks), dim3(256), 0, stream, ws.resid_count, num_nodes,
                     ws.chunk_sums);
  hipLaunchKernelGGL(scan_chunk_offsets_kernel, dim3(1), dim3(256), 0, stream, ws.chunk_sums, nchunks);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nchunks), dim3(256), 0, stream, ws.resid_count, num_nodes, ws.chunk_sums,
                     resid_indptr);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// Phase 2 (same workspace, untouched since phase 1).  resid_indices[E_r], panel_cols[32 (S + 2)],
// panel_bits[(S + 1) * waves * 64]: every element is written (memset + kernel).
inline int panel_plan_fill(const int* indptr, const int* indices, int num_nodes, int num_cols, long long num_edges,
                           int waves, int row_blocks, int tau, void* workspace, const int* panel_ptr,
                           const int* resid_indptr, long long total_ksteps, int* resid_indices, int* panel_cols,
                           uint32_t* panel_bits, hipStream_t stream) {
  if (int rc = plan_check(num_nodes, num_cols, num_edges, waves, row_blocks, tau)) return rc;
  if (((uintptr_t)workspace & 15) || total_ksteps < 0) return kErrBadShape;
  const int panel_rows = waves * row_blocks * 16;
  const int np = (num_nodes + panel_rows - 1) / panel_rows;
  if (hipMemsetAsync(panel_bits, 0, (size_t)(total_ksteps + 1) * waves * kWave * 4, stream) != hipSuccess) return kErrLaunch;
  if (hipMemsetAsync(panel_cols + total_ksteps * kStageK, 0, 2 * kStageK * 4, stream) != hipSuccess) return kErrLaunch;
  if (num_nodes == 0) return kOk;
  const PlanWorkspace ws = plan_workspace(workspace, num_nodes, panel_rows);
  const int nranges = num_cols > 0 ? (num_cols + kPlanRange - 1) / kPlanRange : 1;
  if (int rc = bm_set_lds(panel_plan_fill_kernel, sizeof(PlanLds))) return rc;
  hipLaunchKernelGGL(panel_plan_fill_kernel, dim3(np), dim3(kPlanThreads), sizeof(PlanLds), stream, indptr, indices,
                     num_nodes, panel_rows, waves, row_blocks, (unsigned)tau, nranges, panel_ptr, ws.shared_count,
                     resid_indptr, resid_indices, panel_cols, panel_bits);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

inline long long panel_plan_workspace_bytes(int num_nodes, int waves, int row_blocks) {
  if (num_nodes < 0 || waves <= 0 || row_blocks <= 0) return 0;
  return plan_workspace(nullptr, num_nodes, waves * row_blocks * 16).bytes;
}

}  // namespace voltrix
