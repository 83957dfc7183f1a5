// Voltrix-SpMM for MI355X (gfx950) -- fused GPU preprocess: CSR (device) -> (pointer1, hspa_packed, hind).
//
// Produces, bit for bit, what the reference pipeline
//     voltrix::preprocess (CPU, bmat_kernels.cuh:264-320) -> hmat_cuda (:195-212) -> hmat_packed_swizzle_cuda (:228-242)
// produces, but entirely on the GPU and without the reference's three costs:
//   * the single-threaded std::map condensing on the host,
//   * the O(TC blocks x window edges) rescan in hmat_cuda_kernel (:66,:94),
//   * the transient fp32 `hspa` (512 bytes per TC block).
//
// Two rank algorithms (one workgroup or wave per 16-row window, grid-strided), chosen per call by csr_path(): bitmap,
// sort, or mixed = sort for the windows up to kSortLdsKeys edges and bitmap for the bigger ones:
//  bitmap (column universe <= 2^19 and not much larger than a window's edge list; e.g. reddit):
//   1. csr_bitmap_count_kernel  every edge sets bit `column` of an LDS bitmap; distinct columns = popcount.
//   2. scan_* kernels           pointer1 = exclusive prefix sum (wave shuffles + LDS, three small launches).
//   3. csr_bitmap_fill_kernel   bitmap again; group prefixes by a workgroup scan; hind = set bits in rank order (staged
//                               in LDS, coalesced copy-out); per edge rank = prefix + popcount -> bit OR-ed into an LDS
//                               stage of the window's packed words, written once.  No sort, no key workspace, no global
//                               atomics, no device-scope fences (one __threadfence per window cost 3x: it writes back L2).
//  sort (any universe up to 2^28):
//   1. csr_window_sort_kernel   key = (column << 4) | (row & 15) for every edge of the window; bitonic sort
//                               (all-ascending network, virtual +inf padding) in LDS, or in the global workspace when
//                               the window has more than kSortLdsKeys edges; sorted keys -> workspace; distinct
//                               columns counted with wave ballots -> block_partition[w] = ceil(U_w / 8) (0 -> 1).
//   2. scan_* kernels           as above.
//   3. csr_window_fill_kernel   zero the window's part of the handle; per sorted key: "first of its column" flags -> ballot/popcount prefix sum = condensed
//                               column rank; hind[8*pointer1[w] + rank] = column; bit (row, rank) OR-ed into the
//                               reference's swizzled word/bit position.
//   Windows of kWsKeys .. kSortLdsKeys edges take csr_bucket_count / _fill_kernel instead (ids below 2^27): an O(n) bucket
//   ranking in LDS, no sort (section "mid-size windows" below).
//   Windows with <= kWsKeys edges take the csr_wave_* twins of steps 1 and 3: one WAVE per window -- the sort is a
//   register-resident bitonic network (cross-lane shuffles, no LDS, no barriers), the fill stages the packed words in a
//   wave-private LDS slice and writes them once.
// Limits: num_nodes <= 2^28 (the packed key keeps the column in 28 bits), num_edges + W <= INT32_MAX.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "voltrix/traits.hpp"

namespace voltrix {

constexpr int kSortThreads = 256;
constexpr int kSortLdsKeys = 8192;   // 32 KiB of LDS per workgroup -> 4 workgroups per CU
constexpr int kScanChunk = 2048;     // elements per scan workgroup (256 threads x 8)

// All-ascending bitonic network over keys[0..n): every comparator moves the smaller key to the lower index, so
// indices >= n can be treated as +inf and skipped (no power-of-two padding is materialised).  THREADS cooperating
// threads (a workgroup with __syncthreads, or one wave on a wave-private LDS slice with a wave-level sync).
struct WorkgroupSync {
  __device__ __forceinline__ void operator()() const { __syncthreads(); }
};
struct WaveSync {  // LDS operations of one wave execute in order; this only pins the compiler's ordering
  __device__ __forceinline__ void operator()() const {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
};

// lk_first: first merge stage to run (blocks of 1 << (lk_first - 1) keys are already sorted ascending).
template <int THREADS, class Ptr, class Sync>
__device__ __forceinline__ void bitonic_sort_ascending(Ptr keys, const int n, const int tid, const Sync sync,
                                                       const int lk_first = 1) {
  constexpr int U = 4;  // independent compare-exchanges in flight per thread (hides the LDS / memory round trip)
  int lp = 0;
  while ((1 << lp) < n) ++lp;
  const int half = (1 << lp) >> 1;
  auto exchange = [&](auto index_pair) {
    for (int i0 = tid; i0 < half; i0 += U * THREADS) {
      int lo[U], hi[U];
      uint32_t x[U], y[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * THREADS;
        index_pair(i, lo[u], hi[u]);
        if (i >= half || hi[u] >= n) hi[u] = -1;  // nothing to do (partner is virtual +inf padding)
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (hi[u] >= 0) {
          x[u] = keys[lo[u]];
          y[u] = keys[hi[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (hi[u] >= 0 && x[u] > y[u]) {
          keys[lo[u]] = y[u];
          keys[hi[u]] = x[u];
        }
      }
    }
    sync();
  };
  for (int lk = lk_first; lk <= lp; ++lk) {  // k = 1 << lk
    const int lh = lk - 1, hmask = (1 << lh) - 1;
    exchange([&](const int i, int& lo, int& hi) {  // "flip" step: partner mirrored inside the k-block
      const int base = (i >> lh) << lk, off = i & hmask;
      lo = base + off;
      hi = base + (2 << lh) - 1 - off;
    });
    for (int lj = lk - 2; lj >= 0; --lj) {  // "shear" steps: partner at distance j = 1 << lj
      const int jmask = (1 << lj) - 1;
      exchange([&](const int i, int& lo, int& hi) {
        lo = ((i >> lj) << (lj + 1)) + (i & jmask);
        hi = lo + (1 << lj);
      });
    }
  }
}

// Row pointers of window w (wave-uniform -> scalar loads, one round trip): rp[q] = indptr[min(16w + q, num_nodes)].
__device__ __forceinline__ void load_window_rowptr(const int* __restrict__ indptr, const int w, const int num_nodes,
                                                   int (&rp)[kBlkH + 1]) {
  const long long r0 = (long long)w * kBlkH;
  if (r0 + kBlkH <= num_nodes) {
    const int* const p = indptr + r0;
#pragma unroll
    for (int q = 0; q <= kBlkH; ++q) rp[q] = p[q];
  } else {  // the last, partial window
#pragma unroll
    for (int q = 0; q <= kBlkH; ++q) rp[q] = indptr[r0 + q < num_nodes ? r0 + q : num_nodes];
  }
}

__device__ __forceinline__ int local_row(const int (&rp)[kBlkH + 1], const int e) {
  int rl = 0;
#pragma unroll
  for (int q = 1; q < kBlkH; ++q) rl += (rp[q] <= e) ? 1 : 0;
  return rl;
}

// ---- small windows: one WAVE per window on a wave-private LDS slice (no workgroup barriers) ----------------------
constexpr int kWsBatch = 4;     // independent global loads in flight per lane
constexpr int kWsKeys = 2048;   // edges per window handled by the wave path (32 keys per lane in registers)
constexpr int kWsWaves = 4;     // waves per workgroup (independent windows)
constexpr int kWsGrid = 256 * 16;  // grid-strided

// Bitonic sort of 64 * C keys held in registers, C consecutive keys per lane (index = lane * C + r).  Compare-exchange
// steps at distance < C stay inside a lane (register pairs, fully unrolled); distance >= C pairs lane with
// lane ^ (distance / C) through one cross-lane shuffle per register.  No LDS round trips, no barriers.
template <int C>
__device__ __forceinline__ void wave_bitonic_sort(uint32_t (&v)[C], const int lane) {
  constexpr int P = C * kWave;
#pragma unroll
  for (int k = 2; k <= P; k <<= 1) {
    // keys whose index has bit k set sort descending (k == P: one ascending run)
    const bool lane_desc = (k >= C && k < P) ? ((lane & (k / C)) != 0) : false;
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= C) {
        const int lj = j / C;
        const bool take_max = ((lane & lj) != 0) != lane_desc;
#pragma unroll
        for (int r = 0; r < C; ++r) {
          const uint32_t other = (uint32_t)__shfl_xor((int)v[r], lj, kWave);
          const uint32_t mn = v[r] < other ? v[r] : other, mx = v[r] < other ? other : v[r];
          v[r] = take_max ? mx : mn;
        }
      } else {
#pragma unroll
        for (int r = 0; r < C; ++r) {
          if ((r & j) == 0) {
            const uint32_t x = v[r], y = v[r | j];
            const uint32_t mn = x < y ? x : y, mx = x < y ? y : x;
            const bool desc = k < C ? ((r & k) != 0) : lane_desc;
            v[r] = desc ? mx : mn;
            v[r | j] = desc ? mn : mx;
          }
        }
      }
    }
  }
}

// One window of n <= 64 * C edges: keys (column << 4 | local row) -> sorted -> workspace; returns the distinct columns.
template <int C>
__device__ __forceinline__ int wave_sort_window(const int* __restrict__ indices, const int lo, const int n,
                                                const int (&rp)[kBlkH + 1], const unsigned col_limit,
                                                int* __restrict__ status, uint32_t* __restrict__ dst, const int lane) {
  uint32_t v[C];
#pragma unroll
  for (int r = 0; r < C; ++r) {
    const int i = lane * C + r;
    v[r] = i < n ? (uint32_t)indices[lo + i] : 0u;
  }
#pragma unroll
  for (int r = 0; r < C; ++r) {
    const int i = lane * C + r;
    if (i < n) {
      if (v[r] >= col_limit) atomicAdd(status, 1);  // outside the caller's column universe / the 28-bit key
      v[r] = (v[r] << 4) | (uint32_t)local_row(rp, lo + i);
    } else {
      v[r] = 0xFFFFFFFFu;  // padding sorts last
    }
  }
  wave_bitonic_sort<C>(v, lane);
  const uint32_t prev_last = (uint32_t)__shfl_up((int)v[C - 1], 1, kWave);
  int cnt = 0;
#pragma unroll
  for (int r = 0; r < C; ++r) {
    const int i = lane * C + r;
    if (i < n) {
      dst[i] = v[r];
      const uint32_t prev = r ? v[r - 1] : prev_last;
      cnt += (i == 0 || (v[r] >> 4) != (prev >> 4)) ? 1 : 0;
    }
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, kWave);
  return cnt;
}

// Windows above kWsKeys edges are queued for the workgroup kernels: queue[0 .. *queue_count) (workspace).  Order inside
// the queue is irrelevant (windows are independent, results are deterministic).
// (Measured: splitting this kernel into a low-VGPR pass for n <= 512 and a second pass for the rest does not pay --
// the small-window sort is VALU-bound, not latency-bound.)
static __global__ __launch_bounds__(kWsWaves* kWave) void csr_wave_sort_kernel(const int* __restrict__ indptr,
                                                                        const int* __restrict__ indices,
                                                                        const int num_nodes, const int num_windows,
                                                                        const unsigned col_limit,
                                                                        uint32_t* __restrict__ keys_ws,
                                                                        int* __restrict__ block_partition,
                                                                        int* __restrict__ status,
                                                                        int* __restrict__ queue_count,
                                                                        int* __restrict__ queue) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
  for (int w = blockIdx.x * kWsWaves + wv; w < num_windows; w += gridDim.x * kWsWaves) {
    int rp[kBlkH + 1];
    load_window_rowptr(indptr, w, num_nodes, rp);
    const int lo = rp[0], n = rp[kBlkH] - lo;
    if (n > kWsKeys) {  // csr_window_sort_kernel's share
      if (lane == 0) queue[atomicAdd(queue_count, 1)] = w;
      continue;
    }
    uint32_t* const dst = keys_ws + lo;
    int u;  // distinct columns (wave-uniform)
    if (n <= 4 * kWave) u = wave_sort_window<4>(indices, lo, n, rp, col_limit, status, dst, lane);
    else if (n <= 8 * kWave) u = wave_sort_window<8>(indices, lo, n, rp, col_limit, status, dst, lane);
    else if (n <= 16 * kWave) u = wave_sort_window<16>(indices, lo, n, rp, col_limit, status, dst, lane);
    else u = wave_sort_window<32>(indices, lo, n, rp, col_limit, status, dst, lane);
    if (lane == 0) block_partition[w] = n == 0 ? 1 : (u + kBlkW - 1) / kBlkW;  // empty window -> 1 (reference quirk)
  }
}

static __global__ __launch_bounds__(kWsWaves* kWave) void csr_wave_fill_kernel(const int* __restrict__ indptr,
                                                                        const int num_nodes, const int num_windows,
                                                                        const uint32_t* __restrict__ keys_ws,
                                                                        const int* __restrict__ pointer1,
                                                                        uint32_t* __restrict__ hspa_packed,
                                                                        int* __restrict__ hind) {
  __shared__ uint4 lstage4[kWsWaves][kWsKeys / kBlkW];  // packed words of up to kWsKeys / 8 TC blocks per wave
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
  uint4* const stage4 = lstage4[wv];
  uint32_t* const stage = reinterpret_cast<uint32_t*>(stage4);
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  const WaveSync sync;
  for (int i = lane; i < kWsKeys / kBlkW; i += kWave) stage4[i] = zero4;
  sync();
  for (int w = blockIdx.x * kWsWaves + wv; w < num_windows; w += gridDim.x * kWsWaves) {
    const long long r0 = (long long)w * kBlkH, r1 = r0 + kBlkH;
    const int lo = indptr[r0 < num_nodes ? r0 : num_nodes];
    const int n = indptr[r1 < num_nodes ? r1 : num_nodes] - lo;
    if (n > kWsKeys) continue;  // csr_window_fill_kernel's share (queued by csr_wave_sort_kernel)
    const long long p0 = pointer1[w];
    const int nb = pointer1[w + 1] - (int)p0;
    int carry = 0;  // distinct columns before `base` (wave-uniform)
    for (int base = 0; base < n; base += kWave * kWsBatch) {
      uint32_t keys[kWsBatch], prevs[kWsBatch];
#pragma unroll
      for (int u = 0; u < kWsBatch; ++u) {
        const int i = base + u * kWave + lane;
        keys[u] = i < n ? keys_ws[lo + i] : 0u;
        prevs[u] = (i < n && i > 0) ? keys_ws[lo + i - 1] : 0u;
      }
#pragma unroll
      for (int u = 0; u < kWsBatch; ++u) {
        const int i = base + u * kWave + lane;
        const bool valid = i < n;
        const uint32_t key = keys[u];
        const bool first = valid && (i == 0 || (key >> 4) != (prevs[u] >> 4));  // first key of its column
        const unsigned long long m = __ballot(first);
        if (valid) {
          const int rank = carry + __popcll(m & (~0ull >> (kWave - 1 - lane))) - 1;  // reference edgeToColumn
          const int r = key & 15, c = rank & 7, b = rank >> 3;
          if (first) hind[8 * (p0 + b) + c] = (int)(key >> 4);
          // reference bit order (bmat_kernels.cuh:180-188): word t = (r>>3) + 2*(c>>2), bit 4*(r&7) + (c&3)
          atomicOr(&stage[4 * b + (r >> 3) + 2 * (c >> 2)], 1u << (4 * (r & 7) + (c & 3)));
        }
        carry += __popcll(m);
      }
    }
    for (int q = carry + lane; q < 8 * nb; q += kWave) hind[8 * p0 + q] = 0;  // unused slots of the last block
    sync();
    uint4* const out4 = reinterpret_cast<uint4*>(hspa_packed) + p0;
    for (int i = lane; i < nb; i += kWave) {
      out4[i] = stage4[i];
      stage4[i] = zero4;
    }
    sync();
  }
}

static __global__ __launch_bounds__(kSortThreads) void csr_window_sort_kernel(const int* __restrict__ indptr,
                                                                       const int* __restrict__ indices,
                                                                       const int num_nodes, const int num_windows,
                                                                       const unsigned col_limit,
                                                                       uint32_t* __restrict__ keys_ws,
                                                                       int* __restrict__ block_partition,
                                                                       int* __restrict__ status,
                                                                       const int* __restrict__ counts,
                                                                       const int* __restrict__ queue) {
  __shared__ uint32_t lkeys[kSortLdsKeys];
  __shared__ int rowptr[kBlkH + 1];
  __shared__ int wave_cnt[kSortThreads / kWave];
  const int tid = threadIdx.x;
  const int num_big = *counts;  // windows above kWsKeys edges, queued by csr_wave_sort_kernel
  for (int q = blockIdx.x; q < num_big; q += gridDim.x) {
    const int w = queue[q];
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
      const uint32_t col = (uint32_t)indices[e];
      if (col >= col_limit) atomicAdd(status, 1);  // outside the caller's column universe / the 28-bit key
      const uint32_t key = (col << 4) | (uint32_t)rl;
      if (in_lds) lkeys[i] = key; else dst[i] = key;
    }
    __syncthreads();
    // chunks of kWsKeys keys: sorted by one wave each in registers (stages k <= kWsKeys of the network) ...
    auto sort_chunks = [&](auto keys) {
      const int lane = tid & (kWave - 1), wv = tid / kWave;
      for (int c = wv; c * kWsKeys < n; c += kSortThreads / kWave) {
        constexpr int C = kWsKeys / kWave;
        uint32_t v[C];
        const int base = c * kWsKeys + lane * C;
#pragma unroll
        for (int r = 0; r < C; ++r) v[r] = base + r < n ? keys[base + r] : 0xFFFFFFFFu;
        wave_bitonic_sort<C>(v, lane);
#pragma unroll
        for (int r = 0; r < C; ++r)
          if (base + r < n) keys[base + r] = v[r];
      }
      __syncthreads();
    };
    // ... then the remaining merge stages (k = 2 * kWsKeys and up) by the whole workgroup
    constexpr int kFirstStage = 12;
    static_assert((1 << (kFirstStage - 1)) == kWsKeys, "chunk size = 2^(first workgroup stage - 1)");
    if (in_lds) {
      sort_chunks(lkeys);
      bitonic_sort_ascending<kSortThreads>(lkeys, n, tid, WorkgroupSync{}, kFirstStage);
    } else {
      sort_chunks(dst);
      bitonic_sort_ascending<kSortThreads>(dst, n, tid, WorkgroupSync{}, kFirstStage);
    }

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
static __global__ __launch_bounds__(kSortThreads) void csr_window_fill_kernel(const int* __restrict__ indptr,
                                                                       const int num_nodes, const int num_windows,
                                                                       const uint32_t* __restrict__ keys_ws,
                                                                       const int* __restrict__ pointer1,
                                                                       uint32_t* __restrict__ hspa_packed,
                                                                       int* __restrict__ hind,
                                                                       const int* __restrict__ counts,
                                                                       const int* __restrict__ queue) {
  __shared__ int wave_tot[kSortThreads / kWave];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const int num_big = *counts;
  for (int q = blockIdx.x; q < num_big; q += gridDim.x) {
    const int w = queue[q];
    const long long r0 = (long long)w * kBlkH, r1 = r0 + kBlkH;
    const int lo = indptr[r0 < num_nodes ? r0 : num_nodes];
    const int n = indptr[r1 < num_nodes ? r1 : num_nodes] - lo;
    const long long p0 = pointer1[w];
    {  // zero this window's part of the handle; the barrier (stores acknowledged by L2, same CU) orders it before the ORs
      const int nb = pointer1[w + 1] - (int)p0;
      uint4* const z4 = reinterpret_cast<uint4*>(hspa_packed) + p0;
      uint4* const h4 = reinterpret_cast<uint4*>(hind) + 2 * p0;
      for (int i = tid; i < nb; i += kSortThreads) z4[i] = make_uint4(0u, 0u, 0u, 0u);
      for (int i = tid; i < 2 * nb; i += kSortThreads) h4[i] = make_uint4(0u, 0u, 0u, 0u);
      __syncthreads();
    }
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

// ---- mid-size windows (kWsKeys < n <= kSortLdsKeys edges): bucket ranking, O(n) ------------------------------------
// The bitonic network costs n lg^2 n / 2 compare-exchanges (a 4 k-edge window: ~120 us of a 256-thread workgroup).  Ranks
// do not need a sort: spread the keys over NB = 4096 column buckets of equal width (range [cmin, cmax] of the window,
// width a power of two: bucket order = column order), one LDS counter per bucket -- the returning atomic add is the key's
// slot inside its bucket -- an exclusive scan of the counters, and the keys land bucket by bucket.  Inside a bucket (a few
// keys) "first key of its column" and "distinct columns below mine" are pair comparisons.  Count kernel: distinct columns
// = number of first flags; the bucket-ordered keys, first flag in bit 31, go to the key workspace.  Fill kernel: prefix
// sum of the flags = rank of every bucket's first column, + the smaller first columns of the own bucket.
// Windows whose columns cluster (sum of squared bucket sizes above kBkPairsPerKey x n: the pair loops would cost more
// than the sort -- band graphs put half a window's edges into a handful of buckets) go on to the workgroup sort through
// one queue, windows above kSortLdsKeys edges to the next kernels through another.
// Needs column ids below 2^27 (bit 31 of the key is the flag).
constexpr int kBkThreads = 256;
constexpr int kBkBuckets = 4096;
constexpr int kBkItems = kSortLdsKeys / kBkThreads;   // 32 keys per thread
constexpr int kBkPairsPerKey = 8;    // clustering test: sum of squared bucket sizes <= 8 n (uniform columns: 1-3 n)
constexpr unsigned kBkMaxCols = 1u << 27;
constexpr uint32_t kBkFirst = 0x80000000u;

struct BkLds {
  uint32_t keys[kSortLdsKeys];
  int start[kBkBuckets + 16];   // counters, then exclusive offsets; start[kBkBuckets] = n
  int rowptr[kBlkH + 1];
  int wsum[kBkThreads / kWave];
  long long wsq[kBkThreads / kWave];
  unsigned wmin[kBkThreads / kWave], wmax[kBkThreads / kWave];
};

// smallest shift with ((cmax - cmin) >> shift) < kBkBuckets
__device__ __forceinline__ int bk_shift(const unsigned span) {
  int shift = 0;
  while ((span >> shift) >= (unsigned)kBkBuckets) ++shift;
  return shift;
}

// workgroup min / max of the window's columns (every thread gets them)
template <class Lds>
__device__ __forceinline__ void bk_min_max(Lds& s, unsigned lo, unsigned hi, unsigned& cmin, unsigned& cmax) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    const unsigned a = (unsigned)__shfl_xor((int)lo, off, kWave), b = (unsigned)__shfl_xor((int)hi, off, kWave);
    lo = a < lo ? a : lo;
    hi = b > hi ? b : hi;
  }
  if ((threadIdx.x & (kWave - 1)) == 0) {
    s.wmin[threadIdx.x / kWave] = lo;
    s.wmax[threadIdx.x / kWave] = hi;
  }
  __syncthreads();
  cmin = s.wmin[0];
  cmax = s.wmax[0];
#pragma unroll
  for (int k = 1; k < kBkThreads / kWave; ++k) {
    cmin = s.wmin[k] < cmin ? s.wmin[k] : cmin;
    cmax = s.wmax[k] > cmax ? s.wmax[k] : cmax;
  }
}

static __global__ __launch_bounds__(kBkThreads) void csr_bucket_count_kernel(
    const int* __restrict__ indptr, const int* __restrict__ indices, const int num_nodes, const unsigned col_limit,
    uint32_t* __restrict__ keys_ws, int* __restrict__ block_partition, int* __restrict__ status,
    int* __restrict__ counts /* [0] in, [1] too big, [2] done, [3] clustered */, const int* __restrict__ queue,
    int* __restrict__ queue_big, int* __restrict__ queue_done, int* __restrict__ queue_clustered) {
  __shared__ BkLds s;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const int todo = counts[0];
  for (int q = blockIdx.x; q < todo; q += gridDim.x) {
    const int w = queue[q];
    if (tid <= kBlkH) {
      const long long r = (long long)w * kBlkH + tid;
      s.rowptr[tid] = indptr[r < num_nodes ? r : num_nodes];
    }
    for (int i = tid; i < kBkBuckets / 4; i += kBkThreads) reinterpret_cast<uint4*>(s.start)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const int lo = s.rowptr[0], n = s.rowptr[kBlkH] - lo;
    if (n > kSortLdsKeys) {  // workgroup-uniform: the next kernels' share
      if (tid == 0) queue_big[atomicAdd(&counts[1], 1)] = w;
      __syncthreads();
      continue;
    }
    // three sweeps over the window's edges (the first from HBM, the others out of L2): range, histogram, scatter --
    // nothing per edge is kept in registers between them.  fn(i, column) for every edge i < n, 8 loads in flight.
    auto for_each_edge = [&](auto fn) {
      for (int base = tid; base < n; base += 8 * kBkThreads) {
        uint32_t c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = base + k * kBkThreads < n ? (uint32_t)indices[lo + base + k * kBkThreads] : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (base + k * kBkThreads < n) fn(base + k * kBkThreads, c[k]);
      }
    };
    unsigned mn = 0xFFFFFFFFu, mx = 0u;
    int bad = 0;
    for_each_edge([&](const int, const uint32_t col) {
      bad += col >= col_limit ? 1 : 0;
      const uint32_t m = col & (kBkMaxCols - 1u);  // ids outside the universe are reported (status); arithmetic stays in range
      mn = m < mn ? m : mn;
      mx = m > mx ? m : mx;
    });
    unsigned cmin, cmax;
    bk_min_max(s, mn, mx, cmin, cmax);
    const int shift = bk_shift(cmax - cmin);
    for_each_edge([&](const int, const uint32_t col) { atomicAdd(&s.start[((col & (kBkMaxCols - 1u)) - cmin) >> shift], 1); });
    __syncthreads();
    // exclusive scan of the counters (16 per thread), sum of squares for the clustering test
    int c[16];
    int tot = 0;
    long long sq = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      c[k] = s.start[16 * tid + k];
      sq += (long long)c[k] * c[k];
      tot += c[k];
    }
    const int inc = wave_inclusive_scan(tot);
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) sq += __shfl_xor(sq, off, kWave);
    if (lane == kWave - 1) s.wsum[wv] = inc;
    if (lane == 0) s.wsq[wv] = sq;
    __syncthreads();
    int run = inc - tot;
    long long pairs = 0;
#pragma unroll
    for (int k = 0; k < kBkThreads / kWave; ++k) {
      run += k < wv ? s.wsum[k] : 0;
      pairs += s.wsq[k];
    }
    if (pairs > (long long)kBkPairsPerKey * n) {  // workgroup-uniform: clustered columns -> workgroup sort
      if (tid == 0) queue_clustered[atomicAdd(&counts[3], 1)] = w;
      __syncthreads();
      continue;
    }
    if (bad) atomicAdd(status, bad);   // this kernel keeps the window: its out-of-universe ids are reported here, once
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      s.start[16 * tid + k] = run;
      run += c[k];
    }
    __syncthreads();
    // scatter: the returning add hands out the positions of a bucket; afterwards start[b] = END of bucket b (= start of b + 1)
    for_each_edge([&](const int i, const uint32_t col) {
      const uint32_t m = col & (kBkMaxCols - 1u);
      int rl = 0;
#pragma unroll
      for (int k = 1; k < kBlkH; ++k) rl += (s.rowptr[k] <= lo + i) ? 1 : 0;
      s.keys[atomicAdd(&s.start[(m - cmin) >> shift], 1)] = (m << 4) | (uint32_t)rl;
    });
    __syncthreads();
    // first key of its column inside the bucket = no equal column at a lower position of the bucket.  Readers mask the
    // flag bit, so the owner of a position may set it while others still compare against that key.
    int firsts = 0;
    for (int p = tid; p < n; p += kBkThreads) {
      const uint32_t k = s.keys[p];
      const uint32_t col = k >> 4;
      const int bkt = (col - cmin) >> shift;
      bool first = true;
      for (int t = bkt ? s.start[bkt - 1] : 0; t < p; ++t) first = first && (((s.keys[t] & ~kBkFirst) >> 4) != col);
      firsts += first ? 1 : 0;
      if (first) s.keys[p] = k | kBkFirst;
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) firsts += __shfl_xor(firsts, off, kWave);
    if (lane == 0) s.wsum[wv] = firsts;
    __syncthreads();
    uint32_t* const dst = keys_ws + lo;
    for (int i = tid; i < n; i += kBkThreads) dst[i] = s.keys[i];
    if (tid == 0) {
      int u = 0;
#pragma unroll
      for (int k = 0; k < kBkThreads / kWave; ++k) u += s.wsum[k];
      block_partition[w] = (u + kBlkW - 1) / kBlkW;   // n > kWsKeys here: never the empty-window case
      queue_done[atomicAdd(&counts[2], 1)] = w;
    }
    __syncthreads();
  }
}

struct BkFillLds {
  uint32_t keys[kSortLdsKeys];
  uint16_t pref[kSortLdsKeys + 2];   // first flags below the position
  uint4 stage[kSortLdsKeys / kBlkW]; // packed words of up to 1024 TC blocks
  int wsum[kBkThreads / kWave];
  unsigned wmin[kBkThreads / kWave], wmax[kBkThreads / kWave];
};

static __global__ __launch_bounds__(kBkThreads) void csr_bucket_fill_kernel(
    const int* __restrict__ indptr, const int num_nodes, const uint32_t* __restrict__ keys_ws,
    const int* __restrict__ pointer1, uint32_t* __restrict__ hspa_packed, int* __restrict__ hind,
    const int* __restrict__ counts /* [2] = windows in queue_done */, const int* __restrict__ queue_done) {
  extern __shared__ __attribute__((aligned(16))) char bk_smem[];
  BkFillLds& s = *reinterpret_cast<BkFillLds*>(bk_smem);
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  for (int i = tid; i < kSortLdsKeys / kBlkW; i += kBkThreads) s.stage[i] = zero4;
  __syncthreads();
  const int todo = counts[2];
  for (int q = blockIdx.x; q < todo; q += gridDim.x) {
    const int w = queue_done[q];
    const long long r0 = (long long)w * kBlkH, r1 = r0 + kBlkH;
    const int lo = indptr[r0 < num_nodes ? r0 : num_nodes];
    const int n = indptr[r1 < num_nodes ? r1 : num_nodes] - lo;
    const long long p0 = pointer1[w];
    const int nb = pointer1[w + 1] - (int)p0;
    unsigned mn = 0xFFFFFFFFu, mx = 0u;
    for (int i = tid; i < n; i += kBkThreads) {
      const uint32_t k = keys_ws[lo + i];
      s.keys[i] = k;
      const unsigned col = (k & ~kBkFirst) >> 4;
      mn = col < mn ? col : mn;
      mx = col > mx ? col : mx;
    }
    unsigned cmin, cmax;
    bk_min_max(s, mn, mx, cmin, cmax);
    const int shift = bk_shift(cmax - cmin);
    // exclusive prefix of the first flags: thread tid owns positions [tid * per, (tid + 1) * per)
    const int per = (n + kBkThreads - 1) / kBkThreads;
    const int a = tid * per < n ? tid * per : n, b = a + per < n ? a + per : n;
    int mine = 0;
    for (int p = a; p < b; ++p) mine += (s.keys[p] & kBkFirst) ? 1 : 0;
    const int inc = wave_inclusive_scan(mine);
    if (lane == kWave - 1) s.wsum[wv] = inc;
    __syncthreads();
    int run = inc - mine;
#pragma unroll
    for (int k = 0; k < kBkThreads / kWave; ++k) run += k < wv ? s.wsum[k] : 0;
    for (int p = a; p < b; ++p) {
      s.pref[p] = (uint16_t)run;
      run += (s.keys[p] & kBkFirst) ? 1 : 0;
    }
    __syncthreads();
    int* const hind_w = hind + 8 * p0;
    uint32_t* const stage = reinterpret_cast<uint32_t*>(s.stage);
    for (int p = tid; p < n; p += kBkThreads) {
      const uint32_t k = s.keys[p];
      const unsigned col = (k & ~kBkFirst) >> 4;
      const unsigned bkt = (col - cmin) >> shift;
      int lo_p = p;  // first position of the bucket
      while (lo_p > 0 && ((((s.keys[lo_p - 1] & ~kBkFirst) >> 4) - cmin) >> shift) == bkt) --lo_p;
      int rank = s.pref[lo_p];
      for (int t = lo_p; t < n; ++t) {
        const uint32_t kt = s.keys[t];
        const unsigned ct = (kt & ~kBkFirst) >> 4;
        if (((ct - cmin) >> shift) != bkt) break;
        rank += ((kt & kBkFirst) && ct < col) ? 1 : 0;
      }
      if (k & kBkFirst) hind_w[rank] = (int)col;
      const int rl = k & 15, c = rank & 7;
      // reference bit order (bmat_kernels.cuh:180-188): word t = (r>>3) + 2*(c>>2), bit 4*(r&7) + (c&3)
      atomicOr(&stage[4 * (rank >> 3) + (rl >> 3) + 2 * (c >> 2)], 1u << (4 * (rl & 7) + (c & 3)));
    }
    __syncthreads();
    uint4* const out4 = reinterpret_cast<uint4*>(hspa_packed) + p0;
    for (int i = tid; i < nb; i += kBkThreads) {
      out4[i] = s.stage[i];
      s.stage[i] = zero4;
    }
    // distinct columns = first flags in total = prefix at the last position + its flag
    const int u = n > 0 ? (int)s.pref[n - 1] + ((s.keys[n - 1] & kBkFirst) ? 1 : 0) : 0;
    for (int k = u + tid; k < 8 * nb; k += kBkThreads) hind_w[k] = 0;  // unused slots of the last block
    __syncthreads();
  }
}

// ---- bitmap path: rank columns with an LDS bitmap instead of a sort ---------------------------------------------
// When the column universe fits LDS (num_cols <= kBmMaxCols) and is not much larger than a window's edge list, the
// condensed column of an edge is a population count: every edge of the window sets bit `column` of an LDS bitmap;
// rank(c) = #set bits below c = prefix[c / 128] + popcount inside the 128-bit group.  Work per window is
// O(edges + num_nodes / 32) LDS operations, no sort, no key workspace, and the handle is written exactly once
// (bitmap words of a window are staged in LDS, hind comes out of the bitmap scan in rank order).  Same output, bit
// for bit, as the sort path and as the reference pipeline.
// threads per workgroup: 512 when every window goes through these kernels (single range: 1024 costs 45 % on the
// reddit-like graph, barriers), 1024 for the listed big windows of the mixed path (their sweeps are round-trip-bound:
// power-law 4 M 84 -> 73 ms)
constexpr int kBmThreadsAll = 512;
constexpr int kBmThreadsListed = 1024;
constexpr int kBmStageWords = 4096;      // 16 KiB: the packed words of kBmStageBlocks TC blocks / one sweep of hind
constexpr int kBmStageBlocks = kBmStageWords / 4;
constexpr int kBmGrid = 4096;            // grid-strided; ~5 rounds of workgroups even out the window sizes (measured)
constexpr int kBmMaxCols = 1 << 19;      // 64 KiB bitmap + 16 KiB group prefixes + 16 KiB stage

__host__ __device__ inline int bm_groups(int num_cols) { return (num_cols + 127) / 128; }   // 128 columns = uint4
inline size_t bm_count_lds(int num_cols) { return (size_t)bm_groups(num_cols < kBmMaxCols ? num_cols : kBmMaxCols) * 16; }
inline size_t bm_fill_lds(int num_cols) {
  return (size_t)bm_groups(num_cols < kBmMaxCols ? num_cols : kBmMaxCols) * 20 + kBmStageWords * 4;
}

__device__ __forceinline__ int popc4(const uint4 v) { return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }

constexpr int kBmBatch = 8;  // independent global loads in flight per thread (the loops are latency-bound otherwise)

// Universes above kBmMaxCols are covered in column RANGES of kBmMaxCols: one bitmap pass per range, the distinct-column
// count carried from range to range (the window's edges are re-read from L2 by every pass).
__host__ __device__ inline int bm_range_cols(int num_cols) { return num_cols < kBmMaxCols ? num_cols : kBmMaxCols; }

// marks bit (c - c0) for every edge of the window with c in [c0, c0 + range)
template <int kBmThreads, bool COUNT_INVALID>
__device__ __forceinline__ void bm_mark_window(uint32_t* bitmap, const int* __restrict__ indices, const long long lo,
                                               const long long hi, const int c0, const int range, const int num_cols,
                                               int* status) {
  for (long long base = lo + threadIdx.x; base < hi; base += (long long)kBmBatch * kBmThreads) {
    int c[kBmBatch];
#pragma unroll
    for (int k = 0; k < kBmBatch; ++k) {
      const long long e = base + (long long)k * kBmThreads;
      c[k] = e < hi ? indices[e] : 0;
    }
#pragma unroll
    for (int k = 0; k < kBmBatch; ++k) {
      if (base + (long long)k * kBmThreads < hi) {
        const unsigned rel = (unsigned)(c[k] - c0);
        if (rel < (unsigned)range) atomicOr(&bitmap[rel >> 5], 1u << (rel & 31));
        else if (COUNT_INVALID && c0 == 0 && (unsigned)c[k] >= (unsigned)num_cols)
          atomicAdd(status, 1);  // ids outside the column universe are skipped and reported (once: first range)
      }
    }
  }
}

// ---- listed (big) windows of the mixed path: the window's edges grouped by column range, once --------------------
// Every range pass of the plain kernels sweeps ALL edges of the window (count 1 + fill 2 sweeps per range: 24 n edge reads
// for 8 ranges).  For the listed windows the count kernel first writes the keys (column << 4 | local row) into the key
// workspace grouped by range -- per-wave LDS counters, two sweeps -- and group_ptr[q][0 .. R] (q = position in the list);
// afterwards every pass reads only its own group (count: 3 n edge reads, fill: 2 n).  Rows need not be sorted.
constexpr int kBmMaxGroups = 16;   // = kMixedMaxPasses: the mixed path takes universes of up to 16 ranges

// marks bit (column - c0) for the keys of one group
template <int kBmThreads>
__device__ __forceinline__ void bm_mark_keys(uint32_t* bitmap, const uint32_t* __restrict__ keys, const int count,
                                             const int c0) {
  for (int base = threadIdx.x; base < count; base += kBmBatch * kBmThreads) {
    uint32_t k[kBmBatch];
#pragma unroll
    for (int u = 0; u < kBmBatch; ++u) k[u] = base + u * kBmThreads < count ? keys[base + u * kBmThreads] : 0u;
#pragma unroll
    for (int u = 0; u < kBmBatch; ++u) {
      if (base + u * kBmThreads < count) {
        const unsigned rel = (k[u] >> 4) - (unsigned)c0;
        atomicOr(&bitmap[rel >> 5], 1u << (rel & 31));
      }
    }
  }
}

// keys_out[0 .. valid) <- the window's edges grouped by column range (range = column / range_cols); gptr[0 .. R] <- group
// boundaries; ids outside [0, num_cols) are dropped and counted.  cnt = LDS int[waves][kBmMaxGroups] scratch.
template <int kBmThreads>
__device__ __forceinline__ void bm_group_by_range(const int* __restrict__ indices, const int (&rp)[kBlkH + 1],
                                                  const int num_cols, const int range_shift, const int nranges,
                                                  uint32_t* __restrict__ keys_out, int* __restrict__ gptr,
                                                  int* __restrict__ status, int* cnt /* LDS */) {
  constexpr int kWaves = kBmThreads / kWave;
  const int tid = threadIdx.x, wv = tid / kWave;
  const int lo = rp[0], hi = rp[kBlkH];
  for (int i = tid; i < kWaves * kBmMaxGroups; i += kBmThreads) cnt[i] = 0;
  __syncthreads();
  int bad = 0;
  // sweep 1: edges per (wave, range); the same thread handles the same edges in both sweeps
  for (int base = lo + tid; base < hi; base += kBmBatch * kBmThreads) {
    int c[kBmBatch];
#pragma unroll
    for (int u = 0; u < kBmBatch; ++u) c[u] = base + u * kBmThreads < hi ? indices[base + u * kBmThreads] : -1;
#pragma unroll
    for (int u = 0; u < kBmBatch; ++u) {
      if (base + u * kBmThreads < hi) {
        if ((unsigned)c[u] < (unsigned)num_cols) atomicAdd(&cnt[wv * kBmMaxGroups + (c[u] >> range_shift)], 1);
        else ++bad;
      }
    }
  }
  if (bad) atomicAdd(status, bad);
  __syncthreads();
  // exclusive scan, range-major / wave-minor (one thread: waves x ranges <= 256 values)
  if (tid == 0) {
    int run = 0;
    for (int r = 0; r < nranges; ++r) {
      gptr[r] = run;
      for (int v = 0; v < kWaves; ++v) {
        const int t = cnt[v * kBmMaxGroups + r];
        cnt[v * kBmMaxGroups + r] = run;
        run += t;
      }
    }
    gptr[nranges] = run;
  }
  __syncthreads();
  // sweep 2: scatter
  for (int base = lo + tid; base < hi; base += kBmBatch * kBmThreads) {
    int c[kBmBatch];
#pragma unroll
    for (int u = 0; u < kBmBatch; ++u) c[u] = base + u * kBmThreads < hi ? indices[base + u * kBmThreads] : -1;
#pragma unroll
    for (int u = 0; u < kBmBatch; ++u) {
      const int e = base + u * kBmThreads;
      if (e < hi && (unsigned)c[u] < (unsigned)num_cols) {
        const int pos = atomicAdd(&cnt[wv * kBmMaxGroups + (c[u] >> range_shift)], 1);
        keys_out[pos] = ((uint32_t)c[u] << 4) | (uint32_t)local_row(rp, e);
      }
    }
  }
  __syncthreads();
}

template <int kBmThreads>
static __global__ __launch_bounds__(kBmThreads) void csr_bitmap_count_kernel(const int* __restrict__ indptr,
                                                                      const int* __restrict__ indices,
                                                                      const int num_nodes, const int num_cols,
                                                                      const int num_windows,
                                                                      int* __restrict__ block_partition,
                                                                      int* __restrict__ status,
                                                                      const int* __restrict__ list,
                                                                      const int* __restrict__ list_count,
                                                                      uint32_t* __restrict__ keys_ws,
                                                                      int* __restrict__ group_ptr) {
  constexpr int kBmWaves = kBmThreads / kWave;
  extern __shared__ uint4 bm_lds[];
  __shared__ int wave_cnt[kBmWaves];
  __shared__ int group_cnt[kBmWaves * kBmMaxGroups];
  __shared__ int gptr_s[kBmMaxGroups + 1];
  uint4* const bitmap4 = bm_lds;
  uint32_t* const bitmap = reinterpret_cast<uint32_t*>(bm_lds);
  const int range_cols = bm_range_cols(num_cols);
  const int ng = bm_groups(range_cols);
  const int tid = threadIdx.x;
  for (int i = tid; i < ng; i += kBmThreads) bitmap4[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  const int todo = list ? *list_count : num_windows;  // mixed path: only the listed windows
  for (int q = blockIdx.x; q < todo; q += gridDim.x) {
    const int w = list ? list[q] : q;
    const long long r0 = (long long)w * kBlkH, r1 = r0 + kBlkH;
    const long long lo = indptr[r0 < num_nodes ? r0 : num_nodes], hi = indptr[r1 < num_nodes ? r1 : num_nodes];
    const bool grouped = list != nullptr && group_ptr != nullptr;  // kernel-uniform
    if (grouped) {
      int rp[kBlkH + 1];
      load_window_rowptr(indptr, w, num_nodes, rp);
      const int nranges = (num_cols + range_cols - 1) / range_cols;
      // several ranges: range_cols = 2^19; a single range (forced mixed path on a small universe): everything in group 0
      bm_group_by_range<kBmThreads>(indices, rp, num_cols, nranges > 1 ? 31 - __clz(range_cols) : 31, nranges,
                                    keys_ws + lo, gptr_s, status, group_cnt);
      if (tid <= nranges) group_ptr[(long long)q * (kBmMaxGroups + 1) + tid] = gptr_s[tid];
    }
    int cnt = 0;
    for (int c0 = 0, r = 0; c0 < num_cols; c0 += range_cols, ++r) {
      const int range = num_cols - c0 < range_cols ? num_cols - c0 : range_cols;
      if (grouped) bm_mark_keys<kBmThreads>(bitmap, keys_ws + lo + gptr_s[r], gptr_s[r + 1] - gptr_s[r], c0);
      else bm_mark_window<kBmThreads, true>(bitmap, indices, lo, hi, c0, range, num_cols, status);
      __syncthreads();
      for (int i = tid; i < ng; i += kBmThreads) {  // count and clear in one sweep
        cnt += popc4(bitmap4[i]);
        bitmap4[i] = make_uint4(0u, 0u, 0u, 0u);
      }
      __syncthreads();
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, kWave);
    if ((tid & (kWave - 1)) == 0) wave_cnt[tid / kWave] = cnt;
    __syncthreads();
    if (tid == 0) {
      int u = 0;
#pragma unroll
      for (int i = 0; i < kBmWaves; ++i) u += wave_cnt[i];
      block_partition[w] = u == 0 ? 1 : (u + kBlkW - 1) / kBlkW;  // no columns -> 1 block (reference quirk, :252)
    }
    __syncthreads();
  }
}

template <int kBmThreads>
static __global__ __launch_bounds__(kBmThreads) void csr_bitmap_fill_kernel(const int* __restrict__ indptr,
                                                                     const int* __restrict__ indices,
                                                                     const int num_nodes, const int num_cols,
                                                                     const int num_windows,
                                                                     const int* __restrict__ pointer1,
                                                                     uint32_t* __restrict__ hspa_packed,
                                                                     int* __restrict__ hind,
                                                                     const int* __restrict__ list,
                                                                     const int* __restrict__ list_count,
                                                                     const uint32_t* __restrict__ keys_ws,
                                                                     const int* __restrict__ group_ptr) {
  constexpr int kBmWaves = kBmThreads / kWave;
  __shared__ int gptr_s[kBmMaxGroups + 1];
  constexpr int kBmSweeps = kBmMaxCols / 128 / kBmThreads;  // 128-column groups per thread in a full range
  static_assert(kBmSweeps * kBmThreads * 128 == kBmMaxCols, "range = whole sweeps");
  extern __shared__ uint4 bm_lds[];
  __shared__ int wave_tot[kBmSweeps][kBmWaves];
  const int range_cols = bm_range_cols(num_cols);
  const int ng = bm_groups(range_cols);
  uint4* const bitmap4 = bm_lds;                                        // [ng]   128 columns per entry
  uint32_t* const bitmap = reinterpret_cast<uint32_t*>(bm_lds);
  uint4* const stage4 = bm_lds + ng;                                    // [kBmStageBlocks] packed words of TC blocks
  uint32_t* const stage = reinterpret_cast<uint32_t*>(stage4);
  int* const prefix = reinterpret_cast<int*>(stage4 + kBmStageBlocks);  // [ng] distinct columns before the group
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  for (int i = tid; i < ng; i += kBmThreads) bitmap4[i] = zero4;
  for (int i = tid; i < kBmStageBlocks; i += kBmThreads) stage4[i] = zero4;
  __syncthreads();

  const int todo = list ? *list_count : num_windows;  // mixed path: only the listed windows
  for (int q = blockIdx.x; q < todo; q += gridDim.x) {
    const int w = list ? list[q] : q;
    int rp[kBlkH + 1];  // wave-uniform row pointers of the window (scalar loads)
    load_window_rowptr(indptr, w, num_nodes, rp);
    const long long lo = rp[0], hi = rp[kBlkH];
    const long long p0 = pointer1[w];
    const int nb = pointer1[w + 1] - (int)p0;
    uint4* const out4 = reinterpret_cast<uint4*>(hspa_packed) + p0;
    int* const hind_w = hind + 8 * p0;

    // listed windows of the mixed path: the count kernel left the keys grouped by range (bm_group_by_range)
    const bool grouped = list != nullptr && group_ptr != nullptr;  // kernel-uniform
    if (grouped) {
      const int nranges = (num_cols + range_cols - 1) / range_cols;
      if (tid <= nranges) gptr_s[tid] = group_ptr[(long long)q * (kBmMaxGroups + 1) + tid];
      __syncthreads();
    }

    int carry = 0;    // distinct columns of the ranges before the current one
    int flushed = 0;  // TC blocks of this window already written; block `flushed` starts at stage4[0]
    for (int c0 = 0, rho = 0; c0 < num_cols; c0 += range_cols, ++rho) {
      const int range = num_cols - c0 < range_cols ? num_cols - c0 : range_cols;
      const bool last_range = c0 + range_cols >= num_cols;
      const int ngr = bm_groups(range);
      const uint32_t* const gkeys = grouped ? keys_ws + lo + gptr_s[rho] : nullptr;
      const int gcount = grouped ? gptr_s[rho + 1] - gptr_s[rho] : 0;
      if (grouped) bm_mark_keys<kBmThreads>(bitmap, gkeys, gcount, c0);
      else bm_mark_window<kBmThreads, false>(bitmap, indices, lo, hi, c0, range, num_cols, nullptr);
      __syncthreads();

      // scan the bitmap: group prefixes for the rank lookups, and hind = the set bits in ascending (= rank) order.
      // Group i = j * kBmThreads + tid belongs to sweep j; all sweeps are scanned behind ONE barrier.
      const int sweeps = (ngr + kBmThreads - 1) / kBmThreads;  // <= kBmSweeps
      for (int j = 0; j < sweeps; ++j) {
        const int i = j * kBmThreads + tid;
        const int pc = i < ngr ? popc4(bitmap4[i]) : 0;
        const int incl = wave_inclusive_scan(pc);
        if (i < ngr) prefix[i] = incl - pc;  // rank inside this wave's 64 groups; completed below
        if (lane == kWave - 1) wave_tot[j][wv] = incl;
      }
      __syncthreads();
      int seen = carry;
      for (int j = 0; j < sweeps; ++j) {
        int woff = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < kBmWaves; ++k) {
          const int t = wave_tot[j][k];
          woff += k < wv ? t : 0;
          tot += t;
        }
        const int i = j * kBmThreads + tid;
        if (i < ngr) prefix[i] += seen + woff;
        seen += tot;
      }
      const int new_carry = seen;
      // hind: through LDS (coalesced copy-out) when the columns fit BEHIND the pending partial TC block (stage words
      // 0-3) -- the whole range at once, else sweep by sweep, else straight to global memory.  Each thread reads back
      // only its own prefix entries, so no barrier is needed before the emission.
      constexpr int kSpare = kBmStageWords - 4;
      auto sweep_total = [&](const int j) {
        int tot = 0;
#pragma unroll
        for (int k = 0; k < kBmWaves; ++k) tot += wave_tot[j][k];
        return tot;
      };
      auto emit = [&](const int j, const int stage_base) {  // stage_base < 0: global
        const int i = j * kBmThreads + tid;
        if (i < ngr) {
          int r = prefix[i];
          const uint4 v = bitmap4[i];
          const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            uint32_t m = words[q];
            while (m) {
              const int b = __ffs(m) - 1;
              m &= m - 1;
              const int col = c0 + (i * 4 + q) * 32 + b;
              if (stage_base >= 0) stage[4 + r - stage_base] = (uint32_t)col; else hind_w[r] = col;
              ++r;
            }
          }
        }
      };
      auto copy_out = [&](const int first, const int count) {  // stage[4 .. 4 + count) -> hind_w[first ..); re-zero
        __syncthreads();
        for (int k = tid; k < count; k += kBmThreads) {
          hind_w[first + k] = (int)stage[4 + k];
          stage[4 + k] = 0u;
        }
        __syncthreads();
      };
      if (new_carry - carry <= kSpare) {
        for (int j = 0; j < sweeps; ++j) emit(j, carry);
        copy_out(carry, new_carry - carry);
      } else {
        int first = carry;
        for (int j = 0; j < sweeps; ++j) {
          const int tot = sweep_total(j);
          if (tot <= kSpare) {
            emit(j, first);
            copy_out(first, tot);
          } else {
            emit(j, -1);
          }
          first += tot;
        }
        __syncthreads();
      }

      // every edge of the range: rank of its column -> bit (row, rank) in the reference's swizzled word / bit position,
      // OR-ed into the LDS stage (TC blocks [pass0, pass0 + kBmStageBlocks)) and copied out once.  More blocks than the
      // stage holds: one pass over the edges per stage-full (no global atomics, no fences; edges re-read from L2).
      if (new_carry > carry) {
        const int last_blk = (new_carry - 1) >> 3;
        for (int pass0 = flushed;;) {
          // the edges of the range: (column, local row) from the window's CSR segment (every edge, filtered) or from the
          // range's key group
          const long long sweep_lo = grouped ? 0 : lo, sweep_hi = grouped ? gcount : hi;
          for (long long base = sweep_lo + tid; base < sweep_hi; base += (long long)kBmBatch * kBmThreads) {
            int cs[kBmBatch];
#pragma unroll
            for (int k = 0; k < kBmBatch; ++k) {
              const long long e = base + (long long)k * kBmThreads;
              cs[k] = e < sweep_hi ? (grouped ? (int)gkeys[e] : indices[e]) : -1;
            }
#pragma unroll
            for (int k = 0; k < kBmBatch; ++k) {
              const long long e = base + (long long)k * kBmThreads;
              const unsigned rel = (grouped ? (unsigned)cs[k] >> 4 : (unsigned)cs[k]) - (unsigned)c0;
              if (e >= sweep_hi || rel >= (unsigned)range) continue;
              const int g = rel >> 7, wi = (rel >> 5) & 3;
              const uint4 v = bitmap4[g];
              const uint32_t below = (1u << (rel & 31)) - 1u;
              const int rank = prefix[g] + __popc(v.x & (wi > 0 ? ~0u : (wi == 0 ? below : 0u))) +
                               __popc(v.y & (wi > 1 ? ~0u : (wi == 1 ? below : 0u))) +
                               __popc(v.z & (wi > 2 ? ~0u : (wi == 2 ? below : 0u))) +
                               __popc(v.w & (wi == 3 ? below : 0u));
              const int blk = (rank >> 3) - pass0;
              if ((unsigned)blk >= (unsigned)kBmStageBlocks) continue;  // another pass owns this TC block
              const int rl = grouped ? (cs[k] & 15) : local_row(rp, (int)e);
              const int cc = rank & 7;
              // reference bit order (bmat_kernels.cuh:180-188): word t = (r>>3) + 2*(c>>2), bit 4*(r&7) + (c&3)
              atomicOr(&stage[4 * blk + (rl >> 3) + 2 * (cc >> 2)], 1u << (4 * (rl & 7) + (cc & 3)));
            }
          }
          __syncthreads();
          const bool final_pass = pass0 + kBmStageBlocks > last_blk;
          const int pass_end = final_pass ? last_blk + 1 : pass0 + kBmStageBlocks;  // blocks [pass0, pass_end) touched
          // the last touched block is complete unless later ranges can still add columns to it
          const int done_end = (final_pass && !last_range && (new_carry & 7)) ? pass_end - 1 : pass_end;
          for (int i = tid; i < done_end - pass0; i += kBmThreads) {
            out4[pass0 + i] = stage4[i];
            stage4[i] = zero4;
          }
          __syncthreads();
          if (done_end < pass_end && done_end > pass0) {  // keep the pending partial block at stage4[0]
            if (tid == 0) {
              stage4[0] = stage4[done_end - pass0];
              stage4[done_end - pass0] = zero4;
            }
            __syncthreads();
          }
          flushed = done_end;
          if (final_pass) break;
          pass0 = flushed;
        }
      }
      for (int i = tid; i < ngr; i += kBmThreads) bitmap4[i] = zero4;
      carry = new_carry;
      __syncthreads();
    }
    // at most one block is left: the pending partial block at stage4[0] when the ranges after it added nothing, or --
    // no valid column at all -- the window's single all-zero TC block (reference quirk; the stage is zero then)
    for (int i = flushed + tid; i < nb; i += kBmThreads) {
      out4[i] = stage4[i - flushed];
      stage4[i - flushed] = zero4;
    }
    for (int k = carry + tid; k < 8 * nb; k += kBmThreads) hind_w[k] = 0;  // unused slots of the last block
  }
}

// Path choice (the same in workspace_bytes / count / fill).  num_cols = the caller's column universe (every id is in
// [0, num_cols); <= 0: unknown -> sort path).
//   bitmap  one range (universe <= 2^19) whose sweep costs less than sorting the window's edges (reddit);
//   mixed   2 .. 16 ranges: windows up to kSortLdsKeys edges are sorted (a wave in registers / a workgroup in LDS: cost
//           ~ their edges), only the bigger ones pay the per-range sweeps of the bitmap kernels (cost ~ universe).  On
//           the power-law stand-in (4 M columns = 8 ranges; median window 3.2 k edges, 13 % of the windows above 8 k
//           with 55 % of the edges) every window through the bitmap kernels cost ~100 us of sweeps and barriers;
//   sort    unknown universes; more ranges than that: bitmap or sort for every window by the same cost estimate.
// `forced` (an ARGUMENT of the three entry points, the same value in all of them; kCsrAuto = the rule): kCsrSort /
// kCsrBitmap / kCsrMixed override it (bitmap is honoured only when 0 < num_cols <= 2^25, mixed when <= 2^23) -- how the
// tests reach every path on one graph.  Nothing is read from the environment here.
enum CsrPath { kCsrAuto = -1, kCsrSort = 0, kCsrBitmap = 1, kCsrMixed = 2 };
constexpr int kMixedMaxPasses = kBmMaxGroups;

inline CsrPath csr_path(int num_nodes, int num_cols, long long num_edges, int forced = kCsrAuto) {
  if (num_cols <= 0 || num_nodes <= 0) return kCsrSort;
  const long long passes = ((long long)num_cols + kBmMaxCols - 1) / kBmMaxCols;  // column ranges per window
  if (passes > 64) return kCsrSort;
  if (forced == kCsrSort) return kCsrSort;
  if (forced == kCsrBitmap) return kCsrBitmap;
  if (forced == kCsrMixed && passes <= kMixedMaxPasses) return kCsrMixed;   // the range groups of the listed windows: <= 16
  if (passes > 1 && passes <= kMixedMaxPasses) return kCsrMixed;
  const long long W = ((long long)num_nodes + kBlkH - 1) / kBlkH;
  const double per_window = (double)num_edges / (double)W;       // mean edges per window
  // rough per-window costs in LDS word operations: a bitmap pass sweeps its range's 128-column groups and touches every
  // edge; the register sort of a window <= kWsKeys edges is cheap, the workgroup sort pays ~lg^2 / 2 sweeps of the keys
  const double range_groups = (double)(num_cols < kBmMaxCols ? num_cols : kBmMaxCols) / 128.0;
  const double bitmap_cost = (double)passes * (range_groups + per_window);
  double lg = 1.0;
  for (double x = 2.0; x < per_window; x *= 2.0) lg += 1.0;      // ~ log2
  const double sort_cost = per_window <= (double)kWsKeys ? 2.0 * per_window : per_window * lg * (lg + 1.0) / 8.0;
  return bitmap_cost <= sort_cost ? kCsrBitmap : kCsrSort;
}
inline bool csr_use_bitmap(int num_nodes, int num_cols, long long num_edges, int forced = kCsrAuto) {
  return csr_path(num_nodes, num_cols, num_edges, forced) == kCsrBitmap;
}

template <class K>
inline int bm_set_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return kErrBadConfig;
  return kOk;
}

// ---- host side -------------------------------------------------------------------------------------------------
inline long long align16(long long x) { return (x + 15) & ~15ll; }

// workspace: [sort / mixed: keys uint32[E]] [scan scratch int[nchunks]] [queue counts (16 B)] [sort / mixed: queue int[W]]
// [bucket kernels in use: queue_big int[W], queue_done int[W], queue_clustered int[W]]
struct CsrWorkspace {
  uint32_t* keys;
  int* chunk_sums;
  int* counts;          // [0] windows above kWsKeys edges (queue); bucket kernels: of those, [1] the windows above kSortLdsKeys
                        // edges (queue_big), [2] the ones they ranked (queue_done), [3] clustered columns (queue_clustered)
  int* queue;
  int* queue_big;
  int* queue_done;
  int* queue_clustered;
  int* group_ptr;       // mixed path: int[(E / kSortLdsKeys + 1)][kBmMaxGroups + 1], range groups of the listed windows
  long long bytes;
};
// the bucket kernels flag keys in bit 31: column ids below 2^27 (num_cols <= 0: unknown universe -> not used)
inline bool csr_use_buckets(int num_cols) { return num_cols > 0 && (unsigned)num_cols <= kBkMaxCols; }

inline CsrWorkspace csr_workspace(void* base, int num_nodes, int num_cols, long long num_edges, int forced_path = kCsrAuto) {
  const long long W = ((long long)num_nodes + kBlkH - 1) / kBlkH;
  const long long nchunks = (W + kScanChunk - 1) / kScanChunk + 1;
  const CsrPath path = csr_path(num_nodes, num_cols, num_edges, forced_path);
  const bool buckets = path != kCsrBitmap && csr_use_buckets(num_cols);
  char* p = static_cast<char*>(base);
  CsrWorkspace ws;
  ws.keys = reinterpret_cast<uint32_t*>(p);
  p += path == kCsrBitmap ? 0 : align16(num_edges * 4);
  ws.chunk_sums = reinterpret_cast<int*>(p);
  p += align16(nchunks * 4);
  ws.counts = reinterpret_cast<int*>(p);
  p += 16;
  ws.queue = reinterpret_cast<int*>(p);
  p += path == kCsrBitmap ? 0 : align16(W * 4);
  ws.queue_big = reinterpret_cast<int*>(p);
  p += buckets ? align16(W * 4) : 0;
  ws.queue_done = reinterpret_cast<int*>(p);
  p += buckets ? align16(W * 4) : 0;
  ws.queue_clustered = reinterpret_cast<int*>(p);
  p += buckets ? align16(W * 4) : 0;
  ws.group_ptr = reinterpret_cast<int*>(p);   // every listed window has more than kSortLdsKeys edges
  p += path == kCsrMixed ? align16((num_edges / kSortLdsKeys + 1) * (kBmMaxGroups + 1) * 4) : 0;
  ws.bytes = p - static_cast<char*>(base);
  return ws;
}
inline long long csr_preprocess_workspace_bytes(int num_nodes, int num_cols, long long num_edges, int forced_path = kCsrAuto) {
  return csr_workspace(nullptr, num_nodes, num_cols, num_edges, forced_path).bytes;
}

inline int csr_check(int num_nodes, long long num_edges) {
  if (num_nodes < 0 || num_edges < 0) return kErrBadShape;
  if (num_nodes > (1 << 28)) return kErrBadShape;  // packed sort key keeps the column in 28 bits
  const long long W = ((long long)num_nodes + kBlkH - 1) / kBlkH;
  if (num_edges + W > 0x7FFFFFFFll) return kErrOverflow;  // T <= E + W must fit the int32 handle
  return kOk;
}

// status[0] <- number of edges whose column id lies outside [0, num_cols) (num_cols <= 0: outside [0, 2^28)); the
// handle is only meaningful when it is 0 (the bitmap path skips such edges, the sort path truncates them).
inline int csr_window_count(const int* indptr, const int* indices, int num_nodes, int num_cols, long long num_edges,
                            void* workspace, int* block_partition, int* pointer1, int* status, hipStream_t stream,
                            int forced_path = kCsrAuto) {
  if (int rc = csr_check(num_nodes, num_edges)) return rc;
  if (((uintptr_t)workspace & 15) || status == nullptr || num_cols > (1 << 28)) return kErrBadShape;
  const int W = (num_nodes + kBlkH - 1) / kBlkH;
  if (hipMemsetAsync(status, 0, sizeof(int), stream) != hipSuccess) return kErrLaunch;
  if (W == 0) {
    return hipMemsetAsync(pointer1, 0, sizeof(int), stream) == hipSuccess ? kOk : kErrLaunch;
  }
  const CsrPath path = csr_path(num_nodes, num_cols, num_edges, forced_path);
  const CsrWorkspace ws = csr_workspace(workspace, num_nodes, num_cols, num_edges, forced_path);
  uint32_t* const keys = ws.keys;
  int* const chunk_sums = ws.chunk_sums;
  const int nchunks = (W + kScanChunk - 1) / kScanChunk;
  // Windows up to kWsKeys edges: one wave each (sort in registers).  The rest is queued: bucket ranking for the windows
  // up to kSortLdsKeys edges when the ids fit its key format (clustered ones: workgroup sort in LDS); the bigger windows
  // -- or everything queued, without the bucket kernels -- go to the bitmap kernels (mixed path) or to the workgroup sort
  // (sort path).
  const bool buckets = path != kCsrBitmap && csr_use_buckets(num_cols);
  const int* big_count = buckets ? ws.counts + 1 : ws.counts;
  const int* big_queue = buckets ? ws.queue_big : ws.queue;
  if (path != kCsrBitmap) {
    const unsigned col_limit = num_cols > 0 ? (unsigned)num_cols : (1u << 28);
    const int wgs = (W + kWsWaves - 1) / kWsWaves;  // small windows: one wave each
    if (hipMemsetAsync(ws.counts, 0, 4 * sizeof(int), stream) != hipSuccess) return kErrLaunch;
    hipLaunchKernelGGL(csr_wave_sort_kernel, dim3(wgs < kWsGrid ? wgs : kWsGrid), dim3(kWsWaves * kWave), 0, stream,
                       indptr, indices, num_nodes, W, col_limit, keys, block_partition, status, ws.counts, ws.queue);
    const int grid = W < 256 * 8 ? W : 256 * 8;  // queued windows: one workgroup each
    if (buckets) {
      hipLaunchKernelGGL(csr_bucket_count_kernel, dim3(grid), dim3(kBkThreads), 0, stream, indptr, indices, num_nodes,
                         col_limit, keys, block_partition, status, ws.counts, ws.queue, ws.queue_big, ws.queue_done,
                         ws.queue_clustered);
      hipLaunchKernelGGL(csr_window_sort_kernel, dim3(grid), dim3(kSortThreads), 0, stream, indptr, indices, num_nodes,
                         W, col_limit, keys, block_partition, status, ws.counts + 3, ws.queue_clustered);
    }
    if (path == kCsrSort)
      hipLaunchKernelGGL(csr_window_sort_kernel, dim3(grid), dim3(kSortThreads), 0, stream, indptr, indices, num_nodes,
                         W, col_limit, keys, block_partition, status, big_count, big_queue);
  }
  if (path != kCsrSort) {  // bitmap: every window; mixed: the listed windows
    const size_t lds = bm_count_lds(num_cols);
    const int grid = W < kBmGrid ? W : kBmGrid;
    if (path == kCsrMixed) {
      if (int rc = bm_set_lds(csr_bitmap_count_kernel<kBmThreadsListed>, lds)) return rc;
      hipLaunchKernelGGL(csr_bitmap_count_kernel<kBmThreadsListed>, dim3(grid), dim3(kBmThreadsListed), lds, stream, indptr,
                         indices, num_nodes, num_cols, W, block_partition, status, big_queue, big_count, keys,
                         buckets ? ws.group_ptr : nullptr);
    } else {
      if (int rc = bm_set_lds(csr_bitmap_count_kernel<kBmThreadsAll>, lds)) return rc;
      hipLaunchKernelGGL(csr_bitmap_count_kernel<kBmThreadsAll>, dim3(grid), dim3(kBmThreadsAll), lds, stream, indptr,
                         indices, num_nodes, num_cols, W, block_partition, status, nullptr, nullptr, nullptr, nullptr);
    }
  }
  hipLaunchKernelGGL(scan_chunk_sums_kernel, dim3(nchunks), dim3(256), 0, stream, block_partition, W, chunk_sums);
  hipLaunchKernelGGL(scan_chunk_offsets_kernel, dim3(1), dim3(256), 0, stream, chunk_sums, nchunks);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nchunks), dim3(256), 0, stream, block_partition, W, chunk_sums, pointer1);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

inline int csr_fill(const int* indptr, const int* indices, int num_nodes, int num_cols, long long num_edges,
                    void* workspace, const int* pointer1, uint32_t* hspa_packed, int* hind, hipStream_t stream,
                    int forced_path = kCsrAuto) {
  if (int rc = csr_check(num_nodes, num_edges)) return rc;
  if (((uintptr_t)workspace & 15) || ((uintptr_t)hspa_packed & 15) || ((uintptr_t)hind & 15)) return kErrBadShape;
  const int W = (num_nodes + kBlkH - 1) / kBlkH;
  if (W == 0) return kOk;
  const CsrPath path = csr_path(num_nodes, num_cols, num_edges, forced_path);
  const CsrWorkspace ws = csr_workspace(workspace, num_nodes, num_cols, num_edges, forced_path);
  const bool buckets = path != kCsrBitmap && csr_use_buckets(num_cols);
  const int* big_count = buckets ? ws.counts + 1 : ws.counts;
  const int* big_queue = buckets ? ws.queue_big : ws.queue;
  if (path != kCsrSort) {  // bitmap: every window; mixed: the listed windows (the big ones go first: longest jobs)
    const size_t lds = bm_fill_lds(num_cols);
    const int grid = W < kBmGrid ? W : kBmGrid;
    if (path == kCsrMixed) {
      if (int rc = bm_set_lds(csr_bitmap_fill_kernel<kBmThreadsListed>, lds)) return rc;
      hipLaunchKernelGGL(csr_bitmap_fill_kernel<kBmThreadsListed>, dim3(grid), dim3(kBmThreadsListed), lds, stream, indptr,
                         indices, num_nodes, num_cols, W, pointer1, hspa_packed, hind, big_queue, big_count, ws.keys,
                         buckets ? ws.group_ptr : nullptr);
    } else {
      if (int rc = bm_set_lds(csr_bitmap_fill_kernel<kBmThreadsAll>, lds)) return rc;
      hipLaunchKernelGGL(csr_bitmap_fill_kernel<kBmThreadsAll>, dim3(grid), dim3(kBmThreadsAll), lds, stream, indptr,
                         indices, num_nodes, num_cols, W, pointer1, hspa_packed, hind, nullptr, nullptr, nullptr, nullptr);
    }
  }
  if (path != kCsrBitmap) {
    const uint32_t* const keys = ws.keys;
    const int wgs = (W + kWsWaves - 1) / kWsWaves;
    hipLaunchKernelGGL(csr_wave_fill_kernel, dim3(wgs < kWsGrid ? wgs : kWsGrid), dim3(kWsWaves * kWave), 0, stream,
                       indptr, num_nodes, W, keys, pointer1, hspa_packed, hind);
    const int grid = W < 256 * 8 ? W : 256 * 8;
    if (buckets) {
      if (int rc = bm_set_lds(csr_bucket_fill_kernel, sizeof(BkFillLds))) return rc;
      hipLaunchKernelGGL(csr_bucket_fill_kernel, dim3(grid), dim3(kBkThreads), sizeof(BkFillLds), stream, indptr, num_nodes,
                         keys, pointer1, hspa_packed, hind, ws.counts, ws.queue_done);
      hipLaunchKernelGGL(csr_window_fill_kernel, dim3(grid), dim3(kSortThreads), 0, stream, indptr, num_nodes, W, keys,
                         pointer1, hspa_packed, hind, ws.counts + 3, ws.queue_clustered);
    }
    if (path == kCsrSort)
      hipLaunchKernelGGL(csr_window_fill_kernel, dim3(grid), dim3(kSortThreads), 0, stream, indptr, num_nodes, W, keys,
                         pointer1, hspa_packed, hind, big_count, big_queue);
  }
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace voltrix
