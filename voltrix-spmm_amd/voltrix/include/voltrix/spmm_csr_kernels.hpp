// Voltrix-SpMM for MI355X (gfx950) -- CSR row-gather kernel: C = A * B straight from the CSR, no block format, no matrix cores
// (round 6).
//
// Why it exists.  At mean degree 2-12 the 16 rows of a window share no column: the block format pads almost nothing there, but it
// buys nothing either -- every edge is one gathered row of B, and what a step costs is the number of 128-byte line requests the CUs
// can issue.  For 16-bit features the window / stream kernels (LDS-DMA gathers, MFMA on bitmaps) win that race at F <= 256; for
// fp32 FEATURES they lose it: the exact-fp32 tiles feed 512-byte rows through v_mfma_f32_16x16x4_f32 and a ring sized for
// them, the other route is a two-pass cast in front of the 16-bit kernels.  Measured on the reference's evaluation stand-ins
// (profiles/r06/eval_set.jsonl, the harness's plain kernel of the same shape): DD-like fp32-in F = 128 0.130 ms against 0.162 for
// the exact tiles and 0.172 for rocSPARSE's best algorithm; F = 512 0.449 / 0.650 / 0.661; com-amazon-like 0.145 / 0.180 / 0.191.
// voltrix.spmm times it ONCE per (handle, width, dtype) against the block-format path and keeps the faster (spmm/spmm.py).
//
// Shape.  A group of L = min(64, next_pow2(slab_columns / V)) lanes owns one row (V = 16 bytes of the operand: 4 fp32, 8 fp16 / bf16);
// a 256-thread workgroup owns 256 / L consecutive rows; grid.y walks column slabs of 64 V columns (fp32: 256, 16-bit: 512).  Per edge a
// lane loads its 16 bytes of row B[col] (batches of UNROLL = 4 edges in flight, the last batch masked), accumulates in fp32 in CSR order, and stores 16-byte pieces of C.
// Workgroup -> rows: XCD x = blockIdx.x % 8 owns a contiguous eighth of the row groups (xcd_ranges = 1: neighbouring rows, which
// reference neighbouring rows of B on band / community graphs, share an L2; equal or better than the round-robin order on every graph
// measured: amazon0505-like x 128 fp32 0.297 -> 0.256 ms, ppi-like x 512 0.166 -> 0.111), or workgroup b = row group b (0).  A binary: products are exact; the sum is fp32 in edge order:
// |C - ref| <= deg * 2^-23 * (A |B|).
//
// Bound: the CUs' line-request rate / HBM, like the window kernel on these graphs.  Rows of very different length
// in one wave serialise (a hub row of a web graph: 6 ms against 0.29): the operator only takes this kernel where it measured faster.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "voltrix/spmm_kernels.hpp"

namespace voltrix {

template <typename T>
struct CsrArgs {
  const int* indptr;    // [num_rows + 1]
  const int* indices;   // [nnz] column ids = rows of `input`
  const T* input;       // [*, F] row-major, rows 16-byte aligned
  float* output;        // [num_rows, F]
  const float* values;  // WEIGHTED kernels: [nnz] fp32 edge values in CSR order (duplicate entries add); else unused
  int num_rows;
  int F;
  int lanes_per_row;    // power of two <= 64
  int groups_per_xcd;   // ceil(row groups / 8): sizes the grid; a row group = 256 / lanes_per_row rows
  int xcd_ranges;       // 1: XCD x owns a contiguous eighth of the row groups; 0: consecutive groups go round the XCDs
};

template <typename T>
__device__ __forceinline__ void csr_accumulate(float (&acc)[16 / sizeof(T)], const uint4_t raw) {
  if constexpr (std::is_same<T, float>::value) {
    const float4_t v = __builtin_bit_cast(float4_t, raw);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += v[i];
  } else if constexpr (std::is_same<T, _Float16>::value) {
    const half8_t v = __builtin_bit_cast(half8_t, raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += (float)v[i];
  } else {   // bfloat16 as bits: a 16-bit shift is the conversion
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[2 * i] += __builtin_bit_cast(float, raw[i] << 16);
      acc[2 * i + 1] += __builtin_bit_cast(float, raw[i] & 0xffff0000u);
    }
  }
}

// acc += v * row piece (WEIGHTED kernels: one fused multiply-add per element, a single fp32 rounding)
template <typename T>
__device__ __forceinline__ void csr_accumulate_scaled(float (&acc)[16 / sizeof(T)], const uint4_t raw, const float v) {
  if constexpr (std::is_same<T, float>::value) {
    const float4_t x = __builtin_bit_cast(float4_t, raw);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_fmaf(v, x[i], acc[i]);
  } else if constexpr (std::is_same<T, _Float16>::value) {
    const half8_t x = __builtin_bit_cast(half8_t, raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(v, (float)x[i], acc[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[2 * i] = __builtin_fmaf(v, __builtin_bit_cast(float, raw[i] << 16), acc[2 * i]);
      acc[2 * i + 1] = __builtin_fmaf(v, __builtin_bit_cast(float, raw[i] & 0xffff0000u), acc[2 * i + 1]);
    }
  }
}

template <typename T, int UNROLL, bool WEIGHTED = false>
static __global__ __launch_bounds__(256) void spmm_csr_rows_kernel(const CsrArgs<T> a) {
  constexpr int V = 16 / (int)sizeof(T);
  const int L = a.lanes_per_row;
  const int rows_per_group = 256 / L;
  // xcd_ranges: XCD x owns the row groups [x * groups_per_xcd, (x + 1) * groups_per_xcd); else workgroup b = row group b
  const long long group = a.xcd_ranges ? (long long)(blockIdx.x % kNumXcd) * a.groups_per_xcd + blockIdx.x / kNumXcd
                                       : (long long)blockIdx.x;
  const long long row = group * rows_per_group + (int)threadIdx.x / L;
  if (row >= a.num_rows) return;
  const int lane = (int)threadIdx.x & (L - 1);
  const long long col0 = ((long long)blockIdx.y * 64 + lane) * V;     // this lane's 16 bytes of every gathered row
  if (col0 >= a.F) return;
  float acc[V];
#pragma unroll
  for (int i = 0; i < V; ++i) acc[i] = 0.0f;
  int e = a.indptr[row];
  const int end = a.indptr[row + 1];
  const T* const base = a.input + col0;
  const long long F = a.F;
  // full batches of UNROLL edges, then ONE more batch for the tail with clamped ids (every load of it is issued, unconditionally, before
  // the first is consumed; the slots past the row's end are skipped when adding): at mean degree 5 a row is one or two batches with all
  // their loads in flight -- a scalar tail loop ran most rows of these graphs one load at a time; a tail whose LOADS were predicated
  // serialised them (ppi-like x 128: 0.150 ms against 0.091; profiles/r06/experiment_csr_mapping.log)
  for (; e + UNROLL <= end; e += UNROLL) {
    uint4_t raw[UNROLL];
    float v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      raw[u] = *reinterpret_cast<const uint4_t*>(base + (long long)a.indices[e + u] * F);
      if constexpr (WEIGHTED) v[u] = a.values[e + u];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if constexpr (WEIGHTED) csr_accumulate_scaled<T>(acc, raw[u], v[u]);
      else csr_accumulate<T>(acc, raw[u]);
    }
  }
  if (e < end) {
    uint4_t raw[UNROLL];
    float v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int ee = e + u < end ? e + u : end - 1;
      raw[u] = *reinterpret_cast<const uint4_t*>(base + (long long)a.indices[ee] * F);
      if constexpr (WEIGHTED) v[u] = a.values[ee];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (e + u < end) {
        if constexpr (WEIGHTED) csr_accumulate_scaled<T>(acc, raw[u], v[u]);
        else csr_accumulate<T>(acc, raw[u]);
      }
  }
  float4_t* out = reinterpret_cast<float4_t*>(a.output + row * F + col0);
#pragma unroll
  for (int i = 0; i < V / 4; ++i) out[i] = float4_t{acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]};
}

// dtype: 0 fp32, 1 fp16, 2 bfloat16.  embedding_dim % (16 / sizeof(T)) == 0 (16-byte row pieces).  Every row of `output` is written.
// values (optional): fp32 [nnz] edge values in CSR order -- C = csr(values) * B, the products v * b in fp32 (one fused multiply-add per
// element: exact for fp32 rows up to the sum's rounding), duplicates add.
inline int launch_spmm_csr_rows(const int* indptr, const int* indices, int num_rows, int embedding_dim, const void* input, int dtype,
                                float* output, hipStream_t stream, int xcd_ranges = 0, const float* values = nullptr) {
  if (num_rows < 0 || embedding_dim < 0 || dtype < 0 || dtype > 2) return kErrBadShape;
  if (num_rows == 0 || embedding_dim == 0) return kOk;
  const int v = dtype == 0 ? 4 : 8;
  if (embedding_dim % v || indptr == nullptr || input == nullptr || output == nullptr || ((uintptr_t)input & 15) ||
      ((uintptr_t)output & 15))
    return kErrBadShape;
  const int pieces = embedding_dim / v;                  // 16-byte pieces per row
  const int slab_pieces = pieces < 64 ? pieces : 64;
  int lanes = 1;
  while (lanes < slab_pieces) lanes <<= 1;
  const int slabs = (pieces + 63) / 64;
  const int rows_per_group = 256 / lanes;
  const long long groups = ((long long)num_rows + rows_per_group - 1) / rows_per_group;
  const long long per_xcd = (groups + kNumXcd - 1) / kNumXcd;
  if (per_xcd * kNumXcd > 0x7fffffffLL) return kErrBadShape;
  const dim3 grid((unsigned)(per_xcd * kNumXcd), (unsigned)slabs);
  auto go = [&](auto tag) {
    using T = decltype(tag);
    CsrArgs<T> a{indptr, indices, static_cast<const T*>(input), output, values, num_rows, embedding_dim, lanes, (int)per_xcd,
                 xcd_ranges ? 1 : 0};
    if (values != nullptr)
      hipLaunchKernelGGL((spmm_csr_rows_kernel<T, 4, true>), grid, dim3(256), 0, stream, a);
    else
      hipLaunchKernelGGL((spmm_csr_rows_kernel<T, 4, false>), grid, dim3(256), 0, stream, a);
  };
  if (dtype == 0) go(float{});
  else if (dtype == 1) go(_Float16{});
  else go(bfloat16_bits{});
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- value planes: new edge values on a fixed pattern (voltrix/weighted.py::update_values) ------------------------------------------
// plane[slot[e]] = T(values[e]): `slot` = the element of the flat value plane [T * 128] every CSR entry lands on (weighted.edge_slots;
// duplicate-free patterns: every entry owns its element).  One pass: 12 bytes read + one 2- or 4-byte store per edge.
template <typename T>
static __global__ __launch_bounds__(256) void scatter_values_kernel(const float* __restrict__ values, const long long* __restrict__ slot,
                                                                    T* __restrict__ plane, const long long count) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long long)gridDim.x * 256) {
    const float v = values[e];
    if constexpr (std::is_same<T, float>::value) plane[slot[e]] = v;
    else if constexpr (std::is_same<T, _Float16>::value) plane[slot[e]] = (_Float16)v;
    else {   // bfloat16 bits: round to nearest even, NaN kept quiet
      const unsigned b = __builtin_bit_cast(unsigned, v);
      plane[slot[e]] = (v != v) ? (bfloat16_bits)0x7fc0 : (bfloat16_bits)((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
    }
  }
}

// dtype of the plane: 0 fp32, 1 fp16, 2 bfloat16
inline int scatter_values(const float* values, const long long* slot, void* plane, long long count, int dtype, hipStream_t stream) {
  if (count < 0 || dtype < 0 || dtype > 2) return kErrBadShape;
  if (count == 0) return kOk;
  if (values == nullptr || slot == nullptr || plane == nullptr) return kErrBadShape;
  const long long want = (count + 255) / 256;
  const int blocks = (int)(want < 256 * 32 ? want : 256 * 32);
  if (dtype == 0)
    hipLaunchKernelGGL((scatter_values_kernel<float>), dim3(blocks), dim3(256), 0, stream, values, slot, static_cast<float*>(plane), count);
  else if (dtype == 1)
    hipLaunchKernelGGL((scatter_values_kernel<_Float16>), dim3(blocks), dim3(256), 0, stream, values, slot,
                       static_cast<_Float16*>(plane), count);
  else
    hipLaunchKernelGGL((scatter_values_kernel<bfloat16_bits>), dim3(blocks), dim3(256), 0, stream, values, slot,
                       static_cast<bfloat16_bits*>(plane), count);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace voltrix
