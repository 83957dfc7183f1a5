// Experiment (round 2): the two-level RESIDUAL as a column-ordered edge stream with LDS accumulators -- no TC blocks, no
// MFMA.  One wave owns R consecutive rows (accumulators float[R][128] in LDS, touched by that wave only: the summation
// order per row is the column order, fixed) and walks its edges sorted by column; every edge is one 256-byte load of B's
// row (one dword = two halfs per lane) and two ds_add_f32 per lane.  All waves of the chip sweep the column range from 0
// upwards, so rows of B fetched by one CU are L2 hits for the others of its XCD as long as they keep pace.
// F = 128 fp16 only.  Launcher: grid = ceil(groups / WAVES) workgroups of WAVES waves; stream word = col << 6 | local row.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdint>

#ifndef ES_ROWS
#define ES_ROWS 64   // rows per wave
#endif
#ifndef ES_WAVES
#define ES_WAVES 4
#endif
#ifndef ES_BATCH
#define ES_BATCH 16  // row loads in flight per wave
#endif

static __global__ __launch_bounds__(ES_WAVES * 64) void edge_stream_kernel(const int* __restrict__ stream_ptr,
                                                                          const int* __restrict__ stream,
                                                                          const int* __restrict__ group_order,
                                                                          const int num_groups, const int num_nodes,
                                                                          const uint32_t* __restrict__ b_words /* [N][64] */,
                                                                          float* __restrict__ c, const int atomic_out) {
  extern __shared__ __attribute__((aligned(16))) float es_acc[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* const acc = es_acc + wave * (ES_ROWS * 128);
  const int pos = blockIdx.x * ES_WAVES + wave;
  if (pos >= num_groups) return;
  const int g = group_order ? group_order[pos] : pos;
  for (int i = lane; i < ES_ROWS * 128 / 4; i += 64) reinterpret_cast<float4*>(acc)[i] = float4{0.f, 0.f, 0.f, 0.f};
  const int e0 = stream_ptr[g], e1 = stream_ptr[g + 1];
  // software pipeline: the row loads of batch k + 1 are issued before the adds of batch k (two register buffers); the
  // stream words run one batch further ahead (lane u holds word u of its batch)
  auto fetch_words = [&](int base) { return (lane < ES_BATCH && base + lane < e1) ? stream[base + lane] : -1; };
  auto issue = [&](int words, uint32_t (&bv)[ES_BATCH]) {
#pragma unroll
    for (int u = 0; u < ES_BATCH; ++u) {
      const int w = __builtin_amdgcn_readlane(words, u);
      bv[u] = b_words[(long long)(w >= 0 ? (w >> 6) : 0) * 64 + lane];   // unconditional: straight-line code keeps the waits counted
    }
  };
  auto consume = [&](int words, const uint32_t (&bv)[ES_BATCH]) {
#pragma unroll
    for (int u = 0; u < ES_BATCH; ++u) {
      const int w = __builtin_amdgcn_readlane(words, u);
      if (w >= 0) {  // wave-uniform
        const __half2 h = __builtin_bit_cast(__half2, bv[u]);
        float* const dst = acc + (w & 63) * 128 + 2 * lane;
        __hip_atomic_fetch_add(dst, __low2float(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(dst + 1, __high2float(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  };
  uint32_t bv0[ES_BATCH], bv1[ES_BATCH];
  int w0 = fetch_words(e0), w1 = fetch_words(e0 + ES_BATCH);
  issue(w0, bv0);
  for (int base = e0; base < e1; base += 2 * ES_BATCH) {
    const int w2 = fetch_words(base + 2 * ES_BATCH);
    issue(w1, bv1);
    consume(w0, bv0);
    const int w3 = fetch_words(base + 3 * ES_BATCH);
    issue(w2, bv0);
    consume(w1, bv1);
    w0 = w2;
    w1 = w3;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int row0 = g * ES_ROWS;
  for (int r = 0; r < ES_ROWS; ++r) {
    const int row = row0 + r;
    if (row >= num_nodes) break;
    const float2 v = reinterpret_cast<const float2*>(acc + r * 128)[lane];
    float* const dst = c + (long long)row * 128 + 2 * lane;
    if (atomic_out) {
      unsafeAtomicAdd(dst, v.x);
      unsafeAtomicAdd(dst + 1, v.y);
    } else {
      *reinterpret_cast<float2*>(dst) = v;
    }
  }
}

extern "C" int edge_stream_launch(void* stream_ptr, void* stream, void* group_order, int num_groups, int num_nodes, void* b,
                                  void* c, int atomic_out, void* hip_stream) {
  const int lds = ES_WAVES * ES_ROWS * 128 * 4;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(edge_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
        hipSuccess)
      return 3;
    attr = true;
  }
  hipLaunchKernelGGL(edge_stream_kernel, dim3((num_groups + ES_WAVES - 1) / ES_WAVES), dim3(ES_WAVES * 64), lds,
                     static_cast<hipStream_t>(hip_stream), static_cast<const int*>(stream_ptr), static_cast<const int*>(stream),
                     static_cast<const int*>(group_order), num_groups, num_nodes, static_cast<const uint32_t*>(b),
                     static_cast<float*>(c), atomic_out);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
