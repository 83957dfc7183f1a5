// Issue rate of v_smfmac_f32_16x16x64_f16 vs v_mfma_f32_16x16x32_f16 (8 independent accumulators, registers only).
#include <hip/hip_runtime.h>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half16_t __attribute__((ext_vector_type(16)));
typedef float float4_t __attribute__((ext_vector_type(4)));

template <int SPARSE>
__global__ __launch_bounds__(256) void rate(float* out, int iters, int idx) {
  half8_t a;
  half16_t b;
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)(threadIdx.x & 1 ? 2.0f : 0.0f);
  for (int i = 0; i < 16; ++i) b[i] = (_Float16)(0.001f * (threadIdx.x + i));
  half8_t b8 = {b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7]};
  float4_t acc[8];
  for (int s = 0; s < 8; ++s) acc[s] = float4_t{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (SPARSE == 2) {
        typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
        half4_t a4 = {a[0], a[1], a[2], a[3]};
        acc[s] = __builtin_amdgcn_smfmac_f32_16x16x32_f16(a4, b8, acc[s], idx, 0, 0);
      } else if (SPARSE) acc[s] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, acc[s], idx, 0, 0);
      else acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b8, acc[s], 0, 0, 0);
    }
  }
  float r = 0.f;
  for (int s = 0; s < 8; ++s) r += acc[s][0] + acc[s][1] + acc[s][2] + acc[s][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

extern "C" float rate_ms(int sparse, int iters, void* out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a, 0);
    if (sparse == 2) hipLaunchKernelGGL(rate<2>, dim3(256 * 2), dim3(256), 0, 0, (float*)out, iters, 0x4444);
    else if (sparse) hipLaunchKernelGGL(rate<1>, dim3(256 * 2), dim3(256), 0, 0, (float*)out, iters, 0x4444);
    else hipLaunchKernelGGL(rate<0>, dim3(256 * 2), dim3(256), 0, 0, (float*)out, iters, 0);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}
