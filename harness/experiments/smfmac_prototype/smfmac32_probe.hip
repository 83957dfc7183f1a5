// Probe of v_smfmac_f32_16x16x32_f16 (K = 32: A = 4 kept halves + 8-bit index per lane, B = 8 halves per lane).
#include <hip/hip_runtime.h>
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));

template <int ABID>
__global__ void probe(const _Float16* a /*[64][4]*/, const _Float16* b /*[64][8]*/, const int* idx, float* d) {
  const int lane = threadIdx.x;
  half4_t av;
  half8_t bv;
  for (int i = 0; i < 4; ++i) av[i] = a[lane * 4 + i];
  for (int i = 0; i < 8; ++i) bv[i] = b[lane * 8 + i];
  float4_t acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_smfmac_f32_16x16x32_f16(av, bv, acc, idx[lane], 0, ABID);
  for (int i = 0; i < 4; ++i) d[lane * 4 + i] = acc[i];
}

extern "C" int smfmac32_probe(void* a, void* b, void* idx, void* d, int abid) {
#define CASE(A) if (abid == A) hipLaunchKernelGGL(probe<A>, dim3(1), dim3(64), 0, 0, (const _Float16*)a, (const _Float16*)b, (const int*)idx, (float*)d);
  CASE(0) CASE(1) CASE(2) CASE(3)
  return (int)hipDeviceSynchronize();
}
