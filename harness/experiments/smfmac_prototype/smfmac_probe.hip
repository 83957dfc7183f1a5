// Probe of v_smfmac_f32_16x16x64_f16 operand layouts (no ISA text offline): the host sets raw register contents.
#include <hip/hip_runtime.h>
#include <cstdint>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half16_t __attribute__((ext_vector_type(16)));
typedef float float4_t __attribute__((ext_vector_type(4)));

template <int ABID>
__global__ void probe(const _Float16* a /*[64][8]*/, const _Float16* b /*[64][16]*/, const int* idx /*[64]*/, float* d /*[64][4]*/) {
  const int lane = threadIdx.x;
  half8_t av;
  half16_t bv;
  for (int i = 0; i < 8; ++i) av[i] = a[lane * 8 + i];
  for (int i = 0; i < 16; ++i) bv[i] = b[lane * 16 + i];
  float4_t acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_smfmac_f32_16x16x64_f16(av, bv, acc, idx[lane], 0, ABID);
  for (int i = 0; i < 4; ++i) d[lane * 4 + i] = acc[i];
}

extern "C" int smfmac_probe(void* a, void* b, void* idx, void* d, int abid) {
  if (abid == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, (const _Float16*)a, (const _Float16*)b, (const int*)idx, (float*)d);
  else hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, (const _Float16*)a, (const _Float16*)b, (const int*)idx, (float*)d);
  return (int)hipDeviceSynchronize();
}
