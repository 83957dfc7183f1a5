// Four back-to-back v_smfmac with the same A / index and different B, A written by VALU right before (as in the panel
// kernel), NOPS wait states between that VALU write and the first smfmac and GAP between consecutive smfmacs.
#include <hip/hip_runtime.h>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half16_t __attribute__((ext_vector_type(16)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

template <int NOPS, int GAP>
__global__ void hazard(const unsigned* araw, const _Float16* b, const int* idx, float* d) {
  const int lane = threadIdx.x;
  uint4_t ar;
  for (int i = 0; i < 4; ++i) ar[i] = araw[lane * 4 + i];
  half16_t b0, b1, b2, b3;
  for (int i = 0; i < 16; ++i) {
    b0[i] = b[lane * 16 + i]; b1[i] = b[1024 + lane * 16 + i]; b2[i] = b[2048 + lane * 16 + i]; b3[i] = b[3072 + lane * 16 + i];
  }
  float4_t c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
  int ix = idx[lane];
  asm volatile(
      "s_nop 7\n s_nop 7\n"
      ".rept %10\n s_nop 0\n .endr\n"
      "v_smfmac_f32_16x16x64_f16 %0, %4, %5, %9\n .rept %11\n s_nop 0\n .endr\n"
      "v_smfmac_f32_16x16x64_f16 %1, %4, %6, %9\n .rept %11\n s_nop 0\n .endr\n"
      "v_smfmac_f32_16x16x64_f16 %2, %4, %7, %9\n .rept %11\n s_nop 0\n .endr\n"
      "v_smfmac_f32_16x16x64_f16 %3, %4, %8, %9\n s_nop 7\n s_nop 7\n s_nop 7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
      : "v"(ar), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(ix), "n"(NOPS), "n"(GAP));
  for (int i = 0; i < 4; ++i) {
    d[lane * 4 + i] = c0[i]; d[256 + lane * 4 + i] = c1[i]; d[512 + lane * 4 + i] = c2[i]; d[768 + lane * 4 + i] = c3[i];
  }
}

extern "C" int smfmac_hazard2(void* a, void* b, void* idx, void* d, int nops, int gap) {
#define CASE(N, G) if (nops == N && gap == G) hipLaunchKernelGGL((hazard<N, G>), dim3(1), dim3(64), 0, 0, (const unsigned*)a, (const _Float16*)b, (const int*)idx, (float*)d);
  CASE(0, 0) CASE(1, 0) CASE(2, 0) CASE(4, 0) CASE(8, 0) CASE(0, 1) CASE(0, 2) CASE(0, 4) CASE(0, 8) CASE(4, 4) CASE(8, 8)
  return (int)hipDeviceSynchronize();
}
