// Does a v_smfmac need wait states after another v_smfmac?  Two instructions in one asm block, N s_nop's between.
#include <hip/hip_runtime.h>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half16_t __attribute__((ext_vector_type(16)));
typedef float float4_t __attribute__((ext_vector_type(4)));

template <int MODE, int NOPS>
__global__ void hazard(const _Float16* a, const _Float16* b, const int* idx, float* d) {
  const int lane = threadIdx.x;
  half8_t a0, a1;
  half16_t b0, b1;
  for (int i = 0; i < 8; ++i) { a0[i] = a[lane * 8 + i]; a1[i] = a[512 + lane * 8 + i]; }
  for (int i = 0; i < 16; ++i) { b0[i] = b[lane * 16 + i]; b1[i] = b[1024 + lane * 16 + i]; }
  float4_t c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
  int ix = idx[lane];
  if (MODE == 0) {  // same A and index, different B
    asm volatile("s_nop 7\n s_nop 7\n v_smfmac_f32_16x16x64_f16 %0, %2, %3, %5\n .rept %6\n s_nop 0\n .endr\n"
                 "v_smfmac_f32_16x16x64_f16 %1, %2, %4, %5\n s_nop 7\n s_nop 7\n s_nop 7"
                 : "+v"(c0), "+v"(c1) : "v"(a0), "v"(b0), "v"(b1), "v"(ix), "n"(NOPS));
  } else {          // same B and index register, different A, ABID 0 then 1
    asm volatile("s_nop 7\n s_nop 7\n v_smfmac_f32_16x16x64_f16 %0, %2, %4, %5\n .rept %6\n s_nop 0\n .endr\n"
                 "v_smfmac_f32_16x16x64_f16 %1, %3, %4, %5 abid:1\n s_nop 7\n s_nop 7\n s_nop 7"
                 : "+v"(c0), "+v"(c1) : "v"(a0), "v"(a1), "v"(b0), "v"(ix), "n"(NOPS));
  }
  for (int i = 0; i < 4; ++i) { d[lane * 4 + i] = c0[i]; d[256 + lane * 4 + i] = c1[i]; }
}

extern "C" int smfmac_hazard(void* a, void* b, void* idx, void* d, int mode, int nops) {
#define CASE(M, N) if (mode == M && nops == N) hipLaunchKernelGGL((hazard<M, N>), dim3(1), dim3(64), 0, 0, (const _Float16*)a, (const _Float16*)b, (const int*)idx, (float*)d);
  CASE(0, 0) CASE(0, 1) CASE(0, 2) CASE(0, 3) CASE(0, 4) CASE(0, 6) CASE(0, 8) CASE(0, 12) CASE(0, 16)
  CASE(1, 0) CASE(1, 1) CASE(1, 2) CASE(1, 3) CASE(1, 4) CASE(1, 6) CASE(1, 8) CASE(1, 12) CASE(1, 16)
  return (int)hipDeviceSynchronize();
}
