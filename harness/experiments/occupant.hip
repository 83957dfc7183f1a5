// A kernel that only OCCUPIES a CU the way the panel kernel does -- threads, registers, LDS, time -- and executes nothing:
// no memory traffic, no matrix cores, no LDS accesses (harness/experiments/exp_occupant.py).  What the window kernel loses
// beside it is what co-residency itself costs.
#include <hip/hip_runtime.h>

template <int VGPRS>
__device__ __forceinline__ void touch_registers();
template <>
__device__ __forceinline__ void touch_registers<176>() { asm volatile("v_mov_b32 v175, 0" ::: "v175"); }
template <>
__device__ __forceinline__ void touch_registers<32>() { asm volatile("v_mov_b32 v31, 0" ::: "v31"); }

template <int THREADS, int VGPRS>
__global__ __launch_bounds__(THREADS) void occupant_kernel(const long long ticks /* 100 MHz */) {
  extern __shared__ char smem[];
  touch_registers<VGPRS>();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  if (ticks < 0) smem[threadIdx.x] = 1;   // never: keeps the dynamic LDS allocation
}

extern "C" int occupant_launch(int grid, int threads, int vgprs, int lds_bytes, double microseconds, void* stream) {
  const long long ticks = (long long)(microseconds * 100.0);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CASE(T, V)                                                                                                   \
  if (threads == T && vgprs == V) {                                                                                  \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&occupant_kernel<T, V>),                                   \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)                   \
      return 2;                                                                                                      \
    hipLaunchKernelGGL((occupant_kernel<T, V>), dim3(grid), dim3(T), lds_bytes, s, ticks);                           \
    return hipGetLastError() == hipSuccess ? 0 : 1;                                                                  \
  }
  CASE(512, 176) CASE(512, 32) CASE(256, 176) CASE(256, 32)
#undef CASE
  return 3;
}
