// Microbenchmark: what does the SHAPE of a 16 x 128 fp32 tile store cost?  One wave writes tiles of 16 consecutive rows x 512 B
// (the stream kernel's epilogue) with 8 x global_store_dwordx4, in four lane -> address mappings:
//   0: 16 rows x  64 B per instruction (the swapped-operand MFMA layout as it stands)
//   1:  8 rows x 128 B per instruction (whole cache lines)
//   2:  4 rows x 256 B
//   3:  2 rows x 512 B (whole rows)
//   4: as 0 with global_store_dword x 4 per slot (the window kernel's epilogue: 4 rows x 64 B per instruction, 32 instructions)
#include <hip/hip_runtime.h>
typedef float float4_t __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(64) void store_tiles(float* out, int num_tiles, int F) {
  const int lane = threadIdx.x;
  float4_t v = {(float)lane, 1.f, 2.f, 3.f};
  for (int t = blockIdx.x; t < num_tiles; t += gridDim.x) {
    float* tile = out + (long long)t * 16 * F;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int row, col;
      if (MODE == 0 || MODE == 4) { row = lane & 15; col = 16 * j + 4 * (lane >> 4); }
      else if (MODE == 1) { row = (lane >> 3) + 8 * (j & 1); col = 32 * (j >> 1) + 4 * (lane & 7); }
      else if (MODE == 2) { row = (lane >> 4) + 4 * (j & 3); col = 64 * (j >> 2) + 4 * (lane & 15); }
      else { row = (lane >> 5) + 2 * j; col = 4 * (lane & 31); }
      if (MODE == 4) {
        // window-kernel shape: lane (g, c): rows 4 g + jj, column 16 j + c
        const int g = lane >> 4, c = lane & 15;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) tile[(long long)(4 * g + jj) * F + 16 * j + c] = v[jj];
      } else {
        *reinterpret_cast<float4_t*>(tile + (long long)row * F + col) = v;
      }
    }
  }
}
extern "C" int launch_store(int mode, float* out, int num_tiles, int F, int grid, hipStream_t s) {
  switch (mode) {
    case 0: hipLaunchKernelGGL(store_tiles<0>, dim3(grid), dim3(64), 0, s, out, num_tiles, F); break;
    case 1: hipLaunchKernelGGL(store_tiles<1>, dim3(grid), dim3(64), 0, s, out, num_tiles, F); break;
    case 2: hipLaunchKernelGGL(store_tiles<2>, dim3(grid), dim3(64), 0, s, out, num_tiles, F); break;
    case 3: hipLaunchKernelGGL(store_tiles<3>, dim3(grid), dim3(64), 0, s, out, num_tiles, F); break;
    default: hipLaunchKernelGGL(store_tiles<4>, dim3(grid), dim3(64), 0, s, out, num_tiles, F); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
