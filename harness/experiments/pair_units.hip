// Experiment (round 2): the window kernel with TWO units per wave (spmm_tc16_kernel<T, 2>) on tile (128, 3, 4).
#include "voltrix/spmm_kernels.hpp"
extern "C" int pair_units_launch(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int f, void* input,
                                 void* output, int atomic_out, void* units, void* unit_ptr, int max_units_per_xcd,
                                 void* partials, int units_per_wave, void* stream) {
  return voltrix::launch_spmm_tc16<voltrix::SpmmTile<128, 3, 4, 2, false, false>>(
      static_cast<const int*>(blk_offsets), static_cast<const uint32_t*>(hspa_packed), static_cast<const int*>(hind),
      num_nodes, f, input, static_cast<float*>(output), static_cast<hipStream_t>(stream), nullptr, nullptr, atomic_out,
      static_cast<const int*>(units), static_cast<const int*>(unit_ptr), max_units_per_xcd, static_cast<float*>(partials),
      nullptr, nullptr, units_per_wave);
}
