// Diagnostic build of the shipped two-level panel tile (harness/experiments/exp_corun_diag.py compiles it with
// -DVOLTRIX_PANEL_DIAG=n: bit 0 skips the MFMAs, bit 1 the row DMAs, bit 2 the barrier -- results are wrong by design).
#include "voltrix/spmm_panel_kernels.hpp"
extern "C" int corun_diag_launch(void* panel_ptr, void* panel_cols, void* panel_bits, void* panel_order, int num_nodes, int f,
                                 void* input, void* output, int accumulate, void* stream) {
  return voltrix::launch_spmm_panel<voltrix::PanelTile<128, 3, 8, 4, 1>>(
      static_cast<const int*>(panel_ptr), static_cast<const int*>(panel_cols), static_cast<const uint32_t*>(panel_bits),
      static_cast<const int*>(panel_order), num_nodes, f, input, static_cast<float*>(output), accumulate, nullptr,
      static_cast<hipStream_t>(stream));
}
