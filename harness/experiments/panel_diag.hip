// Diagnostic build of ONE panel tile (harness/experiments/panel_diag.py compiles it with -DVOLTRIX_PANEL_DIAG=n).
#include "voltrix/spmm_panel_kernels.hpp"
#ifndef PD_DEPTH
#define PD_DEPTH 4
#endif
#ifndef PD_WAVES
#define PD_WAVES 8
#endif
#ifndef PD_RB
#define PD_RB 4
#endif
extern "C" int panel_diag_launch(void* panel_ptr, void* panel_cols, void* panel_bits, int num_nodes, int f, void* input,
                                 void* output, int accumulate, void* stream) {
  return voltrix::launch_spmm_panel<voltrix::PanelTile<128, PD_DEPTH, PD_WAVES, PD_RB, 1>>(
      static_cast<const int*>(panel_ptr), static_cast<const int*>(panel_cols), static_cast<const uint32_t*>(panel_bits),
      nullptr, num_nodes, f, input, static_cast<float*>(output), accumulate, nullptr, static_cast<hipStream_t>(stream));
}
