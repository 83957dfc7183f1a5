// libvoltrix_hip.so -- panel kernel entry points (include/voltrix_capi.h): the shared-column half of the two-level
// condensed format (spmm_panel_kernels.hpp).
#include <hip/hip_runtime.h>

#include "voltrix/panel_plan.hpp"
#include "voltrix/spmm_panel_kernels.hpp"
#include "voltrix_capi.h"

namespace {

// X(FS, DEPTH, WAVES, RB, KS) over the ahead-of-time space
#define VOLTRIX_PANEL_SPACE(X)                                                                       \
  X(128, 3, 8, 4, 1) X(128, 4, 8, 4, 1) X(128, 6, 8, 4, 1) X(128, 8, 8, 4, 1) X(128, 4, 4, 4, 1) X(128, 6, 4, 4, 1) X(128, 8, 4, 4, 1) \
  X(128, 4, 8, 2, 1) X(128, 6, 8, 2, 1) X(128, 8, 8, 2, 1) X(128, 6, 4, 2, 1)                                    \
  X(64, 4, 4, 4, 1) X(64, 6, 4, 4, 1) X(64, 4, 8, 4, 2) X(64, 6, 8, 4, 2) X(64, 6, 4, 4, 2)                      \
  X(32, 4, 4, 4, 2) X(32, 6, 4, 4, 2) X(32, 6, 8, 4, 2) X(64, 6, 4, 2, 2) X(64, 6, 8, 2, 2) X(32, 6, 4, 2, 2) X(32, 6, 8, 2, 2) \
  X(128, 3, 4, 4, 1) X(128, 4, 4, 2, 1)

// the software-pipelined k-step loop (PanelTile<..., PIPE = true>; one k-step per ring slot): ksteps = 17
constexpr int kPipelined = 17;
static_assert(kPipelined == VOLTRIX_PANEL_KSTEPS_PIPELINED, "include/voltrix_capi.h");
#define VOLTRIX_PANEL_PIPE_SPACE(X) X(128, 3, 8, 4) X(128, 4, 8, 4) X(128, 4, 8, 2) X(128, 6, 8, 2) X(64, 4, 8, 4) X(64, 6, 8, 4)

template <bool BF16>
int dispatch(int fs, int depth, int waves, int rb, int ks, const int* panel_ptr, const int* panel_cols,
             const uint32_t* panel_bits, const int* panel_order, int num_nodes, int embedding_dim, const void* input,
             float* output, int accumulate, const float* out_scale, hipStream_t stream, int64_t input_rows,
             int slab_policy, const int* xcd_ptr, int max_panels_per_xcd, const int* parts = nullptr, int num_parts = 0,
             float* partials = nullptr) {
#define X(FS, D, W, RB, KS)                                                                                          \
  if (fs == FS && depth == D && waves == W && rb == RB && ks == KS)                                                  \
    return voltrix::launch_spmm_panel<voltrix::PanelTile<FS, D, W, RB, KS, BF16>>(                                   \
        panel_ptr, panel_cols, panel_bits, panel_order, num_nodes, embedding_dim, input, output, accumulate, out_scale, \
        stream, 0, 0, input_rows, slab_policy, xcd_ptr, max_panels_per_xcd, parts, num_parts, partials);
  VOLTRIX_PANEL_SPACE(X)
#undef X
#define X(FS, D, W, RB)                                                                                              \
  if (fs == FS && depth == D && waves == W && rb == RB && ks == kPipelined)                                           \
    return voltrix::launch_spmm_panel<voltrix::PanelTile<FS, D, W, RB, 1, BF16, true>>(                               \
        panel_ptr, panel_cols, panel_bits, panel_order, num_nodes, embedding_dim, input, output, accumulate, out_scale, \
        stream, 0, 0, input_rows, slab_policy, xcd_ptr, max_panels_per_xcd, parts, num_parts, partials);
  VOLTRIX_PANEL_PIPE_SPACE(X)
#undef X
  return voltrix::kErrBadConfig;
}

}  // namespace

extern "C" {

void voltrix_launch_spmm_panel_f16(void* panel_ptr, void* panel_cols, void* panel_bits, void* panel_order, void* xcd_ptr,
                                   int max_panels_per_xcd, int num_nodes, int embedding_dim, void* input, int64_t input_rows, void* output,
                                   int accumulate, int fs, int depth, int waves, int row_blocks, int ksteps,
                                   int slab_policy, void* out_scale, void* stream, int* return_code) {
  *return_code = dispatch<false>(fs, depth, waves, row_blocks, ksteps, static_cast<const int*>(panel_ptr),
                                 static_cast<const int*>(panel_cols), static_cast<const uint32_t*>(panel_bits),
                                 static_cast<const int*>(panel_order), num_nodes, embedding_dim, input,
                                 static_cast<float*>(output), accumulate, static_cast<const float*>(out_scale),
                                 static_cast<hipStream_t>(stream), input_rows, slab_policy,
                                 static_cast<const int*>(xcd_ptr), max_panels_per_xcd);
}

void voltrix_launch_spmm_panel_bf16(void* panel_ptr, void* panel_cols, void* panel_bits, void* panel_order, void* xcd_ptr,
                                    int max_panels_per_xcd, int num_nodes, int embedding_dim, void* input, int64_t input_rows, void* output,
                                    int accumulate, int fs, int depth, int waves, int row_blocks, int ksteps,
                                    int slab_policy, void* out_scale, void* stream, int* return_code) {
  *return_code = dispatch<true>(fs, depth, waves, row_blocks, ksteps, static_cast<const int*>(panel_ptr),
                                static_cast<const int*>(panel_cols), static_cast<const uint32_t*>(panel_bits),
                                static_cast<const int*>(panel_order), num_nodes, embedding_dim, input,
                                static_cast<float*>(output), accumulate, static_cast<const float*>(out_scale),
                                static_cast<hipStream_t>(stream), input_rows, slab_policy,
                                static_cast<const int*>(xcd_ptr), max_panels_per_xcd);
}

void voltrix_launch_spmm_panel_parts_f16(void* panel_ptr, void* panel_cols, void* panel_bits, void* parts, int num_parts,
                                         void* xcd_ptr, int max_parts_per_xcd, void* partials, int num_nodes, int embedding_dim,
                                         void* input, int64_t input_rows, void* output, int accumulate, int fs, int depth,
                                         int waves, int row_blocks, int ksteps, int slab_policy, void* out_scale, void* stream,
                                         int* return_code) {
  *return_code = parts == nullptr ? voltrix::kErrBadShape
                                  : dispatch<false>(fs, depth, waves, row_blocks, ksteps, static_cast<const int*>(panel_ptr),
                                                    static_cast<const int*>(panel_cols), static_cast<const uint32_t*>(panel_bits),
                                                    nullptr, num_nodes, embedding_dim, input, static_cast<float*>(output),
                                                    accumulate, static_cast<const float*>(out_scale),
                                                    static_cast<hipStream_t>(stream), input_rows, slab_policy,
                                                    static_cast<const int*>(xcd_ptr), max_parts_per_xcd,
                                                    static_cast<const int*>(parts), num_parts, static_cast<float*>(partials));
}

void voltrix_launch_spmm_panel_parts_bf16(void* panel_ptr, void* panel_cols, void* panel_bits, void* parts, int num_parts,
                                          void* xcd_ptr, int max_parts_per_xcd, void* partials, int num_nodes, int embedding_dim,
                                          void* input, int64_t input_rows, void* output, int accumulate, int fs, int depth,
                                          int waves, int row_blocks, int ksteps, int slab_policy, void* out_scale, void* stream,
                                          int* return_code) {
  *return_code = parts == nullptr ? voltrix::kErrBadShape
                                  : dispatch<true>(fs, depth, waves, row_blocks, ksteps, static_cast<const int*>(panel_ptr),
                                                   static_cast<const int*>(panel_cols), static_cast<const uint32_t*>(panel_bits),
                                                   nullptr, num_nodes, embedding_dim, input, static_cast<float*>(output),
                                                   accumulate, static_cast<const float*>(out_scale),
                                                   static_cast<hipStream_t>(stream), input_rows, slab_policy,
                                                   static_cast<const int*>(xcd_ptr), max_parts_per_xcd,
                                                   static_cast<const int*>(parts), num_parts, static_cast<float*>(partials));
}

void voltrix_launch_combine_panel_partials(void* cuts, int num_cuts, void* partials, void* output, int num_nodes,
                                           int embedding_dim, int panel_rows, int accumulate, void* stream, int* return_code) {
  *return_code = voltrix::launch_combine_panel_partials(static_cast<const int*>(cuts), num_cuts,
                                                        static_cast<const float*>(partials), static_cast<float*>(output),
                                                        num_nodes, embedding_dim, panel_rows, accumulate,
                                                        static_cast<hipStream_t>(stream));
}

void voltrix_launch_panel_order(void* panel_ptr, int num_panels, int group, void* xcd_ptr, void* order_out, void* stream,
                                int* return_code) {
  *return_code = voltrix::panel_order(static_cast<const int*>(panel_ptr), num_panels, group, static_cast<int*>(order_out),
                                      static_cast<hipStream_t>(stream), static_cast<const int*>(xcd_ptr));
}

int64_t voltrix_panel_plan_workspace_bytes(int num_nodes, int waves, int row_blocks) {
  return voltrix::panel_plan_workspace_bytes(num_nodes, waves, row_blocks);
}

void voltrix_launch_panel_plan_count(void* node_pointer, void* edge_list, int num_nodes, int num_cols, int64_t num_edges,
                                     int waves, int row_blocks, int tau, void* workspace, void* panel_ptr,
                                     void* resid_node_pointer, void* status, void* stream, int* return_code) {
  *return_code = voltrix::panel_plan_count(static_cast<const int*>(node_pointer), static_cast<const int*>(edge_list),
                                           num_nodes, num_cols, num_edges, waves, row_blocks, tau, workspace,
                                           static_cast<int*>(panel_ptr), static_cast<int*>(resid_node_pointer),
                                           static_cast<int*>(status), static_cast<hipStream_t>(stream));
}

void voltrix_launch_panel_plan_fill(void* node_pointer, void* edge_list, int num_nodes, int num_cols, int64_t num_edges,
                                    int waves, int row_blocks, int tau, void* workspace, void* panel_ptr,
                                    void* resid_node_pointer, int64_t total_ksteps, void* resid_edge_list,
                                    void* panel_cols, void* panel_bits, void* stream, int* return_code) {
  *return_code = voltrix::panel_plan_fill(static_cast<const int*>(node_pointer), static_cast<const int*>(edge_list),
                                          num_nodes, num_cols, num_edges, waves, row_blocks, tau, workspace,
                                          static_cast<const int*>(panel_ptr), static_cast<const int*>(resid_node_pointer),
                                          total_ksteps, static_cast<int*>(resid_edge_list), static_cast<int*>(panel_cols),
                                          static_cast<uint32_t*>(panel_bits), static_cast<hipStream_t>(stream));
}

}  // extern "C"
