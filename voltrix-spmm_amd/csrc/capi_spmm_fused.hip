// libvoltrix_hip.so -- the two-level format in one launch (include/voltrix_capi.h; spmm_fused_kernels.hpp) and the
// builder of its residual stage records (fused_plan.hpp).
#include <hip/hip_runtime.h>

#include "voltrix/fused_plan.hpp"
#include "voltrix/spmm_fused_kernels.hpp"
#include "voltrix_capi.h"

namespace {

// X(FS, DP) over the ahead-of-time space
#define VOLTRIX_FUSED_SPACE(X) X(128, 3) X(64, 3) X(64, 4) X(32, 3) X(32, 4)

template <bool BF16>
int dispatch(int fs, int depth, const int* panel_ptr, const int* panel_cols, const uint32_t* panel_bits,
             const int* panel_order, const int* wave_ptr, const uint32_t* records, int num_nodes, int embedding_dim,
             const void* input, float* output, const float* out_scale, hipStream_t stream, int pace_blocks,
             const int* xcd_ptr, int max_panels_per_xcd) {
#define X(FS, D)                                                                                                  \
  if (fs == FS && depth == D)                                                                                     \
    return voltrix::launch_spmm_fused<voltrix::FusedTile<FS, D, BF16>>(panel_ptr, panel_cols, panel_bits, panel_order, \
                                                                       wave_ptr, records, num_nodes, embedding_dim, \
                                                                       input, output, out_scale, stream, pace_blocks, \
                                                                       xcd_ptr, max_panels_per_xcd);
  VOLTRIX_FUSED_SPACE(X)
#undef X
  return voltrix::kErrBadConfig;
}

}  // namespace

extern "C" {

void voltrix_launch_spmm_fused_f16(void* panel_ptr, void* panel_cols, void* panel_bits, void* panel_order, void* xcd_ptr,
                                   int max_panels_per_xcd, void* wave_ptr, void* records, int num_nodes, int embedding_dim, void* input,
                                   void* output, int fs, int depth, int pace_blocks, void* out_scale, void* stream,
                                   int* return_code) {
  *return_code = dispatch<false>(fs, depth, static_cast<const int*>(panel_ptr), static_cast<const int*>(panel_cols),
                                 static_cast<const uint32_t*>(panel_bits), static_cast<const int*>(panel_order),
                                 static_cast<const int*>(wave_ptr), static_cast<const uint32_t*>(records), num_nodes,
                                 embedding_dim, input, static_cast<float*>(output),
                                 static_cast<const float*>(out_scale), static_cast<hipStream_t>(stream), pace_blocks,
                                 static_cast<const int*>(xcd_ptr), max_panels_per_xcd);
}

void voltrix_launch_spmm_fused_bf16(void* panel_ptr, void* panel_cols, void* panel_bits, void* panel_order, void* xcd_ptr,
                                    int max_panels_per_xcd, void* wave_ptr, void* records, int num_nodes, int embedding_dim, void* input,
                                    void* output, int fs, int depth, int pace_blocks, void* out_scale, void* stream,
                                    int* return_code) {
  *return_code = dispatch<true>(fs, depth, static_cast<const int*>(panel_ptr), static_cast<const int*>(panel_cols),
                                static_cast<const uint32_t*>(panel_bits), static_cast<const int*>(panel_order),
                                static_cast<const int*>(wave_ptr), static_cast<const uint32_t*>(records), num_nodes,
                                embedding_dim, input, static_cast<float*>(output),
                                static_cast<const float*>(out_scale), static_cast<hipStream_t>(stream), pace_blocks,
                                static_cast<const int*>(xcd_ptr), max_panels_per_xcd);
}

void voltrix_fused_panel_geometry(int* waves, int* row_blocks) {
  *waves = voltrix::kFusedWaves;
  *row_blocks = voltrix::kFusedRowBlocks;
}

int64_t voltrix_fused_records_workspace_bytes(int num_nodes) { return voltrix::fused_records_workspace_bytes(num_nodes); }

void voltrix_launch_fused_records_count(void* blk_offsets, void* hspa_packed, int num_nodes, void* workspace, void* wave_ptr,
                                        void* stream, int* return_code) {
  *return_code = voltrix::fused_records_count(static_cast<const int*>(blk_offsets), static_cast<const uint32_t*>(hspa_packed),
                                              num_nodes, workspace, static_cast<int*>(wave_ptr),
                                              static_cast<hipStream_t>(stream));
}

void voltrix_launch_fused_records_fill(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, void* wave_ptr,
                                       int64_t num_records, void* records, void* stream, int* return_code) {
  *return_code = voltrix::fused_records_fill(static_cast<const int*>(blk_offsets), static_cast<const uint32_t*>(hspa_packed),
                                             static_cast<const int*>(hind), num_nodes, static_cast<const int*>(wave_ptr),
                                             num_records, static_cast<uint32_t*>(records), static_cast<hipStream_t>(stream));
}

}  // extern "C"
