// libvoltrix_hip.so -- fp16-operand SpMM entry points (include/voltrix_capi.h).
#include "capi_common.hpp"

using namespace voltrix_capi;

extern "C" {

void voltrix_launch_spmm_f16_tile(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                                  int embedding_dim, void* input, void* output, int fs, int depth, int waves,
                                  void* window_order, void* out_scale, void* stream, int* return_code) {
  (void)num_edges;  // unused by the reference's live kernels as well
  *return_code = dispatch_spmm<2, _Float16>(fs, depth, waves, static_cast<const int*>(blk_offsets),
                                            static_cast<const uint32_t*>(hspa_packed), static_cast<const int*>(hind),
                                            num_nodes, embedding_dim, static_cast<const _Float16*>(input),
                                            static_cast<float*>(output), static_cast<hipStream_t>(stream),
                                            static_cast<const int*>(window_order),
                                            static_cast<const float*>(out_scale));
}

void voltrix_launch_spmm_f16_sched(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                                   int embedding_dim, void* input, void* output, int fs, int depth, int waves,
                                   void* window_order, void* out_scale, int atomic_out,
                                   void* units, void* unit_ptr, int max_units_per_xcd, void* partials, void* row_map,
                                   int units_per_wave, void* stream,
                                   int* return_code) {
  (void)num_edges;
  *return_code = dispatch_spmm<2, _Float16>(fs, depth, waves, static_cast<const int*>(blk_offsets),
                                            static_cast<const uint32_t*>(hspa_packed), static_cast<const int*>(hind),
                                            num_nodes, embedding_dim, static_cast<const _Float16*>(input),
                                            static_cast<float*>(output), static_cast<hipStream_t>(stream),
                                            static_cast<const int*>(window_order),
                                            static_cast<const float*>(out_scale), atomic_out,
                                            static_cast<const int*>(units), static_cast<const int*>(unit_ptr),
                                            max_units_per_xcd, static_cast<float*>(partials), static_cast<const int*>(row_map),
                                            units_per_wave);
}

void voltrix_launch_combine_partials(void* cuts, int num_cuts, void* partials, void* output, int num_nodes,
                                     int embedding_dim, int accumulate, void* row_map, void* stream, int* return_code) {
  *return_code = voltrix::combine_partials(static_cast<const int*>(cuts), num_cuts, static_cast<const float*>(partials),
                                           static_cast<float*>(output), num_nodes, embedding_dim, accumulate,
                                           static_cast<hipStream_t>(stream), static_cast<const int*>(row_map));
}

void voltrix_launch_spmm_f16(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                             int embedding_dim, void* input, void* output, void* stream, int* return_code) {
  const TileId t = default_tile(embedding_dim, true);
  voltrix_launch_spmm_f16_tile(blk_offsets, hspa_packed, hind, num_nodes, num_edges, embedding_dim, input, output, t.fs,
                               t.depth, t.waves, nullptr, nullptr, stream, return_code);
}

// The reference's launch() arguments for fp32 features (jit_kernels/spmm.py:78-88) on the 16-bit matrix-core path: the
// operand is rounded to fp16 after one power-of-two rescale (cast_f32_to_f16_scaled: same 10-bit mantissa as the
// reference's TF32 rounding, fp32's range) into the caller's workspace, the epilogue undoes the scale.
int64_t voltrix_spmm_f32_workspace_bytes(int64_t input_rows, int embedding_dim) {
  if (input_rows < 0 || embedding_dim < 0) return -1;
  return 16 + input_rows * (int64_t)embedding_dim * 2;
}

void voltrix_launch_spmm_f32_as_f16(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                                    int embedding_dim, void* input, int64_t input_rows, void* output, void* workspace,
                                    void* stream, int* return_code) {
  if (input_rows < 0 || workspace == nullptr || ((uintptr_t)workspace & 15) || embedding_dim % 8 != 0) {
    *return_code = voltrix::kErrBadShape;
    return;
  }
  float* const scale = static_cast<float*>(workspace);
  _Float16* const operand = reinterpret_cast<_Float16*>(static_cast<char*>(workspace) + 16);
  *return_code = voltrix::cast_f32_to_f16_scaled(static_cast<const float*>(input), operand,
                                                 input_rows * (int64_t)embedding_dim, scale,
                                                 static_cast<hipStream_t>(stream));
  if (*return_code != voltrix::kOk) return;
  const TileId t = default_tile(embedding_dim, true);
  voltrix_launch_spmm_f16_tile(blk_offsets, hspa_packed, hind, num_nodes, num_edges, embedding_dim, operand, output, t.fs,
                               t.depth, t.waves, nullptr, scale, stream, return_code);
}

void voltrix_launch_window_order(void* blk_offsets, int num_nodes, int chunk, void* order_out, void* stream,
                                 int* return_code) {
  *return_code = voltrix::launch_window_order(static_cast<const int*>(blk_offsets), num_nodes, chunk,
                                              static_cast<int*>(order_out), static_cast<hipStream_t>(stream));
}

void voltrix_launch_cast_f32_f16(void* src, void* dst, int64_t count, void* stream, int* return_code) {
  *return_code = voltrix::cast_f32_to_f16(static_cast<const float*>(src), static_cast<_Float16*>(dst), count,
                                          static_cast<hipStream_t>(stream));
}

void voltrix_launch_cast_f32_f16_scaled(void* src, void* dst, int64_t count, void* scale, void* stream,
                                        int* return_code) {
  *return_code = voltrix::cast_f32_to_f16_scaled(static_cast<const float*>(src), static_cast<_Float16*>(dst), count,
                                                 static_cast<float*>(scale), static_cast<hipStream_t>(stream));
}

void voltrix_launch_scale_rows(void* src, void* scale, void* dst, int64_t rows, int num_feats, int dtype, void* stream,
                               int* return_code) {
  *return_code = voltrix::scale_rows(src, static_cast<const float*>(scale), dst, rows, num_feats, dtype,
                                     static_cast<hipStream_t>(stream));
}

}  // extern "C"

namespace voltrix_capi {
int num_tiles_f16() { return num_tiles<2>(); }
bool tile_at_f16(int i, TileId* t) { return tile_at<2>(i, t); }
}  // namespace voltrix_capi
