// libvoltrix_hip.so -- bfloat16-operand SpMM entry points (include/voltrix_capi.h): the fp16 kernels with
// v_mfma_f32_16x16x32_bf16 (same tiles, same LDS image, same A-fragment bits).
#include "capi_common.hpp"

using namespace voltrix_capi;

extern "C" {

void voltrix_launch_spmm_bf16_tile(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                                   int embedding_dim, void* input, void* output, int fs, int depth, int waves,
                                   void* window_order, void* out_scale, void* stream, int* return_code) {
  (void)num_edges;
  *return_code = dispatch_spmm<2, voltrix::bfloat16_bits, true>(
      fs, depth, waves, static_cast<const int*>(blk_offsets), static_cast<const uint32_t*>(hspa_packed),
      static_cast<const int*>(hind), num_nodes, embedding_dim, static_cast<const voltrix::bfloat16_bits*>(input),
      static_cast<float*>(output), static_cast<hipStream_t>(stream), static_cast<const int*>(window_order),
      static_cast<const float*>(out_scale));
}

void voltrix_launch_spmm_bf16_sched(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                                    int embedding_dim, void* input, void* output, int fs, int depth, int waves,
                                    void* window_order, void* out_scale, int atomic_out,
                                    void* units, void* unit_ptr, int max_units_per_xcd, void* partials, void* row_map,
                                   int units_per_wave, void* stream,
                                    int* return_code) {
  (void)num_edges;
  *return_code = dispatch_spmm<2, voltrix::bfloat16_bits, true>(
      fs, depth, waves, static_cast<const int*>(blk_offsets), static_cast<const uint32_t*>(hspa_packed),
      static_cast<const int*>(hind), num_nodes, embedding_dim, static_cast<const voltrix::bfloat16_bits*>(input),
      static_cast<float*>(output), static_cast<hipStream_t>(stream), static_cast<const int*>(window_order),
      static_cast<const float*>(out_scale), atomic_out,
                                            static_cast<const int*>(units), static_cast<const int*>(unit_ptr),
                                            max_units_per_xcd, static_cast<float*>(partials), static_cast<const int*>(row_map),
                                            units_per_wave);
}

void voltrix_launch_spmm_bf16(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                              int embedding_dim, void* input, void* output, void* stream, int* return_code) {
  const TileId t = default_tile(embedding_dim, true);
  voltrix_launch_spmm_bf16_tile(blk_offsets, hspa_packed, hind, num_nodes, num_edges, embedding_dim, input, output,
                                t.fs, t.depth, t.waves, nullptr, nullptr, stream, return_code);
}

}  // extern "C"

