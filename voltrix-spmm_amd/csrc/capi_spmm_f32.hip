// libvoltrix_hip.so -- fp32-operand (exact) SpMM entry points: the reference's own launch() signature.
#include "capi_common.hpp"

using namespace voltrix_capi;

namespace voltrix_capi {
int num_tiles_f16();
bool tile_at_f16(int i, TileId* t);
}  // namespace voltrix_capi

extern "C" {

int voltrix_abi_version(void) { return VOLTRIX_ABI_VERSION; }

void voltrix_launch_spmm_f32_tile(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                                  int embedding_dim, void* input, void* output, int fs, int depth, int waves,
                                  void* window_order, void* out_scale, void* stream, int* return_code) {
  (void)num_edges;
  *return_code = dispatch_spmm<4, float>(fs, depth, waves, static_cast<const int*>(blk_offsets),
                                         static_cast<const uint32_t*>(hspa_packed), static_cast<const int*>(hind),
                                         num_nodes, embedding_dim, static_cast<const float*>(input),
                                         static_cast<float*>(output), static_cast<hipStream_t>(stream),
                                         static_cast<const int*>(window_order), static_cast<const float*>(out_scale));
}

// the reference's launch() argument list (voltrix/jit_kernels/spmm.py:78-88)
void voltrix_launch_spmm(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                         int embedding_dim, void* input, void* output, void* stream, int* return_code) {
  const TileId t = default_tile(embedding_dim, false);
  voltrix_launch_spmm_f32_tile(blk_offsets, hspa_packed, hind, num_nodes, num_edges, embedding_dim, input, output, t.fs,
                               t.depth, t.waves, nullptr, nullptr, stream, return_code);
}

void voltrix_spmm_default_tile(int embedding_dim, int is_f16, int* fs, int* depth, int* waves) {
  const TileId t = default_tile(embedding_dim, is_f16 != 0);
  *fs = t.fs;
  *depth = t.depth;
  *waves = t.waves;
}

int voltrix_spmm_num_tiles(int is_f16) { return is_f16 ? num_tiles_f16() : num_tiles<4>(); }

void voltrix_spmm_tile_at(int is_f16, int index, int* fs, int* depth, int* waves) {
  TileId t{0, 0, 0};
  if (is_f16) tile_at_f16(index, &t); else tile_at<4>(index, &t);
  *fs = t.fs;
  *depth = t.depth;
  *waves = t.waves;
}

}  // extern "C"
