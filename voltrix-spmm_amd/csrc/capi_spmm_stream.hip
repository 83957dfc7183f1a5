// libvoltrix_hip.so -- stream kernel entry points (include/voltrix_capi.h): the window format walked as a stream of stages
// (spmm_stream_kernels.hpp), for the ahead-of-time tiles (fs x depth x waves).
#include <hip/hip_runtime.h>

#include "voltrix/spmm_stream_kernels.hpp"
#include "voltrix_capi.h"

namespace {

// X(FS, DEPTH, WAVES) over the ahead-of-time space (the tuner's stream points: jit_kernels/spmm.py::_tile_space)
#define VOLTRIX_STREAM_SPACE(X)                                                               \
  X(128, 2, 1) X(128, 3, 1) X(128, 4, 1) X(128, 2, 2) X(128, 3, 2) X(64, 2, 1) X(64, 3, 1) X(64, 4, 1) X(64, 2, 2) X(64, 3, 2) \
  X(32, 2, 1) X(32, 3, 1) X(32, 4, 1) X(32, 2, 2) X(32, 3, 2)

template <bool BF16>
int dispatch(int fs, int depth, int waves, const uint32_t* hspa_packed, const int* hind, int num_nodes, int embedding_dim,
             const void* input, int64_t input_rows, float* output, const int* units, const int* runs, const int* run_ptr,
             int max_runs_per_xcd, float* partials, const float* out_scale, int slab_policy, hipStream_t stream) {
  if (fs == 0) {   // the default tile: the slab by width, two ring slots, one wave per workgroup (nine waves per CU)
    fs = embedding_dim <= 32 ? 32 : (embedding_dim <= 64 ? 64 : 128);
    depth = 2;
    waves = 1;
  }
#define X(FS, D, W)                                                                                                   \
  if (fs == FS && depth == D && waves == W)                                                                           \
    return voltrix::launch_spmm_stream<voltrix::SpmmTile<FS, D, W, 2, BF16, false>>(                                  \
        hspa_packed, hind, num_nodes, embedding_dim, input, output, stream, units, runs, run_ptr, max_runs_per_xcd, partials, \
        out_scale, 0, 0, input_rows, slab_policy);
  VOLTRIX_STREAM_SPACE(X)
#undef X
  return voltrix::kErrBadConfig;
}

}  // namespace

extern "C" {

void voltrix_launch_spmm_stream_f16(void* hspa_packed, void* hind, int num_nodes, int embedding_dim, void* input,
                                    int64_t input_rows, void* output, void* units, void* runs, void* run_ptr,
                                    int max_runs_per_xcd, void* partials, void* out_scale, int fs, int depth, int waves,
                                    int slab_policy, void* stream, int* return_code) {
  *return_code = dispatch<false>(fs, depth, waves, static_cast<const uint32_t*>(hspa_packed), static_cast<const int*>(hind),
                                 num_nodes, embedding_dim, input, input_rows, static_cast<float*>(output),
                                 static_cast<const int*>(units), static_cast<const int*>(runs), static_cast<const int*>(run_ptr),
                                 max_runs_per_xcd, static_cast<float*>(partials), static_cast<const float*>(out_scale),
                                 slab_policy, static_cast<hipStream_t>(stream));
}

void voltrix_launch_spmm_stream_bf16(void* hspa_packed, void* hind, int num_nodes, int embedding_dim, void* input,
                                     int64_t input_rows, void* output, void* units, void* runs, void* run_ptr,
                                     int max_runs_per_xcd, void* partials, void* out_scale, int fs, int depth, int waves,
                                     int slab_policy, void* stream, int* return_code) {
  *return_code = dispatch<true>(fs, depth, waves, static_cast<const uint32_t*>(hspa_packed), static_cast<const int*>(hind),
                                num_nodes, embedding_dim, input, input_rows, static_cast<float*>(output),
                                static_cast<const int*>(units), static_cast<const int*>(runs), static_cast<const int*>(run_ptr),
                                max_runs_per_xcd, static_cast<float*>(partials), static_cast<const float*>(out_scale),
                                slab_policy, static_cast<hipStream_t>(stream));
}

}  // extern "C"
