// libvoltrix_hip.so -- stage-list executor entry point (include/voltrix_capi.h).
#include <hip/hip_runtime.h>

#include "voltrix/spmm_list_kernels.hpp"
#include "voltrix_capi.h"

namespace {

template <int FS, int D, int G>
int run(void* hspa_packed, void* hind, int num_nodes, int embedding_dim, void* input, void* output, void* entries,
        void* wave_ptr, int num_waves, void* stream) {
  return voltrix::launch_spmm_list<voltrix::SpmmListTile<FS, D, 1, G>>(
      static_cast<const uint32_t*>(hspa_packed), static_cast<const int*>(hind), num_nodes, embedding_dim,
      static_cast<const _Float16*>(input), static_cast<float*>(output), static_cast<const voltrix::int4v_t*>(entries),
      static_cast<const int*>(wave_ptr), num_waves, static_cast<hipStream_t>(stream));
}

}  // namespace

extern "C" void voltrix_launch_spmm_f16_list(void* hspa_packed, void* hind, int num_nodes, int embedding_dim,
                                             void* input, void* output, void* entries, void* wave_ptr, int num_waves,
                                             int fs, int depth, int groups, void* stream, int* return_code) {
#define X(FS, D, G)                                                                                          \
  if (fs == FS && depth == D && groups == G) {                                                               \
    *return_code = run<FS, D, G>(hspa_packed, hind, num_nodes, embedding_dim, input, output, entries, wave_ptr, \
                                 num_waves, stream);                                                         \
    return;                                                                                                  \
  }
  X(128, 3, 1) X(128, 3, 2) X(128, 3, 4) X(128, 3, 8)
  X(128, 4, 1) X(128, 4, 2) X(128, 4, 4) X(128, 4, 8)
  X(64, 3, 1) X(64, 3, 2) X(64, 3, 4) X(64, 3, 8)
  X(64, 4, 1) X(64, 4, 2) X(64, 4, 4) X(64, 4, 8)
#undef X
  *return_code = voltrix::kErrBadConfig;
}
