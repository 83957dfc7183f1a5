// libvoltrix_hip.so -- Cuthill-McKee row order on the device (include/voltrix_capi.h; voltrix/reorder_kernels.hpp,
// profiles/HISTORY.md section 3.4).  No reference counterpart: the reference reads externally reordered graphs
// (bench/graph_gen.py:42-45).
#include <hip/hip_runtime.h>

#include "voltrix/reorder_kernels.hpp"
#include "voltrix_capi.h"

namespace {
voltrix::BfsGraph graph_of(void* indptr, void* indices, void* t_indptr, void* t_indices, int num_nodes, int t_rows) {
  return voltrix::BfsGraph{static_cast<const int*>(indptr), static_cast<const int*>(indices),
                           static_cast<const int*>(t_indptr), static_cast<const int*>(t_indices), num_nodes, t_rows};
}
}  // namespace

extern "C" {

int64_t voltrix_csr_transpose_workspace_bytes(int64_t num_edges) { return voltrix::csr_transpose_workspace_bytes(num_edges); }

void voltrix_launch_csr_transpose(void* indptr, void* indices, int num_rows, int num_cols, int64_t num_edges, void* workspace,
                                  void* t_indptr, void* t_indices, void* stream, int* return_code) {
  *return_code = voltrix::csr_transpose(static_cast<const int*>(indptr), static_cast<const int*>(indices), num_rows, num_cols,
                                        num_edges, workspace, static_cast<int*>(t_indptr), static_cast<int*>(t_indices),
                                        static_cast<hipStream_t>(stream));
}

void voltrix_launch_bfs_seed(int start, int num_nodes, void* level, void* queue, void* ctrl, void* level_off, void* stream,
                             int* return_code) {
  *return_code = voltrix::bfs_seed(start, num_nodes, static_cast<int*>(level), static_cast<int*>(queue),
                                   static_cast<int*>(ctrl), static_cast<int*>(level_off), static_cast<hipStream_t>(stream));
}

void voltrix_launch_bfs_levels(void* indptr, void* indices, void* t_indptr, void* t_indices, int num_nodes, int t_rows,
                               void* level, void* queue, void* ctrl, void* level_off, int wide_levels, void* stream,
                               int* return_code) {
  *return_code = voltrix::bfs_levels(graph_of(indptr, indices, t_indptr, t_indices, num_nodes, t_rows),
                                     static_cast<int*>(level), static_cast<int*>(queue), static_cast<int*>(ctrl),
                                     static_cast<int*>(level_off), wide_levels, static_cast<hipStream_t>(stream));
}

int64_t voltrix_cm_rank_workspace_bytes(int64_t max_level) { return voltrix::cm_rank_workspace_bytes(max_level); }

void voltrix_launch_cm_rank(void* indptr, void* indices, void* t_indptr, void* t_indices, int num_nodes, int t_rows,
                            void* level, void* rank, void* tie, void* queue, void* level_off, const int* level_off_host,
                            int num_levels, int base, void* workspace, void* stream, int* return_code) {
  *return_code = voltrix::cm_rank(graph_of(indptr, indices, t_indptr, t_indices, num_nodes, t_rows),
                                  static_cast<const int*>(level), static_cast<int*>(rank), static_cast<const int*>(tie),
                                  static_cast<int*>(queue), static_cast<const int*>(level_off), level_off_host, num_levels,
                                  base, workspace, static_cast<hipStream_t>(stream));
}

void voltrix_launch_chol_inv_transposed(void* gram, int k, double eps, void* out, void* stream, int* return_code) {
  *return_code = voltrix::chol_inv_transposed(static_cast<const float*>(gram), k, eps, static_cast<float*>(out),
                                              static_cast<hipStream_t>(stream));
}

}  // extern "C"
