// libvoltrix_hip.so -- preprocess entry points (include/voltrix_capi.h): the reference's three preprocess launch()
// argument lists plus the fused GPU preprocess.
#include <hip/hip_runtime.h>

#include "voltrix/bmat_kernels.hpp"
#include "voltrix/csr_preprocess.hpp"
#include "voltrix_capi.h"

extern "C" {

// voltrix/jit_kernels/preprocess.py:57-65  (host pointers)
void voltrix_launch_preprocess(void* edge_list, void* node_pointer, int num_nodes, void* block_partition,
                               void* edge_to_column, void* edge_to_row, void* pointer1, int* return_code) {
  *return_code = voltrix::preprocess(static_cast<const int32_t*>(edge_list), static_cast<const int32_t*>(node_pointer),
                                     num_nodes, VOLTRIX_BLK_H, VOLTRIX_BLK_W, static_cast<int32_t*>(block_partition),
                                     static_cast<int32_t*>(edge_to_column), static_cast<int32_t*>(edge_to_row),
                                     static_cast<int32_t*>(pointer1));
}

// voltrix/jit_kernels/hmat_gem.py:56-68  (device pointers; null stream like bmat_kernels.cuh:204)
void voltrix_launch_hmat_gen(void* node_pointer, void* edge_list, void* block_partition, void* edge_to_column,
                             void* edge_to_row, void* pointer1, int num_row_windows, int num_nodes, int num_edges,
                             void* hspa, void* hind, int* return_code) {
  *return_code = voltrix::hmat_hip(static_cast<const int32_t*>(node_pointer), static_cast<const int32_t*>(edge_list),
                                   static_cast<const int32_t*>(block_partition),
                                   static_cast<const int32_t*>(edge_to_column), static_cast<const int32_t*>(edge_to_row),
                                   static_cast<const int32_t*>(pointer1), num_row_windows, num_nodes, num_edges,
                                   static_cast<float*>(hspa), static_cast<int*>(hind), nullptr);
}

// voltrix/jit_kernels/bmat_swizzle.py:38-43
void voltrix_launch_hmat_packed_swizzle(int num_row_windows, void* pointer1, void* hspa, void* hspa_packed,
                                        int* return_code) {
  *return_code = voltrix::hmat_packed_swizzle_hip(num_row_windows, static_cast<const int32_t*>(pointer1),
                                                  static_cast<const float*>(hspa), static_cast<uint32_t*>(hspa_packed),
                                                  nullptr);
}

int64_t voltrix_csr_preprocess_workspace_bytes(int num_nodes, int num_cols, int64_t num_edges, int path) {
  return voltrix::csr_preprocess_workspace_bytes(num_nodes, num_cols, num_edges, path);
}

void voltrix_launch_csr_window_count(void* node_pointer, void* edge_list, int num_nodes, int num_cols, int64_t num_edges,
                                     int path, void* workspace, void* block_partition, void* pointer1, void* status,
                                     void* stream, int* return_code) {
  *return_code = voltrix::csr_window_count(static_cast<const int*>(node_pointer), static_cast<const int*>(edge_list),
                                           num_nodes, num_cols, num_edges, workspace, static_cast<int*>(block_partition),
                                           static_cast<int*>(pointer1), static_cast<int*>(status),
                                           static_cast<hipStream_t>(stream), path);
}

void voltrix_launch_csr_fill(void* node_pointer, void* edge_list, int num_nodes, int num_cols, int64_t num_edges,
                             int path, void* workspace, void* pointer1, void* hspa_packed, void* hind, void* stream,
                             int* return_code) {
  *return_code = voltrix::csr_fill(static_cast<const int*>(node_pointer), static_cast<const int*>(edge_list), num_nodes,
                                   num_cols, num_edges, workspace, static_cast<const int*>(pointer1),
                                   static_cast<uint32_t*>(hspa_packed), static_cast<int*>(hind),
                                   static_cast<hipStream_t>(stream), path);
}

}  // extern "C"
