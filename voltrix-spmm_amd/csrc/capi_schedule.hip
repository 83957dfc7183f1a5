// libvoltrix_hip.so -- schedule builders (include/voltrix_capi.h): the window kernel's unit table, built on the device
// from the handle's blk_offsets (no reference counterpart).
#include <hip/hip_runtime.h>

#include "voltrix/unit_table.hpp"
#include "voltrix_capi.h"

extern "C" {

int64_t voltrix_unit_table_workspace_bytes(int num_nodes) { return voltrix::unit_table_workspace_bytes(num_nodes); }

int64_t voltrix_unit_table_fill_workspace_bytes(int64_t num_units) {
  return voltrix::unit_table_fill_workspace_bytes(num_units);
}

void voltrix_launch_unit_table_count(void* blk_offsets, int num_nodes, int max_stages, void* xcd_ptr, void* workspace,
                                     void* header, void* stream, int* return_code) {
  *return_code = voltrix::unit_table_count(static_cast<const int*>(blk_offsets), num_nodes, max_stages, workspace,
                                           static_cast<int*>(header), static_cast<hipStream_t>(stream),
                                           static_cast<const int*>(xcd_ptr));
}

void voltrix_launch_unit_table_fill(void* blk_offsets, int num_nodes, void* xcd_ptr, void* workspace, void* fill_workspace,
                                    int num_units, int num_cuts, int top, void* units, void* unit_ptr, void* cuts,
                                    void* stream, int* return_code) {
  *return_code = voltrix::unit_table_fill(static_cast<const int*>(blk_offsets), num_nodes, workspace, fill_workspace,
                                          num_units, num_cuts, top, static_cast<int*>(units), static_cast<int*>(unit_ptr),
                                          static_cast<int*>(cuts), static_cast<hipStream_t>(stream),
                                          static_cast<const int*>(xcd_ptr));
}

}  // extern "C"
