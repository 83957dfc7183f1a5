// libvoltrix_hip.so -- schedule builders (include/voltrix_capi.h): the window kernel's unit table, built on the device
// from the handle's blk_offsets (no reference counterpart; profiles/HISTORY.md section 3.2); round 4: the XCD ranges of equal work and
// the panel kernel's piece table (schedule_tables.hpp; profiles/HISTORY.md section 3.3).
#include <hip/hip_runtime.h>

#include "voltrix/schedule_tables.hpp"
#include "voltrix/stream_table.hpp"
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

void voltrix_launch_xcd_ranges_of_work(void* work, int num_items, int align, void* xcd_ptr, void* stream, int* return_code) {
  *return_code = voltrix::xcd_ranges_of_work(static_cast<const int*>(work), num_items, align, static_cast<int*>(xcd_ptr),
                                             static_cast<hipStream_t>(stream));
}

void voltrix_launch_xcd_ranges_of_windows(void* blk_offsets, int num_nodes, int align, void* xcd_ptr, void* stream,
                                          int* return_code) {
  *return_code = voltrix::xcd_ranges_of_windows(static_cast<const int*>(blk_offsets), num_nodes, align,
                                                static_cast<int*>(xcd_ptr), static_cast<hipStream_t>(stream));
}

void voltrix_launch_xcd_ranges_of_panels(void* panel_ptr, void* resid_blk_offsets, int num_nodes, int panel_rows,
                                         int kstep_cost_x10, void* xcd_ptr, void* window_xcd_ptr, void* stream,
                                         int* return_code) {
  *return_code = voltrix::xcd_ranges_of_panels(static_cast<const int*>(panel_ptr), static_cast<const int*>(resid_blk_offsets),
                                               num_nodes, panel_rows, kstep_cost_x10, static_cast<int*>(xcd_ptr),
                                               static_cast<int*>(window_xcd_ptr), static_cast<hipStream_t>(stream));
}

int64_t voltrix_panel_parts_workspace_bytes(int num_panels) { return voltrix::panel_parts_workspace_bytes(num_panels); }

void voltrix_launch_panel_parts_count(void* panel_ptr, int num_panels, int cap, void* panel_xcd_ptr, void* workspace,
                                      void* header, void* stream, int* return_code) {
  *return_code = voltrix::panel_parts_count(static_cast<const int*>(panel_ptr), num_panels, cap,
                                            static_cast<const int*>(panel_xcd_ptr), workspace, static_cast<int*>(header),
                                            static_cast<hipStream_t>(stream));
}

void voltrix_launch_panel_parts_fill(void* panel_ptr, int num_panels, int cap, void* panel_xcd_ptr, void* workspace,
                                     void* parts, void* part_xcd_ptr, void* cuts, void* stream, int* return_code) {
  *return_code = voltrix::panel_parts_fill(static_cast<const int*>(panel_ptr), num_panels, cap,
                                           static_cast<const int*>(panel_xcd_ptr), workspace, static_cast<int*>(parts),
                                           static_cast<int*>(part_xcd_ptr), static_cast<int*>(cuts),
                                           static_cast<hipStream_t>(stream));
}

int64_t voltrix_stream_table_workspace_bytes(int num_nodes) { return voltrix::stream_table_workspace_bytes(num_nodes); }

int64_t voltrix_stream_table_fill_workspace_bytes(int64_t num_units) {
  return voltrix::stream_table_fill_workspace_bytes(num_units);
}

void voltrix_launch_stream_table_count(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int run_cost,
                                       int cut_stages, void* workspace, void* header, void* stream, int* return_code) {
  *return_code = voltrix::stream_table_count(static_cast<const int*>(blk_offsets), static_cast<const uint32_t*>(hspa_packed),
                                             static_cast<const int*>(hind), num_nodes, run_cost, cut_stages, workspace,
                                             static_cast<int*>(header), static_cast<hipStream_t>(stream));
}

void voltrix_launch_stream_table_fill(void* blk_offsets, int num_nodes, void* workspace, void* fill_workspace, int num_units,
                                      int num_cuts, int run_bound, int run_cost, void* units, void* cuts, void* runs,
                                      void* run_ptr, void* header2, void* stream, int* return_code) {
  *return_code = voltrix::stream_table_fill(static_cast<const int*>(blk_offsets), num_nodes, workspace, fill_workspace,
                                            num_units, num_cuts, run_bound, run_cost, static_cast<int*>(units),
                                            static_cast<int*>(cuts), static_cast<int*>(runs), static_cast<int*>(run_ptr),
                                            static_cast<int*>(header2), static_cast<hipStream_t>(stream));
}

}  // extern "C"
