/*
 * Voltrix-SpMM for MI355X (gfx950) -- C-ABI of libvoltrix_hip.so (the drop-in boundary).
 *
 * The reference (YaqiXia/Voltrix-SpMM) has exactly one process-internal FFI: each kernel module is
 * JIT-compiled into its own kernel.so exporting
 *
 *     extern "C" void launch(<args...>, int& __return_code);
 *
 * (voltrix/jit/template.py:104-123) which voltrix/jit/runtime.py:38-52 calls through ctypes with tensors
 * as data_ptr() -> void*, Python int -> C int, bool -> bool, torch.cuda.Stream -> void* (stream handle)
 * and `int&` passed as ctypes.byref(c_int), i.e. a pointer at ABI level.
 *
 * This header declares the same four argument lists as named symbols of ONE ahead-of-time library
 * (voltrix_launch_*), plus the gfx950 extensions (fp16 operand, tile selection, fused GPU preprocess).
 * The JIT layer of the Python package (voltrix/jit) still generates per-kernel `launch` wrappers with
 * these exact argument lists; both routes call the same voltrix:: host launchers
 * (the .hpp headers under voltrix-spmm_amd/voltrix/include/voltrix).
 *
 * Contract shared by every entry point
 *   - plain C types only; every buffer is owned by the caller (the reference allocates everything in
 *     Python, voltrix/spmm/spmm.py:28-57,101); the library never allocates or frees device memory and keeps
 *     no state between calls;
 *   - device work is enqueued asynchronously on `stream` (a hipStream_t; NULL = the null stream, which is
 *     where the reference runs its preprocess kernels, bmat_kernels.cuh:204,234); no host synchronisation;
 *   - *return_code is ALWAYS written: 0 on success, a voltrix_rc value otherwise (the reference plumbs
 *     __return_code but never sets it; on errors it throws through ctypes or exit(1)s,
 *     spmm_kernels.cuh:28-45);
 *   - nothing is printed (the reference's preprocess printf's, bmat_kernels.cuh:309-310).
 */
#ifndef VOLTRIX_CAPI_H_
#define VOLTRIX_CAPI_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VOLTRIX_ABI_VERSION 2 /* 2 (round 4): explicit path / slab_policy / input_rows arguments instead of environment reads */

#define VOLTRIX_CSR_AUTO (-1)  /* `path` of the fused preprocess entry points: the library's rule (0 sort, 1 bitmap, 2 mixed) */
#define VOLTRIX_SLAB_AUTO (-1) /* `slab_policy` of the panel kernel (0 one grid over all column slabs, 1 one launch per group) */
#define VOLTRIX_BLK_H 16 /* rows per row window           (voltrix/spmm/spmm.py:12) */
#define VOLTRIX_BLK_W 8  /* condensed columns per TC block (voltrix/spmm/spmm.py:13) */

enum voltrix_rc {
  VOLTRIX_OK = 0,
  VOLTRIX_ERR_BAD_SHAPE = 1,  /* negative sizes, embedding_dim % 8 != 0 (fp16) / % 4 (fp32), misaligned pointer */
  VOLTRIX_ERR_LAUNCH = 2,     /* hipGetLastError() after the launch */
  VOLTRIX_ERR_BAD_CONFIG = 3, /* tile (fs, depth, waves) not instantiated */
  VOLTRIX_ERR_OVERFLOW = 4,   /* total TC blocks exceed int32 (the handle stores int32 offsets) */
  VOLTRIX_ERR_DUPLICATE = 5
};

int voltrix_abi_version(void);

/* ---- the reference's four launch() argument lists ------------------------------------------------------- */

/* Replaces launch() of kernel.preprocess_kernel.* -- voltrix/jit_kernels/preprocess.py:57-65 ->
 * voltrix::preprocess, bmat_kernels.cuh:264-320.  ALL pointers are HOST int32 arrays:
 * edge_list[E], node_pointer[num_nodes+1], block_partition[W], edge_to_column[E], edge_to_row[E],
 * pointer1[W+1], W = ceil(num_nodes/16). */
void voltrix_launch_preprocess(void* edge_list, void* node_pointer, int num_nodes, void* block_partition,
                               void* edge_to_column, void* edge_to_row, void* pointer1, int* return_code);

/* Replaces launch() of kernel.hmat_gen_kernel.* -- voltrix/jit_kernels/hmat_gem.py:56-68 -> voltrix::hmat_cuda,
 * bmat_kernels.cuh:195-212 (kernel :21-111).  DEVICE pointers; hspa float[T*128] and hind int[T*8] are fully
 * written (zero-filled then scattered), T = pointer1[num_row_windows].  Runs on the null stream like the
 * reference (:204). */
void voltrix_launch_hmat_gen(void* node_pointer, void* edge_list, void* block_partition, void* edge_to_column,
                             void* edge_to_row, void* pointer1, int num_row_windows, int num_nodes, int num_edges,
                             void* hspa, void* hind, int* return_code);

/* Replaces launch() of kernel.hmat_packed_swizzle_kernel.* -- voltrix/jit_kernels/bmat_swizzle.py:38-43 ->
 * voltrix::hmat_packed_swizzle_cuda, bmat_kernels.cuh:228-242 (kernel :151-193).  hspa_packed uint32[T*4]. */
void voltrix_launch_hmat_packed_swizzle(int num_row_windows, void* pointer1, void* hspa, void* hspa_packed,
                                        int* return_code);

/* Replaces launch() of kernel.spmm_kernel.* -- voltrix/jit_kernels/spmm.py:78-88 ->
 * voltrix::voltrix_spmm_forward_cuda, spmm_kernels.cuh:2003-2113.  input float32 [*, embedding_dim], output
 * float32 [num_nodes, embedding_dim], both row-major contiguous DEVICE buffers; every row of output
 * (including the num_nodes % 16 tail the reference leaves unwritten) is stored.  Exact fp32 products on
 * v_mfma_f32_16x16x4_f32 (the reference rounds `input` to TF32).  embedding_dim % 4 == 0.  num_edges is unused,
 * as in the reference.  The tile is chosen by voltrix_spmm_default_tile(). */
void voltrix_launch_spmm(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                         int embedding_dim, void* input, void* output, void* stream, int* return_code);

/* ---- gfx950 extensions ------------------------------------------------------------------------------------ */

/* Same as voltrix_launch_spmm with an explicit tile: fs = feature slab per wave (32/64/128), depth = LDS ring
 * depth (2..4), waves = waves per workgroup (1/2/4/8); VOLTRIX_ERR_BAD_CONFIG if not instantiated.
 * window_order: NULL, or the int32[W] schedule written by voltrix_launch_window_order (changes speed only). */
void voltrix_launch_spmm_f32_tile(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                                  int embedding_dim, void* input, void* output, int fs, int depth, int waves,
                                  void* window_order, void* out_scale, void* stream, int* return_code);

/* fp16 dense operand (BASELINE.json's headline configuration): input _Float16 [*, embedding_dim], output float32.
 * v_mfma_f32_16x16x32_f16, fp32 accumulate.  embedding_dim % 8 == 0, input 16-byte aligned. */
void voltrix_launch_spmm_f16(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                             int embedding_dim, void* input, void* output, void* stream, int* return_code);
void voltrix_launch_spmm_f16_tile(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                                  int embedding_dim, void* input, void* output, int fs, int depth, int waves,
                                  void* window_order, void* out_scale, void* stream, int* return_code);

/* fp32 features on the 16-bit matrix-core path -- the fast replacement of voltrix_launch_spmm for hosts that keep the
 * reference's fp32 contract (jit_kernels/spmm.py:53: input float32): `input` float32 [input_rows, embedding_dim] is rounded
 * to fp16 after ONE power-of-two rescale per call (voltrix_launch_cast_f32_f16_scaled below: the reference's TF32 rounding
 * keeps the same 10 mantissa bits; fp32's exponent range is preserved) into `workspace`, the product runs on
 * v_mfma_f32_16x16x32_f16 with the default tile and the epilogue undoes the scale (exact).  What voltrix.spmm does for a
 * float32 `feat`.  workspace: voltrix_spmm_f32_workspace_bytes(input_rows, embedding_dim) bytes, device, 16-byte aligned,
 * owned by the caller (input_rows = rows of `input`: num_nodes for a square adjacency).  embedding_dim % 8 == 0.
 * voltrix_launch_spmm itself keeps exact fp32 products (3.6x slower on the reddit-like headline graph). */
int64_t voltrix_spmm_f32_workspace_bytes(int64_t input_rows, int embedding_dim);
void voltrix_launch_spmm_f32_as_f16(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                                    int embedding_dim, void* input, int64_t input_rows, void* output, void* workspace,
                                    void* stream, int* return_code);

/* bfloat16 dense operand (extension; 8-bit mantissa, fp32's exponent range): input bfloat16 [*, embedding_dim], output
 * float32, v_mfma_f32_16x16x32_bf16.  Same tiles, shapes and arguments as the fp16 entry points. */
void voltrix_launch_spmm_bf16(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                              int embedding_dim, void* input, void* output, void* stream, int* return_code);
void voltrix_launch_spmm_bf16_tile(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                                   int embedding_dim, void* input, void* output, int fs, int depth, int waves,
                                   void* window_order, void* out_scale, void* stream, int* return_code);

/* Schedule / output extensions of the 16-bit-operand launches (no reference counterpart; the math is that of
 * voltrix_launch_spmm_f16_tile):
 *   atomic_out  0: every element of output is stored.  1: the result is ADDED to output with float atomics
 *               (global_atomic_add_f32): the caller zero-fills output and lets the panel kernel (accumulate = 2) add its
 *               part in any order -- two addends per element, so the sum does not depend on the order.
 *   units       NULL, or a unit table int32[U][4] = {window, phase, stride, slot} (16-byte aligned) that replaces
 *               window_order: a unit runs the stages phase, phase + stride, ... (a stage = 4 TC blocks) of its window.
 *               A long window is cut into `stride` interleaved units of bounded length that each sweep the window's whole
 *               (sorted) column range: no wave runs a window several times the usual length (the tail of the launch), and
 *               units listed by length run in step and share gathered rows through L2.  slot < 0: the unit is the
 *               whole window (stride 1) and its result goes to output; slot >= 0: the unit's [16][embedding_dim] float32
 *               tile goes to partials + slot * 16 * embedding_dim, and voltrix_launch_combine_partials sums the tiles
 *               of each cut window in unit order (deterministic) into output.  unit_ptr int32[9]: XCD x owns units
 *               [unit_ptr[x], unit_ptr[x+1]); max_units_per_xcd = the largest of those eight counts.  Every stage of
 *               every window must belong to exactly one unit.
 *   units_per_wave  1, or 2 with a unit table and fs <= 128 (several column slabs: fs = 128, slab-major order): every wave runs two consecutive units of the
 *               table, their stages alternating through one ring into two accumulator sets (same bits as 1): twice the
 *               windows sweep their sorted columns in step per CU at the same LDS and bytes in flight.  Measured beside the
 *               panel kernel on the reddit-like graph with units cut at 1.25 x the median: 1.365 -> 1.293 ms.  Ignored
 *               (= 1) where it does not apply.
 *   row_map     NULL, or int32[16 W]: row i of the handle is row row_map[i] of output (-1: a padding row, never
 *               written).  For handles built from a row-permuted CSR (locality reorder: rows that share columns grouped
 *               into the same 16-row windows -- the reference takes externally reordered graphs, bench/graph_gen.py:42-45):
 *               the product is written through the permutation, there is no un-permute pass, and because column ids are
 *               not relabelled `input` is the caller's B as it is. */
void voltrix_launch_spmm_f16_sched(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                                   int embedding_dim, void* input, void* output, int fs, int depth, int waves,
                                   void* window_order, void* out_scale, int atomic_out, void* units, void* unit_ptr,
                                   int max_units_per_xcd, void* partials, void* row_map, int units_per_wave, void* stream,
                                   int* return_code);
void voltrix_launch_spmm_bf16_sched(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int num_edges,
                                    int embedding_dim, void* input, void* output, int fs, int depth, int waves,
                                    void* window_order, void* out_scale, int atomic_out, void* units, void* unit_ptr,
                                    int max_units_per_xcd, void* partials, void* row_map, int units_per_wave, void* stream,
                                    int* return_code);
/* cuts int32[num_cuts][4] = {window, first slot, units, 0} (16-byte aligned): output rows of `window` = (accumulate ?
 * output : 0) + partials[first slot] + partials[first slot + 1] + ... in that order.  Run it on the stream of the
 * launch that wrote the partials, after the join with the panel kernel when accumulate != 0.  embedding_dim % 4 == 0.
 * row_map as for the launch that wrote the partials (NULL: identity). */
void voltrix_launch_combine_partials(void* cuts, int num_cuts, void* partials, void* output, int num_nodes,
                                     int embedding_dim, int accumulate, void* row_map, void* stream, int* return_code);

/* Unit table of a handle (the `units` / `unit_ptr` / `cuts` arguments above), built on the device from blk_offsets alone,
 * in two phases around the one host read that sizes the outputs (no reference counterpart -- its equal-work scheduler,
 * spmm_kernels.cuh:499-540, is dead code; layout and rules: voltrix/unit_table.hpp, profiles/HISTORY.md section 3.2):
 *   phase 1  voltrix_launch_unit_table_count: header int32[8] (device) = {num_units U, num_cuts C, num_slots, max units per
 *            XCD, max_stages L, top (longest unit), 0, 0}.  max_stages <= 0: L = max(8, floor(1.5 x median stages per window)),
 *            on handles of fewer than 1024 windows at most max(8, ceil(all stages / 1024)) (unit_table.hpp).
 *            workspace: voltrix_unit_table_workspace_bytes(num_nodes) bytes, device, 16-byte aligned.
 *   (caller reads the header; allocates units int32[U][4], unit_ptr int32[9], cuts int32[C][4], partials
 *    float[num_slots * 16 * embedding_dim] and voltrix_unit_table_fill_workspace_bytes(U) bytes of fill workspace)
 *   phase 2  voltrix_launch_unit_table_fill: writes units, unit_ptr, cuts (every element); same workspace, untouched since
 *            phase 1; num_units / num_cuts / top as read from the header.
 * xcd_ptr: NULL, or device int32[9] = first window of every XCD's range (xcd_ptr[0] = 0, xcd_ptr[8] = ceil(num_nodes / 16),
 * non-decreasing; the same array in both phases): ranges of equal WORK instead of equal window counts, for graphs whose
 * stages per row vary along the rows (communities); a two-level host passes 32 x the panel kernel's xcd_ptr so that both
 * kernels keep a panel's rows on one XCD.  Speed only: the SpMM gives the same bits with any ranges.
 * The table depends on blk_offsets (and xcd_ptr) only (ties between units of equal length are broken by window, then unit
 * index), so a handle always gets the same table and the SpMM the same bits. */
int64_t voltrix_unit_table_workspace_bytes(int num_nodes);
int64_t voltrix_unit_table_fill_workspace_bytes(int64_t num_units);
void voltrix_launch_unit_table_count(void* blk_offsets, int num_nodes, int max_stages, void* xcd_ptr, void* workspace,
                                     void* header, void* stream, int* return_code);
void voltrix_launch_unit_table_fill(void* blk_offsets, int num_nodes, void* xcd_ptr, void* workspace, void* fill_workspace,
                                    int num_units, int num_cuts, int top, void* units, void* unit_ptr, void* cuts,
                                    void* stream, int* return_code);

/* Stream kernel (round 5; spmm_stream_kernels.hpp): the window format walked as a STREAM OF STAGES -- the kernel for graphs of
 * short windows (the reference's low-degree evaluation graphs, bench/plot.py:8).  Replaces the reference's one-CTA-per-window
 * dispatch (voltrix_spmm_forward_cuda, spmm_kernels.cuh:2003-2113, grids :2028,2058,2089) for them: a wave owns a RUN of
 * consecutive units (whole windows; a long window's interleaved pieces), one LDS ring that never drains at a window boundary,
 * every finished window stored from the loop with 16-byte stores.  16-bit binary operand, plain stores (every row of the output
 * is written; cut windows go through `partials` and voltrix_launch_combine_partials with accumulate = 0, row_map = NULL).
 *   tables      built by the library from the handle's three tensors, two phases around the host reads that size the outputs:
 *     voltrix_launch_stream_table_count: header int32[8] (device) = {num_units U, cut windows C, partial-tile slots, bound on
 *       the number of runs, run_cost, cut_stages, 0, 0}.  run_cost <= 0: clamp((stages + windows) / 9216, 6, 48) (six runs per
 *       wave slot of the chip); cut_stages <= 0: max(run_cost, the unit table's default L).  run_cost <= 128.
 *       workspace: voltrix_stream_table_workspace_bytes(num_nodes) bytes, device, 16-byte aligned.
 *     (caller reads the header; allocates units int32[U][8], cuts int32[C][4], runs int32[bound][4], run_ptr int32[9],
 *      header2 int32[4], partials float[slots * 16 * embedding_dim], voltrix_stream_table_fill_workspace_bytes(U) bytes)
 *     voltrix_launch_stream_table_fill: writes them all; header2 = {runs R, max runs per XCD, oversized, 0} (the launch's
 *       max_runs_per_xcd); same workspace, untouched since phase 1; run_cost as read from the header: 2 .. 128, anything else is
 *       refused (kErrBadShape) -- a larger value would make runs of more than 64 units, which the kernel (one lane per unit of a
 *       run) cannot walk; `oversized` != 0 reports such a run should one ever be built: do not launch with that table.
 *     units[u] = {first TC block, end TC block of the window, 4 x stride, window, partial-tile slot or -1, stages, columns of the
 *     window's last TC block that carry an edge (0: a window without edges), 0}; runs[r] = {first unit, units (<= 64), stages,
 *     0}; run_ptr[x] .. run_ptr[x + 1] = the runs of XCD x (contiguous windows, equal cost).  The tables depend on the handle
 *     only: a handle always gets the same tables and the product the same bits.
 *   tile        (fs, depth, waves): fs 32 | 64 | 128 (the slab: embedding_dim <= 32 | <= 64 | wider), depth 2..4, waves 1 | 2;
 *               fs = 0 selects the default (fs by width, depth 2, waves 1).
 *   input_rows  rows of the dense operand (0: num_nodes); decides 32- or 64-bit row addressing and the slab launches. */
int64_t voltrix_stream_table_workspace_bytes(int num_nodes);
int64_t voltrix_stream_table_fill_workspace_bytes(int64_t num_units);
void voltrix_launch_stream_table_count(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, int run_cost,
                                       int cut_stages, void* workspace, void* header, void* stream, int* return_code);
void voltrix_launch_stream_table_fill(void* blk_offsets, int num_nodes, void* workspace, void* fill_workspace, int num_units,
                                      int num_cuts, int run_bound, int run_cost, void* units, void* cuts, void* runs,
                                      void* run_ptr, void* header2, void* stream, int* return_code);
void voltrix_launch_spmm_stream_f16(void* hspa_packed, void* hind, int num_nodes, int embedding_dim, void* input,
                                    int64_t input_rows, void* output, void* units, void* runs, void* run_ptr,
                                    int max_runs_per_xcd, void* partials, void* out_scale, int fs, int depth, int waves,
                                    int slab_policy, void* stream, int* return_code);
void voltrix_launch_spmm_stream_bf16(void* hspa_packed, void* hind, int num_nodes, int embedding_dim, void* input,
                                     int64_t input_rows, void* output, void* units, void* runs, void* run_ptr,
                                     int max_runs_per_xcd, void* partials, void* out_scale, int fs, int depth, int waves,
                                     int slab_policy, void* stream, int* return_code);

/* "Balance" schedule for a handle: order_out int32[W] (device) lists the windows of every XCD range, inside chunks of
 * `chunk` (1..4096) consecutive windows, by descending TC-block count, so that co-resident waves sweep their sorted
 * columns at a similar pace and share gathered rows through L2.  Depends on blk_offsets only; results of the SpMM are
 * bit-identical with or without it. */
void voltrix_launch_window_order(void* blk_offsets, int num_nodes, int chunk, void* order_out, void* stream,
                                 int* return_code);

/* Panel kernel: the shared-column half of the two-level condensed format (spmm_panel_kernels.hpp; no reference
 * counterpart).  A panel = waves * row_blocks * 16 consecutive rows; per panel the plan lists the columns referenced by
 * several of its rows, cut into k-steps of 32:
 *   panel_ptr  int32 [NP+1]                  first k-step of every panel, S = panel_ptr[NP]
 *   panel_cols int32 [32 * (S + 2)]          row of `input` per (k-step, k); unused slots repeat a real column; 2 k-steps
 *                                            of padding (valid row ids) at the end
 *   panel_bits uint32 [(S + 1) * waves * 64] word (k-step, wave v, lane 16 g + R), bit 16 (c & 1) + 4 j + (c >> 1) <=>
 *                                            edge (row 16 (row_blocks v + j) + R of the panel, column 8 g + c of the
 *                                            k-step), j < row_blocks, c < 8
 *   panel_order int32 [NP] or NULL           launch position -> panel
 * output [num_nodes, embedding_dim] float32: accumulate == 0 overwrites every row; 1 adds onto it (read-add-store: output
 * already holds the window kernel's result for the remaining edges); 2 adds with float atomics onto a pre-zeroed output that
 * the window kernel (atomic_out = 1) adds to as well, in any order.  input _Float16 (bfloat16 for _bf16) [*, embedding_dim], 16-byte
 * aligned, embedding_dim % 8 == 0.  Tile: fs in {32,64,128}, depth = ring slots, ksteps per ring slot in {1,2}, or
 * VOLTRIX_PANEL_KSTEPS_PIPELINED (17): one k-step per slot walked by the software-pipelined loop (fragment reads of one half of
 * the column slots under the MFMAs of the other half; same bits as ksteps = 1; what the library uses for 8 x 2 workgroups =
 * 256-row panels, the shape it picks when the panel kernel is the critical path -- voltrix/hybrid.py PANEL_DOMINATED_RATIO);
 * VOLTRIX_ERR_BAD_CONFIG if the combination is not instantiated; VOLTRIX_ERR_BAD_SHAPE for accumulate outside 0..2.
 * out_scale as for voltrix_launch_spmm_f16_tile.
 * input_rows = rows of `input` (0: num_nodes, a square adjacency); slab_policy: how an operand wider than the tile's fs is
 * launched -- -1 (VOLTRIX_SLAB_AUTO) one launch per 256-byte group of column slabs when such a group of `input`
 * (input_rows x 256 bytes) fits the 256 MiB Infinity Cache, else one grid over all slabs; 0 always one grid; 1 always the
 * launches.  Same bits either way.
 * xcd_ptr / max_panels_per_xcd: NULL / 0, or device int32[9] = first launch POSITION of every XCD's range (xcd_ptr[8] =
 * NP) and the length of the longest range (sizes the grid): ranges of equal work instead of ceil(NP / 8) positions each
 * (graphs with community structure: the k-steps per panel vary by community); panel_order must have been built with the same
 * xcd_ptr.  Speed only.
 * MEMORY REQUIREMENT of the atomic forms (accumulate == 2 here, atomic_out != 0 in voltrix_launch_spmm_*_sched): they use the
 * hardware's no-return global_atomic_add_f32, which is only defined on ordinary (coarse-grained) device memory -- hipMalloc,
 * torch's allocator.  On fine-grained or host-mapped output (hipHostMalloc, hipMallocManaged with fine-grained coherence) the
 * adds can be lost silently: give such outputs a device-memory staging buffer, or use accumulate 0 / 1. */
#define VOLTRIX_PANEL_KSTEPS_PIPELINED 17
void voltrix_launch_spmm_panel_f16(void* panel_ptr, void* panel_cols, void* panel_bits, void* panel_order, void* xcd_ptr,
                                   int max_panels_per_xcd, int num_nodes, int embedding_dim, void* input, int64_t input_rows, void* output,
                                   int accumulate, int fs, int depth, int waves, int row_blocks, int ksteps,
                                   int slab_policy, void* out_scale, void* stream, int* return_code);
void voltrix_launch_spmm_panel_bf16(void* panel_ptr, void* panel_cols, void* panel_bits, void* panel_order, void* xcd_ptr,
                                    int max_panels_per_xcd, int num_nodes, int embedding_dim, void* input, int64_t input_rows, void* output,
                                    int accumulate, int fs, int depth, int waves, int row_blocks, int ksteps,
                                    int slab_policy, void* out_scale, void* stream, int* return_code);

/* The same kernel over PARTS of panels (round 4; voltrix/hybrid.py::panel_parts builds the table, no reference counterpart).
 * A panel's k-step list is walked by ONE workgroup, so the longest panel is a critical path (a panel inside a dense
 * community carries 5-20 x the k-steps of the median panel).  parts int32 [num_parts][4] = {panel, first k-step inside the
 * panel, k-steps, slot} replaces panel_order: launch position -> a piece of at most a bounded number of k-steps.  slot < 0:
 * the panel is whole and its tile goes to `output` per `accumulate`, exactly as above.  slot >= 0: the panel is cut; every
 * piece STORES its tile to partials[slot] (float32 [slots][16 waves row_blocks][embedding_dim], per-call scratch) and
 * voltrix_launch_combine_panel_partials -- on the same stream, after this launch (and, for accumulate == 2, after the window
 * kernel has been joined) -- adds the pieces to `output` in slot order: a fixed order, whatever the pieces' timing.
 * xcd_ptr / max_parts_per_xcd as above, over part positions (NULL / 0: ranges of ceil(num_parts / 8)).
 *   cuts int32 [num_cuts][4] = {panel, first slot, pieces, 0};  panel_rows = 16 waves row_blocks;  accumulate 0: output rows
 *   of the cut panels = the sum; != 0: added onto output (plain read-add-store). */
void voltrix_launch_spmm_panel_parts_f16(void* panel_ptr, void* panel_cols, void* panel_bits, void* parts, int num_parts,
                                         void* xcd_ptr, int max_parts_per_xcd, void* partials, int num_nodes, int embedding_dim,
                                         void* input, int64_t input_rows, void* output, int accumulate, int fs, int depth,
                                         int waves, int row_blocks, int ksteps, int slab_policy, void* out_scale, void* stream,
                                         int* return_code);
void voltrix_launch_spmm_panel_parts_bf16(void* panel_ptr, void* panel_cols, void* panel_bits, void* parts, int num_parts,
                                          void* xcd_ptr, int max_parts_per_xcd, void* partials, int num_nodes, int embedding_dim,
                                          void* input, int64_t input_rows, void* output, int accumulate, int fs, int depth,
                                          int waves, int row_blocks, int ksteps, int slab_policy, void* out_scale, void* stream,
                                          int* return_code);
void voltrix_launch_combine_panel_partials(void* cuts, int num_cuts, void* partials, void* output, int num_nodes,
                                           int embedding_dim, int panel_rows, int accumulate, void* stream, int* return_code);

/* Builders of the two schedules above, on the device (round 4; voltrix/schedule_tables.hpp; the Python host's
 * voltrix/schedule.py::split_equal_work and voltrix/hybrid.py::panel_parts are their torch-tensor restatements).
 * XCD ranges: xcd_ptr int32[9] <- b[0] = 0 <= ... <= b[8] = n such that the eight ranges of the items' work have about equal
 * sums (boundary = the index whose prefix sum is nearest to ceil(total x / 8), rounded up to `align`; total 0: ceil(n / 8)
 * items each).  _of_work: work int32[num_items]; _of_windows: the stages of a handle's windows (work of window w =
 * ceil(TC blocks / 4)), n = ceil(num_nodes / 16); _of_panels: work of panel p = round(kstep_cost_x10 / 10 x its k-steps) +
 * the stages of its panel_rows / 16 windows in resid_blk_offsets (the Python host passes 66: a k-step costs a CU about 6.6
 * residual stages), and window_xcd_ptr int32[9] (or NULL) <- the same ranges in windows, for the residual's unit table.
 * One workgroup each; stream-ordered, no host read. */
void voltrix_launch_xcd_ranges_of_work(void* work, int num_items, int align, void* xcd_ptr, void* stream, int* return_code);
void voltrix_launch_xcd_ranges_of_windows(void* blk_offsets, int num_nodes, int align, void* xcd_ptr, void* stream,
                                          int* return_code);
void voltrix_launch_xcd_ranges_of_panels(void* panel_ptr, void* resid_blk_offsets, int num_nodes, int panel_rows,
                                         int kstep_cost_x10, void* xcd_ptr, void* window_xcd_ptr, void* stream,
                                         int* return_code);
/* Piece table, two phases around the host read that sizes it.  count: header int32[8] <- {pieces, cut panels, partial-tile
 * slots, pieces in the longest XCD range, cap, 0, 0, 0}; fill (same panel_ptr / cap / panel_xcd_ptr / workspace, untouched in
 * between): parts int32[pieces][4], part_xcd_ptr int32[9], cuts int32[max(1, cut panels)][4] as described above -- panels
 * of more than `cap` k-steps in ceil(k-steps / cap) contiguous pieces of nearly equal length; per XCD range (panel_xcd_ptr in
 * panel units, or NULL = ceil(num_panels / 8) panels each) longest first, ties by (panel, piece).  workspace: 16-byte aligned,
 * voltrix_panel_parts_workspace_bytes(num_panels) bytes. */
int64_t voltrix_panel_parts_workspace_bytes(int num_panels);
void voltrix_launch_panel_parts_count(void* panel_ptr, int num_panels, int cap, void* panel_xcd_ptr, void* workspace,
                                      void* header, void* stream, int* return_code);
void voltrix_launch_panel_parts_fill(void* panel_ptr, int num_panels, int cap, void* panel_xcd_ptr, void* workspace,
                                     void* parts, void* part_xcd_ptr, void* cuts, void* stream, int* return_code);

/* The two-level format in ONE launch (spmm_fused_kernels.hpp; round 3, rebuilt in round 4).  One 256-thread workgroup per
 * 512-row panel -- four waves, one per SIMD, eight 16-row blocks each; the plan keeps its waves = 8 x row_blocks = 4 layout --
 * computes the whole product for its rows: the shared columns from the panel plan (arrays as for
 * voltrix_launch_spmm_panel_f16) and the residual edges from per-wave streams of stage records, into the same accumulators;
 * output [num_nodes, embedding_dim] float32 is written once with plain stores (every row; no zero fill, no atomics, no
 * second stream, no combine pass; the summation order is fixed, so results are run-to-run identical).
 *   wave_ptr int32 [4 NP + 1]    first record of (panel p, wave v) at index 4 p + v; wave v owns windows 32 p + 8 v + j, j < 8
 *   records  uint32 [R + 1][64]  16-byte aligned; one record = one stage (4 TC blocks = 32 condensed columns) of ONE of the
 *                                wave's windows: words 0..31 rows of `input` (unused columns repeat a real one), 32..47 the
 *                                stage's 16 bitmap words (hspa_packed order), word 48 = j; a wave's records are sorted by
 *                                their first column; one record of padding at the end.  Built from the block-format handle of
 *                                the residual matrix by voltrix_launch_fused_records_* below.
 * Tile: fs in {32,64,128}, depth = slots of the shared panel ring (3, or 4 below fs 128); VOLTRIX_ERR_BAD_CONFIG otherwise.
 * pace_blocks: 0 / 1 = none; n > 1 (one column slab only) = the workgroups that share an XCD and a dispatch generation wait
 * for each other at n points of their column sweep (bounded polls on counters the library keeps: advisory, the result never
 * depends on it) so that an XCD's resident rows sweep the sorted columns together and share gathered rows through its L2.
 * input / out_scale / xcd_ptr / max_panels_per_xcd as for voltrix_launch_spmm_panel_f16. */
void voltrix_launch_spmm_fused_f16(void* panel_ptr, void* panel_cols, void* panel_bits, void* panel_order, void* xcd_ptr,
                                   int max_panels_per_xcd, void* wave_ptr, void* records, int num_nodes, int embedding_dim, void* input,
                                   void* output, int fs, int depth, int pace_blocks, void* out_scale, void* stream,
                                   int* return_code);
void voltrix_launch_spmm_fused_bf16(void* panel_ptr, void* panel_cols, void* panel_bits, void* panel_order, void* xcd_ptr,
                                    int max_panels_per_xcd, void* wave_ptr, void* records, int num_nodes, int embedding_dim, void* input,
                                    void* output, int fs, int depth, int pace_blocks, void* out_scale, void* stream,
                                   int* return_code);

/* Builder of the stage records above (fused_plan.hpp): block-format handle of the RESIDUAL matrix (the handle of
 * resid_node_pointer / resid_edge_list from the plan builder below, through voltrix_launch_csr_window_count / _fill) ->
 * (wave_ptr, records), in two phases because the caller owns every buffer:
 *   phase 1  voltrix_launch_fused_records_count: wave_ptr int32 [4 NP + 1] (NP = ceil(num_nodes / 512)); workspace:
 *            voltrix_fused_records_workspace_bytes(num_nodes) bytes, device, 16-byte aligned
 *   (caller reads R = wave_ptr[4 NP] and allocates records uint32 [(R + 1) * 64], 16-byte aligned)
 *   phase 2  voltrix_launch_fused_records_fill: every word of records is written (the padding record is zero).
 * Once the records exist the residual handle is no longer needed by voltrix_launch_spmm_fused_*. */
/* How the one-launch kernel splits a 512-row panel: waves per workgroup (4) x 16-row blocks per wave (8).  wave_ptr has
 * waves * NP + 1 entries; hosts size it from THIS call, not from a constant of their own. */
void voltrix_fused_panel_geometry(int* waves, int* row_blocks);
int64_t voltrix_fused_records_workspace_bytes(int num_nodes);
void voltrix_launch_fused_records_count(void* blk_offsets, void* hspa_packed, int num_nodes, void* workspace, void* wave_ptr,
                                        void* stream, int* return_code);
void voltrix_launch_fused_records_fill(void* blk_offsets, void* hspa_packed, void* hind, int num_nodes, void* wave_ptr,
                                       int64_t num_records, void* records, void* stream, int* return_code);

/* Builder of the panel plan (panel_plan.hpp): CSR on the DEVICE (rows sorted, duplicate-free, ids in [0, num_cols),
 * num_cols <= 2^22) -> residual CSR + plan, in two phases because the caller owns every buffer:
 *   phase 1  voltrix_launch_panel_plan_count: panel_ptr int32[NP+1], resid_node_pointer int32[num_nodes+1], status[1];
 *            workspace: voltrix_panel_plan_workspace_bytes(...) bytes, device, 16-byte aligned
 *   (caller reads S = panel_ptr[NP], E_r = resid_node_pointer[num_nodes] and status[0] -- the number of input
 *    violations, must be 0 -- and allocates resid_edge_list int32[E_r], panel_cols int32[32 (S + 2)],
 *    panel_bits uint32[(S + 1) * waves * 64])
 *   phase 2  voltrix_launch_panel_plan_fill: same arguments and workspace; every output element is written.
 * A column is shared in a panel when >= tau (1..65535) of the panel's rows reference it; waves in {4, 8}, row_blocks in
 * {2, 4}.  VOLTRIX_ERR_BAD_CONFIG: column universe above 2^22 (the two-level format is not built for it). */
int64_t voltrix_panel_plan_workspace_bytes(int num_nodes, int waves, int row_blocks);
void voltrix_launch_panel_plan_count(void* node_pointer, void* edge_list, int num_nodes, int num_cols, int64_t num_edges,
                                     int waves, int row_blocks, int tau, void* workspace, void* panel_ptr,
                                     void* resid_node_pointer, void* status, void* stream, int* return_code);
void voltrix_launch_panel_plan_fill(void* node_pointer, void* edge_list, int num_nodes, int num_cols, int64_t num_edges,
                                    int waves, int row_blocks, int tau, void* workspace, void* panel_ptr,
                                    void* resid_node_pointer, int64_t total_ksteps, void* resid_edge_list,
                                    void* panel_cols, void* panel_bits, void* stream, int* return_code);

/* `panel_order` of the panel launches: order_out int32[num_panels] (device), position -> panel; inside every XCD's range of
 * positions (xcd_ptr int32[9] on the device, or NULL: ceil(num_panels / 8) each), groups of `group` consecutive panels (neighbours share their band columns: side by side
 * they share gathered rows through L2), the groups with the most k-steps first (ties by index), natural order inside a
 * group; group = 1: plain longest-first (what the Python host uses: groups of 4 gained 3 % on the bare kernel pair, nothing
 * through the operator).  Speed only. */
void voltrix_launch_panel_order(void* panel_ptr, int num_panels, int group, void* xcd_ptr, void* order_out, void* stream,
                                int* return_code);

/* Default tile for a feature width; is_f16 selects the operand type.  Always succeeds. */
void voltrix_spmm_default_tile(int embedding_dim, int is_f16, int* fs, int* depth, int* waves);

/* Number of instantiated tiles and the i-th one (for autotuners that enumerate the ahead-of-time space). */
int voltrix_spmm_num_tiles(int is_f16);
void voltrix_spmm_tile_at(int is_f16, int index, int* fs, int* depth, int* waves);

/* fp32 -> fp16 cast of the dense operand into a caller-provided buffer (count % 8 == 0). */
void voltrix_launch_cast_f32_f16(void* src, void* dst, int64_t count, void* stream, int* return_code);

/* Range-safe variant (what voltrix.spmm uses for fp32 features): dst = fp16(src * 2^-e) with one power-of-two scale
 * per call, e = exponent(max |src|) - 14, so fp32 magnitudes beyond fp16's range neither overflow nor flush while the
 * 10-bit mantissa (= the reference's TF32 rounding, spmm_kernels.cuh:1671) is kept.  scale: device float[2], 8-byte
 * aligned; scale[0] <- 2^e, to be passed as `out_scale` of voltrix_launch_spmm_f16_tile (multiplied into every output
 * element in the epilogue; exact).  No host sync.  Inf / NaN in src: scale 1, they propagate as in fp32. */
void voltrix_launch_cast_f32_f16_scaled(void* src, void* dst, int64_t count, void* scale, void* stream,
                                        int* return_code);

/* CSR row-gather kernel (round 6; spmm_csr_kernels.hpp): output[i, :] = sum over the entries of row i of input[indices[e], :],
 * straight from a DEVICE CSR (int32 indptr[num_rows + 1], indices[nnz]; binary A: entries count as 1, a duplicate (row, col) entry
 * counts TWICE here -- callers with duplicates keep to the block format, which counts it once like the reference's bitmaps,
 * bmat_kernels.cuh:100-103).  dtype 0 fp32 / 1 fp16 / 2 bfloat16 rows of `input` (embedding_dim a multiple of 16 bytes), fp32
 * `output` [num_rows, embedding_dim], every row written; exact products, fp32 sum in entry order.  No workspace, no tables.
 * xcd_ranges: 0 = consecutive row groups go round the XCDs (default), 1 = every XCD owns a contiguous eighth of the rows.  The
 * operator takes it for handles of short windows where it measured faster than the block-format kernels (fp32 features; wide
 * operands): voltrix/spmm/spmm.py.  No reference counterpart (the reference has the block format only). */
void voltrix_launch_spmm_csr_rows(void* indptr, void* indices, int num_rows, int embedding_dim, void* input, int dtype, void* output,
                                  int xcd_ranges, void* stream, int* return_code);

/* The same kernel with edge values: output = csr(values) * input, values = device float[nnz] in CSR order (duplicate entries ADD, as in
 * torch.sparse.mm).  One fused multiply-add per element in fp32: with fp32 rows the weighted product is exact up to the rounding of
 * the sum (deg * 2^-23 (|A| |B|)) -- the block format's value planes are 16-bit.  Values can change between calls at no cost (nothing is
 * preprocessed).  voltrix/weighted.py takes it for general values on handles of short windows where it measured faster, and for
 * VOLTRIX_FP32_MODE=exact.  No reference counterpart (the reference has no edge values: spmm_kernels.cuh:1632-1644). */
void voltrix_launch_spmm_csr_rows_weighted(void* indptr, void* indices, void* values, int num_rows, int embedding_dim, void* input, int dtype,
                                           void* output, int xcd_ranges, void* stream, int* return_code);

/* New edge values on a fixed pattern: plane[slots[e]] = T(values[e]) for e < count; values device float[count], slots device int64[count]
 * (the element of the flat value plane [T * 128] every CSR entry lands on: voltrix/weighted.py::edge_slots; the entries of a
 * duplicate-free pattern own their elements), plane of dtype 0 fp32 / 1 fp16 / 2 bfloat16 (round to nearest even).  One pass instead
 * of rebuilding the plane (sorts and searches over all edges).  No reference counterpart. */
void voltrix_launch_scatter_values(void* values, void* slots, void* plane, int64_t count, int dtype, void* stream, int* return_code);

/* Rows of a dense row-major matrix times a per-row factor: dst[i, :] = T(float(src[i, :]) * scale[i]); dst may be src.
 * dtype 0 fp32 / 1 fp16 / 2 bfloat16; a row (num_feats elements) must be a multiple of 16 bytes; scale: device float[rows].
 * What edge values of the form v_ij = r_i * c_j cost on top of the binary product (voltrix/weighted.py: B's rows times c before,
 * C's rows times r after -- the normalised adjacencies of GCN / mean aggregation); the reference has no edge values at all
 * (spmm_kernels.cuh:1632-1644: bits -> 1.0). */
void voltrix_launch_scale_rows(void* src, void* scale, void* dst, int64_t rows, int num_feats, int dtype, void* stream,
                               int* return_code);

/* Fused GPU preprocess: CSR on the DEVICE -> (pointer1, hspa_packed, hind) without the reference's host
 * preprocess, the O(TCb*E) rescan or the 512-byte/TC-block fp32 `hspa` intermediate.  Two phases because the
 * caller owns every buffer and T is data dependent:
 *   phase 1  voltrix_launch_csr_window_count: block_partition[W], pointer1[W+1], status[1] (device int32)
 *            workspace: voltrix_csr_preprocess_workspace_bytes(num_nodes, num_cols, num_edges, path) bytes, device, 16-B aligned
 *   (caller reads T = pointer1[W] and status[0], allocates hspa_packed uint32[4T] and hind int32[8T])
 *   phase 2  voltrix_launch_csr_fill: writes hspa_packed and hind (every word), same workspace, same num_cols.
 * num_cols = the column universe: every id in edge_list lies in [0, num_cols) (square adjacency: num_nodes; a row shard
 * whose ids index a gathered B: the rows of that buffer).  num_cols <= 0 = unknown.  The condensed-column ranks come
 * from an LDS bitmap + popcounts when the universe fits LDS (num_cols <= 2^19) and is small next to a window's edge
 * list, otherwise from a per-window sort (needs the 4-byte-per-edge key workspace); universes of 2 .. 16 bitmap ranges
 * (up to 2^23 columns) sort the windows up to 8192 edges and send only the bigger ones through the bitmap kernels, one
 * sweep per range ("mixed").  path = -1 (VOLTRIX_CSR_AUTO): that rule; 0 sort / 1 bitmap / 2 mixed force a path where it is
 * applicable (the same value in all three calls; how the tests reach every path on one graph).  status[0] = number of edges with an id outside [0, num_cols) (outside
 * [0, 2^28) when num_cols <= 0): the handle is valid only if it is 0 (callers retry with num_cols = 0 or reject).
 * Output is bit-identical to preprocess + hmat_gen + hmat_packed_swizzle on every path. */
int64_t voltrix_csr_preprocess_workspace_bytes(int num_nodes, int num_cols, int64_t num_edges, int path);
void voltrix_launch_csr_window_count(void* node_pointer, void* edge_list, int num_nodes, int num_cols, int64_t num_edges,
                                     int path, void* workspace, void* block_partition, void* pointer1, void* status,
                                     void* stream, int* return_code);
void voltrix_launch_csr_fill(void* node_pointer, void* edge_list, int num_nodes, int num_cols, int64_t num_edges,
                             int path, void* workspace, void* pointer1, void* hspa_packed, void* hind, void* stream,
                             int* return_code);

/* Cuthill-McKee row order on the device (locality reorder; no reference counterpart -- the
 * reference reads externally reordered <name>.reorder.npz files, bench/graph_gen.py:42-45, bench/bench_all.py:120-129).
 * Specification (voltrix/reorder_kernels.hpp; oracle/oracle_np.py::cm_order restates it): nodes = rows u < num_nodes;
 * u, v are neighbours when A[u, v] or A[v, u] is stored (columns >= num_nodes are not nodes); deg(u) = entries of row u of
 * A + entries of row u of A^T; tie(u) = position of u in the stable sort by deg.  A component is searched breadth first
 * from `start`; inside level d the nodes are ordered by (rank of the earliest-ranked neighbour of level d - 1, tie).
 * The result is a function of the CSR alone -- no race decides anything -- whatever the order inside A^T's rows.
 *   voltrix_launch_csr_transpose: CSR of A^T (t_indptr int32[num_cols + 1], t_indices int32[num_edges]; rows sorted,
 *       duplicates kept; entries with a column id outside [0, num_cols) left out) -- the search walks row u of A and row u of
 *       A^T, and the backward pass of the SpMM multiplies with it (voltrix/autograd.py).  Row ids expanded per entry, one
 *       stable radix sort by column, row pointers by binary search: no atomics.  workspace:
 *       voltrix_csr_transpose_workspace_bytes(num_edges) bytes, device, 16-byte aligned.
 *   voltrix_launch_bfs_seed: level[start] = 0 (level int32[num_nodes], -1 = unvisited, kept by the caller across
 *       components), queue[0] = start (queue int32[num_nodes]), ctrl int32[8] = {head, tail, appended, depth, done, ...},
 *       level_off int32[num_nodes + 2] (level_off[d] = queue position of level d's first node).
 *   voltrix_launch_bfs_levels: one single-workgroup launch that walks every level while the frontier stays <= 2048 nodes,
 *       then `wide_levels` whole-chip levels.  The caller reads ctrl (its sync) and calls again until ctrl[4] != 0; then
 *       levels = ctrl[3] + 1, nodes of the component = ctrl[1] = level_off[levels], queue[0 .. nodes) = the component level
 *       by level (order inside a level: arbitrary so far).
 *   voltrix_launch_cm_rank: orders every level's queue segment (levels <= 1024 nodes: one workgroup walks runs of them,
 *       LDS bitonic; larger: keys + radix sort, workspace voltrix_cm_rank_workspace_bytes(largest level above 1024) bytes)
 *       and writes rank[v] = base + position (rank int32[num_nodes]).  level_off_host: HOST copy of level_off[0 .. levels].
 *       tie int32[num_nodes].  Afterwards queue[0 .. nodes) is the component's order. */
int64_t voltrix_csr_transpose_workspace_bytes(int64_t num_edges);
void voltrix_launch_csr_transpose(void* indptr, void* indices, int num_rows, int num_cols, int64_t num_edges, void* workspace,
                                  void* t_indptr, void* t_indices, void* stream, int* return_code);
void voltrix_launch_bfs_seed(int start, int num_nodes, void* level, void* queue, void* ctrl, void* level_off, void* stream,
                             int* return_code);
void voltrix_launch_bfs_levels(void* indptr, void* indices, void* t_indptr, void* t_indices, int num_nodes, int t_rows,
                               void* level, void* queue, void* ctrl, void* level_off, int wide_levels, void* stream,
                               int* return_code);
int64_t voltrix_cm_rank_workspace_bytes(int64_t max_level);
void voltrix_launch_cm_rank(void* indptr, void* indices, void* t_indptr, void* t_indices, int num_nodes, int t_rows,
                            void* level, void* rank, void* tie, void* queue, void* level_off, const int* level_off_host,
                            int num_levels, int base, void* workspace, void* stream, int* return_code);

/* out = inv(chol(gram + eps trace(gram) I))^T for a k x k symmetric positive semi-definite gram (float32, row-major,
 * k <= 64), float32 [k][k]: the small factor of a Cholesky QR (X <- X out has orthonormal columns), on the device so that
 * the spectral row order's subspace iteration (voltrix/reorder.py) has no host sync per step.  One workgroup, double
 * arithmetic.  VOLTRIX_ERR_BAD_SHAPE for k outside 1..64. */
void voltrix_launch_chol_inv_transposed(void* gram, int k, double eps, void* out, void* stream, int* return_code);

#ifdef __cplusplus
}
#endif
#endif /* VOLTRIX_CAPI_H_ */
