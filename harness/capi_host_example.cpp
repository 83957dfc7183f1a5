// A host WITHOUT torch on libvoltrix_hip.so (include/voltrix_capi.h): plain hipMalloc buffers, the two-phase builders,
// every launch on one hipStream_t pair -- the whole Route-B chain of INTEGRATION.md in ~250 lines of C++:
//
//   CSR (device) -> csr_window_count / csr_fill (the handle) -> unit_table_count / _fill -> spmm_f16_sched + combine_partials
//                -> panel_plan_count / _fill + residual handle + panel_order -> zero C; panel kernel || window kernel; combine
//                -> fused_records_count / _fill -> spmm_fused (the same two-level product as ONE launch)
//                -> csr_transpose -> bfs_seed / bfs_levels / cm_rank (the Cuthill-McKee row order of the locality reorder)
//
// and checks all three products against a plain CPU loop over the CSR (fp16-rounded B, double accumulation) and the row order
// against a CPU search.
//   build:  hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include harness/capi_host_example.cpp \
//                 -L voltrix-spmm_amd/lib -lvoltrix_hip -Wl,-rpath,$PWD/voltrix-spmm_amd/lib -o capi_host_example
//   run:    ./capi_host_example [num_nodes] [mean_degree] [embedding_dim]      (exit code 0 = everything agrees with the CPU)
// tests/test_gpu_capi_host.py builds and runs it on the GPU box.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "voltrix_capi.h"

#define HIP_OK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      std::exit(2);                                                                        \
    }                                                                                      \
  } while (0)
#define RC_OK(call)                                                         \
  do {                                                                      \
    int rc_ = -1;                                                           \
    call;                                                                   \
    if (rc_ != 0) {                                                         \
      std::fprintf(stderr, "%s:%d return code %d\n", __FILE__, __LINE__, rc_); \
      std::exit(3);                                                         \
    }                                                                       \
  } while (0)

template <class T>
static T* dev_alloc(size_t count) {
  void* p = nullptr;
  HIP_OK(hipMalloc(&p, std::max<size_t>(count, 4) * sizeof(T)));
  return static_cast<T*>(p);
}
template <class T>
static std::vector<T> to_host(const T* d, size_t count) {
  std::vector<T> h(count);
  if (count) HIP_OK(hipMemcpy(h.data(), d, count * sizeof(T), hipMemcpyDeviceToHost));
  return h;
}

struct Handle {  // the reference's three tensors
  int* blk_offsets;
  uint32_t* hspa_packed;
  int* hind;
  int total_blocks;
};

static Handle preprocess(const int* d_indptr, const int* d_indices, int n, int num_cols, int64_t e, hipStream_t s) {
  const int W = (n + VOLTRIX_BLK_H - 1) / VOLTRIX_BLK_H;
  void* ws = dev_alloc<char>((size_t)voltrix_csr_preprocess_workspace_bytes(n, num_cols, e, VOLTRIX_CSR_AUTO));
  int* block_partition = dev_alloc<int>(W);
  int* status = dev_alloc<int>(1);
  Handle h{};
  h.blk_offsets = dev_alloc<int>(W + 1);
  RC_OK(voltrix_launch_csr_window_count((void*)d_indptr, (void*)d_indices, n, num_cols, e, VOLTRIX_CSR_AUTO, ws, block_partition,
                                        h.blk_offsets, status, s, &rc_));
  int bad = 0;
  HIP_OK(hipMemcpyAsync(&h.total_blocks, h.blk_offsets + W, sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_OK(hipMemcpyAsync(&bad, status, sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));  // the one host read that sizes the handle
  if (bad) {
    std::fprintf(stderr, "column ids outside the universe: %d\n", bad);
    std::exit(4);
  }
  h.hspa_packed = dev_alloc<uint32_t>(4 * (size_t)h.total_blocks);
  h.hind = dev_alloc<int>(8 * (size_t)h.total_blocks);
  RC_OK(voltrix_launch_csr_fill((void*)d_indptr, (void*)d_indices, n, num_cols, e, VOLTRIX_CSR_AUTO, ws, h.blk_offsets, h.hspa_packed,
                                h.hind, s, &rc_));
  HIP_OK(hipStreamSynchronize(s));
  HIP_OK(hipFree(ws));
  HIP_OK(hipFree(block_partition));
  HIP_OK(hipFree(status));
  return h;
}

struct UnitTable {
  int *units, *unit_ptr, *cuts;
  int header[8];
};

static UnitTable build_unit_table(const Handle& h, int n, hipStream_t s, int* xcd_ptr = nullptr /* device int32[9] */) {
  UnitTable t{};
  void* ws = dev_alloc<char>((size_t)voltrix_unit_table_workspace_bytes(n));
  int* d_header = dev_alloc<int>(8);
  RC_OK(voltrix_launch_unit_table_count(h.blk_offsets, n, /*max_stages=*/0, xcd_ptr, ws, d_header, s, &rc_));
  HIP_OK(hipMemcpyAsync(t.header, d_header, sizeof(t.header), hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  const int num_units = t.header[0], num_cuts = t.header[1], top = t.header[5];
  t.units = dev_alloc<int>(4 * (size_t)num_units);
  t.cuts = dev_alloc<int>(4 * (size_t)num_cuts);
  t.unit_ptr = dev_alloc<int>(9);
  void* fill_ws = dev_alloc<char>((size_t)voltrix_unit_table_fill_workspace_bytes(num_units));
  RC_OK(voltrix_launch_unit_table_fill(h.blk_offsets, n, xcd_ptr, ws, fill_ws, num_units, num_cuts, top, t.units, t.unit_ptr,
                                       t.cuts, s, &rc_));
  HIP_OK(hipStreamSynchronize(s));
  HIP_OK(hipFree(ws));
  HIP_OK(hipFree(fill_ws));
  HIP_OK(hipFree(d_header));
  return t;
}

static void window_spmm(const Handle& h, const UnitTable& t, int n, int num_edges, int f, const void* d_b, float* d_c,
                        int atomic_out, float* partials, hipStream_t s) {
  int fs, depth, waves;
  voltrix_spmm_default_tile(f, 1, &fs, &depth, &waves);
  RC_OK(voltrix_launch_spmm_f16_sched(h.blk_offsets, h.hspa_packed, h.hind, n, num_edges, f, (void*)d_b, d_c, fs, depth,
                                      waves, /*window_order=*/nullptr, /*out_scale=*/nullptr, atomic_out, t.units,
                                      t.unit_ptr, t.header[3], partials, /*row_map=*/nullptr, /*units_per_wave=*/1, s, &rc_));
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 6000;
  const int mean_degree = argc > 2 ? std::atoi(argv[2]) : 300;
  const int f = argc > 3 ? std::atoi(argv[3]) : 128;
  if (voltrix_abi_version() != VOLTRIX_ABI_VERSION) return 5;

  // CSR: a third of a row's edges inside a band around the diagonal (shared columns for the panels), the rest anywhere;
  // sorted, duplicate-free rows; a few empty rows and a few hubs (their windows are cut by the unit table)
  std::mt19937 rng(7);
  std::vector<int> indptr(n + 1, 0), indices;
  std::vector<int> row;
  for (int r = 0; r < n; ++r) {
    row.clear();
    const int deg = (r % 97 == 0) ? 0 : (r % 501 == 7 ? 8 * mean_degree : (int)(rng() % (2 * mean_degree)));  // hubs: cut windows
    for (int k = 0; k < deg; ++k) {
      const int c = (k % 3) ? (int)(rng() % n) : (int)(((long long)r + (long long)(rng() % 1024) - 512 + n) % n);
      row.push_back(c);
    }
    std::sort(row.begin(), row.end());
    row.erase(std::unique(row.begin(), row.end()), row.end());
    indices.insert(indices.end(), row.begin(), row.end());
    indptr[r + 1] = (int)indices.size();
  }
  const int64_t e = (int64_t)indices.size();
  std::vector<__half> b((size_t)n * f);
  std::uniform_real_distribution<float> dist(-1.f, 1.f);
  for (auto& x : b) x = __float2half(dist(rng));

  hipStream_t s_main, s_side;
  HIP_OK(hipStreamCreate(&s_main));
  HIP_OK(hipStreamCreate(&s_side));
  int* d_indptr = dev_alloc<int>(n + 1);
  int* d_indices = dev_alloc<int>(e);
  __half* d_b = dev_alloc<__half>((size_t)n * f);
  float* d_c = dev_alloc<float>((size_t)n * f);
  HIP_OK(hipMemcpy(d_indptr, indptr.data(), (n + 1) * sizeof(int), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_indices, indices.data(), e * sizeof(int), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_b, b.data(), b.size() * sizeof(__half), hipMemcpyHostToDevice));

  // CPU reference on the same fp16 operand
  std::vector<double> ref((size_t)n * f, 0.0);
  for (int r = 0; r < n; ++r)
    for (int p = indptr[r]; p < indptr[r + 1]; ++p)
      for (int k = 0; k < f; ++k) ref[(size_t)r * f + k] += (double)__half2float(b[(size_t)indices[p] * f + k]);
  auto max_rel_err = [&](const std::vector<float>& got) {
    double num = 0.0, den = 0.0;
    for (size_t i = 0; i < ref.size(); ++i) {
      num += (got[i] - ref[i]) * (got[i] - ref[i]);
      den += ref[i] * ref[i];
    }
    return std::sqrt(num / std::max(den, 1e-30));
  };

  // ---- 1. window format: handle, unit table, SpMM, combine ------------------------------------------------------------------
  const Handle h = preprocess(d_indptr, d_indices, n, n, e, s_main);
  const UnitTable t = build_unit_table(h, n, s_main);
  float* partials = dev_alloc<float>((size_t)std::max(1, t.header[2]) * 16 * f);
  HIP_OK(hipMemsetAsync(d_c, 0xFF, (size_t)n * f * sizeof(float), s_main));  // NaN pattern: every element must be written
  window_spmm(h, t, n, (int)e, f, d_b, d_c, /*atomic_out=*/0, partials, s_main);
  RC_OK(voltrix_launch_combine_partials(t.cuts, t.header[1], partials, d_c, n, f, /*accumulate=*/0, nullptr, s_main, &rc_));
  HIP_OK(hipStreamSynchronize(s_main));
  const double err_window = max_rel_err(to_host(d_c, (size_t)n * f));

  // ---- 2. two-level format: plan + residual handle, panel kernel || window kernel onto a zeroed C ----------------------------
  const int waves = 8, row_blocks = 4, tau = 3, panel_rows = waves * row_blocks * 16;
  const int num_panels = (n + panel_rows - 1) / panel_rows;
  void* plan_ws = dev_alloc<char>((size_t)voltrix_panel_plan_workspace_bytes(n, waves, row_blocks));
  int* panel_ptr = dev_alloc<int>(num_panels + 1);
  int* resid_indptr = dev_alloc<int>(n + 1);
  int* status = dev_alloc<int>(1);
  RC_OK(voltrix_launch_panel_plan_count(d_indptr, d_indices, n, n, e, waves, row_blocks, tau, plan_ws, panel_ptr,
                                        resid_indptr, status, s_main, &rc_));
  int ksteps = 0, resid_edges = 0, bad = 0;
  HIP_OK(hipMemcpyAsync(&ksteps, panel_ptr + num_panels, sizeof(int), hipMemcpyDeviceToHost, s_main));
  HIP_OK(hipMemcpyAsync(&resid_edges, resid_indptr + n, sizeof(int), hipMemcpyDeviceToHost, s_main));
  HIP_OK(hipMemcpyAsync(&bad, status, sizeof(int), hipMemcpyDeviceToHost, s_main));
  HIP_OK(hipStreamSynchronize(s_main));
  if (bad) return 6;
  int* resid_indices = dev_alloc<int>(resid_edges);
  int* panel_cols = dev_alloc<int>(32 * ((size_t)ksteps + 2));
  uint32_t* panel_bits = dev_alloc<uint32_t>(((size_t)ksteps + 1) * waves * 64);
  int* panel_order = dev_alloc<int>(num_panels);
  RC_OK(voltrix_launch_panel_plan_fill(d_indptr, d_indices, n, n, e, waves, row_blocks, tau, plan_ws, panel_ptr,
                                       resid_indptr, ksteps, resid_indices, panel_cols, panel_bits, s_main, &rc_));
  // xcd_ptr = nullptr everywhere in this example: XCD ranges of equal panel / window COUNTS.  Ranges of equal work (what
  // voltrix/hybrid.py::balance_xcd_ranges computes from panel_ptr and the residual's blk_offsets) are a speed matter only.
  RC_OK(voltrix_launch_panel_order(panel_ptr, num_panels, /*group=*/1, /*xcd_ptr=*/nullptr, panel_order, s_main, &rc_));
  const Handle hr = preprocess(resid_indptr, resid_indices, n, n, resid_edges, s_main);
  const UnitTable tr = build_unit_table(hr, n, s_main);
  float* partials_r = dev_alloc<float>((size_t)std::max(1, tr.header[2]) * 16 * f);

  hipEvent_t fork, join;
  HIP_OK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  HIP_OK(hipMemsetAsync(d_c, 0, (size_t)n * f * sizeof(float), s_main));
  HIP_OK(hipEventRecord(fork, s_main));
  HIP_OK(hipStreamWaitEvent(s_side, fork, 0));
  RC_OK(voltrix_launch_spmm_panel_f16(panel_ptr, panel_cols, panel_bits, panel_order, /*xcd_ptr=*/nullptr, 0, n, f, d_b,
                                      /*input_rows=*/n, d_c,
                                      /*accumulate=*/2, /*fs=*/128, /*depth=*/3, waves, row_blocks, /*ksteps=*/1,
                                      VOLTRIX_SLAB_AUTO, /*out_scale=*/nullptr, s_side, &rc_));
  HIP_OK(hipEventRecord(join, s_side));
  window_spmm(hr, tr, n, resid_edges, f, d_b, d_c, /*atomic_out=*/1, partials_r, s_main);
  HIP_OK(hipStreamWaitEvent(s_main, join, 0));
  RC_OK(voltrix_launch_combine_partials(tr.cuts, tr.header[1], partials_r, d_c, n, f, /*accumulate=*/1, nullptr, s_main,
                                        &rc_));
  HIP_OK(hipStreamSynchronize(s_main));
  const double err_two_level = max_rel_err(to_host(d_c, (size_t)n * f));

  // ---- 2b. the same step under the round-4 schedules, every table built by the library: XCD ranges of equal work for both
  // ---- kernels (panel work = 6.6 x k-steps + residual stages), the residual's unit table over those ranges, and the panel
  // ---- kernel over PIECES of at most `cap` k-steps -- the product cuts panels longer than a CU's fair share (S / 256); a small
  // ---- cap here, so that the combine path runs on any graph.  Pieces of a cut panel STORE to partial tiles;
  // ---- combine_panel_partials adds them to C in slot order after the join. ------------------------------------------------
  int* xcd_ptr = dev_alloc<int>(9);
  int* window_xcd_ptr = dev_alloc<int>(9);
  RC_OK(voltrix_launch_xcd_ranges_of_panels(panel_ptr, hr.blk_offsets, n, panel_rows, /*kstep_cost_x10=*/66, xcd_ptr,
                                            window_xcd_ptr, s_main, &rc_));
  const UnitTable tr2 = build_unit_table(hr, n, s_main, window_xcd_ptr);
  float* partials_r2 = dev_alloc<float>((size_t)std::max(1, tr2.header[2]) * 16 * f);
  const int cap = 4;
  void* parts_ws = dev_alloc<char>((size_t)voltrix_panel_parts_workspace_bytes(num_panels));
  int* d_parts_header = dev_alloc<int>(8);
  RC_OK(voltrix_launch_panel_parts_count(panel_ptr, num_panels, cap, xcd_ptr, parts_ws, d_parts_header, s_main, &rc_));
  int parts_header[8];
  HIP_OK(hipMemcpyAsync(parts_header, d_parts_header, sizeof(parts_header), hipMemcpyDeviceToHost, s_main));
  HIP_OK(hipStreamSynchronize(s_main));
  const int num_parts = parts_header[0], num_pcuts = parts_header[1], slots = parts_header[2], max_parts = parts_header[3];
  int* parts = dev_alloc<int>(4 * (size_t)std::max(1, num_parts));
  int* part_xcd_ptr = dev_alloc<int>(9);
  int* pcuts = dev_alloc<int>(4 * (size_t)std::max(1, num_pcuts));
  float* panel_partials = dev_alloc<float>((size_t)std::max(1, slots) * panel_rows * f);
  RC_OK(voltrix_launch_panel_parts_fill(panel_ptr, num_panels, cap, xcd_ptr, parts_ws, parts, part_xcd_ptr, pcuts, s_main,
                                        &rc_));
  HIP_OK(hipMemsetAsync(d_c, 0, (size_t)n * f * sizeof(float), s_main));
  HIP_OK(hipEventRecord(fork, s_main));
  HIP_OK(hipStreamWaitEvent(s_side, fork, 0));
  RC_OK(voltrix_launch_spmm_panel_parts_f16(panel_ptr, panel_cols, panel_bits, parts, num_parts, part_xcd_ptr, max_parts,
                                            panel_partials, n, f, d_b, /*input_rows=*/n, d_c, /*accumulate=*/2, /*fs=*/128,
                                            /*depth=*/3, waves, row_blocks, /*ksteps=*/1, VOLTRIX_SLAB_AUTO, nullptr, s_side,
                                            &rc_));
  HIP_OK(hipEventRecord(join, s_side));
  window_spmm(hr, tr2, n, resid_edges, f, d_b, d_c, /*atomic_out=*/1, partials_r2, s_main);
  HIP_OK(hipStreamWaitEvent(s_main, join, 0));
  RC_OK(voltrix_launch_combine_partials(tr2.cuts, tr2.header[1], partials_r2, d_c, n, f, /*accumulate=*/1, nullptr, s_main,
                                        &rc_));
  RC_OK(voltrix_launch_combine_panel_partials(pcuts, num_pcuts, panel_partials, d_c, n, f, panel_rows, /*accumulate=*/1,
                                              s_main, &rc_));
  HIP_OK(hipStreamSynchronize(s_main));
  const double err_pieces = max_rel_err(to_host(d_c, (size_t)n * f));

  // ---- 3. the same two-level product as ONE launch: stage records of the residual handle, then spmm_fused (plain stores:
  // ---- no zero fill, no second stream, no combine pass; the residual handle itself is no longer needed afterwards) ----------
  int fused_waves = 0, fused_row_blocks = 0;
  voltrix_fused_panel_geometry(&fused_waves, &fused_row_blocks);   // 4 x 8 since round 4: never a constant of the host's own
  const int num_waves_total = fused_waves * num_panels;
  void* rec_ws = dev_alloc<char>((size_t)voltrix_fused_records_workspace_bytes(n));
  int* wave_ptr = dev_alloc<int>(num_waves_total + 1);
  RC_OK(voltrix_launch_fused_records_count(hr.blk_offsets, hr.hspa_packed, n, rec_ws, wave_ptr, s_main, &rc_));
  int num_records = 0;
  HIP_OK(hipMemcpyAsync(&num_records, wave_ptr + num_waves_total, sizeof(int), hipMemcpyDeviceToHost, s_main));
  HIP_OK(hipStreamSynchronize(s_main));
  uint32_t* records = dev_alloc<uint32_t>(((size_t)num_records + 1) * 64);
  RC_OK(voltrix_launch_fused_records_fill(hr.blk_offsets, hr.hspa_packed, hr.hind, n, wave_ptr, num_records, records, s_main,
                                          &rc_));
  HIP_OK(hipMemsetAsync(d_c, 0xFF, (size_t)n * f * sizeof(float), s_main));  // NaN pattern: every element must be written
  const int fused_fs = f <= 32 ? 32 : (f <= 64 ? 64 : 128);
  RC_OK(voltrix_launch_spmm_fused_f16(panel_ptr, panel_cols, panel_bits, panel_order, /*xcd_ptr=*/nullptr, 0, wave_ptr, records,
                                      n, f, d_b, d_c,
                                      fused_fs, /*depth=*/fused_fs == 128 ? 3 : 4, /*pace_blocks=*/0, /*out_scale=*/nullptr,
                                      s_main, &rc_));
  HIP_OK(hipStreamSynchronize(s_main));
  const double err_fused = max_rel_err(to_host(d_c, (size_t)n * f));

  // ---- 4. Cuthill-McKee row order of the component of the least-degree node (locality reorder; voltrix/reorder_kernels.hpp):
  // ---- transpose -> degrees and tie order (host) -> search from a start node -> ranks; checked against a CPU search -----------
  void* tr_ws = dev_alloc<char>((size_t)voltrix_csr_transpose_workspace_bytes(e));
  int* t_indptr = dev_alloc<int>(n + 1);
  int* t_indices = dev_alloc<int>(std::max<int64_t>(e, 1));
  RC_OK(voltrix_launch_csr_transpose(d_indptr, d_indices, n, n, e, tr_ws, t_indptr, t_indices, s_main, &rc_));
  const std::vector<int> h_tptr = to_host(t_indptr, n + 1), h_tidx = to_host(t_indices, e);
  std::vector<int> deg(n), by_deg(n), tie(n);
  for (int u = 0; u < n; ++u) deg[u] = indptr[u + 1] - indptr[u] + h_tptr[u + 1] - h_tptr[u];
  for (int u = 0; u < n; ++u) by_deg[u] = u;
  std::stable_sort(by_deg.begin(), by_deg.end(), [&](int a, int b) { return deg[a] < deg[b]; });
  for (int k = 0; k < n; ++k) tie[by_deg[k]] = k;
  int start = -1;
  for (int k = 0; k < n && start < 0; ++k)
    if (deg[by_deg[k]] > 0) start = by_deg[k];
  int* d_tie = dev_alloc<int>(n);
  int* d_level = dev_alloc<int>(n);
  int* d_rank = dev_alloc<int>(n);
  int* d_queue = dev_alloc<int>(n);
  int* d_level_off = dev_alloc<int>(n + 2);
  int* d_ctrl = dev_alloc<int>(8);
  HIP_OK(hipMemcpy(d_tie, tie.data(), n * sizeof(int), hipMemcpyHostToDevice));
  HIP_OK(hipMemsetAsync(d_level, 0xFF, n * sizeof(int), s_main));   // -1: unvisited
  HIP_OK(hipMemsetAsync(d_rank, 0xFF, n * sizeof(int), s_main));
  RC_OK(voltrix_launch_bfs_seed(start, n, d_level, d_queue, d_ctrl, d_level_off, s_main, &rc_));
  int ctrl[8] = {0};
  do {   // one single-workgroup launch for the narrow levels + four whole-chip levels per host read
    RC_OK(voltrix_launch_bfs_levels(d_indptr, d_indices, t_indptr, t_indices, n, n, d_level, d_queue, d_ctrl, d_level_off,
                                    /*wide_levels=*/4, s_main, &rc_));
    HIP_OK(hipMemcpyAsync(ctrl, d_ctrl, sizeof(ctrl), hipMemcpyDeviceToHost, s_main));
    HIP_OK(hipStreamSynchronize(s_main));
  } while (!ctrl[4]);
  const int cm_levels = ctrl[3] + 1, cm_nodes = ctrl[1];
  const std::vector<int> level_off = to_host(d_level_off, cm_levels + 1);
  int64_t biggest = 0;
  for (int d = 0; d < cm_levels; ++d)
    if (level_off[d + 1] - level_off[d] > 1024) biggest = std::max<int64_t>(biggest, level_off[d + 1] - level_off[d]);
  void* cm_ws = dev_alloc<char>((size_t)std::max<int64_t>(16, voltrix_cm_rank_workspace_bytes(biggest)));
  RC_OK(voltrix_launch_cm_rank(d_indptr, d_indices, t_indptr, t_indices, n, n, d_level, d_rank, d_tie, d_queue, d_level_off,
                               level_off.data(), cm_levels, /*base=*/0, cm_ws, s_main, &rc_));
  HIP_OK(hipStreamSynchronize(s_main));
  const std::vector<int> order = to_host(d_queue, cm_nodes);
  // CPU search from the same start over rows of A and of A^T: levels, then (earliest-ranked parent, tie) inside a level
  std::vector<int> cpu_level(n, -1), cpu_rank(n, -1), cpu_order{start}, frontier{start};
  cpu_level[start] = 0;
  cpu_rank[start] = 0;
  auto for_neighbours = [&](int u, auto&& fn) {
    for (int p = indptr[u]; p < indptr[u + 1]; ++p) fn(indices[p]);
    for (int p = h_tptr[u]; p < h_tptr[u + 1]; ++p) fn(h_tidx[p]);
  };
  for (int d = 0; !frontier.empty(); ++d) {
    std::vector<int> next;
    for (int u : frontier)
      for_neighbours(u, [&](int v) {
        if (cpu_level[v] < 0) {
          cpu_level[v] = d + 1;
          next.push_back(v);
        }
      });
    std::vector<int> best(n, 0x7fffffff);
    for (int v : next) for_neighbours(v, [&](int u) { if (cpu_level[u] == d) best[v] = std::min(best[v], cpu_rank[u]); });
    std::sort(next.begin(), next.end(), [&](int a, int b) { return best[a] != best[b] ? best[a] < best[b] : tie[a] < tie[b]; });
    for (int v : next) {
      cpu_rank[v] = (int)cpu_order.size();
      cpu_order.push_back(v);
    }
    frontier.swap(next);
  }
  const bool order_ok = order == cpu_order;

  std::printf("N=%d nnz=%lld F=%d | window format: %d TC blocks, %d units (%d cut windows), rel err %.3e | two-level: %d "
              "k-steps, %d residual edges (%d units), rel err %.3e | panels in %d pieces (%d cut): rel err %.3e | one launch: %d "
              "stage records, rel err %.3e\n",
              n, (long long)e, f, h.total_blocks, t.header[0], t.header[1], err_window, ksteps, resid_edges, tr.header[0],
              err_two_level, num_parts, num_pcuts, err_pieces, num_records, err_fused);
  std::printf("Cuthill-McKee search from row %d: %d rows in %d levels, order %s the CPU search\n", start, cm_nodes, cm_levels,
              order_ok ? "equals" : "DIFFERS FROM");
  const bool ok = err_window < 1e-5 && err_two_level < 1e-5 && err_fused < 1e-5 && err_pieces < 1e-5 &&
                  std::isfinite(err_window) && std::isfinite(err_two_level) && std::isfinite(err_fused) &&
                  std::isfinite(err_pieces) && order_ok;
  return ok ? 0 : 1;
}
