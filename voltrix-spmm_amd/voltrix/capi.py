"""ctypes binding of the ahead-of-time C-ABI library ``lib/libvoltrix_hip.so`` (include/voltrix_capi.h).

The library is the drop-in boundary for non-Python hosts and the home of the gfx950 extensions that have no
reference counterpart (fused GPU preprocess, fp16 operand, tile enumeration).  There is NO fallback: if the
library is missing it is built with hipcc (csrc/Makefile); if that fails the import error is raised.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.path.join(_ROOT, "lib", "libvoltrix_hip.so")
CSRC_DIR = os.path.join(_ROOT, "csrc")

_lib = None

# every symbol include/voltrix_capi.h declares (tests/test_capi_symbols.py cross-checks this list with the header)
SYMBOLS = (
    "voltrix_abi_version",
    "voltrix_launch_preprocess",
    "voltrix_launch_hmat_gen",
    "voltrix_launch_hmat_packed_swizzle",
    "voltrix_launch_spmm",
    "voltrix_launch_spmm_f32_tile",
    "voltrix_launch_spmm_f16",
    "voltrix_launch_spmm_f16_tile",
    "voltrix_spmm_f32_workspace_bytes",
    "voltrix_launch_spmm_f32_as_f16",
    "voltrix_launch_spmm_f16_sched",
    "voltrix_launch_spmm_bf16_sched",
    "voltrix_launch_combine_partials",
    "voltrix_stream_table_workspace_bytes",
    "voltrix_stream_table_fill_workspace_bytes",
    "voltrix_launch_stream_table_count",
    "voltrix_launch_stream_table_fill",
    "voltrix_launch_spmm_stream_f16",
    "voltrix_launch_spmm_stream_bf16",
    "voltrix_launch_spmm_panel_f16",
    "voltrix_launch_spmm_panel_bf16",
    "voltrix_launch_spmm_panel_parts_f16",
    "voltrix_launch_spmm_panel_parts_bf16",
    "voltrix_launch_combine_panel_partials",
    "voltrix_launch_xcd_ranges_of_work",
    "voltrix_launch_xcd_ranges_of_windows",
    "voltrix_launch_xcd_ranges_of_panels",
    "voltrix_panel_parts_workspace_bytes",
    "voltrix_launch_panel_parts_count",
    "voltrix_launch_panel_parts_fill",
    "voltrix_launch_spmm_fused_f16",
    "voltrix_launch_spmm_fused_bf16",
    "voltrix_fused_panel_geometry",
    "voltrix_fused_records_workspace_bytes",
    "voltrix_launch_fused_records_count",
    "voltrix_launch_fused_records_fill",
    "voltrix_panel_plan_workspace_bytes",
    "voltrix_launch_panel_plan_count",
    "voltrix_launch_panel_plan_fill",
    "voltrix_launch_panel_order",
    "voltrix_spmm_default_tile",
    "voltrix_spmm_num_tiles",
    "voltrix_spmm_tile_at",
    "voltrix_launch_window_order",
    "voltrix_launch_spmm_bf16",
    "voltrix_launch_spmm_bf16_tile",
    "voltrix_launch_cast_f32_f16",
    "voltrix_launch_cast_f32_f16_scaled",
    "voltrix_launch_scale_rows",
    "voltrix_launch_spmm_csr_rows",
    "voltrix_launch_spmm_csr_rows_weighted",
    "voltrix_launch_scatter_values",
    "voltrix_csr_preprocess_workspace_bytes",
    "voltrix_launch_csr_window_count",
    "voltrix_launch_csr_fill",
    "voltrix_unit_table_workspace_bytes",
    "voltrix_unit_table_fill_workspace_bytes",
    "voltrix_launch_unit_table_count",
    "voltrix_launch_unit_table_fill",
    "voltrix_csr_transpose_workspace_bytes",
    "voltrix_launch_csr_transpose",
    "voltrix_launch_bfs_seed",
    "voltrix_launch_bfs_levels",
    "voltrix_cm_rank_workspace_bytes",
    "voltrix_launch_cm_rank",
    "voltrix_launch_chol_inv_transposed",
)


def build(force: bool = False) -> str:
    """Compile the library for gfx950 (works without a GPU)."""
    cmd = ["make", "-s", "-C", CSRC_DIR, "-j4"]
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC_DIR, "clean"])
    subprocess.check_call(cmd)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.voltrix_abi_version.restype = ctypes.c_int
        _lib.voltrix_spmm_num_tiles.restype = ctypes.c_int
        _lib.voltrix_csr_preprocess_workspace_bytes.restype = ctypes.c_int64
        _lib.voltrix_panel_plan_workspace_bytes.restype = ctypes.c_int64
        _lib.voltrix_spmm_f32_workspace_bytes.restype = ctypes.c_int64
        _lib.voltrix_fused_records_workspace_bytes.restype = ctypes.c_int64
        _lib.voltrix_unit_table_workspace_bytes.restype = ctypes.c_int64
        _lib.voltrix_unit_table_fill_workspace_bytes.restype = ctypes.c_int64
        _lib.voltrix_cm_rank_workspace_bytes.restype = ctypes.c_int64
        _lib.voltrix_csr_transpose_workspace_bytes.restype = ctypes.c_int64
        for name in SYMBOLS:
            if name.startswith("voltrix_launch_") or name in ("voltrix_spmm_default_tile", "voltrix_spmm_tile_at"):
                getattr(_lib, name).restype = None
    return _lib


class VoltrixError(RuntimeError):
    pass


_RC = {1: "bad shape / alignment", 2: "HIP launch error", 3: "tile configuration not instantiated",
       4: "int32 overflow of the block format", 5: "duplicate CSR entries"}


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise VoltrixError(f"{what}: return code {rc} ({_RC.get(rc, 'unknown')})")


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def default_tile(embedding_dim: int, is_f16: bool):
    fs, d, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib().voltrix_spmm_default_tile(ctypes.c_int(embedding_dim), ctypes.c_int(int(is_f16)), ctypes.byref(fs),
                                    ctypes.byref(d), ctypes.byref(w))
    return fs.value, d.value, w.value


def tiles(is_f16: bool):
    out = []
    for i in range(lib().voltrix_spmm_num_tiles(ctypes.c_int(int(is_f16)))):
        fs, d, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib().voltrix_spmm_tile_at(ctypes.c_int(int(is_f16)), ctypes.c_int(i), ctypes.byref(fs), ctypes.byref(d),
                                   ctypes.byref(w))
        out.append((fs.value, d.value, w.value))
    return out


CSR_PATHS = {None: -1, "auto": -1, "sort": 0, "bitmap": 1, "mixed": 2}   # `path` of the fused preprocess entry points
SLAB_AUTO, SLAB_ONE_GRID, SLAB_LAUNCHES = -1, 0, 1                       # `slab_policy` of the SpMM launchers


def csr_preprocess_workspace_bytes(num_nodes: int, num_cols: int, num_edges: int, path=None) -> int:
    return int(lib().voltrix_csr_preprocess_workspace_bytes(ctypes.c_int(num_nodes), ctypes.c_int(num_cols),
                                                            ctypes.c_int64(num_edges), ctypes.c_int(CSR_PATHS[path])))


def launch_csr_window_count(indptr, indices, num_nodes, num_cols, workspace, block_partition, pointer1, status,
                            stream, path=None) -> None:
    rc = ctypes.c_int(-1)
    lib().voltrix_launch_csr_window_count(_ptr(indptr), _ptr(indices), ctypes.c_int(num_nodes), ctypes.c_int(num_cols),
                                          ctypes.c_int64(indices.numel()), ctypes.c_int(CSR_PATHS[path]), _ptr(workspace),
                                          _ptr(block_partition), _ptr(pointer1), _ptr(status), ctypes.c_void_p(stream),
                                          ctypes.byref(rc))
    check(rc.value, "voltrix_launch_csr_window_count")


def launch_csr_fill(indptr, indices, num_nodes, num_cols, workspace, pointer1, hspa_packed, hind, stream, path=None) -> None:
    rc = ctypes.c_int(-1)
    lib().voltrix_launch_csr_fill(_ptr(indptr), _ptr(indices), ctypes.c_int(num_nodes), ctypes.c_int(num_cols),
                                  ctypes.c_int64(indices.numel()), ctypes.c_int(CSR_PATHS[path]), _ptr(workspace),
                                  _ptr(pointer1), _ptr(hspa_packed), _ptr(hind), ctypes.c_void_p(stream), ctypes.byref(rc))
    check(rc.value, "voltrix_launch_csr_fill")


def launch_spmm(blk_offsets, hspa_packed, hind, num_nodes, num_edges, embedding_dim, input_ptr, output_ptr, is_f16,
                tile, stream, window_order=0, out_scale=0) -> int:
    """Raw-pointer launch (used by bench.py and the tests); ``window_order`` is 0 (natural) or the device pointer of
    the schedule written by :func:`launch_window_order`; ``out_scale`` is 0 or the device pointer of the float written
    by :func:`launch_cast_f32_f16_scaled`.  Returns the return code."""
    rc = ctypes.c_int(-1)
    if is_f16 == "bf16":  # operand kind: True = fp16, False = fp32, "bf16" = bfloat16 (same tiles as fp16)
        fn = lib().voltrix_launch_spmm_bf16_tile
    else:
        fn = lib().voltrix_launch_spmm_f16_tile if is_f16 else lib().voltrix_launch_spmm_f32_tile
    fn(ctypes.c_void_p(blk_offsets), ctypes.c_void_p(hspa_packed), ctypes.c_void_p(hind), ctypes.c_int(num_nodes),
       ctypes.c_int(num_edges), ctypes.c_int(embedding_dim), ctypes.c_void_p(input_ptr), ctypes.c_void_p(output_ptr),
       ctypes.c_int(tile[0]), ctypes.c_int(tile[1]), ctypes.c_int(tile[2]), ctypes.c_void_p(window_order),
       ctypes.c_void_p(out_scale), ctypes.c_void_p(stream), ctypes.byref(rc))
    return rc.value


def launch_spmm_sched(blk_offsets, hspa_packed, hind, num_nodes, num_edges, embedding_dim, input_ptr, output_ptr, tile,
                      stream, window_order=0, out_scale=0, atomic_out=False, bf16=False, table=None, partials=0,
                      row_map=0, units_per_wave=1) -> int:
    """16-bit operand launch with the schedule / output extensions (include/voltrix_capi.h): ``atomic_out`` = add the
    result onto a pre-zeroed output with float atomics (two-level format without a join pass); ``table`` = a
    ``voltrix.schedule.UnitTable`` (replaces ``window_order``), ``partials`` = device pointer of its partial tiles.
    Returns the return code."""
    rc = ctypes.c_int(-1)
    fn = lib().voltrix_launch_spmm_bf16_sched if bf16 else lib().voltrix_launch_spmm_f16_sched
    fn(ctypes.c_void_p(blk_offsets), ctypes.c_void_p(hspa_packed), ctypes.c_void_p(hind), ctypes.c_int(num_nodes),
       ctypes.c_int(num_edges), ctypes.c_int(embedding_dim), ctypes.c_void_p(input_ptr), ctypes.c_void_p(output_ptr),
       ctypes.c_int(tile[0]), ctypes.c_int(tile[1]), ctypes.c_int(tile[2]), ctypes.c_void_p(window_order),
       ctypes.c_void_p(out_scale), ctypes.c_int(int(atomic_out)),
       ctypes.c_void_p(table.units.data_ptr() if table is not None else 0),
       ctypes.c_void_p(table.unit_ptr.data_ptr() if table is not None else 0),
       ctypes.c_int(table.max_units_per_xcd if table is not None else 0), ctypes.c_void_p(partials),
       ctypes.c_void_p(row_map), ctypes.c_int(int(units_per_wave)), ctypes.c_void_p(stream), ctypes.byref(rc))
    return rc.value


def launch_combine_partials(table, partials_ptr, output_ptr, num_nodes, embedding_dim, accumulate, stream,
                            row_map=0) -> int:
    """Sum the partial tiles of the cut windows of ``table`` (a ``voltrix.schedule.UnitTable``) into the output."""
    rc = ctypes.c_int(-1)
    lib().voltrix_launch_combine_partials(ctypes.c_void_p(table.cuts.data_ptr()), ctypes.c_int(table.num_cuts),
                                          ctypes.c_void_p(partials_ptr), ctypes.c_void_p(output_ptr),
                                          ctypes.c_int(num_nodes), ctypes.c_int(embedding_dim),
                                          ctypes.c_int(int(accumulate)), ctypes.c_void_p(row_map),
                                          ctypes.c_void_p(stream), ctypes.byref(rc))
    return rc.value


def build_unit_table(blk_offsets, num_nodes: int, max_stages: int = 0, stream=None, xcd_ptr=None):
    """The handle's unit table through the library's two-phase builder (voltrix/unit_table.hpp): returns
    ``(units int32 [U, 4], unit_ptr int32 [9], cuts int32 [C, 4], header)`` with ``header`` = the eight ints of phase 1
    as a Python list (num_units, num_cuts, num_slots, max_units_per_xcd, max_stages, top, 0, 0).  ``xcd_ptr``: None (equal
    window ranges per XCD) or a device int32 [9] tensor of first windows (ranges of equal work).  One host sync."""
    import torch

    dev = blk_offsets.device
    stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
    workspace = torch.empty(max(16, int(lib().voltrix_unit_table_workspace_bytes(ctypes.c_int(num_nodes)))),
                            dtype=torch.uint8, device=dev)
    header = torch.empty(8, dtype=torch.int32, device=dev)
    rc = ctypes.c_int(-1)
    xp = ctypes.c_void_p(xcd_ptr.data_ptr() if xcd_ptr is not None else 0)
    lib().voltrix_launch_unit_table_count(_ptr(blk_offsets), ctypes.c_int(num_nodes), ctypes.c_int(int(max_stages)), xp,
                                          _ptr(workspace), _ptr(header), ctypes.c_void_p(stream), ctypes.byref(rc))
    check(rc.value, "voltrix_launch_unit_table_count")
    head = [int(v) for v in header.tolist()]   # the sync
    num_units, num_cuts, top = head[0], head[1], head[5]
    units = torch.empty((num_units, 4), dtype=torch.int32, device=dev)
    cuts = torch.empty((num_cuts, 4), dtype=torch.int32, device=dev)
    unit_ptr = torch.empty(9, dtype=torch.int32, device=dev)
    fill_ws = torch.empty(max(16, int(lib().voltrix_unit_table_fill_workspace_bytes(ctypes.c_int64(num_units)))),
                          dtype=torch.uint8, device=dev)
    lib().voltrix_launch_unit_table_fill(_ptr(blk_offsets), ctypes.c_int(num_nodes), xp, _ptr(workspace), _ptr(fill_ws),
                                         ctypes.c_int(num_units), ctypes.c_int(num_cuts), ctypes.c_int(top), _ptr(units),
                                         _ptr(unit_ptr), _ptr(cuts), ctypes.c_void_p(stream), ctypes.byref(rc))
    check(rc.value, "voltrix_launch_unit_table_fill")
    return units, unit_ptr, cuts, head


def build_stream_table(blk_offsets, hspa_packed, hind, num_nodes: int, run_cost: int = 0, cut_stages: int = 0, stream=None):
    """The handle's stream tables through the library's two-phase builder (voltrix/stream_table.hpp): returns ``(units int32
    [U, 8], runs int32 [R, 4], run_ptr int32 [9], cuts int32 [C, 4], header)`` with ``header`` = (num_units, num_cuts, num_slots,
    num_runs, max_runs_per_xcd, run_cost, cut_stages).  Two host syncs (the sizes of the outputs, then the number of runs)."""
    import torch

    dev = blk_offsets.device
    stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
    L = lib()
    workspace = torch.empty(max(16, int(L.voltrix_stream_table_workspace_bytes(ctypes.c_int(num_nodes)))), dtype=torch.uint8,
                            device=dev)
    header = torch.empty(8, dtype=torch.int32, device=dev)
    rc = ctypes.c_int(-1)
    L.voltrix_launch_stream_table_count(_ptr(blk_offsets), _ptr(hspa_packed), _ptr(hind), ctypes.c_int(num_nodes),
                                        ctypes.c_int(int(run_cost)), ctypes.c_int(int(cut_stages)), _ptr(workspace),
                                        _ptr(header), ctypes.c_void_p(stream), ctypes.byref(rc))
    check(rc.value, "voltrix_launch_stream_table_count")
    num_units, num_cuts, num_slots, run_bound, rcost, cut = [int(v) for v in header.tolist()][:6]   # the sync
    units = torch.empty((num_units, 8), dtype=torch.int32, device=dev)
    cuts = torch.empty((num_cuts, 4), dtype=torch.int32, device=dev)
    runs = torch.empty((max(1, run_bound), 4), dtype=torch.int32, device=dev)
    run_ptr = torch.empty(9, dtype=torch.int32, device=dev)
    header2 = torch.empty(4, dtype=torch.int32, device=dev)
    fill_ws = torch.empty(max(16, int(L.voltrix_stream_table_fill_workspace_bytes(ctypes.c_int64(num_units)))),
                          dtype=torch.uint8, device=dev)
    L.voltrix_launch_stream_table_fill(_ptr(blk_offsets), ctypes.c_int(num_nodes), _ptr(workspace), _ptr(fill_ws),
                                       ctypes.c_int(num_units), ctypes.c_int(num_cuts), ctypes.c_int(run_bound),
                                       ctypes.c_int(max(2, rcost)), _ptr(units), _ptr(cuts), _ptr(runs), _ptr(run_ptr),
                                       _ptr(header2), ctypes.c_void_p(stream), ctypes.byref(rc))
    check(rc.value, "voltrix_launch_stream_table_fill")
    num_runs, max_runs, oversized = [int(v) for v in header2.tolist()][:3]
    assert oversized == 0, "stream table with a run of more than 64 units (run_cost > 128?): the kernel cannot walk it"
    return units, runs[:num_runs], run_ptr, cuts, (num_units, num_cuts, num_slots, num_runs, max_runs, rcost, cut)


def launch_spmm_stream(hspa_packed, hind, num_nodes: int, embedding_dim: int, feat, output, table, partials, out_scale=None,
                       tile=(0, 0, 0), stream=None, input_rows: int = 0, slab_policy: int = -1) -> int:
    """``voltrix_launch_spmm_stream_f16 / _bf16`` on a ``voltrix.schedule.StreamTable`` (the stream kernel through the C-ABI)."""
    import torch

    stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
    fn = lib().voltrix_launch_spmm_stream_bf16 if feat.dtype == torch.bfloat16 else lib().voltrix_launch_spmm_stream_f16
    from .utils import timed_launch

    rc = ctypes.c_int(-1)
    with timed_launch("spmm_stream", stream):
        fn(_ptr(hspa_packed), _ptr(hind), ctypes.c_int(num_nodes), ctypes.c_int(embedding_dim), _ptr(feat),
           ctypes.c_int64(int(input_rows)), _ptr(output), _ptr(table.units), _ptr(table.runs), _ptr(table.run_ptr),
           ctypes.c_int(table.max_runs_per_xcd), _ptr(partials), ctypes.c_void_p(out_scale.data_ptr() if out_scale is not None else 0),
           ctypes.c_int(tile[0]), ctypes.c_int(tile[1]), ctypes.c_int(tile[2]), ctypes.c_int(int(slab_policy)),
           ctypes.c_void_p(stream), ctypes.byref(rc))
    return rc.value


def xcd_ranges_of_work(work, align: int = 1, stream=None):
    """int32 [9] on ``work``'s device: eight ranges of the int32 items ``work`` with about equal sums
    (voltrix/schedule_tables.hpp; voltrix/schedule.py::split_equal_work is the restatement).  Stream-ordered."""
    import torch

    out = torch.empty(9, dtype=torch.int32, device=work.device)
    rc = ctypes.c_int(-1)
    stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
    lib().voltrix_launch_xcd_ranges_of_work(_ptr(work), ctypes.c_int(work.numel()), ctypes.c_int(int(align)), _ptr(out),
                                            ctypes.c_void_p(stream), ctypes.byref(rc))
    check(rc.value, "voltrix_launch_xcd_ranges_of_work")
    return out


def xcd_ranges_of_windows(blk_offsets, num_nodes: int, align: int = 1, stream=None):
    """int32 [9]: first window of every XCD's range, ranges of equal STAGES of the handle ``blk_offsets``."""
    import torch

    out = torch.empty(9, dtype=torch.int32, device=blk_offsets.device)
    rc = ctypes.c_int(-1)
    stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
    lib().voltrix_launch_xcd_ranges_of_windows(_ptr(blk_offsets), ctypes.c_int(num_nodes), ctypes.c_int(int(align)), _ptr(out),
                                               ctypes.c_void_p(stream), ctypes.byref(rc))
    check(rc.value, "voltrix_launch_xcd_ranges_of_windows")
    return out


def xcd_ranges_of_panels(panel_ptr, resid_blk_offsets, num_nodes: int, panel_rows: int, kstep_cost_x10: int, stream=None):
    """``(xcd_ptr, window_xcd_ptr)`` int32 [9] each: ranges of panels with equal work (``kstep_cost_x10 / 10`` x k-steps +
    residual stages) and the same ranges counted in windows."""
    import torch

    dev = panel_ptr.device
    out, out_w = torch.empty(9, dtype=torch.int32, device=dev), torch.empty(9, dtype=torch.int32, device=dev)
    rc = ctypes.c_int(-1)
    stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
    lib().voltrix_launch_xcd_ranges_of_panels(_ptr(panel_ptr), _ptr(resid_blk_offsets), ctypes.c_int(num_nodes),
                                              ctypes.c_int(panel_rows), ctypes.c_int(int(kstep_cost_x10)), _ptr(out),
                                              _ptr(out_w), ctypes.c_void_p(stream), ctypes.byref(rc))
    check(rc.value, "voltrix_launch_xcd_ranges_of_panels")
    return out, out_w


def build_panel_parts(panel_ptr, cap: int, panel_xcd_ptr=None, stream=None):
    """The panel kernel's piece table through the library's two-phase builder (voltrix/schedule_tables.hpp): returns
    ``(parts int32 [P, 4], part_xcd_ptr int32 [9], cuts int32 [C, 4], header)`` with ``header`` = (pieces, cut panels,
    slots, pieces of the longest XCD range, cap, 0, 0, 0) as a Python list.  One host sync."""
    import torch

    dev = panel_ptr.device
    num_panels = panel_ptr.numel() - 1
    stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
    workspace = torch.empty(max(16, int(lib().voltrix_panel_parts_workspace_bytes(ctypes.c_int(num_panels)))),
                            dtype=torch.uint8, device=dev)
    header = torch.empty(8, dtype=torch.int32, device=dev)
    rc = ctypes.c_int(-1)
    xp = ctypes.c_void_p(panel_xcd_ptr.data_ptr() if panel_xcd_ptr is not None else 0)
    lib().voltrix_launch_panel_parts_count(_ptr(panel_ptr), ctypes.c_int(num_panels), ctypes.c_int(int(cap)), xp,
                                           _ptr(workspace), _ptr(header), ctypes.c_void_p(stream), ctypes.byref(rc))
    check(rc.value, "voltrix_launch_panel_parts_count")
    head = [int(v) for v in header.tolist()]   # the sync
    parts = torch.empty((head[0], 4), dtype=torch.int32, device=dev)
    cuts = torch.empty((max(1, head[1]), 4), dtype=torch.int32, device=dev)
    part_xcd_ptr = torch.empty(9, dtype=torch.int32, device=dev)
    lib().voltrix_launch_panel_parts_fill(_ptr(panel_ptr), ctypes.c_int(num_panels), ctypes.c_int(int(cap)), xp,
                                          _ptr(workspace), _ptr(parts), _ptr(part_xcd_ptr), _ptr(cuts),
                                          ctypes.c_void_p(stream), ctypes.byref(rc))
    check(rc.value, "voltrix_launch_panel_parts_fill")
    return parts, part_xcd_ptr, cuts[:head[1]], head


def spmm_f32_workspace_bytes(input_rows: int, embedding_dim: int) -> int:
    return int(lib().voltrix_spmm_f32_workspace_bytes(ctypes.c_int64(input_rows), ctypes.c_int(embedding_dim)))


def launch_spmm_f32_as_f16(blk_offsets, hspa_packed, hind, num_nodes, num_edges, embedding_dim, feat, output, workspace,
                           stream) -> int:
    """Route B for fp32 features: the reference's launch() arguments + a caller-owned workspace; scaled-fp16 operand on the
    default tile (include/voltrix_capi.h).  Tensors in, return code out."""
    rc = ctypes.c_int(-1)
    lib().voltrix_launch_spmm_f32_as_f16(_ptr(blk_offsets), _ptr(hspa_packed), _ptr(hind), ctypes.c_int(num_nodes),
                                         ctypes.c_int(num_edges), ctypes.c_int(embedding_dim), _ptr(feat),
                                         ctypes.c_int64(feat.shape[0]), _ptr(output), _ptr(workspace),
                                         ctypes.c_void_p(stream), ctypes.byref(rc))
    return rc.value


def launch_spmm_panel(plan, input_ptr, output_ptr, embedding_dim, accumulate, bf16, tile, out_scale, stream,
                      input_rows: int = 0, slab_policy: int = SLAB_AUTO, partials_ptr: int = 0) -> int:
    """Panel kernel (shared-column half of the two-level format); ``plan`` = voltrix.hybrid.PanelPlan, ``tile`` =
    (fs, depth, ksteps); ``input_rows`` = rows of the dense operand (0: the plan's rows).  A plan with a part table
    (``plan.parts``) goes through ``voltrix_launch_spmm_panel_parts_*`` with ``partials_ptr`` = the call's partial-tile
    buffer.  Returns the return code."""
    rc = ctypes.c_int(-1)
    parts = getattr(plan, "parts", None)
    if parts is not None:
        fn = lib().voltrix_launch_spmm_panel_parts_bf16 if bf16 else lib().voltrix_launch_spmm_panel_parts_f16
        fn(_ptr(plan.panel_ptr), _ptr(plan.panel_cols), _ptr(plan.panel_bits), _ptr(parts.parts), ctypes.c_int(parts.num_parts),
           _ptr(parts.xcd_ptr), ctypes.c_int(parts.max_parts_per_xcd), ctypes.c_void_p(partials_ptr),
           ctypes.c_int(plan.num_nodes), ctypes.c_int(embedding_dim), ctypes.c_void_p(input_ptr), ctypes.c_int64(input_rows),
           ctypes.c_void_p(output_ptr), ctypes.c_int(int(accumulate)), ctypes.c_int(tile[0]), ctypes.c_int(tile[1]),
           ctypes.c_int(plan.waves), ctypes.c_int(plan.row_blocks), ctypes.c_int(tile[2]), ctypes.c_int(slab_policy),
           ctypes.c_void_p(out_scale), ctypes.c_void_p(stream), ctypes.byref(rc))
        return rc.value
    fn = lib().voltrix_launch_spmm_panel_bf16 if bf16 else lib().voltrix_launch_spmm_panel_f16
    order = plan.panel_order.data_ptr() if plan.panel_order is not None else 0
    xcd_ptr = getattr(plan, "xcd_ptr", None)
    fn(_ptr(plan.panel_ptr), _ptr(plan.panel_cols), _ptr(plan.panel_bits), ctypes.c_void_p(order),
       ctypes.c_void_p(xcd_ptr.data_ptr() if xcd_ptr is not None else 0),
       ctypes.c_int(plan.max_panels_per_xcd if xcd_ptr is not None else 0), ctypes.c_int(plan.num_nodes), ctypes.c_int(embedding_dim), ctypes.c_void_p(input_ptr), ctypes.c_int64(input_rows),
       ctypes.c_void_p(output_ptr), ctypes.c_int(int(accumulate)), ctypes.c_int(tile[0]), ctypes.c_int(tile[1]),
       ctypes.c_int(plan.waves), ctypes.c_int(plan.row_blocks), ctypes.c_int(tile[2]), ctypes.c_int(slab_policy),
       ctypes.c_void_p(out_scale), ctypes.c_void_p(stream), ctypes.byref(rc))
    return rc.value


def launch_combine_panel_partials(parts, partials_ptr, output_ptr, num_nodes, embedding_dim, panel_rows, accumulate,
                                  stream) -> int:
    """``output (+)= `` the partial tiles of the cut panels, pieces in slot order (``parts`` = voltrix.hybrid.PanelParts)."""
    rc = ctypes.c_int(-1)
    lib().voltrix_launch_combine_panel_partials(_ptr(parts.cuts), ctypes.c_int(parts.num_cuts), ctypes.c_void_p(partials_ptr),
                                                ctypes.c_void_p(output_ptr), ctypes.c_int(num_nodes),
                                                ctypes.c_int(embedding_dim), ctypes.c_int(panel_rows),
                                                ctypes.c_int(int(accumulate)), ctypes.c_void_p(stream), ctypes.byref(rc))
    return rc.value


def launch_spmm_fused(plan, fused, input_ptr, output_ptr, embedding_dim, bf16, tile, out_scale, stream,
                      pace_blocks: int = 0) -> int:
    """The two-level product in one launch; ``plan`` = voltrix.hybrid.PanelPlan (8 waves x 4 row blocks), ``fused`` =
    voltrix.hybrid.FusedRecords (4 waves x 8 row blocks), ``tile`` = (fs, depth), ``pace_blocks`` = sync points per column
    sweep between the workgroups of an XCD (0: none).  Returns the return code."""
    rc = ctypes.c_int(-1)
    fn = lib().voltrix_launch_spmm_fused_bf16 if bf16 else lib().voltrix_launch_spmm_fused_f16
    order = plan.panel_order.data_ptr() if plan.panel_order is not None else 0
    xcd_ptr = getattr(plan, "xcd_ptr", None)
    fn(_ptr(plan.panel_ptr), _ptr(plan.panel_cols), _ptr(plan.panel_bits), ctypes.c_void_p(order),
       ctypes.c_void_p(xcd_ptr.data_ptr() if xcd_ptr is not None else 0),
       ctypes.c_int(plan.max_panels_per_xcd if xcd_ptr is not None else 0), _ptr(fused.wave_ptr),
       _ptr(fused.records), ctypes.c_int(plan.num_nodes), ctypes.c_int(embedding_dim), ctypes.c_void_p(input_ptr),
       ctypes.c_void_p(output_ptr), ctypes.c_int(tile[0]), ctypes.c_int(tile[1]), ctypes.c_int(int(pace_blocks)),
       ctypes.c_void_p(out_scale), ctypes.c_void_p(stream), ctypes.byref(rc))
    return rc.value


def fused_panel_geometry():
    """``(waves, row_blocks)`` of the one-launch kernel's 512-row panel, asked of the library (4 x 8 since round 4)."""
    waves, row_blocks = ctypes.c_int(0), ctypes.c_int(0)
    lib().voltrix_fused_panel_geometry(ctypes.byref(waves), ctypes.byref(row_blocks))
    assert waves.value * row_blocks.value * 16 == 512, (waves.value, row_blocks.value)
    return waves.value, row_blocks.value


def build_fused_records(blk_offsets, hspa_packed, hind, num_nodes: int, stream=None):
    """Stage records of the one-launch kernel through the library's two-phase builder (voltrix/fused_plan.hpp): returns
    ``(wave_ptr int32 [waves NP + 1], records uint32 [R + 1, 64], R)``.  One host sync."""
    import torch

    dev = blk_offsets.device
    stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
    num_waves = fused_panel_geometry()[0] * ((num_nodes + 511) // 512)
    workspace = torch.empty(max(16, int(lib().voltrix_fused_records_workspace_bytes(ctypes.c_int(num_nodes)))),
                            dtype=torch.uint8, device=dev)
    wave_ptr = torch.empty(num_waves + 1, dtype=torch.int32, device=dev)
    rc = ctypes.c_int(-1)
    lib().voltrix_launch_fused_records_count(_ptr(blk_offsets), _ptr(hspa_packed), ctypes.c_int(num_nodes), _ptr(workspace),
                                             _ptr(wave_ptr), ctypes.c_void_p(stream), ctypes.byref(rc))
    check(rc.value, "voltrix_launch_fused_records_count")
    num_records = int(wave_ptr[-1])   # the sync
    assert 0 <= num_records <= 4 * (hspa_packed.numel() // 16 + 1), num_records   # at most one record per TC block
    records = torch.empty((num_records + 1, 64), dtype=torch.int32, device=dev).view(torch.uint32)
    lib().voltrix_launch_fused_records_fill(_ptr(blk_offsets), _ptr(hspa_packed), _ptr(hind), ctypes.c_int(num_nodes),
                                            _ptr(wave_ptr), ctypes.c_int64(num_records), _ptr(records),
                                            ctypes.c_void_p(stream), ctypes.byref(rc))
    check(rc.value, "voltrix_launch_fused_records_fill")
    return wave_ptr, records, num_records


def launch_panel_order(panel_ptr, num_panels: int, order_out, stream, group: int = 1, xcd_ptr=None) -> None:
    rc = ctypes.c_int(-1)
    lib().voltrix_launch_panel_order(_ptr(panel_ptr), ctypes.c_int(num_panels), ctypes.c_int(group),
                                     ctypes.c_void_p(xcd_ptr.data_ptr() if xcd_ptr is not None else 0), _ptr(order_out),
                                     ctypes.c_void_p(stream), ctypes.byref(rc))
    check(rc.value, "voltrix_launch_panel_order")


def panel_plan_workspace_bytes(num_nodes: int, waves: int, row_blocks: int) -> int:
    return int(lib().voltrix_panel_plan_workspace_bytes(ctypes.c_int(num_nodes), ctypes.c_int(waves),
                                                        ctypes.c_int(row_blocks)))


def launch_panel_plan_count(indptr, indices, num_nodes, num_cols, waves, row_blocks, tau, workspace, panel_ptr,
                            resid_indptr, status, stream) -> int:
    rc = ctypes.c_int(-1)
    lib().voltrix_launch_panel_plan_count(_ptr(indptr), _ptr(indices), ctypes.c_int(num_nodes), ctypes.c_int(num_cols),
                                          ctypes.c_int64(indices.numel()), ctypes.c_int(waves), ctypes.c_int(row_blocks),
                                          ctypes.c_int(tau), _ptr(workspace), _ptr(panel_ptr), _ptr(resid_indptr),
                                          _ptr(status), ctypes.c_void_p(stream), ctypes.byref(rc))
    return rc.value


def launch_panel_plan_fill(indptr, indices, num_nodes, num_cols, waves, row_blocks, tau, workspace, panel_ptr,
                           resid_indptr, total_ksteps, resid_indices, panel_cols, panel_bits, stream) -> None:
    rc = ctypes.c_int(-1)
    lib().voltrix_launch_panel_plan_fill(_ptr(indptr), _ptr(indices), ctypes.c_int(num_nodes), ctypes.c_int(num_cols),
                                         ctypes.c_int64(indices.numel()), ctypes.c_int(waves), ctypes.c_int(row_blocks),
                                         ctypes.c_int(tau), _ptr(workspace), _ptr(panel_ptr), _ptr(resid_indptr),
                                         ctypes.c_int64(total_ksteps), _ptr(resid_indices), _ptr(panel_cols),
                                         _ptr(panel_bits), ctypes.c_void_p(stream), ctypes.byref(rc))
    check(rc.value, "voltrix_launch_panel_plan_fill")


def launch_window_order(blk_offsets, num_nodes, order_out, stream, chunk: int = 256) -> None:
    rc = ctypes.c_int(-1)
    lib().voltrix_launch_window_order(_ptr(blk_offsets), ctypes.c_int(num_nodes), ctypes.c_int(chunk), _ptr(order_out),
                                      ctypes.c_void_p(stream), ctypes.byref(rc))
    check(rc.value, "voltrix_launch_window_order")


def launch_cast_f32_f16(src, dst, stream) -> None:
    rc = ctypes.c_int(-1)
    lib().voltrix_launch_cast_f32_f16(_ptr(src), _ptr(dst), ctypes.c_int64(src.numel()), ctypes.c_void_p(stream),
                                      ctypes.byref(rc))
    check(rc.value, "voltrix_launch_cast_f32_f16")


_cast_scaled = None   # the entry point with its argument types set: plain ints convert in C (12 -> 4 us per call on the host)


def launch_cast_f32_f16_scaled(src, dst, scale, stream) -> None:
    """dst = fp16(src * 2^-e), scale[0] = 2^e (``scale``: float32[2] device tensor); see include/voltrix_capi.h."""
    global _cast_scaled
    if _cast_scaled is None:
        fn = lib().voltrix_launch_cast_f32_f16_scaled
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                       ctypes.POINTER(ctypes.c_int)]
        _cast_scaled = fn
    rc = ctypes.c_int(-1)
    _cast_scaled(src.data_ptr(), dst.data_ptr(), src.numel(), scale.data_ptr(), stream, rc)
    if rc.value != 0:
        check(rc.value, "voltrix_launch_cast_f32_f16_scaled")


_spmm_csr_rows = None


def launch_spmm_csr_rows(indptr, indices, num_rows: int, feat, output, stream, xcd_ranges: int = 0, values=None) -> None:
    """``output = csr(ones) @ feat`` -- or ``csr(values) @ feat`` with ``values`` (device float32 [nnz], CSR order) -- with the CSR
    row-gather kernel (device int32 CSR; fp32 / fp16 / bf16 ``feat`` whose rows are a multiple of 16 bytes; fp32 ``output``
    [num_rows, F]); see include/voltrix_capi.h."""
    import torch

    global _spmm_csr_rows, _spmm_csr_rows_weighted
    if _spmm_csr_rows is None:
        fn = lib().voltrix_launch_spmm_csr_rows
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                       ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        _spmm_csr_rows = fn
        fn = lib().voltrix_launch_spmm_csr_rows_weighted
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                       ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        _spmm_csr_rows_weighted = fn
    assert indptr.dtype == torch.int32 and indices.dtype == torch.int32 and indptr.numel() == num_rows + 1
    assert feat.dim() == 2 and feat.is_contiguous() and output.is_contiguous() and output.dtype == torch.float32
    assert output.shape == (num_rows, feat.shape[1])
    dtype = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[feat.dtype]
    rc = ctypes.c_int(-1)
    if values is None:
        _spmm_csr_rows(indptr.data_ptr(), indices.data_ptr(), num_rows, feat.shape[1], feat.data_ptr(), dtype, output.data_ptr(),
                       int(xcd_ranges), stream, rc)
        check(rc.value, "voltrix_launch_spmm_csr_rows")
        return
    assert values.dtype == torch.float32 and values.is_contiguous() and values.numel() == indices.numel() and values.is_cuda
    _spmm_csr_rows_weighted(indptr.data_ptr(), indices.data_ptr(), values.data_ptr(), num_rows, feat.shape[1], feat.data_ptr(), dtype,
                            output.data_ptr(), int(xcd_ranges), stream, rc)
    check(rc.value, "voltrix_launch_spmm_csr_rows_weighted")


_spmm_csr_rows_weighted = None
_scatter_values = None


def launch_scatter_values(values, slots, plane, stream) -> None:
    """``plane.view(-1)[slots[e]] = values[e]`` (device float32 values, int64 slots, fp32 / fp16 / bf16 plane); see
    include/voltrix_capi.h."""
    import torch

    global _scatter_values
    if _scatter_values is None:
        fn = lib().voltrix_launch_scatter_values
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                       ctypes.POINTER(ctypes.c_int)]
        _scatter_values = fn
    assert values.dtype == torch.float32 and slots.dtype == torch.int64 and values.numel() == slots.numel()
    assert values.is_contiguous() and slots.is_contiguous() and plane.is_contiguous() and values.is_cuda and plane.is_cuda
    dtype = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[plane.dtype]
    rc = ctypes.c_int(-1)
    _scatter_values(values.data_ptr(), slots.data_ptr(), plane.data_ptr(), values.numel(), dtype, stream, rc)
    check(rc.value, "voltrix_launch_scatter_values")


_scale_rows = None


def launch_scale_rows(src, scale, dst, stream) -> None:
    """dst[i, :] = src[i, :] * scale[i] (``scale`` float32 [rows]; fp32 / fp16 / bf16 rows of a 16-byte multiple; in place
    allowed); see include/voltrix_capi.h."""
    import torch

    global _scale_rows
    if _scale_rows is None:
        fn = lib().voltrix_launch_scale_rows
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                       ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        _scale_rows = fn
    assert src.dim() == 2 and src.is_contiguous() and dst.is_contiguous() and dst.shape == src.shape and dst.dtype == src.dtype
    assert scale.dtype == torch.float32 and scale.numel() == src.shape[0] and scale.is_contiguous()
    dtype = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[src.dtype]
    rc = ctypes.c_int(-1)
    _scale_rows(src.data_ptr(), scale.data_ptr(), dst.data_ptr(), src.shape[0], src.shape[1], dtype, stream, rc)
    check(rc.value, "voltrix_launch_scale_rows")


# ---- kernel-isolated timing hook (utils.KernelTimer / bench_kineto): every launch wrapper above that takes a stream is
# ---- bracketed by an event pair on that stream while a timer is active; free otherwise.
def _timed(fn, name, stream_index=None, stream_kw="stream"):
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        from .utils import KernelTimer

        if KernelTimer.active is None:
            return fn(*args, **kwargs)
        stream = kwargs.get(stream_kw)
        if stream is None and stream_index is not None and len(args) > stream_index:
            stream = args[stream_index]
        with KernelTimer.active.bracket(name, stream):
            return fn(*args, **kwargs)

    return wrapper


launch_spmm = _timed(launch_spmm, "spmm", 10)
launch_spmm_sched = _timed(launch_spmm_sched, "spmm", 9)
launch_spmm_panel = _timed(launch_spmm_panel, "spmm_panel", 8)
launch_spmm_fused = _timed(launch_spmm_fused, "spmm_fused", 8)
launch_combine_partials = _timed(launch_combine_partials, "combine_partials", 6)
launch_cast_f32_f16_scaled = _timed(launch_cast_f32_f16_scaled, "cast_f32_f16_scaled", 3)
launch_cast_f32_f16 = _timed(launch_cast_f32_f16, "cast_f32_f16", 2)
launch_scale_rows = _timed(launch_scale_rows, "scale_rows", 3)
launch_spmm_csr_rows = _timed(launch_spmm_csr_rows, "spmm_csr_rows", 5)
launch_spmm_f32_as_f16 = _timed(launch_spmm_f32_as_f16, "spmm_f32_as_f16", 9)
launch_window_order = _timed(launch_window_order, "window_order", 3)
launch_csr_window_count = _timed(launch_csr_window_count, "csr_window_count", 8)
launch_csr_fill = _timed(launch_csr_fill, "csr_fill", 8)


# ---------------------------------------------------------------------------------------------------------------------
# Cuthill-McKee row order on the device (voltrix/reorder_kernels.hpp; include/voltrix_capi.h)
def csr_transpose(indptr, indices, num_rows: int, num_cols: int, stream=None):
    """CSR of ``A^T``: int32 ``(t_indptr [num_cols + 1], t_indices [nnz])``, rows sorted, duplicates kept (a stable radix
    sort by column, voltrix/reorder_kernels.hpp).  No host sync."""
    import torch

    dev = indptr.device
    stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
    nnz = int(indices.numel())
    ws = torch.empty(max(16, int(lib().voltrix_csr_transpose_workspace_bytes(ctypes.c_int64(nnz)))), dtype=torch.uint8,
                     device=dev)
    t_indptr = torch.empty(num_cols + 1, dtype=torch.int32, device=dev)
    t_indices = torch.empty(nnz, dtype=torch.int32, device=dev)
    rc = ctypes.c_int(-1)
    lib().voltrix_launch_csr_transpose(_ptr(indptr), _ptr(indices), ctypes.c_int(num_rows), ctypes.c_int(num_cols),
                                       ctypes.c_int64(nnz), _ptr(ws), _ptr(t_indptr), _ptr(t_indices),
                                       ctypes.c_void_p(stream), ctypes.byref(rc))
    check(rc.value, "voltrix_launch_csr_transpose")
    return t_indptr, t_indices


def chol_inv_transposed(gram, eps: float = 1e-10, stream=None):
    """``inv(chol(gram + eps trace I))^T`` of a k x k float32 Gram matrix on its device (k <= 64); no host sync."""
    import torch

    k = gram.shape[0]
    assert gram.is_cuda and gram.dtype == torch.float32 and gram.shape == (k, k) and gram.is_contiguous()
    out = torch.empty_like(gram)
    stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
    rc = ctypes.c_int(-1)
    lib().voltrix_launch_chol_inv_transposed(_ptr(gram), ctypes.c_int(k), ctypes.c_double(eps), _ptr(out),
                                             ctypes.c_void_p(stream), ctypes.byref(rc))
    check(rc.value, "voltrix_launch_chol_inv_transposed")
    return out


class CmSearch:
    """State of the Cuthill-McKee search of one graph: buffers the C-ABI entries work on (all int32, on the CSR's device)."""

    def __init__(self, indptr, indices, t_indptr, t_indices, num_nodes: int, t_rows: int, tie):
        import torch

        dev = indptr.device
        self.graph = (indptr, indices, t_indptr, t_indices)
        self.n, self.t_rows, self.tie = num_nodes, t_rows, tie
        self.level = torch.full((num_nodes,), -1, dtype=torch.int32, device=dev)
        self.rank = torch.full((num_nodes,), -1, dtype=torch.int32, device=dev)
        self.queue = torch.empty(num_nodes, dtype=torch.int32, device=dev)
        self.level_off = torch.zeros(num_nodes + 2, dtype=torch.int32, device=dev)
        self.ctrl = torch.zeros(8, dtype=torch.int32, device=dev)
        self.syncs = 0

    def _graph_args(self):
        a, b, c, d = self.graph
        return (_ptr(a), _ptr(b), _ptr(c), _ptr(d), ctypes.c_int(self.n), ctypes.c_int(self.t_rows))

    def levels(self, start: int, wide_levels: int = 4, stream=None):
        """Breadth-first levels of ``start``'s component: returns ``(nodes, levels)``; ``queue[:nodes]`` holds the component
        level by level, ``level_off[:levels + 1]`` the offsets.  One host read per `wide_levels` whole-chip levels (a graph
        whose frontiers stay narrow is walked by one launch)."""
        import torch

        stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
        rc = ctypes.c_int(-1)
        lib().voltrix_launch_bfs_seed(ctypes.c_int(int(start)), ctypes.c_int(self.n), _ptr(self.level), _ptr(self.queue),
                                      _ptr(self.ctrl), _ptr(self.level_off), ctypes.c_void_p(stream), ctypes.byref(rc))
        check(rc.value, "voltrix_launch_bfs_seed")
        while True:
            lib().voltrix_launch_bfs_levels(*self._graph_args(), _ptr(self.level), _ptr(self.queue), _ptr(self.ctrl),
                                            _ptr(self.level_off), ctypes.c_int(wide_levels), ctypes.c_void_p(stream),
                                            ctypes.byref(rc))
            check(rc.value, "voltrix_launch_bfs_levels")
            ctrl = self.ctrl.tolist()       # the sync
            self.syncs += 1
            if ctrl[4]:
                return ctrl[1], ctrl[3] + 1

    def rank_component(self, levels: int, base: int, stream=None):
        """Cuthill-McKee order inside the levels of the component in ``queue`` (in place) and ``rank`` = base + position."""
        import torch

        stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
        offsets = self.level_off[:levels + 1].cpu()          # the sync
        self.syncs += 1
        sizes = (offsets[1:] - offsets[:-1])
        big = int(sizes[sizes > 1024].max()) if bool((sizes > 1024).any()) else 0
        ws = torch.empty(max(16, int(lib().voltrix_cm_rank_workspace_bytes(ctypes.c_int64(big)))), dtype=torch.uint8,
                         device=self.level.device)
        host = (ctypes.c_int * (levels + 1))(*offsets.tolist())
        rc = ctypes.c_int(-1)
        lib().voltrix_launch_cm_rank(*self._graph_args(), _ptr(self.level), _ptr(self.rank), _ptr(self.tie), _ptr(self.queue),
                                     _ptr(self.level_off), host, ctypes.c_int(levels), ctypes.c_int(int(base)), _ptr(ws),
                                     ctypes.c_void_p(stream), ctypes.byref(rc))
        check(rc.value, "voltrix_launch_cm_rank")
        return offsets
