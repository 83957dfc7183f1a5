"""Unit tables: the schedule of ``spmm_tc16_kernel`` for skewed window lengths (profiles/HISTORY.md section 3.2).

Long windows are cut into interleaved units of bounded length, listed longest first per XCD range; the partial tiles of a cut
window are summed in unit order by ``combine_partials_kernel``.  The table is built on the device by the library
(``voltrix/unit_table.hpp`` behind ``voltrix_launch_unit_table_count / _fill``); ``unit_table_torch`` is its torch-tensor
restatement (tests compare them element by element).  Results do not depend on timing: the order is data.

(The round-1 stage-list executor that used to live here lost to the hardware's own dispatch and moved to
harness/experiments/stage_list_executor/ in round 4.)
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

NUM_XCD = 8


@dataclass
class UnitTable:
    """Schedule of the window kernel in units of bounded length (spmm_kernels.hpp, SpmmArgs::units)."""
    units: torch.Tensor       # int32 [U, 4]: window, phase, stride, slot -- XCD ranges back to back, longest unit first
    unit_ptr: torch.Tensor    # int32 [9]
    cuts: torch.Tensor        # int32 [C, 4]: window, first slot, units, 0 -- one row per cut window
    max_units_per_xcd: int
    num_units: int
    num_cuts: int             # windows cut into more than one unit
    num_slots: int            # partial tiles ([16][F] fp32 each) the cut windows need
    max_stages: int


def length_order(blk_offsets: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """int32 [W]: position -> window, every XCD's window range sorted by TC-block count, longest first (ties by
    index)."""
    dev = blk_offsets.device
    num_windows = (num_nodes + 15) // 16
    idx = torch.arange(num_windows, dtype=torch.int64, device=dev)
    if num_windows == 0:
        return idx.to(torch.int32)
    nblk = (blk_offsets[1:num_windows + 1] - blk_offsets[:num_windows]).to(torch.int64)
    wpx = (num_windows + NUM_XCD - 1) // NUM_XCD
    top = int(nblk.max())
    return torch.argsort((idx // wpx) * (top + 1) + (top - nblk), stable=True).to(torch.int32)


FEW_WINDOWS = 1024        # handles with fewer windows are cut into about this many units: one per SIMD of the chip
STAGE_HIST_BINS = 65536   # the device builder's histogram counts longer windows in its last bin (unit_table.hpp)


def default_max_stages(blk_offsets: torch.Tensor, num_nodes: int) -> int:
    """1.5 x the median window length (in stages of 4 TC blocks): the measured optimum on the reddit-like graph, both for
    the window format and for the residual of the two-level format (profiles/HISTORY.md section 5).  Handles of fewer than
    ``FEW_WINDOWS`` windows (round 5): at most ``ceil(all stages / FEW_WINDOWS)`` -- a window is one wave's serial stream, and a
    few hundred long windows of one length (ddi-like) would leave most SIMDs without a wave
    (profiles/r05/experiment_few_windows.log: 1.7-2.0 x)."""
    num_windows = (num_nodes + 15) // 16
    if num_windows == 0:
        return 1
    nst = ((blk_offsets[1:num_windows + 1] - blk_offsets[:num_windows]) + 3) // 4
    bound = max(8, int(1.5 * float(nst.float().median())))
    if num_windows < FEW_WINDOWS:
        stages = int(nst.clamp(max=STAGE_HIST_BINS).sum())
        bound = min(bound, max(8, -(-stages // FEW_WINDOWS)))
    return bound


def balanced_xcd_windows(blk_offsets: torch.Tensor, num_nodes: int, align: int = 1) -> torch.Tensor:
    """int32 [9] on the handle's device: first window of every XCD's range such that the eight ranges hold about the same
    number of STAGES (the window kernel's work), boundaries on multiples of ``align`` windows.  The equal-windows split the
    kernels default to is the same thing on graphs whose rows are statistically alike (the reddit-like stand-in: +- 2 %); on
    graphs with community structure it is not (a reddit-size block model: 1.57 x the mean on the busiest XCD).  Depends on
    blk_offsets only: deterministic."""
    dev = blk_offsets.device
    num_windows = (num_nodes + 15) // 16
    out = torch.zeros(NUM_XCD + 1, dtype=torch.int64, device=dev)
    if num_windows == 0:
        return out.to(torch.int32)
    if blk_offsets.is_cuda:      # the library's builder (voltrix/schedule_tables.hpp); the lines below are its restatement
        from . import capi

        return capi.xcd_ranges_of_windows(blk_offsets, num_nodes, align)
    nst = ((blk_offsets[1:num_windows + 1] - blk_offsets[:num_windows]).to(torch.int64) + 3) // 4
    return split_equal_work(nst, align)


def split_equal_work(work: torch.Tensor, align: int = 1) -> torch.Tensor:
    """int32 [9]: boundaries b[0] = 0 <= b[1] <= ... <= b[8] = len(work) (multiples of ``align`` inside) such that the eight
    ranges of ``work`` (non-negative int64 per item) have about equal sums: b[x] = the index whose prefix sum is NEAREST to
    x / 8 of the total (the item that crosses the target goes to the side that leaves the smaller error), rounded up to
    ``align``."""
    dev = work.device
    n = work.numel()
    prefix = torch.cumsum(work.to(torch.int64), 0)
    total = int(prefix[-1]) if n else 0
    out = torch.zeros(NUM_XCD + 1, dtype=torch.int64, device=dev)
    out[NUM_XCD] = n
    if total > 0:
        targets = torch.tensor([(total * x + NUM_XCD - 1) // NUM_XCD for x in range(1, NUM_XCD)], dtype=torch.int64, device=dev)
        hi = (torch.searchsorted(prefix, targets, right=False) + 1).clamp(max=n)   # items before the boundary, crossing one in
        over = prefix[hi - 1] - targets
        under = targets - torch.where(hi >= 2, prefix[(hi - 2).clamp(min=0)], torch.zeros_like(targets))
        cut = torch.where(under < over, hi - 1, hi)
        cut = ((cut + align - 1) // align * align).clamp(max=n)
        out[1:NUM_XCD] = torch.cummax(cut, 0).values
    else:
        per = (n + NUM_XCD - 1) // NUM_XCD
        out[1:NUM_XCD] = torch.arange(1, NUM_XCD, device=dev).mul(per).clamp(max=n)
    return out.to(torch.int32)


def unit_table(blk_offsets: torch.Tensor, num_nodes: int, max_stages: int = None, chunk: int = None,
               xcd_ptr: torch.Tensor = None) -> UnitTable:
    """The handle's unit table (layout and rules: :func:`unit_table_torch`).  Built by the library's two-phase device
    builder (``voltrix/unit_table.hpp`` through ``capi.build_unit_table`` -- the same entry points a C host binds);
    the chunked listing (experiments) and CPU tensors go through the torch-tensor restatement, which the tests also use
    to check the native table element by element."""
    if chunk is not None or not blk_offsets.is_cuda:
        return unit_table_torch(blk_offsets, num_nodes, max_stages, chunk, xcd_ptr)
    from . import capi

    units, unit_ptr, cuts, head = capi.build_unit_table(blk_offsets, num_nodes, 0 if max_stages is None else max_stages,
                                                        xcd_ptr=xcd_ptr)
    return UnitTable(units, unit_ptr, cuts, head[3], head[0], head[1], head[2], head[4] if num_nodes > 0 else
                     (1 if max_stages is None else max_stages))


def unit_table_torch(blk_offsets: torch.Tensor, num_nodes: int, max_stages: int = None, chunk: int = None,
                     xcd_ptr: torch.Tensor = None) -> UnitTable:
    """Cut every window of more than ``max_stages`` stages (a stage = 4 TC blocks = one MFMA K step) into
    ``k = ceil(stages / max_stages)`` interleaved units -- unit j runs the stages j, j + k, j + 2k, ... -- so that every
    unit sweeps the window's whole (sorted) column range with at most ``max_stages`` stages.  Units of one XCD's window
    range (optionally: of every chunk of ``chunk`` consecutive windows, which keeps row neighbours together) are listed
    longest first.  No wave is left with a window several times the usual length (the tail of the launch), and units of
    equal length that start together keep pace, so co-resident waves want the same rows of B at the same time and share
    them through L2.  Cut windows leave partial tiles that ``combine_partials`` sums in unit order (deterministic).
    Built with torch tensor ops on the handle's device (plumbing, once per handle)."""
    dev = blk_offsets.device
    num_windows = (num_nodes + 15) // 16
    if max_stages is None:
        max_stages = default_max_stages(blk_offsets, num_nodes)
    assert max_stages >= 1
    if num_windows == 0:
        z = torch.zeros((0, 4), dtype=torch.int32, device=dev)
        return UnitTable(z, torch.zeros(9, dtype=torch.int32, device=dev), z, 0, 0, 0, 0, max_stages)
    nblk = (blk_offsets[1:num_windows + 1] - blk_offsets[:num_windows]).to(torch.int64)
    nst = (nblk + 3) // 4
    k = torch.clamp((nst + max_stages - 1) // max_stages, min=1)
    w = torch.repeat_interleave(torch.arange(num_windows, dtype=torch.int64, device=dev), k)
    first = torch.cumsum(k, 0) - k
    j = torch.arange(w.numel(), dtype=torch.int64, device=dev) - first[w]
    kk = k[w]
    length = (nst[w] - j + kk - 1) // kk
    # partial-tile slots: the units of a cut window take consecutive slots, in unit order
    cut = k > 1
    k_cut = torch.where(cut, k, torch.zeros_like(k))
    slot_first = torch.cumsum(k_cut, 0) - k_cut
    slot = torch.where(cut[w], slot_first[w] + j, torch.full_like(w, -1))
    wpx = (num_windows + NUM_XCD - 1) // NUM_XCD
    if xcd_ptr is not None:      # ranges of equal work (balanced_xcd_windows): XCD of window w = ranges whose start is <= w
        assert chunk is None
        xcd = torch.searchsorted(xcd_ptr.to(torch.int64)[1:NUM_XCD].contiguous(), w, right=True)
    else:
        xcd = w // wpx
    group = xcd if chunk is None else xcd * (wpx // max(1, chunk) + 2) + (w - xcd * wpx) // max(1, chunk)
    top = int(length.max())
    order = torch.argsort(group * (top + 1) + (top - length), stable=True)
    units = torch.stack([w[order], j[order], kk[order], slot[order]], dim=1).to(torch.int32).contiguous()
    counts = torch.bincount(xcd, minlength=NUM_XCD)
    unit_ptr = torch.zeros(NUM_XCD + 1, dtype=torch.int64, device=dev)
    unit_ptr[1:] = torch.cumsum(counts, 0)
    cw = torch.nonzero(cut).flatten()
    cuts = torch.stack([cw, slot_first[cw], k[cw], torch.zeros_like(cw)], dim=1).to(torch.int32).contiguous()
    return UnitTable(units, unit_ptr.to(torch.int32), cuts, int(counts.max()), int(w.numel()), int(cw.numel()),
                     int(k_cut.sum()), max_stages)


# ---- stream tables: the schedule of spmm_stream_kernel (short windows; voltrix/spmm_stream_kernels.hpp) ---------------------
@dataclass
class StreamTable:
    """Runs of consecutive units for the stream kernel: a wave walks the stages of a run as one stream."""
    units: torch.Tensor       # int32 [U, 8]: first TC block, end TC block of the window, TC blocks between two stages (4 x stride),
                              #   window, partial-tile slot or -1, stages, columns of the window's last TC block that carry an
                              #   edge (1 .. 8; 0 = a window without edges), 0 -- in window order
    runs: torch.Tensor        # int32 [R, 4]: first unit, units (<= 64), stages, 0
    run_ptr: torch.Tensor     # int32 [9]: the runs of every XCD (contiguous windows, equal work)
    cuts: torch.Tensor        # int32 [C, 4]: window, first slot, units, 0 (combine_partials)
    max_runs_per_xcd: int
    num_runs: int
    num_units: int
    num_cuts: int
    num_slots: int
    run_cost: int
    cut_stages: int


STREAM_WAVE_SLOTS = 256 * 6      # waves the chip holds at the default ring depth (6 per CU)
STREAM_RUNS_PER_SLOT = 6         # runs every wave slot gets on average: the tail of the launch is a fraction of one run


def last_block_columns(blk_offsets: torch.Tensor, hspa_packed: torch.Tensor, hind: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """int64 [W]: how many of the 8 condensed columns of every window's LAST TC block carry an edge.  Every other block of a
    window is full, padded ``hind`` slots are 0 and real columns ascend, so the count is 1 + the non-zero slots after the first;
    a window without edges (one all-zero block, the reference's quirk) counts 0."""
    num_windows = (num_nodes + 15) // 16
    off = blk_offsets[:num_windows + 1].to(torch.int64)
    last = off[1:] - 1
    h = hind.view(-1, 8)[last]
    ncl = 1 + (h[:, 1:] > 0).sum(1)
    bits = hspa_packed.view(torch.int32).view(-1, 4)[last]
    empty = (bits == 0).all(1)
    return torch.where(empty, torch.zeros_like(ncl), ncl)


def stream_tables(blk_offsets: torch.Tensor, hspa_packed: torch.Tensor, hind: torch.Tensor, num_nodes: int, run_cost: int = None,
                  cut_stages: int = None) -> StreamTable:
    """The handle's stream table (layout and rules: :func:`stream_tables_torch`).  Built by the library's two-phase device
    builder (``voltrix/stream_table.hpp`` through ``capi.build_stream_table`` -- the entry points a C host binds); CPU tensors go
    through the torch-tensor restatement, which the tests also use to check the native tables element by element."""
    if not blk_offsets.is_cuda or num_nodes == 0:
        return stream_tables_torch(blk_offsets, hspa_packed, hind, num_nodes, run_cost, cut_stages)
    from . import capi

    units, runs, run_ptr, cuts, head = capi.build_stream_table(blk_offsets, hspa_packed, hind, num_nodes, run_cost or 0,
                                                               cut_stages or 0)
    return StreamTable(units, runs, run_ptr, cuts, head[4], head[3], head[0], head[1], head[2], head[5], head[6])


def stream_tables_torch(blk_offsets: torch.Tensor, hspa_packed: torch.Tensor, hind: torch.Tensor, num_nodes: int,
                        run_cost: int = None, cut_stages: int = None) -> StreamTable:
    """The stream table as torch tensor ops (the definition the device builder is checked against).  A unit = a whole window, or -- windows longer than ``cut_stages`` stages -- one of its
    ``k = ceil(stages / cut_stages)`` interleaved pieces (unit j runs the stages j, j + k, ...; partial tiles summed in unit
    order by ``combine_partials``, exactly as in :func:`unit_table_torch`).  Units stay in WINDOW order (consecutive windows'
    metadata and rows of C are consecutive in memory; band graphs gather overlapping rows of B).  A unit costs its stages + 1
    (the store of its 16 rows); the eight XCD ranges hold equal cost; inside a range, a run = the units whose cost prefix falls
    into the same bucket of ``run_cost``, so a run costs < run_cost + the cost of one unit and has <= 64 units.
    Defaults: ``run_cost`` such that every wave slot of the chip gets about six runs (6 .. 48), ``cut_stages`` = the larger of
    ``run_cost`` and 1.5 x the median window (graphs of long windows: one window per run, only the tail is cut).
    Torch tensor ops on the handle's device: once per handle; depends on ``blk_offsets`` only (deterministic)."""
    dev = blk_offsets.device
    num_windows = (num_nodes + 15) // 16
    z4 = torch.zeros((0, 4), dtype=torch.int32, device=dev)
    if num_windows == 0:
        return StreamTable(torch.zeros((0, 8), dtype=torch.int32, device=dev), z4, torch.zeros(9, dtype=torch.int32, device=dev),
                           z4, 0, 0, 0, 0, 0, run_cost or 6, cut_stages or 8)
    off = blk_offsets[:num_windows + 1].to(torch.int64)
    nblk = off[1:] - off[:-1]
    nst = (nblk + 3) // 4
    if run_cost is None:
        total = int(nst.sum()) + num_windows
        run_cost = int(min(48, max(6, total // (STREAM_WAVE_SLOTS * STREAM_RUNS_PER_SLOT))))
    assert 2 <= run_cost <= 128, "a run holds at most 64 units"
    if cut_stages is None:
        cut_stages = max(run_cost, default_max_stages(blk_offsets, num_nodes))
    assert cut_stages >= 1
    k = torch.clamp((nst + cut_stages - 1) // cut_stages, min=1)
    w = torch.repeat_interleave(torch.arange(num_windows, dtype=torch.int64, device=dev), k)
    first_unit_of_window = torch.cumsum(k, 0) - k
    j = torch.arange(w.numel(), dtype=torch.int64, device=dev) - first_unit_of_window[w]
    kk = k[w]
    length = (nst[w] - j + kk - 1) // kk
    cut = k > 1
    k_cut = torch.where(cut, k, torch.zeros_like(k))
    slot_first = torch.cumsum(k_cut, 0) - k_cut
    slot = torch.where(cut[w], slot_first[w] + j, torch.full_like(w, -1))
    zero = torch.zeros_like(w)
    ncl = last_block_columns(blk_offsets, hspa_packed, hind, num_nodes)
    units = torch.stack([off[w] + 4 * j, off[w + 1], 4 * kk, w, slot, length, ncl[w], zero], dim=1).to(torch.int32).contiguous()
    cost = length + 1
    ub = split_equal_work(cost).to(torch.int64)                   # unit boundaries of the XCD ranges
    c_before = torch.cumsum(cost, 0) - cost
    idx = torch.arange(w.numel(), dtype=torch.int64, device=dev)
    xcd = torch.searchsorted(ub[1:NUM_XCD].contiguous(), idx, right=True)
    base = c_before[ub[:NUM_XCD].clamp(max=w.numel() - 1)][xcd]
    key = xcd * (int(c_before[-1]) // run_cost + 2) + (c_before - base) // run_cost
    new_run = torch.ones_like(key, dtype=torch.bool)
    new_run[1:] = key[1:] != key[:-1]
    run_id = torch.cumsum(new_run.to(torch.int64), 0) - 1
    num_runs = int(run_id[-1]) + 1
    first = torch.nonzero(new_run).flatten()
    count = torch.bincount(run_id, minlength=num_runs)
    stages = torch.zeros(num_runs, dtype=torch.int64, device=dev).index_add_(0, run_id, length)
    assert int(count.max()) <= 64
    runs = torch.stack([first, count, stages, torch.zeros_like(first)], dim=1).to(torch.int32).contiguous()
    run_ptr = torch.zeros(NUM_XCD + 1, dtype=torch.int64, device=dev)
    run_ptr[1:] = torch.cumsum(torch.bincount(xcd[first], minlength=NUM_XCD), 0)
    cw = torch.nonzero(cut).flatten()
    cuts = torch.stack([cw, slot_first[cw], k[cw], torch.zeros_like(cw)], dim=1).to(torch.int32).contiguous()
    per_xcd = run_ptr[1:] - run_ptr[:-1]
    return StreamTable(units, runs, run_ptr.to(torch.int32), cuts, int(per_xcd.max()), num_runs, int(w.numel()),
                       int(cw.numel()), int(k_cut.sum()), run_cost, cut_stages)
