"""Unit tables: the schedule of ``spmm_tc16_kernel`` for skewed window lengths.

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


def default_max_stages(blk_offsets: torch.Tensor, num_nodes: int) -> int:
    """1.5 x the median window length (in stages of 4 TC blocks): the measured optimum on the reddit-like graph, both for
    the window format and for the residual of the two-level format."""
    num_windows = (num_nodes + 15) // 16
    if num_windows == 0:
        return 1
    nst = ((blk_offsets[1:num_windows + 1] - blk_offsets[:num_windows]) + 3) // 4
    return max(8, int(1.5 * float(nst.float().median())))


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
