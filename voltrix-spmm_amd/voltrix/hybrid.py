"""Two-level condensed format: row-panel "shared columns" + the reference's window format for the rest.

No reference counterpart (the reference leans on externally reordered inputs to raise the
TC-block fill; this raises the *reuse* of every gathered row instead).  A column that is referenced by at least ``tau``
rows of a ``panel_rows``-row panel is gathered ONCE per panel by ``spmm_panel_kernel`` (spmm_panel_kernels.hpp) and
multiplied into all the panel's 16-row blocks on the matrix cores; every other edge stays in a residual CSR that goes
through the unchanged ``csr_preprocess`` -> ``spmm`` path.  ``C = A_resid @ B + A_shared @ B`` -- two addends per
element.

Host API:

    handle = voltrix.csr_preprocess(indptr, indices, num_nodes)      # the reference handle of the WHOLE matrix, byte for
    out = voltrix.spmm(*handle, num_nodes, num_edges, feat)          # byte; when the graph pays for it the two-level form
                                                                     # rides along as an acceleration side-car that only
                                                                     # voltrix.spmm uses (any other consumer of the three
                                                                     # tensors computes the same product from them)
    two = voltrix.csr_preprocess_hybrid(indptr, indices, num_nodes)  # explicit: a TwoLevelHandle (residual + plan)
    out = voltrix.spmm_two_level(two, feat)

Plan layout: see spmm_panel_kernels.hpp; pinned bit-exactly by ``oracle/oracle_np.py::panel_plan``.
"""
from __future__ import annotations

import contextlib
import contextvars
import dataclasses
import os

import torch

from . import capi

# (waves, row_blocks) of the plan <-> PanelTile<FS, DEPTH, WAVES, RB, KS>; panel_rows = waves * row_blocks * 16
DEFAULT_WAVES = 8
DEFAULT_ROW_BLOCKS = 4
DEFAULT_TAU = 3
KSTEP = 32
# the one-launch kernel's own split of a 512-row panel (spmm_fused_kernels.hpp): four waves (one per SIMD) x eight row blocks
FUSED_WAVES = 4
FUSED_ROW_BLOCKS = 8


@dataclasses.dataclass
class PanelParts:
    """Launch table of the panel kernel in PIECES of bounded length (``panel_parts``; spmm_panel_kernels.hpp, PanelArgs::parts)."""
    parts: torch.Tensor       # int32 [P, 4]: panel, first k-step inside the panel, k-steps, slot (-1: the panel is whole)
    cuts: torch.Tensor        # int32 [C, 4]: panel, first slot, pieces, 0 -- one row per cut panel
    xcd_ptr: torch.Tensor     # int32 [9]: XCD x owns the positions [xcd_ptr[x], xcd_ptr[x + 1]) of ``parts``, longest first
    max_parts_per_xcd: int
    num_parts: int
    num_cuts: int
    num_slots: int            # partial tiles ([panel rows][F] fp32 each) a call needs
    cap: int                  # no piece is longer than this many k-steps


@dataclasses.dataclass
class PanelPlan:
    panel_ptr: torch.Tensor      # int32 [NP+1]
    panel_cols: torch.Tensor     # int32 [32 * (S + 2)]
    panel_bits: torch.Tensor     # uint32 [(S + 1) * waves * 64]
    panel_order: torch.Tensor    # int32 [NP] (launch position -> panel) or None = natural order
    num_nodes: int
    waves: int
    row_blocks: int
    tau: int
    num_ksteps: int
    num_shared_edges: int
    num_resid_edges: int
    xcd_ptr: torch.Tensor = None      # int32 [9] or None: XCD x owns the launch positions [xcd_ptr[x], xcd_ptr[x + 1]) -- ranges of
    max_panels_per_xcd: int = 0       # equal WORK (balance_xcd_ranges); None: ranges of ceil(NP / 8) positions
    parts: PanelParts = None          # long panels cut into pieces of bounded length (panel_parts), or None: one workgroup per panel

    @property
    def panel_rows(self) -> int:
        return self.waves * self.row_blocks * 16

    @property
    def num_panels(self) -> int:
        return (self.num_nodes + self.panel_rows - 1) // self.panel_rows


@dataclasses.dataclass(eq=False)
class TwoLevelHandle:
    """The two-level format as an explicit object: the reference-format tensors of the RESIDUAL matrix plus the panel
    plan of the shared columns.  Deliberately not a tuple: the three tensors alone describe only part of the matrix, so
    they cannot be unpacked into ``voltrix.spmm`` / ``spmm_kernel`` / the C-ABI by accident."""
    blk_offsets: torch.Tensor    # int32 [W+1]   residual matrix
    hspa_packed: torch.Tensor    # uint32 [4T]
    hind: torch.Tensor           # int32 [8T]
    plan: PanelPlan
    num_nodes: int
    num_edges: int               # of the whole matrix (shared + residual)
    hash_tag: str = None         # tuner key of the residual launches (like hspa_packed.hash_tag in the reference)
    fused: "FusedRecords" = None  # the residual re-packed for the one-launch kernel (spmm_fused_kernels.hpp), or None
    window_xcd_ptr: torch.Tensor = None   # int32 [9]: the residual's unit table uses the panel kernel's XCD ranges (x 32 windows)
    format_choice: dict = dataclasses.field(default_factory=dict)   # (width, dtype) -> "two-level" | "window" (voltrix.spmm, auto mode)

    @property
    def residual(self):
        return self.blk_offsets, self.hspa_packed, self.hind


@dataclasses.dataclass
class FusedRecords:
    """The residual matrix as per-wave streams of 256-byte stage records for ``spmm_fused_kernel`` (one launch for the
    whole two-level product, spmm_fused_kernels.hpp).  Layout pinned by ``oracle/oracle_np.py::fused_records``."""
    wave_ptr: torch.Tensor       # int32 [4 NP + 1]: first record of (panel, wave); wave v owns windows 32 p + 8 v + j
    records: torch.Tensor        # uint32 [R + 1, 64] (one record of padding)
    num_records: int

    def nbytes(self) -> int:
        return self.wave_ptr.numel() * 4 + self.records.numel() * 4


RECORD_WORDS = 64


def build_fused_records_torch(blk_offsets: torch.Tensor, hspa_packed: torch.Tensor, hind: torch.Tensor, num_nodes: int,
                              waves: int = FUSED_WAVES, row_blocks: int = FUSED_ROW_BLOCKS) -> FusedRecords:
    """Block-format handle of the residual -> ``FusedRecords`` with torch tensor ops on the handle's device (a stable sort
    of the stages by (wave, first column, row block) + gathers).  The HIP builder (fused_plan.hpp) produces the same bytes;
    this form also runs on the CPU and is what the tests compare both against the plain-loop definition with."""
    assert blk_offsets.dtype == torch.int32 and hind.dtype == torch.int32 and hspa_packed.dtype == torch.uint32
    device = blk_offsets.device
    num_windows = (num_nodes + 15) // 16
    panel_rows = waves * row_blocks * 16
    num_waves = waves * ((num_nodes + panel_rows - 1) // panel_rows)
    bo = blk_offsets.to(torch.int64)
    kb0, kb1 = bo[:-1], bo[1:]
    nblk = kb1 - kb0
    packed = hspa_packed.view(torch.int32).view(-1, 4)
    hind64 = hind
    total_blocks = packed.shape[0]
    # the reference's empty-window quirk: one all-zero block, nothing to multiply
    zero_first = (packed[kb0.clamp(max=max(total_blocks - 1, 0))] == 0).all(dim=1) if total_blocks else nblk < 0
    nstage = torch.where((nblk == 1) & zero_first, torch.zeros_like(nblk), (nblk + 3) // 4)
    win = torch.repeat_interleave(torch.arange(num_windows, device=device), nstage)
    first_stage = torch.cumsum(nstage, 0) - nstage
    sidx = torch.arange(win.numel(), device=device) - first_stage[win]
    sb = kb0[win] + 4 * sidx
    first_col = hind64[8 * sb].to(torch.int64)
    ncols = int(first_col.max()) + 1 if win.numel() else 1
    order = torch.argsort(((win // row_blocks) * ncols + first_col) * row_blocks + win % row_blocks, stable=True)
    win, sb = win[order], sb[order]
    wave_ptr = torch.zeros(num_waves + 1, dtype=torch.int64, device=device)
    wave_ptr[1:] = torch.cumsum(torch.bincount(win // row_blocks, minlength=num_waves), 0)
    num_records = int(win.numel())
    records = torch.zeros((num_records + 1, RECORD_WORDS), dtype=torch.int32, device=device)
    chunk = 1 << 20          # bounded temporaries (int64 [chunk, 32])
    for lo in range(0, num_records, chunk):
        hi = min(lo + chunk, num_records)
        w_, sb_ = win[lo:hi], sb[lo:hi]
        end = kb1[w_][:, None]
        k = torch.arange(32, device=device)[None, :]
        blk = sb_[:, None] + k // 8
        inwin = blk < end
        blk = torch.where(inwin, blk, kb0[w_][:, None])
        c = k % 8
        words = packed[blk, 2 * (c >> 2)] | packed[blk, 2 * (c >> 2) + 1]
        colmask = ((0x11111111 << (c & 3)) & 0xFFFFFFFF).to(torch.int64)
        used = inwin & (((words.to(torch.int64) & 0xFFFFFFFF) & colmask) != 0)
        safe = hind64[8 * kb0[w_]][:, None]
        records[lo:hi, :32] = torch.where(used, hind64[8 * blk + c], safe)
        t = torch.arange(16, device=device)[None, :]
        blk = sb_[:, None] + t // 4
        inwin = blk < end
        blk = torch.where(inwin, blk, kb0[w_][:, None])
        records[lo:hi, 32:48] = torch.where(inwin, packed[blk, t % 4], torch.zeros((), dtype=torch.int32, device=device))
        records[lo:hi, 48] = (w_ % row_blocks).to(torch.int32)
    return FusedRecords(wave_ptr=wave_ptr.to(torch.int32), records=records.view(torch.uint32), num_records=num_records)


def build_fused_records(blk_offsets: torch.Tensor, hspa_packed: torch.Tensor, hind: torch.Tensor, num_nodes: int) -> FusedRecords:
    """Block-format handle of the residual (on the GPU) -> ``FusedRecords`` with the library's HIP builder
    (fused_plan.hpp, two launches around the one host sync that sizes the output); same bytes as
    :func:`build_fused_records_torch` (tests/test_gpu_fused.py)."""
    assert blk_offsets.is_cuda and hspa_packed.is_cuda and hind.is_cuda
    assert capi.fused_panel_geometry() == (FUSED_WAVES, FUSED_ROW_BLOCKS), "library / package out of step"
    wave_ptr, records, num_records = capi.build_fused_records(blk_offsets, hspa_packed, hind, num_nodes)
    return FusedRecords(wave_ptr=wave_ptr, records=records, num_records=num_records)


def split_shared_columns(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, num_cols: int, panel_rows: int,
                         tau: int):
    """Device CSR -> ``(resid_indptr int32, resid_indices int32, pc int64 [U], rp int64 [Es], inv int64 [Es])``:
    the residual CSR (sorted, duplicate-free rows), the sorted shared (panel * num_cols + col) keys, and per shared edge
    its row inside the panel and the index of its key."""
    device = indptr.device
    n, p = num_nodes, panel_rows
    deg = (indptr[1:] - indptr[:-1]).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(n, device=device, dtype=torch.int64), deg)
    cols = indices.to(torch.int64)
    key = ((rows // p) * num_cols + cols) * p + (rows % p)      # (panel, col, row in panel)
    del rows, cols
    key = torch.unique(key, sorted=True)                        # sorts; duplicate (row, col) entries count once
    pc_all = key // p
    uniq, inverse, counts = torch.unique_consecutive(pc_all, return_inverse=True, return_counts=True)
    shared_u = counts >= tau
    shared_e = shared_u[inverse]
    # residual CSR: back to (row, col) order
    rk = key[~shared_e]
    r_pc = rk // p
    r_row = (r_pc // num_cols) * p + (rk % p)
    r_key = torch.sort(r_row * num_cols + (r_pc % num_cols)).values
    resid_rows = r_key // num_cols
    resid_indices = (r_key % num_cols).to(torch.int32)
    resid_indptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    resid_indptr[1:] = torch.cumsum(torch.bincount(resid_rows, minlength=n), 0)
    # shared part
    new_index = torch.cumsum(shared_u.to(torch.int64), 0) - 1   # unique key -> index among the shared keys
    sk = key[shared_e]
    return (resid_indptr.to(torch.int32), resid_indices, uniq[shared_u], sk % p, new_index[inverse[shared_e]])


MAX_PLAN_COLS = 1 << 22  # the HIP builder counts per column range of 2^16 (panel_plan.hpp); larger universes: no plan
MAX_PLAN_EDGE_PASSES = 2 * 10 ** 10  # ... and re-reads a panel's edges once per range: (ranges x edges) above this: no plan


def canonical_csr(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, num_cols: int):
    """Rows sorted, duplicate (row, col) entries dropped (they count once: reference quirk 5).  Only used when the
    device-side input check of the builder reports unsorted / duplicate rows."""
    deg = (indptr[1:] - indptr[:-1]).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(num_nodes, device=indptr.device, dtype=torch.int64), deg)
    key = torch.unique(rows * num_cols + indices.to(torch.int64), sorted=True)
    rows = key // num_cols
    out_indptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=indptr.device)
    out_indptr[1:] = torch.cumsum(torch.bincount(rows, minlength=num_nodes), 0)
    return out_indptr.to(torch.int32), (key - rows * num_cols).to(torch.int32)


def empty_plan(num_nodes, waves, row_blocks, tau, device, num_edges):
    np_ = (num_nodes + waves * row_blocks * 16 - 1) // (waves * row_blocks * 16)
    return PanelPlan(panel_ptr=torch.zeros(np_ + 1, dtype=torch.int32, device=device),
                     panel_cols=torch.zeros(2 * KSTEP, dtype=torch.int32, device=device),
                     panel_bits=torch.zeros(waves * 64, dtype=torch.int32, device=device).view(torch.uint32),
                     panel_order=None, num_nodes=num_nodes, waves=waves, row_blocks=row_blocks, tau=tau, num_ksteps=0,
                     num_shared_edges=0, num_resid_edges=num_edges)


# a k-step of the panel kernel costs about this many residual stages of the window kernel (reddit-like pair, each kernel
# alone: 182.6 k k-steps in 0.70 ms, 1.59 M stages in 0.92 ms -- 0.98 us against 0.148 us per CU)
KSTEP_COST_IN_STAGES = 6.6
KSTEP_COST_X10 = 66      # the same figure in tenths: the builders work in integers


def balance_xcd_ranges(two: "TwoLevelHandle") -> None:
    """Round 4: the XCD ranges of BOTH kernels of the two-level step from the work, not from the row count.  Every XCD used
    to own NP / 8 consecutive panels (and the residual's windows of the same rows); on graphs whose rows are statistically
    alike that is an equal split of the work (reddit-like: +- 2 %), on graphs with community structure it is not -- a panel
    inside a 17 k-node community carries 540 k-steps, one inside a 900-node community 30: the busiest XCD of the reddit-size
    block model held 1.57 x the mean (profiles/r04).  Work of panel p = KSTEP_COST_IN_STAGES x its k-steps + the residual
    stages of its 32 windows; boundaries where the prefix sum crosses x / 8 of the total; the plan's launch order is rebuilt
    longest-first inside the new ranges and the residual handle's unit table takes the same ranges (x 32 windows: a panel's
    rows stay on one XCD for both kernels).  Speed only -- same bits with any ranges.  One tiny host read (the longest range
    sizes the panel kernel's grid)."""
    plan = two.plan
    if plan.num_ksteps == 0 or plan.num_panels < 2 * 8:
        return
    cost_x10 = max(1, KSTEP_COST_X10 * plan.panel_rows // 512)   # a k-step's MFMAs scale with the panel's rows
    if plan.panel_ptr.is_cuda:      # the library's builder (schedule_tables.hpp: the entry point a C host binds)
        xcd_ptr, two.window_xcd_ptr = capi.xcd_ranges_of_panels(plan.panel_ptr, two.blk_offsets, two.num_nodes, plan.panel_rows,
                                                                cost_x10)
    else:
        xcd_ptr, two.window_xcd_ptr = xcd_ranges_of_panels_torch(plan.panel_ptr, two.blk_offsets, two.num_nodes,
                                                                 plan.panel_rows, cost_x10)
    longest_range, longest_panel = torch.stack([(xcd_ptr[1:] - xcd_ptr[:-1]).max(),
                                                (plan.panel_ptr[1:] - plan.panel_ptr[:-1]).max()]).tolist()   # the host read
    plan.xcd_ptr = xcd_ptr
    plan.max_panels_per_xcd = int(longest_range)
    plan.panel_order = longest_first_order(plan.panel_ptr, xcd_ptr=xcd_ptr)
    cap = default_part_cap(plan.num_ksteps)
    if PANEL_PART_FACTOR > 0 and longest_panel > cap:          # pieces only where a panel is too long
        plan.parts = panel_parts(plan.panel_ptr, cap, xcd_ptr)


def xcd_ranges_of_panels_torch(panel_ptr: torch.Tensor, resid_blk_offsets: torch.Tensor, num_nodes: int, panel_rows: int,
                               kstep_cost_x10: int):
    """Torch-tensor restatement of ``voltrix_launch_xcd_ranges_of_panels`` (tests compare them): ``(xcd_ptr,
    window_xcd_ptr)``.  Work of a panel = round-half-even(k-steps x cost / 10) + the stages of its windows, integers
    throughout."""
    from .schedule import split_equal_work

    dev = panel_ptr.device
    num_panels = panel_ptr.numel() - 1
    windows_per_panel = panel_rows // 16
    num_windows = (num_nodes + 15) // 16
    nst = ((resid_blk_offsets[1:num_windows + 1] - resid_blk_offsets[:num_windows]).to(torch.int64) + 3) // 4
    pad = num_panels * windows_per_panel - num_windows
    stages = torch.cat([nst, torch.zeros(pad, dtype=torch.int64, device=dev)]).view(num_panels, windows_per_panel).sum(1)
    tenths = (panel_ptr[1:] - panel_ptr[:-1]).to(torch.int64) * kstep_cost_x10
    q, r = tenths // 10, tenths % 10
    q = q + ((r > 5) | ((r == 5) & (q % 2 == 1))).to(torch.int64)
    xcd_ptr = split_equal_work(q + stages)
    return xcd_ptr, (xcd_ptr.to(torch.int64) * windows_per_panel).clamp(max=num_windows).to(torch.int32)


# A panel's k-steps are walked by ONE workgroup: panels longer than this multiple of a CU's fair share of the k-steps
# (S / 256) are cut.  Measured on MI355X (profiles/r04/experiment_panel_parts.log, experiment_schedule_ab.log; DESIGN.md
# section 3.3), step in ms without the table -> 1.0 / 0.75 / 0.5: block model 1.236 -> 1.055 / 1.039 / 1.154, block model
# shuffled + spectral order 1.275 -> 1.173 / 1.176, reddit-like shuffled + spectral order 1.427 -> 1.433 / 1.457 (65 panels
# cut for nothing), headline graph 1.353 -> 1.357 / 1.362 (0 / 2 panels cut).  Shorter pieces balance better, but every piece of
# a cut panel writes and re-reads a 256-KiB tile and the combine pass runs after the join: 1.0 cuts only the panels that ARE
# the critical path.
PANEL_PART_FACTOR = 1.0
NUM_CUS = 256


def default_part_cap(num_ksteps: int) -> int:
    return max(32, int(PANEL_PART_FACTOR * num_ksteps / NUM_CUS))


def panel_parts(panel_ptr: torch.Tensor, cap: int, panel_xcd_ptr: torch.Tensor = None) -> PanelParts:
    """The panel kernel's piece table (layout and rules: :func:`panel_parts_torch`).  Built by the library's two-phase
    device builder (``voltrix/schedule_tables.hpp`` through ``capi.build_panel_parts`` -- the entry points a C host binds);
    CPU tensors go through the torch-tensor restatement, which the tests also use to check the native table element by
    element."""
    if not panel_ptr.is_cuda:
        return panel_parts_torch(panel_ptr, cap, panel_xcd_ptr)
    assert cap >= 1
    parts, part_xcd_ptr, cuts, head = capi.build_panel_parts(panel_ptr, cap, panel_xcd_ptr)
    return PanelParts(parts, cuts, part_xcd_ptr, head[3], head[0], head[1], head[2], cap)


def panel_parts_torch(panel_ptr: torch.Tensor, cap: int, panel_xcd_ptr: torch.Tensor = None) -> PanelParts:
    """Cut every panel of more than ``cap`` k-steps into ``k = ceil(k-steps / cap)`` CONTIGUOUS pieces of nearly equal length
    (the panel's columns are sorted: a piece sweeps a contiguous column range).  Pieces are listed per XCD range (the panel's
    range: ``panel_xcd_ptr`` in panel units, or NP / 8 panels each), longest first.  The pieces of a cut panel take
    consecutive partial-tile slots in k-step order and ``combine_panel_partials`` adds them in that order: a fixed
    summation order whatever the pieces' timing.  Whole panels keep slot -1 (their tile goes straight to C, as without
    the table).  Torch tensor ops on ``panel_ptr``'s device (plumbing, once per handle; one host read for the sizes)."""
    assert cap >= 1
    dev = panel_ptr.device
    num_panels = panel_ptr.numel() - 1
    nks = (panel_ptr[1:] - panel_ptr[:-1]).to(torch.int64)
    k = torch.clamp((nks + cap - 1) // cap, min=1)
    p = torch.repeat_interleave(torch.arange(num_panels, dtype=torch.int64, device=dev), k)
    first = torch.cumsum(k, 0) - k
    j = torch.arange(p.numel(), dtype=torch.int64, device=dev) - first[p]
    base, rem = nks[p] // k[p], nks[p] % k[p]
    length = base + (j < rem).to(torch.int64)
    begin = j * base + torch.minimum(j, rem)
    cut = k > 1
    k_cut = torch.where(cut, k, torch.zeros_like(k))
    slot_first = torch.cumsum(k_cut, 0) - k_cut
    slot = torch.where(cut[p], slot_first[p] + j, torch.full_like(p, -1))
    if panel_xcd_ptr is not None:
        xcd = torch.searchsorted(panel_xcd_ptr.to(torch.int64)[1:8].contiguous(), p, right=True)
    else:
        xcd = p // max(1, (num_panels + 7) // 8)
    top = int(length.max()) if p.numel() else 0
    order = torch.argsort(xcd * (top + 1) + (top - length), stable=True)
    parts = torch.stack([p[order], begin[order], length[order], slot[order]], dim=1).to(torch.int32).contiguous()
    counts = torch.bincount(xcd, minlength=8)
    xcd_ptr = torch.zeros(9, dtype=torch.int64, device=dev)
    xcd_ptr[1:] = torch.cumsum(counts, 0)
    cp = torch.nonzero(cut).flatten()
    cuts = torch.stack([cp, slot_first[cp], k[cp], torch.zeros_like(cp)], dim=1).to(torch.int32).contiguous()
    return PanelParts(parts, cuts, xcd_ptr.to(torch.int32), int(counts.max()) if p.numel() else 0, int(p.numel()),
                      int(cp.numel()), int(k_cut.sum()), cap)


def longest_first_order(panel_ptr: torch.Tensor, group: int = 1, xcd_ptr: torch.Tensor = None) -> torch.Tensor:
    """Launch order of the panel kernel: int32 [NP], position -> panel; inside every XCD's contiguous range of positions
    (spmm_panel_kernel: blockIdx % 8 picks the range) GROUPS of ``group`` consecutive panels, the groups with the most k-steps
    first, the panels of a group in their natural order; ``group`` = 1 (shipped): plain longest-first.  455 workgroups on
    256 CUs is 1.8 rounds: started in natural order, the last round's long panels finish alone (two-level step 1.59 ms),
    longest-first leaves the short ones for the end (1.42 ms, profiles/r02/experiment_tau_reddit.log).  Groups of neighbours
    launched side by side share their band columns through L2: 1.323 -> 1.282 ms for the bare kernel pair
    (experiment_panel_groups.log), but nothing through the operator (1.357 vs 1.360 ms, bench_ab_panel_group.txt: the zero
    fill of C between the steps costs the window kernel more than the panel kernel gains), so the operator keeps group = 1.
    Speed only."""
    num_panels = panel_ptr.numel() - 1
    if num_panels <= 0:
        return torch.zeros(0, dtype=torch.int32, device=panel_ptr.device)
    if panel_ptr.is_cuda:   # the library's kernel (panel_plan.hpp::panel_order_kernel; same order, checked by the tests)
        order = torch.empty(num_panels, dtype=torch.int32, device=panel_ptr.device)
        capi.launch_panel_order(panel_ptr, num_panels, order, torch.cuda.current_stream().cuda_stream, group, xcd_ptr)
        return order
    nks = (panel_ptr[1:] - panel_ptr[:-1]).to(torch.int64)
    per_xcd = (num_panels + 7) // 8
    idx = torch.arange(num_panels, device=panel_ptr.device)
    if xcd_ptr is not None:          # ranges of equal work: XCD of a panel, and its first position
        assert group == 1
        xcd = torch.searchsorted(xcd_ptr.to(torch.int64)[1:8].contiguous(), idx, right=True)
        top = int(nks.max())
        return torch.argsort(xcd * (top + 1) + (top - nks), stable=True).to(torch.int32)
    xcd = idx // per_xcd
    gid = xcd * (per_xcd // group + 2) + (idx - xcd * per_xcd) // group
    gsum = torch.zeros(int(gid.max()) + 1, dtype=torch.int64, device=panel_ptr.device).index_add_(0, gid, nks)
    top = int(gsum.max())
    key = ((xcd * (top + 1) + (top - gsum[gid])) * (per_xcd + 1) + (idx - xcd * per_xcd) // group) * (group + 1) + (
        idx - xcd * per_xcd) % group
    return torch.argsort(key, stable=True).to(torch.int32)


def build_panel_plan(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, num_cols: int = None,
                     waves: int = DEFAULT_WAVES, row_blocks: int = DEFAULT_ROW_BLOCKS, tau: int = DEFAULT_TAU,
                     min_share: float = 0.0):
    """CSR on the GPU -> ``(resid_indptr, resid_indices, PanelPlan)`` with the HIP builder (panel_plan.hpp; two launches
    around one host sync that sizes the outputs).  Universes above 2^22 columns -- or (column ranges x edges) above 2e10,
    the builder's cost -- get an empty plan (everything stays in the window format).  ``min_share``: when the count phase
    finds fewer than that fraction of the edges in shared columns the builder stops there and returns the empty plan (no
    fill phase, no residual CSR: the format that would lose is never built)."""
    assert indptr.is_cuda and indices.is_cuda and indptr.dtype == torch.int32 and indices.dtype == torch.int32
    assert indptr.is_contiguous() and indices.is_contiguous() and indptr.numel() == num_nodes + 1
    assert waves in (4, 8) and row_blocks in (2, 4) and 1 <= tau <= 65535
    device = indptr.device
    num_cols = num_nodes if num_cols is None else int(num_cols)
    if (num_cols > MAX_PLAN_COLS or num_nodes == 0
            or ((num_cols + (1 << 16) - 1) >> 16) * indices.numel() > MAX_PLAN_EDGE_PASSES):
        return indptr, indices, empty_plan(num_nodes, waves, row_blocks, tau, device, indices.numel())
    stream = torch.cuda.current_stream().cuda_stream
    panel_rows = waves * row_blocks * 16
    num_panels = (num_nodes + panel_rows - 1) // panel_rows
    for attempt in range(2):
        workspace = torch.empty(capi.panel_plan_workspace_bytes(num_nodes, waves, row_blocks), dtype=torch.uint8,
                                device=device)
        panel_ptr = torch.empty(num_panels + 1, dtype=torch.int32, device=device)
        resid_indptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=device)
        status = torch.empty(1, dtype=torch.int32, device=device)
        rc = capi.launch_panel_plan_count(indptr, indices, num_nodes, num_cols, waves, row_blocks, tau, workspace, panel_ptr,
                                          resid_indptr, status, stream)
        capi.check(rc, "voltrix_launch_panel_plan_count")
        total_ksteps, num_resid, bad = torch.cat([panel_ptr[-1:], resid_indptr[-1:], status]).tolist()  # the host sync
        if bad == 0:
            break
        if attempt == 1 or int(indices.min()) < 0 or int(indices.max()) >= num_cols:
            # ids outside the declared universe (rectangular operands): no plan, the window format takes everything --
            # exactly what csr_preprocess does with them (universe-free sort path)
            return indptr, indices, empty_plan(num_nodes, waves, row_blocks, tau, device, indices.numel())
        indptr, indices = canonical_csr(indptr, indices, num_nodes, num_cols)   # unsorted rows / duplicates: once
    if total_ksteps == 0 or indices.numel() - num_resid < min_share * max(1, indices.numel()):
        plan = empty_plan(num_nodes, waves, row_blocks, tau, device, indices.numel())
        plan.num_shared_edges = indices.numel() - num_resid      # what the count phase saw (the plan itself is empty)
        return indptr, indices, plan
    resid_indices = torch.empty(num_resid, dtype=torch.int32, device=device)
    panel_cols = torch.empty(KSTEP * (total_ksteps + 2), dtype=torch.int32, device=device)
    panel_bits = torch.empty((total_ksteps + 1) * waves * 64, dtype=torch.int32, device=device).view(torch.uint32)
    capi.launch_panel_plan_fill(indptr, indices, num_nodes, num_cols, waves, row_blocks, tau, workspace, panel_ptr,
                                resid_indptr, total_ksteps, resid_indices, panel_cols, panel_bits, stream)
    plan = PanelPlan(panel_ptr=panel_ptr, panel_cols=panel_cols, panel_bits=panel_bits,
                     panel_order=longest_first_order(panel_ptr) if total_ksteps > 0 else None,
                     num_nodes=num_nodes, waves=waves, row_blocks=row_blocks, tau=tau, num_ksteps=total_ksteps,
                     num_shared_edges=indices.numel() - num_resid, num_resid_edges=num_resid)
    return resid_indptr, resid_indices, plan


def build_panel_plan_torch(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, num_cols: int = None,
                           waves: int = DEFAULT_WAVES, row_blocks: int = DEFAULT_ROW_BLOCKS, tau: int = DEFAULT_TAU):
    """The same plan from torch tensor ops on any device (sort + run lengths): the first implementation, kept as an
    independent cross-check of the HIP builder that also runs on the CPU (tests/test_hybrid_plan.py).  Not used by the
    operator path."""
    assert indptr.device == indices.device and indptr.dtype == torch.int32 and indices.dtype == torch.int32
    assert waves in (4, 8) and row_blocks in (2, 4) and tau >= 1
    device = indptr.device
    num_cols = num_nodes if num_cols is None else int(num_cols)
    p = waves * row_blocks * 16
    num_panels = (num_nodes + p - 1) // p
    resid_indptr, resid_indices, pc, rp, inv = split_shared_columns(indptr, indices, num_nodes, num_cols, p, tau)

    su_panel = pc // num_cols
    su_col = pc % num_cols
    cnt = torch.bincount(su_panel, minlength=num_panels)
    nks = (cnt + KSTEP - 1) // KSTEP
    panel_ptr = torch.zeros(num_panels + 1, dtype=torch.int64, device=device)
    panel_ptr[1:] = torch.cumsum(nks, 0)
    total_ksteps = int(panel_ptr[-1].item())
    assert total_ksteps < 2 ** 26, "k-step offsets are int32 (x 32 columns)"
    col_start = torch.cumsum(cnt, 0) - cnt                        # first shared key of every panel
    rank = torch.arange(pc.numel(), device=device, dtype=torch.int64) - col_start[su_panel]
    slot = (panel_ptr[su_panel] + rank // KSTEP) * KSTEP + rank % KSTEP   # global (k-step, k) of every shared key

    # unused slots of a panel's last k-step repeat its first shared column (finite data, zero adjacency bits)
    panel_of_kstep = torch.repeat_interleave(torch.arange(num_panels, device=device), nks)
    first_col = torch.zeros(num_panels, dtype=torch.int64, device=device)
    has = cnt > 0
    first_col[has] = su_col[col_start[has]]
    panel_cols = torch.zeros(KSTEP * (total_ksteps + 2), dtype=torch.int64, device=device)
    panel_cols[:KSTEP * total_ksteps] = first_col[panel_of_kstep].repeat_interleave(KSTEP)
    panel_cols[slot] = su_col

    # adjacency bits: word (k-step, wave, lane = 16 g + R), bit 16 (c & 1) + 4 j + (c >> 1), c = column inside the
    # lane's group of 8
    e_slot = slot[inv]
    k = e_slot % KSTEP
    v = rp // (16 * row_blocks)
    j = (rp % (16 * row_blocks)) // 16
    word = (e_slot // KSTEP) * (waves * 64) + v * 64 + (k // 8) * 16 + (rp % 16)
    bits = torch.zeros((total_ksteps + 1) * waves * 64, dtype=torch.int64, device=device)
    bits.index_add_(0, word, torch.ones_like(word) << (16 * (k % 2) + 4 * j + (k % 8) // 2))     # distinct bits: add == or
    panel_bits = ((bits + 2 ** 31) % 2 ** 32 - 2 ** 31).to(torch.int32).view(torch.uint32)   # explicit wrap to 32 bits

    plan = PanelPlan(panel_ptr=panel_ptr.to(torch.int32), panel_cols=panel_cols.to(torch.int32), panel_bits=panel_bits,
                     panel_order=None, num_nodes=num_nodes, waves=waves, row_blocks=row_blocks, tau=tau,
                     num_ksteps=total_ksteps, num_shared_edges=int(rp.numel()),
                     num_resid_edges=int(resid_indices.numel()))
    return resid_indptr, resid_indices, plan


# panel kernel tile per feature width: (fs, depth, ks); waves / row_blocks come from the plan.  Depth 3 at FS = 128 keeps
# the workgroup at 36 KB of LDS and 176 registers, so that it fits on a CU NEXT TO a (128, 3, 4) pair window-kernel
# workgroup (103 KB, 160 registers: 2 x 176 + 160 = 512, tests/test_register_budget.py) -- the two kernels overlap when they
# run on two streams (profiles/HISTORY.md section 3.3).
# ksteps value that selects the software-pipelined k-step loop (one k-step per ring slot; include/voltrix_capi.h
# VOLTRIX_PANEL_KSTEPS_PIPELINED, spmm_panel_kernels.hpp PanelTile<..., PIPE = true>)
KSTEPS_PIPELINED = 17


def default_panel_tile(embedding_dim: int, waves: int, row_blocks: int = DEFAULT_ROW_BLOCKS):
    if embedding_dim <= 32:
        return (32, 6, 2)
    if embedding_dim <= 64:
        return (64, 6, 2)
    if waves == 8:
        # 8 x 2 (256-row panels of panel-dominated graphs, PANEL_DOMINATED_RATIO): 111 registers, two workgroups per CU, the
        # pipelined loop (protein-like 0.860 -> 0.825 ms).  8 x 4 beside the window kernel: the classic loop -- squeezed
        # into the pair's 176 registers the pipelined one gains nothing (profiles/r05/experiment_panel_pipe_176.log)
        return (128, 3, 1) if row_blocks == 4 else (128, 4, KSTEPS_PIPELINED)
    return (128, 4, 1)


# The default shape is 8 waves x 4 row blocks = 512-row panels: a gathered k-step serves 512 rows, and the workgroup fits on a
# CU beside the window kernel's (both kernels are busy for about the same time on the graphs the format was built on).  When the
# panel side is the critical path by this factor -- its k-steps, priced in window-kernel stages (KSTEP_COST_IN_STAGES), against the
# residual's stages (~ residual edges / 32: a residual row gather serves about one edge) -- 8 x 2 = 256-row panels win: half the
# registers, two workgroups per CU whose k-step barriers are independent (the matrix cores of one run under the barrier /
# refill phase of the other), more and shorter panels over the 256 CUs.  Measured (profiles/r05/experiment_panel_shapes*.log,
# F = 128): protein-like (ratio 4.2) 1.361 -> 0.825 ms; reddit-like (0.76) 1.37 -> 1.72-1.77, block model (0.94) 1.01 -> 1.34:
# the balanced graphs keep 8 x 4.
PANEL_DOMINATED_RATIO = 2.0
PANEL_DOMINATED_ROW_BLOCKS = 2


def panel_dominated(plan: "PanelPlan") -> bool:
    """Is the panel kernel of this (8 x 4) plan the step's critical path by PANEL_DOMINATED_RATIO?  Host integers only."""
    resid_stages = max(1.0, plan.num_resid_edges / 32.0)
    return plan.num_ksteps * KSTEP_COST_IN_STAGES >= PANEL_DOMINATED_RATIO * resid_stages


_SIDE_STREAMS = {}


def side_stream(device) -> "torch.cuda.Stream":
    """One extra stream per device for the panel kernel (the window kernel stays on the caller's stream)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def run_two_level(plan: PanelPlan, operand: torch.Tensor, output: torch.Tensor, run_window, out_scale=None,
                  tile=None, concurrent: bool = True) -> None:
    """``output = A_resid @ operand + A_shared @ operand``.  ``run_window(atomic)`` enqueues the window kernel for the
    residual handle on the current stream; with ``atomic`` it adds onto ``output``, otherwise it stores every row.  It
    returns None or a ``PendingCombine`` (jit_kernels/spmm.py).

    The panel kernel runs on a side stream while the window kernel runs on the caller's stream (the first is
    matrix-core bound, the second gather bound: they overlap on the same CUs).  They meet in C through float atomics: C is
    zero-filled, both kernels ADD their part -- two addends per element onto a zero, so the sum does not depend on who comes
    first -- and the partial tiles of cut windows are added by the combine pass after the join (C then holds the panel's part,
    complete).  Stream-ordered, no host sync; capturable (fork / join through events).  ``concurrent=False``: one stream, the
    window kernel stores, the panel kernel adds onto it afterwards (read-add-store) -- for callers that cannot spare a second
    stream; slower, same result up to fp32 summation order."""
    def finish(pending):
        if pending is not None:      # cut windows of a unit-table schedule: sum their partial tiles (fixed order)
            pending.run()

    if plan.num_ksteps == 0:
        finish(run_window(False))
        return
    if not concurrent:
        finish(run_window(False))
        launch_panel(plan, operand, output, accumulate=1, out_scale=out_scale, tile=tile)
        return
    from .utils import timed_launch

    main = torch.cuda.current_stream()
    side = side_stream(operand.device)
    with timed_launch("zero_fill", main):
        output.zero_()
    fork = torch.cuda.Event()
    fork.record(main)
    side.wait_event(fork)                        # operand / out_scale / the zero fill were produced on `main`
    pending_panel = launch_panel(plan, operand, output, accumulate=2, out_scale=out_scale, tile=tile,
                                 stream=side.cuda_stream, defer_combine=True)
    join = torch.cuda.Event()
    join.record(side)
    pending = run_window(True)     # (enqueueing the window kernel first changes nothing: 1.374 vs 1.376 ms, profiles/r04)
    main.wait_event(join)
    finish(pending)                              # adds onto rows that now hold the panel kernel's part, complete
    finish(pending_panel)                        # pieces of cut panels, in slot order (both kernels are done with C)


class PendingPanelCombine:
    """The partial tiles of cut panels, still to be added to C (``launch_panel(..., defer_combine=True)``)."""

    def __init__(self, plan, partials, output, num_feats, accumulate):
        self.plan, self.partials, self.output, self.num_feats, self.accumulate = plan, partials, output, num_feats, accumulate

    def run(self, raw_stream: int = None):
        """On the current stream (bracketed for a KernelTimer), or on the raw HIP stream handle given."""
        from .utils import timed_launch

        current = torch.cuda.current_stream()
        with timed_launch("combine_panel_partials", current) if raw_stream is None else contextlib.nullcontext():
            rc = capi.launch_combine_panel_partials(self.plan.parts, self.partials.data_ptr(), self.output.data_ptr(),
                                                    self.plan.num_nodes, self.num_feats, self.plan.panel_rows,
                                                    self.accumulate, current.cuda_stream if raw_stream is None else raw_stream)
        capi.check(rc, "voltrix_launch_combine_panel_partials")
        self.partials = None


def launch_panel(plan: PanelPlan, feat: torch.Tensor, output: torch.Tensor, accumulate, out_scale=None,
                 tile=None, stream=None, slab_policy: int = None, defer_combine: bool = False):
    """``output (+)= A_shared @ feat`` for fp16 / bfloat16 ``feat`` [*, F] and float32 ``output`` [N, F].
    ``accumulate``: 0 / False store, 1 / True read-add-store, 2 float atomics (include/voltrix_capi.h).
    Plans with a part table (``plan.parts``): cut panels leave partial tiles that a combine pass adds to ``output`` in
    fixed order -- right after the launch on the same stream, or, with ``defer_combine``, by the returned
    ``PendingPanelCombine`` (the atomic join runs it after both kernels are done)."""
    assert feat.is_cuda and feat.is_contiguous() and feat.dtype in (torch.float16, torch.bfloat16)
    assert output.is_cuda and output.is_contiguous() and output.dtype == torch.float32
    f = feat.shape[1]
    assert output.shape == (plan.num_nodes, f)
    tile = tile or default_panel_tile(f, plan.waves, plan.row_blocks)
    assert tile is not None, f"no panel tile for F={f} with {plan.waves} waves"
    stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
    if slab_policy is None:
        from .jit_kernels import spmm as _spmm_wrapper    # the one place the operator's slab policy lives (tests flip it)

        slab_policy = _spmm_wrapper.SLAB_POLICY
    parts = plan.parts
    partials = None
    if parts is not None and parts.num_slots > 0:
        partials = torch.empty(parts.num_slots * plan.panel_rows * f, dtype=torch.float32, device=feat.device)
    rc = capi.launch_spmm_panel(plan, feat.data_ptr(), output.data_ptr(), f, int(accumulate),
                                feat.dtype == torch.bfloat16, tile, out_scale.data_ptr() if out_scale is not None else 0,
                                stream, input_rows=feat.shape[0], slab_policy=slab_policy,
                                partials_ptr=partials.data_ptr() if partials is not None else 0)
    capi.check(rc, "voltrix_launch_spmm_panel")
    if partials is None:
        return None
    pending = PendingPanelCombine(plan, partials, output, f, 1 if int(accumulate) else 0)
    if defer_combine:
        return pending
    pending.run(None if stream == torch.cuda.current_stream().cuda_stream else stream)
    return None


# fused kernel tile per feature width: (fs, depth of the shared panel ring)
def default_fused_tile(embedding_dim: int):
    if embedding_dim <= 32:
        return (32, 4)
    if embedding_dim <= 64:
        return (64, 4)
    return (128, 3)


# sync points per column sweep between the workgroups of an XCD (spmm_fused_kernels.hpp "pace"); 0 = none.  A module
# attribute the operator passes down as an argument (experiments set it; nothing is read from the environment at launch).
FUSED_PACE_BLOCKS = 0


def fused_enabled() -> bool:
    """``VOLTRIX_FUSED=1``: run the two-level product as ONE launch (spmm_fused_kernel: plain stores, no zero fill, no
    atomics, no second stream, one fixed summation order) instead of the panel kernel beside the window kernel with the
    atomic join.  Off by default: measured on the reddit-like graph the one-launch kernel takes 2.0-2.1 ms against 1.35 ms for
    the pair (profiles/r03/experiment_fused_*.log, profiles/HISTORY.md section 3.7) -- it stays as the form for hosts that want a
    single stream-ordered launch and run-to-run identical bits."""
    return os.getenv("VOLTRIX_FUSED", "0") in ("1", "on")


def launch_fused(plan: PanelPlan, fused: FusedRecords, feat: torch.Tensor, output: torch.Tensor, out_scale=None,
                 tile=None, stream=None, pace_blocks: int = None) -> None:
    """``output = (A_shared + A_resid) @ feat`` in ONE launch (spmm_fused_kernels.hpp): every row of ``output`` is written
    once, plain stores, fixed summation order.  fp16 / bfloat16 ``feat`` [*, F], float32 ``output`` [N, F]."""
    assert feat.is_cuda and feat.is_contiguous() and feat.dtype in (torch.float16, torch.bfloat16)
    assert output.is_cuda and output.is_contiguous() and output.dtype == torch.float32
    assert plan.waves == DEFAULT_WAVES and plan.row_blocks == DEFAULT_ROW_BLOCKS, "the fused kernel owns 8 x 4 x 16-row panels"
    f = feat.shape[1]
    assert output.shape == (plan.num_nodes, f)
    tile = tile or default_fused_tile(f)
    stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
    pace_blocks = FUSED_PACE_BLOCKS if pace_blocks is None else pace_blocks
    rc = capi.launch_spmm_fused(plan, fused, feat.data_ptr(), output.data_ptr(), f, feat.dtype == torch.bfloat16, tile,
                                out_scale.data_ptr() if out_scale is not None else 0, stream, pace_blocks)
    capi.check(rc, "voltrix_launch_spmm_fused")


def min_shared_fraction() -> float:
    """Fraction of the edges that must sit in shared columns for ``csr_preprocess`` to attach the two-level side-car
    (``VOLTRIX_HYBRID_MIN_SHARE``).  Default 0.38 in ``auto`` mode, from the measured curve over seven reddit-size graphs
    whose local half of the column mixture goes from 0 to 65 % of the edges (profiles/r04/experiment_share_curve.log; two-level
    time / window-format time at F = 128):

        share   0.293  0.339  0.389  0.437  0.492  0.554  0.662
        ratio   1.076  1.049  0.974  0.877  0.776  0.705  0.577

    -- the forms break even at a share of 0.365 (rounds 2-3: 0.4, "between" the two points measured then, 0.29 and 0.55); 0.2
    when the side-car is forced (``VOLTRIX_HYBRID=1``) or the first call is allowed to time both forms
    (``VOLTRIX_HYBRID=tune``)."""
    default = "0.38" if hybrid_mode() == "auto" else "0.2"
    return float(os.getenv("VOLTRIX_HYBRID_MIN_SHARE", default))


_MODE_OVERRIDE = contextvars.ContextVar("voltrix_hybrid_mode", default=None)


@contextlib.contextmanager
def mode_override(value: str):
    """``with mode_override("0"):`` -- VOLTRIX_HYBRID for the calls of this context (thread / task) only; used by
    library-internal callers (the spectral reorder's products never want a side-car) instead of mutating os.environ."""
    token = _MODE_OVERRIDE.set(value)
    try:
        yield
    finally:
        _MODE_OVERRIDE.reset(token)


def hybrid_mode() -> str:
    """``VOLTRIX_HYBRID``:
    ``auto`` (default)  ``csr_preprocess`` decides ONCE, from the plan builder's count phase: the two-level side-car is built
                        when the graph is big and dense enough for the panel kernel to fill the chip AND at least
                        ``min_shared_fraction()`` of its edges sit in shared columns; ``voltrix.spmm`` then uses it for
                        every 16-bit-operand call.  No timing, no host sync in the operator, same choice (hence the same
                        fp32 summation order, the same bits) in every process and on every rank.
    ``tune``            round-2 behaviour, opt-in: the side-car is built from 20 % shared edges on, and the FIRST
                        ``voltrix.spmm`` per (width, dtype) times both forms (3 runs each, one host sync, not capturable in
                        a HIP graph) and keeps the faster, persisted in ``tuned.json`` under the matrix tag + device.
    ``1``               the side-car whenever enough edges sit in shared columns, used unconditionally;  ``0`` never."""
    v = _MODE_OVERRIDE.get()
    v = os.getenv("VOLTRIX_HYBRID", "auto") if v is None else v
    if v in ("0", "", "off"):
        return "off"
    if v in ("1", "on"):
        return "on"
    return "tune" if v == "tune" else "auto"


# auto mode: below this many edges, or this mean degree, the side-car is not even tried (its build costs a few ms and the
# panel kernel needs columns that several rows of a 512-row panel share)
AUTO_MIN_EDGES = 1 << 22
AUTO_MIN_MEAN_DEGREE = 64
# ... and enough rows for the panel kernel to fill the chip.  Rounds 2-3: one 512-row panel per CU (131 k rows).  With panels in
# pieces (round 4) a plan of ~220 panels launches ~400 workgroups, and the measured break-even moved down
# (profiles/r04/experiment_small_graphs.log, window format -> side-car, ms): 116 k rows 0.810 -> 0.749 (reddit-like), 0.664 ->
# 0.471 (block model) -- whole panels: 1.026 / 0.568 --; 58 k rows 0.411 -> 0.553, 0.301 -> 0.370; 29 k rows 0.228 -> 0.322.
# Interpolated break-even 83-106 k rows.
AUTO_MIN_ROWS = 110_000


def handle_bytes(tensors) -> int:
    """Device bytes of a handle's tensors (reporting: bench.py ``handle_bytes``)."""
    return int(sum(t.numel() * t.element_size() for t in tensors if isinstance(t, torch.Tensor)))


def two_level_bytes(two: "TwoLevelHandle") -> int:
    plan = two.plan
    total = handle_bytes((two.blk_offsets, two.hspa_packed, two.hind, plan.panel_ptr, plan.panel_cols, plan.panel_bits,
                          plan.panel_order))
    if two.fused is not None:
        total += two.fused.nbytes()
    return total
