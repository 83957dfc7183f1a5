"""Stage-list schedules for the C-stationary executor (spmm_list_kernels.hpp in this folder; round-1 experiment, not part of the product since round 4).

A schedule assigns every row window to an accumulator set ``g`` of a wave (``groups`` windows stay resident per wave,
several rounds if there are more windows than ``num_waves * groups``) and lists, per wave, the stages (<= 4 consecutive
TC blocks of one window) in the order they are executed:

``plain``   window by window, blocks front to back -- the order of the one-wave-per-window kernel (used to validate the
            executor: results are bit-identical to ``voltrix_launch_spmm_f16``).
``sweep``   first the blocks whose columns lie within ``near_rows`` of the window's own rows (window by window: row
            neighbours run on the same XCD and share them), then all remaining blocks panel by panel of ``panel_rows``
            rows of B, so that the waves of an XCD sweep B together and every panel is brought into that XCD's L2 about
            once per round instead of once per window.

The schedule depends on ``blk_offsets`` and ``hind`` only (not on B), is built once per handle on the GPU with torch
tensor ops (sort / cumsum / scatter -- plumbing, no per-call cost) and is cached by the caller.  Results do not depend
on timing: the order is data.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

NUM_XCD = 8


@dataclass
class StageList:
    entries: torch.Tensor    # int32 [n, 4]: first block, count | g << 8 | flush << 16 | tail << 17, window, 0
    wave_ptr: torch.Tensor   # int32 [num_waves + 1]
    num_waves: int
    groups: int
    depth: int
    mode: str
    rounds: int
    num_stages: int


def build_stage_list(blk_offsets: torch.Tensor, hspa_packed: torch.Tensor, hind: torch.Tensor, num_nodes: int, *,
                     num_waves: int, groups: int, depth: int, mode: str = "sweep", panel_rows: int = 8192,
                     near_rows: int = 6144, balance: bool = True) -> StageList:
    assert num_waves % NUM_XCD == 0 and num_waves > 0 and 1 <= groups <= 8
    dev = blk_offsets.device
    p1 = blk_offsets.to(torch.int64)
    num_windows = p1.numel() - 1
    nblk = p1[1:] - p1[:-1]
    total = int(p1[-1])
    pad = 2 * depth + 1
    if num_windows == 0 or total == 0:
        wave_ptr = (torch.arange(num_waves + 1, device=dev, dtype=torch.int64) * pad).to(torch.int32)
        entries = torch.zeros((num_waves * pad, 4), dtype=torch.int32, device=dev)
        return StageList(entries, wave_ptr, num_waves, groups, depth, mode, 0, 0)

    blk = torch.arange(total, device=dev, dtype=torch.int64)
    w_of = torch.repeat_interleave(torch.arange(num_windows, device=dev, dtype=torch.int64), nblk)
    first_col = hind.view(-1, 8)[:, 0].to(torch.int64)

    # empty windows own one all-zero TC block (reference quirk): nothing to gather, but their rows must be written
    words = hspa_packed.view(torch.int32).view(-1, 4)
    blk_zero = (words == 0).all(dim=1)
    empty_w = (nblk == 1) & blk_zero[p1[:-1]]

    # ---- segments: maximal runs of consecutive blocks of one window that are scheduled together -------------------
    if mode == "plain":
        seg = torch.zeros(total, dtype=torch.int64, device=dev)
        phase = seg
    elif mode == "sweep":
        center = w_of * 16 + 8
        near = (first_col - center).abs() <= near_rows
        panel = torch.div(first_col, panel_rows, rounding_mode="floor")
        seg = torch.where(near, torch.full_like(panel, -1), panel)
        phase = (~near).to(torch.int64)
    else:
        raise ValueError(mode)
    new_run = torch.ones(total, dtype=torch.bool, device=dev)
    new_run[1:] = (w_of[1:] != w_of[:-1]) | (seg[1:] != seg[:-1])
    run_id = torch.cumsum(new_run.to(torch.int64), 0) - 1
    run_start = blk[new_run][run_id]
    stage_in_run = torch.div(blk - run_start, 4, rounding_mode="floor")
    new_stage = new_run.clone()
    new_stage[1:] |= stage_in_run[1:] != stage_in_run[:-1]
    stage_id = torch.cumsum(new_stage.to(torch.int64), 0) - 1
    num_stages = int(stage_id[-1]) + 1
    s_block0 = blk[new_stage]
    s_count = torch.bincount(stage_id, minlength=num_stages)
    s_w = w_of[new_stage]
    s_seg = seg[new_stage]
    s_phase = phase[new_stage]

    # ---- windows -> (wave, round, accumulator set) ---------------------------------------------------------------------
    # XCD x owns a contiguous window range (wave % 8 == xcd: workgroups are dealt round-robin over the XCDs).  A round is
    # a contiguous chunk of waves_per_xcd * G windows of that range (so the windows that run together are row neighbours:
    # they share the near-diagonal columns).  Inside a round the windows are dealt to the waves in "snake" order of their
    # block count, so every wave carries about the same number of blocks and all waves sweep the panels at the same pace.
    wpx = (num_windows + NUM_XCD - 1) // NUM_XCD
    widx = torch.arange(num_windows, device=dev, dtype=torch.int64)
    xcd = torch.div(widx, wpx, rounding_mode="floor")
    local = widx - xcd * wpx
    waves_per_xcd = num_waves // NUM_XCD
    per_round = waves_per_xcd * groups
    w_round = torch.div(local, per_round, rounding_mode="floor")
    rounds = int(w_round.max()) + 1
    if balance:
        chunk = xcd * rounds + w_round                      # (xcd, round) id, ascending with the window index
        size_key = chunk * (int(nblk.max()) + 1) + (int(nblk.max()) - nblk)   # by chunk, then descending size
        by_size = torch.sort(size_key, stable=True).indices
        chunk_start = torch.zeros(NUM_XCD * rounds + 1, dtype=torch.int64, device=dev)
        chunk_start[1:] = torch.cumsum(torch.bincount(chunk, minlength=NUM_XCD * rounds), 0)
        rank = torch.empty(num_windows, dtype=torch.int64, device=dev)
        rank[by_size] = widx - chunk_start[chunk[by_size]]   # rank of the window inside its chunk, 0 = largest
        lap, pos_in_lap = torch.div(rank, waves_per_xcd, rounding_mode="floor"), rank % waves_per_xcd
        wave_in_xcd = torch.where(lap % 2 == 0, pos_in_lap, waves_per_xcd - 1 - pos_in_lap)   # snake
        w_g = lap
    else:
        slot = torch.div(local - w_round * per_round, groups, rounding_mode="floor")
        wave_in_xcd = slot
        w_g = local % groups
    w_wave = wave_in_xcd * NUM_XCD + xcd

    # ---- order: (wave, round, phase, panel, g) then block order (stable sort keeps ascending blocks) -----------------
    panel_key = torch.where(s_phase == 0, torch.zeros_like(s_seg), s_seg + 1)
    g_major = torch.where(s_phase == 0, w_g[s_w], torch.zeros_like(s_seg))   # near phase: window by window
    key = w_wave[s_w]
    key = key * (rounds + 1) + w_round[s_w]
    key = key * 2 + s_phase
    key = key * 8 + g_major
    key = key * (int(panel_key.max()) + 2) + panel_key
    key = key * 8 + w_g[s_w]
    order = torch.sort(key, stable=True).indices
    s_block0, s_count, s_w = s_block0[order], s_count[order], s_w[order]
    s_wave = w_wave[s_w]

    # flush after the last stage of every window (in execution order)
    pos = torch.arange(num_stages, device=dev, dtype=torch.int64)
    last_pos = torch.zeros(num_windows, dtype=torch.int64, device=dev).scatter_reduce(0, s_w, pos, "amax",
                                                                                       include_self=False)
    flush = (last_pos[s_w] == pos).to(torch.int64)
    count = torch.where(empty_w[s_w], torch.zeros_like(s_count), s_count)
    # tail: the stage contains the window's last TC block (the only one with padded hind slots), is partial, or is empty
    holds_last = (s_block0 + s_count) >= p1[1:][s_w]
    tail = (holds_last | (s_count < 4) | (count == 0)).to(torch.int64)
    info = count | (w_g[s_w] << 8) | (flush << 16) | (tail << 17)

    # ---- per-wave lists with 2*depth+1 padding entries -----------------------------------------------------------------
    per_wave = torch.bincount(s_wave, minlength=num_waves)
    wave_ptr = torch.zeros(num_waves + 1, dtype=torch.int64, device=dev)
    wave_ptr[1:] = torch.cumsum(per_wave + pad, 0)
    first_of_wave = torch.zeros(num_waves + 1, dtype=torch.int64, device=dev)
    first_of_wave[1:] = torch.cumsum(per_wave, 0)
    dst = wave_ptr[s_wave] + (pos - first_of_wave[s_wave])
    entries = torch.zeros((int(wave_ptr[-1]), 4), dtype=torch.int32, device=dev)
    entries[dst, 0] = s_block0.to(torch.int32)
    entries[dst, 1] = info.to(torch.int32)
    entries[dst, 2] = s_w.to(torch.int32)
    # padding: count 0, block = the wave's last real block (a valid block of a valid window), or block 0
    last_block = torch.zeros(num_waves, dtype=torch.int64, device=dev)
    has = per_wave > 0
    last_idx = (first_of_wave[1:] - 1).clamp(min=0)
    last_block[has] = s_block0[last_idx[has]]
    pad_dst = (wave_ptr[:-1] + per_wave)[:, None] + torch.arange(pad, device=dev, dtype=torch.int64)[None, :]
    entries[pad_dst.reshape(-1), 0] = last_block[:, None].expand(num_waves, pad).reshape(-1).to(torch.int32)
    entries[pad_dst.reshape(-1), 1] = 1 << 17   # padding: count 0, tail path (rows = a valid row, nothing multiplied)
    return StageList(entries.contiguous(), wave_ptr.to(torch.int32), num_waves, groups, depth, mode, rounds, num_stages)



