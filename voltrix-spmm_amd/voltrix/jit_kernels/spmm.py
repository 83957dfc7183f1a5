"""``spmm_kernel``: the tiled SpMM accumulate (reference voltrix/jit_kernels/spmm.py:39-94).

The tuning space is the gfx950 tile template ``SpmmTile<FS, DEPTH, WAVES, EB>`` (traits.hpp) instead of the
reference's ``model`` 0/1/2; the tuner key adds the feature width, operand dtype and device to the matrix tag
(the reference keys on the tag alone, so the model picked at the first F is reused for every F -- SURVEY.md
section 8a quirk 9).
"""
import contextlib
import contextvars
import os
import warnings

import torch

from ..jit.compiler import hash_to_hex
from ..project import TUNE_SPACE_FLAG
from .tuner import jit_tuner

# Per-context override of VOLTRIX_TUNE_SPACE (library-internal callers such as the spectral reorder run their products with
# the default tiles whatever the process environment says -- without touching os.environ, which every other thread reads)
_TUNE_SPACE_OVERRIDE = contextvars.ContextVar("voltrix_tune_space", default=None)


def tune_space_mode() -> str:
    return _TUNE_SPACE_OVERRIDE.get() or os.getenv(TUNE_SPACE_FLAG, "default")


@contextlib.contextmanager
def tune_space(mode: str):
    """``with tune_space("none"):`` -- VOLTRIX_TUNE_SPACE for the calls of this context (thread / task) only."""
    token = _TUNE_SPACE_OVERRIDE.set(mode)
    try:
        yield
    finally:
        _TUNE_SPACE_OVERRIDE.reset(token)


includes = ('"voltrix/spmm_kernels.hpp"', '"voltrix/spmm_stream_kernels.hpp"')
template = """
if ({SCHED} == 6) {
  __return_code = voltrix::launch_spmm_stream<voltrix::SpmmTile<{FS}, {DEPTH}, {WAVES}, {EB}, {BF16} != 0, {WEIGHTED} != 0>>(
      hspa_packed, hind, num_nodes, embedding_dim, input, output, stream, s_units, s_runs, s_run_ptr, s_max_runs, partials_s,
      out_scale, 0, 0, input_rows, slab_policy);
  if (__return_code == 0 && combine_now != 0)
    __return_code = voltrix::combine_partials(cuts_s, num_cuts_s, partials_s, output, num_nodes, embedding_dim, 0, stream, nullptr);
  return;
}
__return_code = voltrix::launch_spmm_tc16<voltrix::SpmmTile<{FS}, {DEPTH}, {WAVES}, {EB}, {BF16} != 0, {WEIGHTED} != 0>>(
    blk_offsets, hspa_packed, hind,
    num_nodes, embedding_dim, input, output, stream,
    ({SCHED} == 0 || {SCHED} >= 4) ? nullptr : ({SCHED} == 1 ? win_order_a : ({SCHED} == 2 ? win_order_b : win_order_c)),
    out_scale, atomic_out,
    {SCHED} == 4 ? units : ({SCHED} == 5 ? units_p : nullptr), {SCHED} == 5 ? unit_ptr_p : unit_ptr,
    {SCHED} == 5 ? max_units_per_xcd_p : max_units_per_xcd, {SCHED} == 5 ? partials_p : partials,
    has_row_map != 0 ? row_map : nullptr, {WEIGHTED} != 0 ? (const void*)values : nullptr, {SCHED} == 5 ? 2 : 1,
    0, 0, input_rows, slab_policy);
if (__return_code == 0 && {SCHED} == 4 && combine_now != 0)
  __return_code = voltrix::combine_partials(cuts, num_cuts, partials, output, num_nodes, embedding_dim, atomic_out, stream,
                                            has_row_map != 0 ? row_map : nullptr);
if (__return_code == 0 && {SCHED} == 5 && combine_now != 0)
  __return_code = voltrix::combine_partials(cuts_p, num_cuts_p, partials_p, output, num_nodes, embedding_dim, atomic_out,
                                            stream, has_row_map != 0 ? row_map : nullptr);
"""

# windows per length-sorted chunk of the "balance" schedule (spmm_kernels.hpp::launch_window_order) for SCHED 1/2/3.
# Small chunks keep row neighbours together (banded graphs, wide features), wide chunks equalise more (uniform columns).
# Measured optimum on MI355X: reddit-like F=128 -> 512, F=512 -> 128, uniform columns -> 2048 (profiles/HISTORY.md section 5).
ORDER_CHUNKS = {1: 128, 2: 512, 3: 2048}
# SCHED 4: unit table (voltrix/schedule.py::unit_table) -- windows longer than 1.5 x the median cut into interleaved units,
# units listed longest first per XCD range; the partial tiles of cut windows are summed in unit order by
# combine_partials.  Measured on the reddit-like graph: window format 2.19 -> 1.94 ms, two-level residual 1.41 -> 1.04 ms.
SCHED_UNITS = 4
# SCHED 5: the same with TWO units per wave (spmm_tc16_pair_kernel: their stages alternate through one ring into two
# accumulator sets), units cut at 1.25 x the median.  Twice the rows sweep their sorted columns in step per CU at the same LDS
# and bytes in flight: reddit-like two-level residual TCC hits 49 -> 58 %, 1.03 -> 0.92 ms alone, the pair with the panel kernel
# 1.365 -> 1.293 ms (profiles/r02/experiment_pair_units.log).  16-bit binary operand, four-wave tiles; with several column
# slabs the launch is slab-major (slabs of 128 bytes and more) and a pair never straddles two slabs.
SCHED_PAIRS = 5
# SCHED 6: the window format as a STREAM of stages (spmm_stream_kernels.hpp, round 5): a wave walks a run of consecutive
# windows through one ring that never drains at a window boundary, stores every finished window from the loop with 16-byte
# stores, and needs neither a prologue per window nor a 4-byte-per-lane epilogue -- the kernel for the short windows of the
# reference's low-degree evaluation graphs.  16-bit binary operand, plain stores to C (no row map, no atomics).
SCHED_STREAM = 6
STREAM_MAX_BLOCKS_PER_WINDOW = 48   # 12 stages per window on average (the evaluation set's low-degree graphs: 3-30 TC blocks)
PAIR_UNIT_FACTOR = 1.25   # x the median window length (measured: 1.0 .. 1.5 within 1 %, profiles/r02/experiment_pair_units.log)

# How an operand wider than the tile's slab is launched (spmm_kernels.hpp::slab_launch_group): -1 = the library's rule (one
# launch per 256-byte group of column slabs when such a group of B fits the Infinity Cache), 0 = always one grid, 1 = always
# the launches.  A module attribute the operator passes down as an ARGUMENT of every launch (tests flip it to compare the two
# forms bit for bit); nothing on the launch path reads the environment.
SLAB_POLICY = -1


def slab_launches(embedding_dim: int, fs: int, elem_bytes: int, rows: int, policy: int = None) -> int:
    """Kernel launches a call of the window (or panel) kernel makes for an ``embedding_dim``-wide operand of ``rows`` rows:
    the rule of ``spmm_kernels.hpp::slab_launch_group`` restated for reports (bench.py ``config.tile.launches_per_step``,
    harness/pmc_summarize.py)."""
    policy = SLAB_POLICY if policy is None else policy
    slabs = -(-embedding_dim // fs)
    slab_bytes = fs * elem_bytes
    if slab_bytes < 128 or policy == 0 or (policy < 0 and rows * 256 > (256 << 20)):
        return 1
    group = 1 if slab_bytes >= 256 else 256 // slab_bytes
    return -(-slabs // group) if slabs > group else 1


def feature_hash(feature: torch.Tensor) -> str:
    """Tag of the sparse matrix the handle belongs to (reference spmm.py:17-36): the caller-set
    ``hspa_packed.hash_tag`` string, else the buffer address (with the reference's warning)."""
    if hasattr(feature, "hash_tag") and isinstance(feature.hash_tag, str):
        return hash_to_hex(feature.hash_tag)
    warnings.warn(
        "The feature tensor(i.e. `hspa_packed`)'s hash_tag attr is not set. "
        "Voltrix will use the memory address as the key value for profiling, "
        "which may lead to performance degradation of different cases."
    )
    return hash_to_hex(str(feature.data_ptr()))


def _lds_bytes(fs, depth, waves, eb, weighted=False):
    return waves * (depth * 32 * fs * eb + (2 * depth + 1) * (1280 if weighted else 256))


# LDS a window-kernel workgroup may take when a panel-kernel workgroup (two-level format, 44 KB at FS = 128 / DEPTH 3)
# has to fit on the same CU beside it
TWO_LEVEL_LDS_BUDGET = 160 * 1024 - 47 * 1024   # panel workgroup: 24 KiB ring + 20 KiB metadata + slack


def tile_space(embedding_dim: int, elem_bytes: int, bf16: bool = False, max_lds: int = None, weighted: bool = False,
               stream_ok: bool = True, shallow_ok: bool = True):
    """Points of the tile space worth trying for this feature width (``bf16``: the 2-byte operand is bfloat16;
    ``max_lds``: keep only tiles whose workgroup fits that many bytes of LDS; ``weighted``: the A operand is a value
    plane, 1 KiB more per metadata slot)."""
    points = tuple(dict(point, BF16=int(bf16), WEIGHTED=int(weighted)) for point in _tile_space(embedding_dim, elem_bytes))
    if not shallow_ok and tune_space_mode() == "default":
        # two-slot rings of the window kernel (round 6) are candidates for handles of SHORT windows only: on long windows the sweep's
        # sample ranks them first and the full-size step loses (fresh sweeps with them everywhere: protein-like 0.84 -> 1.11 ms,
        # products-like F = 128 3.3 -> 4.1 ms, profiles/r06/experiment_depth2_default_space.log)
        points = tuple(p for p in points if p["DEPTH"] != 2 or p["SCHED"] == SCHED_STREAM)
    if weighted or max_lds is not None or not stream_ok:   # the stream kernel: binary operand, plain stores, alone on the CU
        points = tuple(p for p in points if p["SCHED"] != SCHED_STREAM)
        if not points:   # VOLTRIX_TUNE_SPACE=stream on a launch the stream kernel does not serve: the default tile
            with tune_space("none"):
                points = tuple(dict(point, BF16=int(bf16), WEIGHTED=int(weighted)) for point in _tile_space(embedding_dim, elem_bytes))
    if weighted:
        assert elem_bytes == 2
        points = tuple(p for p in points if p["SCHED"] != SCHED_PAIRS)   # paired units: binary operand only
        points = tuple(p for p in points if _lds_bytes(p["FS"], p["DEPTH"], p["WAVES"], 2, True) <= 160 * 1024
                       and (2 + 32 * p["FS"] * 2 // 1024) * (p["DEPTH"] - 1) <= 63)
    if (max_lds is not None and not weighted and tune_space_mode() == "none"
            and len(points) == 1
            and points[0]["EB"] == 2 and (points[0]["FS"] >= 64 or embedding_dim <= points[0]["FS"])):
        # the single untuned point beside a panel workgroup: two units per wave (measured: 1.365 -> 1.293 ms for the pair)
        points = (dict(points[0], SCHED=SCHED_PAIRS),)
    if max_lds is not None:
        # beside a panel workgroup: the panel tile's slab width only (the two kernels then walk the column slabs in step;
        # the tuner times this kernel ALONE, where a half-width slab-major tile can look as good -- reddit-like F=128:
        # FS=64 picked once, 1.61 ms for the pair against 1.38 ms with FS=128, profiles/r02/bench_fs64_beside_panel.json)
        fs_top = max(p["FS"] for p in points)
        fit = tuple(p for p in points if _lds_bytes(p["FS"], p["DEPTH"], p["WAVES"], p["EB"], weighted) <= max_lds
                    and p["WAVES"] >= 4 and p["FS"] == fs_top)
        points = fit or points
    return points


def _tile_space(embedding_dim: int, elem_bytes: int):
    mode = tune_space_mode()
    fs4 = 32 if embedding_dim <= 32 else 64     # fp32 rows: slabs of 128 or 256 bytes
    fs_max = 128
    fs_fit = 32 if embedding_dim <= 32 else (64 if embedding_dim <= 64 else fs_max)
    if mode == "none":  # the ahead-of-time library's default tile (csrc/capi_common.hpp::default_tile) + unit table
        fs = 32 if embedding_dim <= 32 else (64 if embedding_dim <= 64 else 128)
        if elem_bytes == 4:
            return ({"FS": min(fs, 64), "DEPTH": 3, "WAVES": 1, "EB": 4, "SCHED": 2},)
        return ({"FS": fs, "DEPTH": 4 if fs == 32 else 3, "WAVES": 4, "EB": 2, "SCHED": SCHED_UNITS},)
    if mode == "stream":   # the stream kernel's default point alone (tests, experiments)
        if elem_bytes == 4:
            return ({"FS": fs4, "DEPTH": 3, "WAVES": 1, "EB": 4, "SCHED": SCHED_STREAM},)
        return ({"FS": fs_fit, "DEPTH": 3 if fs_fit >= 128 else 4, "WAVES": 1, "EB": 2, "SCHED": SCHED_STREAM},)
    if mode == "full":
        fs_list = sorted({fs_fit, max(32, fs_fit // 2), min(256, fs_fit * 2) if embedding_dim > 128 else fs_fit})
        depths, waves = (2, 3, 4), (1, 2, 4)
    else:
        fs_list = sorted({fs_fit, max(32, fs_fit // 2)})
        # round 6: ring depth 2 joins the default space.  A FULL sweep on the low-degree stand-ins found half-width, two-slot
        # window tiles 7-17 % ahead of what the default space could offer (profiles/r06/experiment_sweep_compare.log: amazon0505-like
        # 0.160 -> 0.132 ms, amazon0601-like 0.133 -> 0.114, DD-like 0.080 -> 0.069, ppi-like 0.027 -> 0.023, all FS 64 / DEPTH 2):
        # with one or two stages per window a third ring slot only costs LDS, i.e. waves per CU.  The staged sweep starts every
        # shape at its shallowest ring, so this adds no first-stage candidate (6 + <= 4 + 2 = 12 timed at most).  Short-window
        # handles only (tile_space: shallow_ok).
        depths, waves = (2, 3, 4), (1, 4)
    space = []
    for fs in fs_list:
        for d in depths:
            for w in waves:
                ndma = 32 * fs * elem_bytes // 1024
                if _lds_bytes(fs, d, w, elem_bytes) <= 160 * 1024 and (1 + ndma) * (d - 1) <= 63:
                    # natural window order / balance schedule per chunk size / unit table (16-bit operands)
                    scheds = (0,) + tuple(ORDER_CHUNKS) + ((SCHED_UNITS,) if elem_bytes == 2 else ())
                    if elem_bytes == 2 and w == 4 and (fs >= 64 or embedding_dim <= fs):
                        scheds += (SCHED_PAIRS,)   # two units per wave (several slabs: slab-major order, i.e. slabs >= 128 bytes)
                    for sched in scheds:
                        space.append({"FS": fs, "DEPTH": d, "WAVES": w, "EB": elem_bytes, "SCHED": sched})
    # stream points: the full-width slab, one or two waves per workgroup (LDS decides the waves per CU); fp32 rows: exact products
    fs_s = fs_fit if elem_bytes == 2 else fs4
    for w in (1, 2):
        for d in (2, 3, 4) if elem_bytes == 2 else (2, 3):
            if (1 + 32 * fs_s * elem_bytes // 1024) * (d - 1) + d * (fs_s // 16) <= 63:   # loads + stores behind a wait
                space.append({"FS": fs_s, "DEPTH": d, "WAVES": w, "EB": elem_bytes, "SCHED": SCHED_STREAM})
    return tuple(space)


# ---- the bounded sweep (round 4) ----------------------------------------------------------------------
SAMPLE_MIN_WINDOWS = 1 << 14     # below this many windows the sweep times the whole handle
SAMPLE_CHUNKS = 2                # contiguous window ranges of the sample (at 1/4 and 3/4 of the handle)
SAMPLE_FRACTION = 16             # 1 / this of the windows in all


SAMPLE_MIN_STAGES = 1 << 19      # a sample launch must fill the chip for a while: at least this many stages (4 TC blocks each)


def sample_ranges(num_windows: int, total_blocks: int = None):
    """Window ranges ``[(w0, w1)]`` the sweep times its candidates on: the whole handle when it is small, else
    ``SAMPLE_CHUNKS`` contiguous ranges centred at (2 k + 1) / (2 SAMPLE_CHUNKS) of the windows, 1 / ``SAMPLE_FRACTION`` of
    them in all (few, long launches: a launch's drained tail is what distorts a sample -- four ranges of 1/64 ranked the
    power-law graph's tiles 4 % off, profiles/r04/experiment_tuner_sample_powerlaw_v1.log).  A contiguous range of a
    block-format handle is itself a handle (``blk_offsets[w0 : w1 + 1]`` holds absolute TC-block offsets into the same
    ``hspa_packed`` / ``hind``), so a sample launch is the SAME kernel with a shifted pointer, fewer rows and its own schedule
    arrays: nothing is copied.  Round 5: the sample is sized by WORK, not by windows -- on the low-degree graphs of the
    reference's evaluation set a window is one or two stages, 1/16 of the windows is a launch of a few microseconds, and such a
    sample ranked the kernels by their launch latency (the stream kernel, 1.7 x faster on the whole graph, lost it): with
    ``total_blocks`` given, the ranges hold at least ``SAMPLE_MIN_STAGES`` stages (the whole handle when it has fewer)."""
    if num_windows <= SAMPLE_MIN_WINDOWS:
        return [(0, num_windows)]
    size = max(SAMPLE_MIN_WINDOWS // (2 * SAMPLE_CHUNKS), num_windows // (SAMPLE_FRACTION * SAMPLE_CHUNKS))
    if total_blocks is not None:
        stages = max(1, total_blocks // 4)
        if stages <= SAMPLE_MIN_STAGES:
            return [(0, num_windows)]
        size = max(size, -(-num_windows * SAMPLE_MIN_STAGES // (stages * SAMPLE_CHUNKS)))
        if size * SAMPLE_CHUNKS * 2 > num_windows:
            return [(0, num_windows)]
    out = []
    for k in range(SAMPLE_CHUNKS):
        centre = (2 * k + 1) * num_windows // (2 * SAMPLE_CHUNKS)
        w0 = max(0, min(num_windows - size, centre - size // 2))
        out.append((w0, w0 + size))
    return out


def sweep_stages(space, best=None, stage_no=0):
    """The stages of the bounded sweep (<= 12 candidates of the default space's 44).
    Stage 0: one candidate per (FS, WAVES) at the SHALLOWEST ring, each with the most robust schedule its operand type has --
    the unit table for 16-bit operands (balanced on every graph measured: tails 0.7-1.7 %), the chunk-512 balance schedule
    for fp32 ones -- so that shapes are compared on equal terms.  Stage 1: the other schedules of the winning shape (natural
    order, balance chunks 512 / 2048, two units per wave; chunk 128 never won a sweep and is left to
    ``VOLTRIX_TUNE_SPACE=full``).  Stage 2: the winner at the other ring depths."""
    def shape(p):
        return (p["FS"], p["WAVES"], p["EB"], p["SCHED"] == SCHED_STREAM)

    # short-window handles (their space carries two-slot window tiles: tile_space(shallow_ok)): the first-stage schedule of a window
    # shape is the NATURAL order -- nothing is cut there, and the unit table's longest-first order gives up what consecutive windows
    # share in L2 (held-out co-purchase graph: the staged sweep kept the stream kernel at 0.0856 ms, a full sweep found (64, 2, 1,
    # natural) at 0.0706; profiles/r06/experiment_sweep_compare.log)
    short = any(p["DEPTH"] == 2 and p["SCHED"] != SCHED_STREAM for p in space)

    def robust(points):
        for pref in ((SCHED_STREAM, 0, SCHED_UNITS, 2) if short else (SCHED_STREAM, SCHED_UNITS, 2, 0)):
            for p in points:
                if p["SCHED"] == pref:
                    return p
        return points[0]

    if best is None:
        shapes = {}
        for p in space:
            shapes.setdefault(shape(p), []).append(p)
        out = []
        for points in shapes.values():
            depth = min(p["DEPTH"] for p in points)
            out.append(robust([p for p in points if p["DEPTH"] == depth]))
        return out
    if stage_no == 1:
        if short:
            # short windows: the chunk-128 balance schedule is a candidate (it won the full sweep on amazon0505-like: 0.132 ms against
            # 0.160 for the shipped stream tile); the plain unit table gives its place to it where the paired one is in the space
            # (both cut the few long windows there are) -- still at most 12 timed candidates
            has_pairs = any(p["SCHED"] == SCHED_PAIRS for p in space if shape(p) == shape(best))
            skip = (best["SCHED"],) + ((SCHED_UNITS,) if has_pairs else ())
            return [p for p in space if shape(p) == shape(best) and p["DEPTH"] == best["DEPTH"] and p["SCHED"] not in skip]
        return [p for p in space if shape(p) == shape(best) and p["DEPTH"] == best["DEPTH"] and p["SCHED"] not in (best["SCHED"], 1)]
    if stage_no == 2:
        return [p for p in space if shape(p) == shape(best) and p["SCHED"] == best["SCHED"] and p["DEPTH"] != best["DEPTH"]]
    return []


def sweep_budget_s(step_s: float) -> float:
    """Wall-clock cap of one sweep: max(2 s, 20 x the full-size step)."""
    return max(2.0, 20.0 * step_s)


def sweep_bench(fn) -> float:
    """Milliseconds per run of ``fn`` for the sweep: median of 3 batches of 4 back-to-back runs, one event pair per batch, after
    one warm-up (a host sync after every single launch lets the clocks drop between launches and reads 10-15 % high on
    MI355X; the reference's per-run cache flush, utils.py:277-281, prices a cold cache no steady-state caller sees)."""
    fn()
    times = []
    for _ in range(3):
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(4):
            fn()
        end.record()
        end.synchronize()
        times.append(start.elapsed_time(end) / 4)
    return sorted(times)[1]


def window_order(blk_offsets: torch.Tensor, hspa_packed: torch.Tensor, num_nodes: int, sched: int = 1) -> torch.Tensor:
    """The handle's "balance" schedule ``sched`` (key of ORDER_CHUNKS), computed once on the GPU and cached on the
    ``hspa_packed`` tensor object."""
    cache = getattr(hspa_packed, "_voltrix_window_order", None)
    key = (blk_offsets.data_ptr(), num_nodes, sched)
    if isinstance(cache, dict) and key in cache:
        return cache[key]
    from .. import capi

    order = torch.empty((num_nodes + 15) // 16, dtype=torch.int32, device=blk_offsets.device)
    capi.launch_window_order(blk_offsets, num_nodes, order, torch.cuda.current_stream().cuda_stream, ORDER_CHUNKS[sched])
    try:
        if not isinstance(cache, dict):
            cache = {}
            hspa_packed._voltrix_window_order = cache
        cache[key] = order
    except AttributeError:
        pass
    return order


def arg_defs_for(dtype):
    """Argument list of the generated ``launch`` for a dense operand of ``dtype`` (the reference's ten arguments,
    jit_kernels/spmm.py:78-88, then the schedule / output extensions)."""
    return (
        ("blk_offsets", torch.int32),
        ("hspa_packed", torch.uint32),
        ("hind", torch.int32),
        ("num_nodes", int),
        ("num_edges", int),
        ("embedding_dim", int),
        ("input", dtype),
        ("output", torch.float32),
        ("win_order_a", torch.int32),
        ("win_order_b", torch.int32),
        ("win_order_c", torch.int32),
        ("out_scale", torch.float32),
        ("atomic_out", int),
        ("units", torch.int32),
        ("unit_ptr", torch.int32),
        ("max_units_per_xcd", int),
        ("cuts", torch.int32),
        ("num_cuts", int),
        ("partials", torch.float32),
        ("units_p", torch.int32),
        ("unit_ptr_p", torch.int32),
        ("max_units_per_xcd_p", int),
        ("cuts_p", torch.int32),
        ("num_cuts_p", int),
        ("partials_p", torch.float32),
        ("combine_now", int),
        ("row_map", torch.int32),
        ("has_row_map", int),
        ("values", dtype),
        ("input_rows", int),
        ("slab_policy", int),
        ("s_units", torch.int32),
        ("s_runs", torch.int32),
        ("s_run_ptr", torch.int32),
        ("s_max_runs", int),
        ("cuts_s", torch.int32),
        ("num_cuts_s", int),
        ("partials_s", torch.float32),
        ("stream", torch.cuda.Stream),
    )


def handle_unit_table(blk_offsets: torch.Tensor, hspa_packed: torch.Tensor, num_nodes: int, pairs: bool = False,
                      xcd_ptr: torch.Tensor = None):
    """The handle's unit table (voltrix.schedule.unit_table; default length bound, or 1.25 x the median for the paired
    launch), built once on the GPU and cached on the ``hspa_packed`` tensor object.  XCD ranges: the caller's ``xcd_ptr``
    (the two-level step: the panel kernel's ranges) or, round 4, ranges of equal STAGES instead of equal window counts
    (``schedule.balanced_xcd_windows``: graphs whose rows are not statistically alike)."""
    attr = "_voltrix_unit_table_pairs" if pairs else "_voltrix_unit_table"
    cache = getattr(hspa_packed, attr, None)
    key = (blk_offsets.data_ptr(), num_nodes, xcd_ptr.data_ptr() if xcd_ptr is not None else 0)
    if isinstance(cache, tuple) and cache[0] == key:
        return cache[1]
    from ..schedule import balanced_xcd_windows, default_max_stages, unit_table

    ranges = xcd_ptr if xcd_ptr is not None else balanced_xcd_windows(blk_offsets, num_nodes)
    if pairs:
        median_x_1_5 = default_max_stages(blk_offsets, num_nodes)
        table = unit_table(blk_offsets, num_nodes, max(8, int(PAIR_UNIT_FACTOR * median_x_1_5 / 1.5)), xcd_ptr=ranges)
    else:
        table = unit_table(blk_offsets, num_nodes, xcd_ptr=ranges)
    try:
        setattr(hspa_packed, attr, (key, table))
    except AttributeError:
        pass
    return table


def handle_stream_table(blk_offsets: torch.Tensor, hspa_packed: torch.Tensor, hind: torch.Tensor, num_nodes: int):
    """The handle's stream table (voltrix.schedule.stream_tables), built once and cached on the ``hspa_packed`` tensor object."""
    cache = getattr(hspa_packed, "_voltrix_stream_table", None)
    key = (blk_offsets.data_ptr(), num_nodes)
    if isinstance(cache, tuple) and cache[0] == key:
        return cache[1]
    from ..schedule import stream_tables

    table = stream_tables(blk_offsets, hspa_packed, hind, num_nodes)
    try:
        hspa_packed._voltrix_stream_table = (key, table)
    except AttributeError:
        pass
    return table


def graph_bucket_keys(blk_offsets: torch.Tensor, num_nodes: int, keys: dict):
    """The tuner's coarse key (the reference memoises per process only, jit_kernels/tuner.py:44):
    ``keys`` with the matrix tag replaced by a bucket of statistics of the handle -- log2 of the row count, log2 of the mean
    TC blocks per window (what the mean degree becomes in the block format), the quartiles of the TC blocks per window in
    half-octaves, and the coefficient of variation of the window lengths in steps of 0.25 (band graphs, uniform graphs and
    power-law graphs of one size land in different buckets).  Graphs of one bucket keep the tile and schedule tuned on the
    first of them.  One small device reduction + one host sync, only when a sweep would otherwise run; cached on the tensor."""
    import math

    cached = getattr(blk_offsets, "_voltrix_bucket", None)
    if cached is None or cached[0] != (blk_offsets.data_ptr(), num_nodes):
        num_windows = (num_nodes + 15) // 16
        if num_windows == 0:
            return None
        nblk = (blk_offsets[1:num_windows + 1] - blk_offsets[:num_windows]).float()
        sample = nblk[:: max(1, num_windows // (1 << 20))]
        q = torch.quantile(sample, torch.tensor([0.25, 0.5, 0.75], device=nblk.device))
        stats = torch.cat([q, nblk.mean()[None], nblk.std(unbiased=False)[None]]).tolist()

        def half_octave(v):
            return int(round(2 * math.log2(max(v, 1.0))))

        bucket = {"log2_rows": int(round(math.log2(max(num_nodes, 1)))),
                  "log2_mean_blocks_x2": half_octave(stats[3]),
                  "quartiles_x2": [half_octave(v) for v in stats[:3]],
                  "cv_x4": int(round(4 * stats[4] / max(stats[3], 1e-9)))}
        cached = ((blk_offsets.data_ptr(), num_nodes), bucket)
        try:
            blk_offsets._voltrix_bucket = cached
        except AttributeError:
            pass
    # the ARCHITECTURE, not the marketing name, keys a bucket: the same gfx950 part reports "AMD Instinct MI355X" or "AMD Radeon
    # Graphics" depending on the driver stack, and the shipped defaults must hit on both
    out = {k: v for k, v in keys.items() if k not in ("feature_hash", "device")}
    try:
        out["arch"] = torch.cuda.get_device_properties(blk_offsets.device).gcnArchName.split(":")[0]
    except (AttributeError, RuntimeError):
        out["arch"] = keys.get("device", "unknown")
    out["graph_bucket"] = str(cached[1])
    if isinstance(out.get("embedding_dim"), int) and out["embedding_dim"] > 128:
        # wide operands run as 128-column slabs, one launch per slab (spmm_kernels.hpp::slab_launch_group): a width the
        # store has not seen takes the choice of any other wide operand of the bucket (the exact width is asked first)
        return [out, dict(out, embedding_dim="wide")]
    return out


_UNIT_SCALE = {}


def unit_scale(device) -> torch.Tensor:
    """float32[2] = {1, 0} on ``device``: the neutral ``out_scale`` (operands that were not rescaled)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _UNIT_SCALE:
        _UNIT_SCALE[key] = torch.tensor([1.0, 0.0], dtype=torch.float32, device=device)
    return _UNIT_SCALE[key]


class PendingCombine:
    """Partial tiles of cut windows that still have to be summed into the output (``spmm_kernel(..., defer_combine=True)``:
    the two-level step runs the pass after the join with the panel kernel, when the output holds that kernel's part)."""

    def __init__(self, table, partials, output, num_nodes, embedding_dim, accumulate, row_map=None):
        self.table, self.partials, self.output, self.row_map = table, partials, output, row_map
        self.num_nodes, self.embedding_dim, self.accumulate = num_nodes, embedding_dim, accumulate

    def run(self, stream=None):
        from .. import capi

        stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
        rc = capi.launch_combine_partials(self.table, self.partials.data_ptr(), self.output.data_ptr(), self.num_nodes,
                                          self.embedding_dim, self.accumulate, stream,
                                          self.row_map.data_ptr() if self.row_map is not None else 0)
        capi.check(rc, "voltrix_launch_combine_partials")


def _raw_stream(device) -> int:
    """The current stream's handle without building a ``torch.cuda.Stream`` object (9 us of the 57 us a call used to cost)."""
    try:
        return torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())
    except AttributeError:
        return torch.cuda.current_stream(device).cuda_stream


class _LaunchPlan:
    """Everything of a ``spmm_kernel`` call that does not change between calls on one handle -- the chosen kernel's entry point,
    its argument list already marshalled to ctypes, which partial-tile buffer the schedule needs, what ``defer_combine`` returns --
    so that a repeated call patches five pointers and launches (round 5: the wrapper cost 57 us of host time per call, more than
    the kernel on the small graphs of the reference's evaluation set: ppi, ddi, FraudYelp).  Kept on the ``hspa_packed`` tensor
    OBJECT (it dies with it; a copy of the tensor starts without plans)."""
    __slots__ = ("fn", "cargs", "generation", "partials_index", "partials_floats", "combine_table", "combine_args", "device",
                 "keepalive")

    def launch(self, input, output, out_scale, values, defer_combine):
        import ctypes

        cargs = list(self.cargs)
        cargs[6] = ctypes.c_void_p(input.data_ptr())
        cargs[7] = ctypes.c_void_p(output.data_ptr())
        cargs[11] = ctypes.c_void_p(out_scale.data_ptr())
        cargs[28] = ctypes.c_void_p((values if values is not None else input).data_ptr())
        partials = None
        if self.partials_index is not None:
            partials = torch.empty(self.partials_floats, dtype=torch.float32, device=input.device)
            cargs[self.partials_index] = ctypes.c_void_p(partials.data_ptr())
        cargs[38] = ctypes.c_void_p(_raw_stream(self.device))
        rc = ctypes.c_int(-1)
        self.fn(*cargs, ctypes.byref(rc))
        assert rc.value == 0, f"spmm_kernel failed with return code {rc.value}"
        if self.combine_table is not None:
            pending = PendingCombine(self.combine_table, partials, output, *self.combine_args)
            if defer_combine:
                return pending
        return None


def spmm_kernel(blk_offsets, hspa_packed, hind, num_nodes, num_edges, embedding_dim, input, output, out_scale=None,
                atomic_out=False, beside_panel=False, defer_combine=False, row_map=None, values=None, xcd_ptr=None):
    """Extensions over the reference wrapper (all default to its behaviour):
    ``out_scale``      float32 device tensor whose first element multiplies every output element (the power-of-two
                       written by ``capi.launch_cast_f32_f16_scaled``); default 1.
    ``atomic_out``     add the product onto ``output`` with float atomics instead of storing it (two-level format: the
                       caller zero-fills ``output`` and the panel kernel adds its part the same way).
    ``beside_panel``   the launch runs beside a panel-kernel workgroup: only tiles that leave it room on the CU.
    ``defer_combine``  with a unit-table schedule, do not sum the cut windows' partial tiles now: return a
                       ``PendingCombine`` (or None when nothing is pending) for the caller to ``run()`` later.
    ``row_map``        int32 [16 W] device tensor: row i of the handle is row ``row_map[i]`` of ``output`` (-1 = padding);
                       handles of a row-permuted CSR (voltrix/reorder.py) write the product through it.  INJECTIVE apart from
                       the -1 entries: the combine pass of the unit-table schedules adds a cut window's tiles to its rows of C
                       by read-add-store, which two windows mapped onto the same rows would race on
                       (profiles/r06/experiment_column_sliced_residual.log ran into exactly that with a many-to-one map).
    ``values``         weighted SpMM (voltrix/weighted.py): the value plane [T, 16, 8] of ``input``'s 16-bit dtype that
                       replaces the bitmaps as the A operand.
    ``xcd_ptr``        int32 [9] device tensor: first window of every XCD's range for the unit-table schedules (the two-level
                       step passes the panel kernel's ranges); default: ranges of equal stages.
    """
    # ---- repeated call on this handle: the plan of the first one (same kernel, same tables, same bits) ----------------------
    from ..utils import KernelTimer

    plan_key = (blk_offsets.data_ptr(), hind.data_ptr(), num_nodes, embedding_dim, input.dtype, input.shape[0], bool(atomic_out),
                bool(beside_panel), bool(defer_combine), row_map.data_ptr() if row_map is not None else 0,
                values.data_ptr() if values is not None else 0, xcd_ptr.data_ptr() if xcd_ptr is not None else 0,
                getattr(hspa_packed, "hash_tag", None), tune_space_mode(), SLAB_POLICY)
    plans = getattr(hspa_packed, "_voltrix_plans", None)
    plan = plans.get(plan_key) if plans is not None else None
    if plan is not None and plan.generation == jit_tuner.generation and KernelTimer.active is None:
        assert input.is_contiguous() and output.is_contiguous() and input.shape[1] == embedding_dim
        assert output.shape[0] == num_nodes and output.shape[1] == embedding_dim and output.dtype == torch.float32
        return plan.launch(input, output, out_scale if out_scale is not None else unit_scale(input.device), values, defer_combine)

    assert blk_offsets.is_cuda and blk_offsets.dtype == torch.int32
    assert hspa_packed.is_cuda and hspa_packed.dtype == torch.uint32
    assert hind.is_cuda and hind.dtype == torch.int32
    assert input.is_cuda and input.dtype in (torch.float, torch.float16, torch.bfloat16) and input.is_contiguous()
    assert output.is_cuda and output.dtype == torch.float and output.is_contiguous()
    assert input.dim() == 2 and input.shape[1] == embedding_dim
    assert output.shape[0] == num_nodes and output.shape[1] == embedding_dim
    elem_bytes = input.element_size()
    assert embedding_dim % (16 // elem_bytes) == 0, "embedding_dim must keep rows 16-byte aligned (voltrix.spmm pads)"
    if out_scale is None:
        out_scale = unit_scale(input.device)
    assert out_scale.is_cuda and out_scale.dtype == torch.float32 and out_scale.numel() >= 1
    if row_map is not None:
        assert row_map.is_cuda and row_map.dtype == torch.int32 and row_map.numel() == 16 * ((num_nodes + 15) // 16)

    if values is not None:
        assert values.is_cuda and values.dtype == input.dtype and elem_bytes == 2 and values.is_contiguous()
        assert values.numel() * 4 == hspa_packed.numel() * 128, "value plane: 128 values per TC block"
    # the stream kernel is a candidate for handles of SHORT windows only (at most STREAM_MAX_BLOCKS_PER_WINDOW TC blocks per
    # window on average; the TC-block count is the size of hspa_packed: no host sync).  On the HBM-resident graphs of long windows
    # the sweep's SAMPLE ranks it first and the full-size step loses 2-13 % to the half-width window tiles (products-like F = 128
    # 3.74 vs 3.30 ms, F = 512 14.8 vs 13.5, papers-like 64.4 vs 62.2, power-law 117.8 vs 112.0: profiles/r05/retune_with_stream_points.log)
    short_windows = hspa_packed.numel() // 4 <= STREAM_MAX_BLOCKS_PER_WINDOW * ((num_nodes + 15) // 16)
    space = tile_space(embedding_dim, elem_bytes, input.dtype == torch.bfloat16,
                       TWO_LEVEL_LDS_BUDGET if beside_panel else None, weighted=values is not None,
                       stream_ok=not atomic_out and row_map is None and (short_windows or tune_space_mode() == "stream"),
                       shallow_ok=short_windows)
    keys = {
        "feature_hash": feature_hash(hspa_packed),
        "embedding_dim": embedding_dim,
        "dtype": str(input.dtype),
        "device": torch.cuda.get_device_name(input.device),
        "two_level": bool(beside_panel),
        "weighted": values is not None,
    }
    if row_map is not None:
        keys["row_map"] = True   # a different space (no stream points): a different choice
    # unit tables / partial-tile buffers: before the choice is made, those of every schedule in the space (the sweep runs
    # them all); afterwards only the chosen schedule's
    chosen = jit_tuner.tuned_point("spmm_kernel", keys).get("SCHED") if jit_tuner.is_tuned("spmm_kernel", keys) else None
    if chosen is not None:
        want = lambda sched: chosen == sched                                    # noqa: E731
    else:
        want = lambda sched: any(p["SCHED"] == sched for p in space)           # noqa: E731
    if want(SCHED_UNITS):
        table = handle_unit_table(blk_offsets, hspa_packed, num_nodes, xcd_ptr=xcd_ptr)
        partials = torch.empty(max(1, table.num_slots) * 16 * embedding_dim, dtype=torch.float32, device=input.device)
        units, unit_ptr, cuts = table.units, table.unit_ptr, table.cuts
        max_units, num_cuts = table.max_units_per_xcd, table.num_cuts
    else:
        table, partials = None, out_scale
        units = unit_ptr = cuts = blk_offsets   # never dereferenced (no SCHED 4 point will run)
        max_units = num_cuts = 0
    if want(SCHED_PAIRS):
        table_p = handle_unit_table(blk_offsets, hspa_packed, num_nodes, pairs=True, xcd_ptr=xcd_ptr)
        partials_p = torch.empty(max(1, table_p.num_slots) * 16 * embedding_dim, dtype=torch.float32, device=input.device)
    else:
        table_p, partials_p = None, out_scale
    if want(SCHED_STREAM):
        table_s = handle_stream_table(blk_offsets, hspa_packed, hind, num_nodes)
        partials_s = torch.empty(max(1, table_s.num_slots) * 16 * embedding_dim, dtype=torch.float32, device=input.device)
    else:
        table_s, partials_s = None, out_scale
    needs_orders = chosen is None or chosen in ORDER_CHUNKS

    def order(sched):
        return window_order(blk_offsets, hspa_packed, num_nodes, sched) if needs_orders else blk_offsets

    def make_args(out, combine_now):
        return (blk_offsets, hspa_packed, hind, num_nodes, num_edges, embedding_dim, input, out,
                order(1), order(2), order(3), out_scale, int(bool(atomic_out)), units, unit_ptr,
                max_units, cuts, num_cuts, partials,
                table_p.units if table_p is not None else blk_offsets, table_p.unit_ptr if table_p is not None else blk_offsets,
                table_p.max_units_per_xcd if table_p is not None else 0, table_p.cuts if table_p is not None else blk_offsets,
                table_p.num_cuts if table_p is not None else 0, partials_p, int(combine_now),
                row_map if row_map is not None else blk_offsets, int(row_map is not None),
                values if values is not None else input, int(input.shape[0]), int(SLAB_POLICY),
                table_s.units if table_s is not None else blk_offsets, table_s.runs if table_s is not None else blk_offsets,
                table_s.run_ptr if table_s is not None else blk_offsets, table_s.max_runs_per_xcd if table_s is not None else 0,
                table_s.cuts if table_s is not None else blk_offsets, table_s.num_cuts if table_s is not None else 0,
                partials_s, torch.cuda.current_stream())

    args = make_args(output, not defer_combine)
    # tuning runs: every candidate is timed with its COMPLETE work (the unit-table schedules with their combine pass, also
    # when the caller defers it), and a launch that adds onto its output gets a scratch one
    tune_args = args
    if len(space) > 1 and chosen is None and (atomic_out or defer_combine):
        tune_args = make_args(torch.zeros_like(output) if atomic_out else output, True)

    def sample_args():
        """(argument tuples of the sample launches, fraction of the handle they cover): called by the tuner only when a
        sweep really runs.  Every range gets its own schedule arrays (window orders, unit tables, partial tiles)."""
        from ..schedule import default_max_stages, stream_tables, unit_table

        num_windows = (num_nodes + 15) // 16
        ranges = sample_ranges(num_windows, int(blk_offsets[num_windows]))
        if len(ranges) == 1 and ranges[0] == (0, num_windows):
            return [tune_args], 1.0
        out_full = tune_args[7]
        launches, covered = [], 0
        for w0, w1 in ranges:
            sub = blk_offsets[w0:w1 + 1]
            n_sub = min(num_nodes, 16 * w1) - 16 * w0
            covered += w1 - w0
            a = list(tune_args)
            a[0], a[3] = sub, n_sub
            a[7] = out_full if row_map is not None else out_full[16 * w0:16 * w0 + n_sub]
            for slot, sched in ((8, 1), (9, 2), (10, 3)):
                a[slot] = window_order(sub, hspa_packed, n_sub, sched) if needs_orders else sub
            if want(SCHED_UNITS):
                t = unit_table(sub, n_sub)
                a[13], a[14], a[15], a[16], a[17] = t.units, t.unit_ptr, t.max_units_per_xcd, t.cuts, t.num_cuts
                a[18] = torch.empty(max(1, t.num_slots) * 16 * embedding_dim, dtype=torch.float32, device=input.device)
            if want(SCHED_PAIRS):
                t = unit_table(sub, n_sub, max(8, int(PAIR_UNIT_FACTOR * default_max_stages(sub, n_sub) / 1.5)))
                a[19], a[20], a[21], a[22], a[23] = t.units, t.unit_ptr, t.max_units_per_xcd, t.cuts, t.num_cuts
                a[24] = torch.empty(max(1, t.num_slots) * 16 * embedding_dim, dtype=torch.float32, device=input.device)
            if want(SCHED_STREAM):
                t = stream_tables(sub, hspa_packed, hind, n_sub, cut_stages=table_s.cut_stages)   # run cost: the sample's own
                a[31], a[32], a[33], a[34], a[35], a[36] = t.units, t.runs, t.run_ptr, t.max_runs_per_xcd, t.cuts, t.num_cuts
                a[37] = torch.empty(max(1, t.num_slots) * 16 * embedding_dim, dtype=torch.float32, device=input.device)
            a[25] = 1                                       # combine now: the candidate's complete work
            if row_map is not None:
                a[26] = row_map[16 * w0:16 * w1]
            launches.append(tuple(a))
        return launches, covered / max(1, num_windows)

    staged = tune_space_mode() != "full"    # "full": every point of the (larger) space, still on the sample

    def tune(use_store=True):
        return jit_tuner.compile_and_tune(
            name="spmm_kernel",
            keys=keys,
            space=space,
            includes=includes,
            arg_defs=arg_defs_for(input.dtype),
            template=template,
            args=tune_args,
            kernel_tag="spmm",
            bench=sweep_bench,
            bucket_keys=lambda: graph_bucket_keys(blk_offsets, num_nodes, keys),
            use_store=use_store,
            sample_args=sample_args,
            stages=sweep_stages if staged else None,
            budget_s=sweep_budget_s,
        )

    runtime = tune()
    rc = runtime(*args)
    signature = jit_tuner._signature("spmm_kernel", keys)
    if rc != 0 and signature in jit_tuner.unvalidated:
        # a persisted / bucket choice that is illegal for THESE arguments: forget it and sweep (its first launch was the
        # validation -- no extra launch is spent on choices that are fine)
        jit_tuner.forget("spmm_kernel", keys)
        runtime = tune(use_store=False)
        rc = runtime(*args)
    jit_tuner.unvalidated.discard(signature)
    assert rc == 0, f"spmm_kernel failed with return code {rc}"
    sched = jit_tuner.tuned_point("spmm_kernel", keys).get("SCHED")
    # ---- the plan for the next call: the marshalled argument list of THIS launch; only the schedule's own partial-tile buffer
    # ---- (when it has cut windows) is allocated per call
    try:
        import ctypes

        from ..jit.template import map_ctype

        fn, _ = runtime.launcher()
        plan = _LaunchPlan()
        plan.fn, plan.generation, plan.device = fn, jit_tuner.generation, input.device
        plan.cargs = [map_ctype(a) for a in args]
        # every tensor whose ADDRESS the list holds stays alive with the plan (round 6): the schedule-table caches on
        # hspa_packed keep one entry each, so a call with another xcd_ptr / blk_offsets replaces a table this plan still points at
        # -- the handle's own tensors, window orders, the three schedules' tables, row map, XCD ranges; NOT the per-call operands
        # (input, output, out_scale, values, partial tiles: patched at every launch) and not hspa_packed (the plan lives on it)
        plan.keepalive = tuple(args[i] for i in (0, 2, 8, 9, 10, 13, 14, 16, 19, 20, 22, 26, 31, 32, 33, 35)
                               if isinstance(args[i], torch.Tensor)) + tuple(
            t for t in (table, table_p, table_s, xcd_ptr) if t is not None)
        plan.partials_index, plan.partials_floats, plan.combine_table, plan.combine_args = None, 0, None, None
        chosen_table = {SCHED_UNITS: (table, 18), SCHED_PAIRS: (table_p, 24), SCHED_STREAM: (table_s, 37)}.get(sched)
        if chosen_table is not None and chosen_table[0] is not None and chosen_table[0].num_slots > 0:
            plan.partials_index = chosen_table[1]
            plan.partials_floats = max(1, chosen_table[0].num_slots) * 16 * embedding_dim
            if chosen_table[0].num_cuts > 0 and defer_combine:
                plan.combine_table = chosen_table[0]
                plan.combine_args = ((num_nodes, embedding_dim, False, None) if sched == SCHED_STREAM else
                                     (num_nodes, embedding_dim, bool(atomic_out), row_map))
        if plans is None:
            plans = {}
            hspa_packed._voltrix_plans = plans
        plans[plan_key] = plan
        del ctypes
    except AttributeError:
        pass
    if defer_combine:
        if sched == SCHED_UNITS and table is not None and table.num_cuts > 0:
            return PendingCombine(table, partials, output, num_nodes, embedding_dim, bool(atomic_out), row_map)
        if sched == SCHED_PAIRS and table_p is not None and table_p.num_cuts > 0:
            return PendingCombine(table_p, partials_p, output, num_nodes, embedding_dim, bool(atomic_out), row_map)
        if sched == SCHED_STREAM and table_s is not None and table_s.num_cuts > 0:
            return PendingCombine(table_s, partials_s, output, num_nodes, embedding_dim, False, None)
    return None
