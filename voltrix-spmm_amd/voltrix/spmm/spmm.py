"""Public operator API: ``csr_preprocess`` and ``spmm`` (reference voltrix/spmm/spmm.py:16-114).

Drop-in contract: same names, argument meaning, assertion behaviour and return types;
the returned handle ``(blk_offsets, hspa_packed, hind)`` has the reference's exact byte layout.
"""
import math
import os

import torch

from ..jit_kernels import (
    csr_fused_preprocess_kernel,
    hmat_gen_kernel,
    hmat_packed_swizzle_kernel,
    preprocess_kernel,
    spmm_kernel,
)
from .. import capi, hybrid, sidecar
from ..project import CSR_PATH_FLAG, FP32_MODE_FLAG, PREPROCESS_FLAG

BLK_H = 16
BLK_W = 8


def csr_preprocess(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, num_cols: int = None):
    """CSR (CPU int32, as in the reference :21-22) -> ``(blk_offsets int32 [W+1], hspa_packed uint32 [4T],
    hind int32 [8T])`` on the current CUDA device: the reference's handle of the WHOLE matrix, byte for byte, whatever
    the environment says.  ``num_cols`` (extension, default ``num_nodes``): the column universe when the ids index
    something else than the ``num_nodes`` rows (row shards over a gathered B).

    Default: one H2D copy of the CSR and the fused GPU preprocess.  ``VOLTRIX_PREPROCESS=reference`` runs the
    reference's own three-stage pipeline (host ``preprocess_kernel``, ``hmat_gen_kernel``,
    ``hmat_packed_swizzle_kernel``, with the transient fp32 ``hspa``); both give identical bytes.
    Duplicate (row, col) entries count once (bitmap), whereas ``torch.sparse.mm`` sums them (quirk 5).

    Acceleration side-car (``VOLTRIX_HYBRID``, default ``auto``; voltrix/hybrid.py::hybrid_mode): when the graph is big
    and dense enough and enough of its edges sit in columns that several rows of a 512-row panel share, the two-level form of
    the same matrix is built as well and attached to the ``hspa_packed`` tensor object; ``voltrix.spmm`` uses it when it
    finds it.  The decision is made HERE, once, from the plan builder's counts (deterministic: no timing, no host sync in
    ``spmm``); the form that is not chosen is never built.  It is a hint, never part of the contract -- a copy of the
    tensor, ``spmm_kernel``, the C-ABI launches all compute the same product from the three tensors alone.  Memory: the
    side-car of the reddit-like graph is 0.68 GB (residual handle 305 MB + plan 374 MB) beside the 617 MB reference handle
    this function must return whatever the format (``voltrix.hybrid.two_level_bytes``).
    """
    assert indptr.is_cpu and indptr.dtype == torch.int32
    assert indices.is_cpu and indices.dtype == torch.int32
    assert indptr.numel() == num_nodes + 1

    if preprocess_mode()[0] != "reference":
        return csr_preprocess_device(indptr.contiguous().cuda(), indices.contiguous().cuda(), num_nodes, num_cols)

    num_edges = indices.numel()
    num_row_windows = math.ceil(num_nodes / BLK_H)
    edge_to_column = torch.zeros(num_edges, dtype=torch.int32)
    edge_to_row = torch.zeros(num_edges, dtype=torch.int32)
    block_partition = torch.zeros(num_row_windows, dtype=torch.int32)
    pointer1 = torch.zeros(block_partition.numel() + 1, dtype=torch.int32)
    preprocess_kernel(edge_list=indices.contiguous(), node_pointer=indptr.contiguous(),
                      block_partition=block_partition, edge_to_column=edge_to_column, edge_to_row=edge_to_row,
                      pointer1=pointer1)

    total_blocks = int(pointer1[-1].item())
    hspa = torch.empty(total_blocks * BLK_H * BLK_W, dtype=torch.float32, device="cuda")
    hind = torch.empty(total_blocks * BLK_W, dtype=torch.int32, device="cuda")
    hspa_packed = torch.empty(hspa.numel() // 32, dtype=torch.uint32, device="cuda")

    indptr_d, indices_d = indptr.cuda(), indices.cuda()
    edge_to_column, edge_to_row = edge_to_column.cuda(), edge_to_row.cuda()
    block_partition, pointer1 = block_partition.cuda(), pointer1.cuda()
    hmat_gen_kernel(node_pointer=indptr_d, edge_list=indices_d, block_partition=block_partition,
                    edge_to_column=edge_to_column, edge_to_row=edge_to_row, pointer1=pointer1, hspa=hspa, hind=hind)
    hmat_packed_swizzle_kernel(block_partition=block_partition, pointer1=pointer1, hspa=hspa, hspa_packed=hspa_packed)
    sidecar.register(hspa_packed, None)   # decided: the reference pipeline never builds a side-car (no "undecided" warning)
    return pointer1, hspa_packed, hind


def preprocess_mode():
    """``VOLTRIX_PREPROCESS``: ``fused`` (default: one H2D copy + the fused GPU preprocess, rank algorithm chosen by the
    library) | ``fused:sort`` / ``fused:bitmap`` / ``fused:mixed`` (the same with that rank algorithm forced where it
    applies: tests, experiments) | ``reference`` (the reference's own three-stage pipeline).  -> (mode, path or None)."""
    mode, _, path = os.getenv(PREPROCESS_FLAG, "fused").partition(":")
    assert mode in ("fused", "reference") and (path or "auto") in ("auto", "sort", "bitmap", "mixed"), \
        f"{PREPROCESS_FLAG}={os.getenv(PREPROCESS_FLAG)}"
    return mode, (path or None)


def csr_preprocess_device(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, num_cols: int = None):
    """``csr_preprocess`` for a CSR that already lives on the GPU (extension: the reference takes CPU tensors only; graph
    pipelines and the row-sharded operator build their shards on the device).  Same handle, same side-car policy.  Handles of
    short windows keep ``indptr`` / ``indices`` THEMSELVES as their CSR side-car (no copy: 4 (nnz + N) bytes saved): do not write into
    them while the handle is in use -- the CSR kernel would see the change, the block format would not."""
    assert indptr.is_cuda and indptr.dtype == torch.int32 and indices.is_cuda and indices.dtype == torch.int32
    assert indptr.numel() == num_nodes + 1
    indptr, indices = indptr.contiguous(), indices.contiguous()
    path = preprocess_mode()[1]
    pointer1, hspa_packed, hind, _ = csr_fused_preprocess_kernel(indptr, indices, num_nodes, num_cols, path=path)
    mode = hybrid.hybrid_mode()
    big_enough = (indices.numel() >= hybrid.AUTO_MIN_EDGES and num_nodes >= hybrid.AUTO_MIN_ROWS
                  and indices.numel() >= hybrid.AUTO_MIN_MEAN_DEGREE * max(1, num_nodes))
    two = None
    if mode == "on" or (mode in ("auto", "tune") and big_enough):
        two = _build_two_level(indptr, indices, num_nodes, num_cols)
    # the decision -- a side-car or "window format" -- is recorded for the MEMORY of hspa_packed (voltrix/sidecar.py): views
    # and re-packed tuples of the handle keep it, it dies with the storage
    sidecar.register(hspa_packed, two)
    if two is None:
        _attach_csr_side_car(hspa_packed, indptr, indices, num_nodes, num_cols)
    return pointer1, hspa_packed, hind


# The CSR row-gather kernel (round 6; spmm_csr_kernels.hpp) is a candidate for handles of SHORT windows (the same bound as the stream
# kernel's: at most this many TC blocks per window on average -- 16 rows that share no column), of at most this many edges (the
# duplicate check below sorts them), without duplicate (row, col) entries (the CSR kernel would count them twice, the bitmaps once).
CSR_MAX_BLOCKS_PER_WINDOW = 48
CSR_MAX_EDGES = 1 << 26


def csr_path_mode() -> str:
    """``VOLTRIX_CSR_PATH``: ``auto`` (default: handles of short windows keep their CSR; the first ``voltrix.spmm`` per (width, dtype)
    times the CSR row-gather kernel against the block-format path -- three calls each, one host sync -- and keeps the faster) |
    ``1`` (always the CSR kernel where a CSR side-car exists: tests) | ``0`` (never)."""
    v = os.getenv(CSR_PATH_FLAG, "auto")
    return "off" if v in ("0", "off") else ("on" if v in ("1", "on") else "auto")


def _attach_csr_side_car(hspa_packed, indptr, indices, num_nodes, num_cols) -> None:
    if csr_path_mode() == "off" or num_nodes == 0 or indices.numel() == 0 or indices.numel() > CSR_MAX_EDGES:
        return
    windows = (num_nodes + 15) // 16
    if hspa_packed.numel() // 4 > CSR_MAX_BLOCKS_PER_WINDOW * windows:
        return
    cols = num_nodes if num_cols is None else int(num_cols)
    deg = (indptr[1:] - indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(num_nodes, device=indptr.device, dtype=torch.int64), deg)
    key = rows * max(cols, 1) + indices.long()
    lo, hi = torch.aminmax(indices)
    distinct = torch.unique(key).numel()
    if int(lo) < 0 or int(hi) >= cols or distinct != indices.numel():      # ids outside the universe / duplicates: block format only
        return
    sidecar.register_csr(hspa_packed, sidecar.CsrSideCar(indptr, indices, num_nodes, cols))


def two_level_of(hspa_packed: torch.Tensor):
    """The ``TwoLevelHandle`` that ``csr_preprocess`` recorded for this handle (any tensor over the same memory), or None."""
    return sidecar.lookup(hspa_packed)[1]


def _build_two_level(indptr_d, indices_d, num_nodes, num_cols, waves=hybrid.DEFAULT_WAVES,
                     row_blocks=hybrid.DEFAULT_ROW_BLOCKS, tau=hybrid.DEFAULT_TAU, min_share=None):
    """Device CSR -> TwoLevelHandle, or None when fewer than ``min_share`` (hybrid.min_shared_fraction()) of the edges land
    on the panel side: too few for the panel kernel to pay for itself (uniform-random graphs, low degrees) -- the window
    format of the whole matrix is the better form then."""
    min_share = hybrid.min_shared_fraction() if min_share is None else min_share
    resid_indptr, resid_indices, plan = hybrid.build_panel_plan(indptr_d, indices_d, num_nodes, num_cols, waves, row_blocks,
                                                                tau, min_share=min_share)
    if plan.num_ksteps == 0 or plan.num_shared_edges < min_share * max(1, indices_d.numel()):
        return None   # the builder stopped after its count phase: nothing of the two-level form was built
    if (waves, row_blocks) == (hybrid.DEFAULT_WAVES, hybrid.DEFAULT_ROW_BLOCKS) and hybrid.panel_dominated(plan) \
            and not hybrid.fused_enabled():
        # the panel kernel would be the critical path: 256-row panels instead (hybrid.PANEL_DOMINATED_RATIO); one more plan build
        del resid_indptr, resid_indices, plan
        resid_indptr, resid_indices, plan = hybrid.build_panel_plan(indptr_d, indices_d, num_nodes, num_cols, waves,
                                                                    hybrid.PANEL_DOMINATED_ROW_BLOCKS, tau, min_share=min_share)
        # columns shared only at 512-row granularity can leave the 256-row plan (nearly) empty: the same test as for the first
        # plan (round 6) -- the window format of the whole matrix is the better form then
        if plan.num_ksteps == 0 or plan.num_shared_edges < min_share * max(1, indices_d.numel()):
            return None
    pointer1, hspa_packed, hind, _ = csr_fused_preprocess_kernel(resid_indptr, resid_indices, num_nodes, num_cols,
                                                                 path=preprocess_mode()[1])
    two = hybrid.TwoLevelHandle(pointer1, hspa_packed, hind, plan, num_nodes, int(indices_d.numel()))
    hybrid.balance_xcd_ranges(two)
    _attach_fused(two)
    return two


def _attach_fused(two) -> None:
    """Stage records of the one-launch kernel (8 x 4 x 16-row panels with a non-empty plan only)."""
    plan = two.plan
    if (hybrid.fused_enabled() and plan.num_ksteps > 0 and plan.waves == hybrid.DEFAULT_WAVES
            and plan.row_blocks == hybrid.DEFAULT_ROW_BLOCKS):
        two.fused = hybrid.build_fused_records(two.blk_offsets, two.hspa_packed, two.hind, two.num_nodes)


def csr_preprocess_hybrid(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, num_cols: int = None,
                          waves: int = hybrid.DEFAULT_WAVES, row_blocks: int = hybrid.DEFAULT_ROW_BLOCKS,
                          tau: int = hybrid.DEFAULT_TAU) -> "hybrid.TwoLevelHandle":
    """Extension (no reference counterpart): the two-level condensed format as an explicit ``TwoLevelHandle`` for
    ``spmm_two_level``.  Columns referenced by >= ``tau`` rows of a ``waves * row_blocks * 16``-row panel go to a panel
    plan (gathered once per panel, ``spmm_panel_kernel``); the remaining edges keep the reference's window format (the
    handle's three tensors describe that RESIDUAL matrix only, which is why this is not a tuple).  The plan may be empty
    (no column reaches ``tau``, or the universe is beyond the builder's limits): then the residual is the whole matrix.
    Same argument checks as ``csr_preprocess``."""
    assert indptr.is_cpu and indptr.dtype == torch.int32
    assert indices.is_cpu and indices.dtype == torch.int32
    assert indptr.numel() == num_nodes + 1
    indptr_d, indices_d = indptr.contiguous().cuda(), indices.contiguous().cuda()
    resid_indptr, resid_indices, plan = hybrid.build_panel_plan(indptr_d, indices_d, num_nodes, num_cols, waves, row_blocks,
                                                                tau)
    pointer1, hspa_packed, hind, _ = csr_fused_preprocess_kernel(resid_indptr, resid_indices, num_nodes, num_cols,
                                                                 path=preprocess_mode()[1])
    two = hybrid.TwoLevelHandle(pointer1, hspa_packed, hind, plan, num_nodes, int(indices.numel()))
    hybrid.balance_xcd_ranges(two)
    _attach_fused(two)
    return two


# fp32 features, VOLTRIX_FP32_MODE=auto (the default since round 5): a handle of SHORT windows -- at most this many gathered rows
# of B per output row -- multiplies the fp32 rows as they are (exact products on v_mfma_f32_16x16x4_f32: the stream kernel's
# EB = 4 tiles); every other handle casts B to fp16 with one power-of-two scale per call.  The cast is two passes over B (an
# absolute maximum, then the conversion: 10 bytes per element); on the low-degree graphs of the reference's evaluation set that
# was more traffic than the product itself (YeastH-like F = 128: 0.80 ms of cast in front of a 0.52 ms product), while gathering
# fp32 rows only doubles the gathered bytes.  Measured break-even: between 5.7 (com-amazon-like: exact wins) and 8.6
# (amazon0601-like: the cast wins) gathered rows per output row.
EXACT_FP32_MAX_GATHERED_ROWS_PER_ROW = 6.0
# ... and up to this many when the operand is at most 32 columns wide: an fp32 row of 32 columns is ONE 128-byte line, exactly what
# the half-line fp16 row costs the CU's request path, so the exact tiles gather "for free" while the cast keeps its launches
# (profiles/r05/experiment_fp32_modes.log, F = 32: ppi-like 0.036 -> 0.025 ms, amazon0601-like 0.108 -> 0.074, DD-like 0.078 -> 0.046;
# FraudYelp-like at 112 gathered rows per row: 0.054 -> 0.155, stays with the cast)
EXACT_FP32_MAX_GATHERED_ROWS_PER_ROW_NARROW = 16.0
EXACT_FP32_NARROW_COLUMNS = 32


def fp32_mode(hspa_packed: torch.Tensor = None, num_nodes: int = 0, num_feats: int = None) -> str:
    """``VOLTRIX_FP32_MODE``: ``fp16`` (scaled cast), ``exact`` (fp32 rows, exact products) or ``auto`` (default: by the handle
    and the operand's width, above) -> "fp16" | "exact".  No host sync: the TC-block count is the size of ``hspa_packed``."""
    mode = os.getenv(FP32_MODE_FLAG, "auto")
    assert mode in ("fp16", "exact", "auto"), f"{FP32_MODE_FLAG}={mode}"
    if mode != "auto":
        return mode
    from ..jit_kernels.spmm import tune_space_mode

    if tune_space_mode() == "none":        # the untuned default tiles: the round-4 behaviour
        return "fp16"
    if hspa_packed is None or num_nodes <= 0 or sidecar.lookup(hspa_packed)[1] is not None:
        return "fp16"
    gathered_rows = 2.0 * hspa_packed.numel()           # 8 per TC block = 8 x numel / 4
    narrow = num_feats is not None and num_feats <= EXACT_FP32_NARROW_COLUMNS
    limit = EXACT_FP32_MAX_GATHERED_ROWS_PER_ROW_NARROW if narrow else EXACT_FP32_MAX_GATHERED_ROWS_PER_ROW
    return "exact" if gathered_rows <= limit * num_nodes else "fp16"


def _operand(feat: torch.Tensor, mode: str = None):
    """``feat`` -> (operand for the kernels, out_scale or None, padded width, exact-fp32 flag).  ``mode``: what fp32 features
    become ("fp16" | "exact"; default: the environment's, ``auto`` counting as fp16 -- callers with a handle decide with it)."""
    assert feat.is_cuda and feat.dim() == 2
    feat = feat.contiguous()
    num_feats = feat.shape[1]
    assert feat.dtype in (torch.float32, torch.float16, torch.bfloat16), f"unsupported feature dtype {feat.dtype}"
    exact = feat.dtype == torch.float32 and (mode or fp32_mode()) == "exact"
    align = 4 if exact else 8
    padded = (num_feats + align - 1) // align * align
    if padded != num_feats:  # keep gathered rows 16-byte aligned
        feat = torch.nn.functional.pad(feat, (0, padded - num_feats))
    if exact or feat.dtype in (torch.float16, torch.bfloat16):
        return feat, None, padded, exact
    # fp32 -> fp16 with one power-of-two scale per call (undone in the kernel's epilogue): keeps fp32's range, which
    # the reference's TF32 multiply has and a plain fp16 cast has not; on the stream, no host sync.
    operand = torch.empty(feat.shape, dtype=torch.float16, device=feat.device)
    out_scale = torch.empty(2, dtype=torch.float32, device=feat.device)
    from ..jit_kernels.spmm import _raw_stream     # the stream's handle without a torch.cuda.Stream object (8 us)

    capi.launch_cast_f32_f16_scaled(feat, operand, out_scale, _raw_stream(feat.device))
    return operand, out_scale, padded, exact


def spmm(blk_offsets: torch.Tensor, hspa_packed: torch.Tensor, hind: torch.Tensor, num_nodes: int, num_edges: int,
         feat: torch.Tensor):
    """``csr(ones) @ feat`` -> new float32 ``[num_nodes, F]`` on ``feat.device``, on the current stream.

    ``feat``: CUDA, 2-D, contiguous; float32 (the reference's only dtype, jit_kernels/spmm.py:53), float16
    (BASELINE.json's headline) or bfloat16.  float32 (``VOLTRIX_FP32_MODE``, default ``auto``): handles of short windows
    multiply the fp32 rows as they are -- exact products, no cast pass (``fp32_mode``) -- every other handle rounds B to fp16
    for the MFMA -- the same 10-bit mantissa as the reference's TF32 rounding (spmm_kernels.cuh:1671), after a per-call
    power-of-two rescale that keeps fp32's range; ``fp16`` / ``exact`` force either.  NOTE the decision depends on the handle AND
    on the operand's width (<= 32 columns: up to 16 gathered rows per output row run exact, wider: up to 6) and on
    ``VOLTRIX_TUNE_SPACE`` (``none`` = always the cast): a forward and a backward product of different widths on one graph may
    round differently (exact vs 2^-11 relative per element of B); pin ``VOLTRIX_FP32_MODE`` when that matters.  ``spmm_reordered``
    makes the same decision for its handle; ``spmm_weighted``, ``spmm_two_level`` and the sharded operator always take the 16-bit
    operand (their kernels have no fp32 tiles).  Every output row is written, including the
    ``num_nodes % 16`` tail the reference skips.  When ``csr_preprocess`` attached the two-level side-car to this very
    ``hspa_packed`` tensor, the 16-bit-operand product runs in that form (same result up to fp32 summation order).
    """
    num_feats = feat.shape[1]
    known, two, csr = sidecar.lookup_both(hspa_packed)
    if _CSR_DEPTH[0]:
        csr = None
    if csr is not None and csr.num_rows == num_nodes and feat.is_cuda and feat.dim() == 2:
        if _csr_choice(csr, blk_offsets, hspa_packed, hind, num_nodes, num_edges, feat) == "csr":
            return _spmm_csr(csr, feat)
    operand, out_scale, padded, exact = _operand(feat, fp32_mode(hspa_packed, num_nodes, feat.shape[1])
                                                 if feat.dtype == torch.float32 else None)
    output = torch.empty((num_nodes, padded), dtype=torch.float32, device=feat.device)
    mode = hybrid.hybrid_mode()
    if not known and mode == "auto" and not exact:
        sidecar.warn_if_unknown(hspa_packed, num_nodes, num_edges, hybrid.AUTO_MIN_EDGES, hybrid.AUTO_MIN_ROWS,
                                hybrid.AUTO_MIN_MEAN_DEGREE)

    def window():
        spmm_kernel(blk_offsets, hspa_packed, hind, num_nodes=num_nodes, num_edges=num_edges, embedding_dim=padded,
                    input=operand, output=output, out_scale=out_scale)

    def two_level():
        _run_two_level(two, operand, output, out_scale, tag_source=hspa_packed)

    if two is None or exact or two.num_nodes != num_nodes or mode == "off":
        assert not sidecar.is_slim(hspa_packed), \
            "this handle was slimmed (voltrix.slim_handle): only its two-level side-car is left, the window-format paths " \
            "(VOLTRIX_HYBRID=0, VOLTRIX_FP32_MODE=exact) need the full reference handle"
        window()
    elif mode != "tune" or sidecar.is_slim(hspa_packed):
        two_level()   # csr_preprocess decided (auto) or the caller did (VOLTRIX_HYBRID=1): stream-ordered, capturable
    else:   # opt-in: the first call for this (width, dtype) times both forms and keeps the faster (host sync!)
        key = (padded, str(operand.dtype))
        if key not in two.format_choice:
            assert not torch.cuda.is_current_stream_capturing(), \
                "VOLTRIX_HYBRID=tune times both formats on the first call: make that call outside the stream capture"
            two.format_choice[key] = _choose_format(hspa_packed, key, window, two_level)
        (two_level if two.format_choice[key] == "two-level" else window)()
    return output if padded == num_feats else output[:, :num_feats].contiguous()


_CSR_DEPTH = [0]      # > 0 while _csr_choice times the block-format path through voltrix.spmm itself
CSR_MIN_GAIN = 0.03   # the CSR kernel must beat the block-format path by this much to be taken


def _spmm_csr(csr, feat: torch.Tensor) -> torch.Tensor:
    """``csr(ones) @ feat`` with the CSR row-gather kernel: fp32 / fp16 / bf16 rows as they are (no cast pass), fp32 result."""
    from ..jit_kernels.spmm import _raw_stream

    feat = feat.contiguous()
    num_feats = feat.shape[1]
    align = 4 if feat.dtype == torch.float32 else 8
    padded = (num_feats + align - 1) // align * align
    if padded != num_feats:
        feat = torch.nn.functional.pad(feat, (0, padded - num_feats))
    output = torch.empty((csr.num_rows, padded), dtype=torch.float32, device=feat.device)
    capi.launch_spmm_csr_rows(csr.indptr, csr.indices, csr.num_rows, feat, output, _raw_stream(feat.device), 1)
    return output if padded == num_feats else output[:, :num_feats].contiguous()


def _csr_choice(csr, blk_offsets, hspa_packed, hind, num_nodes, num_edges, feat) -> str:
    """"csr" | "block" for this (handle, width, dtype): decided once.  ``VOLTRIX_CSR_PATH=1`` forces "csr".  Otherwise the first call
    times both paths (one warm-up -- the block path's own first-call tuning included -- then three calls each; one host sync; not
    inside a stream capture, where the block path is taken) and the choice is remembered on the side-car and, for tagged handles,
    persisted next to the tile choices."""
    mode = csr_path_mode()
    if mode == "off" or feat.dtype not in (torch.float32, torch.float16, torch.bfloat16) or feat.shape[0] < csr.num_cols:
        return "block"
    if mode == "on":
        return "csr"
    # pinned numerics / pinned kernels stay pinned: VOLTRIX_FP32_MODE=fp16 asks for the scaled 16-bit operand (this kernel would
    # compute the exact fp32 product), VOLTRIX_TUNE_SPACE=none / stream name the block-format kernel to run (tests, experiments)
    from ..jit_kernels.spmm import tune_space_mode

    if (feat.dtype == torch.float32 and os.getenv(FP32_MODE_FLAG, "auto") == "fp16") or tune_space_mode() in ("none", "stream"):
        return "block"
    key = (int(feat.shape[1]), str(feat.dtype))
    if key in csr.choice:
        return csr.choice[key]
    if torch.cuda.is_current_stream_capturing():
        return "block"
    from ..jit_kernels import jit_tuner
    from ..jit_kernels.spmm import feature_hash

    tagged = isinstance(getattr(hspa_packed, "hash_tag", None), str)
    signature = ("spmm_csr_path", f"{{'device': '{torch.cuda.get_device_name(hspa_packed.device)}', 'dtype': '{key[1]}', "
                                  f"'embedding_dim': {key[0]}, 'feature_hash': '{feature_hash(hspa_packed) if tagged else ''}'}}")
    if tagged:
        stored = jit_tuner._load_store().get(f"{signature[0]}|{signature[1]}")
        if stored in ("csr", "block"):
            csr.choice[key] = stored
            return stored
    times = {}
    _CSR_DEPTH[0] += 1
    try:
        for name, fn in (("block", lambda: spmm(blk_offsets, hspa_packed, hind, num_nodes, num_edges, feat)),
                         ("csr", lambda: _spmm_csr(csr, feat))):
            fn()
            fn()
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            for _ in range(3):
                fn()
            end.record()
            end.synchronize()
            times[name] = start.elapsed_time(end) / 3
    finally:
        _CSR_DEPTH[0] -= 1
    best = "csr" if times["csr"] < (1.0 - CSR_MIN_GAIN) * times["block"] else "block"
    if os.getenv("VOLTRIX_PRINT_AUTO_TUNE") or os.getenv("VOLTRIX_JIT_DEBUG"):
        print(f"voltrix.spmm path for width {key[0]} {key[1]}: {times} -> {best}")
    csr.choice[key] = best
    if tagged:
        jit_tuner._save_choice(signature, best)
    return best


def _choose_format(hspa_packed, key, window, two_level) -> str:
    """``VOLTRIX_HYBRID=tune`` only.  Time ``window()`` and ``two_level()`` (both write the caller's output: the last one run
    is the chosen one's, run again by the caller) and return the faster form's name.  One host sync; the choice is
    persisted next to the tile choices (``tuned.json``) under the matrix tag and the device, so that a later process skips
    the comparison.  The two forms sum in a different fp32 order: a choice that rests on three timed repetitions can differ
    between processes and ranks, which is why the default mode decides from the plan's statistics instead."""
    from ..jit_kernels import jit_tuner
    from ..jit_kernels.spmm import feature_hash

    signature = ("spmm_format", f"{{'device': '{torch.cuda.get_device_name(hspa_packed.device)}', 'dtype': '{key[1]}', "
                                f"'embedding_dim': {key[0]}, 'feature_hash': '{feature_hash(hspa_packed)}'}}")
    stored = jit_tuner._load_store().get(f"{signature[0]}|{signature[1]}")
    if stored in ("two-level", "window") and hasattr(hspa_packed, "hash_tag"):
        return stored
    times = {}
    for name, fn in (("window", window), ("two-level", two_level)):
        fn()   # tile / schedule sweep of the first call
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(3):
            fn()
        end.record()
        end.synchronize()
        times[name] = start.elapsed_time(end)
    best = min(times, key=times.get)
    if os.getenv("VOLTRIX_PRINT_AUTO_TUNE") or os.getenv("VOLTRIX_JIT_DEBUG"):
        print(f"voltrix.spmm format for width {key[0]} {key[1]}: {times} -> {best}")
    if hasattr(hspa_packed, "hash_tag"):
        jit_tuner._save_choice(signature, best)
    return best


def _run_two_level(two, operand, output, out_scale, tag_source=None, concurrent=True):
    if two.fused is not None and hybrid.fused_enabled():
        hybrid.launch_fused(two.plan, two.fused, operand, output, out_scale=out_scale)   # one launch, C written once
        return
    resid = two.hspa_packed
    if getattr(resid, "hash_tag", None) is None:   # tuner key of the residual launches: the caller's tag + a suffix
        tag = two.hash_tag or getattr(tag_source, "hash_tag", None)
        if isinstance(tag, str):
            resid.hash_tag = tag + "/residual"

    def run_window(atomic):
        return spmm_kernel(two.blk_offsets, resid, two.hind, num_nodes=two.num_nodes,
                           num_edges=two.plan.num_resid_edges, embedding_dim=operand.shape[1], input=operand,
                           output=output, out_scale=out_scale, atomic_out=atomic,
                           beside_panel=two.plan.num_ksteps > 0, defer_combine=True, xcd_ptr=two.window_xcd_ptr)

    hybrid.run_two_level(two.plan, operand, output, run_window, out_scale=out_scale, concurrent=concurrent)


def spmm_two_level(handle: "hybrid.TwoLevelHandle", feat: torch.Tensor, concurrent: bool = True):
    """``csr(ones) @ feat`` for a ``TwoLevelHandle`` (``csr_preprocess_hybrid``): float32 ``[num_nodes, F]``, on the
    current stream (the panel kernel runs beside the window kernel on a side stream, joined through events;
    ``concurrent=False``: both on the caller's stream, the panel kernel adding onto the window kernel's result).  ``feat``
    as for ``spmm``; the panel kernel takes a 16-bit operand, so ``VOLTRIX_FP32_MODE=exact`` is refused unless the plan
    is empty."""
    assert isinstance(handle, hybrid.TwoLevelHandle)
    num_feats = feat.shape[1]
    operand, out_scale, padded, exact = _operand(feat)
    assert not (exact and handle.plan.num_ksteps > 0), \
        "the panel kernel takes a 16-bit operand (unset VOLTRIX_FP32_MODE=exact or use csr_preprocess)"
    output = torch.empty((handle.num_nodes, padded), dtype=torch.float32, device=feat.device)
    _run_two_level(handle, operand, output, out_scale, concurrent=concurrent)
    return output if padded == num_feats else output[:, :num_feats].contiguous()
