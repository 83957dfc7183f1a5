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
from .. import capi, hybrid
from ..project import FP32_MODE_FLAG, PREPROCESS_FLAG

BLK_H = 16
BLK_W = 8


def csr_preprocess(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, num_cols: int = None):
    """CSR (CPU int32, as in the reference :21-22) -> ``(blk_offsets int32 [W+1], hspa_packed uint32 [4T],
    hind int32 [8T])`` on the current CUDA device.  ``num_cols`` (extension, default ``num_nodes``): the column
    universe when the ids index something else than the ``num_nodes`` rows (row shards over a gathered B).

    Default: one H2D copy of the CSR and the fused GPU preprocess.  ``VOLTRIX_PREPROCESS=reference`` runs the
    reference's own three-stage pipeline (host ``preprocess_kernel``, ``hmat_gen_kernel``,
    ``hmat_packed_swizzle_kernel``, with the transient fp32 ``hspa``); both give identical bytes.
    Duplicate (row, col) entries count once (bitmap), whereas ``torch.sparse.mm`` sums them (quirk 5).
    """
    assert indptr.is_cpu and indptr.dtype == torch.int32
    assert indices.is_cpu and indices.dtype == torch.int32
    assert indptr.numel() == num_nodes + 1

    if hybrid.hybrid_enabled():  # VOLTRIX_HYBRID=1: two-level format (voltrix/hybrid.py); the handle is the residual's
        return csr_preprocess_hybrid(indptr, indices, num_nodes, num_cols)
    if os.getenv(PREPROCESS_FLAG, "fused") != "reference":
        pointer1, hspa_packed, hind, _ = csr_fused_preprocess_kernel(
            indptr.contiguous().cuda(), indices.contiguous().cuda(), num_nodes, num_cols)
        return pointer1, hspa_packed, hind

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
    return pointer1, hspa_packed, hind


def csr_preprocess_hybrid(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, num_cols: int = None,
                          waves: int = hybrid.DEFAULT_WAVES, row_blocks: int = hybrid.DEFAULT_ROW_BLOCKS,
                          tau: int = hybrid.DEFAULT_TAU):
    """Extension (no reference counterpart): two-level condensed format.  Columns referenced by >= ``tau`` rows of a
    ``waves * row_blocks * 16``-row panel go to a panel plan (gathered once per panel, ``spmm_panel_kernel``); the
    remaining edges go through the ordinary ``csr_preprocess``.  Returns the reference-format handle of the RESIDUAL
    matrix with the plan attached as ``hspa_packed.panel_plan`` -- ``spmm`` adds both parts.  When fewer than
    ``VOLTRIX_HYBRID_MIN_SHARE`` (default 0.2) of the edges land on the panel side the plan is dropped and the handle is
    the plain window format of the whole matrix.  The same arguments and assertions as ``csr_preprocess``."""
    assert indptr.is_cpu and indptr.dtype == torch.int32
    assert indices.is_cpu and indices.dtype == torch.int32
    assert indptr.numel() == num_nodes + 1
    indptr_d, indices_d = indptr.contiguous().cuda(), indices.contiguous().cuda()
    resid_indptr, resid_indices, plan = hybrid.build_panel_plan(indptr_d, indices_d, num_nodes, num_cols, waves, row_blocks,
                                                                tau)
    if plan.num_shared_edges < hybrid.min_shared_fraction() * max(1, indices.numel()):
        # too few edges sit in shared columns for the panel kernel to pay for itself (uniform-random graphs, low degrees):
        # keep the whole matrix in the window format; the empty plan makes spmm skip the panel kernel
        resid_indptr, resid_indices = indptr_d, indices_d
        plan = hybrid.empty_plan(num_nodes, waves, row_blocks, tau, indptr_d.device, indices.numel())
    pointer1, hspa_packed, hind, _ = csr_fused_preprocess_kernel(resid_indptr, resid_indices, num_nodes, num_cols)
    hspa_packed.panel_plan = plan
    return pointer1, hspa_packed, hind


def spmm(blk_offsets: torch.Tensor, hspa_packed: torch.Tensor, hind: torch.Tensor, num_nodes: int, num_edges: int,
         feat: torch.Tensor):
    """``csr(ones) @ feat`` -> new float32 ``[num_nodes, F]`` on ``feat.device``, on the current stream.

    ``feat``: CUDA, 2-D, contiguous; float32 (the reference's only dtype, jit_kernels/spmm.py:53), float16
    (BASELINE.json's headline) or bfloat16.  float32 is rounded to fp16 for the MFMA -- the same 10-bit mantissa as the
    reference's TF32 rounding (spmm_kernels.cuh:1671), after a per-call power-of-two rescale that keeps fp32's range --
    unless ``VOLTRIX_FP32_MODE=exact``, which keeps exact fp32 products.  Every output row is written, including the ``num_nodes % 16`` tail the reference skips.
    """
    assert feat.is_cuda and feat.dim() == 2
    feat = feat.contiguous()
    num_feats = feat.shape[1]
    assert feat.dtype in (torch.float32, torch.float16, torch.bfloat16), f"unsupported feature dtype {feat.dtype}"

    exact = feat.dtype == torch.float32 and os.getenv(FP32_MODE_FLAG, "fp16") == "exact"
    align = 4 if exact else 8
    padded = (num_feats + align - 1) // align * align
    if padded != num_feats:  # keep gathered rows 16-byte aligned
        feat = torch.nn.functional.pad(feat, (0, padded - num_feats))
    out_scale = None
    if exact or feat.dtype in (torch.float16, torch.bfloat16):
        operand = feat
    else:
        # fp32 -> fp16 with one power-of-two scale per call (undone in the kernel's epilogue): keeps fp32's range, which
        # the reference's TF32 multiply has and a plain fp16 cast has not; on the stream, no host sync.
        operand = torch.empty(feat.shape, dtype=torch.float16, device=feat.device)
        out_scale = torch.empty(2, dtype=torch.float32, device=feat.device)
        capi.launch_cast_f32_f16_scaled(feat, operand, out_scale, torch.cuda.current_stream().cuda_stream)
    output = torch.empty((num_nodes, padded), dtype=torch.float32, device=feat.device)

    def run_window():
        spmm_kernel(blk_offsets, hspa_packed, hind, num_nodes=num_nodes, num_edges=num_edges, embedding_dim=padded,
                    input=operand, output=output, out_scale=out_scale)

    plan = getattr(hspa_packed, "panel_plan", None)
    if plan is None:
        run_window()
    else:
        # two-level format: the handle covers the residual edges, the panel kernel adds the shared-column part
        assert not exact, "the panel kernel takes a 16-bit operand (unset VOLTRIX_FP32_MODE=exact or use csr_preprocess)"
        assert plan.num_nodes == num_nodes
        hybrid.spmm_two_level(plan, operand, output, run_window, out_scale=out_scale)
    return output if padded == num_feats else output[:, :num_feats].contiguous()
