"""Weighted SpMM: ``C = A @ B`` with edge VALUES (no reference counterpart -- the reference
stores A as 1-bit-per-entry bitmaps and ignores values, bmat_kernels.cuh:100-103, jit_kernels/spmm.py:53-54).

The block format keeps its three tensors (they still say where the edges are and which rows of B a TC block gathers); a
VALUE PLANE rides beside the bitmaps: ``values[T, 16, 8]`` in the dense operand's 16-bit type, zeros where there is no edge.
The window kernel's ``WEIGHTED`` tiles (traits.hpp) fetch a stage's 4 x 128 values with one more 1-KiB LDS-DMA and use
them as the MFMA A fragment as they are -- row R of TC block g is exactly what lane 16 g + R holds -- instead of expanding
bitmap nibbles; everything else (gathers, ring, schedules, unit tables, tuner) is the binary kernel's.

    handle = voltrix.csr_preprocess_weighted(indptr, indices, values, num_nodes)
    out = voltrix.spmm_weighted(handle, feat)            # == torch.sparse_csr_tensor(indptr, indices, values) @ feat

Duplicate (row, col) entries ADD their values, as in ``torch.sparse.mm`` (the binary format counts them once).
A values are rounded to the operand's 16-bit type (fp16 keeps the 10 mantissa bits the reference's TF32 multiply would
keep); float32 features take the scaled-fp16 path of ``voltrix.spmm``.
"""
from __future__ import annotations

import dataclasses

import torch

from .jit_kernels import csr_fused_preprocess_kernel, spmm_kernel


@dataclasses.dataclass(eq=False)
class WeightedHandle:
    blk_offsets: torch.Tensor    # int32 [W+1]      the reference's handle of the pattern
    hspa_packed: torch.Tensor    # uint32 [4T]
    hind: torch.Tensor           # int32 [8T]
    values32: torch.Tensor       # float32 [T, 16, 8]   summed edge values per (TC block, row, condensed column)
    num_nodes: int
    num_edges: int
    planes: dict = dataclasses.field(default_factory=dict)   # dtype -> value plane in that 16-bit type


def value_plane(indptr: torch.Tensor, indices: torch.Tensor, values: torch.Tensor, blk_offsets: torch.Tensor,
                num_nodes: int, num_cols: int) -> torch.Tensor:
    """float32 [T, 16, 8]: edge (row, col) of window w lands at TC block ``blk_offsets[w] + q // 8``, row ``row % 16``,
    column ``q % 8`` with q = rank of col among the window's sorted distinct columns (the format's definition, SURVEY.md
    Appendix A).  Device tensor ops (sort / searchsorted / index_add): plumbing, once per matrix."""
    dev = indptr.device
    deg = (indptr[1:] - indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(num_nodes, device=dev, dtype=torch.int64), deg)
    win = rows // 16
    key = win * num_cols + indices.long()
    uniq = torch.unique(key)                                   # sorted: window-major, columns ascending inside a window
    num_windows = (num_nodes + 15) // 16
    first = torch.searchsorted(uniq, torch.arange(num_windows, device=dev, dtype=torch.int64) * num_cols)
    q = torch.searchsorted(uniq, key) - first[win]
    block = blk_offsets.long()[win] + q // 8
    flat = (block * 16 + rows % 16) * 8 + q % 8
    total = int(blk_offsets[-1])
    plane = torch.zeros(total * 128, dtype=torch.float32, device=dev)
    plane.index_add_(0, flat, values.float())
    return plane.view(total, 16, 8)


def csr_preprocess_weighted(indptr: torch.Tensor, indices: torch.Tensor, values: torch.Tensor, num_nodes: int,
                            num_cols: int = None) -> WeightedHandle:
    """CSR with values (CPU or CUDA; int32 ``indptr`` / ``indices``, floating ``values``) -> ``WeightedHandle``."""
    assert indptr.dtype == torch.int32 and indices.dtype == torch.int32 and indptr.numel() == num_nodes + 1
    assert values.numel() == indices.numel() and values.is_floating_point()
    indptr_d, indices_d, values_d = indptr.contiguous().cuda(), indices.contiguous().cuda(), values.contiguous().cuda()
    num_cols = num_nodes if num_cols is None else int(num_cols)
    pointer1, hspa_packed, hind, _ = csr_fused_preprocess_kernel(indptr_d, indices_d, num_nodes, num_cols)
    universe = max(num_cols, int(indices_d.max()) + 1) if indices_d.numel() else num_cols
    plane = value_plane(indptr_d, indices_d, values_d, pointer1, num_nodes, universe)
    return WeightedHandle(pointer1, hspa_packed, hind, plane, num_nodes, int(indices.numel()))


def spmm_weighted(handle: WeightedHandle, feat: torch.Tensor, hash_tag: str = None) -> torch.Tensor:
    """``csr(values) @ feat`` -> float32 [num_nodes, F] on the current stream.  ``feat``: CUDA, 2-D; float16 / bfloat16, or
    float32 (rounded to fp16 after one power-of-two rescale, as in ``voltrix.spmm``)."""
    from .spmm.spmm import _operand

    assert isinstance(handle, WeightedHandle)
    if hash_tag is not None and getattr(handle.hspa_packed, "hash_tag", None) is None:
        handle.hspa_packed.hash_tag = hash_tag
    num_feats = feat.shape[1]
    operand, out_scale, padded, exact = _operand(feat)
    assert not exact, "the weighted kernel takes a 16-bit operand (unset VOLTRIX_FP32_MODE=exact)"
    if operand.dtype not in handle.planes:
        handle.planes[operand.dtype] = handle.values32.to(operand.dtype).contiguous()
    output = torch.empty((handle.num_nodes, padded), dtype=torch.float32, device=feat.device)
    spmm_kernel(handle.blk_offsets, handle.hspa_packed, handle.hind, num_nodes=handle.num_nodes,
                num_edges=handle.num_edges, embedding_dim=padded, input=operand, output=output, out_scale=out_scale,
                values=handle.planes[operand.dtype])
    return output if padded == num_feats else output[:, :num_feats].contiguous()
