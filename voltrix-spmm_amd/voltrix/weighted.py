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


# A value plane is 512 B per TC block in fp32 (the master the 16-bit planes are rounded from) and 256 B in a 16-bit type.  Above
# this many bytes of fp32 master the handle keeps NO master: the 16-bit plane of ``plane_dtype`` is built directly, window chunk
# by window chunk (papers-like: 105 GB of master beside 52 GB of fp16 plane did not fit next to B and C on one GPU).
MASTER_PLANE_MAX_BYTES = 32 << 30
CHUNK_EDGES = 1 << 27            # edges per chunk of windows: bounds the builder's edge-sized temporaries (five int64 arrays)


@dataclasses.dataclass(eq=False)
class WeightedHandle:
    blk_offsets: torch.Tensor    # int32 [W+1]      the reference's handle of the pattern
    hspa_packed: torch.Tensor    # uint32 [4T]
    hind: torch.Tensor           # int32 [8T]
    values32: torch.Tensor       # float32 [T, 16, 8]   summed edge values per (TC block, row, condensed column); None on handles
                                 # too large for a master (MASTER_PLANE_MAX_BYTES): only the planes built at preprocess exist
    num_nodes: int
    num_edges: int
    planes: dict = dataclasses.field(default_factory=dict)   # dtype -> value plane in that 16-bit type


def _chunk_plane(indptr, indices, values, blk_offsets, w0, w1, num_nodes, num_cols):
    """float32 [blocks of windows w0 .. w1) * 128]: the value plane of a range of windows (definition: ``value_plane``)."""
    dev = indptr.device
    r0, r1 = 16 * w0, min(16 * w1, num_nodes)
    e0, e1 = int(indptr[r0]), int(indptr[r1])
    deg = (indptr[r0 + 1:r1 + 1] - indptr[r0:r1]).long()
    rows = torch.repeat_interleave(torch.arange(r0, r1, device=dev, dtype=torch.int64), deg)
    win = rows // 16 - w0
    key = win * num_cols + indices[e0:e1].long()
    uniq = torch.unique(key)                                   # sorted: window-major, columns ascending inside a window
    first = torch.searchsorted(uniq, torch.arange(w1 - w0, device=dev, dtype=torch.int64) * num_cols)
    q = torch.searchsorted(uniq, key) - first[win]
    del key, uniq
    b0 = int(blk_offsets[w0])
    block = blk_offsets.long()[win + w0] - b0 + q // 8
    flat = (block * 16 + rows % 16) * 8 + q % 8
    del block, rows, win
    plane = torch.zeros((int(blk_offsets[w1]) - b0) * 128, dtype=torch.float32, device=dev)
    plane.index_add_(0, flat, values[e0:e1].float())
    return plane


def value_plane(indptr: torch.Tensor, indices: torch.Tensor, values: torch.Tensor, blk_offsets: torch.Tensor,
                num_nodes: int, num_cols: int, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """``dtype`` [T, 16, 8]: edge (row, col) of window w lands at TC block ``blk_offsets[w] + q // 8``, row ``row % 16``,
    column ``q % 8`` with q = rank of col among the window's sorted distinct columns (the format's definition, SURVEY.md
    Appendix A); duplicates add (in fp32, rounded once).  Device tensor ops (sort / searchsorted / index_add) over chunks of
    windows of about ``CHUNK_EDGES`` edges: plumbing, once per matrix."""
    dev = indptr.device
    num_windows = (num_nodes + 15) // 16
    total = int(blk_offsets[-1])
    out = torch.empty(total * 128, dtype=dtype, device=dev)
    if num_windows == 0:
        return out.view(total, 16, 8)
    # window boundaries of the chunks: prefix of edges per window, cut every CHUNK_EDGES
    win_edges = indptr[(torch.arange(0, num_windows + 1, device=dev) * 16).clamp(max=num_nodes)].long()
    targets = torch.arange(CHUNK_EDGES, int(win_edges[-1]) + CHUNK_EDGES, CHUNK_EDGES, device=dev)
    cuts = torch.unique(torch.cat([torch.zeros(1, dtype=torch.int64, device=dev),
                                   torch.searchsorted(win_edges, targets).clamp(max=num_windows),
                                   torch.full((1,), num_windows, dtype=torch.int64, device=dev)])).tolist()
    for w0, w1 in zip(cuts[:-1], cuts[1:]):
        if w1 <= w0:
            continue
        chunk = _chunk_plane(indptr, indices, values, blk_offsets, w0, w1, num_nodes, num_cols)
        out[int(blk_offsets[w0]) * 128:int(blk_offsets[w1]) * 128] = chunk.to(dtype)
        del chunk
    return out.view(total, 16, 8)


def csr_preprocess_weighted(indptr: torch.Tensor, indices: torch.Tensor, values: torch.Tensor, num_nodes: int,
                            num_cols: int = None, plane_dtype: torch.dtype = None) -> WeightedHandle:
    """CSR with values (CPU or CUDA; int32 ``indptr`` / ``indices``, floating ``values``) -> ``WeightedHandle``.
    ``plane_dtype`` (float16 / bfloat16): build that 16-bit plane now; handles whose fp32 master would exceed
    ``MASTER_PLANE_MAX_BYTES`` keep only it (default float16) -- ``spmm_weighted`` with an operand of the other type then raises."""
    assert indptr.dtype == torch.int32 and indices.dtype == torch.int32 and indptr.numel() == num_nodes + 1
    assert values.numel() == indices.numel() and values.is_floating_point()
    assert plane_dtype in (None, torch.float16, torch.bfloat16)
    indptr_d, indices_d, values_d = indptr.contiguous().cuda(), indices.contiguous().cuda(), values.contiguous().cuda()
    num_cols = num_nodes if num_cols is None else int(num_cols)
    pointer1, hspa_packed, hind, _ = csr_fused_preprocess_kernel(indptr_d, indices_d, num_nodes, num_cols)
    universe = max(num_cols, int(indices_d.max()) + 1) if indices_d.numel() else num_cols
    total = int(pointer1[-1])
    if total * 512 > MASTER_PLANE_MAX_BYTES:
        dt = plane_dtype or torch.float16
        handle = WeightedHandle(pointer1, hspa_packed, hind, None, num_nodes, int(indices.numel()))
        handle.planes[dt] = value_plane(indptr_d, indices_d, values_d, pointer1, num_nodes, universe, dtype=dt)
        return handle
    plane = value_plane(indptr_d, indices_d, values_d, pointer1, num_nodes, universe)
    handle = WeightedHandle(pointer1, hspa_packed, hind, plane, num_nodes, int(indices.numel()))
    if plane_dtype is not None:
        handle.planes[plane_dtype] = plane.to(plane_dtype).contiguous()
    return handle


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
        assert handle.values32 is not None, (f"this handle was built without an fp32 master (too large) and holds only the "
                                             f"{list(handle.planes)} plane(s): pass plane_dtype={operand.dtype} to csr_preprocess_weighted")
        handle.planes[operand.dtype] = handle.values32.to(operand.dtype).contiguous()
    output = torch.empty((handle.num_nodes, padded), dtype=torch.float32, device=feat.device)
    spmm_kernel(handle.blk_offsets, handle.hspa_packed, handle.hind, num_nodes=handle.num_nodes,
                num_edges=handle.num_edges, embedding_dim=padded, input=operand, output=output, out_scale=out_scale,
                values=handle.planes[operand.dtype])
    return output if padded == num_feats else output[:, :num_feats].contiguous()
