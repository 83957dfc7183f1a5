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
    # round 6 -- values of the form v_ij = r_i * c_j (the normalised adjacencies of GCN / mean aggregation): no value plane at
    # all; C = diag(r) (A_binary (diag(c) B)) on the BINARY operator, two-level side-car included
    row_scale: torch.Tensor = None     # float32 [num_nodes]
    col_scale: torch.Tensor = None     # float32 [num_cols]

    csr: tuple = None                  # (indptr, indices, values or None, num_cols) on the device -- what a separable handle builds a value
                                       # plane from should a call be better off with one (``separable_pays``), and what ``update_values``
                                       # derives the edge -> plane map of a general handle from (kept up to KEEP_CSR_MAX_EDGES edges)
    edge_slot: torch.Tensor = None     # int64 [nnz]: ``edge_slots`` of this handle, built by the first ``update_values``
    slot_duplicates: bool = None       # some (row, col) pair occurs more than once: updates ADD per slot instead of storing
    path_choice: dict = dataclasses.field(default_factory=dict)   # (width, dtype) -> "csr" | "plane": ``_weighted_path``'s measured choice

    @property
    def separable(self) -> bool:
        return self.row_scale is not None


def _own_values(values: torch.Tensor) -> torch.Tensor:
    """float32 contiguous copy of the edge values that the HANDLE owns: the CSR kernel with values reads them at every call, so a caller
    that later writes into its own tensor must not change what the handle computes (the planes would still hold the old values)."""
    v = values.float().contiguous()
    return v.clone() if v.data_ptr() == values.data_ptr() else v


def _chunk_slots(indptr, indices, blk_offsets, w0, w1, num_nodes, num_cols):
    """int64 [edges of windows w0 .. w1)]: where every edge of a range of windows sits in the range's part of the value plane
    (definition: ``value_plane``; offsets relative to the first TC block of window w0)."""
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
    return (block * 16 + rows % 16) * 8 + q % 8, e0, e1


def _chunk_plane(indptr, indices, values, blk_offsets, w0, w1, num_nodes, num_cols):
    """float32 [blocks of windows w0 .. w1) * 128]: the value plane of a range of windows (definition: ``value_plane``)."""
    flat, e0, e1 = _chunk_slots(indptr, indices, blk_offsets, w0, w1, num_nodes, num_cols)
    plane = torch.zeros((int(blk_offsets[w1]) - int(blk_offsets[w0])) * 128, dtype=torch.float32, device=indptr.device)
    plane.index_add_(0, flat, values[e0:e1].float())
    return plane


def _window_chunks(indptr, num_nodes):
    """Window boundaries of chunks of about ``CHUNK_EDGES`` edges (bounded temporaries in the builders)."""
    dev = indptr.device
    num_windows = (num_nodes + 15) // 16
    win_edges = indptr[(torch.arange(0, num_windows + 1, device=dev) * 16).clamp(max=num_nodes)].long()
    targets = torch.arange(CHUNK_EDGES, int(win_edges[-1]) + CHUNK_EDGES, CHUNK_EDGES, device=dev)
    return torch.unique(torch.cat([torch.zeros(1, dtype=torch.int64, device=dev),
                                   torch.searchsorted(win_edges, targets).clamp(max=num_windows),
                                   torch.full((1,), num_windows, dtype=torch.int64, device=dev)])).tolist()


def edge_slots(indptr: torch.Tensor, indices: torch.Tensor, blk_offsets: torch.Tensor, num_nodes: int, num_cols: int) -> torch.Tensor:
    """int64 [nnz]: the element of the flat value plane [T * 128] every CSR entry lands on (``value_plane``'s definition), in CSR
    order.  What ``update_values`` scatters new values through: the pattern's part of building a plane (sorts, ranks) is done
    once, a change of values is one pass."""
    out = torch.empty(indices.numel(), dtype=torch.int64, device=indptr.device)
    if (num_nodes + 15) // 16 == 0:
        return out
    cuts = _window_chunks(indptr, num_nodes)
    for w0, w1 in zip(cuts[:-1], cuts[1:]):
        if w1 <= w0:
            continue
        flat, e0, e1 = _chunk_slots(indptr, indices, blk_offsets, w0, w1, num_nodes, num_cols)
        out[e0:e1] = flat + int(blk_offsets[w0]) * 128
    return out


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
    cuts = _window_chunks(indptr, num_nodes)       # prefix of edges per window, cut every CHUNK_EDGES
    for w0, w1 in zip(cuts[:-1], cuts[1:]):
        if w1 <= w0:
            continue
        chunk = _chunk_plane(indptr, indices, values, blk_offsets, w0, w1, num_nodes, num_cols)
        out[int(blk_offsets[w0]) * 128:int(blk_offsets[w1]) * 128] = chunk.to(dtype)
        del chunk
    return out.view(total, 16, 8)


SEPARABLE_TOLERANCE = 2.0 ** -13     # |v_ij / (r_i c_j) - 1| below this counts as separable: a quarter of the fp16 rounding the
                                      # value plane applies to every value anyway (2^-11)
SEPARABLE_SWEEPS = 64
KEEP_CSR_MAX_EDGES = 1 << 30         # general handles keep their device CSR (4 bytes per edge) up to this size: update_values
SEPARABLE_MAX_EDGES = 1 << 29        # the detection holds five edge-sized 64-bit arrays and sorts one of them: above this many edges it is
                                      # not tried (papers-like: 1.6 G edges beside a 28 GB B and a 57 GB C) -- state the factors instead


def separable_scales(indptr: torch.Tensor, indices: torch.Tensor, values: torch.Tensor, num_nodes: int, num_cols: int,
                     tolerance: float = SEPARABLE_TOLERANCE, sweeps: int = SEPARABLE_SWEEPS):
    """``(row_scale float32 [N], col_scale float32 [M])`` with ``values[e] == row_scale[row(e)] * col_scale[col(e)]`` within
    ``tolerance`` (relative) for every edge, or None.  Positive values only; duplicate (row, col) entries: None (they ADD in the
    weighted product but count once in the binary format).  log v_ij = a_i + b_j solved by alternating row / column means
    (exact after one sweep for the common normalisations -- 1 / sqrt(d_i d_j), 1 / d_i, 1 / d_j --, a few dozen sweeps for general
    separable values on a well-connected graph); the answer is CHECKED edge by edge, so a slow or failed convergence only means
    "not separable" and the general value plane.  Torch tensor ops on the CSR's device; a handful of host reads."""
    dev = indptr.device
    e = int(indices.numel())
    if e == 0:
        return torch.ones(num_nodes, device=dev), torch.ones(num_cols, device=dev)
    v = values.double()
    if not bool((v > 0).all()) or not bool(torch.isfinite(v).all()):
        return None
    deg = (indptr[1:] - indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(num_nodes, device=dev, dtype=torch.int64), deg)
    cols = indices.long()
    if int(cols.max()) >= num_cols:
        return None
    key = rows * num_cols + cols
    if int(torch.unique(key).numel()) != e:
        return None
    del key
    logv = torch.log(v)
    cdeg = torch.bincount(cols, minlength=num_cols).double().clamp(min=1.0)
    rdeg = deg.double().clamp(min=1.0)
    a = torch.zeros(num_nodes, dtype=torch.float64, device=dev)
    b = torch.zeros(num_cols, dtype=torch.float64, device=dev)
    log_tol = float(torch.log1p(torch.tensor(tolerance, dtype=torch.float64)))
    for sweep in range(sweeps):
        a = torch.zeros_like(a).index_add_(0, rows, logv - b[cols]) / rdeg
        b = torch.zeros_like(b).index_add_(0, cols, logv - a[rows]) / cdeg
        if sweep in (0, 1, 3, 7, 15, 31, sweeps - 1):
            if float((logv - a[rows] - b[cols]).abs().max()) <= 0.5 * log_tol:
                break
    else:
        return None
    if float((logv - a[rows] - b[cols]).abs().max()) > 0.5 * log_tol:
        return None
    # balance the two factors (their product is what is defined): both near 1 keeps the scaled B inside fp16's range
    shift = 0.5 * (float(a.mean()) - float(b.mean()))
    return torch.exp(a - shift).float(), torch.exp(b + shift).float()


def csr_preprocess_weighted(indptr: torch.Tensor, indices: torch.Tensor, values: torch.Tensor, num_nodes: int,
                            num_cols: int = None, plane_dtype: torch.dtype = None, separable="auto",
                            row_scale: torch.Tensor = None, col_scale: torch.Tensor = None) -> WeightedHandle:
    """CSR with values (CPU or CUDA; int32 ``indptr`` / ``indices``, floating ``values``) -> ``WeightedHandle``.
    ``plane_dtype`` (float16 / bfloat16): build that 16-bit plane now; handles whose fp32 master would exceed
    ``MASTER_PLANE_MAX_BYTES`` keep only it (default float16) -- ``spmm_weighted`` with an operand of the other type then raises.

    Round 6 -- SEPARABLE values ``v_ij = r_i c_j`` (symmetric / row / column normalised adjacencies: what a GCN or a mean
    aggregator multiplies by) need no value plane: ``C = diag(r) A (diag(c) B)`` runs on the binary operator, with its two-level
    side-car and every schedule of it, at the cost of one pass over B before and one over C after (``scale_rows``).
    ``separable="auto"`` (default) detects it (``separable_scales``: exact check, edge by edge); ``row_scale`` / ``col_scale``
    (float [N] / [M]) state it (``values`` may then be None); ``separable=False`` forces the general value plane."""
    assert indptr.dtype == torch.int32 and indices.dtype == torch.int32 and indptr.numel() == num_nodes + 1
    assert plane_dtype in (None, torch.float16, torch.bfloat16)
    num_cols = num_nodes if num_cols is None else int(num_cols)
    indptr_d, indices_d = indptr.contiguous().cuda(), indices.contiguous().cuda()
    scales = None
    if row_scale is not None or col_scale is not None:
        dev = indptr_d.device
        scales = (torch.ones(num_nodes, device=dev) if row_scale is None else row_scale.to(dev, torch.float32).contiguous(),
                  torch.ones(num_cols, device=dev) if col_scale is None else col_scale.to(dev, torch.float32).contiguous())
        assert scales[0].numel() == num_nodes and scales[1].numel() == num_cols
    else:
        assert values is not None and values.numel() == indices.numel() and values.is_floating_point()
        if separable is True or (separable == "auto" and indices.numel() <= SEPARABLE_MAX_EDGES):
            scales = separable_scales(indptr_d, indices_d, values.contiguous().cuda(), num_nodes, num_cols)
            assert scales is not None or separable == "auto", "separable=True but the values do not factor as r_i * c_j"
    if scales is not None:
        from .spmm.spmm import csr_preprocess_device

        pointer1, hspa_packed, hind = csr_preprocess_device(indptr_d, indices_d, num_nodes, num_cols)   # side-car policy included
        return WeightedHandle(pointer1, hspa_packed, hind, None, num_nodes, int(indices.numel()), row_scale=scales[0],
                              col_scale=scales[1],
                              csr=(indptr_d, indices_d, _own_values(values.cuda()) if values is not None else None, num_cols))
    assert values.numel() == indices.numel() and values.is_floating_point()
    values_d = values.contiguous().cuda()
    pointer1, hspa_packed, hind, _ = csr_fused_preprocess_kernel(indptr_d, indices_d, num_nodes, num_cols)
    universe = max(num_cols, int(indices_d.max()) + 1) if indices_d.numel() else num_cols
    total = int(pointer1[-1])
    # kept for update_values and for the CSR row-gather kernel with values (``_weighted_path``): 8 bytes per edge
    csr = (indptr_d, indices_d, _own_values(values_d), universe) if indices_d.numel() <= KEEP_CSR_MAX_EDGES else None
    if total * 512 > MASTER_PLANE_MAX_BYTES:
        dt = plane_dtype or torch.float16
        handle = WeightedHandle(pointer1, hspa_packed, hind, None, num_nodes, int(indices.numel()), csr=csr)
        handle.planes[dt] = value_plane(indptr_d, indices_d, values_d, pointer1, num_nodes, universe, dtype=dt)
        return handle
    plane = value_plane(indptr_d, indices_d, values_d, pointer1, num_nodes, universe)
    handle = WeightedHandle(pointer1, hspa_packed, hind, plane, num_nodes, int(indices.numel()), csr=csr)
    if plane_dtype is not None:
        handle.planes[plane_dtype] = plane.to(plane_dtype).contiguous()
    return handle


def update_values(handle: WeightedHandle, values: torch.Tensor) -> WeightedHandle:
    """New edge values on the SAME sparsity pattern, in the CSR order the handle was built from (attention coefficients, edge
    weights that are trained: they change every step, the pattern never does).  The pattern's share of building a value plane --
    ranking every column inside its window: sorts and searches over all edges -- is done once (``edge_slots``, cached on the
    handle: 8 bytes per edge) and a change of values is one scatter per plane the handle holds; the handle (tile choices,
    launch plans, unit tables) is otherwise untouched.  Returns ``handle``.

    Separable handles: the new values are checked like the first ones (``separable_scales``); if they factor the two scale
    vectors are replaced, otherwise the handle becomes a general one (value plane built now) -- build with ``separable=False``
    when the values are known to be general and this check is not wanted."""
    assert isinstance(handle, WeightedHandle) and values.is_floating_point() and values.numel() == handle.num_edges
    assert handle.csr is not None, (f"this handle did not keep its CSR (more than {KEEP_CSR_MAX_EDGES} edges): values cannot be "
                                    f"replaced in place -- rebuild it with csr_preprocess_weighted")
    indptr, indices, _, num_cols = handle.csr
    values = values.contiguous().to(indptr.device)
    if handle.separable:
        scales = separable_scales(indptr, indices, values, handle.num_nodes, num_cols) if values.numel() <= SEPARABLE_MAX_EDGES else None
        if scales is not None:
            handle.row_scale, handle.col_scale = scales
            handle.csr = (indptr, indices, _own_values(values), num_cols)
            handle.planes.clear()                     # lazily built planes of the old values (separable_pays)
            return handle
        handle.row_scale = handle.col_scale = None    # a general handle from here on: the plane path of spmm_weighted
        handle.planes.clear()
        handle.csr = (indptr, indices, _own_values(values), num_cols)
        total = int(handle.blk_offsets[-1])
        if total * 512 > MASTER_PLANE_MAX_BYTES:
            handle.planes[torch.float16] = value_plane(indptr, indices, values, handle.blk_offsets, handle.num_nodes, num_cols,
                                                       dtype=torch.float16)
        else:
            handle.values32 = value_plane(indptr, indices, values, handle.blk_offsets, handle.num_nodes, num_cols)
        return handle
    if handle.edge_slot is None:
        handle.edge_slot = edge_slots(indptr, indices, handle.blk_offsets, handle.num_nodes, num_cols)
        handle.slot_duplicates = bool(values.numel()) and int(torch.unique(handle.edge_slot).numel()) != int(values.numel())
    slot = handle.edge_slot
    if handle.slot_duplicates:      # duplicate (row, col) entries ADD (fp32, rounded once): through the master, like the first build
        assert handle.values32 is not None, "duplicate entries need the fp32 master plane to add in: rebuild the handle"
        flat = handle.values32.view(-1)
        flat.zero_()
        flat.index_add_(0, slot, values.float())
        for dt in list(handle.planes):
            handle.planes[dt].copy_(handle.values32)
        handle.csr = (indptr, indices, _own_values(values), num_cols)
        return handle
    # no duplicates: every edge owns its element and every other element of the plane is a structural zero that stays one.  One scatter
    # pass per 16-bit plane the handle holds (``scatter_values_kernel``; torch's own scatter took 2.5 ms for 114.6 M edges, a rebuild
    # 107-144 ms); the fp32 master is only written when it is the ONLY plane -- otherwise it is dropped, and a plane of another 16-bit
    # type is later made from the latest values through the same map (``_plane_for``)
    from . import capi
    from .jit_kernels.spmm import _raw_stream

    values32 = _own_values(values)
    handle.csr = (indptr, indices, values32, num_cols)
    stream = _raw_stream(values32.device)
    if handle.planes:
        handle.values32 = None
        for dt in list(handle.planes):
            capi.launch_scatter_values(values32, slot, handle.planes[dt], stream)
    else:
        capi.launch_scatter_values(values32, slot, handle.values32, stream)
    return handle


def _plane_for(handle: WeightedHandle, dtype: torch.dtype) -> torch.Tensor:
    """The handle's value plane in ``dtype``, made on first use: from the fp32 master, or -- after ``update_values`` dropped it --
    from the latest values through the edge -> plane map."""
    if dtype not in handle.planes:
        if handle.values32 is not None:
            handle.planes[dtype] = handle.values32.to(dtype).contiguous()
        else:
            latest = handle.csr[2] if handle.csr is not None else None
            assert latest is not None and handle.edge_slot is not None and not handle.slot_duplicates, (
                f"this handle was built without an fp32 master (too large) and holds only the {list(handle.planes)} plane(s): pass "
                f"plane_dtype={dtype} to csr_preprocess_weighted")
            from . import capi
            from .jit_kernels.spmm import _raw_stream

            plane = torch.zeros(int(handle.blk_offsets[-1]) * 128, dtype=dtype, device=latest.device)
            capi.launch_scatter_values(latest, handle.edge_slot, plane, _raw_stream(latest.device))
            handle.planes[dtype] = plane.view(-1, 16, 8)
    return handle.planes[dtype]


def spmm_weighted(handle: WeightedHandle, feat: torch.Tensor, hash_tag: str = None, prescaled: bool = False,
                  postscale: bool = True) -> torch.Tensor:
    """``csr(values) @ feat`` -> float32 [num_nodes, F] on the current stream.  ``feat``: CUDA, 2-D; float16 / bfloat16, or
    float32 (rounded to fp16 after one power-of-two rescale, as in ``voltrix.spmm``).

    Separable handles only (round 6): ``prescaled=True`` -- the caller has already multiplied row j of ``feat`` by
    ``handle.col_scale[j]`` (e.g. in the epilogue of the GEMM that produced it): no pass over B here; ``postscale=False`` -- return
    ``A (diag(c) feat)`` WITHOUT the row factors, for callers that fold ``handle.row_scale`` into what consumes C (bias, activation):
    no pass over C here.  With both the weighted product costs exactly the binary one."""
    from .spmm.spmm import _operand

    assert isinstance(handle, WeightedHandle)
    if hash_tag is not None and getattr(handle.hspa_packed, "hash_tag", None) is None:
        handle.hspa_packed.hash_tag = hash_tag
    num_feats = feat.shape[1]
    assert handle.separable or (not prescaled and postscale), "prescaled / postscale=False: handles of separable values only"
    if handle.separable:
        if prescaled or not postscale or separable_pays(handle, num_feats, feat.element_size()):
            return _spmm_separable(handle, feat, prescaled, postscale)
    if not handle.separable and _weighted_path(handle, feat) == "csr":
        return _spmm_weighted_csr(handle, feat)
    operand, out_scale, padded, exact = _operand(feat)
    assert not exact, ("the value planes are 16-bit: VOLTRIX_FP32_MODE=exact needs the CSR kernel with values, i.e. a general handle that "
                       "kept its CSR and values (csr_preprocess_weighted(..., separable=False), at most KEEP_CSR_MAX_EDGES edges)")
    if handle.separable:
        _materialise_plane(handle, operand.dtype)
    output = torch.empty((handle.num_nodes, padded), dtype=torch.float32, device=feat.device)
    spmm_kernel(handle.blk_offsets, handle.hspa_packed, handle.hind, num_nodes=handle.num_nodes,
                num_edges=handle.num_edges, embedding_dim=padded, input=operand, output=output, out_scale=out_scale,
                values=_plane_for(handle, operand.dtype))
    return output if padded == num_feats else output[:, :num_feats].contiguous()


WEIGHTED_CSR_MAX_BLOCKS_PER_WINDOW = 48     # as for the binary operator's CSR side-car (spmm/spmm.py): short windows only
WEIGHTED_CSR_MIN_GAIN = 0.03


def _spmm_weighted_csr(handle: WeightedHandle, feat: torch.Tensor) -> torch.Tensor:
    """``csr(values) @ feat`` straight from the handle's CSR with the row-gather kernel (``spmm_csr_rows_kernel<T, 4, true>``): fp32 /
    fp16 / bf16 rows as they are, fp32 values, one fused multiply-add per element -- no plane, nothing to update when values change."""
    from . import capi
    from .jit_kernels.spmm import _raw_stream

    indptr, indices, values, _ = handle.csr
    feat = feat.contiguous()
    num_feats = feat.shape[1]
    align = 4 if feat.dtype == torch.float32 else 8
    padded = (num_feats + align - 1) // align * align
    if padded != num_feats:
        feat = torch.nn.functional.pad(feat, (0, padded - num_feats))
    output = torch.empty((handle.num_nodes, padded), dtype=torch.float32, device=feat.device)
    capi.launch_spmm_csr_rows(indptr, indices, handle.num_nodes, feat, output, _raw_stream(feat.device), 1, values=values)
    return output if padded == num_feats else output[:, :num_feats].contiguous()


def _weighted_path(handle: WeightedHandle, feat: torch.Tensor) -> str:
    """"csr" | "plane" for a general handle and this operand.  The CSR kernel with values needs the handle's CSR and latest values
    (kept up to ``KEEP_CSR_MAX_EDGES`` edges).  ``VOLTRIX_FP32_MODE=exact`` on fp32 features: "csr" (the planes are 16-bit).
    ``VOLTRIX_CSR_PATH=1 / 0``: forced / never.  Otherwise, on handles of short windows only, the first call per (width, dtype) times
    both (three calls each, one host sync, not inside a stream capture) and the faster is remembered on the handle; pinned kernels
    (``VOLTRIX_TUNE_SPACE=none / stream``) keep the plane."""
    import os

    from .jit_kernels.spmm import tune_space_mode
    from .project.const import FP32_MODE_FLAG
    from .spmm.spmm import csr_path_mode

    have = (handle.csr is not None and handle.csr[2] is not None and feat.is_cuda and feat.dim() == 2
            and feat.dtype in (torch.float32, torch.float16, torch.bfloat16) and feat.shape[0] >= handle.csr[3])
    mode = csr_path_mode()
    if not have or mode == "off":
        return "plane"
    if mode == "on" or (feat.dtype == torch.float32 and os.getenv(FP32_MODE_FLAG, "auto") == "exact"):
        return "csr"
    if tune_space_mode() in ("none", "stream") or (feat.dtype == torch.float32 and os.getenv(FP32_MODE_FLAG, "auto") == "fp16"):
        return "plane"
    windows = (handle.num_nodes + 15) // 16
    if handle.hspa_packed.numel() // 4 > WEIGHTED_CSR_MAX_BLOCKS_PER_WINDOW * windows:
        return "plane"
    key = (int(feat.shape[1]), str(feat.dtype))
    if key in handle.path_choice:
        return handle.path_choice[key]
    if torch.cuda.is_current_stream_capturing():
        return "plane"
    handle.path_choice[key] = "plane"          # what the timed calls below take
    times = {}
    for name, fn in (("plane", lambda: spmm_weighted(handle, feat)), ("csr", lambda: _spmm_weighted_csr(handle, feat))):
        fn()
        fn()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(3):
            fn()
        end.record()
        end.synchronize()
        times[name] = start.elapsed_time(end) / 3
    best = "csr" if times["csr"] < (1.0 - WEIGHTED_CSR_MIN_GAIN) * times["plane"] else "plane"
    if os.getenv("VOLTRIX_PRINT_AUTO_TUNE") or os.getenv("VOLTRIX_JIT_DEBUG"):
        print(f"voltrix.spmm_weighted path for width {key[0]} {key[1]}: {times} -> {best}")
    handle.path_choice[key] = best
    return best


def separable_pays(handle: WeightedHandle, num_feats: int, elem_bytes: int) -> bool:
    """The two row scalings of the separable path move B twice and C twice -- 2 N F (s_in + 4) bytes --, the value plane 256 bytes per TC
    block whatever the width.  On graphs whose B and C are far beyond the caches and whose windows are short the passes cost more than
    the plane (papers-like x 128: 94 ms separable against 80.5 with the plane; the binary product is 62): the separable path is taken
    when the handle carries the two-level side-car (the panel kernel has no value plane: reddit-like 1.43 against 2.47 ms) or when its
    extra bytes are fewer.  Host integers only."""
    from . import sidecar

    if handle.csr is None or sidecar.lookup(handle.hspa_packed)[1] is not None:
        return True
    rows = max(handle.num_nodes, handle.col_scale.numel())
    return 2 * rows * num_feats * (elem_bytes + 4) <= 64 * handle.hspa_packed.numel()      # 256 bytes per TC block = 64 x 4 words


def _materialise_plane(handle: WeightedHandle, dtype: torch.dtype) -> None:
    """A separable handle that is better off with a value plane for this call's width builds it now, once per 16-bit type (chunk by
    chunk, no fp32 master; values restated from the factors when the caller gave those)."""
    if dtype in handle.planes:
        return
    indptr, indices, values, num_cols = handle.csr
    if values is None:
        deg = (indptr[1:] - indptr[:-1]).long()
        values = torch.repeat_interleave(handle.row_scale, deg) * handle.col_scale[indices.long()]
    universe = max(num_cols, int(indices.max()) + 1) if indices.numel() else num_cols
    handle.planes[dtype] = value_plane(indptr, indices, values, handle.blk_offsets, handle.num_nodes, universe, dtype=dtype)


def _spmm_separable(handle: WeightedHandle, feat: torch.Tensor, prescaled: bool = False, postscale: bool = True) -> torch.Tensor:
    """``diag(r) (A (diag(c) feat))``: rows of B times c (one HIP pass, in B's own dtype -- fp32 features are scaled in fp32 and
    then take ``voltrix.spmm``'s own fp32 path), the BINARY operator on the handle (two-level side-car, tuned tiles, launch
    plans), rows of C times r in place (one HIP pass)."""
    from . import capi
    from .jit_kernels.spmm import _raw_stream
    from .spmm.spmm import spmm

    assert feat.is_cuda and feat.dim() == 2 and feat.shape[0] == handle.col_scale.numel()
    feat = feat.contiguous()
    num_feats = feat.shape[1]
    align = 4 if feat.dtype == torch.float32 else 8
    padded = (num_feats + align - 1) // align * align
    if padded != num_feats:
        feat = torch.nn.functional.pad(feat, (0, padded - num_feats))
    stream = _raw_stream(feat.device)
    scaled = feat
    if not prescaled:
        scaled = torch.empty_like(feat)
        capi.launch_scale_rows(feat, handle.col_scale, scaled, stream)
    out = spmm(handle.blk_offsets, handle.hspa_packed, handle.hind, num_nodes=handle.num_nodes, num_edges=handle.num_edges,
               feat=scaled)
    if postscale:
        capi.launch_scale_rows(out, handle.row_scale, out, stream)
    return out if padded == num_feats else out[:, :num_feats].contiguous()


def scale_rows_of(feat: torch.Tensor, scale: torch.Tensor, in_place: bool = False) -> torch.Tensor:
    """``feat[i, :] * scale[i]`` with the library's ``scale_rows`` pass (one HBM pass in ``feat``'s dtype; widths that are not a
    16-byte multiple are padded for the pass and cut again).  ``in_place``: overwrite ``feat`` when its layout allows."""
    from . import capi
    from .jit_kernels.spmm import _raw_stream

    assert feat.is_cuda and feat.dim() == 2 and scale.numel() == feat.shape[0] and scale.dtype == torch.float32
    num_feats = feat.shape[1]
    align = 4 if feat.dtype == torch.float32 else 8
    padded = (num_feats + align - 1) // align * align
    src = feat.contiguous()
    if padded != num_feats:
        src = torch.nn.functional.pad(src, (0, padded - num_feats))
    if src.data_ptr() % 16:
        src = src.clone()
    dst = src if (in_place or src is not feat) else torch.empty_like(src)
    capi.launch_scale_rows(src, scale.contiguous(), dst, _raw_stream(feat.device))
    return dst if padded == num_feats else dst[:, :num_feats].contiguous()


def transpose_order(indptr: torch.Tensor, indices: torch.Tensor, num_rows: int) -> torch.Tensor:
    """int64 [nnz]: entry k of the CSR of ``A^T`` (``transpose_weighted``) is entry ``order[k]`` of the CSR of ``A``."""
    deg = (indptr[1:] - indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(num_rows, device=indptr.device, dtype=torch.int64), deg)
    return torch.argsort(indices.long() * num_rows + rows)


def transpose_weighted(indptr: torch.Tensor, indices: torch.Tensor, values: torch.Tensor, num_rows: int, num_cols: int):
    """Device CSR with values of ``A`` -> ``(t_indptr, t_indices, t_values)`` of ``A^T`` (rows sorted; one stable sort by column)."""
    dev = indptr.device
    deg = (indptr[1:] - indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(num_rows, device=dev, dtype=torch.int64), deg)
    order = torch.argsort(indices.long() * num_rows + rows)
    t_indptr = torch.zeros(num_cols + 1, dtype=torch.int64, device=dev)
    t_indptr[1:] = torch.cumsum(torch.bincount(indices.long(), minlength=num_cols), 0)
    return t_indptr.to(torch.int32), rows[order].to(torch.int32), values[order].contiguous()
